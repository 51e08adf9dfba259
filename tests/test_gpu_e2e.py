"""-m gpu: the product models (HIP warps + MIOpen convs) against the reference's end-to-end golden
values and against the CPU oracle on identical weights and inputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# north_star: flow fields within 1e-4 px (fp32), interp PSNR within 0.01 dB of the reference path
FLOW_ATOL = 1e-4
PSNR_TOL_DB = 0.01


def _product(nd):
    if nd == 3:
        from opticalflowscivis_amd.flow3d.model.RIFE import Model
    else:
        from opticalflowscivis_amd.flow2d.model.RIFE import Model
    torch.manual_seed(1234)
    return Model(local_rank=-1, device=DEV)


def _psums(net):
    return np.array([float(p.detach().double().sum()) for p in net.parameters()])


@pytest.mark.parametrize("nd", [3, 2])
def test_product_matches_reference_golden(golden, nd):
    """Same seed => the reference's initial weights; same inputs => its outputs and losses."""
    from opticalflowscivis_amd.data.synthetic import psnr
    g = golden("flow%dd_e2e" % nd)
    m = _product(nd)
    np.testing.assert_allclose(_psums(m.flownet), g["param_sums"], rtol=0, atol=1e-6)
    data = torch.from_numpy(g["data"]).to(DEV)
    imgs, gt = data[:, :2], data[:, 2:3]
    m.eval()
    with torch.no_grad():
        merged, flows, mask = m.inference(imgs[:, :1], imgs[:, 1:2])
    if nd == 2:
        merged, mask = merged[2], mask[2]
    assert np.abs(flows[2].cpu().numpy() - g["inf_flow2"]).max() < FLOW_ATOL
    assert np.abs(merged.cpu().numpy() - g["inf_merged"]).max() < 5e-5
    ref_psnr = psnr(torch.from_numpy(g["inf_merged"]), gt.cpu())
    assert abs(psnr(merged.cpu(), gt.cpu()) - ref_psnr) < PSNR_TOL_DB
    keys = ("loss_l1", "loss_tea", "loss_distill", "loss_G") if nd == 3 else \
        [str(k) for k in g["update_loss_keys"]]
    for step in range(2):
        if nd == 3:
            pred, info = m.update(imgs, gt, learning_rate=1e-4, training=True)
        else:
            pred, info = m.update(imgs, gt, "droplet2d", learning_rate=1e-4, training=True)
        got = [float(info[k].detach()) if torch.is_tensor(info[k]) else float(info[k]) for k in keys]
        np.testing.assert_allclose(got, g["update_losses"][step], rtol=5e-4, atol=2e-6)
    assert np.abs(pred.detach().cpu().numpy() - g["update_pred_last"]).max() < 5e-4
    np.testing.assert_allclose(_psums(m.flownet), g["param_sums_after"], rtol=1e-4, atol=5e-3)


def test_flow3d_vs_oracle_droplet():
    """Droplet-like binary volumes, same weights on both sides.  48^3 is the regular path; 40^3 is
    not a multiple of 16, so block outputs (32^3) are smaller than the frames and the reference's
    crop logic plus the input-extent != flow-extent warp are exercised."""
    from opticalflowscivis_amd.data import synthetic
    from oracle.ifnet_ref import ModelRef
    for S in (48, 40):
        m = _product(3)
        torch.manual_seed(1234)
        o = ModelRef(3)
        o.flownet.load_state_dict({k: v.cpu() for k, v in m.flownet.state_dict().items()})
        data = synthetic.droplet3d_batch(1, S, seed=3)
        imgs, gt = data[:, :2], data[:, 2:3]
        with torch.no_grad():
            om, of, _ = o.inference(imgs[:, :1], imgs[:, 1:2])
            pm, pf, _ = m.inference(imgs[:, :1].to(DEV), imgs[:, 1:2].to(DEV))
        assert pm.shape == om.shape
        assert float((pf[2].cpu() - of[2]).abs().max()) < FLOW_ATOL
        gtc = gt[(slice(None), slice(None)) + tuple(slice(0, s) for s in om.shape[2:])]
        assert abs(synthetic.psnr(pm.cpu(), gtc) - synthetic.psnr(om, gtc)) < PSNR_TOL_DB
        _, oi = o.update(imgs, gt, learning_rate=1e-4)
        _, pi = m.update(imgs.to(DEV), gt.to(DEV), learning_rate=1e-4)
        for k in ("loss_l1", "loss_tea", "loss_distill", "loss_G"):
            assert abs(float(pi[k]) - float(oi[k])) < 5e-4 * max(1.0, abs(float(oi[k]))), k


def test_upflow_matches_reference_golden(golden):
    """UPFlow_net (HIP correlation / warps / census / photo losses, MIOpen convs) against the
    reference's forward + backward on identical weights (same seed) and inputs."""
    from opticalflowscivis_amd.upflow.model.upflow import UPFlow_net
    g = golden("upflow_e2e")
    conf = UPFlow_net.config()
    conf.update({'if_norm_before_cost_volume': True, 'norm_moments_across_channels': False,
                 'norm_moments_across_images': False, 'photo_loss_census_weight': 1,
                 'multi_scale_distillation_weight': 1})
    torch.manual_seed(0)
    net = conf()
    np.testing.assert_allclose(_psums(net), g["param_sums"], rtol=0, atol=1e-9)
    net = net.to(DEV)
    out = net({'im1': torch.from_numpy(g["im1"]), 'im2': torch.from_numpy(g["im2"]), 'if_loss': True})
    ref_f = torch.from_numpy(g["flow_f_out"])
    scale = float(ref_f.abs().max())
    # UPFlow is chaotic at fp32 level: WarpingLayer_no_div zeroes a feature pixel when its fp32 weight
    # sum is < 1.0, which on ~1-2 % of in-bounds pixels is decided by the last ulp, at
    # each of 4 pyramid levels.  The reference's OWN torch ops run on this GPU deviate from its CPU
    # result by median 0.20 px / max 2.0 px at a flow scale of 68 px; the HIP path lands in the same band.
    # The band also has to hold MIOpen's run-to-run noise: its 2-D convolutions are not reproducible from one call
    # to the next (the feature pyramid of the SAME input differs by ~1e-6 between two calls in one process; two
    # forwards of this net in one process differ by up to 4.8 px), and every flipped mask pixel amplifies that.
    # Measured over 26 runs on two boxes: median error 0.30-1.02 % of the flow scale, 99th percentile 1.6-4.2 %,
    # losses within 6e-4 .. 7e-3, occlusion-mask mismatch 0.3-0.8 %, gradient sums median 0.3-1.2 % / max 1.2-4.8 %.
    # The limits below are about twice the worst of those; the ops themselves are pinned tightly in
    # test_gpu_warps / test_gpu_losses and level by level in test_upflow_levels_teacher_forced.
    err = (out['flow_f_out'].detach().cpu() - ref_f).abs()
    assert float(err.median()) < 0.02 * scale
    assert float(err.flatten().quantile(0.99)) < 0.08 * scale
    occ_ref = torch.from_numpy(g["occ_fw"])
    assert float((out['occ_fw'].cpu() != occ_ref).float().mean()) < 2e-2
    keys = [str(k) for k in g["loss_keys"]]
    got = np.array([float(out['loss_dict'][k]) for k in keys])
    np.testing.assert_allclose(got, g["losses"], rtol=1.5e-2)
    sum(out['loss_dict'][k] for k in keys).backward()
    gsum = np.array([float(p.grad.detach().double().abs().sum()) if p.grad is not None else 0.0
                     for p in net.parameters()])
    # gradient magnitude per parameter tensor: these sums move with every flipped mask / occlusion pixel, so the bound
    # is tied to the fraction of occlusion-mask pixels that differ from the reference's (measured: 0.3-0.6 % of the
    # pixels, median deviation 0.3-0.9 %, worst tensor 1.9-3.5 %; the stock torch ops on this GPU: 0.4-0.6 % / 1.4-2.5 %)
    occ_diff = float((out['occ_fw'].cpu() != occ_ref).float().mean())
    rel = np.abs(gsum - g["grad_abs_sums"]) / (np.abs(g["grad_abs_sums"]) + 1e-3)
    assert np.median(rel) < 0.02 + 3 * occ_diff and rel.max() < 0.03 + 10 * occ_diff, (np.median(rel), rel.max(), occ_diff)


def _upflow_deviation(g, stock, sgu=False):
    """One forward + backward of the UPFlow mirror on this GPU against the reference's CPU golden: with the HIP ops, or
    (stock=True) with every hot-path op swapped for the reference's own formulation in stock torch ops on the GPU
    (grid_sample, unfold correlation, 49-channel census: oracle/upflow_port.py::stock_ops)."""
    import contextlib
    from opticalflowscivis_amd.upflow.model.upflow import UPFlow_net
    from oracle.upflow_port import stock_ops
    conf = UPFlow_net.config()
    conf.update({'if_norm_before_cost_volume': True, 'norm_moments_across_channels': False,
                 'norm_moments_across_images': False, 'photo_loss_census_weight': 1,
                 'multi_scale_distillation_weight': 1, 'if_sgu_upsample': sgu})
    torch.manual_seed(0)
    net = conf().to(DEV)
    keys = [str(k) for k in g["loss_keys"]]
    with (stock_ops() if stock else contextlib.nullcontext()):
        out = net({'im1': torch.from_numpy(g["im1"]), 'im2': torch.from_numpy(g["im2"]), 'if_loss': True})
        got = np.array([float(out['loss_dict'][k].detach()) for k in keys])
        sum(out['loss_dict'][k] for k in keys).backward()
    ref_f = torch.from_numpy(g["flow_f_out"])
    scale = float(ref_f.abs().max())
    err = (out['flow_f_out'].detach().cpu() - ref_f).abs()
    gsum = np.array([float(p.grad.detach().double().abs().sum()) if p.grad is not None else 0.0
                     for p in net.parameters()])
    rel = np.abs(gsum - g["grad_abs_sums"]) / (np.abs(g["grad_abs_sums"]) + 1e-3)
    return dict(flow_median=float(err.median()) / scale, flow_p99=float(err.flatten().quantile(0.99)) / scale,
                loss=float(np.max(np.abs(got - g["losses"]) / np.abs(g["losses"]))),
                occ=float((out['occ_fw'].cpu() != torch.from_numpy(g["occ_fw"])).float().mean()),
                grad_median=float(np.median(rel)), grad_max=float(rel.max()))


def test_upflow_hip_path_is_no_further_from_the_reference_than_its_own_torch_ops(golden):
    """the comparator of the UPFlow end-to-end band, measured in the test instead of quoted in a
    comment.  The reference's own op formulations, run as stock torch ops on THIS GPU, do not reproduce its CPU
    result either (MIOpen's 2-D convolutions are not reproducible from call to call; fp32-borderline validity masks
    flip): three runs of each side, and for every metric the HIP path's best run must be within 1.5 x the stock path's
    worst -- the HIP kernels add no deviation of their own beyond what the platform's noise already gives the
    reference's ops.  (Measured, 4 runs each: flow median 0.28-0.45 % vs 0.30-0.60 % of the flow scale, 99th
    percentile 1.7-3.8 % vs 1.6-3.8 %, losses 0.05-0.42 % vs 0.07-0.29 %, gradient sums worst tensor 1.9-3.5 % vs
    1.4-2.5 %.  `torch.backends.cudnn.deterministic = True` is not an option: MIOpen answers
    miopenStatusUnknownError for one of the PWC shapes in that mode on this image.)  The epsilon-level statement about
    the same network lives on the CPU: tests/test_oracle_e2e.py::test_upflow_mirror_on_oracle_ops_matches_reference
    (the mirror on the oracle's ops equals the reference bit for bit), and op by op in test_gpu_warps / test_gpu_losses
    / test_gpu_c3_batch."""
    g = golden("upflow_e2e")
    stock = [_upflow_deviation(g, True) for _ in range(3)]
    hip = [_upflow_deviation(g, False) for _ in range(3)]
    for k in stock[0]:
        s_worst, h_best = max(r[k] for r in stock), min(r[k] for r in hip)
        print("%-12s stock %s | hip %s" % (k, " ".join("%.2e" % r[k] for r in stock), " ".join("%.2e" % r[k] for r in hip)))
        assert h_best <= 1.5 * s_worst + 1e-4, (k, stock, hip)
    # and every HIP run stays inside the absolute band of test_upflow_matches_reference_golden
    for r in hip:
        assert r["flow_median"] < 0.02 and r["flow_p99"] < 0.08 and r["loss"] < 1.5e-2 and r["occ"] < 2e-2


def test_upflow_sgu_variant_vs_reference_and_its_own_torch_ops(golden):
    """The self-guided upsampling variant (`if_sgu_upsample=True`; UPFlow/model/upflow.py:21-92, 612-616, 629-631) on the HIP
    ops against the reference's forward + backward (tests/golden/upflow_sgu.npz, 96 x 128, flow scale 12 px).  The variant
    adds ten masked feature warps and flow warps to the chain that amplifies fp32-borderline validity decisions, and the
    fixture's flow scale is small, so the absolute band of the plain test does not transfer; the comparator does: three
    runs of the reference's own op formulations as stock torch ops on this GPU vs three runs on the HIP ops, every metric's
    best HIP run within 1.5 x the stock path's worst (measured, 3 runs each: flow median 1.1-2.2 % stock vs 1.2-2.0 % HIP
    of the 12 px scale, 99th percentile 7.2-9.8 % vs 7.9-11.4 %, losses 0.35-1.6 % vs 0.8-1.3 %, occlusion mismatch
    0.7-1.1 % both, gradient sums worst tensor 12-15 % vs 9-15 %).  The epsilon statement is on the CPU:
    tests/test_oracle_e2e.py::test_upflow_sgu_mirror_on_oracle_ops_matches_reference (bit for bit)."""
    g = golden("upflow_sgu")
    stock = [_upflow_deviation(g, True, sgu=True) for _ in range(3)]
    hip = [_upflow_deviation(g, False, sgu=True) for _ in range(3)]
    for k in stock[0]:
        s_worst, h_best = max(r[k] for r in stock), min(r[k] for r in hip)
        print("%-12s stock %s | hip %s" % (k, " ".join("%.2e" % r[k] for r in stock), " ".join("%.2e" % r[k] for r in hip)))
        assert h_best <= 1.5 * s_worst + 1e-4, (k, stock, hip)
    for r in hip:  # and nothing is grossly off in absolute terms: 0.6 px median / 3 px at the 99th percentile, losses 3 %
        assert r["flow_median"] < 0.05 and r["flow_p99"] < 0.25 and r["loss"] < 3e-2 and r["occ"] < 4e-2


def _proj(t, seed):
    """(dot with a seeded Gaussian tensor, absolute sum): the two numbers tests/golden/make_golden.py::_proj stores."""
    if t is None:
        return np.array([0.0, 0.0])
    t = t.detach().double().cpu()
    R = torch.randn(t.shape, generator=torch.Generator().manual_seed(seed))
    return np.array([float((t * R.double()).sum()), float(t.abs().sum())])


def test_upflow_levels_teacher_forced(golden):
    """UPFlow end to end is a band, not an epsilon (fp32-borderline validity masks flip at every level and the
    difference is amplified down the pyramid) -- so every pyramid level is ALSO checked on its own, fed the
    reference's inputs of that level (tests/golden/upflow_levels.npz, captured by running the reference):
    `decode_level_res` outputs at 1e-4, the gradients w.r.t. every level input and every parameter of the
    estimator / context networks at 5e-3 (projections: dot with a seeded Gaussian tensor + absolute sum; both agree
    to ~3e-5 with MIOpen on its direct / implicit-GEMM backward solvers for the estimator's 2-D convolutions, which
    tests/conftest.py pins -- its fp32 Winograd solvers, picked or not from one run to the next by the find step's
    timings, move single bias-gradient entries by ~1e-3, measured up to 5e-3 on the projection scale).
    The forward value of the two feature warps is forced to the reference's (their mask decisions included); the
    HIP warp itself is compared with it away from pixels where the two masks disagree, and its own gradient is
    what flows back."""
    from opticalflowscivis_amd.upflow.model.upflow import UPFlow_net
    g = golden("upflow_levels")
    conf = UPFlow_net.config()
    conf.update({'if_norm_before_cost_volume': True, 'norm_moments_across_channels': False,
                 'norm_moments_across_images': False, 'photo_loss_census_weight': 1,
                 'multi_scale_distillation_weight': 1})
    torch.manual_seed(0)
    net = conf()
    np.testing.assert_allclose(_psums(net), g["param_sums"], rtol=0, atol=1e-9)
    net = net.to(DEV)
    params = dict(net.named_parameters())
    pnames = [str(n) for n in g["param_names"]]
    hip_warp = net.warping_layer.forward
    names = ["flow_1", "flow_2", "feature_1", "feature_1_1x1", "feature_2", "feature_2_1x1"]
    for level in range(int(g["nlevels"])):
        tag = "L%d_" % level
        ins = [torch.from_numpy(g[tag + n]).to(DEV).requires_grad_() for n in names]
        forced = ([torch.from_numpy(g[tag + "feature_2_warp"]).to(DEV), torch.from_numpy(g[tag + "feature_1_warp"]).to(DEV)]
                  if level > 0 else [])
        calls, flipped = [], []

        def warp(x, flow):
            out = hip_warp(x, flow)
            ref = forced[len(calls)]
            calls.append(1)
            # pixels where the two validity masks disagree (one side zeroed the whole channel vector)
            mh, mr = (out.detach().abs().sum(1) > 0), (ref.abs().sum(1) > 0)
            agree = (mh == mr)
            flipped.append(float((~agree).float().mean()))
            err = ((out.detach() - ref).abs() * agree.unsqueeze(1)).max()
            assert float(err) < 2e-5 * max(1.0, float(ref.abs().max())), (level, float(err))
            return out + (ref - out).detach()  # forward: the reference's value; backward: the HIP warp's gradient

        net.warping_layer.forward = warp
        try:
            outs = net.decode_level_res(level, *ins)
        finally:
            net.warping_layer.forward = hip_warp
        assert all(f < 0.03 for f in flipped), (level, flipped)  # fp32-borderline pixels only
        print("level %d: validity-mask pixels that differ from the reference's: %s" % (level, flipped))
        for n, t in zip(["flow_1_up", "flow_2_up", "res_1", "res_2"], outs):
            ref = torch.from_numpy(g[tag + "out_" + n])
            err = float((t.detach().cpu() - ref).abs().max())
            assert err < 1e-4 * max(1.0, float(ref.abs().max())), (level, n, err)
        G1 = torch.randn(outs[2].shape, generator=torch.Generator().manual_seed(100 + level)).to(DEV)
        G2 = torch.randn(outs[3].shape, generator=torch.Generator().manual_seed(200 + level)).to(DEV)
        wrt = ins + [params[n] for n in pnames]
        grads = torch.autograd.grad((outs[2] * G1).sum() + (outs[3] * G2).sum(), wrt, allow_unused=True)
        exact = not any(flipped)

        def check(got, want, what, tol):
            # the dot product of an error vector e with a Gaussian tensor is ~ |e|_2 <= |e|_1
            assert abs(got[1] - want[1]) <= tol * max(want[1], 1e-6), (level, what, got, want)
            assert abs(got[0] - want[0]) <= tol * max(abs(want[0]), 0.05 * want[1], 1e-6), (level, what, got, want)

        for i, n in enumerate(names):
            # inputs that reach the outputs through the warp too see the HIP mask where it differs from the
            # reference's: exact only when no mask pixel flipped at this level
            on_warp_path = level > 0 and n in ("flow_1", "flow_2", "feature_1", "feature_2")
            check(_proj(grads[i], 300 + 10 * level + i), g[tag + "gin"][i], n,
                  5e-3 if (exact or not on_warp_path) else 6e-2)
        for i, n in enumerate(pnames):
            check(_proj(grads[6 + i], 1000 + i), g[tag + "gparam"][i], n, 5e-3)


def test_upflow_c3_b32_equals_its_b2_slices():
    """BASELINE config C3 at its full batch (32 x 3 x 150 x 450, census on): samples are independent through every
    op of the forward pass (per-plane feature normalisation), so the B = 32 launch geometry of every HIP kernel must
    reproduce what the same network computes on the B = 2 slices of the batch -- the size at which the golden
    end-to-end and teacher-forced tests pin it to the reference.  Losses are batch means: the B = 32 value is the
    mean of the 16 slice values.  Same band as the end-to-end test (MIOpen picks batch-dependent convolution
    algorithms and is not reproducible call to call; validity masks flip on fp32 noise)."""
    from opticalflowscivis_amd.data import synthetic
    from opticalflowscivis_amd.upflow.model.upflow import UPFlow_net
    conf = UPFlow_net.config()
    conf.update({'if_norm_before_cost_volume': True, 'norm_moments_across_channels': False,
                 'norm_moments_across_images': False, 'photo_loss_census_weight': 1,
                 'multi_scale_distillation_weight': 1})
    torch.manual_seed(0)
    net = conf().to(DEV)
    pairs = synthetic.vortex2d_pairs(32, 150, 450, seed=0, device=DEV)
    im1, im2 = pairs[:, 0].contiguous(), pairs[:, 1].contiguous()
    keys = ['photo_loss', 'smooth_loss', 'census_loss', 'msd_loss']
    with torch.no_grad():
        full = net({'im1': im1, 'im2': im2, 'if_loss': True})
        flow32 = full['flow_f_out']
        loss32 = np.array([float(full['loss_dict'][k]) for k in keys])
        assert np.isfinite(loss32).all() and bool(torch.isfinite(flow32).all())
        acc = np.zeros(4)
        scale = float(flow32.abs().max())
        for i in range(0, 32, 2):
            part = net({'im1': im1[i:i + 2], 'im2': im2[i:i + 2], 'if_loss': True})
            acc += np.array([float(part['loss_dict'][k]) for k in keys])
            err = (part['flow_f_out'] - flow32[i:i + 2]).abs()
            assert float(err.median()) < 0.02 * scale and float(err.flatten().quantile(0.99)) < 0.08 * scale, i
    np.testing.assert_allclose(loss32, acc / 16, rtol=1.5e-2)


def test_upflow_c3_train_step_runs():
    """BASELINE config C3 shape (150 x 450, census on) through one optimiser step, B=4."""
    from opticalflowscivis_amd.upflow.scripts.simple_train import Trainer
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        conf = Trainer.Config(n_epoch=1, batchsize=4, batch_per_epoch=2, exp_dir=d)
        conf.net_params = dict(conf.net_params, photo_loss_census_weight=1, multi_scale_distillation_weight=1)
        torch.manual_seed(0)
        tr = Trainer(conf, device=DEV)
        before = _psums(tr.net)
        tr.training()
        after = _psums(tr.net)
        assert np.isfinite(after).all() and not np.allclose(before, after)


def test_flow2d_c2_train_step_vs_oracle():
    """BASELINE config C2: Flow-2D Droplet 160 x 224, batch 16, one unsupervised train step (2-D
    warp-pair + photometric HIP kernels) against the CPU oracle on identical weights and data."""
    from opticalflowscivis_amd.data import synthetic
    from oracle.ifnet_ref import ModelRef
    m = _product(2)
    torch.manual_seed(1234)
    o = ModelRef(2)
    o.flownet.load_state_dict({k: v.cpu() for k, v in m.flownet.state_dict().items()})
    data = synthetic.droplet2d_batch(16, 160, 224, seed=1234)
    imgs, gt = data[:, :2], data[:, 2:3]
    po, oi = o.update(imgs, gt, learning_rate=1e-4)
    pp, pi = m.update(imgs.to(DEV), gt.to(DEV), "droplet2d", learning_rate=1e-4)
    for k in ("loss_l1", "loss_tea", "loss_distill", "loss_photo", "loss_G"):
        a, b = float(pi[k].detach()), float(oi[k].detach())
        assert abs(a - b) < 1e-3 * max(1e-3, abs(b)), (k, a, b)
    assert float((pi["flow"].detach().cpu() - oi["flow"][:, :2].detach()).abs().max()) < 1e-3
    assert abs(synthetic.psnr(pp.detach().cpu(), gt) - synthetic.psnr(po.detach(), gt)) < PSNR_TOL_DB
    np.testing.assert_allclose(_psums(m.flownet), _psums(o.flownet), rtol=1e-4, atol=5e-3)


def test_flow3d_train_step_is_hip_graph_capturable():
    """The whole Flow-3D training step (forward, losses, backward, AdamW) captures into ONE HIP graph
    (`Model.graphed_update`): no entry point allocates outside the caching allocator, synchronises or
    copies from the host.  Replaying the graph reproduces eager steps from the same weights, follows the
    learning rate passed per step, and building it leaves the model untouched."""
    import copy
    from opticalflowscivis_amd.flow3d.model.RIFE import Model
    from opticalflowscivis_amd.data import synthetic
    torch.manual_seed(7)
    m = Model(local_rank=-1, device=DEV)
    data = synthetic.droplet3d_batch(1, 32, seed=3, device=DEV)
    imgs, gt = data[:, :2].contiguous(), data[:, 2:3].contiguous()
    data2 = synthetic.droplet3d_batch(1, 32, seed=4, device=DEV)
    imgs2, gt2 = data2[:, :2].contiguous(), data2[:, 2:3].contiguous()
    state0 = copy.deepcopy(m.flownet.state_dict())
    lrs = (1e-4, 3e-4, 1e-4)
    batches = ((imgs, gt), (imgs2, gt2), (imgs, gt))  # the third loss sees the effect of the second update
    eager = [float(m.update(bi, bg, learning_rate=lr, training=True)[1]["loss_G"].detach())
             for (bi, bg), lr in zip(batches, lrs)]
    # fresh model state, graphed
    m2 = Model(local_rank=-1, device=DEV)
    m2.flownet.load_state_dict(state0)
    step = m2.graphed_update(imgs, gt)
    for a, b in zip(m2.flownet.state_dict().values(), state0.values()):
        assert torch.equal(a, b)  # building the graph did not train
    graphed = []
    for (bi, bg), lr in zip(batches, lrs):
        _, info = step(bi, bg, lr)
        graphed.append(float(info["loss_G"].detach()))
    for a, b in zip(graphed, eager):
        assert abs(a - b) < 2e-4 * max(1.0, abs(b)), (graphed, eager)
    # AdamW's first steps move every weight by ~lr * sign(grad): the replays honoured the three learning
    # rates (total displacement ~ their sum; elementwise agreement with the eager run is not defined --
    # most of these gradients are at the noise level of the atomics-based reductions and Adam normalises
    # them to +-lr)
    tot = sum(lrs)
    w0 = state0["block2.convblock0.0.0.weight"].to(DEV)
    wg = dict(m2.flownet.named_parameters())["block2.convblock0.0.0.weight"].detach()
    assert 0.3 * tot < float((wg - w0).abs().median()) < 1.2 * tot
    assert float((wg - w0).abs().max()) <= tot * 1.001 + 1e-7


def test_flow2d_train_step_is_hip_graph_capturable():
    """Flow-2D's NaN / > 10 guard on the distillation loss (Flow-2D/model/RIFE.py:295-296) runs
    on the device, so the launch-bound C2 step (~600 launches) captures into one HIP graph; replays reproduce the
    eager losses step by step."""
    import copy
    from opticalflowscivis_amd.flow2d.model.RIFE import Model
    from opticalflowscivis_amd.data import synthetic
    torch.manual_seed(7)
    m = Model(local_rank=-1, device=DEV)
    d1 = synthetic.droplet2d_batch(4, 64, 96, seed=3, radius=(8, 16)).to(DEV)
    d2 = synthetic.droplet2d_batch(4, 64, 96, seed=4, radius=(8, 16)).to(DEV)
    batches = [(d[:, :2].contiguous(), d[:, 2:3].contiguous()) for d in (d1, d2, d1)]
    lrs = (1e-4, 3e-4, 1e-4)
    state0 = copy.deepcopy(m.flownet.state_dict())
    keys = ("loss_l1", "loss_tea", "loss_distill", "loss_photo", "loss_G")
    eager = [[float(v[k].detach()) for k in keys]
             for v in (m.update(bi, bg, "droplet2d", learning_rate=lr, training=True)[1]
                       for (bi, bg), lr in zip(batches, lrs))]
    m2 = Model(local_rank=-1, device=DEV)
    m2.flownet.load_state_dict(state0)
    step = m2.graphed_update(batches[0][0], batches[0][1], dataset="droplet2d")
    for a, b in zip(m2.flownet.state_dict().values(), state0.values()):
        assert torch.equal(a, b)
    for i, ((bi, bg), lr) in enumerate(zip(batches, lrs)):
        _, info = step(bi, bg, lr)
        got = [float(info[k].detach()) for k in keys]
        for k, a, b in zip(keys, got, eager[i]):
            assert abs(a - b) < 5e-4 * max(abs(b), 1e-6) + 1e-7, (i, k, got, eager[i])


def test_flow2d_distill_guard_on_device():
    """RIFE.py:295-296 on the device: a distillation loss that is NaN or > 10 leaves loss_G and yields a gradient
    of exactly 0 into the flows, also through non-finite operands; a regular one passes through untouched."""
    from opticalflowscivis_amd import ops
    g = torch.Generator().manual_seed(2)
    shape, fshape = (2, 1, 24, 40), (2, 4, 24, 40)
    gt = torch.rand(shape, generator=g).to(DEV)
    mt = (gt + 0.01 * torch.randn(shape, generator=g).to(DEV))
    ms = [(gt + 0.3 * torch.randn(shape, generator=g).to(DEV)) for _ in range(3)]
    ft = torch.randn(fshape, generator=g).to(DEV)
    for scale, bad in ((1.0, False), (1e3, True), (float("nan"), True)):
        fl = [(scale * torch.randn(fshape, generator=g).to(DEV)).requires_grad_() for _ in range(3)]
        ld = ops.distill_terms3(ms, mt, gt, fl, ft)
        discard = torch.isnan(ld) | (ld > 10.)
        assert bool(discard) == bad
        kept = torch.where(discard, torch.zeros_like(ld), ld)
        grads = torch.autograd.grad(kept * 0.01, fl)
        if bad:
            assert float(kept) == 0.0 and all(bool((x == 0).all()) for x in grads)
        else:
            assert float(kept) == float(ld) and any(float(x.abs().max()) > 0 for x in grads)


def test_flow3d_training_trajectory_tracks_oracle():
    """25 AdamW steps from the same weights on the same batch: the product's losses follow the CPU oracle's
    step by step while loss_G falls (tests/tools/soak_vs_oracle.py follows them for 100+ steps, to 1/4 of the
    initial loss).  A wrong gradient in any kernel of the step shows up here within a few steps."""
    from opticalflowscivis_amd.data import synthetic
    from oracle.ifnet_ref import ModelRef
    torch.manual_seed(0)
    m = _product(3)
    o = ModelRef(3)
    o.flownet.load_state_dict({k: v.cpu() for k, v in m.flownet.state_dict().items()})
    data = synthetic.droplet3d_batch(2, 32, seed=5)
    imgs, gt = data[:, :2].contiguous(), data[:, 2:3].contiguous()
    gi, gg = imgs.to(DEV), gt.to(DEV)
    first = last = None
    for i in range(26):
        _, pi = m.update(gi, gg, learning_rate=3e-5, training=True)
        _, oi = o.update(imgs, gt, learning_rate=3e-5)
        a, b = float(pi["loss_G"].detach()), float(oi["loss_G"].detach())
        assert abs(a - b) < 2e-3 * abs(b) + 1e-5, (i, a, b)
        if i == 0:
            first = b
        last = b
    assert last < 0.97 * first  # it actually trained
