"""-m gpu: parity at (and towards) the bench scale -- VERDICT r1 "next round" item 1.

* the full Flow-3D train step, product vs CPU oracle, at 128^3 (the size bench.py's cpu_baseline / parity
  witness runs at) and on the C5 workload (5Jets-like density, 64^3);
* the IFNet-3D convolution kernels at the REAL layer shapes of the 2 x 256^3 step (conv0a 11 -> 32 with
  128^3 out, the flow head's last deconv 32 -> 6 with 256^3 out) against MIOpen and against directly
  evaluated weight-gradient taps;
* Flow-2D at an extent where IFNet crops its outputs below the input (rife2d_photometric).
north_star tolerances: losses 5e-4 relative, flow 1e-4 px, interpolation PSNR 0.01 dB."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _pair(nd, seed=1234):
    """(product model, oracle model) with identical seed-`seed` weights."""
    from oracle.ifnet_ref import ModelRef
    if nd == 3:
        from opticalflowscivis_amd.flow3d.model.RIFE import Model
    else:
        from opticalflowscivis_amd.flow2d.model.RIFE import Model
    torch.manual_seed(seed)
    m = Model(local_rank=-1, device=DEV)
    torch.manual_seed(seed)
    o = ModelRef(nd)
    for (ka, a), (kb, b) in zip(m.flownet.state_dict().items(), o.flownet.state_dict().items()):
        assert ka == kb and torch.equal(a.cpu(), b), ka  # same seed => same weights on both sides
    return m, o


def _check_step(m, o, data, nd, extra=()):
    from opticalflowscivis_amd.data import synthetic
    imgs, gt = data[:, :2].contiguous(), data[:, 2:3].contiguous()
    if nd == 3:
        po, oi = o.update(imgs, gt, learning_rate=1e-4, training=True)
        pp, pi = m.update(imgs.to(DEV), gt.to(DEV), learning_rate=1e-4, training=True)
    else:
        po, oi = o.update(imgs, gt, learning_rate=1e-4)
        pp, pi = m.update(imgs.to(DEV), gt.to(DEV), "droplet2d", learning_rate=1e-4, training=True)
    torch.cuda.synchronize()
    for k in ("loss_l1", "loss_tea", "loss_distill", "loss_G") + tuple(extra):
        a, b = float(pi[k].detach()), float(oi[k].detach())
        assert abs(a - b) <= 5e-4 * max(abs(b), 1e-6), (k, a, b)
    fo = oi["flow"].detach()
    fo = fo[:, :2] if nd == 2 else fo
    assert pi["flow"].shape == fo.shape
    assert float((pi["flow"].detach().cpu() - fo).abs().max()) < 1e-4
    gtc = gt[(slice(None), slice(None)) + tuple(slice(0, n) for n in po.shape[2:])]
    assert pp.shape == po.shape
    assert abs(synthetic.psnr(pp.detach().cpu(), gtc) - synthetic.psnr(po.detach(), gtc)) < 0.01
    # the weights after the AdamW step: per-tensor sums track the oracle's
    ps = np.array([float(p.detach().double().sum()) for p in m.flownet.parameters()])
    os_ = np.array([float(p.detach().double().sum()) for p in o.flownet.parameters()])
    np.testing.assert_allclose(ps, os_, rtol=1e-4, atol=5e-3)


def test_flow3d_step_vs_oracle_128():
    """Droplet-3D, B = 1 at 128^3: every layer runs the full-size kernel instantiations of the bench."""
    from opticalflowscivis_amd.data import synthetic
    m, o = _pair(3)
    _check_step(m, o, synthetic.droplet3d_batch(1, 128, seed=1234), 3)


def test_flow3d_step_at_256_vs_reference_golden(golden):
    """the train step at the size BASELINE's metric is quoted on, against values the REFERENCE
    itself produced (tests/golden/flow3d_256.npz, written by `make_golden.py flow3d_256` running
    Flow-3D/model/RIFE.py:81 `update` on the CPU: B = 1 at 256^3 -- the build host cannot hold B = 2 -- droplet
    triplet seed 1234, two AdamW steps at lr 1e-4).  Step 1 pins the forward (losses 5e-4 relative, final flow
    1e-4 px on every 8th voxel per axis plus whole-tensor moments, interpolation PSNR 0.01 dB); the weights after
    step 1 and the losses of step 2 pin the backward and the optimiser at this size."""
    from opticalflowscivis_amd.data import synthetic
    from opticalflowscivis_amd.flow3d.model.RIFE import Model
    g = golden("flow3d_256")
    S = int(g["size"])
    assert S == 256
    data = synthetic.droplet3d_batch(1, S, seed=1234)
    np.testing.assert_array_equal(np.array([float(data[0, c].double().sum()) for c in range(3)]), g["data_sums"])
    torch.manual_seed(1234)
    m = Model(local_rank=-1, device=DEV)
    ps0 = np.array([float(p.detach().double().sum()) for p in m.flownet.parameters()])
    np.testing.assert_allclose(ps0, g["param_sums"], rtol=0, atol=1e-9)  # same seed => the reference's weights
    imgs, gt = data[:, :2].to(DEV), data[:, 2:3].to(DEV)
    sl = (slice(None), slice(None), slice(0, None, 8), slice(0, None, 8), slice(0, None, 8))
    for step in range(2):
        pred, info = m.update(imgs, gt, learning_rate=1e-4, training=True)
        torch.cuda.synchronize()
        got = [float(info[k].detach()) for k in ("loss_l1", "loss_tea", "loss_distill", "loss_G")]
        for k, a, b in zip(("loss_l1", "loss_tea", "loss_distill", "loss_G"), got, g["update_losses"][step]):
            assert abs(a - b) <= 5e-4 * abs(b), (step, k, a, b)
        if step == 0:
            for name, t, tol in (("flow", info["flow"], 1e-4), ("flow_tea", info["flow_tea"], 1e-4),
                                 ("merged", pred, 2e-5), ("merged_tea", info["merged_tea"], 2e-5)):
                t = t.detach()
                d = float((t[sl].cpu() - torch.from_numpy(g[name + "_s8"])).abs().max())
                assert d < tol, (name, d)
                mom = np.array([float(t.double().mean()), float(t.double().abs().mean()),
                                float(t.double().pow(2).mean()), float(t.abs().max())])
                np.testing.assert_allclose(mom, g[name + "_moments"], rtol=2e-5, atol=1e-7, err_msg=name)
            assert abs(synthetic.psnr(pred.detach(), gt) - float(g["psnr"])) < 0.01
            assert abs(synthetic.psnr(info["merged_tea"].detach(), gt) - float(g["psnr_tea"])) < 0.01
            ps1 = np.array([float(p.detach().double().sum()) for p in m.flownet.parameters()])
            # AdamW's first update is lr * sign(g) per weight (1e-4 each): an entry whose gradient is within rounding
            # noise of 0 may go the other way and shifts its tensor's sum by 2e-4 -- the band allows 25 of them per
            # tensor (of up to 442 368 entries), the same band as the 128^3 oracle comparison above
            np.testing.assert_allclose(ps1, g["param_sums_after1"], rtol=1e-4, atol=5e-3)
        del pred, info
    ps2 = np.array([float(p.detach().double().sum()) for p in m.flownet.parameters()])
    np.testing.assert_allclose(ps2, g["param_sums_after2"], rtol=1e-4, atol=1e-2)


def _trunk_dispatch(B, S):
    """(forward slab kind, weight-gradient kernel id) the library picks for the 64 -> 64 k3 trunk layers of a
    scale-1 block at volume edge S (trunk edge S / 4): asked of its own dispatch, nothing launched."""
    from opticalflowscivis_amd import _lib, ops
    t = S // 4
    buf = (_lib.FsWprepJob * 4)()
    n = _lib.lib().fs_conv3d_fwd_wprep_jobs(buf, 4, 0x1000, 0x1000, 0x1000, B, 64, 64, t, t, t, t, t, t, 3, 1, 1, 0)
    assert n == 1
    return buf[0].kind, ops.conv3d_wrw_kernel_id(0x1000, 0x1000, B, 64, 64, (t, t, t), (t, t, t), 3, 1, 1)


def test_flow3d_training_drift_at_256_vs_reference_trajectory(golden):
    """the reference trains by repeated `update` on the running weights (Flow-3D/train.py:165-169).
    tests/golden/flow3d_256_traj.npz (`make_golden.py flow3d_256_traj`) holds what the REFERENCE's `Model.update`
    produced over EIGHT AdamW steps at B = 1, 256^3, seed 1234, lr 1e-4: the four losses and the PSNR of every step,
    every parameter's sum after steps 4 and 8.  The product takes the same eight steps on the kernels the bench times:
    the test first asserts that the trunk layers at this size dispatch to the Winograd-domain kernels (slab kind 6 =
    F(2,3) x F(4,3) forward / input gradient, weight-gradient kernel F(4,3)), so a threshold change cannot silently
    turn this into a test of the direct kernels, and that the step really launched them.
    Bands: the step is a chaotic map of its rounding errors -- AdamW's first updates are lr * sign(g) and the
    distillation term grows 150x over the eight steps -- so the band widens with the step (values below)."""
    from opticalflowscivis_amd import ops
    from opticalflowscivis_amd.data import synthetic
    from opticalflowscivis_amd.flow3d.model.RIFE import Model
    g = golden("flow3d_256_traj")
    S, n = int(g["size"]), int(g["steps"])
    assert S == 256 and n == 8
    kind, wrw = _trunk_dispatch(1, S)
    assert kind == 6 and wrw == ops.WRW_KERNEL_WINO43, (kind, wrw)
    data = synthetic.droplet3d_batch(1, S, seed=1234)
    np.testing.assert_array_equal(np.array([float(data[0, c].double().sum()) for c in range(3)]), g["data_sums"])
    torch.manual_seed(1234)
    m = Model(local_rank=-1, device=DEV)
    psum = lambda: np.array([float(p.detach().double().sum()) for p in m.flownet.parameters()])
    np.testing.assert_allclose(psum(), g["param_sums"], rtol=0, atol=1e-9)
    imgs, gt = data[:, :2].to(DEV), data[:, 2:3].to(DEV)
    names = ("loss_l1", "loss_tea", "loss_distill", "loss_G")
    drift, psnr_d, sums = [], [], {}
    ops.enable_kernel_timing(True)
    for step in range(n):
        pred, info = m.update(imgs, gt, learning_rate=1e-4, training=True)
        torch.cuda.synchronize()
        if step == 0:
            rec = ops.kernel_timings()
            ops.enable_kernel_timing(False)
            syms = {r[4] for rs in rec.values() for r in rs if r[4]}
            assert "conv3d_wino2d_ps_kernel<0, 16>" in syms and "conv3d_wrw_wino4_kernel<0>" in syms, syms
        got = [float(info[k].detach()) for k in names]
        drift.append([abs(a - b) / abs(b) for a, b in zip(got, g["update_losses"][step])])
        psnr_d.append(abs(synthetic.psnr(pred.detach(), gt) - float(g["psnr"][step])))
        del pred, info
        if step + 1 in (n // 2, n):
            sums[step + 1] = psum()
    drift = np.array(drift)
    print("relative drift per step (l1, tea, distill, G):\n", drift, "\nPSNR drift dB:", psnr_d)
    # Measured on MI355X (round 4, profiles/r04_traj_drift.txt): loss_l1 / loss_tea / loss_G agree with the reference to
    # <= 4e-7 relative through step 4 and then separate ~10x per step (4e-6, 1.5e-5, 1.4e-4 at steps 6, 7, 8: the map
    # amplifies rounding differences, the weight-gradient atomics make the last digits differ from run to run);
    # loss_distill to <= 5.5e-5 at every step; PSNR to 7e-4 dB at step 8.  Bands: >= 2x the measured worst.
    band = [2e-6, 2e-6, 2e-6, 2e-6, 4e-6, 2e-5, 6e-5, 5e-4]
    for step in range(n):
        assert max(drift[step, 0], drift[step, 1], drift[step, 3]) <= band[step], (step, drift)
        assert drift[step, 2] <= 2e-4, (step, drift[:, 2])
        assert psnr_d[step] <= (1e-4 if step < 6 else 5e-3), (step, psnr_d)
    np.testing.assert_allclose(sums[4], g["param_sums_after4"], rtol=1e-4, atol=2e-2)
    np.testing.assert_allclose(sums[8], g["param_sums_after8"], rtol=1e-4, atol=4e-2)


def test_flow3d_b2_at_256_equals_its_b1_slices():
    """the reference fixtures pin the 256^3 step at B = 1 (the build host cannot hold B = 2); the
    bench runs B = 2.  This ties the B = 2 LAUNCH GEOMETRY of every kernel of the step (twice the bricks / tiles, batch
    strides) to those pinned B = 1 runs: from the same weights, the forward of the two-sample batch equals the two
    one-sample forwards per sample (flow 1e-5 px, frames 2e-5: the coarse blocks' layers pick other brick sizes at half
    the batch, i.e. another summation order, nothing else differs), and
    the parameter gradients of the batch loss are the mean of the two samples' gradients (batch-mean losses: 5e-4 of
    each tensor's norm, the weight-gradient atomics' order is the noise).  A dropped sample or a wrong batch stride in
    any forward / backward kernel is an O(1) error here."""
    from opticalflowscivis_amd import ops
    from opticalflowscivis_amd.data import synthetic
    from opticalflowscivis_amd.flow3d.model.RIFE import Model
    S = 256
    assert _trunk_dispatch(2, S) == _trunk_dispatch(1, S) == (6, ops.WRW_KERNEL_WINO43)
    data = synthetic.droplet3d_batch(2, S, seed=4321, device=DEV)
    torch.manual_seed(1234)
    m = Model(local_rank=-1, device=DEV)
    params = [p for p in m.flownet.parameters()]

    def grads(batch):
        m.optimG.zero_grad(set_to_none=True)
        m.train()
        with ops.prepared_weights():
            flow, mask, merged, flow_tea, merged_tea, loss_distill = m.flownet((batch[:, :2].contiguous(), batch[:, 2:3].contiguous()),
                                                                               scale=[4, 2, 1])
            gt = batch[:, 2:3].contiguous()
            loss = ops.l1_loss(merged[2], gt) + ops.l1_loss(merged_tea, gt) + loss_distill * 0.1
            loss.backward()
        out = (flow[2].detach().clone(), merged[2].detach().clone(), float(loss.detach()),
               [p.grad.detach().double().clone() for p in params])
        return out

    f2, m2, l2, g2 = grads(data)
    acc = [torch.zeros_like(g) for g in g2]
    lsum = 0.0
    for i in range(2):
        f1, m1, l1, g1 = grads(data[i:i + 1])
        assert float((f2[i:i + 1] - f1).abs().max()) < 1e-5, i
        assert float((m2[i:i + 1] - m1).abs().max()) < 2e-5, i
        lsum += l1
        for a, g in zip(acc, g1):
            a += g
        del f1, m1, g1
    assert abs(l2 - lsum / 2) < 1e-5 * abs(l2)
    worst = max(float((g - a / 2).norm()) / max(float(g.norm()), 1e-30) for g, a in zip(g2, acc))
    print("B = 2 parameter gradients vs the mean of the B = 1 gradients: worst relative L2 error %.2e" % worst)
    assert worst < 5e-4  # measured 9.8e-5


def test_flow3d_step_vs_oracle_jets_c5():
    """BASELINE config C5's workload (5Jets-like smooth density field), per-GPU batch 2 at 64^3."""
    from opticalflowscivis_amd.data import synthetic
    m, o = _pair(3)
    _check_step(m, o, synthetic.jets3d_batch(2, 64, seed=1234), 3)


def test_flow2d_step_vs_oracle_cropped_extent():
    """H = 146: floor(H/4) % 4 == 0 and H % 4 != 0, so IFNet's outputs are 144 rows and the photometric
    term bilinearly resizes the frames to the warped extent (Flow-2D/model/RIFE.py:267)."""
    from opticalflowscivis_amd.data import synthetic
    m, o = _pair(2)
    data = synthetic.droplet2d_batch(2, 146, 96, seed=5, radius=(10, 20))
    _check_step(m, o, data, 2, extra=("loss_photo",))


# ---- the convolution kernels at the real layer shapes of the 2 x 256^3 step --------------------------
def _rel(a, b):
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-12)


def _wrw_taps(src, g, k, taps):
    """dW[:, ci, kz, ky, kx] evaluated directly (fp64 sums of strided slices) for a few (ci, kz, ky, kx):
    dW[co] = sum_{b,o} g[b,co,o] * src_pad[b,ci,2o + k - 1]   (k4 s2 p1)."""
    B, Cs, D, H, W = src.shape
    Do, Ho, Wo = g.shape[2:]
    out = {}
    for (ci, kz, ky, kx) in taps:
        xs = F.pad(src[:, ci:ci + 1], (1, 1, 1, 1, 1, 1))[:, :, kz:kz + 2 * Do:2, ky:ky + 2 * Ho:2, kx:kx + 2 * Wo:2]
        out[(ci, kz, ky, kx)] = (g.double() * xs.double()).sum(dim=(0, 2, 3, 4))
    return out


def test_conv0a_256_step_shape():
    """conv0a of the scale-1 blocks: Conv3d(11 -> 32, k4 s2 p1) on [2, 11, 256^3] -> [2, 32, 128^3]:
    forward vs MIOpen, input gradient (fs_conv3d_tr) vs MIOpen's transposed convolution, weight gradient
    (fs_conv3d_wrw) vs directly evaluated taps."""
    from opticalflowscivis_amd import ops
    torch.manual_seed(0)
    x = torch.randn(2, 11, 256, 256, 256, device=DEV)
    w = torch.randn(32, 11, 4, 4, 4, device=DEV) / (11 * 64) ** 0.5
    b = torch.randn(32, device=DEV)
    y = ops.conv3d_fwd(x, w, b, 4, 2, 1, 0)
    ref = F.conv3d(x, w, b, 2, 1)
    assert y.shape == ref.shape == (2, 32, 128, 128, 128)
    assert _rel(y, ref) < 2e-5
    del ref
    gy = torch.randn_like(y)
    del y
    gx = ops.conv3d_tr(gy, w, None, x.shape[2:])
    ref = F.conv_transpose3d(gy, w, None, 2, 1)
    assert gx.shape == ref.shape == x.shape
    assert _rel(gx, ref) < 2e-5
    del gx, ref
    gw = ops.conv3d_wrw(gy, x, 4, 2, 1)
    taps = [(0, 0, 0, 0), (10, 3, 3, 3), (5, 1, 2, 0), (7, 2, 0, 3)]
    for (ci, kz, ky, kx), want in _wrw_taps(x, gy, 4, taps).items():
        got = gw[:, ci, kz, ky, kx].double()
        assert float((got - want).abs().max()) < 2e-4 * float(want.abs().max()) + 0.05, (ci, kz, ky, kx)


def test_flow_head_256_step_shape():
    """Last layer of the flow head at scale 1: ConvTranspose3d(32 -> 6, k4 s2 p1) on [2, 32, 128^3] ->
    [2, 6, 256^3]: forward (fs_conv3d_tr) vs MIOpen, input gradient (fs_conv3d_fwd on grad_out) vs MIOpen's
    strided convolution, weight gradient vs directly evaluated taps."""
    from opticalflowscivis_amd import ops
    torch.manual_seed(1)
    x = torch.randn(2, 32, 128, 128, 128, device=DEV)
    w = torch.randn(32, 6, 4, 4, 4, device=DEV) / (32 * 8) ** 0.5
    b = torch.randn(6, device=DEV)
    y = ops.conv3d_tr(x, w, b)
    ref = F.conv_transpose3d(x, w, b, 2, 1)
    assert y.shape == ref.shape == (2, 6, 256, 256, 256)
    assert _rel(y, ref) < 2e-5
    del ref
    gy = torch.randn_like(y)
    del y
    gx = ops.conv3d_fwd(gy, w, None, 4, 2, 1, 0)  # the weight read as [out = 32][in = 6]
    ref = F.conv3d(gy, w, None, 2, 1)
    assert gx.shape == ref.shape == x.shape
    assert _rel(gx, ref) < 2e-5
    del gx, ref
    gw = ops.conv3d_wrw(x, gy, 4, 2, 1)  # transposed layer: (g, src) = (x, grad_out) -> [32, 6, 4,4,4]
    assert gw.shape == w.shape
    taps = [(0, 0, 0, 0), (5, 3, 3, 3), (2, 1, 2, 0), (3, 2, 0, 3)]
    for (ci, kz, ky, kx), want in _wrw_taps(gy, x, 4, taps).items():
        got = gw[:, ci, kz, ky, kx].double()
        assert float((got - want).abs().max()) < 2e-4 * float(want.abs().max()) + 0.05, (ci, kz, ky, kx)
