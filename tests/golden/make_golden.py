"""Generates the golden vectors under tests/golden/ by RUNNING THE REFERENCE on CPU.

Build-container only: needs /root/reference (read-only).  The reference never travels to the
GPU box; only the .npz fixtures written here (inputs + expected outputs / gradients) do.
Usage (each group imports a different reference package layout, so one process per group):

    python tests/golden/make_golden.py rife_ops     # Flow-2D / Flow-3D warplayer.warp
    python tests/golden/make_golden.py upflow_ops   # Corr_pyTorch, warps, census, photo losses
    python tests/golden/make_golden.py flow3d_e2e   # Flow-3D Model.update / inference
    python tests/golden/make_golden.py flow2d_e2e   # Flow-2D Model.update / inference
    python tests/golden/make_golden.py flow3d_256   # Flow-3D Model.update at the BASELINE size (B=1, 256^3; ~35 GB, minutes)
    python tests/golden/make_golden.py flow3d_256_traj  # the same, eight AdamW steps (training drift; ~10 min)
    python tests/golden/make_golden.py upflow_e2e   # UPFlow_net forward losses / flows / grads
    python tests/golden/make_golden.py upflow_sgu   # UPFlow_net with the self-guided upsampling module on
    python tests/golden/make_golden.py upflow_levels  # per pyramid level: decode_level_res inputs / outputs / grads
    python tests/golden/make_golden.py rife_next    # Flow-2D LapLoss
    python tests/golden/make_golden.py upflow_next  # occ_check_model, normalize_features
    python tests/golden/make_golden.py ckpt         # Model.save_model / load_model on-disk format
    python tests/golden/make_golden.py all          # runs the groups above as subprocesses

Third-party modules the reference imports at module scope but that are absent from this
image (cv2, torchvision, ...) are replaced by inert stubs; none of them is on the hot path.
"""
import importlib.util
import os
import subprocess
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _install_stubs():
    class _Any:
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return _Any()

        def __call__(self, *a, **k):
            return _Any()

    for name in ["cv2", "imageio", "png", "pyimof", "plotly", "plotly.graph_objects", "torchvision",
                 "torchvision.models", "skimage", "skimage.transform", "correlation_cuda",
                 "ensurepip", "matplotlib", "matplotlib.pyplot", "matplotlib.colors",
                 "mpl_toolkits", "mpl_toolkits.axes_grid1"]:
        if name in sys.modules:
            continue
        try:
            if name.startswith("matplotlib") or name.startswith("mpl_toolkits"):
                __import__(name)
                continue
        except Exception:
            pass
        m = types.ModuleType(name)
        m.__file__ = "/dev/null/%s.py" % name

        def _ga(k, _Any=_Any):
            if k.startswith("__"):
                raise AttributeError(k)
            return _Any()

        m.__getattr__ = _ga
        sys.modules[name] = m


def _load_file(modname, path):
    spec = importlib.util.spec_from_file_location(modname, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _grads(out, inputs, seed):
    g = torch.Generator().manual_seed(seed)
    G = torch.randn(out.shape, generator=g)
    grads = torch.autograd.grad((out * G).sum(), inputs, allow_unused=True)
    return G, grads


def _np(t):
    return t.detach().cpu().numpy()


def _flow_cases(shape, gen, amp):
    """A smooth-ish random flow with a few wild vectors that leave the domain."""
    f = (torch.rand(shape, generator=gen) * 2 - 1) * amp
    wild = torch.rand(shape, generator=gen) < 0.03
    f = torch.where(wild, f * 6.0, f)
    return f


# ------------------------------------------------------------------------------------------
def rife_ops():
    w2 = _load_file("ref_warp2d", REF + "/Flow-2D/model/warplayer.py")
    w3 = _load_file("ref_warp3d", REF + "/Flow-3D/model/warplayer.py")
    w2.device = torch.device("cpu")
    w3.device = torch.device("cpu")
    gen = torch.Generator().manual_seed(20241)
    store = {}
    # 2-D: a1
    for tag, (B, C, H, W), amp in [("a", (2, 3, 16, 24), 2.5), ("b", (1, 1, 9, 7), 1.5)]:
        x = torch.rand(B, C, H, W, generator=gen).requires_grad_()
        f = _flow_cases((B, 2, H, W), gen, amp).requires_grad_()
        out = w2.warp(x, f)
        G, (gx, gf) = _grads(out, [x, f], 7)
        for k, v in dict(x=x, f=f, out=out, G=G, gx=gx, gf=gf).items():
            store["w2_%s_%s" % (tag, k)] = _np(v)
    x = torch.rand(1, 2, 8, 12, generator=gen)
    store["w2_zero_x"] = _np(x)
    store["w2_zero_out"] = _np(w2.warp(x, torch.zeros(1, 2, 8, 12)))
    # 3-D: a2, non-cubic (pins every axis role) and cubic
    for tag, (B, C, D, H, W), amp in [("nc", (1, 2, 6, 8, 10), 1.5), ("cu", (2, 1, 8, 8, 8), 2.0),
                                      ("tile", (1, 1, 3, 70, 37), 2.0)]:
        x = torch.rand(B, C, D, H, W, generator=gen).requires_grad_()
        f = _flow_cases((B, 3, D, H, W), gen, amp).requires_grad_()
        out = w3.warp(x, f)
        G, (gx, gf) = _grads(out, [x, f], 11)
        for k, v in dict(x=x, f=f, out=out, G=G, gx=gx, gf=gf).items():
            store["w3_%s_%s" % (tag, k)] = _np(v)
    x = torch.rand(1, 1, 5, 6, 7, generator=gen)
    store["w3_zero_x"] = _np(x)
    store["w3_zero_out"] = _np(w3.warp(x, torch.zeros(1, 3, 5, 6, 7)))
    # input extent != flow extent (IFNet-3D on sizes that are not multiples of 16)
    x = torch.rand(1, 2, 10, 9, 12, generator=gen).requires_grad_()
    f = _flow_cases((1, 3, 8, 8, 8), gen, 1.5).requires_grad_()
    out = w3.warp(x, f)
    G, (gx, gf) = _grads(out, [x, f], 13)
    for k, v in dict(x=x, f=f, out=out, G=G, gx=gx, gf=gf).items():
        store["w3_mixed_%s" % k] = _np(v)
    np.savez_compressed(os.path.join(OUT, "rife_ops.npz"), **store)
    print("wrote rife_ops.npz", len(store), "arrays")


# ------------------------------------------------------------------------------------------
def upflow_ops():
    _install_stubs()
    sys.path[:0] = [REF + "/UPFlow"]
    from utils.pytorch_correlation import Corr_pyTorch
    from utils.loss import loss_functions
    from utils.tools import tools
    from model.pwc_modules import WarpingLayer_no_div
    import model.upflow as U
    U.device = torch.device("cpu")
    gen = torch.Generator().manual_seed(777)
    store = {}
    # a3/a4 correlation (Corr_pyTorch, the reference's stand-in for correlation_cuda)
    corr = Corr_pyTorch(pad_size=4, kernel_size=1, max_displacement=4, stride1=1, stride2=1,
                        corr_multiply=1)
    for tag, (B, C, H, W) in [("c3", (2, 3, 10, 14)), ("c32", (1, 32, 10, 14)), ("tiny", (1, 5, 3, 8))]:
        f1 = torch.randn(B, C, H, W, generator=gen).requires_grad_()
        f2 = torch.randn(B, C, H, W, generator=gen).requires_grad_()
        out = corr(f1, f2)
        G, (g1, g2) = _grads(out, [f1, f2], 3)
        for k, v in dict(f1=f1, f2=f2, out=out, G=G, g1=g1, g2=g2).items():
            store["corr_%s_%s" % (tag, k)] = _np(v)
    # a5 WarpingLayer_no_div, a6 torch_warp
    wl = WarpingLayer_no_div()
    B, C, H, W = 2, 4, 12, 20
    x = torch.rand(B, C, H, W, generator=gen).requires_grad_()
    f = _flow_cases((B, 2, H, W), gen, 2.0).requires_grad_()
    out = wl(x, f)
    G, (gx, gf) = _grads(out, [x, f], 5)
    for k, v in dict(x=x, f=f, out=out, G=G, gx=gx, gf=gf).items():
        store["pwcmask_%s" % k] = _np(v)
    out = tools.torch_warp(x, f)
    G, (gx, gf) = _grads(out, [x, f], 6)
    for k, v in dict(out=out, G=G, gx=gx, gf=gf).items():
        store["pwc_%s" % k] = _np(v)
    # a7 boundary_dilated_warp.warp_im
    B, C, H, W = 2, 3, 14, 18
    I = torch.rand(B, C, H, W, generator=gen).requires_grad_()
    f = _flow_cases((B, 2, H, W), gen, 2.0).requires_grad_()
    for tag, start in [("s0", torch.zeros(B, 2, 1, 1)),
                       ("s1", torch.tensor([[1.0, 2.0], [0.0, -1.0]]).view(B, 2, 1, 1))]:
        out = tools.boundary_dilated_warp.warp_im(I, f, start)
        G, (gI, gf) = _grads(out, [I, f], 8)
        for k, v in dict(I=I, f=f, start=start, out=out, G=G, gI=gI, gf=gf).items():
            store["dil_%s_%s" % (tag, k)] = _np(v)
    # a8 census loss, a10 photo_loss_function
    B, H, W = 2, 32, 48
    im1 = torch.rand(B, 3, H, W, generator=gen).requires_grad_()
    im2 = (im1.detach() + 0.1 * torch.randn(B, 3, H, W, generator=gen)).requires_grad_()
    occ = (torch.rand(B, 1, H, W, generator=gen) > 0.3).float()
    store.update(cen_im1=_np(im1), cen_im2=_np(im2), cen_occ=_np(occ))
    for tag, (cha, useocc) in [("abs", (False, False)), ("absocc", (False, True)),
                               ("cha", (True, False)), ("chaocc", (True, True))]:
        loss = loss_functions.census_loss_torch(im1, im2, occ, q=0.4, charbonnier_or_abs_robust=cha,
                                                if_use_occ=useocc, averge=True)
        g1, g2 = torch.autograd.grad(loss, [im1, im2])
        store["cen_%s_loss" % tag] = _np(loss)
        store["cen_%s_g1" % tag] = _np(g1)
        store["cen_%s_g2" % tag] = _np(g2)
    # a9 photo_loss_multi_type
    for typ in ["abs_robust", "charbonnier", "L1", "SSIM"]:
        for useocc in [False, True]:
            loss = U.network_tools.photo_loss_multi_type(im1, im2, occ, photo_loss_type=typ,
                                                         photo_loss_delta=0.4,
                                                         photo_loss_use_occ=useocc)
            g1, g2 = torch.autograd.grad(loss, [im1, im2])
            tag = "%s_%d" % (typ, int(useocc))
            store["photo_%s_loss" % tag] = _np(loss)
            store["photo_%s_g1" % tag] = _np(g1)
            store["photo_%s_g2" % tag] = _np(g2)
    np.savez_compressed(os.path.join(OUT, "upflow_ops.npz"), **store)
    print("wrote upflow_ops.npz", len(store), "arrays")


# ------------------------------------------------------------------------------------------
def _quiet(fn, *a, **k):
    """The reference prints on every forward; keep the generator's output readable."""
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def flow3d_e2e():
    _install_stubs()
    sys.path[:0] = [REF + "/Flow-3D", REF]
    import model.RIFE as R
    import model.warplayer as WL
    R.device = torch.device("cpu")
    WL.device = torch.device("cpu")
    torch.manual_seed(1234)
    m = _quiet(R.Model, local_rank=-1)
    gen = torch.Generator().manual_seed(99)
    S = 32
    data = torch.rand(1, 3, S, S, S, generator=gen)
    imgs, gt = data[:, :2], data[:, 2:3]
    store = dict(data=_np(data))
    nparam = sum(p.numel() for p in m.flownet.parameters())
    store["nparam"] = np.array(nparam)
    # parameter fingerprint (seed-compatibility check for the oracle / product models)
    store["param_sums"] = np.array([float(p.detach().double().sum()) for p in m.flownet.parameters()])
    m.eval()
    with torch.no_grad():
        merged, flows, mask = _quiet(m.inference, imgs[:, :1], imgs[:, 1:2], [4, 2, 1])
    store["inf_merged"] = _np(merged)
    store["inf_flow2"] = _np(flows[2])
    store["inf_mask"] = _np(mask)
    losses = []
    for step in range(2):
        pred, info = _quiet(m.update, imgs, gt, learning_rate=1e-4, training=True)
        losses.append([float(info[k]) for k in ("loss_l1", "loss_tea", "loss_distill", "loss_G")])
    store["update_losses"] = np.array(losses)
    store["update_pred_last"] = _np(pred)
    store["param_sums_after"] = np.array([float(p.detach().double().sum()) for p in m.flownet.parameters()])
    np.savez_compressed(os.path.join(OUT, "flow3d_e2e.npz"), **store)
    print("wrote flow3d_e2e.npz; losses", losses, "nparam", nparam)

def flow3d_256():
    """The reference's Flow-3D `Model.update` (Flow-3D/model/RIFE.py:81) at the size BASELINE's metric is quoted
    on: one 256^3 droplet triplet (B = 1: the host cannot hold B = 2), seed 1234, two AdamW steps at lr 1e-4.
    The input is NOT stored (201 MB): it is `synthetic.droplet3d_batch(1, 256, seed=1234)` of this repo, a
    deterministic function of (seed, shape) whose per-frame sums are stored as a check.  Stored: the four losses
    of both steps, every parameter's sum before / after, the interpolation PSNR, and every 8th voxel per axis of
    the first step's final flow / merged frame / teacher frame / mask (plus whole-tensor moments)."""
    _install_stubs()
    repo = os.path.dirname(os.path.dirname(OUT))
    sys.path[:0] = [REF + "/Flow-3D", REF]
    import model.RIFE as R
    import model.warplayer as WL
    R.device = torch.device("cpu")
    WL.device = torch.device("cpu")
    spec = importlib.util.spec_from_file_location(
        "flowsci_synthetic", os.path.join(repo, "opticalflowscivis_amd", "data", "synthetic.py"))
    syn = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(syn)
    S = int(os.environ.get("FLOW3D_256_SIZE", "256"))
    torch.manual_seed(1234)
    m = _quiet(R.Model, local_rank=-1)
    data = syn.droplet3d_batch(1, S, seed=1234)
    imgs, gt = data[:, :2], data[:, 2:3]
    store = dict(size=np.int64(S), data_sums=np.array([float(data[0, c].double().sum()) for c in range(3)]))
    store["param_sums"] = np.array([float(p.detach().double().sum()) for p in m.flownet.parameters()])
    losses = []
    import time
    for step in range(2):
        t0 = time.time()
        pred, info = _quiet(m.update, imgs, gt, learning_rate=1e-4, training=True)
        losses.append([float(info[k].detach()) for k in ("loss_l1", "loss_tea", "loss_distill", "loss_G")])
        print("step", step, "%.1f s" % (time.time() - t0), losses[-1], flush=True)
        if step == 0:
            sl = (slice(None), slice(None), slice(0, None, 8), slice(0, None, 8), slice(0, None, 8))
            for name, t in (("flow", info["flow"]), ("merged", pred), ("merged_tea", info["merged_tea"]),
                            ("flow_tea", info["flow_tea"])):
                t = t.detach()
                store[name + "_s8"] = _np(t[sl].contiguous())
                store[name + "_moments"] = np.array([float(t.double().mean()), float(t.double().abs().mean()),
                                                     float(t.double().pow(2).mean()), float(t.abs().max())])
            store["psnr"] = np.array(syn.psnr(pred.detach(), gt))
            store["psnr_tea"] = np.array(syn.psnr(info["merged_tea"].detach(), gt))
            store["param_sums_after1"] = np.array([float(p.detach().double().sum()) for p in m.flownet.parameters()])
        del pred, info
    store["update_losses"] = np.array(losses)
    store["param_sums_after2"] = np.array([float(p.detach().double().sum()) for p in m.flownet.parameters()])
    name = "flow3d_256.npz" if S == 256 else "flow3d_%d_probe.npz" % S
    np.savez_compressed(os.path.join(OUT, name), **store)
    print("wrote", name, "; losses", losses, os.path.getsize(os.path.join(OUT, name)) // 1024, "KB")


def flow3d_256_traj():
    """Training drift at the BASELINE size: the reference's `train()` is repeated `Model.update` on the running
    weights (Flow-3D/train.py:165-169).  Same model / triplet / learning rate as `flow3d_256`, EIGHT AdamW steps
    (B = 1, 256^3, seed 1234; ~1 min per step on 8 cores).  Stored: the four losses of every step, every
    parameter's sum before and after steps 4 and 8, PSNR of every step's prediction.  The first two rows equal
    `flow3d_256.npz::update_losses` (same process, same seed) -- asserted here before anything is written."""
    _install_stubs()
    repo = os.path.dirname(os.path.dirname(OUT))
    sys.path[:0] = [REF + "/Flow-3D", REF]
    import model.RIFE as R
    import model.warplayer as WL
    R.device = torch.device("cpu")
    WL.device = torch.device("cpu")
    spec = importlib.util.spec_from_file_location(
        "flowsci_synthetic", os.path.join(repo, "opticalflowscivis_amd", "data", "synthetic.py"))
    syn = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(syn)
    S = int(os.environ.get("FLOW3D_256_SIZE", "256"))
    NSTEP = int(os.environ.get("FLOW3D_TRAJ_STEPS", "8"))
    torch.manual_seed(1234)
    m = _quiet(R.Model, local_rank=-1)
    data = syn.droplet3d_batch(1, S, seed=1234)
    imgs, gt = data[:, :2], data[:, 2:3]
    store = dict(size=np.int64(S), steps=np.int64(NSTEP),
                 data_sums=np.array([float(data[0, c].double().sum()) for c in range(3)]))
    psum = lambda: np.array([float(p.detach().double().sum()) for p in m.flownet.parameters()])
    store["param_sums"] = psum()
    losses, psnrs = [], []
    import time
    for step in range(NSTEP):
        t0 = time.time()
        pred, info = _quiet(m.update, imgs, gt, learning_rate=1e-4, training=True)
        losses.append([float(info[k].detach()) for k in ("loss_l1", "loss_tea", "loss_distill", "loss_G")])
        psnrs.append(float(syn.psnr(pred.detach(), gt)))
        print("step", step, "%.1f s" % (time.time() - t0), losses[-1], psnrs[-1], flush=True)
        del pred, info
        if step + 1 in (NSTEP // 2, NSTEP):
            store["param_sums_after%d" % (step + 1)] = psum()
    store["update_losses"] = np.array(losses)
    store["psnr"] = np.array(psnrs)
    if S == 256:
        two = np.load(os.path.join(OUT, "flow3d_256.npz"))["update_losses"]
        assert np.array_equal(two, store["update_losses"][:2]), (two, store["update_losses"][:2])
    name = "flow3d_256_traj.npz" if S == 256 else "flow3d_%d_traj_probe.npz" % S
    np.savez_compressed(os.path.join(OUT, name), **store)
    print("wrote", name, "; losses", losses, os.path.getsize(os.path.join(OUT, name)) // 1024, "KB")


def flow2d_e2e():
    _install_stubs()
    sys.path[:0] = [REF + "/Flow-2D", REF]
    import model.RIFE as R
    import model.warplayer as WL
    R.device = torch.device("cpu")
    WL.device = torch.device("cpu")
    torch.manual_seed(1234)
    m = _quiet(R.Model, local_rank=-1)
    gen = torch.Generator().manual_seed(98)
    H, W = 64, 96
    data = torch.rand(2, 3, H, W, generator=gen)
    imgs, gt = data[:, :2], data[:, 2:3]
    store = dict(data=_np(data))
    store["nparam"] = np.array(sum(p.numel() for p in m.flownet.parameters()))
    store["param_sums"] = np.array([float(p.detach().double().sum()) for p in m.flownet.parameters()])
    m.eval()
    with torch.no_grad():
        merged, flows, mask = _quiet(m.inference, imgs[:, :1], imgs[:, 1:2], [4, 2, 1])
    store["inf_merged"] = _np(merged[2])  # Flow-2D inference returns all three frames / masks
    store["inf_flow2"] = _np(flows[2])
    store["inf_mask"] = _np(mask[2])
    losses, keys = [], None
    for step in range(2):
        pred, info = _quiet(m.update, imgs, gt, "droplet2d", learning_rate=1e-4, training=True)
        keys = [k for k in info if k.startswith("loss")]
        losses.append([float(info[k]) for k in keys])
    store["update_loss_keys"] = np.array(keys)
    store["update_losses"] = np.array(losses)
    store["update_pred_last"] = _np(pred)
    store["param_sums_after"] = np.array([float(p.detach().double().sum()) for p in m.flownet.parameters()])
    np.savez_compressed(os.path.join(OUT, "flow2d_e2e.npz"), **store)
    print("wrote flow2d_e2e.npz; keys", keys, "losses", losses)


def upflow_e2e():
    _install_stubs()
    sys.path[:0] = [REF + "/UPFlow"]
    import model.upflow as U
    U.device = torch.device("cpu")
    conf = U.UPFlow_net.config()
    _quiet(conf.update, {'if_norm_before_cost_volume': True, 'norm_moments_across_channels': False,
                         'norm_moments_across_images': False, 'if_use_cor_pytorch': True,
                         'if_sgu_upsample': False, 'photo_loss_census_weight': 1,
                         'multi_scale_distillation_weight': 1})
    torch.manual_seed(0)
    net = _quiet(conf)
    # a smooth moving pattern (so that flows / occlusion masks are not pure noise), seeded
    gen = torch.Generator().manual_seed(4)
    H, W = 128, 192
    base = torch.nn.functional.interpolate(torch.rand(2, 3, H // 8, W // 8 + 2, generator=gen), size=(H, W + 16),
                                           mode="bicubic", align_corners=True).clamp(0, 1)
    im1, im2 = base[:, :, :, :W].contiguous(), base[:, :, :, 3:W + 3].contiguous()
    store = dict(im1=_np(im1), im2=_np(im2))
    store["nparam"] = np.array(sum(p.numel() for p in net.parameters()))
    store["param_sums"] = np.array([float(p.detach().double().sum()) for p in net.parameters()])
    out = _quiet(net, {'im1': im1.numpy(), 'im2': im2.numpy(), 'if_loss': True})
    ld = out['loss_dict']
    keys = ['photo_loss', 'smooth_loss', 'census_loss', 'msd_loss']
    store["loss_keys"] = np.array(keys)
    store["losses"] = np.array([float(ld[k]) for k in keys])
    store["flow_f_out"] = _np(out['flow_f_out'])
    store["flow_b_out"] = _np(out['flow_b_out'])
    store["occ_fw"] = _np(out['occ_fw'])
    store["im1_warp"] = _np(out['im1_warp'])
    total = sum(ld[k] for k in keys)
    total.backward()
    store["grad_abs_sums"] = np.array([float(p.grad.detach().double().abs().sum()) if p.grad is not None else 0.0
                                       for p in net.parameters()])
    np.savez_compressed(os.path.join(OUT, "upflow_e2e.npz"), **store)
    print("wrote upflow_e2e.npz; losses", dict(zip(keys, store["losses"])), "nparam", int(store["nparam"]))


def upflow_sgu():
    """UPFlow_net with the self-guided upsampling module on (`if_sgu_upsample=True`, UPFlow/model/upflow.py:21-92,
    362-363, 612-616, 629-631, 677-679): forward + backward on a seeded pair, same recipe as `upflow_e2e` at 96 x 128."""
    _install_stubs()
    sys.path[:0] = [REF + "/UPFlow"]
    import model.upflow as U
    U.device = torch.device("cpu")
    conf = U.UPFlow_net.config()
    _quiet(conf.update, {'if_norm_before_cost_volume': True, 'norm_moments_across_channels': False,
                         'norm_moments_across_images': False, 'if_use_cor_pytorch': True,
                         'if_sgu_upsample': True, 'photo_loss_census_weight': 1,
                         'multi_scale_distillation_weight': 1})
    torch.manual_seed(0)
    net = _quiet(conf)
    gen = torch.Generator().manual_seed(5)
    H, W = 96, 128
    base = torch.nn.functional.interpolate(torch.rand(2, 3, H // 8, W // 8 + 2, generator=gen), size=(H, W + 16),
                                           mode="bicubic", align_corners=True).clamp(0, 1)
    im1, im2 = base[:, :, :, :W].contiguous(), base[:, :, :, 3:W + 3].contiguous()
    store = dict(im1=_np(im1), im2=_np(im2))
    store["nparam"] = np.array(sum(p.numel() for p in net.parameters()))
    store["param_names"] = np.array([n for n, _ in net.named_parameters()])
    store["param_sums"] = np.array([float(p.detach().double().sum()) for p in net.parameters()])
    out = _quiet(net, {'im1': im1.numpy(), 'im2': im2.numpy(), 'if_loss': True})
    ld = out['loss_dict']
    keys = ['photo_loss', 'smooth_loss', 'census_loss', 'msd_loss']
    store["loss_keys"] = np.array(keys)
    store["losses"] = np.array([float(ld[k]) for k in keys])
    store["flow_f_out"] = _np(out['flow_f_out'])
    store["flow_b_out"] = _np(out['flow_b_out'])
    store["occ_fw"] = _np(out['occ_fw'])
    store["im1_warp"] = _np(out['im1_warp'])
    total = sum(ld[k] for k in keys)
    total.backward()
    store["grad_abs_sums"] = np.array([float(p.grad.detach().double().abs().sum()) if p.grad is not None else 0.0
                                       for p in net.parameters()])
    np.savez_compressed(os.path.join(OUT, "upflow_sgu.npz"), **store)
    print("wrote upflow_sgu.npz; losses", dict(zip(keys, store["losses"])), "nparam", int(store["nparam"]))


# ------------------------------------------------------------------------------------------
def _proj(t, seed):
    """Two numbers that pin a gradient tensor without storing it: its dot product with a seeded Gaussian
    tensor of the same shape, and its absolute sum."""
    if t is None:
        return np.array([0.0, 0.0])
    R = torch.randn(t.shape, generator=torch.Generator().manual_seed(seed))
    return np.array([float((t.double() * R.double()).sum()), float(t.double().abs().sum())])


def upflow_levels():
    """Teacher-forcing fixture: for every pyramid level of one reference forward pass, the
    inputs the reference fed to `decode_level_res` (UPFlow/model/upflow.py:621-663), the two warped feature maps
    its `warping_layer` produced (the fp32-borderline validity-mask decisions included), its four outputs, and
    -- for the cotangents G1, G2 = seeded Gaussians on the two residual flows -- projections of the gradients
    w.r.t. every level input and every parameter of the estimator / context networks."""
    _install_stubs()
    sys.path[:0] = [REF + "/UPFlow"]
    import model.upflow as U
    U.device = torch.device("cpu")
    conf = U.UPFlow_net.config()
    _quiet(conf.update, {'if_norm_before_cost_volume': True, 'norm_moments_across_channels': False,
                         'norm_moments_across_images': False, 'if_use_cor_pytorch': True,
                         'if_sgu_upsample': False, 'photo_loss_census_weight': 1,
                         'multi_scale_distillation_weight': 1})
    torch.manual_seed(0)
    net = _quiet(conf)
    gen = torch.Generator().manual_seed(4)
    H, W = 128, 192
    base = torch.nn.functional.interpolate(torch.rand(1, 3, H // 8, W // 8 + 2, generator=gen), size=(H, W + 16),
                                           mode="bicubic", align_corners=True).clamp(0, 1)
    im1, im2 = base[:, :, :, :W].contiguous(), base[:, :, :, 3:W + 3].contiguous()
    store = dict(im1=_np(im1), im2=_np(im2))
    store["param_sums"] = np.array([float(p.detach().double().sum()) for p in net.parameters()])
    pnames = [n for n, _ in net.named_parameters() if n.startswith(("flow_estimators.", "context_networks."))]
    store["param_names"] = np.array(pnames)
    params = dict(net.named_parameters())
    orig_decode = net.decode_level_res
    orig_warp = net.warping_layer.forward
    names = ["flow_1", "flow_2", "feature_1", "feature_1_1x1", "feature_2", "feature_2_1x1"]

    def decode(level, flow_1, flow_2, feature_1, feature_1_1x1, feature_2, feature_2_1x1, img_ori_1, img_ori_2):
        ins = [t.detach().clone().requires_grad_() for t in (flow_1, flow_2, feature_1, feature_1_1x1, feature_2,
                                                              feature_2_1x1)]
        warped = []

        def warp(x, flow):
            out = orig_warp(x, flow)
            warped.append(out)
            return out

        net.warping_layer.forward = warp
        try:
            outs = orig_decode(level=level, flow_1=ins[0], flow_2=ins[1], feature_1=ins[2], feature_1_1x1=ins[3],
                               feature_2=ins[4], feature_2_1x1=ins[5], img_ori_1=img_ori_1, img_ori_2=img_ori_2)
        finally:
            net.warping_layer.forward = orig_warp
        tag = "L%d_" % level
        for n, t in zip(names, ins):
            store[tag + n] = _np(t)
        for n, t in zip(["flow_1_up", "flow_2_up", "res_1", "res_2"], outs):
            store[tag + "out_" + n] = _np(t)
        if warped:
            store[tag + "feature_2_warp"] = _np(warped[0])
            store[tag + "feature_1_warp"] = _np(warped[1])
        G1 = torch.randn(outs[2].shape, generator=torch.Generator().manual_seed(100 + level))
        G2 = torch.randn(outs[3].shape, generator=torch.Generator().manual_seed(200 + level))
        loss = (outs[2] * G1).sum() + (outs[3] * G2).sum()
        wrt = ins + [params[n] for n in pnames]
        grads = torch.autograd.grad(loss, wrt, allow_unused=True)
        store[tag + "gin"] = np.stack([_proj(g, 300 + 10 * level + i) for i, g in enumerate(grads[:6])])
        store[tag + "gparam"] = np.stack([_proj(g, 1000 + i) for i, g in enumerate(grads[6:])])
        return tuple(t.detach() for t in outs)

    net.decode_level_res = decode
    with torch.no_grad():
        pass
    flow_f, flow_b, flows = _quiet(net.forward_2_frame_v3, im1, im2)
    store["flow_f_out"] = _np(flow_f)
    store["nlevels"] = np.int64(len(flows))
    np.savez_compressed(os.path.join(OUT, "upflow_levels.npz"), **store)
    print("wrote upflow_levels.npz:", len(flows), "levels,", len(store), "arrays,",
          os.path.getsize(os.path.join(OUT, "upflow_levels.npz")) // 1024, "KB")


# ------------------------------------------------------------------------------------------
def rife_next():
    """the Flow-2D Laplacian-pyramid loss (Flow-2D/model/laplacian.py)."""
    lap = _load_file("ref_lap2d", REF + "/Flow-2D/model/laplacian.py")
    lap.device = torch.device("cpu")
    gen = torch.Generator().manual_seed(3131)
    store = {}
    for tag, (B, C, H, W), levels in [("even", (2, 1, 48, 64), 5), ("odd", (1, 1, 37, 51), 5),
                                      ("l3", (2, 1, 13, 18), 3), ("c2", (1, 2, 24, 40), 4)]:
        a = torch.rand(B, C, H, W, generator=gen).requires_grad_()
        b = (a.detach() + 0.2 * torch.randn(B, C, H, W, generator=gen)).clamp(0, 1).requires_grad_()
        crit = lap.LapLoss(max_levels=levels, channels=C)
        loss = crit(a, b)
        ga, gb = torch.autograd.grad(loss, [a, b])
        for k, v in dict(a=a, b=b, loss=loss, ga=ga, gb=gb).items():
            store["lap_%s_%s" % (tag, k)] = _np(v)
        store["lap_%s_levels" % tag] = np.int64(levels)
    np.savez_compressed(os.path.join(OUT, "rife_next.npz"), **store)
    print("wrote rife_next.npz", len(store), "arrays")


# ------------------------------------------------------------------------------------------
def upflow_next():
    """SURVEY §8f rows on the UPFlow side: occlusion check (f2), normalize_features (f4)."""
    _install_stubs()
    sys.path[:0] = [REF + "/UPFlow"]
    from utils.tools import tools
    import model.upflow as U
    U.device = torch.device("cpu")
    gen = torch.Generator().manual_seed(4242)
    store = {}
    B, H, W = 2, 24, 40
    # a roughly consistent pair (flow_b ~ -flow_f) plus noise, plus a stripe leaving the frame
    ff = 3.0 * torch.randn(B, 2, 1, 1, generator=gen) + 0.8 * torch.randn(B, 2, H, W, generator=gen)
    fb = -ff + 0.6 * torch.randn(B, 2, H, W, generator=gen)
    ff[:, 0, :, -6:] += 9.0
    fb[:, 1, :4] -= 7.0
    store.update(occ_ff=_np(ff), occ_fb=_np(fb))
    for mode in ("all", "obj", "out"):
        for scale in (1, 4):
            m = tools.occ_check_model(occ_type='for_back_check', occ_alpha_1=0.1, occ_alpha_2=0.5,
                                      obj_out_all=mode)
            of, ob = m(flow_f=ff, flow_b=fb, scale=scale)
            store["occ_%s_s%d_f" % (mode, scale)] = _np(of)
            store["occ_%s_s%d_b" % (mode, scale)] = _np(ob)
    # normalize_features: the four flag combinations, gradients included
    f1 = (1.5 * torch.randn(2, 5, 9, 13, generator=gen) + 0.7).requires_grad_()
    f2 = (0.5 * torch.randn(2, 5, 9, 13, generator=gen) - 0.2).requires_grad_()
    store.update(nf_f1=_np(f1), nf_f2=_np(f2))
    for ch in (False, True):
        for im in (False, True):
            o1, o2 = U.network_tools.normalize_features((f1, f2), normalize=True, center=True,
                                                        moments_across_channels=ch,
                                                        moments_across_images=im)
            G1 = torch.randn(o1.shape, generator=torch.Generator().manual_seed(11))
            G2 = torch.randn(o2.shape, generator=torch.Generator().manual_seed(12))
            g1, g2 = torch.autograd.grad((o1 * G1).sum() + (o2 * G2).sum(), [f1, f2])
            tag = "nf_c%d_i%d_" % (int(ch), int(im))
            for k, v in dict(o1=o1, o2=o2, G1=G1, G2=G2, g1=g1, g2=g2).items():
                store[tag + k] = _np(v)
    np.savez_compressed(os.path.join(OUT, "upflow_next.npz"), **store)
    print("wrote upflow_next.npz", len(store), "arrays")


# ------------------------------------------------------------------------------------------
def ckpt():
    """SURVEY §8f.4 second half: the on-disk checkpoint format of Flow-3D / Flow-2D `Model.save_model`
    (RIFE.py:61-64) -- key list, shapes, dtypes, and the full values of the small tensors -- for the
    unwrapped model (local_rank = -1) and for the DDP-wrapped one train.py always builds (RIFE.py:33-34;
    here wrapped on CPU over a one-rank gloo group, because `device_ids=[local_rank]` needs a GPU), plus
    what the reference's own `load_model` (RIFE.py:44-58) accepts."""
    import tempfile
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    _install_stubs()
    nd = int(os.environ.get("CKPT_ND", "3"))
    sys.path[:0] = [REF + "/Flow-%dD" % nd, REF]
    import model.RIFE as R
    import model.warplayer as WL
    R.device = torch.device("cpu")
    WL.device = torch.device("cpu")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29611")
    dist.init_process_group("gloo", rank=0, world_size=1)
    store = {}
    with tempfile.TemporaryDirectory() as d:
        torch.manual_seed(1234)
        m = _quiet(R.Model, local_rank=-1)
        _quiet(m.save_model, "plain.pkl", d)
        plain = torch.load(os.path.join(d, "plain.pkl"))
        m.flownet = DDP(m.flownet)  # what Model(local_rank >= 0) does (RIFE.py:33-34)
        _quiet(m.save_model, "ddp.pkl", d)
        wrapped = torch.load(os.path.join(d, "ddp.pkl"))
        store["plain_keys"] = np.array(list(plain.keys()))
        store["ddp_keys"] = np.array(list(wrapped.keys()))
        store["shapes"] = np.array(["x".join(str(int(n)) for n in v.shape) for v in wrapped.values()])
        store["dtypes"] = np.array([str(v.dtype) for v in wrapped.values()])
        store["sums"] = np.array([float(v.double().sum()) for v in wrapped.values()])
        small = [k for k, v in wrapped.items() if v.numel() <= 128]
        store["small_keys"] = np.array(small)
        for i, k in enumerate(small):
            store["small_%d" % i] = _np(wrapped[k])
        # the reference's loader: a DDP-wrapped model loads the DDP file; the plain file is filtered to an
        # empty dict (keys without "module.") and load_state_dict raises
        _quiet(m.load_model, "ddp.pkl", d)
        try:
            _quiet(m.load_model, "plain.pkl", d)
            store["ref_loads_plain"] = np.array(1)
        except Exception:
            store["ref_loads_plain"] = np.array(0)
    dist.destroy_process_group()
    np.savez_compressed(os.path.join(OUT, "ckpt_flow%dd.npz" % nd), **store)
    print("wrote ckpt_flow%dd.npz:" % nd, len(store["ddp_keys"]), "keys,", len(small), "small tensors; reference "
          "loads its own unprefixed file:", int(store["ref_loads_plain"]))


GROUPS = dict(flow3d_256=flow3d_256, flow3d_256_traj=flow3d_256_traj, ckpt=ckpt, rife_ops=rife_ops, upflow_ops=upflow_ops, upflow_next=upflow_next, rife_next=rife_next, flow3d_e2e=flow3d_e2e, flow2d_e2e=flow2d_e2e,
              upflow_e2e=upflow_e2e, upflow_sgu=upflow_sgu, upflow_levels=upflow_levels)

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    torch.set_num_threads(8)
    if which == "all":
        for g in GROUPS:
            if g in ("flow3d_256", "flow3d_256_traj") and os.environ.get("GOLDEN_WITH_256") != "1":
                continue  # ~35 GB of host memory and minutes: opt in
            if g == "ckpt":
                for nd in ("3", "2"):
                    subprocess.check_call([sys.executable, os.path.abspath(__file__), g],
                                          env=dict(os.environ, CKPT_ND=nd))
                continue
            subprocess.check_call([sys.executable, os.path.abspath(__file__), g])
    else:
        GROUPS[which]()
