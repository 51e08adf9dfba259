"""The oracle (oracle/) against golden vectors captured from the reference itself
(tests/golden/make_golden.py).  CPU only.  This is what pins the oracle."""
import numpy as np
import torch

from oracle import corr as ocorr
from oracle import losses as olosses
from oracle import warps as owarps


def T(a, grad=False):
    t = torch.from_numpy(a).clone()
    return t.requires_grad_() if grad else t


def close(a, b, atol, rtol=0.0):
    a = a.detach() if isinstance(a, torch.Tensor) else torch.as_tensor(a)
    b = torch.as_tensor(b)
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    assert bool((err <= tol).all()), "max err %.3e (tol %.1e)" % (float(err.max()), atol)


def _check_warp(fn, g, pre, names=("x", "f"), gnames=("gx", "gf"), atol=1e-6, gatol=1e-5, **kw):
    x, f = T(g[pre + names[0]], True), T(g[pre + names[1]], True)
    out = fn(x, f, **kw)
    close(out, g[pre + "out"], atol)
    gx, gf = torch.autograd.grad((out * T(g[pre + "G"])).sum(), [x, f])
    close(gx, g[pre + gnames[0]], gatol)
    close(gf, g[pre + gnames[1]], gatol, rtol=1e-5)


def test_warp2d_rife_ref(golden):
    g = golden("rife_ops")
    for tag in ("a", "b"):
        _check_warp(owarps.warp2d_rife_ref, g, "w2_%s_" % tag)
    close(owarps.warp2d_rife_ref(T(g["w2_zero_x"]), torch.zeros(1, 2, 8, 12)), g["w2_zero_out"], 1e-6)


def test_warp2d_rife_closed(golden):
    g = golden("rife_ops")
    for tag in ("a", "b"):
        x, f = T(g["w2_%s_x" % tag]), T(g["w2_%s_f" % tag])
        close(owarps.warp2d_rife_closed(x, f), g["w2_%s_out" % tag], 2e-5)


def test_warp3d_ref(golden):
    g = golden("rife_ops")
    for tag in ("nc", "cu", "tile", "mixed"):
        _check_warp(owarps.warp3d_ref, g, "w3_%s_" % tag)
    close(owarps.warp3d_ref(T(g["w3_zero_x"]), torch.zeros(1, 3, 5, 6, 7)), g["w3_zero_out"], 1e-6)


def test_warp3d_closed(golden):
    g = golden("rife_ops")
    for tag in ("nc", "cu", "tile"):
        x, f = T(g["w3_%s_x" % tag]), T(g["w3_%s_f" % tag])
        close(owarps.warp3d_closed(x, f), g["w3_%s_out" % tag], 5e-5)


def test_warp3d_axis_rotation_cubic(golden):
    """zero flow is NOT the identity, it rotates axes: out[d,h,w] = in[w,d,h]."""
    x = torch.rand(1, 1, 6, 6, 6)
    out = owarps.warp3d_ref(x, torch.zeros(1, 3, 6, 6, 6))
    close(out, x.permute(0, 1, 3, 4, 2), 2e-6)


def test_warp2d_pwc(golden):
    g = golden("upflow_ops")
    x, f = T(g["pwcmask_x"], True), T(g["pwcmask_f"], True)
    out = owarps.warp2d_pwc_ref(x, f, with_mask=True)
    close(out, g["pwcmask_out"], 1e-6)
    gx, gf = torch.autograd.grad((out * T(g["pwcmask_G"])).sum(), [x, f])
    close(gx, g["pwcmask_gx"], 1e-5)
    close(gf, g["pwcmask_gf"], 1e-5, 1e-5)
    out = owarps.warp2d_pwc_ref(x, f, with_mask=False)
    close(out, g["pwc_out"], 1e-6)
    gx, gf = torch.autograd.grad((out * T(g["pwc_G"])).sum(), [x, f])
    close(gx, g["pwc_gx"], 1e-5)
    close(gf, g["pwc_gf"], 1e-5, 1e-5)
    close(owarps.warp2d_pwc_closed(x.detach(), f.detach()), g["pwc_out"], 2e-5)


def test_warp2d_dilated(golden):
    g = golden("upflow_ops")
    for tag in ("s0", "s1"):
        pre = "dil_%s_" % tag
        I, f = T(g[pre + "I"], True), T(g[pre + "f"], True)
        out = owarps.warp2d_dilated_ref(I, f, T(g[pre + "start"]))
        close(out, g[pre + "out"], 1e-6)
        gI, gf = torch.autograd.grad((out * T(g[pre + "G"])).sum(), [I, f])
        close(gI, g[pre + "gI"], 1e-5)
        close(gf, g[pre + "gf"], 1e-5, 1e-5)


def test_corr2d(golden):
    g = golden("upflow_ops")
    for tag in ("c3", "c32", "tiny"):
        pre = "corr_%s_" % tag
        f1, f2 = T(g[pre + "f1"], True), T(g[pre + "f2"], True)
        for fn in (ocorr.corr2d_unfold_ref, ocorr.corr2d_closed):
            out = fn(f1, f2)
            close(out, g[pre + "out"], 2e-6)
            g1, g2 = torch.autograd.grad((out * T(g[pre + "G"])).sum(), [f1, f2])
            close(g1, g[pre + "g1"], 1e-5)
            close(g2, g[pre + "g2"], 1e-5)


def test_corr3d_degenerates_to_corr2d():
    f1, f2 = torch.randn(1, 4, 1, 6, 7), torch.randn(1, 4, 1, 6, 7)
    c3 = ocorr.corr3d_closed(f1, f2, md=2)  # [1,125,1,6,7]
    c2 = ocorr.corr2d_closed(f1[:, :, 0], f2[:, :, 0], md=2)  # [1,25,6,7]
    close(c3[:, 2 * 25:3 * 25, 0], c2, 1e-6)  # the dz = 0 plane
    assert float(c3[:, :2 * 25].abs().max()) == 0.0  # dz != 0 falls in the zero padding


def test_census_and_photo_losses(golden):
    g = golden("upflow_ops")
    occ = T(g["cen_occ"])
    for tag, (cha, useocc) in [("abs", (False, False)), ("absocc", (False, True)),
                               ("cha", (True, False)), ("chaocc", (True, True))]:
        im1, im2 = T(g["cen_im1"], True), T(g["cen_im2"], True)
        loss = olosses.census_loss(im1, im2, occ, 0.4, cha, useocc)
        close(loss, g["cen_%s_loss" % tag], 1e-5, 1e-5)
        g1, g2 = torch.autograd.grad(loss, [im1, im2])
        close(g1, g["cen_%s_g1" % tag], 1e-7, 1e-4)
        close(g2, g["cen_%s_g2" % tag], 1e-7, 1e-4)
    for typ in ["abs_robust", "charbonnier", "L1", "SSIM"]:
        for useocc in (False, True):
            tag = "%s_%d" % (typ, int(useocc))
            im1, im2 = T(g["cen_im1"], True), T(g["cen_im2"], True)
            loss = olosses.photo_loss_multi_type(im1, im2, occ, typ, 0.4, useocc)
            close(loss, g["photo_%s_loss" % tag], 1e-6, 1e-5)
            g1, g2 = torch.autograd.grad(loss, [im1, im2])
            close(g1, g["photo_%s_g1" % tag], 1e-8, 1e-4)
            close(g2, g["photo_%s_g2" % tag], 1e-8, 1e-4)


def test_occ_check(golden):
    """§8f.2: the oracle's occ_check_model against the reference's masks (bit-exact: same torch ops)."""
    g = golden("upflow_next")
    ff, fb = T(g["occ_ff"]), T(g["occ_fb"])
    for mode in ("all", "obj", "out"):
        for scale in (1, 4):
            of, ob = owarps.occ_check_ref(ff, fb, 0.1, 0.5, scale, mode)
            assert torch.equal(of, T(g["occ_%s_s%d_f" % (mode, scale)])), (mode, scale)
            assert torch.equal(ob, T(g["occ_%s_s%d_b" % (mode, scale)])), (mode, scale)
    # the fixture exercises both values of every mask
    for k in ("occ_all_s1_f", "occ_obj_s1_b", "occ_out_s1_f", "occ_out_s1_b"):
        m = float(np.mean(g[k]))
        assert 0.05 < m < 0.95, (k, m)


def test_lap_loss(golden):
    """§8f.3: the oracle's LapLoss restatement against the reference's value and gradients."""
    from oracle import ifnet_ref
    g = golden("rife_next")
    for tag in ("even", "odd", "l3", "c2"):
        a, b = T(g["lap_%s_a" % tag], True), T(g["lap_%s_b" % tag], True)
        loss = ifnet_ref.lap_loss(a, b, int(g["lap_%s_levels" % tag]))
        close(loss, g["lap_%s_loss" % tag], 1e-7, 1e-6)
        ga, gb = torch.autograd.grad(loss, [a, b])
        close(ga, g["lap_%s_ga" % tag], 1e-9, 1e-5)
        close(gb, g["lap_%s_gb" % tag], 1e-9, 1e-5)


def test_normalize_features(golden):
    """§8f.4: the oracle's normalize_features against the reference, all four flag combinations."""
    g = golden("upflow_next")
    for ch in (False, True):
        for im in (False, True):
            tag = "nf_c%d_i%d_" % (int(ch), int(im))
            f1, f2 = T(g["nf_f1"], True), T(g["nf_f2"], True)
            o1, o2 = ocorr.normalize_features((f1, f2), True, True, ch, im)
            close(o1, g[tag + "o1"], 1e-6, 1e-6)
            close(o2, g[tag + "o2"], 1e-6, 1e-6)
            g1, g2 = torch.autograd.grad((o1 * T(g[tag + "G1"])).sum() + (o2 * T(g[tag + "G2"])).sum(), [f1, f2])
            close(g1, g[tag + "g1"], 1e-5, 1e-5)
            close(g2, g[tag + "g2"], 1e-5, 1e-5)


def test_trajectory_fixture_continues_the_two_step_fixture(golden):
    """flow3d_256_traj.npz (eight reference steps at 256^3) starts with exactly the two steps of flow3d_256.npz: same
    process recipe, same seed, same input -- the reference's CPU path is deterministic here."""
    two, traj = golden("flow3d_256"), golden("flow3d_256_traj")
    assert int(traj["steps"]) == 8 and traj["update_losses"].shape == (8, 4)
    np.testing.assert_array_equal(traj["update_losses"][:2], two["update_losses"])
    np.testing.assert_array_equal(traj["param_sums"], two["param_sums"])
    np.testing.assert_array_equal(traj["data_sums"], two["data_sums"])
    assert np.all(np.isfinite(traj["param_sums_after4"])) and np.all(np.isfinite(traj["param_sums_after8"]))
    # training does move: loss_G falls by ~40 % over the eight steps
    assert traj["update_losses"][7, 3] < 0.7 * traj["update_losses"][0, 3]
