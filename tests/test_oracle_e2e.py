"""The oracle's IFNet / Model.update restatement against end-to-end golden values captured from
the real reference (same seed => same initial weights; same inputs).  CPU only."""
import numpy as np
import torch

from oracle.ifnet_ref import ModelRef


def _psums(net):
    return np.array([float(p.detach().double().sum()) for p in net.parameters()])


def _run(nd, g):
    torch.manual_seed(1234)
    m = ModelRef(nd)
    assert sum(p.numel() for p in m.flownet.parameters()) == int(g["nparam"])
    np.testing.assert_allclose(_psums(m.flownet), g["param_sums"], rtol=0, atol=1e-9)  # seed-compatible init
    data = torch.from_numpy(g["data"])
    imgs, gt = data[:, :2], data[:, 2:3]
    m.flownet.eval()
    with torch.no_grad():
        merged, flows, mask = m.inference(imgs[:, :1], imgs[:, 1:2])
    if nd == 2:
        merged, mask = merged[2], mask[2]
    assert np.abs(merged.numpy() - g["inf_merged"]).max() < 2e-5
    assert np.abs(flows[2].numpy() - g["inf_flow2"]).max() < 2e-5
    assert np.abs(mask.numpy() - g["inf_mask"]).max() < 2e-5
    return m, imgs, gt


def test_flow3d_update_matches_reference(golden):
    g = golden("flow3d_e2e")
    m, imgs, gt = _run(3, g)
    for step in range(2):
        pred, info = m.update(imgs, gt, learning_rate=1e-4, training=True)
        got = [float(info[k]) for k in ("loss_l1", "loss_tea", "loss_distill", "loss_G")]
        np.testing.assert_allclose(got, g["update_losses"][step], rtol=2e-4, atol=1e-6)
    assert np.abs(pred.detach().numpy() - g["update_pred_last"]).max() < 1e-4
    np.testing.assert_allclose(_psums(m.flownet), g["param_sums_after"], rtol=1e-5, atol=1e-3)


def test_flow2d_update_matches_reference(golden):
    g = golden("flow2d_e2e")
    m, imgs, gt = _run(2, g)
    keys = [str(k) for k in g["update_loss_keys"]]
    for step in range(2):
        pred, info = m.update(imgs, gt, learning_rate=1e-4, training=True)
        got = [float(info[k]) for k in keys]
        np.testing.assert_allclose(got, g["update_losses"][step], rtol=2e-4, atol=1e-6)
    assert np.abs(pred.detach().numpy() - g["update_pred_last"]).max() < 1e-4
    np.testing.assert_allclose(_psums(m.flownet), g["param_sums_after"], rtol=1e-5, atol=1e-3)
