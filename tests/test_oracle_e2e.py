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


def test_upflow_mirror_on_oracle_ops_matches_reference(golden):
    """The product's UPFlow network code (feature pyramid, estimators, context networks, loss assembly -- stock torch
    modules) with its seven hot-path ops swapped for the oracle's CPU restatements (oracle/upflow_port.py::cpu_ops)
    against the reference's own forward + backward on the same seed-0 weights and inputs.  On the CPU both sides run
    the same ATen kernels, so this is an EPSILON test of everything in the UPFlow step that is not a HIP kernel --
    which the GPU end-to-end comparison (tests/test_gpu_e2e.py, a band because of MIOpen's run-to-run noise and
    fp32-borderline validity masks) cannot give."""
    from opticalflowscivis_amd.upflow.model.upflow import UPFlow_net
    from oracle.upflow_port import cpu_ops
    g = golden("upflow_e2e")
    conf = UPFlow_net.config()
    conf.update({'if_norm_before_cost_volume': True, 'norm_moments_across_channels': False,
                 'norm_moments_across_images': False, 'photo_loss_census_weight': 1,
                 'multi_scale_distillation_weight': 1})
    torch.manual_seed(0)
    net = conf()
    np.testing.assert_allclose(_psums(net), g["param_sums"], rtol=0, atol=1e-9)
    with cpu_ops():
        out = net({'im1': torch.from_numpy(g["im1"]), 'im2': torch.from_numpy(g["im2"]), 'if_loss': True})
        keys = [str(k) for k in g["loss_keys"]]
        got = np.array([float(out['loss_dict'][k].detach()) for k in keys])
        np.testing.assert_allclose(got, g["losses"], rtol=1e-6)  # measured here: bit-identical
        assert float((out['flow_f_out'].detach() - torch.from_numpy(g["flow_f_out"])).abs().max()) < 1e-5
        assert float((out['flow_b_out'].detach() - torch.from_numpy(g["flow_b_out"])).abs().max()) < 1e-5
        assert float((out['occ_fw'] != torch.from_numpy(g["occ_fw"])).float().mean()) == 0.0
        assert float((out['im1_warp'].detach() - torch.from_numpy(g["im1_warp"])).abs().max()) < 1e-6
        sum(out['loss_dict'][k] for k in keys).backward()
    gsum = np.array([float(p.grad.detach().double().abs().sum()) if p.grad is not None else 0.0
                     for p in net.parameters()])
    np.testing.assert_allclose(gsum, g["grad_abs_sums"], rtol=1e-5, atol=1e-9)


def test_upflow_sgu_mirror_on_oracle_ops_matches_reference(golden):
    """The self-guided upsampling variant (`if_sgu_upsample=True`, UPFlow/model/upflow.py:21-92, 612-616, 629-631): the
    mirror's `sgu_model` -- same module tree, same construction order -- with the hot-path ops swapped for the oracle's
    CPU restatements against the reference's own forward + backward (tests/golden/upflow_sgu.npz, seed 0, 96 x 128)."""
    from opticalflowscivis_amd.upflow.model.upflow import UPFlow_net
    from oracle.upflow_port import cpu_ops
    g = golden("upflow_sgu")
    conf = UPFlow_net.config()
    conf.update({'if_norm_before_cost_volume': True, 'norm_moments_across_channels': False,
                 'norm_moments_across_images': False, 'photo_loss_census_weight': 1,
                 'multi_scale_distillation_weight': 1, 'if_sgu_upsample': True})
    torch.manual_seed(0)
    net = conf()
    assert [n for n, _ in net.named_parameters()] == [str(n) for n in g["param_names"]]  # checkpoints carry over
    assert sum(p.numel() for p in net.parameters()) == int(g["nparam"])
    np.testing.assert_allclose(_psums(net), g["param_sums"], rtol=0, atol=1e-9)      # and so do seeds
    with cpu_ops():
        out = net({'im1': torch.from_numpy(g["im1"]), 'im2': torch.from_numpy(g["im2"]), 'if_loss': True})
        keys = [str(k) for k in g["loss_keys"]]
        got = np.array([float(out['loss_dict'][k].detach()) for k in keys])
        np.testing.assert_allclose(got, g["losses"], rtol=1e-6)
        assert float((out['flow_f_out'].detach() - torch.from_numpy(g["flow_f_out"])).abs().max()) < 1e-5
        assert float((out['flow_b_out'].detach() - torch.from_numpy(g["flow_b_out"])).abs().max()) < 1e-5
        assert float((out['occ_fw'] != torch.from_numpy(g["occ_fw"])).float().mean()) == 0.0
        assert float((out['im1_warp'].detach() - torch.from_numpy(g["im1_warp"])).abs().max()) < 1e-6
        sum(out['loss_dict'][k] for k in keys).backward()
    gsum = np.array([float(p.grad.detach().double().abs().sum()) if p.grad is not None else 0.0
                     for p in net.parameters()])
    np.testing.assert_allclose(gsum, g["grad_abs_sums"], rtol=1e-5, atol=1e-9)
