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
        got = [float(info[k]) for k in keys]
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
