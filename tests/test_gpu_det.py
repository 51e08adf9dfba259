"""-m gpu: deterministic weight gradients (fs_conv3d_wrw_det, round 4).  The default weight-gradient kernels finish with float
atomics (order of arrival decides the last bits); under torch.use_deterministic_algorithms(True) the binding runs the same
kernels with per-run copies of dW in a workspace and a fixed-order reduce launch: bitwise reproducible, and with the other
kernels of the Flow-3D step already order-fixed, so is the whole training step."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture
def deterministic():
    torch.use_deterministic_algorithms(True)
    yield
    torch.use_deterministic_algorithms(False)


# (B, Cg, Cs, g extent, src extent, k, stride, pad): the Winograd F(4,3) trunk layer, loader-wave k3 / k4 layers incl. a
# 16-column one and the 1-channel mask head, and shapes only the register-staged brick kernel takes
CASES = [(2, 64, 64, (32, 32, 64), (32, 32, 64), 3, 1, 1), (2, 64, 64, (16, 16, 32), (16, 16, 32), 3, 1, 1),
         (1, 128, 128, (16, 16, 16), (16, 16, 16), 3, 1, 1), (2, 32, 11, (32, 32, 32), (64, 64, 64), 4, 2, 1),
         (2, 64, 32, (16, 16, 32), (32, 32, 64), 4, 2, 1), (1, 32, 1, (16, 32, 32), (32, 64, 64), 4, 2, 1),
         (1, 20, 7, (9, 10, 11), (9, 10, 11), 3, 1, 1), (1, 6, 32, (10, 12, 14), (20, 24, 28), 4, 2, 1)]


@pytest.mark.parametrize("case", CASES)
def test_wrw_det_is_reproducible_and_equals_the_atomic_form(case, deterministic):
    from opticalflowscivis_amd import ops
    B, Cg, Cs, gd, sd, k, s, p = case
    gen = torch.Generator().manual_seed(Cg * 100 + Cs)
    g = torch.randn((B, Cg) + gd, generator=gen).to(DEV)
    src = torch.randn((B, Cs) + sd, generator=gen).to(DEV)
    a = ops.conv3d_wrw(g, src, k, s, p)
    b = ops.conv3d_wrw(g, src, k, s, p)
    assert torch.equal(a, b)
    torch.use_deterministic_algorithms(False)
    c = ops.conv3d_wrw(g, src, k, s, p)   # float atomics into a zero-filled dW
    scale = float(c.abs().max())
    assert float((a - c).abs().max()) <= 2e-5 * scale, (case, float((a - c).abs().max()), scale)


def test_flow3d_training_is_bitwise_reproducible_under_the_flag(deterministic):
    from opticalflowscivis_amd.data import synthetic
    from opticalflowscivis_amd.flow3d.model.RIFE import Model
    data = synthetic.droplet3d_batch(2, 64, seed=11, device=DEV)
    imgs, gt = data[:, :2].contiguous(), data[:, 2:3].contiguous()

    def run():
        torch.manual_seed(77)
        m = Model(local_rank=-1, device=DEV)
        losses = [float(m.update(imgs, gt, learning_rate=1e-4, training=True)[1]["loss_G"].detach()) for _ in range(3)]
        return losses, [p.detach().clone() for p in m.flownet.parameters()]

    la, pa = run()
    lb, pb = run()
    assert la == lb, (la, lb)
    assert all(torch.equal(x, y) for x, y in zip(pa, pb))
    assert abs(la[0] - la[2]) > 1e-6 * abs(la[0])  # the three steps did move the weights
