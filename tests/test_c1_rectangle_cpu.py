"""BASELINE config C1: 2-D synthetic rectangle pair through Flow-2D RIFE IFNet inference on CPU
PyTorch (plumbing, no GPU).  The CPU side is the oracle (the reference's CPU path restated); the
product never runs on the CPU."""
import torch

from opticalflowscivis_amd.data import synthetic
from oracle.ifnet_ref import ModelRef


def test_rectangle_generator_is_seeded_and_bounded():
    a, vx, vy = synthetic.rectangle2d_sequence(40, seed=1234)
    b, _, _ = synthetic.rectangle2d_sequence(40, seed=1234)
    c, _, _ = synthetic.rectangle2d_sequence(40, seed=7)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert a.shape == (40, 128, 128) and float(a.min()) == 0.0 and float(a.max()) <= 1.0
    # the box (60 x 80 of 10 x 10 tiles with values >= 30/255) is always fully inside the grid
    assert all(int((a[t] > 0).sum()) == 60 * 80 for t in range(40))
    assert float(vx.abs().max()) <= 6 and float(vy.abs().max()) <= 6


def test_flow2d_inference_on_rectangle_pair_cpu():
    trip = synthetic.rectangle2d_triplet(t=5, seed=1234)
    torch.manual_seed(1234)
    m = ModelRef(2)
    m.flownet.eval()
    with torch.no_grad():
        merged, flows, masks = m.inference(trip[:, :1], trip[:, 1:2])
    assert merged[2].shape == (1, 1, 128, 128) and flows[2].shape == (1, 4, 128, 128)
    assert torch.isfinite(merged[2]).all() and torch.isfinite(flows[2]).all()
    psnr = synthetic.psnr(merged[2], trip[:, 2:3])
    assert 0 < psnr < 100  # untrained net: just a finite, sane number
