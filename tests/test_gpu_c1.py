"""-m gpu: BASELINE config C1 (2-D synthetic rectangle pair through Flow-2D RIFE IFNet inference) on the HIP path,
against the CPU side of the same workload -- the oracle's `inference` from the same seed on the same pair
(tests/test_c1_rectangle_cpu.py runs that side alone).  Reference: Datasets/create_rectangle_2d.py:81-199 (the
generator, restated seeded in data/synthetic.py), Flow-2D/model/RIFE.py:66-78 (`inference`),
Flow-2D/inference_img.py:90-97 (recursive bisection)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pair():
    from opticalflowscivis_amd.flow2d.model.RIFE import Model
    from oracle.ifnet_ref import ModelRef
    torch.manual_seed(1234)
    m = Model(local_rank=-1, device=DEV)
    torch.manual_seed(1234)
    o = ModelRef(2)
    m.eval()
    o.flownet.eval()
    return m, o


@pytest.mark.parametrize("t", [5, 14, 33])  # a frame before, at and after a velocity re-draw (every 15 steps)
def test_flow2d_inference_on_rectangle_pair_matches_the_cpu_side(t):
    from opticalflowscivis_amd.data import synthetic
    trip = synthetic.rectangle2d_triplet(t=t, seed=1234)
    m, o = _pair()
    with torch.no_grad():
        om, of, omask = o.inference(trip[:, :1], trip[:, 1:2])
        gm, gf, gmask = m.inference(trip[:, :1].to(DEV), trip[:, 1:2].to(DEV), [4, 2, 1])
    torch.cuda.synchronize()
    gm = gm[2] if isinstance(gm, (list, tuple)) else gm
    for i in range(3):  # the flow after every student block, in pixels
        assert float((gf[i].cpu() - of[i]).abs().max()) < 1e-4, i
    assert float((gm.cpu() - om[2]).abs().max()) < 2e-5
    gt = trip[:, 2:3]
    assert abs(synthetic.psnr(gm.cpu(), gt) - synthetic.psnr(om[2], gt)) < 0.01


def test_inference_img_bisection_on_the_rectangle_pair(tmp_path):
    """`inference_img --exp 2` (a fresh child process, random-init weights from seed 1234): img0 / img4 are the inputs,
    img2 = inference(a, b), img1 = inference(a, img2), img3 = inference(img2, b) -- against the oracle's bisection."""
    from opticalflowscivis_amd.data import synthetic
    trip = synthetic.rectangle2d_triplet(t=5, seed=1234)
    a, b = trip[:, :1], trip[:, 1:2]
    pa, pb = str(tmp_path / "a.npy"), str(tmp_path / "b.npy")
    np.save(pa, a[0, 0].numpy())
    np.save(pb, b[0, 0].numpy())
    out = str(tmp_path / "out")
    code = ("import sys, torch; torch.manual_seed(1234); "
            "from opticalflowscivis_amd.flow2d import inference_img as I; "
            "sys.argv = ['inference_img', '--img', %r, %r, '--exp', '2', '--model', %r, '--out', %r]; I.main()"
            % (pa, pb, str(tmp_path / "nomodel"), out))
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    got = [torch.from_numpy(np.load(os.path.join(out, "img%d.npy" % i))) for i in range(5)]
    _, o = _pair()
    with torch.no_grad():
        mid = o.inference(a, b)[0][2]
        q1 = o.inference(a, mid)[0][2]
        q3 = o.inference(mid, b)[0][2]
    want = [a, q1, mid, q3, b]
    for i, (g, w) in enumerate(zip(got, want)):
        assert g.shape == (128, 128)
        assert float((g - w[0, 0]).abs().max()) < (1e-7 if i in (0, 4) else 5e-5), i
