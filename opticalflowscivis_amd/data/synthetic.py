"""Seeded synthetic stand-ins for the reference's datasets.  There is no network and
the reference's pickles are not in the tree, so every benchmark / test input is generated here.
All generators are deterministic in (seed, shape) and produce float32 in [0, 1].
"""
import math

import torch


def _gen(seed, device):
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    return g


def droplet3d_batch(B, S, seed=1234, device="cpu", radius=(40, 80), max_shift=4.0):
    """Droplet-3D-like triplets [B,3,S,S,S]: a binary {0,1} sphere (Droplet-3D is 0/255 bytes,
    README.md:24) translating by <= max_shift voxels per frame; radius range is given for S=256 and
    scales with S.  Channels = (frame t, frame t+2, frame t+1): (img0, img1, gt) as
    Flow-3D/train.py:150-151 slices them."""
    g = _gen(seed, device)
    scale = S / 256.0
    r = (torch.rand(B, generator=g) * (radius[1] - radius[0]) + radius[0]) * scale
    c0 = torch.rand(B, 3, generator=g) * (S - 2 * r.max() - 4 * max_shift) + r.max() + 2 * max_shift
    v = (torch.rand(B, 3, generator=g) * 2 - 1) * max_shift
    ax = torch.arange(S, dtype=torch.float32, device=device)
    out = torch.empty(B, 3, S, S, S, dtype=torch.float32, device=device)
    for b in range(B):
        for slot, t in ((0, 0.0), (1, 2.0), (2, 1.0)):
            c = (c0[b] + v[b] * t).to(device)
            d2 = ((ax - c[0]) ** 2).view(S, 1, 1) + ((ax - c[1]) ** 2).view(1, S, 1) + \
                 ((ax - c[2]) ** 2).view(1, 1, S)
            out[b, slot] = (d2 <= float(r[b]) ** 2).float()
    return out


def jets3d_batch(B, S, seed=1234, device="cpu", njets=5, max_shift=4.0):
    """5Jets-like smooth density triplets [B,3,S,S,S]: a sum of `njets` anisotropic Gaussian
    plumes advected by <= max_shift voxels per frame, min-max normalised to [0,1]."""
    g = _gen(seed, device)
    ax = torch.linspace(0, 1, S, device=device)
    out = torch.empty(B, 3, S, S, S, dtype=torch.float32, device=device)
    for b in range(B):
        cen = torch.rand(njets, 3, generator=g) * 0.6 + 0.2
        sig = torch.rand(njets, 3, generator=g) * 0.08 + 0.04
        vel = (torch.rand(njets, 3, generator=g) * 2 - 1) * max_shift / S
        for slot, t in ((0, 0.0), (1, 2.0), (2, 1.0)):
            vol = torch.zeros(S, S, S, device=device)
            for j in range(njets):
                c = cen[j] + vel[j] * t
                e = ((ax - float(c[0])) / float(sig[j, 0])).pow(2).view(S, 1, 1) + \
                    ((ax - float(c[1])) / float(sig[j, 1])).pow(2).view(1, S, 1) + \
                    ((ax - float(c[2])) / float(sig[j, 2])).pow(2).view(1, 1, S)
                vol += torch.exp(-0.5 * e)
            out[b, slot] = vol
        lo, hi = out[b].min(), out[b].max()
        out[b] = (out[b] - lo) / (hi - lo + 1e-12)
    return out


def droplet2d_batch(B, H=160, W=224, seed=1234, device="cpu", radius=(20, 40), max_shift=4.0):
    """Droplet-2D-like triplets [B,3,H,W]: a disc translating <= max_shift px/frame, blurred with a
    sigma=1 Gaussian; channels (t, t+2, t+1)."""
    g = _gen(seed, device)
    r = torch.rand(B, generator=g) * (radius[1] - radius[0]) + radius[0]
    cy = torch.rand(B, generator=g) * (H - 2 * radius[1] - 4 * max_shift) + radius[1] + 2 * max_shift
    cx = torch.rand(B, generator=g) * (W - 2 * radius[1] - 4 * max_shift) + radius[1] + 2 * max_shift
    v = (torch.rand(B, 2, generator=g) * 2 - 1) * max_shift
    ys = torch.arange(H, dtype=torch.float32, device=device).view(H, 1)
    xs = torch.arange(W, dtype=torch.float32, device=device).view(1, W)
    k = torch.exp(-0.5 * (torch.arange(-3, 4, dtype=torch.float32, device=device)) ** 2)
    k = (k / k.sum())
    out = torch.empty(B, 3, H, W, dtype=torch.float32, device=device)
    for b in range(B):
        for slot, t in ((0, 0.0), (1, 2.0), (2, 1.0)):
            d2 = (ys - float(cy[b] + v[b, 1] * t)) ** 2 + (xs - float(cx[b] + v[b, 0] * t)) ** 2
            out[b, slot] = (d2 <= float(r[b]) ** 2).float()
    flat = out.view(B * 3, 1, H, W)
    flat = torch.nn.functional.conv2d(torch.nn.functional.pad(flat, (3, 3, 0, 0), mode="replicate"),
                                      k.view(1, 1, 1, 7))
    flat = torch.nn.functional.conv2d(torch.nn.functional.pad(flat, (0, 0, 3, 3), mode="replicate"),
                                      k.view(1, 1, 7, 1))
    return flat.view(B, 3, H, W).clamp(0, 1)


def vortex2d_pairs(B, H=150, W=450, seed=0, device="cpu", nvort=8, max_shift=3.0):
    """Cylinder-ensemble-like pairs for UPFlow [B,2,3,H,W]: a sum of `nvort` Gaussian vortices
    advected between the two frames, min-max normalised, grey replicated to 3 channels
    (UPFlow/model/upflow.py:384-386)."""
    g = _gen(seed, device)
    ys = torch.arange(H, dtype=torch.float32, device=device).view(H, 1)
    xs = torch.arange(W, dtype=torch.float32, device=device).view(1, W)
    out = torch.empty(B, 2, 3, H, W, dtype=torch.float32, device=device)
    for b in range(B):
        cx = torch.rand(nvort, generator=g) * W
        cy = torch.rand(nvort, generator=g) * H
        sg = torch.rand(nvort, generator=g) * 18 + 8
        am = torch.rand(nvort, generator=g) * 2 - 1
        vx = torch.rand(nvort, generator=g) * max_shift
        vy = (torch.rand(nvort, generator=g) * 2 - 1) * 0.5 * max_shift
        for t in (0, 1):
            f = torch.zeros(H, W, device=device)
            for j in range(nvort):
                d2 = (xs - float(cx[j] + vx[j] * t)) ** 2 + (ys - float(cy[j] + vy[j] * t)) ** 2
                f += float(am[j]) * torch.exp(-0.5 * d2 / float(sg[j]) ** 2)
            out[b, t] = f
        lo, hi = out[b].min(), out[b].max()
        out[b] = (out[b] - lo) / (hi - lo + 1e-12)
    return out


def psnr(pred, gt):
    """PSNR on [0,1] data: -10 log10(mean((pred-gt)^2)) (Flow-3D/train.py:385)."""
    mse = torch.mean((pred.double() - gt.double()) ** 2)
    return float(-10.0 * math.log10(max(float(mse), 1e-20)))
