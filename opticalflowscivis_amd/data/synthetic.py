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


def rectangle2d_sequence(n_frames=64, seed=1234, grid=(128, 128), box=(60, 80), tile=10, vel=(-6, 6),
                         max_seq=15):
    """Seeded restatement of Datasets/create_rectangle_2d.py:81-199 (BASELINE config C1): a box of
    `tile`-sized patches with values randint(30,256)/255 bouncing on a `grid`; the velocity is
    re-drawn every `max_seq` steps or when a wall is hit.  Keeps the original's axis swap
    (pos_x += vel_y, pos_y += vel_x, :165-167).  The original is unseeded and blocks on input();
    here numpy's and random's streams are replaced by one seeded numpy Generator.
    Returns (frames [T,H,W] float32 in [0,1], vel_x [T,H,W], vel_y [T,H,W])."""
    import numpy as np
    rng = np.random.default_rng(seed)
    gx, gy = grid
    bx, by = box
    b = np.ones((bx, by), dtype=np.float32)
    for i in range(0, bx, tile):
        for j in range(0, by, tile):
            b[i:i + tile, j:j + tile] = rng.integers(30, 256) / 255.0
    frames = np.zeros((n_frames, gx, gy), dtype=np.float32)
    vxs = np.zeros_like(frames)
    vys = np.zeros_like(frames)
    px, py = int(rng.integers(0, gx - bx + 1)), int(rng.integers(0, gy - by + 1))
    vx, vy = int(rng.integers(vel[0], vel[1] + 1)), int(rng.integers(vel[0], vel[1] + 1))
    seq = max_seq
    for t in range(n_frames):
        if seq == 0:
            vx, vy = int(rng.integers(vel[0], vel[1] + 1)), int(rng.integers(vel[0], vel[1] + 1))
            seq = max_seq
        px = min(max(px + vy, 0), gx - bx)
        py = min(max(py + vx, 0), gy - by)
        frames[t, px:px + bx, py:py + by] = b
        vxs[t, px:px + bx, py:py + by] = vx
        vys[t, px:px + bx, py:py + by] = vy
        seq -= 1
        if px == 0 or py == 0 or px == gx - bx or py == gy - by:
            seq = 0
    return torch.from_numpy(frames), torch.from_numpy(vxs), torch.from_numpy(vys)


def rectangle2d_triplet(t=0, seed=1234):
    """(img0, img1, gt) = frames (t, t+2, t+1) as [1,3,128,128] (C1: pair + middle frame)."""
    f, _, _ = rectangle2d_sequence(t + 3, seed)
    return torch.stack([f[t], f[t + 2], f[t + 1]], 0).unsqueeze(0)
