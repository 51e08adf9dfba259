"""Drop-in for Flow-3D/inference_img.py: interpolate 2**exp - 1 frames between two inputs by
recursive bisection (inference_img.py:88-97).  Inputs / outputs are .npy arrays in [0,1]
(the reference reads PNG/EXR through cv2, which this image does not have); spatial sizes are padded
to a multiple of 32 like the reference (:56-61)."""
import argparse
import os

import numpy as np
import torch
import torch.nn.functional as F

from .model.RIFE import Model

ND = 3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--img', nargs=2, required=True, help='two .npy frames/volumes')
    ap.add_argument('--exp', default=1, type=int)
    ap.add_argument('--model', default='train_log', help='directory holding flownet.pkl')
    ap.add_argument('--out', default='output')
    args = ap.parse_args()
    dev = torch.device('cuda')
    model = Model(-1, device=dev)
    try:
        model.load_model('flownet.pkl', args.model)
    except FileNotFoundError:
        print('no flownet.pkl under %s: using random-init weights' % args.model)
    model.eval()
    a, b = (torch.from_numpy(np.load(p).astype(np.float32)).to(dev) for p in args.img)
    a, b = a.reshape((1, 1) + a.shape[-ND:]), b.reshape((1, 1) + b.shape[-ND:])
    sp = a.shape[2:]
    pad = []
    for s in reversed(sp):
        pad += [0, ((s - 1) // 32 + 1) * 32 - s]
    a, b = F.pad(a, pad), F.pad(b, pad)
    frames = [a, b]
    with torch.no_grad():
        for _ in range(args.exp):
            nxt = []
            for x, y in zip(frames[:-1], frames[1:]):
                mid = model.inference(x, y)[0]
                mid = mid[2] if isinstance(mid, list) else mid
                nxt += [x, mid]
            frames = nxt + [frames[-1]]
    os.makedirs(args.out, exist_ok=True)
    cut = (0, 0) + tuple(slice(0, s) for s in sp)
    for i, f in enumerate(frames):
        np.save(os.path.join(args.out, 'img%d.npy' % i), f[cut].cpu().numpy())


if __name__ == '__main__':
    main()
