"""Oracle (TEST INFRASTRUCTURE): local-window correlation (cost volume) on CPU PyTorch fp32.

Reference boundary: `CorrelationFunction` -> `correlation_cuda.forward/backward`
(UPFlow/model/correlation_package/correlation.py:8-45).  The CUDA extension's sources are NOT in
the reference tree (NVIDIA flownet2-pytorch correlation_package, unpinned), so its arithmetic is
restated from its published definition and from the reference's own PyTorch stand-in
`Corr_pyTorch` (UPFlow/utils/pytorch_correlation.py:27-50), which the reference uses
interchangeably (UPFlow/model/upflow.py:643-652).  Parity is pinned against `Corr_pyTorch`
golden vectors; it is UNPINNED at the CUDA-extension boundary itself.

Published definition (kernel_size=1, stride1=stride2=1, pad=max_displacement=md):
  out[b, (dy+md)*(2md+1) + (dx+md), y, x] = (1/C) * sum_c f1[b,c,y,x] * f2[b,c,y+dy,x+dx]
with f2 read as 0 outside the image; dy-major channel order.
"""
import torch
import torch.nn.functional as F


def corr2d_unfold_ref(in1, in2, md=4):
    """Follows Corr_pyTorch.forward (pytorch_correlation.py:27-50) for kernel_size=1."""
    bz, cn, hei, wid = in1.shape
    f1 = F.unfold(in1, kernel_size=1, padding=0, stride=1)  # :30  [B, C, H*W]
    f2 = F.unfold(in2, kernel_size=1, padding=0, stride=1)  # :31
    k = f2.shape[1]
    f2_ = f2.reshape(bz * k, hei, wid).unsqueeze(1)  # :35-36
    # :38 a second unfold with an (H, W) kernel and `md` padding enumerates the (2md+1)^2 shifts
    f2 = F.unfold(f2_, kernel_size=(hei, wid), padding=md, stride=1)
    _, kernel_number, window_number = f2.shape
    f2_ = f2.reshape(bz, k, kernel_number, window_number)
    f2_2 = f2_.transpose(1, 3).transpose(2, 3)  # :42  [B, win, C, H*W]
    res = f2_2 * f1.unsqueeze(1)  # :46
    res = torch.mean(res, dim=2)  # :47 channel MEAN
    return res.reshape(bz, window_number, hei, wid)  # :48


def corr2d_closed(f1, f2, md=4):
    """The published definition, shift by shift."""
    B, C, H, W = f1.shape
    f2p = F.pad(f2, (md, md, md, md))
    outs = []
    for dy in range(-md, md + 1):
        for dx in range(-md, md + 1):
            sh = f2p[:, :, md + dy:md + dy + H, md + dx:md + dx + W]
            outs.append((f1 * sh).mean(dim=1))
    return torch.stack(outs, dim=1)


def corr3d_closed(f1, f2, md=4):
    """This build's generalisation to volumes (NOT in the reference): dz-major, then dy, dx;
    (2md+1)^3 channels, channel mean, zero padding.  Pinned to the reference only through
    D == 1 slices (must equal corr2d on the central dz plane)."""
    B, C, D, H, W = f1.shape
    f2p = F.pad(f2, (md, md, md, md, md, md))
    outs = []
    for dz in range(-md, md + 1):
        for dy in range(-md, md + 1):
            for dx in range(-md, md + 1):
                sh = f2p[:, :, md + dz:md + dz + D, md + dy:md + dy + H, md + dx:md + dx + W]
                outs.append((f1 * sh).mean(dim=1))
    return torch.stack(outs, dim=1)


# --------------------------------------------------------------------------------------------
# §8f.4  UPFlow/model/upflow.py:96-138 network_tools.normalize_features
# --------------------------------------------------------------------------------------------
def normalize_features(feature_list, normalize=True, center=True, moments_across_channels=True,
                       moments_across_images=True):
    axes = [1, 2, 3] if moments_across_channels else [2, 3]                       # :113
    means = [torch.mean(f, dim=axes, keepdim=True) for f in feature_list]         # :115
    variances = [torch.var(f, dim=axes, keepdim=True) for f in feature_list]      # :116 (unbiased)
    if moments_across_images:                                                     # :120-126
        means = [torch.mean(torch.stack(means, dim=0), dim=(0,))] * len(feature_list)
        # the reference takes the VARIANCE of the per-image variances here (:126), not their mean
        variances = [torch.var(torch.stack(variances, dim=0), dim=(0,))] * len(feature_list)
    stds = [torch.sqrt(v + 1e-16) for v in variances]                             # :128
    if center:
        feature_list = [f - m for f, m in zip(feature_list, means)]               # :132-135
    if normalize:
        feature_list = [f / s for f, s in zip(feature_list, stds)]                # :136-137
    return feature_list


def corr2d_normalized_ref(f1, f2, md=4):
    """The C3 path: per-plane moments (both flags False), then the cost volume (upflow.py:635-652)."""
    n1, n2 = normalize_features((f1, f2), True, True, False, False)
    return corr2d_closed(n1, n2, md)
