"""Oracle (TEST INFRASTRUCTURE): photometric / census losses and the merge / distillation
epilogues of the reference, restated on CPU PyTorch fp32."""
import numpy as np
import torch
import torch.nn.functional as F

from .warps import warp2d_photo_ref


# --------------------------------------------------------------------------------------------
# a10  UPFlow/utils/loss.py:17-48  photo_loss_function
# --------------------------------------------------------------------------------------------
def photo_loss_function(diff, mask, q, charbonnier_or_abs_robust, if_use_occ, averge=True):
    if charbonnier_or_abs_robust:
        if if_use_occ:  # :20-29
            p = ((diff) ** 2 + 1e-6).pow(q) * mask
            if averge:
                p, ap = p.mean(), mask.mean()
            else:
                p, ap = p.sum(), mask.sum()
            return p / (ap * 2 + 1e-6)
        p = ((diff) ** 2 + 1e-8).pow(q)  # :31-36
        return p.mean() if averge else p.sum()
    if if_use_occ:  # :38-42
        d = (torch.abs(diff) + 0.01).pow(q) * mask
        return torch.sum(d) / (torch.sum(mask) * 2 + 1e-6)
    d = (torch.abs(diff) + 0.01).pow(q)  # :44-48
    return d.mean() if averge else d.sum()


# --------------------------------------------------------------------------------------------
# a8  UPFlow/utils/loss.py:51-91  census_loss_torch
# --------------------------------------------------------------------------------------------
def census_dist(img1, img1_warp, max_distance=3):
    """The per-pixel soft Hamming distance between the soft ternary transforms, [B,1,H,W]."""
    patch = 2 * max_distance + 1

    def ternary(image):
        R, G, B = torch.split(image, 1, 1)
        gray = 0.2989 * R + 0.5870 * G + 0.1140 * B  # :56
        oc = patch * patch
        # :59-63 identity convolution = gather of the patch x patch neighbourhood, zero padded
        w = np.eye(oc).reshape((patch, patch, 1, oc))
        weight = torch.from_numpy(np.transpose(w, (3, 2, 0, 1))).float().to(image.device)
        patches = torch.conv2d(gray, weight, None, [1, 1], [max_distance, max_distance])
        t = patches - gray  # :65
        return t / torch.sqrt(0.81 + t ** 2)  # :66

    t1, t2 = ternary(img1), ternary(img1_warp)
    d = (t1 - t2) ** 2  # :70
    return torch.sum(d / (0.1 + d), 1, keepdim=True)  # :71


def census_loss(img1, img1_warp, mask, q, charbonnier_or_abs_robust, if_use_occ, averge=True,
                max_distance=3):
    dist = census_dist(img1, img1_warp, max_distance)
    # create_mask_torch :74-82 -- inner ones padded back with zeros.  NOTE the reference pads
    # [p00, p01] on the LAST dim and [p10, p11] on dim 2; symmetric here, so it is a border mask.
    B, c, H, W = mask.shape
    m = max_distance
    inner = torch.ones(B, c, H - 2 * m, W - 2 * m, dtype=mask.dtype, device=mask.device)
    tmask = F.pad(inner, [m, m, m, m])
    return photo_loss_function(dist, mask * tmask, q, charbonnier_or_abs_robust, if_use_occ, averge)


# --------------------------------------------------------------------------------------------
# a9  UPFlow/model/upflow.py:141-196, 267-289
# --------------------------------------------------------------------------------------------
def weighted_ssim(x, y, weight, c1=float('inf'), c2=9e-6, weight_epsilon=0.01):
    pool = lambda z: F.avg_pool2d(z, (3, 3), (1, 1))  # :164-167
    apw = pool(weight)
    wpe = weight + weight_epsilon
    inv = 1.0 / (apw + weight_epsilon)
    wpool = lambda z: pool(z * wpe) * inv  # :175-177
    mu_x, mu_y = wpool(x), wpool(y)
    sigma_x = wpool(x ** 2) - mu_x ** 2
    sigma_y = wpool(y ** 2) - mu_y ** 2
    sigma_xy = wpool(x * y) - mu_x * mu_y
    if c1 == float('inf'):
        n, d = (2 * sigma_xy + c2), (sigma_x + sigma_y + c2)
    elif c2 == float('inf'):
        n, d = 2 * mu_x * mu_y + c1, mu_x ** 2 + mu_y ** 2 + c1
    else:
        n = (2 * mu_x * mu_y + c1) * (2 * sigma_xy + c2)
        d = (mu_x ** 2 + mu_y ** 2 + c1) * (sigma_x + sigma_y + c2)
    return torch.clamp((1 - n / d) / 2, 0, 1), apw  # :195-196


def photo_loss_multi_type(x, y, occ_mask, photo_loss_type='abs_robust', photo_loss_delta=0.4,
                          photo_loss_use_occ=False):
    occ_weight = occ_mask
    if photo_loss_type == 'abs_robust':
        loss_diff = (torch.abs(x - y) + 0.01).pow(photo_loss_delta)  # :272-274
    elif photo_loss_type == 'charbonnier':
        loss_diff = ((x - y) ** 2 + 1e-6).pow(photo_loss_delta)  # :275-277
    elif photo_loss_type == 'L1':
        loss_diff = torch.abs(x - y + 1e-6)  # :278-280
    elif photo_loss_type == 'SSIM':
        loss_diff, occ_weight = weighted_ssim(x, y, occ_mask)  # :281-282
    else:
        raise ValueError('wrong photo_loss type: %s' % photo_loss_type)
    if photo_loss_use_occ:
        return torch.sum(loss_diff * occ_weight) / (torch.sum(occ_weight) + 1e-6)  # :286-287
    return torch.mean(loss_diff)  # :289


# --------------------------------------------------------------------------------------------
# a11  Flow-2D/model/RIFE.py:190-191, 244-278
# --------------------------------------------------------------------------------------------
def rife2d_photometric(flow4, merged, img0, img1):
    """loss_photo of Model.update: two backward warps of `merged` + Charbonnier, averaged."""
    def charbonnier(x, alpha=0.25, epsilon=1.e-9):  # :190-191
        return torch.pow(torch.pow(x, 2) + epsilon ** 2, alpha)

    def photometric_loss(wraped, frame1):  # :267-272
        h, w = wraped.shape[2:]
        frame1 = F.interpolate(frame1, (h, w), mode='bilinear', align_corners=False)
        p = torch.sum(charbonnier(wraped - frame1), dim=1) / 3
        return torch.sum(p) / frame1.size(0)

    loss = photometric_loss(warp2d_photo_ref(merged, flow4[:, 2:4]), img0)  # :274-275
    loss = loss + photometric_loss(warp2d_photo_ref(merged, flow4[:, :2]), img1)  # :277-278
    return loss / 2  # :279


# --------------------------------------------------------------------------------------------
# a12  Flow-2D/model/IFNet.py:239-248, Flow-3D/model/IFNet.py:241-267
# --------------------------------------------------------------------------------------------
def merge(w0, w1, mask_logit):
    """merged = w0 * sigmoid(m) + w1 * (1 - sigmoid(m))."""
    s = torch.sigmoid(mask_logit)
    return w0 * s + w1 * (1 - s)


def distill_term(merged_i, merged_teacher, gt, flow_i, flow_teacher):
    """One block's contribution to loss_distill (IFNet.py:244-246 / Flow-3D :261)."""
    loss_mask = ((merged_i - gt).abs().mean(1, True) >
                 (merged_teacher - gt).abs().mean(1, True) + 0.01).float().detach()
    return (((flow_teacher.detach() - flow_i) ** 2).mean(1, True) ** 0.5 * loss_mask).mean()
