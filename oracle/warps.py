"""Oracle (TEST INFRASTRUCTURE): the reference's backward warps restated on CPU PyTorch fp32.

Two flavours per warp:
  *_ref     follows the reference line by line (same ATen calls, same arithmetic order);
            this is what the reference's CPU path executes and what `cpu_baseline` times.
  *_closed  the closed-form sampling rule of SURVEY §8(a) as an explicit gather, with no
            grid_sample call -- an independent statement of the same function.
Both are checked against golden vectors captured from the reference (tests/golden).
"""
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------
# a1  Flow-2D/model/warplayer.py:7-26
# --------------------------------------------------------------------------------------------
def warp2d_rife_ref(tenInput, tenFlow):
    B, _, H, W = tenFlow.shape
    # :12-17 base grid: linspace(-1,1) along W (x) and along H (y)
    gx = torch.linspace(-1.0, 1.0, W, dtype=tenFlow.dtype).view(1, 1, 1, W).expand(B, -1, H, -1)
    gy = torch.linspace(-1.0, 1.0, H, dtype=tenFlow.dtype).view(1, 1, H, 1).expand(B, -1, -1, W)
    base = torch.cat([gx, gy], 1)
    # :18-19 flow normalised by (size-1)/2 of the INPUT
    nflow = torch.cat([tenFlow[:, 0:1] / ((tenInput.shape[3] - 1.0) / 2.0),
                       tenFlow[:, 1:2] / ((tenInput.shape[2] - 1.0) / 2.0)], 1)
    g = (base + nflow).permute(0, 2, 3, 1)  # :24
    return F.grid_sample(tenInput, g, mode='bilinear', padding_mode='border',
                         align_corners=True)  # :25


def _bilinear_gather(img, ix, iy, zeros_pad):
    """img [B,C,H,W]; ix, iy [B,H',W'] float coordinates.  Corners outside the image add 0."""
    B, C, H, W = img.shape
    x0 = torch.floor(ix)
    y0 = torch.floor(iy)
    ax, ay = ix - x0, iy - y0
    bx, by = (x0 + 1) - ix, (y0 + 1) - iy
    out = 0
    flat = img.reshape(B, C, H * W)
    for dx, dy, w in ((0, 0, bx * by), (1, 0, ax * by), (0, 1, bx * ay), (1, 1, ax * ay)):
        xi, yi = (x0 + dx).long(), (y0 + dy).long()
        ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
        idx = (yi.clamp(0, H - 1) * W + xi.clamp(0, W - 1)).view(B, 1, -1).expand(-1, C, -1)
        val = torch.gather(flat, 2, idx).view(B, C, *ix.shape[1:])
        out = out + val * (w * ok.to(img.dtype)).unsqueeze(1)
    return out


def warp2d_rife_closed(tenInput, tenFlow):
    """out[y,x] = bilinear(in, x+u, y+v) with the coordinate clamped to the border."""
    B, C, H, W = tenInput.shape
    xs = torch.arange(W, dtype=tenFlow.dtype).view(1, 1, W)
    ys = torch.arange(H, dtype=tenFlow.dtype).view(1, H, 1)
    ix = (xs + tenFlow[:, 0]).clamp(0, W - 1)
    iy = (ys + tenFlow[:, 1]).clamp(0, H - 1)
    return _bilinear_gather(tenInput, ix, iy, zeros_pad=False)


# --------------------------------------------------------------------------------------------
# a2  Flow-3D/model/warplayer.py:9-41
# --------------------------------------------------------------------------------------------
def warp3d_ref(tenInput, tenFlow):
    B, _, D, H, W = tenFlow.shape
    dt = tenFlow.dtype
    # :15-20 -- channel 0 is a linspace over dim 3 (H), channel 1 over dim 2 (D), channel 2
    # over dim 4 (W); 5-D grid_sample reads them as (x->W, y->H, z->D): the axis rotation.
    g0 = torch.linspace(-1.0, 1.0, H, dtype=dt).view(1, 1, 1, H, 1).expand(B, -1, D, -1, W)
    g1 = torch.linspace(-1.0, 1.0, D, dtype=dt).view(1, 1, D, 1, 1).expand(B, -1, -1, H, W)
    g2 = torch.linspace(-1.0, 1.0, W, dtype=dt).view(1, 1, 1, 1, W).expand(B, -1, D, H, -1)
    base = torch.cat([g0, g1, g2], 1)
    # :24-26 -- divisors come from input dims 3, 2, 4
    nflow = torch.cat([tenFlow[:, 0:1] / ((tenInput.shape[3] - 1.0) / 2.0),
                       tenFlow[:, 1:2] / ((tenInput.shape[2] - 1.0) / 2.0),
                       tenFlow[:, 2:3] / ((tenInput.shape[4] - 1.0) / 2.0)], 1)
    g = (base + nflow).permute(0, 2, 3, 4, 1)  # :31
    return F.grid_sample(tenInput, g, mode='bilinear', padding_mode='border',
                         align_corners=True)  # :36


def warp3d_closed(tenInput, tenFlow):
    """SURVEY §8 a2 closed form:
    out[d,h,w] = trilinear(in; iz=(w+F2)(D-1)/(W-1), iy=(d+F1)(H-1)/(D-1), ix=(h+F0)(W-1)/(H-1))
    with the coordinates clamped to the volume."""
    B, C, D, H, W = tenInput.shape
    dt = tenFlow.dtype
    dd = torch.arange(D, dtype=dt).view(1, D, 1, 1)
    hh = torch.arange(H, dtype=dt).view(1, 1, H, 1)
    ww = torch.arange(W, dtype=dt).view(1, 1, 1, W)
    ix = ((hh + tenFlow[:, 0]) * ((W - 1) / (H - 1))).clamp(0, W - 1)
    iy = ((dd + tenFlow[:, 1]) * ((H - 1) / (D - 1))).clamp(0, H - 1)
    iz = ((ww + tenFlow[:, 2]) * ((D - 1) / (W - 1))).clamp(0, D - 1)
    x0, y0, z0 = torch.floor(ix), torch.floor(iy), torch.floor(iz)
    ax, ay, az = ix - x0, iy - y0, iz - z0
    flat = tenInput.reshape(B, C, D * H * W)
    out = 0
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                wgt = (ax if dx else 1 - ax) * (ay if dy else 1 - ay) * (az if dz else 1 - az)
                xi, yi, zi = (x0 + dx).long(), (y0 + dy).long(), (z0 + dz).long()
                ok = (xi < W) & (yi < H) & (zi < D)
                idx = (zi.clamp(max=D - 1) * H + yi.clamp(max=H - 1)) * W + xi.clamp(max=W - 1)
                val = torch.gather(flat, 2, idx.view(B, 1, -1).expand(-1, C, -1)).view(B, C, D, H, W)
                out = out + val * (wgt * ok.to(dt)).unsqueeze(1)
    return out


# --------------------------------------------------------------------------------------------
# a5 / a6  UPFlow/model/pwc_modules.py:184-207, UPFlow/utils/tools.py:1317-1361
# --------------------------------------------------------------------------------------------
def _pwc_vgrid(flow):
    B, _, H, W = flow.shape
    # (device=...: these restatements also run as "the reference's stock torch ops on this GPU" -- the comparator
    # of tests/test_gpu_e2e.py's UPFlow band; on the CPU nothing changes)
    xx = torch.arange(0, W, device=flow.device).view(1, -1).repeat(H, 1).view(1, 1, H, W).repeat(B, 1, 1, 1)
    yy = torch.arange(0, H, device=flow.device).view(-1, 1).repeat(1, W).view(1, 1, H, W).repeat(B, 1, 1, 1)
    vgrid = torch.cat((xx, yy), 1).float() + flow  # pwc_modules.py:193-197
    # :199-200 scale to [-1, 1] with (W-1), (H-1); out-of-place so autograd stays simple
    vx = 2.0 * vgrid[:, 0] / max(W - 1, 1) - 1.0
    vy = 2.0 * vgrid[:, 1] / max(H - 1, 1) - 1.0
    return torch.stack((vx, vy), dim=3)  # B,H,W,2 (:201)


def warp2d_pwc_ref(x, flow, with_mask):
    vgrid = _pwc_vgrid(flow)
    # :202 default grid_sample: bilinear, zeros, align_corners=False
    x_warp = F.grid_sample(x, vgrid, padding_mode='zeros', align_corners=False)
    if not with_mask:
        return x_warp  # tools.torch_warp (tools.py:1345)
    mask = F.grid_sample(torch.ones_like(x), vgrid, align_corners=False)  # :203-207
    mask = (mask >= 1.0).float()
    return x_warp * mask


def pwc_mask_borderline(x, flow):
    """Pixels whose validity mask is decided by fp32 rounding: the in-bounds weight sum lies in
    [1 - 2^-21, 1 + 2^-21] (SURVEY §7 hard parts).  Parity is not defined there."""
    vgrid = _pwc_vgrid(flow).double()
    m = F.grid_sample(torch.ones_like(x[:, :1]).double(), vgrid, align_corners=False)
    return (m - 1.0).abs() <= 2.0 ** -21


def warp2d_pwc_closed(x, flow):
    """Sample at ((x+u) W/(W-1) - 0.5, (y+v) H/(H-1) - 0.5), zeros outside; no mask."""
    B, C, H, W = x.shape
    xs = torch.arange(W, dtype=flow.dtype).view(1, 1, W)
    ys = torch.arange(H, dtype=flow.dtype).view(1, H, 1)
    ix = (xs + flow[:, 0]) * (W / max(W - 1, 1)) - 0.5
    iy = (ys + flow[:, 1]) * (H / max(H - 1, 1)) - 0.5
    return _bilinear_gather(x, ix, iy, zeros_pad=True)


# --------------------------------------------------------------------------------------------
# a11 (warp part)  Flow-2D/model/RIFE.py:244-262 `backwrd_warp`
# --------------------------------------------------------------------------------------------
def warp2d_photo_ref(frame, flow):
    b, c, h, w = flow.size()
    frame = F.interpolate(frame, size=(h, w), mode='bilinear', align_corners=True)  # :248
    fl = flow.transpose(1, 2).transpose(2, 3)  # :249-250  B,H,W,2
    xx = torch.arange(0, w).view(1, -1).repeat(h, 1).view(1, 1, h, w).repeat(b, 1, 1, 1)
    yy = torch.arange(0, h).view(-1, 1).repeat(1, w).view(1, 1, h, w).repeat(b, 1, 1, 1)
    grid = torch.cat((xx, yy), 1).float().transpose(1, 2).transpose(2, 3)  # :228-242
    grid = fl + grid  # :255
    factor = torch.FloatTensor([[[[2 / w, 2 / h]]]])  # :258
    grid = grid * factor - 1  # :259
    return F.grid_sample(frame, grid, align_corners=False)  # :260 (torch default)


def warp2d_photo_closed(frame, flow):
    """Sample at (x+u-0.5, y+v-0.5), zeros outside."""
    B, C, H, W = frame.shape
    xs = torch.arange(W, dtype=flow.dtype).view(1, 1, W)
    ys = torch.arange(H, dtype=flow.dtype).view(1, H, 1)
    return _bilinear_gather(frame, xs + flow[:, 0] - 0.5, ys + flow[:, 1] - 0.5, zeros_pad=True)


# --------------------------------------------------------------------------------------------
# a7  UPFlow/utils/tools.py:393-541 boundary_dilated_warp.warp_im
# --------------------------------------------------------------------------------------------
def warp2d_dilated_ref(I, flow, start=None):
    B, C, H, W = I.shape
    _, _, ph, pw = flow.shape
    xs = torch.arange(pw, dtype=flow.dtype, device=flow.device).view(1, 1, pw)
    ys = torch.arange(ph, dtype=flow.dtype, device=flow.device).view(1, ph, 1)
    if start is None:
        start = torch.zeros(B, 2, 1, 1, dtype=flow.dtype, device=flow.device)
    # get_grid :396-410 adds the patch offset, warp_im :538 adds the flow
    x = (xs + start[:, 0]) + flow[:, 0]
    y = (ys + start[:, 1]) + flow[:, 1]
    # _interpolate :444-454: floor, +1, then clamp the INDICES
    x0 = torch.floor(x).int()
    x1 = x0 + 1
    y0 = torch.floor(y).int()
    y1 = y0 + 1
    x0, x1 = x0.clamp(0, W - 1), x1.clamp(0, W - 1)
    y0, y1 = y0.clamp(0, H - 1), y1.clamp(0, H - 1)
    flat = I.reshape(B, C, H * W)

    def take(yy, xx):  # :480-497 gathers
        idx = (yy.long() * W + xx.long()).view(B, 1, -1).expand(-1, C, -1)
        return torch.gather(flat, 2, idx).view(B, C, ph, pw)

    Ia, Ib, Ic, Id = take(y0, x0), take(y1, x0), take(y0, x1), take(y1, x1)
    x0f, x1f, y0f, y1f = x0.float(), x1.float(), y0.float(), y1.float()
    # :504-508 weights use the CLAMPED corners against the UNCLAMPED coordinate
    wa = ((x1f - x) * (y1f - y)).unsqueeze(1)
    wb = ((x1f - x) * (y - y0f)).unsqueeze(1)
    wc = ((x - x0f) * (y1f - y)).unsqueeze(1)
    wd = ((x - x0f) * (y - y0f)).unsqueeze(1)
    return wa * Ia + wb * Ib + wc * Ic + wd * Id


# --------------------------------------------------------------------------------------------
# §8f.2  UPFlow/utils/tools.py:543-719 occ_check_model (forward-backward check + outgoing mask)
# --------------------------------------------------------------------------------------------
def occ_fb_lhs_thresh(flow_fw, flow_bw, alpha1, alpha2, scale=1):
    """tools.py:592-622: the two tested quantities and the threshold (before the `<`)."""
    def mag(x):  # length_sq_v0 (:596-601)
        return torch.sum(torch.pow(x ** 2, 0.5), dim=1, keepdim=True)
    mag_sq = mag(flow_fw) + mag(flow_bw)                                  # :615
    flow_bw_warped = warp2d_pwc_ref(flow_bw, flow_fw, with_mask=False)    # :616 tools.torch_warp
    flow_fw_warped = warp2d_pwc_ref(flow_fw, flow_bw, with_mask=False)    # :617
    thresh = alpha1 * mag_sq + alpha2 / scale                             # :620
    return mag(flow_fw + flow_bw_warped), mag(flow_bw + flow_fw_warped), thresh


def occ_outgoing_ref(flow):
    """tools.py:683-709."""
    B, C, H, W = flow.size()
    xx = torch.arange(0, W, device=flow.device).view(1, -1).repeat(H, 1).view(1, 1, H, W).repeat(B, 1, 1, 1).float()
    yy = torch.arange(0, H, device=flow.device).view(-1, 1).repeat(1, W).view(1, 1, H, W).repeat(B, 1, 1, 1).float()
    pos_x, pos_y = xx + flow[:, 0:1], yy + flow[:, 1:2]
    m = torch.ones_like(pos_x)
    m[pos_x > W - 1] = 0
    m[pos_x < 0] = 0
    m[pos_y > H - 1] = 0
    m[pos_y < 0] = 0
    return m.float()


def occ_check_ref(flow_f, flow_b, alpha1, alpha2, scale=1, obj_out_all='obj'):
    """tools.py:560-590 dispatch; returns (occ_fw, occ_bw), 0 = occluded."""
    if obj_out_all == 'out':
        return occ_outgoing_ref(flow_f), occ_outgoing_ref(flow_b)
    lf, lb, th = occ_fb_lhs_thresh(flow_f, flow_b, alpha1, alpha2, scale)
    occ_1, occ_2 = (lf < th).float(), (lb < th).float()                  # :621-622
    if obj_out_all == 'all':
        return occ_1, occ_2

    def obj(occ, out):  # :711-719
        o = torch.zeros_like(occ)
        o[occ == 1] = 1
        o[out == 0] = 1
        return o
    return obj(occ_1, occ_outgoing_ref(flow_f)), obj(occ_2, occ_outgoing_ref(flow_b))
