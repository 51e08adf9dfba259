/*
 * flowsci_hip.h -- C-ABI of libflowsci_hip.so: the MI355X (gfx950) hot path of
 * HamidGadirov/OpticalFlowSciVis (backward warps, local-window correlation,
 * photometric / census losses), as hand-written HIP kernels.
 *
 * Conventions (every entry point):
 *   - plain pointers to DEVICE memory, fp32, contiguous NCHW / NCDHW; sizes as int;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *   - returns FS_OK (0) or an FS_ERR_* code; never throws, never allocates,
 *     never synchronises; the caller owns all memory;
 *   - stateless and re-entrant (the reference's module-level grid cache,
 *     Flow-2D/model/warplayer.py:5, has no equivalent: grids are computed in-kernel).
 *
 * Each declaration cites the reference interface it replaces (paths relative to the
 * reference tree).  The Python binding a maintainer would add is in INTEGRATION.md.
 */
#ifndef FLOWSCI_HIP_H
#define FLOWSCI_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef void* fs_stream_t; /* hipStream_t */

enum {
  FS_OK = 0,
  FS_ERR_NULLPTR = 1, /* a required pointer is NULL */
  FS_ERR_SHAPE = 2,   /* a size is out of the supported range */
  FS_ERR_ARG = 3,     /* an option / mode value is invalid */
  FS_ERR_LAUNCH = 4,  /* hipGetLastError() != hipSuccess after the launch */
  FS_ERR_UNSUPPORTED = 5 /* a fused entry point has no kernel for this shape / alignment: use the unfused ones */
};

/* ABI version of this header (major*10000 + minor*100 + patch).  ANY change to the argument list of an existing
 * entry point bumps it, and a binding must refuse a library whose fs_version() differs from the header it was
 * written against -- a stale .so would otherwise run with shifted arguments instead of failing.
 *   100  round 1.
 *   300  round 2 inserted `const int* in_hw` into fs_warp2d_{fwd,bwd} and fs_warp2d_pair_{fwd,bwd} (without bumping
 *        the number: fixed in round 3) and added the f1-f4 / conv3d entry points.
 *   310  round 3: fs_conv3d_fwd* / fs_conv3d_tr* accept w = NULL ("ws is prepared"), fs_conv3d_*_wprep_jobs,
 *        fs_conv3d_wprep_batch.
 *   320  round 4: fs_conv3d_wrw_kernel_id (which weight-gradient kernel a call dispatches to; nothing launched);
 *        fs_conv3d_wrw_det / fs_conv3d_wrw_det_ws_floats (workspace form without float atomics).
 *        The library reads no environment variable any more (measurement switches live in the -DFS_ABLATION build).
 *   330  round 5: fs_warp3d_kernel_id (which kernel a trilinear-warp call dispatches to; nothing launched). */
#define FS_ABI_VERSION 330
int fs_version(void);
/* Static string for an FS_* code. */
const char* fs_error_string(int code);

/* ------------------------------------------------------------------------------------
 * a2. Flow-3D trilinear backward warp -- Flow-3D/model/warplayer.py:9-41 `warp`.
 *   in   [B,C,D,H,W], flow [B,3,D,H,W] -> out [B,C,D,H,W].
 *   grid = (linspace(H) , linspace(D), linspace(W)) + flow/((dim-1)/2)   (:15-26)
 *   5-D grid_sample(bilinear, border, align_corners=True)               (:36)
 *   i.e. the axis-rotating sampling  out[d,h,w] = in[(w+F2)(D-1)/(W-1), (d+F1)(H-1)/(D-1),
 *   (h+F0)(W-1)/(H-1)]  with border clamp.  D,H,W >= 2.
 *   D,H,W are the extent of the FLOW (= of the output).  The reference builds its grid from
 *   the flow's shape and its divisors from the input's (:11-26), so the sampled volume may
 *   have a different extent: `in_dhw` = {Din,Hin,Win} (HOST pointer to 3 ints) or NULL when
 *   the input has the flow's extent.  (IFNet-3D hits this for sizes that are not multiples
 *   of 16: Flow-3D/model/IFNet.py:151-191.)
 * bwd: grad_in (nullable; must be zero-filled by the caller, accumulated with float
 *   atomics) and grad_flow (nullable; fully overwritten).
 */
int fs_warp3d_fwd(const float* in, const float* flow, float* out,
                  int B, int C, const int* in_dhw, int D, int H, int W, fs_stream_t stream);
int fs_warp3d_bwd(const float* in, const float* flow, const float* grad_out,
                  float* grad_in, float* grad_flow,
                  int B, int C, const int* in_dhw, int D, int H, int W, fs_stream_t stream);

/* The IFNet call site warps BOTH frames with the two halves of one 6-channel flow:
 *   warped_img0 = warp(img0, flow[:, :3]); warped_img1 = warp(img1, flow[:, 3:6])
 *   (Flow-3D/model/IFNet.py:190-191, 233-234).  One launch serves both and uses flow6 /
 * grad_flow6 [B,6,D,H,W] in place (no slice copies).  grad_img0/grad_img1: both or neither.
 */
int fs_warp3d_pair_fwd(const float* img0, const float* img1, const float* flow6,
                       float* out0, float* out1,
                       int B, int C, const int* in_dhw, int D, int H, int W, fs_stream_t stream);
int fs_warp3d_pair_bwd(const float* img0, const float* img1, const float* flow6,
                       const float* grad_out0, const float* grad_out1,
                       float* grad_img0, float* grad_img1, float* grad_flow6,
                       int B, int C, const int* in_dhw, int D, int H, int W, fs_stream_t stream);
/* The same with the gradient that reaches the flow from its OTHER consumers summed in by the launch:
 * grad_flow6 = d(warps)/d(flow) + grad_flow_add (nullable; may BE grad_flow6 -- in-place accumulation).
 * An IFNet block's flow feeds the warp, the next block's input and accumulation, and the distillation
 * term (Flow-3D/model/IFNet.py:190-191, 210-214, 261); autograd would add the warp's share to the others'
 * in a separate pass over 805 MB tensors. */
int fs_warp3d_pair_bwd_acc(const float* img0, const float* img1, const float* flow6,
                           const float* grad_out0, const float* grad_out1,
                           float* grad_img0, float* grad_img1,
                           const float* grad_flow_add, float* grad_flow6,
                           int B, int C, const int* in_dhw, int D, int H, int W, fs_stream_t stream);

/* f1: "upsample flow x scale -> warp" in ONE kernel -- Flow-3D/model/IFNet.py:118
 *   flow = F.interpolate(flow_d, scale_factor=s, mode="trilinear", align_corners=False) * s   (+ the running
 *   flow of :213-214) followed by :190-191 warp(img0, flow[:, :3]), warp(img1, flow[:, 3:6]).
 * delta [B,6,Ds,Hs,Ws] is the head's output at the block's working resolution; prev_flow (nullable),
 * flow_out, out0, out1 live at factor x that extent (factor 2 or 4).  flow_out = prev_flow + scale *
 * upsample(delta) is formed per tile in registers, written once (it has three more consumers) and
 * handed to the warp without a second trip through HBM; values are bit-identical to
 * fs_upsample3d_scale_add + fs_warp3d_pair_fwd.
 * bwd: grad_flow_total = d(warps)/d(flow_out) + grad_flow_add (gradient reaching flow_out from its other
 * consumers; nullable; may alias grad_flow_total) -- also the gradient of prev_flow -- and
 * grad_delta = scale * adjoint_upsample(grad_flow_total).  ws: B*6*(D*H*Ws + D*Hs*Ws) floats. */
int fs_upsample_warp3d_pair_fwd(const float* img0, const float* img1, const float* delta,
                                const float* prev_flow, float* flow_out, float* out0, float* out1,
                                int B, int C, const int* in_dhw, int Ds, int Hs, int Ws, int factor,
                                float scale, fs_stream_t stream);
int fs_upsample_warp3d_pair_bwd(const float* img0, const float* img1, const float* flow6,
                                const float* grad_out0, const float* grad_out1,
                                const float* grad_flow_add, float* grad_flow_total, float* grad_delta,
                                float* ws, int B, int C, const int* in_dhw, int Ds, int Hs, int Ws,
                                int factor, float scale, fs_stream_t stream);
/* fs_warp3d_pair_bwd_acc / fs_upsample_warp3d_pair_bwd with up to THREE gradients reaching the flow from its
 * other consumers (Flow-3D/model/IFNet.py: the flow of a block feeds :190-191 the warps, :183 the next block's
 * input `torch.cat`, :213 the running-flow accumulation and :262 the distillation term).  Each add* (nullable)
 * is a [B,6,D,H,W] tensor or a 6-channel slice of a wider one: batch_stride* = its batch stride in floats
 * (>= 6*D*H*W), channel stride D*H*W.  add0 may alias grad_flow6 / grad_flow_total.  The gradients of the two
 * warped frames may be channel slices too (they are channels 2 and 3 of the next block's input gradient):
 * gout_batch_stride* = their batch stride in floats, 0 = dense (C*D*H*W). */
int fs_warp3d_pair_bwd_acc3(const float* img0, const float* img1, const float* flow6,
                            const float* grad_out0, long long gout_batch_stride0, const float* grad_out1,
                            long long gout_batch_stride1, float* grad_img0, float* grad_img1,
                            const float* add0, long long batch_stride0, const float* add1, long long batch_stride1,
                            const float* add2, long long batch_stride2, float* grad_flow6,
                            int B, int C, const int* in_dhw, int D, int H, int W, fs_stream_t stream);
int fs_upsample_warp3d_pair_bwd3(const float* img0, const float* img1, const float* flow6,
                                 const float* grad_out0, long long gout_batch_stride0, const float* grad_out1,
                                 long long gout_batch_stride1, const float* add0, long long batch_stride0, const float* add1, long long batch_stride1,
                                 const float* add2, long long batch_stride2, float* grad_flow_total, float* grad_delta,
                                 float* ws, int B, int C, const int* in_dhw, int Ds, int Hs, int Ws,
                                 int factor, float scale, fs_stream_t stream);

/* Which kernel the trilinear-warp entry points above dispatch a call of this geometry to (nothing is launched; the
 * pointers are inspected for alignment only, `in0` / `in1` = the sampled volumes, in1 NULL for a single warp):
 *   FS_W3_KERNEL_GATHER (0)  the global-gather kernels (rounds 1-4: warp3d_fwd_ring_kernel / warp3d_bwd_kernel)
 *   FS_W3_KERNEL_RC     (1)  round 5: ring pipeline with the gather source in an LDS row cache (warp3d_rc_kernel;
 *                            C == 1, W_in % 4 == 0, the 37-plane x 72-column window fits the sampled volume, 16-byte
 *                            aligned tensors, flow gradient only)
 * `backward` != 0: fs_warp3d*_bwd* with `with_grad_in` saying whether grad_in / grad_img* are asked for; a negative
 * return value is -FS_ERR_*.  (bench.py names the kernel symbol of its roofline records with it.) */
enum { FS_W3_KERNEL_GATHER = 0, FS_W3_KERNEL_RC = 1 };
int fs_warp3d_kernel_id(const float* in0, const float* in1, const float* flow, int B, int C, const int* in_dhw,
                        int D, int H, int W, int backward, int with_grad_in);

/* ------------------------------------------------------------------------------------
 * 2-D bilinear backward warps.  in [B,C,H,W], flow [B,2,H,W] (ch0 = x, ch1 = y),
 * out [B,C,H,W].  `mode` selects the reference call site whose coordinate convention is
 * reproduced:
 *   FS_WARP2D_RIFE    a1  Flow-2D/model/warplayer.py:7-26  border pad, align_corners=True,
 *                         samples at (x+u, y+v).
 *   FS_WARP2D_PWC     a5/a6 UPFlow/model/pwc_modules.py:184-207 (WarpingLayer_no_div) and
 *                         UPFlow/utils/tools.py:1317-1361 (torch_warp): vgrid = 2(x+u)/(W-1)-1,
 *                         zeros pad, align_corners=False.
 *   FS_WARP2D_PHOTO   a11 Flow-2D/model/RIFE.py:244-262 (`backwrd_warp`): grid = (x+u)*2/W-1,
 *                         zeros pad, align_corners=False => samples at (x+u-0.5, y+v-0.5).
 *   FS_WARP2D_DILATED a7  UPFlow/utils/tools.py:412-541 (boundary_dilated_warp.warp_im):
 *                         samples at (x+start_x+u, y+start_y+v), indices clamped, weights from
 *                         clamped corners vs unclamped coordinate.  `start` = [B,2] device
 *                         floats (x,y) or NULL for zeros.
 * `with_mask` (FS_WARP2D_PWC only): multiply by (sum of in-bounds weights >= 1.0), the
 *   validity mask of pwc_modules.py:200-207.
 * bwd: grad_in nullable (caller zero-fills; float atomics), grad_flow nullable (overwritten).
 */
enum { FS_WARP2D_RIFE = 0, FS_WARP2D_PWC = 1, FS_WARP2D_PHOTO = 2, FS_WARP2D_DILATED = 3 };

/* H, W = extent of the flow (= of the output).  `in_hw` = {Hin, Win} (HOST pointer to 2 ints) or NULL
 * when the image has the flow's extent; only FS_WARP2D_RIFE defines the other case (the reference builds
 * its grid from the flow's shape and its divisors from the input's, warplayer.py:10-20; IFNet-2D hits it
 * for extents where its blocks return fewer rows than the frames have, e.g. H = 146). */
int fs_warp2d_fwd(const float* in, const float* flow, const float* start, float* out,
                  int B, int C, const int* in_hw, int H, int W, int mode, int with_mask,
                  fs_stream_t stream);
int fs_warp2d_bwd(const float* in, const float* flow, const float* start,
                  const float* grad_out, float* grad_in, float* grad_flow,
                  int B, int C, const int* in_hw, int H, int W, int mode, int with_mask,
                  fs_stream_t stream);

/* Pair form for the IFNet call site (Flow-2D/model/IFNet.py:191-192, 230-231):
 *   warp(img0, flow[:, :2]); warp(img1, flow[:, 2:4])  with flow4 [B,4,H,W] used in place.
 * No mask, no start.  * Aliasing: grad_in / grad_img0 / grad_img1 are accumulated INTO (zero-fill them first).  Planes of <= 48 KB are added
 * by plane-owning workgroups with plain read-modify-writes, which needs every grad_img of a pair launch to be present and
 * the two to be different tensors; a pair launch whose grad_img0 == grad_img1 (or with one of them NULL where that is
 * allowed) takes the float-atomic kernel instead, which tolerates the aliasing. */
int fs_warp2d_pair_fwd(const float* img0, const float* img1, const float* flow4,
                       float* out0, float* out1,
                       int B, int C, const int* in_hw, int H, int W, int mode, fs_stream_t stream);
int fs_warp2d_pair_bwd(const float* img0, const float* img1, const float* flow4,
                       const float* grad_out0, const float* grad_out1,
                       float* grad_img0, float* grad_img1, float* grad_flow4,
                       int B, int C, const int* in_hw, int H, int W, int mode, fs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * f2. Forward-backward occlusion check + outgoing mask, fused.
 *   UPFlow/utils/tools.py:560-590 (occ_check_model.__call__ dispatch), :592-630
 *   (_forward_backward_occ_check with length_sq_v0 = sum_c |x_c| and two torch_warp calls),
 *   :683-709 (torch_outgoing_occ_check), :711-719 (torch_get_obj_occ_check).
 *   flow_f, flow_b [B,2,H,W] -> occ_f, occ_b [B,1,H,W] in {0,1} (0 = occluded / ignore).
 *   alpha1, alpha2_over_scale: occ_thresh = alpha1 * (|flow_f|_1 + |flow_b|_1) + alpha2/scale.
 *   mode FS_OCC_ALL: the consistency masks; FS_OCC_OUT: the outgoing masks only;
 *   FS_OCC_OBJ: (consistent == 1) or (outgoing == 0) -- the reference's 'obj' setting.
 *   No gradient: the reference's masks are bool -> float.
 */
enum { FS_OCC_ALL = 0, FS_OCC_OBJ = 1, FS_OCC_OUT = 2 };
int fs_occ_check2d(const float* flow_f, const float* flow_b, float* occ_f, float* occ_b,
                   int B, int H, int W, float alpha1, float alpha2_over_scale, int mode,
                   fs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * a3/a4. Local-window correlation (cost volume) -- replaces the `correlation_cuda` torch
 * extension bound at UPFlow/model/correlation_package/correlation.py:4,26-27,42-43
 * (sources absent from the reference tree; semantics pinned by Corr_pyTorch,
 * UPFlow/utils/pytorch_correlation.py:27-50) for the only configuration the reference uses:
 * pad_size = max_displacement = md, kernel_size = 1, stride1 = stride2 = 1, corr_multiply = 1
 * (UPFlow/model/upflow.py:649,652: md = 4).
 *   f1, f2 [B,C,H,W] -> out [B,(2md+1)^2,H,W],
 *   out[b,(dy+md)(2md+1)+(dx+md),y,x] = (1/C) sum_c f1[b,c,y,x] f2[b,c,y+dy,x+dx], zero pad.
 * md in 1..4.  bwd: grad_f1 / grad_f2 nullable (at least one), fully overwritten; no atomics,
 * bitwise reproducible.  The reference's rbot1/rbot2 scratch tensors have no equivalent.
 * Sizes: B*C*H*W and (2md+1)^2*H*W below 2^29 floats (32-bit byte offsets inside a tensor / a sample's
 * cost volume), FS_ERR_SHAPE otherwise; any alignment of rows (W need not be a multiple of 4).
 */
int fs_corr2d_fwd(const float* f1, const float* f2, float* out,
                  int B, int C, int H, int W, int max_displacement, fs_stream_t stream);
int fs_corr2d_bwd(const float* f1, const float* f2, const float* grad_out,
                  float* grad_f1, float* grad_f2,
                  int B, int C, int H, int W, int max_displacement, fs_stream_t stream);

/* f4. normalize_features folded into the cost volume: UPFlow/model/upflow.py:96-138
 * with normalize = center = True, moments_across_channels = moments_across_images = False (the
 * configuration UPFlow_net.demo() / C3 uses, upflow.py:635-641): every (b, c) plane of each feature map
 * is centred and scaled by its own mean / sqrt(unbiased var + 1e-16) immediately before
 * correlation.py:26.  The normalised maps are never materialised:
 *   fs_plane_moments   : stats[plane] = (mean, 1/sqrt(var + 1e-16)) for `planes` planes of S floats (S >= 2);
 *   fs_corr2d_norm_fwd : fs_corr2d_fwd of the normalised maps, normalisation applied as tiles are
 *                        staged (the zero padding of f2 applies AFTER normalisation, as in the reference);
 *   fs_corr2d_norm_bwd : gradients w.r.t. the NORMALISED maps (either may be NULL);
 *   fs_plane_norm_bwd  : chains one of them through the normalisation,
 *                        grad_f = r (grad_n - mean(grad_n) - n sum(grad_n n)/(S-1)).
 */
int fs_plane_moments(const float* f, float* stats, int planes, int S, fs_stream_t stream);
int fs_plane_norm_bwd(const float* f, const float* stats, const float* grad_n, float* grad_f,
                      int planes, int S, fs_stream_t stream);
int fs_corr2d_norm_fwd(const float* f1, const float* f2, const float* stats1, const float* stats2,
                       float* out, int B, int C, int H, int W, int max_displacement,
                       fs_stream_t stream);
int fs_corr2d_norm_bwd(const float* f1, const float* f2, const float* stats1, const float* stats2,
                       const float* grad_out, float* grad_n1, float* grad_n2,
                       int B, int C, int H, int W, int max_displacement, fs_stream_t stream);

/* Both directions of one pyramid level in ONE launch (UPFlow/model/upflow.py:649 and :652 correlate
 * (x1, x2_warp) and (x2, x1_warp) at every level): set a = (f1a, f2a) -> outa, set b = (f1b, f2b) -> outb,
 * identical shapes.  `stats` (nullable) = [4][B*C][2] (mean, rstd) tables of (f1a, f2a, f1b, f2b) as
 * fs_plane_moments4 writes them: the normalize_features-folded variant; with it, the bwd gradients are
 * w.r.t. the NORMALISED maps (chain them with fs_plane_norm_bwd4; a NULL grad_f skips that tensor).
 * The five levels of a step cannot share a launch: level l's features are warped with the flow estimated at
 * level l-1 (upflow.py:621-633). */
int fs_corr2d_pair_fwd(const float* f1a, const float* f2a, const float* f1b, const float* f2b,
                       const float* stats, float* outa, float* outb,
                       int B, int C, int H, int W, int max_displacement, fs_stream_t stream);
int fs_corr2d_pair_bwd(const float* f1a, const float* f2a, const float* f1b, const float* f2b,
                       const float* stats, const float* grad_outa, const float* grad_outb,
                       float* grad_f1a, float* grad_f2a, float* grad_f1b, float* grad_f2b,
                       int B, int C, int H, int W, int max_displacement, fs_stream_t stream);
int fs_plane_moments4(const float* fa, const float* fb, const float* fc, const float* fd, float* stats,
                      int planes, int S, fs_stream_t stream);
int fs_plane_norm_bwd4(const float* fa, const float* fb, const float* fc, const float* fd,
                       const float* stats, const float* grad_na, const float* grad_nb,
                       const float* grad_nc, const float* grad_nd, float* grad_fa, float* grad_fb,
                       float* grad_fc, float* grad_fd, int planes, int S, fs_stream_t stream);

/* 3-D correlation: NEW capability named by BASELINE.json (config 4); the reference has no 3-D cost
 * volume, so this generalises the 2-D layer above (dz-major, then dy, dx; channel mean; zero pad):
 *   f1, f2 [B,C,D,H,W] -> out [B,(2md+1)^3,D,H,W].  Pinned to the reference only through D = 1.
 * Sizes: B*C*D*H*W and (2md+1)^3*D*H*W below 2^29 floats, FS_ERR_SHAPE otherwise.
 */
int fs_corr3d_fwd(const float* f1, const float* f2, float* out,
                  int B, int C, int D, int H, int W, int max_displacement, fs_stream_t stream);
int fs_corr3d_bwd(const float* f1, const float* f2, const float* grad_out,
                  float* grad_f1, float* grad_f2,
                  int B, int C, int D, int H, int W, int max_displacement, fs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * a9/a10. Robust penalty + (masked) reduction -- the arithmetic of
 *   loss_functions.photo_loss_function   UPFlow/utils/loss.py:17-48   (tail of the census loss)
 *   network_tools.photo_loss_multi_type  UPFlow/model/upflow.py:267-289
 *   torch.nn.functional.l1_loss          Flow-3D/model/RIFE.py:132-134 (FS_PEN_L1)
 * in one pass:  S1 = sum_e pen(x[e] - y[e]) * w,  S2 = sum w   (w broadcast over C; S2 counts
 * each pixel once).  x, y [B,C,S] (y NULL => pen(x)), w [B,1,S] or NULL (then S2 = 0).
 * pen: ABS_ROBUST (|v|+0.01)^q | CHARBONNIER (v^2+eps)^q | L1_EPS |v+1e-6| | L1 |v|.
 * `border` > 0 (needs S == H*W): weights within `border` px of the image edge are zeroed --
 * the inner mask of census_loss_torch (loss.py:74-88, max_distance = 3).
 * sums: 2 device floats.  ws: device scratch, 2*FS_REDUCE_BLOCKS floats (caller-owned).
 * The caller turns (S1, S2) into the reference's mean / ratio forms.  Deterministic.
 * bwd: grad_x[e] = coef[0] * pen'(x-y) * w ; grad_y = -grad_x.  coef = device scalar
 * (upstream gradient times d loss / d S1).
 */
enum { FS_PEN_ABS_ROBUST = 0, FS_PEN_CHARBONNIER = 1, FS_PEN_L1_EPS = 2, FS_PEN_L1 = 3 };
#define FS_REDUCE_BLOCKS 1024

int fs_robust_sum(const float* x, const float* y, const float* w, float* sums, float* ws,
                  int B, int C, int S, int H, int W, int border, int mode, float q, float eps,
                  fs_stream_t stream);
int fs_robust_sum_bwd(const float* x, const float* y, const float* w, const float* coef,
                      float* grad_x, float* grad_y,
                      int B, int C, int S, int H, int W, int border, int mode, float q, float eps,
                      fs_stream_t stream);

/* a9, 'SSIM' branch: network_tools.weighted_ssim (UPFlow/model/upflow.py:141-196, c1 = inf,
 * c2 = 9e-6, weight_epsilon = 0.01, 3x3 valid average pools) fused with the reduction of
 * photo_loss_multi_type (:285-289):  sums[0] = sum ld * (use_occ ? avgpool(weight) : 1),
 * sums[1] = sum avgpool(weight);  x, y [B,C,H,W], weight [B,1,H,W], H, W >= 3.
 * loss = sums[0]/(sums[1]+1e-6) (use_occ) or sums[0]/(B*C*(H-2)*(W-2)).  ws as for fs_robust_sum.
 * bwd: grad_x / grad_y (nullable) = coef[0] * d sums[0] / d x, y  (the weight carries no gradient).
 */
int fs_wssim_fwd(const float* x, const float* y, const float* weight, float* sums, float* ws,
                 int B, int C, int H, int W, int use_occ, fs_stream_t stream);
int fs_wssim_bwd(const float* x, const float* y, const float* weight, const float* coef,
                 float* grad_x, float* grad_y, int B, int C, int H, int W, int use_occ,
                 fs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * a8. Census (soft ternary) distance -- loss_functions.census_loss_torch,
 * UPFlow/utils/loss.py:51-72: grey = .2989R+.5870G+.1140B; 7x7 zero-padded neighbourhood
 * differences t = d/sqrt(0.81+d^2); dist = sum_49 (t1-t2)^2/(0.1+(t1-t2)^2).
 *   img1, img2 [B,3,H,W] -> dist [B,1,H,W].  max_distance must be 3.
 * The loss is fs_robust_sum(dist, w = occ mask, border = 3, ...).
 * bwd: grad_dist [B,1,H,W] -> grad_img1 / grad_img2 [B,3,H,W] (nullable, overwritten);
 * gather-formulated, no atomics.
 */
int fs_census_dist_fwd(const float* img1, const float* img2, float* dist,
                       int B, int H, int W, int max_distance, fs_stream_t stream);
int fs_census_dist_bwd(const float* img1, const float* img2, const float* grad_dist,
                       float* grad_img1, float* grad_img2,
                       int B, int H, int W, int max_distance, fs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * a12. IFNet epilogues (Flow-2D/model/IFNet.py:239-248, Flow-3D/model/IFNet.py:241-267).
 * merge:   merged[B,C,S] = w0 * sigmoid(m) + w1 * (1 - sigmoid(m)),  m [B,1,S];
 *          sigmoid_out [B,1,S] nullable (the reference keeps it as mask_list[i]).
 * distill: sums[0] = sum_voxels sqrt(mean_c (flow_tea - flow_i)^2) * loss_mask,
 *          sums[1] = sum loss_mask,  loss_mask = mean_c|merged_i-gt| > mean_c|merged_tea-gt| + 0.01;
 *          merged_*, gt [B,C,S]; flow_* [B,F,S].  The term of loss_distill is sums[0] / (B*S).
 *          bwd: grad_flow_i only (teacher flow and mask are detached in the reference).
 */
int fs_merge_fwd(const float* w0, const float* w1, const float* mask_logit,
                 float* merged, float* sigmoid_out, int B, int C, int S, fs_stream_t stream);
int fs_merge_bwd(const float* w0, const float* w1, const float* mask_logit,
                 const float* grad_merged, const float* grad_sigmoid,
                 float* grad_w0, float* grad_w1, float* grad_mask_logit,
                 int B, int C, int S, fs_stream_t stream);
int fs_distill_fwd(const float* merged_i, const float* merged_tea, const float* gt,
                   const float* flow_i, const float* flow_tea, float* sums, float* ws,
                   int B, int C, int F, int S, fs_stream_t stream);
int fs_distill_bwd(const float* merged_i, const float* merged_tea, const float* gt,
                   const float* flow_i, const float* flow_tea, const float* coef,
                   float* grad_flow_i, int B, int C, int F, int S, fs_stream_t stream);
/* The three student terms of loss_distill (one per block, all against the same teacher: Flow-3D/model/
 * IFNet.py:259-262, Flow-2D/model/IFNet.py:244-248) in one launch each way: the teacher's flow / merged frame
 * and the ground truth are read once instead of three times.  sums[0..2] = the per-term sums of
 * sqrt(mean_c dflow^2) * mask (divide by B*S for the means; sums[3] is scratch: pass 4 floats);
 * ws: 4 * FS_REDUCE_BLOCKS floats.  bwd: coef = device scalar d(objective)/d(sum), shared by the terms. */
int fs_distill3_fwd(const float* merged0, const float* merged1, const float* merged2,
                    const float* merged_tea, const float* gt, const float* flow0, const float* flow1,
                    const float* flow2, const float* flow_tea, float* sums, float* ws,
                    int B, int C, int F, int S, fs_stream_t stream);
int fs_distill3_bwd(const float* merged0, const float* merged1, const float* merged2,
                    const float* merged_tea, const float* gt, const float* flow0, const float* flow1,
                    const float* flow2, const float* flow_tea, const float* coef,
                    float* grad_flow0, float* grad_flow1, float* grad_flow2,
                    int B, int C, int F, int S, fs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * §8f.1  Backward of the trilinear resizes inside IFBlock (Flow-3D/model/IFNet.py:85,88,118-119):
 * F.interpolate(mode="trilinear", align_corners=False, scale_factor = `factor` (upsample = 1) or
 * 1/`factor` (upsample = 0)), factor in {2, 4}.  grad_out [B,C,Dout,Hout,Wout] ->
 * grad_in [B,C,Din,Hin,Win], with out = in*factor (up) or in/factor (down, floor).
 * Gather-formulated adjoint (ATen scatters with atomics): no atomics, reproducible.
 * ws: device scratch for the up-sampling case, B*C*(Dout*Hout*Win + Dout*Hin*Win) floats (the
 * adjoint then runs as three separable 1-D passes); NULL selects the single-pass kernel.
 */
int fs_interp3d_bwd(const float* grad_out, float* grad_in, float* ws, int B, int C,
                    int Din, int Hin, int Win, int Dout, int Hout, int Wout,
                    int factor, int upsample, fs_stream_t stream);
/* grad_in = scale * adjoint(grad_out): the `* scale` of `F.interpolate(flow_d, ...) * scale` (IFNet.py:118)
 * folded into the last separable pass.  scale != 1 needs upsample = 1 and ws != NULL (FS_ERR_ARG otherwise). */
int fs_interp3d_bwd_scaled(const float* grad_out, float* grad_in, float* ws, int B, int C,
                           int Din, int Hin, int Win, int Dout, int Hout, int Wout,
                           int factor, int upsample, float scale, fs_stream_t stream);

/* Forward of the down-scaling side (Flow-3D/model/IFNet.py:85, 88):
 *   out = scale * F.interpolate(in, scale_factor = 1/factor, mode="trilinear", align_corners=False)
 * in [B,C,Din,Hin,Win] -> out [B,C,Din/factor,...] (floor), factor in {2, 4}; `scale` carries the flow's
 * `* 1. / scale` (:88).  ATen's arithmetic and summation order (every lambda is 1/2: bit-identical to ATen).  Adjoint: fs_interp3d_bwd
 * with upsample = 0. */
int fs_downsample3d_fwd(const float* in, float* out, int B, int C, int Din, int Hin, int Win,
                        int factor, float scale, fs_stream_t stream);
/* the same over an input whose C <= 12 channels are planes of different tensors (IFBlock's `torch.cat` in front of its
 * down-sampling, Flow-3D/model/IFNet.py:183 + :85): src[c] = channel c of sample 0, batch_strides[c] = that tensor's
 * batch stride in floats (host arrays, read at launch). */
int fs_downsample3d_fwd_ms(const float* const* src, const long long* batch_strides, float* out, int B, int C,
                           int Din, int Hin, int Win, int factor, float scale, fs_stream_t stream);

/* The 2-D pair (Flow-2D/model/IFNet.py:89, 92, 115-116): out = scale * F.interpolate(in, scale_factor =
 * factor (upsample = 1) or 1/factor (upsample = 0), mode="bilinear", align_corners=False), factor in {2, 4},
 * in [B,C,Hin,Win] -> out [B,C,Hout,Wout] with out = in*factor or floor(in/factor); ATen's arithmetic.
 * bwd: grad_in = scale * adjoint(grad_out), a gather per input pixel (no atomics, reproducible). */
int fs_resize2d_fwd(const float* in, float* out, int B, int C, int Hin, int Win, int Hout, int Wout,
                    int factor, int upsample, float scale, fs_stream_t stream);
int fs_resize2d_bwd(const float* grad_out, float* grad_in, int B, int C, int Hin, int Win, int Hout,
                    int Wout, int factor, int upsample, float scale, fs_stream_t stream);

/* Forward companion for the up-sampling side (Flow-3D/model/IFNet.py:118-119 followed by the
 * accumulation at :213-214 / :228-229):
 *   out = prev + scale * interpolate(small, scale_factor = factor, trilinear, align_corners=False)
 * small [B,C,Din,Hin,Win] -> out [B,C,factor*Din,...]; prev (same shape as out) may be NULL (= 0);
 * factor in {2, 4}.  ATen's index / weight arithmetic and summation order.  The adjoint of the
 * interpolation is fs_interp3d_bwd (times scale); d out / d prev is the identity. */
int fs_upsample3d_scale_add(const float* small, const float* prev, float* out, int B, int C,
                            int Din, int Hin, int Win, int factor, float scale, fs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * f3. Laplacian-pyramid L1 loss of Flow-2D: Flow-2D/model/laplacian.py:10-88
 * (gauss_kernel :10-19, downsample :21-22, upsample :24-36, conv_gauss :38-47,
 * laplacian_pyramid :49-74, LapLoss.forward :81-88).
 *   loss = sum_{l < levels} mean | pyr_l(input) - pyr_l(target) |  computed as ONE pyramid of
 *   (input - target) -- every pyramid step is linear.
 * input, target [N,H,W] (N = B*C; the 5x5 filter is depthwise), target may be NULL (= zeros).
 * Every level needs extent >= 3 (reflect padding by 2; torch raises there too) -> FS_ERR_SHAPE.
 * fs_laploss2d_sizes: host-only; element counts of the three caller-owned fp32 buffers.
 *   sgn  : sign(pyr_l) / numel_l for every level, written by fwd, read by bwd (save it);
 *   ws   : scratch (fwd: ws_fwd_floats, bwd: ws_bwd_floats), contents need not survive.
 * fwd : loss[0] = the loss (loss[1] is scratch; pass 2 floats).
 * bwd : grad_loss = device pointer to d(objective)/d(loss) (1 float);
 *       grad_diff [N,H,W] = d/d(input) = -d/d(target).  Gathers only, deterministic.
 */
int fs_laploss2d_sizes(int N, int H, int W, int levels, long long* sgn_floats,
                       long long* ws_fwd_floats, long long* ws_bwd_floats);
int fs_laploss2d_fwd(const float* input, const float* target, float* sgn, float* ws, float* loss,
                     int N, int H, int W, int levels, fs_stream_t stream);
int fs_laploss2d_bwd(const float* sgn, const float* grad_loss, float* ws, float* grad_diff,
                     int N, int H, int W, int levels, fs_stream_t stream);

/* f3, second half: the 3-D Laplacian-pyramid L1 loss.  PARITY UNPINNED: Flow-3D/model/laplacian.py:37-91 is
 * dead code in the reference (commented out of Model.update, Flow-3D/model/RIFE.py:126,132) whose conv_gauss
 * (:44-58) ignores its kernel, round-trips through scipy.ndimage.gaussian_filter on the CPU over all five
 * axes and detaches the result.  Built here: the 3-D analogue of the 2-D loss above -- filtered = G3(reflect
 * pad 2)(cur) with G3 = g (x) g (x) g, g = [1,4,6,4,1]/16; down = filtered[::2,::2,::2]; up = (8 G3)(reflect
 * pad 2)(zero-interleave(down)) (the reference's `x_up * 8`, laplacian.py:41); pyr = cur - up; loss =
 * sum_l mean |pyr_l(input) - pyr_l(target)|.  Same buffer protocol as fs_laploss2d_*; input, target
 * [N,D,H,W] (N = B*C); every level needs extent >= 3 per axis. */
int fs_laploss3d_sizes(int N, int D, int H, int W, int levels, long long* sgn_floats,
                       long long* ws_fwd_floats, long long* ws_bwd_floats);
int fs_laploss3d_fwd(const float* input, const float* target, float* sgn, float* ws, float* loss,
                     int N, int D, int H, int W, int levels, fs_stream_t stream);
int fs_laploss3d_bwd(const float* sgn, const float* grad_loss, float* ws, float* grad_diff,
                     int N, int D, int H, int W, int levels, fs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Forward pass of the IFNet-3D convolutions as an implicit GEMM on the fp32 matrix cores
 * (companion of fs_conv3d_wrw; `conv()` / IFBlock.conv0 / convblock in Flow-3D/model/IFNet.py:13-29,
 * 31-76: Conv3d(k=3,s=1,p=1) and Conv3d(k=4,s=2,p=1), NCDHW fp32).
 *   y[b,co,o] = bias[co] + sum_{ci,k} W[co,ci,k] * x[b,ci, o*stride + k - pad]   (zero padding)
 * x [B,Cin,Di,Hi,Wi] -> y [B,Cout,Do,Ho,Wo], Do = (Di + 2 pad - kernel)/stride + 1 (checked).
 * (kernel,stride) in {(3,1),(4,2)}.  bias may be NULL.
 * wmode 0: w is [Cout][Cin][k^3] (Conv3d forward; also the INPUT GRADIENT of a ConvTranspose3d, whose
 *          weight tensor [Cin_t][Cout_t][k^3] already reads as [out][in] of that convolution);
 * wmode 1: w is [Cin][Cout][k^3] and is applied flipped (tap k^3-1-k): the INPUT GRADIENT of a
 *          stride-1 "same" Conv3d is this convolution of grad_out with the layer's own weight.
 * ws: device scratch of fs_conv3d_fwd_ws_floats(Cin, Cout, kernel) floats (re-laid-out weights).
 */
long long fs_conv3d_fwd_ws_floats(int Cin, int Cout, int kernel);
int fs_conv3d_fwd(const float* x, const float* w, const float* bias, float* y, float* ws,
                  int B, int Cin, int Cout, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                  int kernel, int stride, int pad, int wmode, fs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * ConvTranspose3d(k=4, s=2, p=1) forward -- the IFNet-3D heads (`conv1` / `conv2` of IFBlock,
 * Flow-3D/model/IFNet.py:62-75) -- which is also the INPUT GRADIENT of Conv3d(k=4, s=2, p=1)
 * (IFBlock.conv0, Flow-3D/model/IFNet.py:34-37) applied to grad_out.
 *   y[b,co,o] = bias[co] + sum_{ci,k : (o + 1 - k) even} x[b,ci,(o + 1 - k)/2] * w[ci,co,k]
 * x [B,Cin,Di,Hi,Wi], w [Cin][Cout][4*4*4] (a ConvTranspose3d weight as stored; a Conv3d weight
 * [Cout_conv][Cin_conv][64] reads the same way for its input gradient), bias may be NULL,
 * y [B,Cout,Dout,Hout,Wout] with out = 2*in per axis (or 2*in + 1: input gradient of a convolution
 * whose odd input extent left its last plane unused -- that plane receives zeros).
 * Cout <= 32 (FS_ERR_ARG otherwise).  ws: fs_conv3d_tr_ws_floats(Cin, Cout) floats of device scratch
 * (0 for Cout <= 6: those run on the vector ALUs with scalar-loaded weights).
 */
/* fs_conv3d_fwd_prelu / fs_conv3d_tr_prelu: the same convolutions with the PReLU that follows every one of
 * them in IFNet (`conv()` / `deconv()`, Flow-3D/model/IFNet.py:13-29) applied in the epilogue:
 * y = conv(x) + bias (kept: the PReLU backward needs it) and z = y > 0 ? y : prelu_weight[c] * y are written
 * in the same pass (num_prelu_weights = 1 or Cout).  fwd_prelu is wmode 0 only; its `residual` (may be
 * NULL, z's shape) is added to z: the `convblock(x) + x` of IFBlock.forward (Flow-3D/model/IFNet.py:101-104).
 * fs_conv3d_fwd_add: y = conv(x) + bias + addend (y's shape) -- with wmode 1 the input gradient of such a
 * residual unit, whose skip branch contributes grad_out itself. */
int fs_conv3d_fwd_add(const float* x, const float* w, const float* bias, const float* addend, float* y,
                      float* ws, int B, int Cin, int Cout, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                      int kernel, int stride, int pad, int wmode, fs_stream_t stream);
/* The backward counterpart for the heads (Flow-3D/model/IFNet.py:66-77: deconv -> PReLU -> deconv): the input gradient
 * of the SECOND deconvolution (a strided convolution of its grad_out, weight read as [out][in], wmode 0) with the PReLU
 * backward of the layer in between folded into the epilogue -- grad_act_y = conv(x) * prelu'(act_y), plus the PReLU
 * weight gradient and the first deconvolution's bias gradient (deterministic: per-wave partials in `part`,
 * fs_conv3d_fwd_dprelu_part_floats floats, summed in a fixed order).  FS_ERR_UNSUPPORTED: no such kernel for this
 * shape / alignment -- use fs_conv3d_fwd followed by fs_prelu_bwd.
 * kernel 3 (stride 1, pad 1): the same for the INNER PReLU of an IFBlock residual unit (Flow-3D/model/IFNet.py:101-104,
 * `convblock(x) + x` with convblock = conv, PReLU, conv, PReLU): x = grad w.r.t. the second convolution's output,
 * w = that convolution's weight as stored (read flipped + transposed, as fs_conv3d_fwd wmode 1), act_y = the first
 * convolution's output. */
long long fs_conv3d_fwd_dprelu_part_floats(int B, int Cout, int Do, int Ho, int Wo);    /* kernel 4 */
long long fs_conv3d_fwd_dprelu_part_floats_k3(int B, int Cout, int Do, int Ho, int Wo); /* kernel 3 */
int fs_conv3d_fwd_dprelu(const float* x, const float* w, const float* act_y, const float* prelu_weight,
                         int num_prelu_weights, float* grad_act_y, float* grad_prelu_weight, float* grad_bias,
                         float* part, float* ws, int B, int Cin, int Cout, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                         int kernel, int stride, int pad, fs_stream_t stream);
int fs_conv3d_fwd_prelu(const float* x, const float* w, const float* bias, const float* prelu_weight,
                        const float* residual, float* y, float* z, float* ws,
                        int B, int Cin, int Cout, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                        int kernel, int stride, int pad, int num_prelu_weights, fs_stream_t stream);
int fs_conv3d_tr_prelu(const float* x, const float* w, const float* bias, const float* prelu_weight,
                       float* y, float* z, float* ws,
                       int B, int Cin, int Cout, int Di, int Hi, int Wi, int Dout, int Hout, int Wout,
                       int num_prelu_weights, fs_stream_t stream);
/* fs_conv3d_tr_add: y = conv_transpose(x) + bias + addend (addend has y's shape): the head's flow / mask
 * delta accumulated onto the running flow / mask (Flow-3D/model/IFNet.py:213-214, 228-229) in the epilogue. */
int fs_conv3d_tr_add(const float* x, const float* w, const float* bias, const float* addend, float* y,
                     float* ws, int B, int Cin, int Cout, int Di, int Hi, int Wi,
                     int Dout, int Hout, int Wout, fs_stream_t stream);
long long fs_conv3d_tr_ws_floats(int Cin, int Cout);
int fs_conv3d_tr(const float* x, const float* w, const float* bias, float* y, float* ws,
                 int B, int Cin, int Cout, int Di, int Hi, int Wi, int Dout, int Hout, int Wout,
                 fs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Backward of torch.nn.PReLU(num_parameters = C or 1) as used after every IFNet convolution
 * (`conv()` in Flow-2D/model/IFNet.py and Flow-3D/model/IFNet.py):  y = x > 0 ? x : a[c] x.
 *   grad_x[e] = x > 0 ? g : a[c] g ;  grad_weight[c] = sum_{b,spatial} (x > 0 ? 0 : x g).
 * x, grad_out, grad_x [B,C,S]; weight, grad_weight [num_weights]; ws: device scratch of
 * B*C*FS_PRELU_MAX_CHUNKS floats.  One pass + a tiny deterministic finishing kernel.
 * grad_bias (may be NULL) [C]: additionally sum_{b,spatial} grad_x per channel -- the bias gradient
 * of the convolution that produced x, for free in the same pass (ws must then hold
 * 2*B*C*FS_PRELU_MAX_CHUNKS floats).
 */
#define FS_PRELU_MAX_CHUNKS 64
int fs_prelu_bwd(const float* x, const float* grad_out, const float* weight,
                 float* grad_x, float* grad_weight, float* grad_bias, float* ws,
                 int B, int C, int S, int num_weights, fs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Weight gradient of the IFNet-3D convolutions (callers of the hot path; MIOpen's own weight-gradient
 * solvers have no gfx950 tuning) as an implicit GEMM on the fp32 matrix cores:
 *   dw[g, c, kz,ky,kx] += sum_{b,o} g[b,g,o] * src[b,c, o*stride + k - pad]     (zero outside src)
 * g [B,Cg,Do,Ho,Wo], src [B,Cs,Di,Hi,Wi], dw [Cg,Cs,k,k,k] (caller zero-fills; float atomics).
 *   torch.nn.Conv3d:          g = grad_output, src = input        -> dw = weight.grad [Cout,Cin,k,k,k]
 *   torch.nn.ConvTranspose3d: g = input,       src = grad_output  -> dw = weight.grad [Cin,Cout,k,k,k]
 * (kernel, stride) in {(3,1), (4,2)} -- the IFBlock layers (Flow-3D/model/IFNet.py:33-78).
 */
int fs_conv3d_wrw(const float* g, const float* src, float* dw, int B, int Cg, int Cs,
                  int Do, int Ho, int Wo, int Di, int Hi, int Wi,
                  int kernel, int stride, int pad, fs_stream_t stream);
/* Which kernel fs_conv3d_wrw runs for these arguments (its own dispatch, nothing launched, pointers only inspected
 * for alignment; `dw` is not needed): FS_WRW_KERNEL_* >= 0, or -(FS_ERR_*) for arguments it would refuse.  For flop
 * accounting (the Winograd form executes half the direct form's multiply-adds) and for tests that must know which
 * kernel they exercised. */
enum { FS_WRW_KERNEL_BRICK = 0,   /* register-staged position bricks (any shape) */
       FS_WRW_KERNEL_DMA = 1,     /* loader-wave form, direct implicit GEMM */
       FS_WRW_KERNEL_WINO23 = 2,  /* Winograd F(2,3) along x (ablation build only) */
       FS_WRW_KERNEL_WINO43 = 3   /* Winograd F(4,3) along x: the 64 -> 64 k3 trunk layers */ };
int fs_conv3d_wrw_kernel_id(const float* g, const float* src, int B, int Cg, int Cs,
                            int Do, int Ho, int Wo, int Di, int Hi, int Wi, int kernel, int stride, int pad);

/* IFBlock's first convolution without its `torch.cat` (Flow-3D/model/IFNet.py:183 `x = torch.cat((x, flow), 1)`
 * over `torch.cat((img0, img1, warped_img0, warped_img1, mask), 1)`, :190-191): the Cin <= 12 input channels are
 * planes of the tensors they already live in.  src[c] = device pointer to channel c of sample 0 (16-byte aligned),
 * batch_strides[c] = the batch stride of its tensor in floats (a multiple of 4, >= D*H*W); both are HOST arrays of
 * Cin entries, read at launch.  fs_conv3d_fwd_prelu_ms == fs_conv3d_fwd_prelu without `residual`;
 * fs_conv3d_wrw_ms == fs_conv3d_wrw with that source.  FS_ERR_UNSUPPORTED when the shape has no loader-wave
 * kernel (k = 4, <= 32 output / gradient channels, W % 4 == 0, enough bricks): concatenate and use the plain
 * entry points. */
int fs_conv3d_fwd_prelu_ms(const float* const* src, const long long* batch_strides, const float* w, const float* bias,
                           const float* prelu_weight, float* y, float* z, float* ws,
                           int B, int Cin, int Cout, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                           int kernel, int stride, int pad, int num_prelu_weights, fs_stream_t stream);
int fs_conv3d_wrw_ms(const float* g, const float* const* src, const long long* batch_strides, float* dw,
                     int B, int Cg, int Cs, int Do, int Ho, int Wo, int Di, int Hi, int Wi,
                     int kernel, int stride, int pad, fs_stream_t stream);

/* Deterministic weight gradient (round 4).  fs_conv3d_wrw / _ms split the positions into runs whose partial tiles
 * meet in dw through float atomics: the order of arrival decides the last bits.  Here every run STORES its partial
 * tile into its own copy of dw inside a caller-owned workspace and one more launch adds the copies in run order:
 * same kernels, same dispatch, bitwise reproducible from run to run; dw is overwritten (no zero fill).
 *   fs_conv3d_wrw_det_ws_floats: floats of workspace this call needs (runs x Cg*Cs*k^3; its own dispatch, nothing
 *   launched), or -(FS_ERR_*).  src_planes / batch_strides both NULL: one `src` tensor; both non-NULL: the
 *   multi-source form of fs_conv3d_wrw_ms (`src` ignored).
 * The Python binding takes this path when torch.are_deterministic_algorithms_enabled(). */
long long fs_conv3d_wrw_det_ws_floats(const float* g, const float* src, const float* const* src_planes,
                                      const long long* batch_strides, int B, int Cg, int Cs,
                                      int Do, int Ho, int Wo, int Di, int Hi, int Wi, int kernel, int stride, int pad);
int fs_conv3d_wrw_det(const float* g, const float* src, const float* const* src_planes, const long long* batch_strides,
                      float* dw, float* ws, long long ws_floats, int B, int Cg, int Cs,
                      int Do, int Ho, int Wo, int Di, int Hi, int Wi, int kernel, int stride, int pad,
                      fs_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Weight preparation in one launch per optimiser step.  fs_conv3d_fwd* / fs_conv3d_tr* re-lay their weights into
 * `ws` with a small launch in front of every convolution (~110 launches of ~5 us per 256^3 Flow-3D step).  A caller
 * that keeps one `ws` per (layer, use) alive can instead (1) ask each entry point ONCE which re-layout it would run
 * for its shape -- fs_conv3d_fwd_wprep_jobs / fs_conv3d_tr_wprep_jobs take the arguments of fs_conv3d_fwd /
 * fs_conv3d_tr (the same dispatch code runs, nothing is launched) and write FsWprepJob records to a HOST array,
 * returning their number (0: this shape reads `w` directly; < 0: -FS_ERR_*; more than `cap`: array too small);
 * (2) after every weight update run all jobs with ONE fs_conv3d_wprep_batch launch (`jobs` in DEVICE memory);
 * (3) call the convolutions with w = NULL: "`ws` is already prepared".  Passing w != NULL keeps the classic
 * behaviour.  The fused variants (_add, _prelu, _prelu_ms, _dprelu) use the layout of their plain entry point
 * (fs_conv3d_fwd_dprelu: wmode = kernel == 3).  `x` is only inspected for its alignment; `has_prelu_out`: the
 * fs_conv3d_tr_prelu form (its two outputs rule out the all-parities kernel). */
typedef struct FsWprepJob {
  const float* w; /* source weights (device) */
  float* ws;      /* destination slab (device) */
  int kind;       /* layout, private to the library */
  int total;      /* floats written */
  int p[6];       /* layout parameters, private to the library */
} FsWprepJob;
int fs_conv3d_fwd_wprep_jobs(FsWprepJob* jobs_host, int cap, const float* x, const float* w, float* ws,
                             int B, int Cin, int Cout, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                             int kernel, int stride, int pad, int wmode);
int fs_conv3d_tr_wprep_jobs(FsWprepJob* jobs_host, int cap, const float* x, const float* w, float* ws, int B, int Cin,
                            int Cout, int Di, int Hi, int Wi, int Dout, int Hout, int Wout, int has_prelu_out);
int fs_conv3d_wprep_batch(const FsWprepJob* jobs_dev, int njobs, fs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FLOWSCI_HIP_H */
