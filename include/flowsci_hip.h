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
  FS_ERR_LAUNCH = 4   /* hipGetLastError() != hipSuccess after the launch */
};

/* Library version (major*10000 + minor*100 + patch). */
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

int fs_warp2d_fwd(const float* in, const float* flow, const float* start, float* out,
                  int B, int C, int H, int W, int mode, int with_mask, fs_stream_t stream);
int fs_warp2d_bwd(const float* in, const float* flow, const float* start,
                  const float* grad_out, float* grad_in, float* grad_flow,
                  int B, int C, int H, int W, int mode, int with_mask, fs_stream_t stream);

/* Pair form for the IFNet call site (Flow-2D/model/IFNet.py:191-192, 230-231):
 *   warp(img0, flow[:, :2]); warp(img1, flow[:, 2:4])  with flow4 [B,4,H,W] used in place.
 * No mask, no start. */
int fs_warp2d_pair_fwd(const float* img0, const float* img1, const float* flow4,
                       float* out0, float* out1,
                       int B, int C, int H, int W, int mode, fs_stream_t stream);
int fs_warp2d_pair_bwd(const float* img0, const float* img1, const float* flow4,
                       const float* grad_out0, const float* grad_out1,
                       float* grad_img0, float* grad_img1, float* grad_flow4,
                       int B, int C, int H, int W, int mode, fs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FLOWSCI_HIP_H */
