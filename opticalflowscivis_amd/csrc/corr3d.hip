// corr3d.hip -- 3-D local-window correlation (cost volume) for gfx950.
//
// NEW CAPABILITY named by BASELINE.json (config 4: "3D correlation + trilinear warp HIP kernels");
// the reference has NO 3-D correlation (its Flow-3D IFNet has no cost volume at all), so
// the semantics are this build's generalisation of the 2-D layer (corr2d.hip / Corr_pyTorch):
//
//   out[b, ((dz+md)(2md+1) + (dy+md))(2md+1) + (dx+md), z, y, x]
//        = (1/C) sum_c f1[b,c,z,y,x] * f2[b,c,z+dz,y+dy,x+dx],     zero outside, dz-major.
//
// pinned to the reference only in the degenerate D = 1 case (== corr2d on the dz = 0 plane); oracle:
// oracle/corr.py::corr3d_closed.  The (2md+1)^3-channel output makes this a coarse-pyramid-level op
// (md = 4: 729 channels).
//
// A 3-D problem is 2md+1 2-D problems per slice -- f1 slice z against f2 slice z + dz -- so both directions run
// the tiled kernels of the 2-D layer (csrc/corr_q.hpp: buffer-load staging at dword alignment, packed-FP32 FMAs on
// aligned register pairs, 16-byte stores, XCD-contiguous tile order) with the channel stride D*H*W:
//   forward : one workgroup per (8 x 32 tile, slice z, displacement plane dz): 2md+1 waves = dy rows; the f1 tile
//             is re-staged per dz from L2 (1 KB per channel against the 81 x 1 KB it produces);
//   backward: one workgroup per (tile, slice z, 32 channels); it visits the displacement planes dz one after the
//             other -- the window of slice z + dz, the plane's (2md+1)^2 gradient planes streamed through LDS --
//             accumulating in registers; grad_f2 through the transposed-displacement identity; no atomics.
// Algorithmic HBM bytes per voxel: 4 (2C + (2md+1)^3) forward -- the output stream dominates (729 of the 793
// floats at C = 32, md = 4) -- and 4 (4C + (2md+1)^3) backward.  Arithmetic: (2md+1)^3 C FMAs per voxel forward,
// twice that backward: at md = 4 both directions are bound by VALU issue, not by HBM (23 K FMAs per voxel).
#include "common.hpp"

namespace {

#include "corr_q.hpp"

int check_shape(int B, int C, int D, int H, int W, int md) {
  if (B < 1 || C < 1 || D < 1 || H < 1 || W < 1) return FS_ERR_SHAPE;
  if (md < 1 || md > 4) return FS_ERR_ARG;
  const long long nd = 2 * md + 1, vol = (long long)D * H * W;
  // 32-bit byte offsets inside a tensor / inside one sample's cost volume; 1-D grids
  if ((long long)B * C * vol >= (1ll << 29) || nd * nd * nd * vol >= (1ll << 29)) return FS_ERR_SHAPE;
  if ((long long)fs::cdiv(W, TX) * fs::cdiv(H, TY) * D * B * nd * 2 * fs::cdiv(C, CBW) >= (1ll << 31)) return FS_ERR_SHAPE;
  return FS_OK;
}

template <int MD>
int launch(const float* f1, const float* f2, const float* gout, float* out, float* g1, float* g2, int B,
           int C, int D, int H, int W, bool bwd, hipStream_t st) {
  constexpr int ND = 2 * MD + 1;
  C2Set a = {};
  a.f1[0] = f1; a.f2[0] = f2; a.out[0] = out; a.gout[0] = gout; a.g1[0] = g1; a.g2[0] = g2;
  a.nsets = 1; a.D = D; a.NZ = ND;
  const long long tiles = (long long)fs::cdiv(W, TX) * fs::cdiv(H, TY) * D * B;
  if (!bwd) {
    constexpr int DPW = FS_C2_FWD_DPW;
    hipLaunchKernelGGL((corr_fwd_q_kernel<MD, DPW>), dim3((unsigned)(tiles * ND)), dim3(64 * ((2 * MD + DPW) / DPW)), 0, st,
                       a, B, C, H, W);
  } else {
    hipLaunchKernelGGL(corr_bwd_q_kernel<MD>, dim3((unsigned)(tiles * 2 * fs::cdiv(C, CBW))), dim3(512), 0, st, a, B, C, H, W);
  }
  FS_LAUNCH_CHECK();
  return FS_OK;
}

int dispatch(const float* f1, const float* f2, const float* gout, float* out, float* g1, float* g2, int B,
             int C, int D, int H, int W, int md, bool bwd, hipStream_t st) {
  switch (md) {
    case 1: return launch<1>(f1, f2, gout, out, g1, g2, B, C, D, H, W, bwd, st);
    case 2: return launch<2>(f1, f2, gout, out, g1, g2, B, C, D, H, W, bwd, st);
    case 3: return launch<3>(f1, f2, gout, out, g1, g2, B, C, D, H, W, bwd, st);
    default: return launch<4>(f1, f2, gout, out, g1, g2, B, C, D, H, W, bwd, st);
  }
}

}  // namespace

extern "C" int fs_corr3d_fwd(const float* f1, const float* f2, float* out, int B, int C, int D, int H,
                             int W, int max_displacement, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(f1); FS_REQUIRE_PTR(f2); FS_REQUIRE_PTR(out);
  const int rc = check_shape(B, C, D, H, W, max_displacement);
  if (rc != FS_OK) return rc;
  return dispatch(f1, f2, nullptr, out, nullptr, nullptr, B, C, D, H, W, max_displacement, false,
                  (hipStream_t)stream);
}

extern "C" int fs_corr3d_bwd(const float* f1, const float* f2, const float* grad_out, float* grad_f1,
                             float* grad_f2, int B, int C, int D, int H, int W, int max_displacement,
                             fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(f1); FS_REQUIRE_PTR(f2); FS_REQUIRE_PTR(grad_out);
  if (grad_f1 == nullptr && grad_f2 == nullptr) return FS_ERR_NULLPTR;
  const int rc = check_shape(B, C, D, H, W, max_displacement);
  if (rc != FS_OK) return rc;
  return dispatch(f1, f2, grad_out, nullptr, grad_f1, grad_f2, B, C, D, H, W, max_displacement, true,
                  (hipStream_t)stream);
}
