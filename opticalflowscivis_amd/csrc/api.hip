// api.hip -- version / error-string entry points of the C-ABI.
#include "common.hpp"

extern "C" int fs_version(void) {
  FS_ENTER(); return FS_ABI_VERSION; }

extern "C" const char* fs_error_string(int code) {
  switch (code) {
    case FS_OK: return "ok";
    case FS_ERR_NULLPTR: return "required pointer is NULL";
    case FS_ERR_SHAPE: return "size out of the supported range";
    case FS_ERR_ARG: return "invalid option / mode";
    case FS_ERR_LAUNCH: return "kernel launch failed";
    case FS_ERR_UNSUPPORTED: return "no fused kernel for this shape / alignment (use the unfused entry points)";
    default: return "unknown error code";
  }
}

namespace {
#include "wprep.hpp"
}

// All weight re-layouts of a model in one launch (see FsWprepJob in flowsci_hip.h): grid.y = job.
extern "C" int fs_conv3d_wprep_batch(const FsWprepJob* jobs_dev, int njobs, fs_stream_t stream) {
  FS_ENTER();
  FS_REQUIRE_PTR(jobs_dev);
  if (njobs < 1 || njobs > 65535) return FS_ERR_SHAPE;
  // 64 workgroups x 256 threads stride over a job (the largest slab, 128 -> 128 k3, is 442 K floats)
  hipLaunchKernelGGL(wprep_batch_kernel, dim3(64, njobs), dim3(256), 0, (hipStream_t)stream, jobs_dev);
  FS_LAUNCH_CHECK();
  return FS_OK;
}
