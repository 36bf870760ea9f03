// Error reporting and misc entry points of libhnd_hip.so.
#include "common.h"

#include <string.h>

namespace hnd {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int relay_timeouts(int reset);     // conv_bstream.hip

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) return HND_OK;
  set_error("%s: launch failed: %s", what, hipGetErrorString(e));
  return HND_ERR_LAUNCH;
}

}  // namespace hnd

extern "C" {

const char* hnd_last_error_string(void) { return hnd::g_err; }

int hnd_abi_version(void) { return HND_ABI_VERSION; }

int hnd_sync_check(void* stream) {
  hipError_t e = hipStreamSynchronize(hnd::as_stream(stream));
  if (e == hipSuccess) e = hipGetLastError();
  if (e == hipSuccess && hnd::relay_timeouts(0) != 0) {
    hnd::set_error("hnd_sync_check: a B-streamed GEMM gave up waiting for a neighbour's partial tile (relay time-out): "
                   "its output is invalid; hnd_relay_timeouts(1) acknowledges");
    return HND_ERR_ASYNC;
  }
  if (e == hipSuccess) return HND_OK;
  hnd::set_error("hnd_sync_check: %s", hipGetErrorString(e));
  return HND_ERR_ASYNC;
}

int hnd_relay_timeouts(int reset) { return hnd::relay_timeouts(reset); }

size_t hnd_workspace_size(int op, const void* desc, int64_t arg) {
  switch (op) {
    case HND_OP_CONV2D_WGRAD: return desc ? hnd_conv2d_wgrad_workspace((const hnd_wgrad_desc*)desc) : 0;
    case HND_OP_MSE: return hnd_mse_scratch_elems() * sizeof(double);
    case HND_OP_QUANTIZE_U8: return hnd_minmax_scratch_elems() * sizeof(float);
    case HND_OP_CHANNEL_SUM: return hnd_channel_sum_scratch_elems((int)arg) * sizeof(float);
    case HND_OP_COMM_UNIQUE_ID: return 128;
    case HND_OP_NMS: return hnd_nms_workspace(arg);
    default: return 0;
  }
}

const char* hnd_device_arch(void) {
  static thread_local char arch[256] = "";
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    arch[0] = 0;
    (void)hipGetLastError();
    return arch;
  }
  strncpy(arch, prop.gcnArchName, sizeof(arch) - 1);
  return arch;
}

}  // extern "C"
