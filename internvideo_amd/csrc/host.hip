// Host-side plumbing of libinternvideo_hip.so: error reporting, device query, hardware-semantics probes.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "common.h"
#include "../../include/internvideo_hip.h"

namespace ivh_host {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return -2;
  }
  return 0;
}
}  // namespace ivh_host

extern "C" const char* ivh_last_error(void) { return ivh_host::g_err; }
extern "C" int ivh_version(void) { return 100; }

extern "C" int ivh_device_info(int* n_cu, int* lds_bytes, char* arch, int arch_len) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    ivh_host::set_error("no HIP device");
    return -1;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    ivh_host::set_error("hipGetDeviceProperties failed");
    return -1;
  }
  if (n_cu) *n_cu = prop.multiProcessorCount;
  if (lds_bytes) *lds_bytes = (int)prop.sharedMemPerBlock;
  if (arch && arch_len > 0) {
    strncpy(arch, prop.gcnArchName, arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  return 0;
}

// ---- probes: pin the two hardware layouts every MFMA kernel in this library is built on --------------------
namespace ivh {
// in: 4 rows x 64 cols bf16 (row stride 64) ; every 16-lane group g reads the 4x16 block at cols 16g..16g+15.
// out[lane][j] = what lane received in element j.  Expected: in[j][16*(lane>>4) + (lane&15)].
__global__ void probe_tr16_kernel(const bf16_t* in, bf16_t* out) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[4 * 64];
  const int lane = threadIdx.x;
  for (int i = lane; i < 256; i += 64) tile[i] = in[i];
  __syncthreads();
  const int i = lane & 15, g = lane >> 4;
  const s16x4 t = lds_tr16(&tile[(i >> 2) * 64 + g * 16 + 4 * (i & 3)]);
#pragma unroll
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (bf16_t)t[j];
}
// c[n][m] = sum_k a[m][k] b[n][k] using D = mfma(first = b-rows, second = a-rows) exactly as gemm.hip does;
// written out with the "rows n = 4g + r, col m = lane & 15" assumption.  a, b: [16][32] bf16 row-major.
__global__ void probe_mfma16_kernel(const bf16_t* a, const bf16_t* b, float* c) {
  const int lane = threadIdx.x;
  const int i = lane & 15, g = lane >> 4;
  const s16x8 af = *reinterpret_cast<const s16x8*>(a + i * 32 + 8 * g);
  const s16x8 bf = *reinterpret_cast<const s16x8*>(b + i * 32 + 8 * g);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = mfma16(bf, af, acc);
#pragma unroll
  for (int r = 0; r < 4; ++r) c[(4 * g + r) * 16 + i] = acc[r];   // c[n][m]
}
}  // namespace ivh

extern "C" int ivh_probe_tr16(const uint16_t* in, uint16_t* out, void* stream) {
  hipLaunchKernelGGL(ivh::probe_tr16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, in, out);
  return ivh_host::check_launch("probe_tr16");
}
extern "C" int ivh_probe_mfma16(const uint16_t* a, const uint16_t* b, float* c, void* stream) {
  hipLaunchKernelGGL(ivh::probe_mfma16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, c);
  return ivh_host::check_launch("probe_mfma16");
}
