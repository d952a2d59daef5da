// Host-side plumbing of libinternvideo_hip.so: error reporting, device query, hardware-semantics probes.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "common.h"
#include "../../include/internvideo_hip.h"
#include "../../include/internvideo_hip_debug.h"

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
extern "C" int ivh_abi_version(void) { return IVH_ABI_VERSION; }
// Device-side dropout epoch (common.h DropCfg): every dropout mask of the text-tower kernels and of the dropout attention kernels is
// hash(seed + *epoch * 0x9E3779B1, element index) while a pointer is registered; NULL (default) = the seed alone.
namespace ivh_host { static const unsigned* g_drop_epoch = nullptr; const unsigned* dropout_epoch() { return g_drop_epoch; } }
extern "C" int ivh_set_dropout_epoch(const void* dev_u32) { ivh_host::g_drop_epoch = (const unsigned*)dev_u32; return 0; }

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
// c[i][j] = sum_k a[i][k] b[j][k] with one v_mfma_f32_32x32x16_bf16 (a, b: [32][16] bf16 row-major), written out under the layout the
// 32x32 attention kernels assume: A lane -> row lane & 31, k = 8 (lane >> 5) + e; C reg r of lane -> row (r & 3) + 8 (r >> 2) + 4 (lane >> 5),
// col lane & 31.
typedef __attribute__((ext_vector_type(16))) float probe_f32x16;
__global__ void probe_mfma32_kernel(const bf16_t* a, const bf16_t* b, float* c) {
  const int lane = threadIdx.x;
  const int i = lane & 31, hi = lane >> 5;
  const s16x8 af = *reinterpret_cast<const s16x8*>(a + i * 16 + 8 * hi);
  const s16x8 bf = *reinterpret_cast<const s16x8*>(b + i * 16 + 8 * hi);
  probe_f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bf), acc, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 16; ++r) c[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + i] = acc[r];   // c[row = a-row][col = b-row]
}
// Known-rate MFMA stream for counter calibration (tools/pmc_mfma.py): every wave issues `iters` x 8 back-to-back 32x32x16 bf16 MFMAs on
// four independent accumulators; one wave per SIMD (256-thread workgroups, one per CU by the grid size).  FLOPs = 2 * 32 * 32 * 16 per MFMA.
__global__ __launch_bounds__(256) void probe_mfma_rate_kernel(int iters, float* sink) {
  probe_f32x16 acc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  s16x8 a, b;
#pragma unroll
  for (int e = 0; e < 8; ++e) { a[e] = (short)(0x3c00 + threadIdx.x + e); b[e] = (short)(0x3b80 + 3 * threadIdx.x + e); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[q], 0, 0, 0);
  }
  float t = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) t += acc[q][r];
  if (t == 123.456f) sink[0] = t;                 // keeps the accumulators live without a store on the timed path
}
// Same FLOPs per wave and iteration (8 x 32x32x16 = 16 x 16x16x32 = 262144) for the two bf16 MFMA shapes, 1 or 2 waves per SIMD: the
// sustained (power / clock limited) rate each shape reaches with every CU busy -- which shape a GEMM inner loop should be built on.
template <int SHAPE>
__global__ void probe_mfma_rate2_kernel(int iters, float* sink) {
  s16x8 a, b;
#pragma unroll
  for (int e = 0; e < 8; ++e) { a[e] = (short)(0x3c00 + threadIdx.x * 7 + e * 13); b[e] = (short)(0x3b80 + 3 * threadIdx.x + e * 5); }
  float t = 0.f;
  if constexpr (SHAPE == 0) {
    probe_f32x16 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[q], 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) t += acc[q][r];
  } else {
    f32x4 acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = mfma16(a, b, acc[q]);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) t += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
  }
  if (t == 123.456f) sink[0] = t;
}
}  // namespace ivh

extern "C" int ivh_probe_mfma_rate2(int shape, int waves_per_simd, int iters, int workgroups, float* sink, void* stream) {
  IVH_REQUIRE((shape == 0 || shape == 1) && (waves_per_simd == 1 || waves_per_simd == 2) && iters > 0 && workgroups > 0 && sink,
              "probe_mfma_rate2: bad args");
  if (shape == 0)
    hipLaunchKernelGGL(ivh::probe_mfma_rate2_kernel<0>, dim3(workgroups), dim3(256 * waves_per_simd), 0, (hipStream_t)stream, iters, sink);
  else
    hipLaunchKernelGGL(ivh::probe_mfma_rate2_kernel<1>, dim3(workgroups), dim3(256 * waves_per_simd), 0, (hipStream_t)stream, iters, sink);
  return ivh_host::check_launch("probe_mfma_rate2");
}

// A stand-in for a collective's kernels (bench.py `comm_contention`): `workgroups` workgroups of 256 threads copy `bytes` from src to dst
// (grid-stride, 16 bytes per lane and trip) -- a bounded number of CUs kept busy with memory traffic beside the training step.
namespace ivh {
__global__ __launch_bounds__(256) void probe_cu_hog_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, long n16) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) dst[i] = src[i];
}
}  // namespace ivh
extern "C" int ivh_probe_cu_hog(const void* src, void* dst, int64_t bytes, int workgroups, void* stream) {
  IVH_REQUIRE(src && dst && bytes > 0 && bytes % 16 == 0 && workgroups > 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0, "probe_cu_hog: bad args");
  hipLaunchKernelGGL(ivh::probe_cu_hog_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, (const ivh::u32x4*)src, (ivh::u32x4*)dst, (long)(bytes / 16));
  return ivh_host::check_launch("probe_cu_hog");
}

extern "C" int ivh_probe_mfma32(const uint16_t* a, const uint16_t* b, float* c, void* stream) {
  hipLaunchKernelGGL(ivh::probe_mfma32_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, c);
  return ivh_host::check_launch("probe_mfma32");
}
extern "C" int ivh_probe_mfma_rate(int iters, int workgroups, float* sink, void* stream) {
  IVH_REQUIRE(iters > 0 && workgroups > 0 && sink, "probe_mfma_rate: bad args");
  hipLaunchKernelGGL(ivh::probe_mfma_rate_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, iters, sink);
  return ivh_host::check_launch("probe_mfma_rate");
}
extern "C" int ivh_probe_tr16(const uint16_t* in, uint16_t* out, void* stream) {
  hipLaunchKernelGGL(ivh::probe_tr16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, in, out);
  return ivh_host::check_launch("probe_tr16");
}
extern "C" int ivh_probe_mfma16(const uint16_t* a, const uint16_t* b, float* c, void* stream) {
  hipLaunchKernelGGL(ivh::probe_mfma16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, c);
  return ivh_host::check_launch("probe_mfma16");
}
