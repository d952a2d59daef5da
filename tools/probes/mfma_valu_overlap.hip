// Do MFMA and VALU instructions overlap on one gfx950 SIMD?  (measurement aid for the attention kernels -- not part of the library)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_valu_overlap tools/probes/mfma_valu_overlap.hip && /tmp/mfma_valu_overlap
// One workgroup per CU, 8 waves = 2 per SIMD (waves w and w + 4 share a SIMD).  Roles per wave: M = a stream of independent
// v_mfma_f32_32x32x16_bf16 (8 accumulators), V = a stream of independent fp32 VALU work with the softmax mix (1 v_exp_f32 per 2 packed fma +
// 2 plain ops), I = idle (exits at once), X = both streams interleaved in ONE wave (instruction-level overlap inside a wave).
// Modes: MI (MFMA alone), IV (VALU alone), MV (one wave of each per SIMD), MM, VV, XI, XX.  If the pipes overlap, MV ~ max(MI, IV) and
// XI ~ max; if they serialise, MV ~ MI + IV.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ void mfma_block(f32x16 (&acc)[8], s16x8 a, s16x8 b) {
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
}
// 8 elements: 8 exp + 8 packed fma (16 flop pairs) + 16 plain VALU ~ (8 x 16 + 8 x 4 + 16 x 4) = 224 issue cycles
__device__ __forceinline__ void valu_block(float (&v)[16], float s) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    f32x2 p = {v[2 * i], v[2 * i + 1]};
    p = p * f32x2{s, s} + f32x2{0.25f, -0.25f};
    v[2 * i] = __builtin_amdgcn_exp2f(p[0]);
    v[2 * i + 1] = fmaxf(p[1], v[2 * i]) + s;
    v[2 * i + 1] = v[2 * i + 1] * 0.999f;
  }
}

template <int ROLE_LO, int ROLE_HI>      // roles of waves 0-3 / 4-7: 0 idle, 1 MFMA, 2 VALU, 3 both interleaved
__global__ __launch_bounds__(512) void probe(float* sink, int iters, float s) {
  const int wave = threadIdx.x >> 6;
  const int role = wave < 4 ? ROLE_LO : ROLE_HI;
  if (role == 0) return;
  f32x16 acc[8];
  float v[16];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = 0.001f * (threadIdx.x + i);
  s16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
  asm volatile("" : "+v"(a), "+v"(b));
  for (int it = 0; it < iters; ++it) {
    if (role == 1) { mfma_block(acc, a, b); }                        // 8 MFMA = 256 matrix-pipe cycles
    else if (role == 2) { valu_block(v, s); }                       // ~224 VALU issue cycles
    else { mfma_block(acc, a, b); valu_block(v, s); }
  }
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][7];
#pragma unroll
  for (int i = 0; i < 16; ++i) r += v[i];
  if (r == 12345.678f) sink[0] = r;
}

template <int LO, int HI>
static float run(float* sink, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((probe<LO, HI>), dim3(256), dim3(512), 0, 0, sink, iters, 1.0001f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<LO, HI>), dim3(256), dim3(512), 0, 0, sink, iters, 1.0001f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* sink;
  hipMalloc(&sink, 4);
  const int iters = 20000;
  const float mi = run<1, 0>(sink, iters), iv = run<0, 2>(sink, iters), mv = run<1, 2>(sink, iters), mm = run<1, 1>(sink, iters),
              vv = run<2, 2>(sink, iters), xi = run<3, 0>(sink, iters), xx = run<3, 3>(sink, iters);
  printf("{\"iters\": %d, \"MI_ms\": %.3f, \"IV_ms\": %.3f, \"MV_ms\": %.3f, \"MM_ms\": %.3f, \"VV_ms\": %.3f, \"XI_ms\": %.3f, \"XX_ms\": %.3f, "
         "\"MV_over_max\": %.3f, \"MV_over_sum\": %.3f, \"XI_over_sum\": %.3f, \"XX_over_2sum\": %.3f}\n",
         iters, mi, iv, mv, mm, vv, xi, xx, mv / fmaxf(mi, iv), mv / (mi + iv), xi / (mi + iv), xx / (2 * (mi + iv)));
  return 0;
}
