// Which VALU instruction types run beside another wave's MFMAs on the same SIMD?  (round 6)
// 512-thread workgroups, one per CU (100 KiB of LDS each): waves w and w + 4 share a SIMD (tools/probes/wave_simd_map.hip).  Waves 0-3 run role RA,
// waves 4-7 role RB; a role is `iters` x 32 instructions of one type on independent registers.  Prints the time of every pair next to the two
// roles alone: pair / (a + b) = 1 means the two streams serialise, pair / max(a, b) = 1 means they overlap completely.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_valu_mix.hip -o tools/probes/mfma_valu_mix.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f2;
enum { IDLE = 0, MFMA = 1, FMA = 2, PKFMA = 3, EXP = 4, MAXF = 5, CVT = 6, PKMUL = 7, MFMA16 = 8, NROLE = 9 };
static const char* names[NROLE] = {"idle", "mfma32x32x16", "v_fma_f32", "v_pk_fma_f32", "v_exp_f32", "v_max_f32", "v_cvt_pk_bf16_f32", "v_pk_mul_f32", "mfma16x16x32"};

template <int R> __device__ __forceinline__ float run_role(int iters, float seed) {
  float acc = 0.f;
  if constexpr (R == MFMA) {
    f32x16 c[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) c[i][r] = seed;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)seed; b[e] = (__bf16)(seed + 1.f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) acc += c[i][0] + c[i][7];
  } else if constexpr (R == MFMA16) {
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    f32x4 c[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) c[i][r] = seed;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)seed; b[e] = (__bf16)(seed + 1.f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) acc += c[i][0];
  } else if constexpr (R == FMA || R == EXP || R == MAXF) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = seed + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if constexpr (R == FMA) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(seed));
          else if constexpr (R == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
          else asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(seed));
        }
    }
    for (int i = 0; i < 8; ++i) acc += x[i];
  } else if constexpr (R == PKFMA || R == PKMUL) {
    f2 x[8];
    const f2 sv = {seed, seed};
    for (int i = 0; i < 8; ++i) x[i] = f2{seed + i, seed - i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if constexpr (R == PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(sv));
          else asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(sv));
        }
    }
    for (int i = 0; i < 8; ++i) acc += x[i][0] + x[i][1];
  } else if constexpr (R == CVT) {
    float x[8]; unsigned y[8];
    for (int i = 0; i < 8; ++i) { x[i] = seed + i; y[i] = 0; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(y[i]) : "v"(x[i]));
    }
    for (int i = 0; i < 8; ++i) acc += (float)y[i];
  }
  return acc;
}

template <int RA, int RB, int PA = 0> __global__ __launch_bounds__(512) void mix_kernel(int iters, float* sink, int iters_b = 0) {
  extern __shared__ char pad[];
  const int wave = threadIdx.x >> 6;
  if (PA && wave < 4) __builtin_amdgcn_s_setprio(PA);        // group A (the MFMA stream in the balanced runs) at raised priority
  float seed = (float)(threadIdx.x & 3) * 0.25f + 0.5f;
  float r;
  if (wave < 4) r = run_role<RA>(iters, seed); else r = run_role<RB>(iters_b ? iters_b : iters, seed);
  if (r == 12345.678f) sink[threadIdx.x] = r + pad[0];
}

template <int RA, int RB, int PA = 0> float time_pair(int iters, float* sink, int iters_b = 0) {
  hipFuncSetAttribute((const void*)mix_kernel<RA, RB, PA>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  mix_kernel<RA, RB, PA><<<256, 512, 100 * 1024>>>(iters, sink, iters_b);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(s);
    mix_kernel<RA, RB, PA><<<256, 512, 100 * 1024>>>(iters, sink, iters_b);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    if (ms < best) best = ms;
  }
  return best;
}

template <int RB> void row(int iters, float* sink, float mfma_alone, float mfma16_alone) {
  const float alone = time_pair<IDLE, RB>(iters, sink);
  const float both = time_pair<MFMA, RB>(iters, sink);
  const float both16 = time_pair<MFMA16, RB>(iters, sink);
  const float self2 = time_pair<RB, RB>(iters, sink);
  printf("{\"valu\": \"%s\", \"alone_ms\": %.3f, \"two_waves_of_it_ms\": %.3f, \"beside_mfma32_ms\": %.3f, \"over_sum\": %.3f, \"over_max\": %.3f, \"beside_mfma16_ms\": %.3f, \"over_sum16\": %.3f}\n",
         names[RB], alone, self2, both, both / (alone + mfma_alone), both / fmaxf(alone, mfma_alone), both16, both16 / (alone + mfma16_alone));
}

// balanced: the VALU wave runs `mult` times the iterations, so that its time alone is about the MFMA wave's: how much VALU issue is left beside MFMAs
template <int RM, int RB> void balanced(int iters, float* sink, const char* mname) {
  const float m = time_pair<RM, IDLE>(iters, sink);
  const float v1 = time_pair<IDLE, RB>(iters, sink, iters);
  const int ib = (int)(iters * m / v1 + 0.5f);
  const float v = time_pair<IDLE, RB>(iters, sink, ib);
  const float both = time_pair<RM, RB>(iters, sink, ib);
  const float both_p = time_pair<RM, RB, 3>(iters, sink, ib);               // the MFMA waves at s_setprio 3
  const float half = time_pair<RM, RB>(iters, sink, ib / 2);                 // half the VALU work: does it hide?
  const float half_p = time_pair<RM, RB, 3>(iters, sink, ib / 2);
  printf("{\"balanced\": \"%s beside %s\", \"mfma_alone_ms\": %.3f, \"valu_alone_ms\": %.3f, \"valu_iters\": %d, \"pair_ms\": %.3f, \"over_sum\": %.3f, \"over_max\": %.3f, "
         "\"pair_mfma_prio3_ms\": %.3f, \"half_valu_pair_ms\": %.3f, \"half_valu_pair_mfma_prio3_ms\": %.3f}\n",
         names[RB], mname, m, v, ib, both, both / (m + v), both / fmaxf(m, v), both_p, half, half_p);
}

int main() {
  float* sink; hipMalloc(&sink, 4096);
  const int iters = 4000;
  const float m = time_pair<MFMA, IDLE>(iters, sink);
  const float m16 = time_pair<MFMA16, IDLE>(iters, sink);
  const float mm = time_pair<MFMA, MFMA>(iters, sink);
  printf("{\"mfma32_alone_ms\": %.3f, \"mfma16_alone_ms\": %.3f, \"two_mfma32_waves_ms\": %.3f, \"iters\": %d, \"instructions_per_iter\": 32}\n", m, m16, mm, iters);
  row<FMA>(iters, sink, m, m16);
  row<PKFMA>(iters, sink, m, m16);
  row<PKMUL>(iters, sink, m, m16);
  row<EXP>(iters, sink, m, m16);
  row<MAXF>(iters, sink, m, m16);
  row<CVT>(iters, sink, m, m16);
  balanced<MFMA, FMA>(iters, sink, "mfma32x32x16");
  balanced<MFMA, PKFMA>(iters, sink, "mfma32x32x16");
  balanced<MFMA, EXP>(iters, sink, "mfma32x32x16");
  balanced<MFMA, MAXF>(iters, sink, "mfma32x32x16");
  balanced<MFMA, CVT>(iters, sink, "mfma32x32x16");
  balanced<MFMA16, FMA>(iters, sink, "mfma16x16x32");
  balanced<MFMA16, PKFMA>(iters, sink, "mfma16x16x32");
  balanced<MFMA16, EXP>(iters, sink, "mfma16x16x32");
  return 0;
}
