// Fused AdamW on flat HBM-resident buffers + global grad-norm pieces.
// Reference: DeepSpeed FusedAdam (adam_w_mode) configured by InternVideo2/single_modality/utils.py:821-871
// (betas (0.9, 0.98), eps 1e-6, bias correction, decoupled weight decay, gradient clipping 3.0 at :860-861).
// One launch per parameter group region; 16 B/param read (master, m, v) + 2..4 B grad, 14 B/param written
// (master, m, v, bf16 compute copy): purely HBM-bound.
#include <stdlib.h>
#include "common.h"
#include "../../include/internvideo_hip.h"

namespace ivh {

// U = independent 4-element groups per thread and trip (all loads of a trip are issued before the first use); NT = non-temporal
// loads / stores (30 GB stream through once per step: nothing of it is worth a cache line).  Element-wise arithmetic is identical for
// every variant: results are bit-identical.
// SCALED: layer-wise learning-rate decay (single_modality/optim_factory.py:24-98, engines/engine_for_finetuning.py:56: the step of a
// parameter group is lr * lr_scale, for the Adam term AND the decoupled weight decay).  The flat buffer is a run of segments (whole
// parameters, 64-element aligned, so the 4 elements of a group never straddle one); `seg_end[i]` = exclusive end offset of segment i
// inside the REGION this launch's slice was cut from, `seg_base` = offset of the slice in that region (zero1 shards start anywhere).
// The table (<= 1024 entries) sits in LDS and a group finds its segment by binary search: ~10 ds_reads beside 52 bytes of HBM traffic.
constexpr int ADAMW_MAX_SEG = 1024;
template <bool GRAD_BF16, int U = 1, bool NT = false, bool SCALED = false>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ master, float* __restrict__ m, float* __restrict__ v,
                                                    const void* __restrict__ grad, bf16_t* __restrict__ shadow, long n,
                                                    float lr, float b1, float b2, float eps, float wd, float bc1, float bc2,
                                                    float grad_scale, const float* __restrict__ clip_coef,
                                                    const long* __restrict__ seg_end = nullptr, const float* __restrict__ seg_scale = nullptr,
                                                    int nseg = 0, long seg_base = 0) {
  __shared__ long s_end[SCALED ? ADAMW_MAX_SEG : 1];
  __shared__ float s_scale[SCALED ? ADAMW_MAX_SEG : 1];
  if constexpr (SCALED) {
    for (int i = threadIdx.x; i < nseg; i += 256) { s_end[i] = seg_end[i]; s_scale[i] = seg_scale[i]; }
    __syncthreads();
  }
  auto lr_of = [&](long group) -> float {                 // lr of the 4-element group `group` of this launch's slice
    if constexpr (!SCALED) return lr;
    const long e = seg_base + group * 4;
    int lo = 0, hi = nseg - 1;                             // first segment whose end lies beyond e (past the table: the last scale)
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_end[mid] <= e) lo = mid + 1; else hi = mid; }
    return lr * s_scale[lo];
  };
  const float gs = clip_coef ? grad_scale * clip_coef[0] : grad_scale;
  const long nv = n >> 2;
  const long stride = (long)gridDim.x * 256;
  auto ld4 = [](const f32x4* p) { if constexpr (NT) return __builtin_nontemporal_load(p); else return *p; };
  auto st4 = [](f32x4* p, f32x4 x) { if constexpr (NT) __builtin_nontemporal_store(x, p); else *p = x; };
  for (long i0 = (long)blockIdx.x * 256 + threadIdx.x; i0 < nv; i0 += stride * U) {
    f32x4 p4[U], m4[U], v4[U];
    float g[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = i0 + u * stride;
      if (i < nv) {
        p4[u] = ld4(reinterpret_cast<const f32x4*>(master) + i);
        m4[u] = ld4(reinterpret_cast<const f32x4*>(m) + i);
        v4[u] = ld4(reinterpret_cast<const f32x4*>(v) + i);
        if constexpr (GRAD_BF16) {
          u32x2 gg;
          if constexpr (NT) gg = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(grad) + i); else gg = reinterpret_cast<const u32x2*>(grad)[i];
          g[u][0] = __uint_as_float(gg[0] << 16); g[u][1] = __uint_as_float(gg[0] & 0xffff0000u);
          g[u][2] = __uint_as_float(gg[1] << 16); g[u][3] = __uint_as_float(gg[1] & 0xffff0000u);
        } else {
          const f32x4 gg = ld4(reinterpret_cast<const f32x4*>(grad) + i);
          g[u][0] = gg[0]; g[u][1] = gg[1]; g[u][2] = gg[2]; g[u][3] = gg[3];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = i0 + u * stride;
      if (i < nv) {
        f32x4 po, mo, vo;
        const float lr = lr_of(i);                         // shadows the launch-wide lr (identical value when !SCALED)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float ge = g[u][e] * gs;
          const float mm = b1 * m4[u][e] + (1.f - b1) * ge;
          const float vv = b2 * v4[u][e] + (1.f - b2) * ge * ge;
          const float upd = (mm / bc1) / (sqrtf(vv / bc2) + eps) + wd * p4[u][e];
          po[e] = p4[u][e] - lr * upd;
          mo[e] = mm; vo[e] = vv;
        }
        st4(reinterpret_cast<f32x4*>(master) + i, po);
        st4(reinterpret_cast<f32x4*>(m) + i, mo);
        st4(reinterpret_cast<f32x4*>(v) + i, vo);
        if (shadow) {
          const u32x2 pk = pack4(po[0], po[1], po[2], po[3]);
          if constexpr (NT) __builtin_nontemporal_store(pk, reinterpret_cast<u32x2*>(shadow) + i); else reinterpret_cast<u32x2*>(shadow)[i] = pk;
        }
      }
    }
  }
}

template <bool BF16>
__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const void* __restrict__ g, long n, float* __restrict__ partial) {
  __shared__ float red[256];
  float s = 0.f;
  const long nv = n >> 2;
  const long stride = (long)gridDim.x * 256;
  // four independent 8 / 16-byte loads per thread and trip: one load at a time left this 2 GB read at 3.8 TB/s
  for (long i0 = (long)blockIdx.x * 256 + threadIdx.x; i0 < nv; i0 += 4 * stride) {
    if constexpr (BF16) {
      u32x2 gg[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const long i = i0 + u * stride; gg[u] = i < nv ? reinterpret_cast<const u32x2*>(g)[i] : u32x2{0u, 0u}; }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float a = __uint_as_float(gg[u][0] << 16), b = __uint_as_float(gg[u][0] & 0xffff0000u);
        const float c = __uint_as_float(gg[u][1] << 16), d = __uint_as_float(gg[u][1] & 0xffff0000u);
        s += a * a + b * b + c * c + d * d;
      }
    } else {
      f32x4 gg[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const long i = i0 + u * stride; gg[u] = i < nv ? reinterpret_cast<const f32x4*>(g)[i] : f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int u = 0; u < 4; ++u) s += gg[u][0] * gg[u][0] + gg[u][1] * gg[u][1] + gg[u][2] * gg[u][2] + gg[u][3] * gg[u][3];
    }
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void sqnorm_final_kernel(const float* __restrict__ partial, int np, float* __restrict__ out, int accumulate) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < np; i += 256) s += partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = accumulate ? out[0] + red[0] : red[0];
}

// coef = min(1, max_norm / (sqrt(sumsq) + 1e-6)) ; norm_out = sqrt(sumsq)      (torch.nn.utils.clip_grad_norm_ formula)
__global__ void clip_coef_kernel(const float* sumsq, float max_norm, float* coef, float* norm_out) {
  const float nrm = sqrtf(sumsq[0]);
  if (norm_out) norm_out[0] = nrm;
  const float c = max_norm / (nrm + 1e-6f);
  coef[0] = (max_norm > 0.f && c < 1.f) ? c : 1.f;
}

// out[i] = sum_r float(in[r * chunk + i]), r ascending: the fp32 accumulation of the W bf16 gradient chunks one rank receives from an
// all-to-all (the reduce half of a reduce-scatter whose wire format is bf16 but whose sum is exact in fp32).  W = 1: bf16 -> fp32.
__global__ __launch_bounds__(256) void shard_sum_kernel(const bf16_t* __restrict__ in, int W, long chunk, float* __restrict__ out) {
  const long nv = chunk >> 3;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < W; ++r) {
      float f[8];
      unpack8(reinterpret_cast<const u32x4*>(in + (long)r * chunk)[i], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += f[e];
    }
    reinterpret_cast<f32x4*>(out)[2 * i] = f32x4{acc[0], acc[1], acc[2], acc[3]};
    reinterpret_cast<f32x4*>(out)[2 * i + 1] = f32x4{acc[4], acc[5], acc[6], acc[7]};
  }
}

}  // namespace ivh

using namespace ivh;
constexpr int SQNORM_BLOCKS = 1024;

extern "C" int ivh_shard_sum_bf16(const uint16_t* in, int W, int64_t chunk, float* out, void* stream) {
  IVH_REQUIRE(in && out && W >= 1 && chunk > 0 && chunk % 8 == 0, "shard_sum_bf16: bad args (chunk must be a multiple of 8)");
  IVH_REQUIRE(((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0, "shard_sum_bf16: buffers must be 16-byte aligned");
  long blocks = (chunk / 8 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(shard_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, W, (long)chunk, out);
  return ivh_host::check_launch("shard_sum_bf16");
}

static int adamw_launch(float* master, float* exp_avg, float* exp_avg_sq, const void* grad, int grad_bf16, uint16_t* shadow_bf16, int64_t n,
                        float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, const float* clip_coef,
                        const int64_t* seg_end, const float* seg_scale, int nseg, int64_t seg_base, void* stream);

extern "C" int ivh_adamw_step(float* master, float* exp_avg, float* exp_avg_sq, const void* grad, int grad_bf16,
                              uint16_t* shadow_bf16, int64_t n, float lr, float beta1, float beta2, float eps,
                              float weight_decay, int step, float grad_scale, const float* clip_coef, void* stream) {
  return adamw_launch(master, exp_avg, exp_avg_sq, grad, grad_bf16, shadow_bf16, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, clip_coef,
                      nullptr, nullptr, 0, 0, stream);
}

extern "C" int ivh_adamw_step_scaled(float* master, float* exp_avg, float* exp_avg_sq, const void* grad, int grad_bf16,
                                     uint16_t* shadow_bf16, int64_t n, float lr, float beta1, float beta2, float eps,
                                     float weight_decay, int step, float grad_scale, const float* clip_coef,
                                     const int64_t* seg_end, const float* seg_scale, int nseg, int64_t seg_base, void* stream) {
  IVH_REQUIRE(seg_end && seg_scale && nseg >= 1 && nseg <= ADAMW_MAX_SEG && seg_base >= 0 && seg_base % 4 == 0,
              "adamw_step_scaled: needs 1..1024 segments (device arrays) and a slice offset that is a multiple of 4");
  return adamw_launch(master, exp_avg, exp_avg_sq, grad, grad_bf16, shadow_bf16, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, clip_coef,
                      seg_end, seg_scale, nseg, seg_base, stream);
}

static int adamw_launch(float* master, float* exp_avg, float* exp_avg_sq, const void* grad, int grad_bf16,
                        uint16_t* shadow_bf16, int64_t n, float lr, float beta1, float beta2, float eps,
                        float weight_decay, int step, float grad_scale, const float* clip_coef,
                        const int64_t* seg_end, const float* seg_scale, int nseg, int64_t seg_base, void* stream) {
  IVH_REQUIRE(master && exp_avg && exp_avg_sq && grad && n > 0 && n % 4 == 0, "adamw_step: bad args (n must be a multiple of 4)");
  IVH_REQUIRE(step >= 1, "adamw_step: step counts from 1");
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  // measurement switches (tools/bench_adamw.py): IVH_ADAMW_VARIANT bit 0 = two groups per thread and trip, bit 1 = non-temporal accesses;
  // IVH_ADAMW_BLOCKS = grid cap.  Defaults = the best of profiles/r3_adamw_variants_v1.jsonl (1.07e9 parameters: 4.83 ms = 6.2 TB/s;
  // one group, ordinary accesses, 8192 workgroups: 5.50 ms = 5.4 TB/s).  Every variant produces the same bits.
  static const int variant = [] { const char* e = getenv("IVH_ADAMW_VARIANT"); return e ? atoi(e) : 3; }();
  static const long cap = [] { const char* e = getenv("IVH_ADAMW_BLOCKS"); const long c = e ? atol(e) : 32768; return c > 0 ? c : 32768; }();
  long blocks = (n / 4 + 255) / 256;
  if (blocks > cap) blocks = cap;
  hipStream_t s = (hipStream_t)stream;
  if (nseg > 0) {                                          // layer-wise lr decay: the shipped access pattern (two groups, non-temporal) + the table
    const long* se = reinterpret_cast<const long*>(seg_end);
    if (grad_bf16) hipLaunchKernelGGL((adamw_kernel<true, 2, true, true>), dim3((unsigned)blocks), dim3(256), 0, s, master, exp_avg, exp_avg_sq, grad,
        shadow_bf16, (long)n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale, clip_coef, se, seg_scale, nseg, (long)seg_base);
    else hipLaunchKernelGGL((adamw_kernel<false, 2, true, true>), dim3((unsigned)blocks), dim3(256), 0, s, master, exp_avg, exp_avg_sq, grad,
        shadow_bf16, (long)n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale, clip_coef, se, seg_scale, nseg, (long)seg_base);
    return ivh_host::check_launch("adamw_step_scaled");
  }
#define IVH_ADAMW(G, U, NT) hipLaunchKernelGGL((adamw_kernel<G, U, NT>), dim3((unsigned)blocks), dim3(256), 0, s, master, exp_avg, exp_avg_sq, grad, \
    shadow_bf16, (long)n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale, clip_coef, nullptr, nullptr, 0, 0L)
  if (grad_bf16) {
    switch (variant & 3) { case 1: IVH_ADAMW(true, 2, false); break; case 2: IVH_ADAMW(true, 1, true); break; case 3: IVH_ADAMW(true, 2, true); break;
                           default: IVH_ADAMW(true, 1, false); }
  } else {
    switch (variant & 3) { case 1: IVH_ADAMW(false, 2, false); break; case 2: IVH_ADAMW(false, 1, true); break; case 3: IVH_ADAMW(false, 2, true); break;
                           default: IVH_ADAMW(false, 1, false); }
  }
#undef IVH_ADAMW
  return ivh_host::check_launch("adamw_step");
}

extern "C" int ivh_sqnorm_scratch_floats(void) { return SQNORM_BLOCKS; }
extern "C" int ivh_sqnorm(const void* g, int g_bf16, int64_t n, float* partial, float* out, int accumulate, void* stream) {
  IVH_REQUIRE(g && partial && out && n > 0 && n % 4 == 0, "sqnorm: bad args");
  hipStream_t s = (hipStream_t)stream;
  if (g_bf16) hipLaunchKernelGGL((sqnorm_partial_kernel<true>), dim3(SQNORM_BLOCKS), dim3(256), 0, s, g, (long)n, partial);
  else hipLaunchKernelGGL((sqnorm_partial_kernel<false>), dim3(SQNORM_BLOCKS), dim3(256), 0, s, g, (long)n, partial);
  hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(256), 0, s, partial, SQNORM_BLOCKS, out, accumulate);
  return ivh_host::check_launch("sqnorm");
}
extern "C" int ivh_clip_coef(const float* sumsq, float max_norm, float* coef, float* norm_out, void* stream) {
  IVH_REQUIRE(sumsq && coef, "clip_coef: bad args");
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sumsq, max_norm, coef, norm_out);
  return ivh_host::check_launch("clip_coef");
}
