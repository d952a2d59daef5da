// Fused AdamW on flat HBM-resident buffers + global grad-norm pieces.
// Reference: DeepSpeed FusedAdam (adam_w_mode) configured by InternVideo2/single_modality/utils.py:821-871
// (betas (0.9, 0.98), eps 1e-6, bias correction, decoupled weight decay, gradient clipping 3.0 at :860-861).
// One launch per parameter group region; 16 B/param read (master, m, v) + 2..4 B grad, 14 B/param written
// (master, m, v, bf16 compute copy): purely HBM-bound.
#include "common.h"
#include "../../include/internvideo_hip.h"

namespace ivh {

template <bool GRAD_BF16>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ master, float* __restrict__ m, float* __restrict__ v,
                                                    const void* __restrict__ grad, bf16_t* __restrict__ shadow, long n,
                                                    float lr, float b1, float b2, float eps, float wd, float bc1, float bc2,
                                                    float grad_scale, const float* __restrict__ clip_coef) {
  const float gs = clip_coef ? grad_scale * clip_coef[0] : grad_scale;
  const long nv = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
    const f32x4 p4 = reinterpret_cast<const f32x4*>(master)[i];
    const f32x4 m4 = reinterpret_cast<const f32x4*>(m)[i];
    const f32x4 v4 = reinterpret_cast<const f32x4*>(v)[i];
    float g[4];
    if constexpr (GRAD_BF16) {
      const u32x2 gg = reinterpret_cast<const u32x2*>(grad)[i];
      g[0] = __uint_as_float(gg[0] << 16); g[1] = __uint_as_float(gg[0] & 0xffff0000u);
      g[2] = __uint_as_float(gg[1] << 16); g[3] = __uint_as_float(gg[1] & 0xffff0000u);
    } else {
      const f32x4 gg = reinterpret_cast<const f32x4*>(grad)[i];
      g[0] = gg[0]; g[1] = gg[1]; g[2] = gg[2]; g[3] = gg[3];
    }
    f32x4 po, mo, vo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float ge = g[e] * gs;
      const float mm = b1 * m4[e] + (1.f - b1) * ge;
      const float vv = b2 * v4[e] + (1.f - b2) * ge * ge;
      const float upd = (mm / bc1) / (sqrtf(vv / bc2) + eps) + wd * p4[e];
      po[e] = p4[e] - lr * upd;
      mo[e] = mm; vo[e] = vv;
    }
    reinterpret_cast<f32x4*>(master)[i] = po;
    reinterpret_cast<f32x4*>(m)[i] = mo;
    reinterpret_cast<f32x4*>(v)[i] = vo;
    if (shadow) reinterpret_cast<u32x2*>(shadow)[i] = pack4(po[0], po[1], po[2], po[3]);
  }
}

template <bool BF16>
__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const void* __restrict__ g, long n, float* __restrict__ partial) {
  __shared__ float red[256];
  float s = 0.f;
  const long nv = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
    if constexpr (BF16) {
      const u32x2 gg = reinterpret_cast<const u32x2*>(g)[i];
      const float a = __uint_as_float(gg[0] << 16), b = __uint_as_float(gg[0] & 0xffff0000u);
      const float c = __uint_as_float(gg[1] << 16), d = __uint_as_float(gg[1] & 0xffff0000u);
      s += a * a + b * b + c * c + d * d;
    } else {
      const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i];
      s += gg[0] * gg[0] + gg[1] * gg[1] + gg[2] * gg[2] + gg[3] * gg[3];
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

extern "C" int ivh_adamw_step(float* master, float* exp_avg, float* exp_avg_sq, const void* grad, int grad_bf16,
                              uint16_t* shadow_bf16, int64_t n, float lr, float beta1, float beta2, float eps,
                              float weight_decay, int step, float grad_scale, const float* clip_coef, void* stream) {
  IVH_REQUIRE(master && exp_avg && exp_avg_sq && grad && n > 0 && n % 4 == 0, "adamw_step: bad args (n must be a multiple of 4)");
  IVH_REQUIRE(step >= 1, "adamw_step: step counts from 1");
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipStream_t s = (hipStream_t)stream;
  if (grad_bf16)
    hipLaunchKernelGGL((adamw_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, s, master, exp_avg, exp_avg_sq, grad, shadow_bf16, (long)n,
                       lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale, clip_coef);
  else
    hipLaunchKernelGGL((adamw_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, s, master, exp_avg, exp_avg_sq, grad, shadow_bf16, (long)n,
                       lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale, clip_coef);
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
