// Teacher-side output kernels: the per-frame CLIP teacher
// (InternVideo2/single_modality/models/internvl_clip_vision.py, "T:") runs the student's block kernels on B*T frame sequences;
// what is specific to it is the tail: frame merge + l2 normalisation of the tapped features (T:445-458) and the head-averaged
// attention map of the 1-query pooling attention that drives attention-guided masking (T:69-85,443,462-463).
// Both are HBM-bound row kernels.
#include "common.h"
#include "../../include/internvideo_hip.h"

namespace ivh {

// x [B][T][L][C] (fp32 | bf16) -> out [B][1 + T*(L-1)][C] (bf16 | fp32):
//   out[b, 0]               = mean_t x[b, t, 0]           (the per-frame cls tokens are averaged, T:449-450)
//   out[b, 1 + t*(L-1) + j] = x[b, t, 1 + j]
// each output row divided by its l2 norm when `l2` (no epsilon, T:453 / T:456).  L = 1: out[b, 0] = mean over frames (T:455).
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void frames_merge_l2_kernel(const TI* __restrict__ x, int T, int L, int C, int l2,
                                                              TO* __restrict__ out) {
  __shared__ float red[4];
  const int Lo = 1 + T * (L - 1);
  const long row = blockIdx.x;                 // b * Lo + r
  const int b = row / Lo, r = row % Lo;
  const int tid = threadIdx.x;
  constexpr int MAXE = 16;                     // C <= 256 * 16 = 4096
  float v[MAXE];
  float ss = 0.f;
#pragma unroll
  for (int n = 0; n < MAXE; ++n) {
    const int c = tid + n * 256;
    float a = 0.f;
    if (c < C) {
      if (r == 0) {
        for (int t = 0; t < T; ++t) {
          const long so = (((long)b * T + t) * L) * C + c;
          if constexpr (sizeof(TI) == 4) a += x[so];
          else a += bf2f(x[so]);
        }
        a *= 1.0f / (float)T;
      } else {
        const int t = (r - 1) / (L - 1), j = (r - 1) % (L - 1);
        const long so = (((long)b * T + t) * L + 1 + j) * C + c;
        if constexpr (sizeof(TI) == 4) a = x[so];
        else a = bf2f(x[so]);
      }
    }
    v[n] = a;
    ss += a * a;
  }
  float inv = 1.0f;
  if (l2) {
    ss = wave_sum(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    inv = 1.0f / sqrtf(red[0] + red[1] + red[2] + red[3]);
  }
#pragma unroll
  for (int n = 0; n < MAXE; ++n) {
    const int c = tid + n * 256;
    if (c < C) {
      const float o = v[n] * inv;
      if constexpr (sizeof(TO) == 4) out[row * C + c] = o;
      else out[row * C + c] = f2bf(o);
    }
  }
}

// Head-averaged probabilities of a 1-query attention: q [S][H][hd], k [S][L][H][hd] (bf16, k rows strided by ks_l elements,
// sequences by ks_s) -> out[s][l - skip] = (1/H) sum_h softmax_l(scale * <q[s,h], k[s,l,h]>)  for l >= skip   (fp32)
// (`attn.mean(1)` of T:82-83 then `attn[:, 0, 1:]` T:463 with skip = 1).  One workgroup per sequence; L <= 1024.
__global__ __launch_bounds__(256) void pool_attn_map_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                            long ks_s, long ks_l, int L, int H, int hd, float scale, int skip,
                                                            float* __restrict__ out) {
  __shared__ float red[4];
  __shared__ float qs[256];                    // one head of q (hd <= 256)
  const int s = blockIdx.x, tid = threadIdx.x;
  constexpr int MAXL = 4;                      // L <= 1024
  float acc[MAXL] = {0.f, 0.f, 0.f, 0.f};
  for (int h = 0; h < H; ++h) {
    __syncthreads();
    if (tid < hd) qs[tid] = bf2f(q[((long)s * H + h) * hd + tid]);
    __syncthreads();
    float lg[MAXL];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < MAXL; ++i) {
      const int l = tid + i * 256;
      lg[i] = -INFINITY;
      if (l < L) {
        const bf16_t* kr = k + (long)s * ks_s + (long)l * ks_l + (long)h * hd;
        float d = 0.f;
        for (int e = 0; e < hd; e += 8) {
          float kv[8];
          unpack8(*reinterpret_cast<const u32x4*>(kr + e), kv);
#pragma unroll
          for (int u = 0; u < 8; ++u) d += kv[u] * qs[e + u];
        }
        lg[i] = d * scale;
        mx = fmaxf(mx, lg[i]);
      }
    }
    mx = wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXL; ++i) {
      lg[i] = (tid + i * 256 < L) ? __expf(lg[i] - mx) : 0.f;
      sum += lg[i];
    }
    sum = wave_sum(sum);
    if ((tid & 63) == 0) red[tid >> 6] = sum;
    __syncthreads();
    const float inv = 1.0f / ((red[0] + red[1] + red[2] + red[3]) * (float)H);
#pragma unroll
    for (int i = 0; i < MAXL; ++i) acc[i] += lg[i] * inv;
  }
#pragma unroll
  for (int i = 0; i < MAXL; ++i) {
    const int l = tid + i * 256;
    if (l < L && l >= skip) out[(long)s * (L - skip) + l - skip] = acc[i];
  }
}

}  // namespace ivh

using namespace ivh;

extern "C" int ivh_frames_merge_l2(const void* x, int x_fp32, int B, int T, int L, int C, int l2, void* out, int out_fp32, void* stream) {
  IVH_REQUIRE(x && out && B > 0 && T > 0 && L > 0 && C > 0 && C <= 4096, "frames_merge_l2: bad args (C <= 4096)");
  const long rows = (long)B * (1 + (long)T * (L - 1));
  IVH_REQUIRE(rows < (1L << 31), "frames_merge_l2: too many rows");
  dim3 grid((unsigned)rows);
  hipStream_t s = (hipStream_t)stream;
  if (x_fp32 && !out_fp32) hipLaunchKernelGGL((frames_merge_l2_kernel<float, bf16_t>), grid, dim3(256), 0, s, (const float*)x, T, L, C, l2, (bf16_t*)out);
  else if (x_fp32 && out_fp32) hipLaunchKernelGGL((frames_merge_l2_kernel<float, float>), grid, dim3(256), 0, s, (const float*)x, T, L, C, l2, (float*)out);
  else if (!x_fp32 && !out_fp32) hipLaunchKernelGGL((frames_merge_l2_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)x, T, L, C, l2, (bf16_t*)out);
  else hipLaunchKernelGGL((frames_merge_l2_kernel<bf16_t, float>), grid, dim3(256), 0, s, (const bf16_t*)x, T, L, C, l2, (float*)out);
  return ivh_host::check_launch("frames_merge_l2");
}

extern "C" int ivh_pool_attn_map(const uint16_t* q, const uint16_t* k, int64_t ks_s, int64_t ks_l, int S, int L, int H, int hd,
                                 float scale, int skip, float* out, void* stream) {
  IVH_REQUIRE(q && k && out && S > 0 && H > 0, "pool_attn_map: bad args");
  IVH_REQUIRE(L > 0 && L <= 1024 && hd > 0 && hd <= 256 && hd % 8 == 0, "pool_attn_map: L=%d (<= 1024), hd=%d (multiple of 8, <= 256)", L, hd);
  IVH_REQUIRE(skip >= 0 && skip < L && ks_l % 8 == 0 && ks_s % 8 == 0 && ((uintptr_t)k & 15) == 0, "pool_attn_map: k must be 16-byte aligned with strides multiple of 8");
  hipLaunchKernelGGL(pool_attn_map_kernel, dim3(S), dim3(256), 0, (hipStream_t)stream, q, k, (long)ks_s, (long)ks_l, L, H, hd, scale, skip, out);
  return ivh_host::check_launch("pool_attn_map");
}

// ------------------------------------------------------------------------------------------------------------------------------
// 1-query multi-head attention for head dims the MFMA flash kernel does not tile (hd > 128): the attention-pool projector of the
// 6B models is 16 heads over D = 3200 -> hd = 200 (internvideo2_pretrain.py:18-114 with P:758-766; internvl_clip_vision.py:23-86).
// One query per sequence makes this a bandwidth problem (K and V are read once, 2 * L * hd MACs per head), so plain VALU code:
// one workgroup per (sequence, head), logits / probabilities staged in LDS.
// ------------------------------------------------------------------------------------------------------------------------------
namespace ivh {

constexpr int POOL_MAXL = 8192;                 // keys per sequence (LDS: 2 * 4 * MAXL = 64 KiB in the backward)

__device__ __forceinline__ float block_sum256(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max256(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// o[s, h, :] = sum_l softmax_l(scale <q[s,h], k[s,l,h]>) v[s,l,h,:] ;  lse[s,h] = log sum_l exp(logit)
__global__ __launch_bounds__(256) void pool_attn_fwd_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                            const bf16_t* __restrict__ v, long ks_s, long ks_l, int L, int H, int hd,
                                                            float scale, bf16_t* __restrict__ o, float* __restrict__ lse) {
  extern __shared__ float sm[];                 // p[L]
  __shared__ float red[4];
  __shared__ float qs[256];
  const int s = blockIdx.x / H, h = blockIdx.x % H, tid = threadIdx.x;
  if (tid < hd) qs[tid] = bf2f(q[((long)s * H + h) * hd + tid]);
  __syncthreads();
  float mx = -INFINITY;
  for (int l = tid; l < L; l += 256) {
    const bf16_t* kr = k + (long)s * ks_s + (long)l * ks_l + (long)h * hd;
    float d = 0.f;
    for (int e = 0; e < hd; e += 8) {
      float kv[8];
      unpack8(*reinterpret_cast<const u32x4*>(kr + e), kv);
#pragma unroll
      for (int u = 0; u < 8; ++u) d += kv[u] * qs[e + u];
    }
    d *= scale;
    sm[l] = d;
    mx = fmaxf(mx, d);
  }
  mx = block_max256(mx, red);
  float sum = 0.f;
  for (int l = tid; l < L; l += 256) {
    const float p = __expf(sm[l] - mx);
    sm[l] = p;
    sum += p;
  }
  sum = block_sum256(sum, red);                 // (its barriers also publish sm[])
  const float inv = 1.0f / sum;
  if (tid == 0 && lse) lse[(long)s * H + h] = mx + __logf(sum);
  for (int d = tid; d < hd; d += 256) {
    const bf16_t* vc = v + (long)s * ks_s + (long)h * hd + d;
    float acc = 0.f;
    for (int l = 0; l < L; ++l) acc += sm[l] * bf2f(vc[(long)l * ks_l]);
    o[((long)s * H + h) * hd + d] = f2bf(acc * inv);
  }
}

// backward of the above.  dq [S][H*hd], dk / dv [S][L][H*hd] contiguous (bf16)
__global__ __launch_bounds__(256) void pool_attn_bwd_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                            const bf16_t* __restrict__ v, long ks_s, long ks_l,
                                                            const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                            int L, int H, int hd, float scale, bf16_t* __restrict__ dq,
                                                            bf16_t* __restrict__ dk, bf16_t* __restrict__ dv) {
  extern __shared__ float sm[];                 // p[L], dS[L]
  __shared__ float red[4];
  __shared__ float qs[256], dos[256];
  float* ps = sm;
  float* ds = sm + L;
  const int s = blockIdx.x / H, h = blockIdx.x % H, tid = threadIdx.x;
  const int D = H * hd;
  if (tid < hd) {
    qs[tid] = bf2f(q[((long)s * H + h) * hd + tid]);
    dos[tid] = bf2f(dout[((long)s * H + h) * hd + tid]);
  }
  __syncthreads();
  const float ls = lse[(long)s * H + h];
  float part = 0.f;
  for (int l = tid; l < L; l += 256) {
    const bf16_t* kr = k + (long)s * ks_s + (long)l * ks_l + (long)h * hd;
    const bf16_t* vr = v + (long)s * ks_s + (long)l * ks_l + (long)h * hd;
    float d = 0.f, dp = 0.f;
    for (int e = 0; e < hd; e += 8) {
      float kv[8], vv[8];
      unpack8(*reinterpret_cast<const u32x4*>(kr + e), kv);
      unpack8(*reinterpret_cast<const u32x4*>(vr + e), vv);
#pragma unroll
      for (int u = 0; u < 8; ++u) { d += kv[u] * qs[e + u]; dp += vv[u] * dos[e + u]; }
    }
    const float p = __expf(d * scale - ls);
    ps[l] = p;
    ds[l] = dp;
    part += p * dp;
  }
  const float delta = block_sum256(part, red);
  for (int l = tid; l < L; l += 256) {
    const float p = ps[l];
    const float g = p * (ds[l] - delta);        // dS[l]
    ds[l] = g;
    bf16_t* dkr = dk + ((long)s * L + l) * D + (long)h * hd;
    bf16_t* dvr = dv + ((long)s * L + l) * D + (long)h * hd;
    const float gs = g * scale;
    for (int e = 0; e < hd; e += 8) {
      float a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { a[u] = gs * qs[e + u]; b[u] = p * dos[e + u]; }
      *reinterpret_cast<u32x4*>(dkr + e) = pack8(a);
      *reinterpret_cast<u32x4*>(dvr + e) = pack8(b);
    }
  }
  __syncthreads();
  for (int d = tid; d < hd; d += 256) {
    const bf16_t* kc = k + (long)s * ks_s + (long)h * hd + d;
    float acc = 0.f;
    for (int l = 0; l < L; ++l) acc += ds[l] * bf2f(kc[(long)l * ks_l]);
    dq[((long)s * H + h) * hd + d] = f2bf(acc * scale);
  }
}

}  // namespace ivh

extern "C" int ivh_pool_attn_fwd(const uint16_t* q, const uint16_t* k, const uint16_t* v, int64_t ks_s, int64_t ks_l, int S, int L, int H,
                                 int hd, float scale, uint16_t* o, float* lse, void* stream) {
  IVH_REQUIRE(q && k && v && o && S > 0 && H > 0 && L > 0, "pool_attn_fwd: bad args");
  IVH_REQUIRE(L <= POOL_MAXL && hd > 0 && hd <= 256 && hd % 8 == 0, "pool_attn_fwd: L=%d (<= %d), hd=%d (multiple of 8, <= 256)", L, POOL_MAXL, hd);
  IVH_REQUIRE(ks_l % 8 == 0 && ks_s % 8 == 0 && ((uintptr_t)k & 15) == 0 && ((uintptr_t)v & 15) == 0, "pool_attn_fwd: k / v must be 16-byte aligned with strides multiple of 8");
  hipLaunchKernelGGL(pool_attn_fwd_kernel, dim3(S * H), dim3(256), (size_t)L * 4, (hipStream_t)stream, q, k, v, (long)ks_s, (long)ks_l, L, H, hd, scale, o, lse);
  return ivh_host::check_launch("pool_attn_fwd");
}

extern "C" int ivh_pool_attn_bwd(const uint16_t* q, const uint16_t* k, const uint16_t* v, int64_t ks_s, int64_t ks_l, const uint16_t* dout,
                                 const float* lse, int S, int L, int H, int hd, float scale, uint16_t* dq, uint16_t* dk, uint16_t* dv, void* stream) {
  IVH_REQUIRE(q && k && v && dout && lse && dq && dk && dv && S > 0 && H > 0 && L > 0, "pool_attn_bwd: bad args");
  IVH_REQUIRE(L <= POOL_MAXL && hd > 0 && hd <= 256 && hd % 8 == 0, "pool_attn_bwd: L=%d (<= %d), hd=%d (multiple of 8, <= 256)", L, POOL_MAXL, hd);
  IVH_REQUIRE(ks_l % 8 == 0 && ks_s % 8 == 0 && ((uintptr_t)k & 15) == 0 && ((uintptr_t)v & 15) == 0, "pool_attn_bwd: k / v must be 16-byte aligned with strides multiple of 8");
  static bool attr_set = false;
  if (!attr_set) { hipFuncSetAttribute((const void*)pool_attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, POOL_MAXL * 8); attr_set = true; }
  hipLaunchKernelGGL(pool_attn_bwd_kernel, dim3(S * H), dim3(256), (size_t)L * 8, (hipStream_t)stream, q, k, v, (long)ks_s, (long)ks_l, dout, lse, L, H, hd, scale, dq, dk, dv);
  return ivh_host::check_launch("pool_attn_bwd");
}
