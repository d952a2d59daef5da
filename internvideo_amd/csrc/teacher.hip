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
