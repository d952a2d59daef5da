// VideoMAE pixel-reconstruction path (reference: InternVideo1/Pretrain/VideoMAE/modeling_pretrain.py "MP:",
// engine_for_pretraining.py "ME:"): token edges of a cls-free masked auto-encoder -- visible-token assembly, the mask-token
// scatter that builds the decoder input, row windows of the token stream, the normalised-pixel regression target and the MSE.
// All HBM-bound; index math is integer and bit-exact.
#include "common.h"
#include "../../include/internvideo_hip.h"

namespace ivh {

// x0[b, j] = tok[b, j] + pos[vis_idx[b, j + 1] - 1]       (fp32; vis_idx carries a leading pseudo-cls 0 so that the im2col and
// index kernels of the cls-carrying student are shared; MP:127-133 `x + pos_embed` then `x[~mask]`)
__global__ __launch_bounds__(256) void assemble_tokens_nocls_kernel(const bf16_t* __restrict__ tok, const float* __restrict__ pos,
                                                                    const int32_t* __restrict__ vis_idx, int B, int L, int D,
                                                                    float* __restrict__ x0) {
  const int nch = D >> 3, Lo = L - 1;
  const long id = (long)blockIdx.x * 256 + threadIdx.x;
  if (id >= (long)B * Lo * nch) return;
  const int c = id % nch;
  const long bj = id / nch;
  const int j = bj % Lo, b = bj / Lo;
  const int n = vis_idx[(long)b * L + j + 1] - 1;
  float v[8];
  unpack8(*reinterpret_cast<const u32x4*>(tok + bj * D + c * 8), v);
  const float* pr = pos + (long)n * D + c * 8;
  float* o = x0 + bj * D + c * 8;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = v[e] + pr[e];
}

// decoder input (MP:381-389): out[b, j] = xvis[b, j] + pos[vis[b, j + 1] - 1]            j <  Nvis
//                             out[b, j] = mask_token + pos[msk[b, j - Nvis] - 1]          j >= Nvis        (fp32 stream rows)
__global__ __launch_bounds__(256) void mae_decoder_input_kernel(const bf16_t* __restrict__ xvis, const float* __restrict__ mask_token,
                                                                const float* __restrict__ pos, const int32_t* __restrict__ vis_idx,
                                                                const int32_t* __restrict__ msk_idx, int B, int Nvis, int Nmask, int D,
                                                                float* __restrict__ out) {
  const int nch = D >> 3, N = Nvis + Nmask;
  const long id = (long)blockIdx.x * 256 + threadIdx.x;
  if (id >= (long)B * N * nch) return;
  const int c = id % nch;
  const long bj = id / nch;
  const int j = bj % N, b = bj / N;
  float v[8];
  int n;
  if (j < Nvis) {
    n = vis_idx[(long)b * (Nvis + 1) + j + 1] - 1;
    unpack8(*reinterpret_cast<const u32x4*>(xvis + ((long)b * Nvis + j) * D + c * 8), v);
  } else {
    n = msk_idx[(long)b * Nmask + j - Nvis] - 1;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = mask_token[c * 8 + e];
  }
  const float* pr = pos + (long)n * D + c * 8;
  float* o = out + bj * D + c * 8;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = v[e] + pr[e];
}

// dst[b, j] = bf16(src[b, start + j]), j < count         (fp32 [B][L][D] stream rows -> a bf16 row window)
__global__ __launch_bounds__(256) void rows_window_kernel(const float* __restrict__ src, int B, int L, int D, int start, int count,
                                                          bf16_t* __restrict__ dst) {
  const int nch = D >> 3;
  const long id = (long)blockIdx.x * 256 + threadIdx.x;
  if (id >= (long)B * count * nch) return;
  const int c = id % nch;
  const long bj = id / nch;
  const int j = bj % count, b = bj / count;
  const float* s = src + ((long)b * L + start + j) * D + c * 8;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = s[e];
  *reinterpret_cast<u32x4*>(dst + bj * D + c * 8) = pack8(v);
}

// dst[b, start + j] = src[b, j] (j < count), every other row of dst zeroed        (the backward of rows_window, fp32 <- bf16|fp32)
template <typename T>
__global__ __launch_bounds__(256) void rows_window_bwd_kernel(const T* __restrict__ src, int B, int L, int D, int start, int count,
                                                              float* __restrict__ dst) {
  const int nch = D >> 3;
  const long id = (long)blockIdx.x * 256 + threadIdx.x;
  if (id >= (long)B * L * nch) return;
  const int c = id % nch;
  const long bj = id / nch;
  const int j = bj % L, b = bj / L;
  float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (j >= start && j < start + count) {
    const long so = ((long)b * count + j - start) * D + c * 8;
    if constexpr (sizeof(T) == 4) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = src[so + e];
    } else {
      unpack8(*reinterpret_cast<const u32x4*>(src + so), v);
    }
  }
  float* d = dst + bj * D + c * 8;
#pragma unroll
  for (int e = 0; e < 8; ++e) d[e] = v[e];
}

// Regression target of one masked tubelet (ME:66-98): un-normalise (x * std + mean), take the (tub, p, p) cube of token
// msk_idx[b, j] - 1, per channel subtract the cube mean and divide by (unbiased std + 1e-6) when `normalize`, write in
// (p0 p1 p2 c) order.  One workgroup per (b, j); fp32 [B][Nmask][tub*p*p*3].
template <typename T>
__global__ __launch_bounds__(256) void pixel_target_kernel(const T* __restrict__ video, const int32_t* __restrict__ msk_idx,
                                                           int C, int Tn, int Hn, int Wn, int tub, int p, int Nmask, int normalize,
                                                           float m0, float m1, float m2, float s0, float s1, float s2,
                                                           float* __restrict__ out) {
  __shared__ float red[4][2];
  __shared__ float stat[3][2];
  const long row = blockIdx.x;                // b * Nmask + j
  const int b = row / Nmask;
  const int tok = msk_idx[row] - 1;
  const int gw = Wn / p, gh = Hn / p;
  const int pw = tok % gw, ph = (tok / gw) % gh, t = tok / (gw * gh);
  const int P = tub * p * p;                  // pixels per channel in the cube
  const int tid = threadIdx.x;
  float* o = out + row * (long)P * C;
  for (int c = 0; c < C; ++c) {
    const float mean_c = c == 0 ? m0 : (c == 1 ? m1 : m2), std_c = c == 0 ? s0 : (c == 1 ? s1 : s2);
    float sum = 0.f, sq = 0.f;
    for (int i = tid; i < P; i += 256) {
      const int dx = i % p, dy = (i / p) % p, dt = i / (p * p);
      const long src = (((long)b * C + c) * Tn + t * tub + dt) * Hn * Wn + (long)(ph * p + dy) * Wn + pw * p + dx;
      float v;
      if constexpr (sizeof(T) == 4) v = video[src];
      else v = bf2f(video[src]);
      v = v * std_c + mean_c;
      sum += v; sq += v * v;
    }
    if (normalize) {
      sum = wave_sum(sum); sq = wave_sum(sq);
      __syncthreads();
      if ((tid & 63) == 0) { red[tid >> 6][0] = sum; red[tid >> 6][1] = sq; }
      __syncthreads();
      if (tid == 0) {
        const float S = red[0][0] + red[1][0] + red[2][0] + red[3][0], Q = red[0][1] + red[1][1] + red[2][1] + red[3][1];
        const float mu = S / (float)P;
        const float var = fmaxf((Q - S * mu) / (float)(P - 1), 0.f);           // unbiased
        stat[c][0] = mu; stat[c][1] = 1.0f / (sqrtf(var) + 1e-6f);
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < P; i += 256) {
    const int dx = i % p, dy = (i / p) % p, dt = i / (p * p);
    for (int c = 0; c < C; ++c) {
      const float mean_c = c == 0 ? m0 : (c == 1 ? m1 : m2), std_c = c == 0 ? s0 : (c == 1 ? s1 : s2);
      const long src = (((long)b * C + c) * Tn + t * tub + dt) * Hn * Wn + (long)(ph * p + dy) * Wn + pw * p + dx;
      float v;
      if constexpr (sizeof(T) == 4) v = video[src];
      else v = bf2f(video[src]);
      v = v * std_c + mean_c;
      if (normalize) v = (v - stat[c][0]) * stat[c][1];
      o[(long)i * C + c] = v;
    }
  }
}

// rows[m] = sum_c (pred[m,c] - target[m,c])^2 ;  dpred[m,c] = bf16(dscale * 2 (pred - target))      (nn.MSELoss pieces, ME:101-106)
template <typename T>
__global__ __launch_bounds__(256) void mse_rows_kernel(const T* __restrict__ pred, const float* __restrict__ target, int Cc,
                                                       float dscale, float* __restrict__ rows, bf16_t* __restrict__ dpred) {
  __shared__ float red[4];
  const long m = blockIdx.x;
  const int tid = threadIdx.x;
  float acc = 0.f;
  for (int c = tid; c < Cc; c += 256) {
    float pv;
    if constexpr (sizeof(T) == 4) pv = pred[m * Cc + c];
    else pv = bf2f(pred[m * Cc + c]);
    const float d = pv - target[m * Cc + c];
    acc += d * d;
    if (dpred) dpred[m * Cc + c] = f2bf(2.0f * dscale * d);
  }
  acc = wave_sum(acc);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) rows[m] = red[0] + red[1] + red[2] + red[3];
}

// rows[m] = 2 - 2 <s[m], t[m]> ;  ds[m, c] = bf16(-2 dscale t[m, c])        (the alignment loss of l2-normalised features on
// MATERIALISED student outputs: multi_modality/models/criterions.py:480-482 new_UTA_Loss, engines/engine_for_pretraining.py:131-136)
template <typename TS, typename TT>
__global__ __launch_bounds__(256) void cosine_rows_kernel(const TS* __restrict__ sfeat, const TT* __restrict__ tfeat, int Cc, float dscale,
                                                          float* __restrict__ rows, bf16_t* __restrict__ ds) {
  __shared__ float red[4];
  const long m = blockIdx.x;
  const int tid = threadIdx.x;
  float acc = 0.f;
  for (int c = tid; c < Cc; c += 256) {
    float sv, tv;
    if constexpr (sizeof(TS) == 4) sv = sfeat[m * Cc + c]; else sv = bf2f(sfeat[m * Cc + c]);
    if constexpr (sizeof(TT) == 4) tv = tfeat[m * Cc + c]; else tv = bf2f(tfeat[m * Cc + c]);
    acc += sv * tv;
    if (ds) ds[m * Cc + c] = f2bf(-2.0f * dscale * tv);
  }
  acc = wave_sum(acc);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) rows[m] = 2.0f - 2.0f * (red[0] + red[1] + red[2] + red[3]);
}

}  // namespace ivh

using namespace ivh;
static inline dim3 grid1d(long n) { return dim3((unsigned)((n + 255) / 256)); }

extern "C" int ivh_assemble_tokens_nocls(const uint16_t* tok, const float* pos, const int32_t* vis_idx, int B, int L, int D,
                                         float* x0, void* stream) {
  IVH_REQUIRE(tok && pos && vis_idx && x0 && B > 0 && L > 1 && D % 8 == 0, "assemble_tokens_nocls: bad args");
  hipLaunchKernelGGL(assemble_tokens_nocls_kernel, grid1d((long)B * (L - 1) * (D / 8)), dim3(256), 0, (hipStream_t)stream, tok, pos, vis_idx, B, L, D, x0);
  return ivh_host::check_launch("assemble_tokens_nocls");
}

extern "C" int ivh_mae_decoder_input(const uint16_t* xvis, const float* mask_token, const float* pos, const int32_t* vis_idx,
                                     const int32_t* msk_idx, int B, int Nvis, int Nmask, int D, float* out, void* stream) {
  IVH_REQUIRE(xvis && mask_token && pos && vis_idx && msk_idx && out && B > 0 && Nvis > 0 && Nmask > 0 && D % 8 == 0, "mae_decoder_input: bad args");
  hipLaunchKernelGGL(mae_decoder_input_kernel, grid1d((long)B * (Nvis + Nmask) * (D / 8)), dim3(256), 0, (hipStream_t)stream,
                     xvis, mask_token, pos, vis_idx, msk_idx, B, Nvis, Nmask, D, out);
  return ivh_host::check_launch("mae_decoder_input");
}

extern "C" int ivh_rows_window(const float* src, int B, int L, int D, int start, int count, uint16_t* dst, void* stream) {
  IVH_REQUIRE(src && dst && B > 0 && D % 8 == 0 && start >= 0 && count > 0 && start + count <= L, "rows_window: bad args");
  hipLaunchKernelGGL(rows_window_kernel, grid1d((long)B * count * (D / 8)), dim3(256), 0, (hipStream_t)stream, src, B, L, D, start, count, dst);
  return ivh_host::check_launch("rows_window");
}

extern "C" int ivh_rows_window_bwd(const void* src, int src_bf16, int B, int L, int D, int start, int count, float* dst, void* stream) {
  IVH_REQUIRE(src && dst && B > 0 && D % 8 == 0 && start >= 0 && count > 0 && start + count <= L, "rows_window_bwd: bad args");
  dim3 grid = grid1d((long)B * L * (D / 8));
  if (src_bf16) hipLaunchKernelGGL((rows_window_bwd_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, B, L, D, start, count, dst);
  else hipLaunchKernelGGL((rows_window_bwd_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)src, B, L, D, start, count, dst);
  return ivh_host::check_launch("rows_window_bwd");
}

extern "C" int ivh_pixel_target(const void* video, int video_fp32, const int32_t* msk_idx, int B, int C, int T, int H, int W,
                                int tubelet, int patch, int Nmask, int normalize, const float* mean3, const float* std3,
                                float* out, void* stream) {
  IVH_REQUIRE(video && msk_idx && out && mean3 && std3 && B > 0 && Nmask > 0, "pixel_target: bad args");
  IVH_REQUIRE(C == 3, "pixel_target: 3 channels expected (got %d)", C);
  IVH_REQUIRE(H % patch == 0 && W % patch == 0 && T % tubelet == 0, "pixel_target: frame %dx%dx%d not divisible by (%d,%d,%d)", T, H, W, tubelet, patch, patch);
  IVH_REQUIRE(!normalize || tubelet * patch * patch > 1, "pixel_target: the unbiased variance needs more than one pixel per cube");
  dim3 grid((unsigned)((long)B * Nmask));
  hipStream_t s = (hipStream_t)stream;
  if (video_fp32)
    hipLaunchKernelGGL((pixel_target_kernel<float>), grid, dim3(256), 0, s, (const float*)video, msk_idx, C, T, H, W, tubelet, patch, Nmask, normalize,
                       mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], out);
  else
    hipLaunchKernelGGL((pixel_target_kernel<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)video, msk_idx, C, T, H, W, tubelet, patch, Nmask, normalize,
                       mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], out);
  return ivh_host::check_launch("pixel_target");
}

extern "C" int ivh_mse_rows(const void* pred, int pred_fp32, const float* target, int M, int C, float dscale, float* rows,
                            uint16_t* dpred, void* stream) {
  IVH_REQUIRE(pred && target && rows && M > 0 && C > 0, "mse_rows: bad args");
  if (pred_fp32) hipLaunchKernelGGL((mse_rows_kernel<float>), dim3(M), dim3(256), 0, (hipStream_t)stream, (const float*)pred, target, C, dscale, rows, dpred);
  else hipLaunchKernelGGL((mse_rows_kernel<bf16_t>), dim3(M), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)pred, target, C, dscale, rows, dpred);
  return ivh_host::check_launch("mse_rows");
}

extern "C" int ivh_cosine_rows(const void* s, int s_fp32, const void* t, int t_fp32, int M, int C, float dscale, float* rows,
                               uint16_t* ds, void* stream) {
  IVH_REQUIRE(s && t && rows && M > 0 && C > 0, "cosine_rows: bad args");
  hipStream_t st = (hipStream_t)stream;
  if (s_fp32 && t_fp32) hipLaunchKernelGGL((cosine_rows_kernel<float, float>), dim3(M), dim3(256), 0, st, (const float*)s, (const float*)t, C, dscale, rows, ds);
  else if (s_fp32) hipLaunchKernelGGL((cosine_rows_kernel<float, bf16_t>), dim3(M), dim3(256), 0, st, (const float*)s, (const bf16_t*)t, C, dscale, rows, ds);
  else if (t_fp32) hipLaunchKernelGGL((cosine_rows_kernel<bf16_t, float>), dim3(M), dim3(256), 0, st, (const bf16_t*)s, (const float*)t, C, dscale, rows, ds);
  else hipLaunchKernelGGL((cosine_rows_kernel<bf16_t, bf16_t>), dim3(M), dim3(256), 0, st, (const bf16_t*)s, (const bf16_t*)t, C, dscale, rows, ds);
  return ivh_host::check_launch("cosine_rows");
}
