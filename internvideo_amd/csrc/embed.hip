// Token-edge kernels (SURVEY.md 8(a) rows a1-a3, a11, a13/a14 inputs): mask -> gather indices, visible-only
// tubelet im2col of the (B,3,T,H,W) frame tensor, cls / positional-embedding assembly of the fp32 residual
// stream, decoder-input gathers and the scatter-free (inverse-index) positional-embedding gradients.
#include "common.h"
#include "../../include/internvideo_hip.h"

namespace ivh {

// one workgroup per clip: ascending ids of the kept tokens (mask == 0) and the inverse map.
__global__ __launch_bounds__(256) void mask_to_indices_kernel(const uint8_t* __restrict__ mask, int N1, int L,
                                                              int32_t* __restrict__ vis_idx, int32_t* __restrict__ inv_idx,
                                                              int32_t* __restrict__ count_out) {
  __shared__ int sc[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int per = (N1 + 255) / 256;
  const int beg = tid * per, end = min(beg + per, N1);
  const uint8_t* mrow = mask + (long)b * N1;
  int cnt = 0;
  for (int n = beg; n < end; ++n) cnt += mrow[n] == 0;
  sc[tid] = cnt;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const int v = tid >= o ? sc[tid - o] : 0;
    __syncthreads();
    sc[tid] += v;
    __syncthreads();
  }
  int pos = sc[tid] - cnt;
  for (int n = beg; n < end; ++n) {
    if (mrow[n] == 0) {
      if (pos < L) vis_idx[(long)b * L + pos] = n;
      inv_idx[(long)b * N1 + n] = pos < L ? pos : -1;
      ++pos;
    } else {
      inv_idx[(long)b * N1 + n] = -1;
    }
  }
  if (tid == 255) count_out[b] = sc[255];
}

// cols[(b, j), k] = bf16(video[b, c, t*tub + dt, ph*p + dy, pw*p + dx]), k = ((c*tub + dt)*p + dy)*p + dx (Conv3d weight
// flattening order), token = vis_idx[b, j+1] - 1 = (t*gh + ph)*gw + pw ; k >= Kreal -> 0.
template <typename T>
__global__ __launch_bounds__(256) void patch_im2col_kernel(const T* __restrict__ video, const int32_t* __restrict__ vis_idx,
                                                           int C, int Tn, int Hn, int Wn, int tub, int p, int L, int Kp,
                                                           bf16_t* __restrict__ cols) {
  const int row = blockIdx.x;            // b * (L-1) + j
  const int b = row / (L - 1), j = row % (L - 1);
  const int tok = vis_idx[(long)b * L + j + 1] - 1;
  const int gw = Wn / p, gh = Hn / p;
  const int pw = tok % gw, ph = (tok / gw) % gh, t = tok / (gw * gh);
  const int Kreal = C * tub * p * p;
  for (int k = threadIdx.x; k < Kp; k += 256) {
    float val = 0.f;
    if (k < Kreal) {
      const int dx = k % p, dy = (k / p) % p, dt = (k / (p * p)) % tub, c = k / (p * p * tub);
      const long src = (((long)b * C + c) * Tn + t * tub + dt) * Hn * Wn + (long)(ph * p + dy) * Wn + pw * p + dx;
      if constexpr (sizeof(T) == 4) val = video[src];
      else val = bf2f(video[src]);
    }
    cols[(long)row * Kp + k] = f2bf(val);
  }
}

// x0[b,0] = cls + pos[0] ; x0[b,j] = tok[b,j-1] + pos[vis_idx[b,j]]     (fp32)
__global__ __launch_bounds__(256) void assemble_tokens_kernel(const bf16_t* __restrict__ tok, const float* __restrict__ cls,
                                                              const float* __restrict__ pos, const int32_t* __restrict__ vis_idx,
                                                              int B, int L, int D, float* __restrict__ x0) {
  const int nch = D >> 3;
  const long id = (long)blockIdx.x * 256 + threadIdx.x;
  if (id >= (long)B * L * nch) return;
  const int c = id % nch;
  const long bj = id / nch;
  const int j = bj % L, b = bj / L;
  const int n = vis_idx[bj];
  float v[8];
  if (j == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = cls[c * 8 + e];
  } else {
    unpack8(*reinterpret_cast<const u32x4*>(tok + ((long)b * (L - 1) + j - 1) * D + c * 8), v);
  }
  const float* pr = pos + (long)n * D + c * 8;
  float* o = x0 + bj * D + c * 8;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = v[e] + pr[e];
}

// y[b,j] = bf16(x[b,j+skip] + pos[vis_idx[b,j+skip] - skip]);  x = a tap of the residual stream, fp32 or bf16
template <typename TX>
__global__ __launch_bounds__(256) void add_pos_gather_kernel(const TX* __restrict__ x, const float* __restrict__ pos,
                                                             const int32_t* __restrict__ vis_idx, int B, int L, int D, int skip,
                                                             bf16_t* __restrict__ y) {
  const int nch = D >> 3, Lo = L - skip;
  const long id = (long)blockIdx.x * 256 + threadIdx.x;
  if (id >= (long)B * Lo * nch) return;
  const int c = id % nch;
  const long bj = id / nch;
  const int j = bj % Lo, b = bj / Lo;
  const int n = vis_idx[(long)b * L + j + skip] - skip;
  const TX* xr = x + ((long)b * L + j + skip) * D + c * 8;
  const float* pr = pos + (long)n * D + c * 8;
  float v[8];
  if constexpr (sizeof(TX) == 4) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = xr[e];
  } else {
    unpack8(*reinterpret_cast<const u32x4*>(xr), v);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] += pr[e];
  *reinterpret_cast<u32x4*>(y + bj * D + c * 8) = pack8(v);
}

// dst[b, j+skip] = src[b, j] on bf16 rows, rows j < skip zeroed: the gradient of a bf16 tap whose consumer dropped the first `skip` rows
__global__ __launch_bounds__(256) void rows_shift_bf16_kernel(bf16_t* __restrict__ dst, const bf16_t* __restrict__ src, int B, int L, int D, int skip) {
  const int nch = D >> 3;
  const long id = (long)blockIdx.x * 256 + threadIdx.x;
  if (id >= (long)B * L * nch) return;
  const int c = id % nch;
  const long bj = id / nch;
  const int j = bj % L, b = bj / L;
  u32x4 v = {0u, 0u, 0u, 0u};
  if (j >= skip) v = *reinterpret_cast<const u32x4*>(src + ((long)b * (L - skip) + j - skip) * D + c * 8);
  *reinterpret_cast<u32x4*>(dst + bj * D + c * 8) = v;
}

// dst[b, j+skip] (+)= src[b, j]   (fp32 <- bf16 or fp32);  rows j < skip of a fresh dst are zeroed
template <typename T>
__global__ __launch_bounds__(256) void accum_rows_kernel(float* __restrict__ dst, const T* __restrict__ src, int B, int L, int D,
                                                         int skip, int accumulate) {
  const int nch = D >> 3;
  const long id = (long)blockIdx.x * 256 + threadIdx.x;
  if (id >= (long)B * L * nch) return;
  const int c = id % nch;
  const long bj = id / nch;
  const int j = bj % L, b = bj / L;
  float* d = dst + bj * D + c * 8;
  float v[8];
  if (j < skip) {
    if (!accumulate) {
#pragma unroll
      for (int e = 0; e < 8; ++e) d[e] = 0.f;
    }
    return;
  }
  const long so = ((long)b * (L - skip) + j - skip) * D + c * 8;
  if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = src[so + e];
  } else {
    unpack8(*reinterpret_cast<const u32x4*>(src + so), v);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) d[e] = accumulate ? d[e] + v[e] : v[e];
}

// fp32 rows [B][L][D] -> bf16 rows without the first `skip` rows of every clip
__global__ __launch_bounds__(256) void rows_to_bf16_kernel(const float* __restrict__ src, int B, int L, int D, int skip,
                                                           bf16_t* __restrict__ dst) {
  const int nch = D >> 3, Lo = L - skip;
  const long id = (long)blockIdx.x * 256 + threadIdx.x;
  if (id >= (long)B * Lo * nch) return;
  const int c = id % nch;
  const long bj = id / nch;
  const int j = bj % Lo, b = bj / Lo;
  const float* s = src + ((long)b * L + j + skip) * D + c * 8;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = s[e];
  *reinterpret_cast<u32x4*>(dst + bj * D + c * 8) = pack8(v);
}

// dpos[n] (+)= sum_k sum_b ( j = inv[b][n+skip] ; j >= 0 ? src[k][b][j-skip] : 0 )     scatter-free & deterministic
template <typename T>
__global__ __launch_bounds__(256) void pos_grad_kernel(const T* __restrict__ src, int K, int B, int Lsrc, int D,
                                                       const int32_t* __restrict__ inv_idx, int N1, int skip, int Npos,
                                                       float* __restrict__ dpos, int accumulate) {
  const int nch = D >> 3;
  const long id = (long)blockIdx.x * 256 + threadIdx.x;
  if (id >= (long)Npos * nch) return;
  const int c = id % nch;
  const int n = id / nch;
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // four samples per trip: their index loads, then their row loads, are independent (one sample at a time was a chain of B dependent
  // index -> row round trips: 104 us for 150 MB); the order of the sum is fixed (b ascending; for K > 1, k inside groups of four b)
  for (int b0 = 0; b0 < B; b0 += 4) {
    int j[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) j[u] = (b0 + u < B) ? inv_idx[(long)(b0 + u) * N1 + n + skip] : -1;
    for (int k = 0; k < K; ++k) {
      float v[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (j[u] >= 0) {
          const long so = (((long)k * B + b0 + u) * Lsrc + (j[u] - skip)) * D + c * 8;
          if constexpr (sizeof(T) == 4) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[u][e] = src[so + e];
          } else {
            unpack8(*reinterpret_cast<const u32x4*>(src + so), v[u]);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (j[u] >= 0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) a[e] += v[u][e];
        }
      }
    }
  }
  float* o = dpos + (long)n * D + c * 8;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = accumulate ? o[e] + a[e] : a[e];
}

// dst[k, b, j, :] = src[k, b, idx[b, j + skip] - skip, :]   raw 16-byte chunks (bit-exact for any element type): the teacher-target
// gather `norm_clip[~mask].reshape(K, B, -1, C)` (engines/engine_for_pretraining.py:118-125)
__global__ __launch_bounds__(256) void gather_rows_kernel(const u32x4* __restrict__ src, const int32_t* __restrict__ idx, int K, int B,
                                                          int Nsrc, int L, int skip, int nch, u32x4* __restrict__ dst) {
  const int Lo = L - skip;
  const long id = (long)blockIdx.x * 256 + threadIdx.x;
  if (id >= (long)K * B * Lo * nch) return;
  const int c = id % nch;
  const long kbj = id / nch;
  const int j = kbj % Lo;
  const long kb = kbj / Lo;
  const int b = kb % B;
  const int n = idx[(long)b * L + j + skip] - skip;
  dst[kbj * nch + c] = src[(kb * Nsrc + n) * nch + c];
}

}  // namespace ivh

using namespace ivh;
static inline dim3 grid1d(long n) { return dim3((unsigned)((n + 255) / 256)); }

extern "C" int ivh_mask_to_indices(const uint8_t* mask, int B, int N1, int L, int32_t* vis_idx, int32_t* inv_idx,
                                   int32_t* count, void* stream) {
  IVH_REQUIRE(mask && vis_idx && inv_idx && count && B > 0 && N1 > 0 && L > 0 && L <= N1, "mask_to_indices: bad args");
  hipLaunchKernelGGL(mask_to_indices_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, mask, N1, L, vis_idx, inv_idx, count);
  return ivh_host::check_launch("mask_to_indices");
}

extern "C" int ivh_patch_im2col(const void* video, int video_fp32, const int32_t* vis_idx, int B, int C, int T, int H, int W,
                                int tubelet, int patch, int L, int Kp, uint16_t* cols, void* stream) {
  IVH_REQUIRE(video && vis_idx && cols && B > 0 && L > 1, "patch_im2col: bad args");
  IVH_REQUIRE(H % patch == 0 && W % patch == 0 && T % tubelet == 0, "patch_im2col: frame %dx%dx%d not divisible by (%d,%d,%d)", T, H, W, tubelet, patch, patch);
  IVH_REQUIRE(Kp % 8 == 0 && Kp >= C * tubelet * patch * patch, "patch_im2col: Kp=%d too small / not a multiple of 8", Kp);
  dim3 grid(B * (L - 1));
  if (video_fp32)
    hipLaunchKernelGGL((patch_im2col_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)video, vis_idx, C, T, H, W, tubelet, patch, L, Kp, cols);
  else
    hipLaunchKernelGGL((patch_im2col_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)video, vis_idx, C, T, H, W, tubelet, patch, L, Kp, cols);
  return ivh_host::check_launch("patch_im2col");
}

extern "C" int ivh_assemble_tokens(const uint16_t* tok, const float* cls, const float* pos, const int32_t* vis_idx,
                                   int B, int L, int D, float* x0, void* stream) {
  IVH_REQUIRE(tok && cls && pos && vis_idx && x0 && D % 8 == 0, "assemble_tokens: bad args");
  hipLaunchKernelGGL(assemble_tokens_kernel, grid1d((long)B * L * (D / 8)), dim3(256), 0, (hipStream_t)stream, tok, cls, pos, vis_idx, B, L, D, x0);
  return ivh_host::check_launch("assemble_tokens");
}

extern "C" int ivh_add_pos_gather(const float* x, const float* pos, const int32_t* vis_idx, int B, int L, int D, int skip,
                                  uint16_t* y, void* stream) {
  IVH_REQUIRE(x && pos && vis_idx && y && D % 8 == 0 && skip >= 0 && skip < L, "add_pos_gather: bad args");
  hipLaunchKernelGGL((add_pos_gather_kernel<float>), grid1d((long)B * (L - skip) * (D / 8)), dim3(256), 0, (hipStream_t)stream, x, pos, vis_idx, B, L, D, skip, y);
  return ivh_host::check_launch("add_pos_gather");
}

extern "C" int ivh_add_pos_gather_bf16(const uint16_t* x, const float* pos, const int32_t* vis_idx, int B, int L, int D, int skip,
                                       uint16_t* y, void* stream) {
  IVH_REQUIRE(x && pos && vis_idx && y && D % 8 == 0 && skip >= 0 && skip < L, "add_pos_gather_bf16: bad args");
  hipLaunchKernelGGL((add_pos_gather_kernel<bf16_t>), grid1d((long)B * (L - skip) * (D / 8)), dim3(256), 0, (hipStream_t)stream, x, pos, vis_idx, B, L, D, skip, y);
  return ivh_host::check_launch("add_pos_gather_bf16");
}

extern "C" int ivh_rows_shift_bf16(uint16_t* dst, const uint16_t* src, int B, int L, int D, int skip, void* stream) {
  IVH_REQUIRE(dst && src && D % 8 == 0 && skip >= 0 && skip < L, "rows_shift_bf16: bad args");
  hipLaunchKernelGGL(rows_shift_bf16_kernel, grid1d((long)B * L * (D / 8)), dim3(256), 0, (hipStream_t)stream, dst, src, B, L, D, skip);
  return ivh_host::check_launch("rows_shift_bf16");
}

extern "C" int ivh_accum_rows(float* dst, const void* src, int src_bf16, int B, int L, int D, int skip, int accumulate, void* stream) {
  IVH_REQUIRE(dst && src && D % 8 == 0 && skip >= 0 && skip < L, "accum_rows: bad args");
  dim3 grid = grid1d((long)B * L * (D / 8));
  if (src_bf16) hipLaunchKernelGGL((accum_rows_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, dst, (const bf16_t*)src, B, L, D, skip, accumulate);
  else hipLaunchKernelGGL((accum_rows_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, dst, (const float*)src, B, L, D, skip, accumulate);
  return ivh_host::check_launch("accum_rows");
}

extern "C" int ivh_rows_to_bf16(const float* src, int B, int L, int D, int skip, uint16_t* dst, void* stream) {
  IVH_REQUIRE(dst && src && D % 8 == 0 && skip >= 0 && skip < L, "rows_to_bf16: bad args");
  hipLaunchKernelGGL(rows_to_bf16_kernel, grid1d((long)B * (L - skip) * (D / 8)), dim3(256), 0, (hipStream_t)stream, src, B, L, D, skip, dst);
  return ivh_host::check_launch("rows_to_bf16");
}

extern "C" int ivh_pos_grad(const void* src, int src_bf16, int K, int B, int Lsrc, int D, const int32_t* inv_idx, int N1, int skip,
                            float* dpos, int accumulate, void* stream) {
  IVH_REQUIRE(src && inv_idx && dpos && D % 8 == 0 && K > 0 && B > 0 && skip >= 0, "pos_grad: bad args");
  const int Npos = N1 - skip;
  dim3 grid = grid1d((long)Npos * (D / 8));
  if (src_bf16) hipLaunchKernelGGL((pos_grad_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, K, B, Lsrc, D, inv_idx, N1, skip, Npos, dpos, accumulate);
  else hipLaunchKernelGGL((pos_grad_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)src, K, B, Lsrc, D, inv_idx, N1, skip, Npos, dpos, accumulate);
  return ivh_host::check_launch("pos_grad");
}

extern "C" int ivh_gather_rows(const void* src, int row_bytes, int K, int B, int Nsrc, const int32_t* idx, int L, int skip,
                               void* dst, void* stream) {
  IVH_REQUIRE(src && idx && dst && K > 0 && B > 0 && Nsrc > 0 && L > 0 && skip >= 0 && skip < L, "gather_rows: bad args");
  IVH_REQUIRE(row_bytes > 0 && row_bytes % 16 == 0, "gather_rows: rows must be a multiple of 16 bytes (got %d)", row_bytes);
  IVH_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "gather_rows: src / dst must be 16-byte aligned");
  const int nch = row_bytes / 16;
  hipLaunchKernelGGL(gather_rows_kernel, grid1d((long)K * B * (L - skip) * nch), dim3(256), 0, (hipStream_t)stream,
                     (const u32x4*)src, idx, K, B, Nsrc, L, skip, nch, (u32x4*)dst);
  return ivh_host::check_launch("gather_rows");
}
