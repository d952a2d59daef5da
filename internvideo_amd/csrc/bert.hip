// Row kernels of the stage-2 text / fusion tower (post-LN BERT, multi_modality/models/backbones/bert/xbert.py):
//   * word + token-type + position embedding gather -> LayerNorm (xbert.py:298-334) and its backward (scatter of the row gradients
//     into the three tables),
//   * LayerNorm(dense_out + residual) of BertSelfOutput / BertOutput (xbert.py:508-512, 592-596) and its backward,
//   * the row-wise cross entropy of the MLM head (xbert.py:1677-1682) and of the VTM head (criterions.py:177-181) with
//     ignore_index, forward and gradient in one pass over the logits.
// Same layout rules as norms.hip: one 64-lane wave owns one row, a lane owns 16-byte chunks lane, lane+64, ... so every wave-level
// access is 1 KiB contiguous; the row stays in registers between the statistics and the scaling pass.  The cross-entropy rows are
// up to 30528 logits long (61 KB): one 256-thread workgroup per row, an online (max, sum) pass and a gradient pass (the second
// read of the row comes from L2).
#include "common.h"
#include "../../include/internvideo_hip.h"

namespace ivh {
namespace bert {

__device__ __forceinline__ void ld8f(const float* p, float* o) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p);
  const f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
  o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
}
__device__ __forceinline__ void st8f(float* p, const float* v) {
  *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}
__device__ __forceinline__ void ld8b(const bf16_t* p, float* o) { unpack8(*reinterpret_cast<const u32x4*>(p), o); }
__device__ __forceinline__ void st8b(bf16_t* p, const float* v) { *reinterpret_cast<u32x4*>(p) = pack8(v); }

// Row source of the two LayerNorm flavours.  SUM: s = a + r (bf16 rows).  EMBED: s = (word[id] + type[0]) + pos[row % L] (fp32 tables,
// the association order of xbert.py:325-329).
struct RowSrc {
  const bf16_t* a;
  const bf16_t* r;
  const int* ids;
  const float* word;
  const float* pos;
  const float* type;
  int L;
  int act;   // SUM only: 1 = s = gelu_erf(a) (the MLM head transform, xbert.py:839-843: dense -> GELU -> LayerNorm; r must be NULL)
  DropCfg drop;   // hidden dropout (xbert.py:288,331,506,510,590,594), thresh 0 = off.  SUM: on `a` before the residual add; EMBED: on the output
};

template <bool EMBED>
__device__ __forceinline__ void load_row_chunk(const RowSrc& s, int row, int C, int c, float* v) {
  if constexpr (EMBED) {
    float p[8], t[8];
    ld8f(s.word + (long)s.ids[row] * C + c * 8, v);
    ld8f(s.type + c * 8, t);
    ld8f(s.pos + (long)(row % s.L) * C + c * 8, p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (v[e] + t[e]) + p[e];
  } else {
    ld8b(s.a + (long)row * C + c * 8, v);
    if (s.act) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
    }
    if (s.drop.thresh) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= drop_scale(s.drop, (unsigned long long)row * C + c * 8 + e);
    }
    if (s.r) {
      float q[8];
      ld8b(s.r + (long)row * C + c * 8, q);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += q[e];
    }
  }
}

template <int NCH, bool EMBED>
__global__ __launch_bounds__(256) void ln_fwd_kernel(RowSrc src, const float* __restrict__ w, const float* __restrict__ b, float eps, int M,
                                                     int C, bf16_t* __restrict__ y, float* __restrict__ stats) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = C >> 3;
  for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        load_row_chunk<EMBED>(src, row, C, c, v[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[i][e];
      }
    }
    const float mu = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mu; q += d * d; }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    if (lane == 0) { stats[row * 2] = mu; stats[row * 2 + 1] = rstd; }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        float wv[8], bv[8], o[8];
        ld8f(w + c * 8, wv); ld8f(b + c * 8, bv);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mu) * rstd * wv[e] + bv[e];
        if constexpr (EMBED) {
          if (src.drop.thresh) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] *= drop_scale(src.drop, (unsigned long long)row * C + c * 8 + e);
          }
        }
        st8b(y + (long)row * C + c * 8, o);
      }
    }
  }
}

// dx = LayerNorm'(dy (+ dy2)) at the recomputed row; SUM: dx (bf16) is the gradient of both addends; EMBED: the row gradient is added
// (fp32 atomics) to word[id] (skipped for the padding id, as nn.Embedding(padding_idx) does), pos[row % L] and type[0].
// dw_part / db_part [gridDim.x][C]: per-workgroup partial sums of the LayerNorm weight / bias gradient (finished by ivh_colsum_finish).
template <int NCH, bool EMBED>
__global__ __launch_bounds__(256) void ln_bwd_kernel(RowSrc src, const float* __restrict__ w, const float* __restrict__ stats,
                                                     const bf16_t* __restrict__ dy, const bf16_t* __restrict__ dy2, int M, int C,
                                                     bf16_t* __restrict__ dx, bf16_t* __restrict__ dx_a, float* __restrict__ dword, float* __restrict__ dpos,
                                                     float* __restrict__ dtype, int pad_id, float* __restrict__ dw_part,
                                                     float* __restrict__ db_part) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [4][C]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = C >> 3;
  float aw[NCH][8], ab[NCH][8];
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) { aw[i][e] = 0.f; ab[i][e] = 0.f; }
  for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
    const float mu = stats[row * 2], rstd = stats[row * 2 + 1];
    float xh[NCH][8], g[NCH][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        float xv[8], d1[8], wv[8];
        load_row_chunk<EMBED>(src, row, C, c, xv);
        ld8b(dy + (long)row * C + c * 8, d1);
        if constexpr (EMBED) {
          if (src.drop.thresh) {
#pragma unroll
            for (int e = 0; e < 8; ++e) d1[e] *= drop_scale(src.drop, (unsigned long long)row * C + c * 8 + e);
          }
        }
        if (dy2) {
          float d2[8];
          ld8b(dy2 + (long)row * C + c * 8, d2);
#pragma unroll
          for (int e = 0; e < 8; ++e) d1[e] += d2[e];
        }
        ld8f(w + c * 8, wv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          xh[i][e] = (xv[e] - mu) * rstd;
          g[i][e] = d1[e] * wv[e];
          aw[i][e] += d1[e] * xh[i][e];
          ab[i][e] += d1[e];
          s1 += g[i][e];
          s2 += g[i][e] * xh[i][e];
        }
      }
    }
    s1 = wave_sum(s1) / (float)C;
    s2 = wave_sum(s2) / (float)C;
    int id = 0;
    if constexpr (EMBED) id = src.ids[row];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        float r[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = rstd * (g[i][e] - s1 - xh[i][e] * s2);
        if constexpr (EMBED) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            if (id != pad_id) unsafeAtomicAdd(dword + (long)id * C + c * 8 + e, r[e]);
            unsafeAtomicAdd(dpos + (long)(row % src.L) * C + c * 8 + e, r[e]);
            unsafeAtomicAdd(dtype + c * 8 + e, r[e]);
          }
        } else {
          if (src.act) {                                   // d/da of gelu(a): the raw row comes back from L2
            float raw[8];
            ld8b(src.a + (long)row * C + c * 8, raw);
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] *= dgelu_erf(raw[e]);
          }
          st8b(dx + (long)row * C + c * 8, r);
          if (dx_a) {                                        // dropout on the branch: its gradient carries the mask, the residual's does not
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] *= drop_scale(src.drop, (unsigned long long)row * C + c * 8 + e);
            st8b(dx_a + (long)row * C + c * 8, r);
          }
        }
      }
    }
  }
  for (int pass = 0; pass < 2; ++pass) {
    float* dst = pass == 0 ? dw_part : db_part;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) st8f(red + wave * C + c * 8, pass == 0 ? aw[i] : ab[i]);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < C; d += 256) dst[(long)blockIdx.x * C + d] = red[d] + red[C + d] + red[2 * C + d] + red[3 * C + d];
  }
}

// ---- cross entropy over rows ---------------------------------------------------------------------------------------------------
// inv_count[0] = 1 / #{m : labels[m] != ignore}  (CrossEntropyLoss's mean over the kept rows; 0 kept rows -> inf -> nan loss, as torch)
__global__ __launch_bounds__(256) void ce_count_kernel(const int* __restrict__ labels, int M, int ignore, float* __restrict__ inv_count) {
  __shared__ int part[4];
  int n = 0;
  for (int m = threadIdx.x; m < M; m += 256) n += labels[m] != ignore;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) inv_count[0] = 1.0f / (float)(part[0] + part[1] + part[2] + part[3]);
}

__device__ __forceinline__ void online_merge(float& m, float& s, float m2, float s2) {
  const float mx = fmaxf(m, m2);
  s = (m == -INFINITY ? 0.f : s * __expf(m - mx)) + (m2 == -INFINITY ? 0.f : s2 * __expf(m2 - mx));
  m = mx;
}

// rows[m] = inv_count * (logsumexp(x[m, :V]) - x[m, label]) (0 for ignored rows);  dx[m, :] = bf16(dscale * inv_count * (softmax - onehot))
// (zeros for ignored rows and for the padding columns V..ld-1).
template <typename TX>
__global__ __launch_bounds__(256) void ce_rows_kernel(const TX* x, int ld, int V, const int* __restrict__ labels, int ignore,
                                                      const float* __restrict__ inv_count, float dscale, const float* __restrict__ dscale_dev,
                                                      float* __restrict__ rows, bf16_t* dx, int ldd) {
  __shared__ float sm[4], ss[4];
  const int row = blockIdx.x;
  const int label = labels[row];
  const TX* xr = x + (long)row * ld;
  bf16_t* dr = dx ? dx + (long)row * ldd : nullptr;
  const int tid = threadIdx.x;
  if (label == ignore) {
    if (tid == 0) rows[row] = 0.f;
    if (dr)
      for (int c = tid; c < (ldd >> 3); c += 256) *reinterpret_cast<u32x4*>(dr + c * 8) = u32x4{0u, 0u, 0u, 0u};
    return;
  }
  const int nfull = V >> 3;
  float xl;                                                  // read before any gradient is written (dx may alias x)
  if constexpr (sizeof(TX) == 4) xl = reinterpret_cast<const float*>(xr)[label];
  else xl = bf2f(reinterpret_cast<const bf16_t*>(xr)[label]);
  float m = -INFINITY, s = 0.f;
  for (int c = tid; c < nfull; c += 256) {
    float v[8];
    if constexpr (sizeof(TX) == 4) ld8f(reinterpret_cast<const float*>(xr) + c * 8, v);
    else ld8b(reinterpret_cast<const bf16_t*>(xr) + c * 8, v);
    float cm = v[0];
#pragma unroll
    for (int e = 1; e < 8; ++e) cm = fmaxf(cm, v[e]);
    float cs = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) cs += __expf(v[e] - cm);
    online_merge(m, s, cm, cs);
  }
  for (int e = nfull * 8 + tid; e < V; e += 256) {
    float v;
    if constexpr (sizeof(TX) == 4) v = reinterpret_cast<const float*>(xr)[e];
    else v = bf2f(reinterpret_cast<const bf16_t*>(xr)[e]);
    online_merge(m, s, v, 1.0f);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
    online_merge(m, s, m2, s2);
  }
  if ((tid & 63) == 0) { sm[tid >> 6] = m; ss[tid >> 6] = s; }
  __syncthreads();
  m = sm[0]; s = ss[0];
#pragma unroll
  for (int i = 1; i < 4; ++i) online_merge(m, s, sm[i], ss[i]);
  const float lse = m + __logf(s);
  const float ic = inv_count[0];
  if (tid == 0) rows[row] = ic * (lse - xl);
  if (!dr) return;
  const float gs = dscale * ic * (dscale_dev ? dscale_dev[0] : 1.0f);
  for (int c = tid; c < (ldd >> 3); c += 256) {
    float v[8], o[8];
    const int base = c * 8;
    if (base + 8 <= V) {
      if constexpr (sizeof(TX) == 4) ld8f(reinterpret_cast<const float*>(xr) + base, v);
      else ld8b(reinterpret_cast<const bf16_t*>(xr) + base, v);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (base + e < V) {
          if constexpr (sizeof(TX) == 4) v[e] = reinterpret_cast<const float*>(xr)[base + e];
          else v[e] = bf2f(reinterpret_cast<const bf16_t*>(xr)[base + e]);
        } else v[e] = -INFINITY;
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = gs * (__expf(v[e] - lse) - ((base + e) == label ? 1.0f : 0.0f));
    st8b(dr + base, o);
  }
}

}  // namespace bert
}  // namespace ivh

using namespace ivh;
using namespace ivh::bert;

static inline int nch_for(int C) { return (C / 8 + 63) / 64; }
static inline int row_grid(int M, int cap) { int g = (M + 3) / 4; return g < cap ? (g < 1 ? 1 : g) : cap; }

#define IVH_BERT_DISPATCH(nch, KERNEL, EMBED, grid, block, shmem, s, ...)                                    \
  switch (nch) {                                                                                             \
    case 1: hipLaunchKernelGGL((KERNEL<1, EMBED>), grid, block, shmem, s, __VA_ARGS__); break;               \
    case 2: hipLaunchKernelGGL((KERNEL<2, EMBED>), grid, block, shmem, s, __VA_ARGS__); break;               \
    case 3: hipLaunchKernelGGL((KERNEL<3, EMBED>), grid, block, shmem, s, __VA_ARGS__); break;               \
    case 4: hipLaunchKernelGGL((KERNEL<4, EMBED>), grid, block, shmem, s, __VA_ARGS__); break;               \
    default: ivh_host::set_error("row width %d not supported by the text-tower kernels (max 2048)", (nch) * 512); return -1; \
  }

static int make_drop(float p, uint32_t seed, DropCfg* d) {
  IVH_REQUIRE(p >= 0.0f && p < 1.0f, "dropout: p = %f outside [0, 1)", (double)p);
  const double t = (double)p * 4294967296.0;
  d->thresh = (unsigned)(t > 4294967295.0 ? 4294967295.0 : t);
  d->inv_keep = 1.0f / (1.0f - p);
  d->seed = seed;
  d->epoch = ivh_host::dropout_epoch();
  return 0;
}

extern "C" int ivh_add_layernorm_fwd(const uint16_t* a, const uint16_t* r, int act, const float* w, const float* b, float eps, int M, int C,
                                     uint16_t* y, float* stats, float drop_p, uint32_t seed, void* stream) {
  IVH_REQUIRE(a && w && b && y && stats && M > 0 && C > 0 && C % 8 == 0, "add_layernorm_fwd: bad args");
  IVH_REQUIRE(act == 0 || (act == 1 && !r), "add_layernorm_fwd: act must be 0, or 1 (GELU(erf) of a) without a residual");
  IVH_REQUIRE(!(act && drop_p > 0.f), "add_layernorm_fwd: dropout and the GELU flavour are not combined anywhere on the path");
  DropCfg dc;
  if (make_drop(drop_p, seed, &dc)) return -1;
  RowSrc src{a, r, nullptr, nullptr, nullptr, nullptr, 1, act, dc};
  IVH_BERT_DISPATCH(nch_for(C), ln_fwd_kernel, false, dim3(row_grid(M, 8192)), dim3(256), 0, (hipStream_t)stream, src, w, b, eps, M, C, y, stats);
  return ivh_host::check_launch("add_layernorm_fwd");
}

extern "C" int ivh_add_layernorm_bwd(const uint16_t* a, const uint16_t* r, int act, const float* w, const float* stats, const uint16_t* dy,
                                     const uint16_t* dy2, int M, int C, uint16_t* dx, uint16_t* dx_a, float* dw_part, float* db_part,
                                     float drop_p, uint32_t seed, void* stream) {
  IVH_REQUIRE(a && w && stats && dy && dx && dw_part && db_part && M > 0 && C > 0 && C % 8 == 0, "add_layernorm_bwd: bad args");
  IVH_REQUIRE(act == 0 || (act == 1 && !r), "add_layernorm_bwd: act must be 0, or 1 (GELU(erf) of a) without a residual");
  IVH_REQUIRE((drop_p > 0.f) == (dx_a != nullptr), "add_layernorm_bwd: dx_a (the masked gradient of the branch) goes with drop_p > 0");
  IVH_REQUIRE(!(act && drop_p > 0.f), "add_layernorm_bwd: dropout and the GELU flavour are not combined anywhere on the path");
  DropCfg dc;
  if (make_drop(drop_p, seed, &dc)) return -1;
  RowSrc src{a, r, nullptr, nullptr, nullptr, nullptr, 1, act, dc};
  const int grid = ivh_norm_bwd_parts(M);
  IVH_BERT_DISPATCH(nch_for(C), ln_bwd_kernel, false, dim3(grid), dim3(256), (size_t)4 * C * sizeof(float), (hipStream_t)stream, src, w, stats,
                    dy, dy2, M, C, dx, dx_a, (float*)nullptr, (float*)nullptr, (float*)nullptr, 0, dw_part, db_part);
  return ivh_host::check_launch("add_layernorm_bwd");
}

extern "C" int ivh_bert_embed_fwd(const int* ids, int M, int L, const float* word, const float* pos, const float* type, const float* w,
                                  const float* b, float eps, int C, uint16_t* y, float* stats, float drop_p, uint32_t seed, void* stream) {
  IVH_REQUIRE(ids && word && pos && type && w && b && y && stats && M > 0 && L > 0 && M % L == 0 && C > 0 && C % 8 == 0,
              "bert_embed_fwd: bad args");
  DropCfg dc;
  if (make_drop(drop_p, seed, &dc)) return -1;
  RowSrc src{nullptr, nullptr, ids, word, pos, type, L, 0, dc};
  IVH_BERT_DISPATCH(nch_for(C), ln_fwd_kernel, true, dim3(row_grid(M, 8192)), dim3(256), 0, (hipStream_t)stream, src, w, b, eps, M, C, y, stats);
  return ivh_host::check_launch("bert_embed_fwd");
}

extern "C" int ivh_bert_embed_bwd(const int* ids, int M, int L, const float* word, const float* pos, const float* type, const float* w,
                                  const float* stats, const uint16_t* dy, int C, int pad_id, float* dword, float* dpos, float* dtype,
                                  float* dw_part, float* db_part, float drop_p, uint32_t seed, void* stream) {
  IVH_REQUIRE(ids && word && pos && type && w && stats && dy && dword && dpos && dtype && dw_part && db_part && M > 0 && L > 0 &&
                  M % L == 0 && C > 0 && C % 8 == 0, "bert_embed_bwd: bad args");
  DropCfg dc;
  if (make_drop(drop_p, seed, &dc)) return -1;
  RowSrc src{nullptr, nullptr, ids, word, pos, type, L, 0, dc};
  const int grid = ivh_norm_bwd_parts(M);
  IVH_BERT_DISPATCH(nch_for(C), ln_bwd_kernel, true, dim3(grid), dim3(256), (size_t)4 * C * sizeof(float), (hipStream_t)stream, src, w, stats,
                    dy, (const bf16_t*)nullptr, M, C, (bf16_t*)nullptr, (bf16_t*)nullptr, dword, dpos, dtype, pad_id, dw_part, db_part);
  return ivh_host::check_launch("bert_embed_bwd");
}

extern "C" int ivh_ce_rows(const void* logits, int logits_fp32, int ld, int M, int V, const int* labels, int ignore_index, float dscale,
                           const float* dscale_dev, float* inv_count, float* rows, uint16_t* dlogits, int ldd, void* stream) {
  IVH_REQUIRE(logits && labels && inv_count && rows && M > 0 && V > 0 && ld >= V, "ce_rows: bad args");
  IVH_REQUIRE(ld % 8 == 0 && (!dlogits || (ldd % 8 == 0 && ldd >= V)), "ce_rows: leading dimensions must be multiples of 8 and >= V");
  IVH_REQUIRE((const void*)dlogits != logits || (!logits_fp32 && ldd == ld), "ce_rows: in-place gradients need bf16 logits and ldd == ld");
  hipLaunchKernelGGL(ce_count_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, labels, M, ignore_index, inv_count);
  if (logits_fp32)
    hipLaunchKernelGGL(ce_rows_kernel<float>, dim3(M), dim3(256), 0, (hipStream_t)stream, (const float*)logits, ld, V, labels, ignore_index,
                       (const float*)inv_count, dscale, dscale_dev, rows, dlogits, ldd);
  else
    hipLaunchKernelGGL(ce_rows_kernel<bf16_t>, dim3(M), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)logits, ld, V, labels, ignore_index,
                       (const float*)inv_count, dscale, dscale_dev, rows, dlogits, ldd);
  return ivh_host::check_launch("ce_rows");
}
