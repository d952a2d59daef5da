// Non-causal softmax attention for equal-length video-token sequences, gfx950.
//
// Forward : O = softmax(Q K^T * scale) V          (flash style: online softmax, S never leaves registers)
// Backward: two recompute kernels, no atomics, deterministic:
//            dq  : one workgroup per 64-query tile, loops over key tiles  -> dQ (and delta = <dO, O> per query, from its prologue)
//            dkdv: one workgroup per 64-key tile, loops over query tiles  -> dK, dV
//
// Head dims 64 / 88 (InternVideo2-1B: 1408 / 16) / 128 are handled by padding the contraction to HDP = 64 / 96 /
// 128 with zero chunks supplied at staging time (the padded columns never touch HBM).
//
// MFMA plan (16x16x32 bf16, 4 waves x 16 rows): the score tile is computed TRANSPOSED, S^T = K Q^T, so that a
// lane owns one query column (softmax statistics are per lane, reduced over the 4 lane groups with two
// shuffles) and the 4 consecutive keys a lane holds per 16-key tile are exactly the k-slots the next MFMA
// (O^T += V^T P^T) wants for its B operand: P goes from accumulator to operand registers with a bf16 pack
// and no cross-lane traffic.  V^T / K^T / Q^T / dO^T A-operands are produced from the ROW-MAJOR tiles in LDS
// with ds_read_b64_tr_b16, so no transposed copy of any activation exists in HBM.
#include <type_traits>
#include "common.h"
#include "../../include/internvideo_hip.h"

namespace ivh {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

constexpr int ATTN_FWD_QW = 1;

// v_exp_f32 as it is (2^x, denormal results flushed): libm's exp2f() wraps it in a range test, two selects, an add and an ldexp --
// six VALU issues per probability instead of one, in the innermost loop of all three attention kernels
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }


template <int HDP> struct AttnCfg {
  static constexpr int RS = HDP * 2 + 32;          // LDS row stride in bytes.  +32: both the b128 row reads (4 x 16-lane groups) and the
                                                   // b64 transposing reads (2 x 32-lane groups) are conflict-free for HDP 64 / 96 / 128 under the
                                                   // gfx950 bank map ((addr / 4) mod 64); +16 was 2-way on every read (tools/lds_bank_sim.py)
  static constexpr int CPR = HDP / 8;              // 16-byte chunks per row
  static constexpr int CPT = (64 * CPR) / 256;     // chunks per thread per 64-row tile
  static constexpr int KS = HDP / 32;              // k-steps over the head dim
  static constexpr int DT = HDP / 16;              // 16-wide output tiles over the head dim
  static constexpr int TILE = 64 * RS;
};

// global -> registers for one 64 x HDP tile (rows row0.. of a (b,h) slice); zero for rows >= nrows / cols >= hd
template <int HDP>
__device__ __forceinline__ void tile_load(const bf16_t* __restrict__ base, long sl, int row0, int nrows, int hd,
                                          u32x4* regs, int tid) {
  using C = AttnCfg<HDP>;
#pragma unroll
  for (int i = 0; i < C::CPT; ++i) {
    const int id = tid + 256 * i;
    const int r = id / C::CPR, cc = id % C::CPR;
    const int row = row0 + r;
    if (row < nrows && cc * 8 < hd) regs[i] = *reinterpret_cast<const u32x4*>(base + (long)row * sl + cc * 8);
    else regs[i] = u32x4{0u, 0u, 0u, 0u};
  }
}
template <int HDP>
__device__ __forceinline__ void tile_store(char* lds_tile, const u32x4* regs, int tid) {
  using C = AttnCfg<HDP>;
#pragma unroll
  for (int i = 0; i < C::CPT; ++i) {
    const int id = tid + 256 * i;
    const int r = id / C::CPR, cc = id % C::CPR;
    *reinterpret_cast<u32x4*>(lds_tile + r * C::RS + cc * 16) = regs[i];
  }
}
// operand fragment straight from HBM: row (one per lane & 15), 8 contiguous head-dim elements at 32*ks + 8*g
template <int HDP>
__device__ __forceinline__ void row_frags(const bf16_t* __restrict__ base, long sl, int row, int nrows, int hd,
                                          s16x8* f, int lane) {
  using C = AttnCfg<HDP>;
  const int g = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) {
    const int d = 32 * ks + 8 * g;
    if (row < nrows && d < hd) f[ks] = *reinterpret_cast<const s16x8*>(base + (long)row * sl + d);
    else f[ks] = s16x8{0, 0, 0, 0, 0, 0, 0, 0};
  }
}
// A operand [16 rows x 32 head-dim] of rows rbase.. from a row-major LDS tile
template <int HDP>
__device__ __forceinline__ s16x8 frag_rows(const char* tile, int rbase, int ks, int lane) {
  using C = AttnCfg<HDP>;
  return *reinterpret_cast<const s16x8*>(tile + (rbase + (lane & 15)) * C::RS + (32 * ks + 8 * (lane >> 4)) * 2);
}
// A operand [16 head-dim x 32 rows] = transposed read: head-dim cols 16*dt.., k-slots 8g+e <-> rows
// 32c + 4g + e (e < 4), 32c + 16 + 4g + e - 4 (e >= 4): the slot order of the packed S^T / S accumulators.
template <int HDP>
__device__ __forceinline__ s16x8 frag_cols_tr(const char* tile, int dt, int c, int lane) {
  using C = AttnCfg<HDP>;
  const int i = lane & 15, g = lane >> 4;
  const char* p = tile + (32 * c + 4 * g + (i >> 2)) * C::RS + (16 * dt + 4 * (i & 3)) * 2;
  const s16x4 t0 = lds_tr16(p);
  const s16x4 t1 = lds_tr16(p + 16 * C::RS);
  s16x8 r;
  r[0] = t0[0]; r[1] = t0[1]; r[2] = t0[2]; r[3] = t0[3];
  r[4] = t1[0]; r[5] = t1[1]; r[6] = t1[2]; r[7] = t1[3];
  return r;
}
__device__ __forceinline__ s16x8 pack_frag(const f32x4& lo, const f32x4& hi) {
  const u32x2 a = pack4(lo[0], lo[1], lo[2], lo[3]);
  const u32x2 b = pack4(hi[0], hi[1], hi[2], hi[3]);
  const u32x4 v = {a[0], a[1], b[0], b[1]};
  return __builtin_bit_cast(s16x8, v);
}

// =========================================================================================================
// Forward.  64 * QW queries per workgroup: each of the 4 waves owns QW 16-query column blocks, so every K fragment (ds_read_b128)
// and every transposed V fragment (2 x ds_read_b64_tr_b16) read from LDS feeds QW MFMAs.  Measured on the 1B shape (L = 417,
// hd 88): QW = 2 halves the LDS reads per MFMA but needs 208 VGPRs (2 waves / SIMD) and 4 x 128-query tiles for 417 queries
// (18 % padding instead of 7 %): 113 us against 106 us for QW = 1, so QW = 1 it is.
// DROP: dropout on the attention probabilities (BERT's attention_probs_dropout_prob, xbert.py:361,469): the probabilities that multiply V
// are masked / rescaled element by element (mask = hash(seed, ((b H + h) Lq + query) Lk_max + key), common.h), the softmax normaliser and
// the saved lse stay those of the undropped row -- O = (softmax(S) o M) V.  The backward kernels regenerate the same mask.
template <int HDP, bool DROP = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HDP <= 96 ? 4 : 3))) void attn_fwd_kernel(const bf16_t* __restrict__ q, long qsb, long qsl, long qsh,
                                                       const bf16_t* __restrict__ k,
                                                       const bf16_t* __restrict__ v, long sb, long sl, long sh,
                                                       bf16_t* __restrict__ out, long ob, long ol, long oh,
                                                       float* __restrict__ lse, int H, int Lq, int Lk_max, int hd, float scale,
                                                       const int32_t* __restrict__ kv_len, DropCfg drop = DropCfg{0u, 1.0f, 0u}) {
  using C = AttnCfg<HDP>;
  constexpr int QW = ATTN_FWD_QW;                        // 16-query blocks per wave
  __shared__ __attribute__((aligned(16))) char lds[2 * C::TILE];
  char* Kt = lds;
  char* Vt = lds + C::TILE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  // 1-D grid, XCD-contiguous remap, query tile fastest: the tiles of one (b, h) run on ONE XCD and share its L2 copy of K / V.
  const int ntq = (Lq + 64 * QW - 1) / (64 * QW);
  const int wid = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = wid / ntq;
  const int b = bh / H, h = bh - b * H, q0 = (wid - bh * ntq) * 64 * QW;
  const int Lk = kv_len ? max(1, min(kv_len[b], Lk_max)) : Lk_max;      // right-padded batches: clip b has kv_len[b] valid keys
  const bf16_t* qb = q + (long)b * qsb + (long)h * qsh;
  const bf16_t* kb = k + (long)b * sb + (long)h * sh;
  const bf16_t* vb = v + (long)b * sb + (long)h * sh;

  s16x8 qf[QW][C::KS];
  f32x4 o[QW][C::DT];
  float m[QW], l[QW];
  int qrow[QW];
#pragma unroll
  for (int w = 0; w < QW; ++w) {
    qrow[w] = q0 + (wave * QW + w) * 16 + (lane & 15);
    row_frags<HDP>(qb, qsl, qrow[w], Lq, hd, qf[w], lane);
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) o[w][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    m[w] = -INFINITY; l[w] = 0.f;
  }
  const float c2 = scale * LOG2E;

  u32x4 kr[C::CPT], vr[C::CPT];
  tile_load<HDP>(kb, sl, 0, Lk, hd, kr, tid);
  tile_load<HDP>(vb, sl, 0, Lk, hd, vr, tid);
  const int nt = (Lk + 63) / 64;
  // one key tile; RAGGED (compile time) only for the last one: the bounds tests / selects of the masking are ~20 % of the loop's VALU
  auto tile = [&](const int t, auto ragged_tag) __attribute__((always_inline)) {
    constexpr bool RAGGED = decltype(ragged_tag)::value;
    __syncthreads();                       // every wave is done reading the previous tile
    tile_store<HDP>(Kt, kr, tid);
    tile_store<HDP>(Vt, vr, tid);
    __syncthreads();
    if (!RAGGED) {                         // next tile's HBM reads fly under this tile's MFMAs (the ragged tile is the last)
      tile_load<HDP>(kb, sl, (t + 1) * 64, Lk, hd, kr, tid);
      tile_load<HDP>(vb, sl, (t + 1) * 64, Lk, hd, vr, tid);
    }
    // S^T tiles: rows = keys 16j + 4g + r, col = this lane's query (one per 16-query block)
    f32x4 s[QW][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int w = 0; w < QW; ++w) s[w][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        const s16x8 kfrag = frag_rows<HDP>(Kt, 16 * j, ks, lane);
#pragma unroll
        for (int w = 0; w < QW; ++w) s[w][j] = mfma16(kfrag, qf[w][ks], s[w][j]);
      }
    }
    s16x8 pf[QW][2];
#pragma unroll
    for (int w = 0; w < QW; ++w) {
      float mt = -INFINITY;                               // max of the RAW scores: the scale enters once, in the exp2 fma
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if constexpr (RAGGED) {
            const int key = t * 64 + 16 * j + 4 * g + r;
            if (key >= Lk) s[w][j][r] = -INFINITY;
          }
          mt = fmaxf(mt, s[w][j][r]);
        }
      mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      const float mn = fmaxf(m[w], mt * c2);              // c2 > 0
      const float alpha = fast_exp2(m[w] - mn);
      m[w] = mn;
      float ps = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s[w][j][r] = fast_exp2(fmaf(s[w][j][r], c2, -mn)); ps += s[w][j][r]; }
      l[w] = l[w] * alpha + ps;
      if constexpr (DROP) {
        const unsigned long long rowbase = ((unsigned long long)bh * Lq + qrow[w]) * (unsigned long long)Lk_max;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[w][j][r] *= drop_scale(drop, rowbase + (unsigned)(t * 64 + 16 * j + 4 * g + r));
      }
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[w][dt][r] *= alpha;
      pf[w][0] = pack_frag(s[w][0], s[w][1]);
      pf[w][1] = pack_frag(s[w][2], s[w][3]);
    }
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt) {
        const s16x8 vfrag = frag_cols_tr<HDP>(Vt, dt, c, lane);
#pragma unroll
        for (int w = 0; w < QW; ++w) o[w][dt] = mfma16(vfrag, pf[w][c], o[w][dt]);
      }
  };
  for (int t = 0; t + 1 < nt; ++t) tile(t, std::false_type{});
  tile(nt - 1, std::true_type{});
#pragma unroll
  for (int w = 0; w < QW; ++w) {
    float lw = l[w];
    lw += __shfl_xor(lw, 16, 64);
    lw += __shfl_xor(lw, 32, 64);
    const float inv = 1.0f / lw;
    if (qrow[w] < Lq) {
      if (g == 0 && lse) lse[((long)b * H + h) * Lq + qrow[w]] = m[w] * LN2 + logf(lw);
      bf16_t* op = out + (long)b * ob + (long)qrow[w] * ol + (long)h * oh;
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt) {
        const int d = 16 * dt + 4 * g;
        if (d < hd) *reinterpret_cast<u32x2*>(op + d) = pack4(o[w][dt][0] * inv, o[w][dt][1] * inv, o[w][dt][2] * inv, o[w][dt][3] * inv);
      }
    }
  }
}

// =========================================================================================================
// dK, dV for one 64-key tile; each wave owns 16 keys (one per lane & 15) and loops over all query tiles.
template <int HDP, bool DROP = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_bwd_dkdv_kernel(
    const bf16_t* __restrict__ q, long qsb, long qsl, long qsh,
    const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, long sb, long sl, long sh,
    const bf16_t* __restrict__ dout, long ob, long ol, long oh, const float* __restrict__ lse, const float* __restrict__ delta,
    bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, long dsb, long dsl, long dsh,
    int H, int Lq, int Lk, int hd, float scale, const int32_t* __restrict__ kv_len, DropCfg drop = DropCfg{0u, 1.0f, 0u},
    const int32_t* __restrict__ nb_dev = nullptr) {
  using C = AttnCfg<HDP>;
  __shared__ __attribute__((aligned(16))) char lds[2 * C::TILE + 512];
  char* Qt = lds;
  char* Dt = lds + C::TILE;
  float* lse_s = reinterpret_cast<float*>(lds + 2 * C::TILE);
  float* del_s = lse_s + 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  const int ntk = (Lk + 63) >> 6;                        // same XCD-aware 1-D grid as the forward: key tile fastest
  int nwg = gridDim.x;                                   // device-side clip count (ivh_flash_attn_bwd_dyn: head dims above 96 take this dK / dV
  if (nb_dev) {                                          // kernel): the remap runs over the REAL grid, see attn32_fwd_kernel
    nwg = min(nwg, ntk * H * max(0, *nb_dev));
    if ((int)blockIdx.x >= nwg) return;
  }
  const int wid = xcd_remap(blockIdx.x, nwg);
  const int bh = wid / ntk;
  const int b = bh / H, h = bh - b * H, k0 = (wid - bh * ntk) * 64;
  const bf16_t* qb = q + (long)b * qsb + (long)h * qsh;
  const bf16_t* kb = k + (long)b * sb + (long)h * sh;
  const bf16_t* vb = v + (long)b * sb + (long)h * sh;
  const bf16_t* dob = dout + (long)b * ob + (long)h * oh;
  const float* lseb = lse + ((long)b * H + h) * Lq;
  const float* delb = delta + ((long)b * H + h) * Lq;
  const int key = k0 + wave * 16 + (lane & 15);
  const int Lk_b = kv_len ? max(1, min(kv_len[b], Lk)) : Lk;            // keys >= Lk_b are padding: their dK / dV rows are written as zeros

  s16x8 kf[C::KS], vf[C::KS];
  row_frags<HDP>(kb, sl, key, Lk_b, hd, kf, lane);
  row_frags<HDP>(vb, sl, key, Lk_b, hd, vf, lane);
  f32x4 dkt[C::DT], dvt[C::DT];
#pragma unroll
  for (int dt = 0; dt < C::DT; ++dt) { dkt[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dvt[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const float c2 = scale * LOG2E;

  u32x4 qr[C::CPT], dr[C::CPT];
  tile_load<HDP>(qb, qsl, 0, Lq, hd, qr, tid);
  tile_load<HDP>(dob, ol, 0, Lq, hd, dr, tid);
  const int nt = (Lq + 63) / 64;
  for (int t = 0; t < nt; ++t) {
    __syncthreads();
    tile_store<HDP>(Qt, qr, tid);
    tile_store<HDP>(Dt, dr, tid);
    if (tid < 64) {
      const int qi = t * 64 + tid;
      lse_s[tid] = qi < Lq ? lseb[qi] * LOG2E : INFINITY;    // +inf -> P = 0 for padded queries
      del_s[tid] = qi < Lq ? delb[qi] : 0.f;
    }
    __syncthreads();
    if (t + 1 < nt) {
      tile_load<HDP>(qb, qsl, (t + 1) * 64, Lq, hd, qr, tid);
      tile_load<HDP>(dob, ol, (t + 1) * 64, Lq, hd, dr, tid);
    }
    // S and dP tiles: rows = queries 16qi + 4g + r, col = this lane's key
    f32x4 p[4], ds[4];
#pragma unroll
    for (int qi = 0; qi < 4; ++qi) {
      f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        s = mfma16(frag_rows<HDP>(Qt, 16 * qi, ks, lane), kf[ks], s);
        dp = mfma16(frag_rows<HDP>(Dt, 16 * qi, ks, lane), vf[ks], dp);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qq = 16 * qi + 4 * g + r;
        const float pv = fast_exp2(s[r] * c2 - lse_s[qq]);
        if constexpr (DROP) {                                  // dV sees the dropped probabilities, dS the dropped dP (delta already does)
          const float mk = drop_scale(drop, ((unsigned long long)bh * Lq + (unsigned)(t * 64 + qq)) * (unsigned long long)Lk + (unsigned)key);
          p[qi][r] = pv * mk;
          ds[qi][r] = pv * (dp[r] * mk - del_s[qq]);
        } else {
          p[qi][r] = pv;
          ds[qi][r] = pv * (dp[r] - del_s[qq]);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const s16x8 pf = pack_frag(p[2 * c], p[2 * c + 1]);
      const s16x8 dsf = pack_frag(ds[2 * c], ds[2 * c + 1]);
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt) {
        dvt[dt] = mfma16(frag_cols_tr<HDP>(Dt, dt, c, lane), pf, dvt[dt]);
        dkt[dt] = mfma16(frag_cols_tr<HDP>(Qt, dt, c, lane), dsf, dkt[dt]);
      }
    }
  }
  if (key < Lk) {
    bf16_t* dkp = dk + (long)b * dsb + (long)key * dsl + (long)h * dsh;
    bf16_t* dvp = dv + (long)b * dsb + (long)key * dsl + (long)h * dsh;
    const float live = key < Lk_b ? 1.0f : 0.0f;
    const float ks_ = scale * live;
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) {
      const int d = 16 * dt + 4 * g;
      if (d < hd) {
        *reinterpret_cast<u32x2*>(dkp + d) = pack4(dkt[dt][0] * ks_, dkt[dt][1] * ks_, dkt[dt][2] * ks_, dkt[dt][3] * ks_);
        *reinterpret_cast<u32x2*>(dvp + d) = pack4(dvt[dt][0] * live, dvt[dt][1] * live, dvt[dt][2] * live, dvt[dt][3] * live);
      }
    }
  }
}

// =========================================================================================================
// dQ for one 64-query tile; each wave owns 16 queries (one per lane & 15) and loops over all key tiles.
template <int HDP, bool DROP = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HDP <= 64 ? 4 : (HDP <= 96 ? 3 : 2)))) void attn_bwd_dq_kernel(
    const bf16_t* __restrict__ q, long qsb, long qsl, long qsh,
    const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, long sb, long sl, long sh,
    const bf16_t* __restrict__ out, const bf16_t* __restrict__ dout, long ob, long ol, long oh, const float* __restrict__ lse,
    float* __restrict__ delta, bf16_t* __restrict__ dq, long dqb, long dql, long dqh, int H, int Lq, int Lk_max, int hd, float scale,
    const int32_t* __restrict__ kv_len, DropCfg drop = DropCfg{0u, 1.0f, 0u}) {
  using C = AttnCfg<HDP>;
  __shared__ __attribute__((aligned(16))) char lds[2 * C::TILE];
  char* Kt = lds;
  char* Vt = lds + C::TILE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  const int ntq = (Lq + 63) >> 6;
  const int wid = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = wid / ntq;
  const int b = bh / H, h = bh - b * H, q0 = (wid - bh * ntq) * 64;
  const int Lk = kv_len ? max(1, min(kv_len[b], Lk_max)) : Lk_max;
  const bf16_t* qb = q + (long)b * qsb + (long)h * qsh;
  const bf16_t* kb = k + (long)b * sb + (long)h * sh;
  const bf16_t* vb = v + (long)b * sb + (long)h * sh;
  const bf16_t* dob = dout + (long)b * ob + (long)h * oh;
  const int qrow = q0 + wave * 16 + (lane & 15);

  s16x8 qf[C::KS], dof[C::KS];
  row_frags<HDP>(qb, qsl, qrow, Lq, hd, qf, lane);
  row_frags<HDP>(dob, ol, qrow, Lq, hd, dof, lane);
  const float lse2 = qrow < Lq ? lse[((long)b * H + h) * Lq + qrow] * LOG2E : INFINITY;
  // delta[q] = <dO[q], O[q]>: the lane holds 8 * KS elements of its query row (the other three 16-lane groups hold the rest), so the
  // row dot product is a register loop and two shuffles -- no separate pass over O and dO.  Written out for the dK/dV kernel.
  float del = 0.f;
  {
    s16x8 of[C::KS];
    row_frags<HDP>(out + (long)b * ob + (long)h * oh, ol, qrow, Lq, hd, of, lane);
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      float a[8], g8[8];
      unpack8(__builtin_bit_cast(u32x4, of[ks]), a);
      unpack8(__builtin_bit_cast(u32x4, dof[ks]), g8);
#pragma unroll
      for (int e = 0; e < 8; ++e) del += a[e] * g8[e];
    }
    del += __shfl_xor(del, 16, 64);
    del += __shfl_xor(del, 32, 64);
    if (g == 0 && qrow < Lq) delta[((long)b * H + h) * Lq + qrow] = del;
  }
  f32x4 dqt[C::DT];
#pragma unroll
  for (int dt = 0; dt < C::DT; ++dt) dqt[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float c2 = scale * LOG2E;

  u32x4 kr[C::CPT], vr[C::CPT];
  tile_load<HDP>(kb, sl, 0, Lk, hd, kr, tid);
  tile_load<HDP>(vb, sl, 0, Lk, hd, vr, tid);
  const int nt = (Lk + 63) / 64;
  auto tile = [&](const int t, auto ragged_tag) __attribute__((always_inline)) {
    constexpr bool RAGGED = decltype(ragged_tag)::value;
    __syncthreads();
    tile_store<HDP>(Kt, kr, tid);
    tile_store<HDP>(Vt, vr, tid);
    __syncthreads();
    if (!RAGGED) {
      tile_load<HDP>(kb, sl, (t + 1) * 64, Lk, hd, kr, tid);
      tile_load<HDP>(vb, sl, (t + 1) * 64, Lk, hd, vr, tid);
    }
    f32x4 ds[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        s = mfma16(frag_rows<HDP>(Kt, 16 * j, ks, lane), qf[ks], s);
        dp = mfma16(frag_rows<HDP>(Vt, 16 * j, ks, lane), dof[ks], dp);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float pv = fast_exp2(s[r] * c2 - lse2);
        if constexpr (RAGGED) {
          const int key = t * 64 + 16 * j + 4 * g + r;
          if (key >= Lk) pv = 0.f;
        }
        float dpr = dp[r];
        if constexpr (DROP)
          dpr *= drop_scale(drop, ((unsigned long long)bh * Lq + (unsigned)qrow) * (unsigned long long)Lk_max + (unsigned)(t * 64 + 16 * j + 4 * g + r));
        ds[j][r] = pv * (dpr - del);
      }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const s16x8 dsf = pack_frag(ds[2 * c], ds[2 * c + 1]);
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt) dqt[dt] = mfma16(frag_cols_tr<HDP>(Kt, dt, c, lane), dsf, dqt[dt]);
    }
  };
  for (int t = 0; t + 1 < nt; ++t) tile(t, std::false_type{});
  tile(nt - 1, std::true_type{});
  if (qrow < Lq) {
    bf16_t* dqp = dq + (long)b * dqb + (long)qrow * dql + (long)h * dqh;
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) {
      const int d = 16 * dt + 4 * g;
      if (d < hd) *reinterpret_cast<u32x2*>(dqp + d) = pack4(dqt[dt][0] * scale, dqt[dt][1] * scale, dqt[dt][2] * scale, dqt[dt][3] * scale);
    }
  }
}

}  // namespace ivh

using namespace ivh;

static int attn_check(const void* q, const void* k, const void* v, int64_t qsb, int64_t qsl, int64_t qsh,
                      int64_t sb, int64_t sl, int64_t sh, int B, int H, int Lq, int Lk, int hd) {
  IVH_REQUIRE(q && k && v, "flash_attn: null q/k/v");
  IVH_REQUIRE(B > 0 && H > 0 && Lq > 0 && Lk > 0, "flash_attn: empty problem B=%d H=%d Lq=%d Lk=%d", B, H, Lq, Lk);
  IVH_REQUIRE(hd > 0 && hd % 8 == 0 && hd <= 128, "flash_attn: head dim %d not supported (multiple of 8, <= 128)", hd);
  IVH_REQUIRE(sb % 8 == 0 && sl % 8 == 0 && sh % 8 == 0 && qsb % 8 == 0 && qsl % 8 == 0 && qsh % 8 == 0,
              "flash_attn: strides must be multiples of 8 elements");
  IVH_REQUIRE(((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)v % 16) == 0, "flash_attn: q/k/v must be 16-byte aligned");
  IVH_REQUIRE(H <= 65535 && B <= 65535, "flash_attn: B, H must be <= 65535");
  return 0;
}

// 32x32x16-MFMA kernels (flash_attn32.hip)
extern "C" int ivh_attn32_supported(int64_t qsb, int64_t qsl, int64_t qsh, int64_t sb, int64_t sl, int64_t sh,
                                    int64_t ob, int64_t ol, int64_t oh, int Lq, int Lk, int hd);
extern "C" int ivh_attn32_fwd_launch(const uint16_t* q, int64_t qsb, int64_t qsl, int64_t qsh, const uint16_t* k, const uint16_t* v, int64_t sb, int64_t sl, int64_t sh,
                                     uint16_t* out, int64_t ob, int64_t ol, int64_t oh, float* lse, int B, int H, int Lq, int Lk, int hd, float scale,
                                     const int32_t* kv_len, const int32_t* nb_dev, void* stream);
extern "C" int ivh_attn32_bwd_dq_launch(const uint16_t* q, int64_t qsb, int64_t qsl, int64_t qsh, const uint16_t* k, const uint16_t* v, int64_t sb, int64_t sl, int64_t sh,
                                        const uint16_t* out, const uint16_t* dout, int64_t ob, int64_t ol, int64_t oh, const float* lse, float* delta,
                                        uint16_t* dq, int64_t dqb, int64_t dql, int64_t dqh, int B, int H, int Lq, int Lk, int hd, float scale,
                                        const int32_t* kv_len, const int32_t* nb_dev, void* stream);
extern "C" int ivh_attn32_dkdv_lds_bytes(int Lq, int hd);
extern "C" int ivh_attn32_bwd_dkdv_launch(const uint16_t* q, int64_t qsb, int64_t qsl, int64_t qsh, const uint16_t* k, const uint16_t* v, int64_t sb, int64_t sl, int64_t sh,
                                          const uint16_t* dout, int64_t ob, int64_t ol, int64_t oh, const float* lse, const float* delta,
                                          uint16_t* dk, uint16_t* dv, int64_t dsb, int64_t dsl, int64_t dsh, int B, int H, int Lq, int Lk, int hd, float scale,
                                          const int32_t* kv_len, const int32_t* nb_dev, void* stream);
// 0 = automatic (the 32x32 kernels whenever they support the problem), 1 = the 16x16x32 kernels of this file, 2 = 32x32 or error
static int g_attn_impl = 0;
extern "C" int ivh_set_attn_kernel(int choice) {
  IVH_REQUIRE(choice >= 0 && choice <= 2, "ivh_set_attn_kernel: choice must be 0 (auto), 1 (16x16x32) or 2 (32x32x16)");
  g_attn_impl = choice;
  return 0;
}

#define IVH_ATTN_DISPATCH(hd, KERNEL, grid, s, ...)                                                     \
  if ((hd) <= 64) hipLaunchKernelGGL((KERNEL<64>), grid, dim3(256), 0, s, __VA_ARGS__);                 \
  else if ((hd) <= 96) hipLaunchKernelGGL((KERNEL<96>), grid, dim3(256), 0, s, __VA_ARGS__);            \
  else hipLaunchKernelGGL((KERNEL<128>), grid, dim3(256), 0, s, __VA_ARGS__);

// nb_dev (ivh_flash_attn_fwd_dyn / _bwd_dyn, ABI 2): int32 in HBM, the number of clips that exist -- the launch is sized for B, workgroups of
// clips at or past *nb_dev leave at once and none of their outputs is written (DropPath skipping: the kept clips of a branch are compacted
// to the front).  32x32 kernels only.
extern "C" int ivh_flash_attn_fwd_dyn(const uint16_t* q, int64_t qsb, int64_t qsl, int64_t qsh,
                                      const uint16_t* k, const uint16_t* v, int64_t sb, int64_t sl, int64_t sh,
                                      uint16_t* out, int64_t ob, int64_t ol, int64_t oh, float* lse,
                                      int B, int H, int Lq, int Lk, int hd, float scale, const int32_t* kv_len, const int32_t* nb_dev, void* stream) {
  if (attn_check(q, k, v, qsb, qsl, qsh, sb, sl, sh, B, H, Lq, Lk, hd)) return -1;
  IVH_REQUIRE(out && ((uintptr_t)out % 8) == 0 && ob % 4 == 0 && ol % 4 == 0 && oh % 4 == 0, "flash_attn_fwd: bad out");
  const bool ok32 = ((uintptr_t)out % 16) == 0 && ivh_attn32_supported(qsb, qsl, qsh, sb, sl, sh, ob, ol, oh, Lq, Lk, hd);
  IVH_REQUIRE(g_attn_impl != 2 || ok32, "flash_attn_fwd: the 32x32 kernel was requested but does not support these strides / sizes");
  IVH_REQUIRE(!nb_dev || (ok32 && g_attn_impl != 1), "flash_attn_fwd_dyn: a device-side clip count needs the 32x32 kernels, which do not support this layout");
  if (g_attn_impl != 1 && ok32)
    return ivh_attn32_fwd_launch(q, qsb, qsl, qsh, k, v, sb, sl, sh, out, ob, ol, oh, lse, B, H, Lq, Lk, hd, scale, kv_len, nb_dev, stream);
  dim3 grid((unsigned)((long)((Lq + 64 * ivh::ATTN_FWD_QW - 1) / (64 * ivh::ATTN_FWD_QW)) * H * B), 1, 1);
  IVH_ATTN_DISPATCH(hd, attn_fwd_kernel, grid, (hipStream_t)stream, q, (long)qsb, (long)qsl, (long)qsh, k, v, (long)sb, (long)sl, (long)sh,
                    out, (long)ob, (long)ol, (long)oh, lse, H, Lq, Lk, hd, scale, kv_len);
  return ivh_host::check_launch("flash_attn_fwd");
}
extern "C" int ivh_flash_attn_fwd(const uint16_t* q, int64_t qsb, int64_t qsl, int64_t qsh,
                                  const uint16_t* k, const uint16_t* v, int64_t sb, int64_t sl, int64_t sh,
                                  uint16_t* out, int64_t ob, int64_t ol, int64_t oh, float* lse,
                                  int B, int H, int Lq, int Lk, int hd, float scale, const int32_t* kv_len, void* stream) {
  return ivh_flash_attn_fwd_dyn(q, qsb, qsl, qsh, k, v, sb, sl, sh, out, ob, ol, oh, lse, B, H, Lq, Lk, hd, scale, kv_len, nullptr, stream);
}

static int drop_cfg(float p, uint32_t seed, DropCfg* d) {
  IVH_REQUIRE(p >= 0.0f && p < 1.0f, "flash_attn dropout: p = %f outside [0, 1)", (double)p);
  const double t = (double)p * 4294967296.0;
  d->thresh = (unsigned)(t > 4294967295.0 ? 4294967295.0 : t);
  d->inv_keep = 1.0f / (1.0f - p);
  d->seed = seed;
  d->epoch = ivh_host::dropout_epoch();
  return 0;
}

// Attention with dropout on the probabilities (the 16x16x32 kernels of this file, head dims <= 64: the text tower's).  Same arguments as
// ivh_flash_attn_fwd / _bwd plus the drop probability and the seed of the counter-based mask; the same (p, seed) must be given to both.
extern "C" int ivh_flash_attn_fwd_dropout(const uint16_t* q, int64_t qsb, int64_t qsl, int64_t qsh,
                                          const uint16_t* k, const uint16_t* v, int64_t sb, int64_t sl, int64_t sh,
                                          uint16_t* out, int64_t ob, int64_t ol, int64_t oh, float* lse,
                                          int B, int H, int Lq, int Lk, int hd, float scale, const int32_t* kv_len,
                                          float p_drop, uint32_t seed, void* stream) {
  if (attn_check(q, k, v, qsb, qsl, qsh, sb, sl, sh, B, H, Lq, Lk, hd)) return -1;
  IVH_REQUIRE(out && ((uintptr_t)out % 8) == 0 && ob % 4 == 0 && ol % 4 == 0 && oh % 4 == 0, "flash_attn_fwd_dropout: bad out");
  IVH_REQUIRE(hd <= 64, "flash_attn_fwd_dropout: built for head dims <= 64 (got %d)", hd);
  DropCfg d;
  if (drop_cfg(p_drop, seed, &d)) return -1;
  dim3 grid((unsigned)((long)((Lq + 64 * ivh::ATTN_FWD_QW - 1) / (64 * ivh::ATTN_FWD_QW)) * H * B), 1, 1);
  hipLaunchKernelGGL((attn_fwd_kernel<64, true>), grid, dim3(256), 0, (hipStream_t)stream, q, (long)qsb, (long)qsl, (long)qsh, k, v, (long)sb, (long)sl,
                     (long)sh, out, (long)ob, (long)ol, (long)oh, lse, H, Lq, Lk, hd, scale, kv_len, d);
  return ivh_host::check_launch("flash_attn_fwd_dropout");
}

extern "C" int ivh_flash_attn_bwd_dropout(const uint16_t* q, int64_t qsb, int64_t qsl, int64_t qsh,
                                          const uint16_t* k, const uint16_t* v, int64_t sb, int64_t sl, int64_t sh,
                                          const uint16_t* out, const uint16_t* dout, int64_t ob, int64_t ol, int64_t oh,
                                          const float* lse, float* delta, uint16_t* dq, int64_t dqb, int64_t dql, int64_t dqh,
                                          uint16_t* dk, uint16_t* dv, int64_t dsb, int64_t dsl, int64_t dsh,
                                          int B, int H, int Lq, int Lk, int hd, float scale, const int32_t* kv_len,
                                          float p_drop, uint32_t seed, void* stream) {
  if (attn_check(q, k, v, qsb, qsl, qsh, sb, sl, sh, B, H, Lq, Lk, hd)) return -1;
  IVH_REQUIRE(out && dout && lse && delta && dq && dk && dv, "flash_attn_bwd_dropout: null argument");
  IVH_REQUIRE(ob % 8 == 0 && ol % 8 == 0 && oh % 8 == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)dout % 16) == 0, "flash_attn_bwd_dropout: out/dout alignment");
  IVH_REQUIRE(dsb % 4 == 0 && dsl % 4 == 0 && dsh % 4 == 0 && dqb % 4 == 0 && dql % 4 == 0 && dqh % 4 == 0, "flash_attn_bwd_dropout: dq/dk/dv strides must be multiples of 4");
  IVH_REQUIRE(hd <= 64, "flash_attn_bwd_dropout: built for head dims <= 64 (got %d)", hd);
  DropCfg d;
  if (drop_cfg(p_drop, seed, &d)) return -1;
  hipStream_t s = (hipStream_t)stream;
  dim3 gk((unsigned)((long)((Lk + 63) / 64) * H * B), 1, 1), gq((unsigned)((long)((Lq + 63) / 64) * H * B), 1, 1);
  hipLaunchKernelGGL((attn_bwd_dq_kernel<64, true>), gq, dim3(256), 0, s, q, (long)qsb, (long)qsl, (long)qsh, k, v, (long)sb, (long)sl, (long)sh, out, dout,
                     (long)ob, (long)ol, (long)oh, lse, delta, dq, (long)dqb, (long)dql, (long)dqh, H, Lq, Lk, hd, scale, kv_len, d);
  hipLaunchKernelGGL((attn_bwd_dkdv_kernel<64, true>), gk, dim3(256), 0, s, q, (long)qsb, (long)qsl, (long)qsh, k, v, (long)sb, (long)sl, (long)sh, dout,
                     (long)ob, (long)ol, (long)oh, lse, delta, dk, dv, (long)dsb, (long)dsl, (long)dsh, H, Lq, Lk, hd, scale, kv_len, d);
  return ivh_host::check_launch("flash_attn_bwd_dropout");
}

extern "C" int ivh_flash_attn_bwd_dyn(const uint16_t* q, int64_t qsb, int64_t qsl, int64_t qsh,
                                      const uint16_t* k, const uint16_t* v, int64_t sb, int64_t sl, int64_t sh,
                                      const uint16_t* out, const uint16_t* dout, int64_t ob, int64_t ol, int64_t oh,
                                      const float* lse, float* delta, uint16_t* dq, int64_t dqb, int64_t dql, int64_t dqh,
                                      uint16_t* dk, uint16_t* dv, int64_t dsb, int64_t dsl, int64_t dsh,
                                      int B, int H, int Lq, int Lk, int hd, float scale, const int32_t* kv_len, const int32_t* nb_dev, void* stream) {
  if (attn_check(q, k, v, qsb, qsl, qsh, sb, sl, sh, B, H, Lq, Lk, hd)) return -1;
  IVH_REQUIRE(out && dout && lse && delta && dq && dk && dv, "flash_attn_bwd: null argument");
  IVH_REQUIRE(ob % 8 == 0 && ol % 8 == 0 && oh % 8 == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)dout % 16) == 0, "flash_attn_bwd: out/dout alignment");
  IVH_REQUIRE(dsb % 4 == 0 && dsl % 4 == 0 && dsh % 4 == 0 && dqb % 4 == 0 && dql % 4 == 0 && dqh % 4 == 0, "flash_attn_bwd: dq/dk/dv strides must be multiples of 4");
  hipStream_t s = (hipStream_t)stream;
  dim3 gk((unsigned)((long)((Lk + 63) / 64) * H * B), 1, 1), gq((unsigned)((long)((Lq + 63) / 64) * H * B), 1, 1);
  const bool al16 = ((uintptr_t)dq % 16) == 0 && ((uintptr_t)dk % 16) == 0 && ((uintptr_t)dv % 16) == 0 && dqb % 8 == 0 && dql % 8 == 0 && dqh % 8 == 0 &&
                    dsb % 8 == 0 && dsl % 8 == 0 && dsh % 8 == 0;
  const bool ok32 = al16 && ivh_attn32_supported(qsb, qsl, qsh, sb, sl, sh, ob, ol, oh, Lq, Lk, hd);
  IVH_REQUIRE(g_attn_impl != 2 || ok32, "flash_attn_bwd: the 32x32 kernels were requested but do not support these strides / sizes");
  const bool use32 = g_attn_impl != 1 && ok32;
  IVH_REQUIRE(!nb_dev || use32, "flash_attn_bwd_dyn: a device-side clip count needs the 32x32 dQ kernel, which does not support this layout");
  // dQ first: its prologue computes delta = <dO, O> per query row and leaves it in `delta` for the dK/dV kernel that follows
  if (use32) {
    if (ivh_attn32_bwd_dq_launch(q, qsb, qsl, qsh, k, v, sb, sl, sh, out, dout, ob, ol, oh, lse, delta, dq, dqb, dql, dqh, B, H, Lq, Lk, hd, scale, kv_len, nb_dev, stream)) return -1;
  } else {
    IVH_ATTN_DISPATCH(hd, attn_bwd_dq_kernel, gq, s, q, (long)qsb, (long)qsl, (long)qsh, k, v, (long)sb, (long)sl, (long)sh, out, dout, (long)ob, (long)ol, (long)oh,
                      lse, delta, dq, (long)dqb, (long)dql, (long)dqh, H, Lq, Lk, hd, scale, kv_len);
  }
  if (use32 && ivh_attn32_dkdv_lds_bytes(Lq, hd) > 0)       // head dims above 96 stay on the 16x16 dK/dV kernel (the 32x32 one would spill)
    return ivh_attn32_bwd_dkdv_launch(q, qsb, qsl, qsh, k, v, sb, sl, sh, dout, ob, ol, oh, lse, delta, dk, dv, dsb, dsl, dsh, B, H, Lq, Lk, hd, scale, kv_len, nb_dev, stream);
  IVH_ATTN_DISPATCH(hd, attn_bwd_dkdv_kernel, gk, s, q, (long)qsb, (long)qsl, (long)qsh, k, v, (long)sb, (long)sl, (long)sh, dout, (long)ob, (long)ol, (long)oh,
                    lse, delta, dk, dv, (long)dsb, (long)dsl, (long)dsh, H, Lq, Lk, hd, scale, kv_len, DropCfg{0u, 1.0f, 0u}, nb_dev);
  return ivh_host::check_launch("flash_attn_bwd");
}
extern "C" int ivh_flash_attn_bwd(const uint16_t* q, int64_t qsb, int64_t qsl, int64_t qsh,
                                  const uint16_t* k, const uint16_t* v, int64_t sb, int64_t sl, int64_t sh,
                                  const uint16_t* out, const uint16_t* dout, int64_t ob, int64_t ol, int64_t oh,
                                  const float* lse, float* delta, uint16_t* dq, int64_t dqb, int64_t dql, int64_t dqh,
                                  uint16_t* dk, uint16_t* dv, int64_t dsb, int64_t dsl, int64_t dsh,
                                  int B, int H, int Lq, int Lk, int hd, float scale, const int32_t* kv_len, void* stream) {
  return ivh_flash_attn_bwd_dyn(q, qsb, qsl, qsh, k, v, sb, sl, sh, out, dout, ob, ol, oh, lse, delta, dq, dqb, dql, dqh, dk, dv, dsb, dsl, dsh,
                                B, H, Lq, Lk, hd, scale, kv_len, nullptr, stream);
}
