// Non-causal softmax attention on 32x32x16 bf16 MFMAs, gfx950 -- the round-2 rewrite of flash_attn.hip.
//
// Why a second set of kernels: the 16x16x32 / 16-queries-per-wave kernels of flash_attn.hip are issue- and latency-bound (24 MFMA
// beside ~110 VALU, 17 v_exp and 36 LDS reads per 64 x 64 tile and wave; two barriers and a register -> LDS copy per key tile).
// Here a wave owns 32 queries (one 32-wide MFMA column block), so every LDS fragment feeds an MFMA of four times the work, the
// per-tile fixed costs (statistics, rescale, waits) are paid once per 32 x 64 scores, and the key / value tiles arrive by LDS-DMA
// (buffer_load ... lds, 16 bytes per lane) into a double buffer: no staging registers, no ds_write, ONE barrier per key tile.
//
// MFMA plan (v_mfma_f32_32x32x16_bf16; C layout col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)):
//   S^T[key][query] = K Q^T : A = K rows from LDS (ds_read_b128: lane -> key lane & 31, 8 head-dim elements 16 s + 8 (lane >> 5)),
//                             B = Q fragments held in registers for the whole kernel.  A lane owns ONE query column and 16 of every
//                             32 keys: the softmax statistics are per lane; the other half of the keys sits in lane ^ 32
//                             (one v_permlane32_swap per reduction, no LDS round trip).
//   O^T[d][query]  += V^T P^T: B = P straight from the S^T accumulator registers (regs 8 c .. 8 c + 7 <-> the 16 keys of k-slice c in
//                             the order (e & 3) + 8 (e >> 2) + 4 (lane >> 5)), A = V^T produced by ds_read_b64_tr_b16 from the
//                             ROW-MAJOR V tile with the same key order: no cross-lane traffic, no transposed copy anywhere.
//   Backward (two recompute kernels as before, deterministic, no atomics): dQ kernel = S^T, dP^T = V dO^T, dQ^T += K^T dS^T;
//   dK/dV kernel (a wave owns 32 keys) = S = Q K^T, dP = dO V^T, dV^T += dO^T P, dK^T += Q^T dS.
//
// LDS image of a 64-row tile: rows of HDP * 2 bytes with NO padding (the DMA image is lane-linear), 16-byte chunks permuted per row
// (rotation for the 12-chunk rows of HDP = 96, XOR for 8 / 16 chunks) on the DMA SOURCE address and on every read, chosen so that
// both the row reads (ds_read_b128, four 16-lane groups) and the transposing reads (two 32-lane groups) are conflict-free under the
// gfx950 bank map; tools/attn32_layout_check.py models the banks and emulates the whole index math against a dense reference.
// Head dims 64 / 88 / 96 / 128: the contraction is padded to HDP = 64 / 96 / 128 with zeros supplied by the DMA's bounds check.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "../../include/internvideo_hip.h"
#include "../../include/internvideo_hip_debug.h"

namespace ivh {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void a32_lds_void_t;

constexpr float A32_LOG2E = 1.4426950408889634f;
constexpr float A32_LN2 = 0.6931471805599453f;
constexpr unsigned A32_OOB = 0x80000000u;

template <int HDP> struct A32 {
  static constexpr int CPR = HDP / 8;              // 16-byte chunks per row
  static constexpr int RS = HDP * 2;               // row stride in bytes
  static constexpr int TILE = 64 * RS;             // 8 / 12 / 16 KiB
  static constexpr int RPW = TILE / 1024 / 4;      // 1 KiB DMA requests per wave and tile (4 waves): 2 / 3 / 4
  static constexpr int KS = HDP / 16;              // k-slices over the head dim
  static constexpr int MT = HDP / 32;              // 32-wide output tiles over the head dim
};

// physical chunk position of logical chunk c in tile row r, and its inverse
template <int HDP> __device__ __forceinline__ int a32_phys(int r, int c) {
  if constexpr (HDP == 96) { const int x = c + ((r >> 2) & 3); return x >= 12 ? x - 12 : x; }
  else if constexpr (HDP == 64) { const int u = (r >> 1) & 7; return c ^ (((u & 1) << 2) | (u >> 1)); }
  else return c ^ (((r & 3) << 2) | ((r >> 2) & 3));
}
template <int HDP> __device__ __forceinline__ int a32_logical(int r, int x) {
  if constexpr (HDP == 96) { const int c = x - ((r >> 2) & 3); return c < 0 ? c + 12 : c; }
  else return a32_phys<HDP>(r, x);
}

__device__ __forceinline__ f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ float a32_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float a32_max_halves(float x) {
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float a32_sum_halves(float x) {
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

typedef __attribute__((ext_vector_type(2))) float a32_f2;
// Packed fp32 arithmetic (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32: two elements per issue slot).  These kernels are bound by
// instruction issue (one instruction per ~4.5 cycles and SIMD, whatever its type), not by VALU lane throughput, so halving the number
// of softmax instructions is worth more than the packed ops' longer execution.
//   p[r] = exp2(x[r] * c - mneg[r]) for 16 accumulator registers, returns the sum of the p (accumulated pairwise)
__device__ __forceinline__ float a32_exp_rows(f32x16& x, float c, float m) {
  const a32_f2 cv = {c, c}, mv = {m, m};
  a32_f2 acc = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a32_f2 t = {x[2 * i], x[2 * i + 1]};
    t = t * cv - mv;
    t[0] = a32_exp2(t[0]); t[1] = a32_exp2(t[1]);
    x[2 * i] = t[0]; x[2 * i + 1] = t[1];
    acc += t;
  }
  return acc[0] + acc[1];
}

// per-lane DMA source offsets (bytes, relative to row 0 of the (b, h) slice) of the wave's RPW requests of a tile
template <int HDP>
__device__ __forceinline__ void a32_dma_offsets(int lane, int wave, long sl, int hd, unsigned* voff) {
  using C = A32<HDP>;
#pragma unroll
  for (int i = 0; i < C::RPW; ++i) {
    const int n = (wave * C::RPW + i) * 64 + lane;          // chunk index inside the tile = LDS position
    const int r = n / C::CPR, x = n - r * C::CPR;
    const int c = a32_logical<HDP>(r, x);
    voff[i] = (c * 8 < hd) ? (unsigned)(((long)r * sl + c * 8) * 2) : A32_OOB;
  }
}
// Buffer descriptor of one (b, h) slice as four plain dwords (raw buffer, no stride, bounds-checked: offsets past `bytes` read 0).
__device__ __forceinline__ u32x4 a32_rsrc(const void* base, int bytes) {
  const unsigned long long a = (unsigned long long)base;
  return u32x4{(unsigned)a, (unsigned)(a >> 32) & 0xffffu, (unsigned)bytes, 0x00020000u};
}
// One LDS-DMA request: 64 lanes x 16 bytes -> LDS [dst, dst + 1 KiB).  Inline asm on purpose: hipcc does not count it, so it never
// drains the queue (s_waitcnt vmcnt(0)) in front of an LDS read it cannot prove disjoint from the DMA's destination -- with the
// builtin it did exactly that before the first transposing read of every tile.  The kernels wait themselves (A32_WAIT_DMA) right
// before the barrier that publishes the tile.  M0 (the DMA's LDS base) is written in the statement that uses it and not restored:
// nothing else in these kernels reads M0 (gfx9+ DS instructions do not), and the three SALU per request it cost were 18 of the
// ~310 issue slots of a forward tile (the kernels are issue-bound: SQ counters in profiles/r2_attn_sq_counters_v1.md).
__device__ __forceinline__ void a32_dma16(u32x4 rs, unsigned lds_dst, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(lds_dst), "v"(voff), "s"(rs) : "memory");
}
// rows beyond the descriptor's range (past the last valid row) and the head-dim padding chunks read zeros.  `tile` = byte offset of
// the tile inside the kernel's (only) LDS array, which starts at LDS address 0.
template <int HDP>
__device__ __forceinline__ void a32_dma_tile(u32x4 rs, const unsigned* voff, unsigned toff, unsigned tile, int wave) {
  using C = A32<HDP>;
#pragma unroll
  for (int i = 0; i < C::RPW; ++i) a32_dma16(rs, tile + (unsigned)((wave * C::RPW + i) * 1024), voff[i] + toff);
}

// per-lane LDS byte offsets of the fragment reads (relative to the tile, sub-tile / slice terms are compile-time immediates)
template <int HDP> struct A32Lane {
  unsigned row[A32<HDP>::KS];           // row fragment of k-slice s: + (32 j) * RS
  unsigned tr[A32<HDP>::MT][2];         // transposed fragment of output tile mt, first / second 4-row group: + (32 j + 16 c) * RS
  __device__ __forceinline__ void init(int lane) {
    using C = A32<HDP>;
    const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int s = 0; s < C::KS; ++s) row[s] = (unsigned)(l31 * C::RS + a32_phys<HDP>(l31, 2 * s + hi) * 16);
    const int i = lane & 15, g = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
      for (int sec = 0; sec < 2; ++sec) {
        const int r = 4 * (g >> 1) + (i >> 2) + 8 * sec;     // + 32 j + 16 c: multiples of 16 leave the chunk permutation unchanged
        const int cl = 4 * mt + 2 * (g & 1) + ((i & 3) >> 1);
        tr[mt][sec] = (unsigned)(r * C::RS + a32_phys<HDP>(r, cl) * 16 + (i & 1) * 8) - (unsigned)(8 * sec * C::RS);
      }
  }
};
// The chunk permutation only depends on (r >> 1) & 7 (HDP 64), (r >> 2) & 3 (HDP 96) or r & 15 (HDP 128): adding 32 j + 16 c to the
// row never changes it, so sub-tile and k-slice offsets are plain immediates on top of the lane bases above.
template <int HDP> __device__ __forceinline__ u32x4 a32_row_frag(const char* tile, const A32Lane<HDP>& ln, int j, int s) {
  return *reinterpret_cast<const u32x4*>(tile + ln.row[s] + j * 32 * A32<HDP>::RS);
}
template <int HDP> __device__ __forceinline__ u32x4 a32_tr_frag(const char* tile, const A32Lane<HDP>& ln, int j, int c, int mt) {
  using C = A32<HDP>;
  const s16x4 t0 = lds_tr16(tile + ln.tr[mt][0] + (32 * j + 16 * c) * C::RS);
  const s16x4 t1 = lds_tr16(tile + ln.tr[mt][1] + (32 * j + 16 * c + 8) * C::RS);
  s16x8 r;
  r[0] = t0[0]; r[1] = t0[1]; r[2] = t0[2]; r[3] = t0[3];
  r[4] = t1[0]; r[5] = t1[1]; r[6] = t1[2]; r[7] = t1[3];
  return __builtin_bit_cast(u32x4, r);
}
__device__ __forceinline__ u32x4 a32_pack8(const f32x16& v, int c) {
  const u32x2 a = pack4(v[8 * c + 0], v[8 * c + 1], v[8 * c + 2], v[8 * c + 3]);
  const u32x2 b = pack4(v[8 * c + 4], v[8 * c + 5], v[8 * c + 6], v[8 * c + 7]);
  return u32x4{a[0], a[1], b[0], b[1]};
}
// B-operand fragments of one row (query or key = lane & 31) straight from HBM: 8 elements at 16 s + 8 (lane >> 5)
template <int HDP>
__device__ __forceinline__ void a32_row_frags_global(const bf16_t* __restrict__ base, long sl, int row, int nrows, int hd, u32x4* f, int lane) {
  using C = A32<HDP>;
  const int hi = lane >> 5;
#pragma unroll
  for (int s = 0; s < C::KS; ++s) {
    const int d = 16 * s + 8 * hi;
    if (row < nrows && d < hd) f[s] = *reinterpret_cast<const u32x4*>(base + (long)row * sl + d);
    else f[s] = u32x4{0u, 0u, 0u, 0u};
  }
}
// store a 32 x HDP^T accumulator set (lane: row = lane & 31, cols 32 mt + (r & 3) + 8 (r >> 2) + 4 hi) as bf16 rows, 16 bytes per
// lane and store: the 4-column pieces of the two lane halves are exchanged with v_permlane32_swap (guide T21)
template <int HDP>
__device__ __forceinline__ void a32_store_rows(const f32x16* acc, float mul, bf16_t* rowp, bool row_ok, int hd, int lane) {
  using C = A32<HDP>;
  const int hi = lane >> 5;
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      u32x2 a = pack4(acc[mt][8 * p + 0] * mul, acc[mt][8 * p + 1] * mul, acc[mt][8 * p + 2] * mul, acc[mt][8 * p + 3] * mul);
      u32x2 b = pack4(acc[mt][8 * p + 4] * mul, acc[mt][8 * p + 5] * mul, acc[mt][8 * p + 6] * mul, acc[mt][8 * p + 7] * mul);
      const auto r0 = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);
      const auto r1 = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
      const u32x4 v = {r0[0], r1[0], r0[1], r1[1]};
      const int d = 32 * mt + 16 * p + 8 * hi;
      if (row_ok && d < hd) *reinterpret_cast<u32x4*>(rowp + d) = v;
    }
}

#define A32_WAIT_DMA() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// (measured and dropped, profiles/r2_attn_variants_v1.md: s_setprio 1 around the MFMA clusters -- guide T5 -- changed nothing here; a
// run-time switch for it split the tile body into several basic blocks, which broke the read-ahead pipelines above: +12 %)

// Scheduling directive for a run of NM MFMAs that each consume RPM LDS fragment reads: the reads run AHEAD MFMAs ahead of their
// consumers.  (Left alone, the machine scheduler serialises read -> s_waitcnt -> MFMA through one register set to save VGPRs.)
template <int NM, int RPM, int AHEAD>
__device__ __forceinline__ void a32_sched_pipeline() {
  __builtin_amdgcn_sched_group_barrier(0x100, AHEAD * RPM, 0);
#pragma unroll
  for (int i = 0; i < NM - AHEAD; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, RPM, 0);
  }
  __builtin_amdgcn_sched_group_barrier(0x008, AHEAD, 0);
}

// =========================================================================================================
// Forward: one workgroup = 4 waves = 128 queries of one (b, h); grid = B * H * ceil(Lq / 128), query pass fastest, XCD-contiguous.
// DEFER: the running maximum is only raised (and O / l rescaled) when some query of the wave sees a score more than 8 (log2 units) above
// it (guide T13); until then P = exp2(s - m_stale) <= 256, which bf16 represents with the same relative precision.  Measurement aid
// (IVH_ATTN_DEFER=1), off by default: the default path rescales every tile.
// QKN (round-5 PROTOTYPE of the q/k-norm fusion, ivh_probe_attn32_fwd_qkn; not on the product path): q and k arrive UN-normalised (the qkv
// GEMM's output as it is) together with their per-token rstd over all heads (rq, rk: [B * L]) and the product of the two norm weights
// (wqk = q_norm.weight * k_norm.weight, [H * hd]).  softmax(scale q_hat k_hat^T) with q_hat = q rq wq, k_hat = k rk wk is computed as
// S[i][j] = scale rk[j] sum_d (q rq wq wk)[i][d] k[j][d]: Q is scaled once, at its load into registers; K stays raw (it arrives by LDS-DMA)
// and its per-key factor multiplies the scores after the MFMAs -- 16 packed multiplies + 8 LDS reads per 64-key tile and wave, which is
// what the prototype exists to price (the standalone qk_rmsnorm_fwd pass it would remove is 4.6 ms per step).
template <int HDP, bool DEFER, bool QKN>
__device__ __forceinline__ void attn32_fwd_body(
    const bf16_t* __restrict__ q, long qsb, long qsl, long qsh, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, long sb, long sl, long sh,
    bf16_t* __restrict__ out, long ob, long ol, long oh, float* __restrict__ lse, int H, int Lq, int Lk_max, int hd, float scale,
    const int32_t* __restrict__ kv_len, unsigned long long* __restrict__ stamps,
    const float* __restrict__ qkn_rq, const float* __restrict__ qkn_rk, const float* __restrict__ qkn_wqk,
    const int32_t* __restrict__ nb_dev) {
  using C = A32<HDP>;
  __shared__ __attribute__((aligned(16))) char lds[4 * C::TILE];            // [buffer][K, V]
  __shared__ __attribute__((aligned(16))) float rk_s[QKN ? 512 : 4];         // QKN: scale * log2 e * rk[key] of this clip (Lk <= 512)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // measurement aid (ivh_attn32_debug_stamps, tools/attn_timeline.py): shader-clock stamps of wave 0 at entry / loop start / loop end / exit
  unsigned long long t_in = 0, t_loop = 0, t_tail = 0;
  if (stamps) t_in = __builtin_readcyclecounter();
  const int hi = lane >> 5;
  const int npass = (Lq + 127) >> 7;
  // device-side clip count (DropPath skipping): only the first *nb_dev clips exist.  The launch is sized for all of them; the workgroups
  // BEHIND the real work in dispatch order leave at once and the XCD-contiguous remap runs over the real grid -- every XCD loses the same
  // share (dropping the tail of the remapped ids instead would idle one XCD: they are the last XCD's chunk)
  int nwg = gridDim.x;
  if (nb_dev) {
    nwg = min(nwg, npass * H * max(0, *nb_dev));
    if ((int)blockIdx.x >= nwg) return;
  }
  const int wid = xcd_remap(blockIdx.x, nwg);
  const int bh = wid / npass;
  const int b = bh / H, h = bh - b * H;
  const int q0 = (wid - bh * npass) * 128 + wave * 32;
  const int Lk = kv_len ? max(1, min(kv_len[b], Lk_max)) : Lk_max;
  const bf16_t* qb = q + (long)b * qsb + (long)h * qsh;
  const bf16_t* kb = k + (long)b * sb + (long)h * sh;
  const bf16_t* vb = v + (long)b * sb + (long)h * sh;
  const int range = (int)((((long)Lk - 1) * sl + hd) * 2);                   // bytes up to the end of the last valid row
  const u32x4 rs_k = a32_rsrc(kb, range);
  const u32x4 rs_v = a32_rsrc(vb, range);
  unsigned voff[C::RPW];
  a32_dma_offsets<HDP>(lane, wave, sl, hd, voff);
  A32Lane<HDP> ln;
  ln.init(lane);
  const unsigned tstep = (unsigned)(64 * sl * 2);

  a32_dma_tile<HDP>(rs_k, voff, 0u, 0u, wave);
  a32_dma_tile<HDP>(rs_v, voff, 0u, (unsigned)C::TILE, wave);

  const bool active = q0 < Lq;                                               // wave-uniform: a wave without queries only moves tiles
  const int qrow = q0 + (lane & 31);
  u32x4 qf[C::KS];
  a32_row_frags_global<HDP>(qb, qsl, qrow, Lq, hd, qf, lane);
  f32x16 o[C::MT];
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[mt][r] = 0.f;
  float m = -INFINITY, l = 0.f;
  const float c2 = QKN ? 1.0f : scale * A32_LOG2E;                           // QKN: the scale rides on the per-key factors
  const int nt = (Lk + 63) >> 6;
  if constexpr (QKN) {
    for (int i = threadIdx.x; i < nt * 64; i += 256) rk_s[i] = i < Lk ? qkn_rk[(long)b * Lk_max + i] * (scale * A32_LOG2E) : 0.f;
    const float rqv = qrow < Lq ? qkn_rq[(long)b * Lq + qrow] : 0.f;
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      const int d = 16 * ks + 8 * hi;
      if (d < hd) {
        float qv[8], wv[8];
        unpack8(qf[ks], qv);
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(qkn_wqk + (long)h * hd + d), w1 = *reinterpret_cast<const f32x4*>(qkn_wqk + (long)h * hd + d + 4);
        wv[0] = w0[0]; wv[1] = w0[1]; wv[2] = w0[2]; wv[3] = w0[3]; wv[4] = w1[0]; wv[5] = w1[1]; wv[6] = w1[2]; wv[7] = w1[3];
#pragma unroll
        for (int e = 0; e < 8; ++e) qv[e] *= rqv * wv[e];
        qf[ks] = pack8(qv);
      }
    }
  }

#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) asm volatile("" : "+v"(qf[ks]));      // the Q loads are waited for HERE: a tracked load still pending
  A32_WAIT_DMA();                                                             // inside the loop would make hipcc drain the DMA queue there
  if constexpr (QKN) __syncthreads();                                         // (+ the staged per-key factors)
  else __builtin_amdgcn_s_barrier();

  if (stamps) t_loop = __builtin_readcyclecounter();
  // PAR = buffer parity of tile t (compile time: every LDS offset of the tile body is an immediate)
  auto tile = [&](const int t, auto par_tag, auto ragged_tag) __attribute__((always_inline)) {
    constexpr bool RAGGED = decltype(ragged_tag)::value;
    constexpr int PAR = decltype(par_tag)::value;
    const char* Kt = lds + PAR * 2 * C::TILE;
    const char* Vt = Kt + C::TILE;
    if (!RAGGED) {                                      // the ragged tile is the last: nothing left to fetch
      constexpr unsigned nxt = (unsigned)((PAR ^ 1) * 2 * C::TILE);
      a32_dma_tile<HDP>(rs_k, voff, (unsigned)(t + 1) * tstep, nxt, wave);
      a32_dma_tile<HDP>(rs_v, voff, (unsigned)(t + 1) * tstep, nxt + (unsigned)C::TILE, wave);
    }
    if (active) {
      f32x16 s[2];
      {
        // fragment reads run four ahead of the MFMAs that consume them (the scheduler on its own re-serialises read -> wait ->
        // MFMA through one register set); the two accumulation chains alternate
        u32x4 kfr[2 * C::KS];
#pragma unroll
        for (int i = 0; i < 2 * C::KS; ++i) kfr[i] = a32_row_frag<HDP>(Kt, ln, i & 1, i >> 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[0][r] = 0.f; s[1][r] = 0.f; }
#pragma unroll
        for (int i = 0; i < 2 * C::KS; ++i) s[i & 1] = mfma32(kfr[i], qf[i >> 1], s[i & 1]);
        a32_sched_pipeline<2 * C::KS, 1, 4>();
      }
      if constexpr (QKN) {                                                   // S[key][query] *= scale log2 e rk[key]: registers 4 g .. 4 g + 3 <-> keys 8 g + 4 hi ..
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 rv = *reinterpret_cast<const f32x4*>(rk_s + t * 64 + 32 * j + 8 * g4 + 4 * hi);
            a32_f2 lo = a32_f2{s[j][4 * g4], s[j][4 * g4 + 1]} * a32_f2{rv[0], rv[1]};
            a32_f2 hi2 = a32_f2{s[j][4 * g4 + 2], s[j][4 * g4 + 3]} * a32_f2{rv[2], rv[3]};
            s[j][4 * g4] = lo[0]; s[j][4 * g4 + 1] = lo[1]; s[j][4 * g4 + 2] = hi2[0]; s[j][4 * g4 + 3] = hi2[1];
          }
      }
      float mt_ = -INFINITY;                                                 // max of the RAW scores: the scale enters once, in the exp2 fma
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if constexpr (RAGGED) {
            const int key = t * 64 + 32 * j + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (key >= Lk) s[j][r] = -INFINITY;
          }
          mt_ = fmaxf(mt_, s[j][r]);
        }
      mt_ = a32_max_halves(mt_);
      float mn = fmaxf(m, mt_ * c2);                                         // c2 > 0
      bool rescale = true;
      if constexpr (DEFER) rescale = __builtin_amdgcn_ballot_w64(mn - m > 8.0f) != 0;   // wave-uniform; first tile: m = -inf
      if (!rescale) mn = m;
      const float alpha = a32_exp2(m - mn);
      m = mn;
      const float ps = a32_exp_rows(s[0], c2, mn) + a32_exp_rows(s[1], c2, mn);
      if (rescale) {
        l = l * alpha + ps;
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[mt][r] *= alpha;
      } else {
        l += ps;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const u32x4 pf = a32_pack8(s[j], c);
#pragma unroll
          for (int mt = 0; mt < C::MT; ++mt) o[mt] = mfma32(a32_tr_frag<HDP>(Vt, ln, j, c, mt), pf, o[mt]);
        }
      a32_sched_pipeline<4 * C::MT, 2, 3>();
    }
    A32_WAIT_DMA();                                     // this wave's share of the next tile has landed ...
    __builtin_amdgcn_s_barrier();                       // ... everyone's has, and everyone is done reading this tile
  };
  {
    constexpr std::integral_constant<int, 0> P0{};
    constexpr std::integral_constant<int, 1> P1{};
    int t = 0;
    for (; t + 2 < nt; t += 2) { tile(t, P0, std::false_type{}); tile(t + 1, P1, std::false_type{}); }
    if (nt - t == 2) { tile(t, P0, std::false_type{}); tile(t + 1, P1, std::true_type{}); }
    else tile(t, P0, std::true_type{});
  }

  if (stamps) t_tail = __builtin_readcyclecounter();
  if (active) {
    const float lt = a32_sum_halves(l);
    const float inv = 1.0f / lt;
    const bool row_ok = qrow < Lq;
    if (row_ok && hi == 0 && lse) lse[((long)b * H + h) * Lq + qrow] = m * A32_LN2 + logf(lt);
    a32_store_rows<HDP>(o, inv, out + (long)b * ob + (long)qrow * ol + (long)h * oh, row_ok, hd, lane);
  }
  if (stamps && wave == 0 && lane == 0) {
    unsigned long long* r = stamps + (long)blockIdx.x * 4;
    r[0] = t_in; r[1] = t_loop; r[2] = t_tail; r[3] = __builtin_readcyclecounter();
  }
}

#define A32_FWD_PARAMS                                                                                                                              \
    const bf16_t* __restrict__ q, long qsb, long qsl, long qsh, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, long sb, long sl, long sh, \
    bf16_t* __restrict__ out, long ob, long ol, long oh, float* __restrict__ lse, int H, int Lq, int Lk_max, int hd, float scale,                       \
    const int32_t* __restrict__ kv_len, unsigned long long* __restrict__ stamps,                                                                      \
    const float* __restrict__ qkn_rq = nullptr, const float* __restrict__ qkn_rk = nullptr, const float* __restrict__ qkn_wqk = nullptr,              \
    const int32_t* __restrict__ nb_dev = nullptr
#define A32_FWD_ARGS q, qsb, qsl, qsh, k, v, sb, sl, sh, out, ob, ol, oh, lse, H, Lq, Lk_max, hd, scale, kv_len, stamps, qkn_rq, qkn_rk, qkn_wqk, nb_dev
template <int HDP, bool DEFER = false, bool QKN = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HDP <= 96 ? 3 : 2))) void attn32_fwd_kernel(A32_FWD_PARAMS) {
  attn32_fwd_body<HDP, DEFER, QKN>(A32_FWD_ARGS);
}
// The same body compiled WITHOUT the packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 become two plain instructions each).
// Measured in round 6 (tools/probes/mfma_valu_mix.hip, profiles/r6_mfma_valu_mix_v1.jsonl): the packed ops of one wave do NOT run beside another
// wave's MFMAs on the same SIMD (pair time = sum), while v_fma_f32 / v_mul_f32 / v_exp_f32 / v_max_f32 / v_cvt_pk_bf16_f32 overlap them completely.
template <int HDP, bool DEFER = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HDP <= 96 ? 3 : 2), target("no-packed-fp32-ops"))) void attn32_fwd_np_kernel(A32_FWD_PARAMS) {
  attn32_fwd_body<HDP, DEFER, false>(A32_FWD_ARGS);
}

// =========================================================================================================
// Forward, two wave groups one phase apart (round 6, VERDICT r5 next 2: "make the two pipes overlap").  One workgroup = 8 waves = 256 queries of
// one (b, h) = two wave per SIMD, one of group A (waves 0-3) and one of group B (waves 4-7).  A wave's tile is cut into an MFMA segment
//   X(i) = S(i) = K(i) Q^T  and  O += V(i-1)^T P(i-1)        (24 MFMAs, LDS fragment reads, nothing else)
// and a VALU segment
//   Y(i) = max / exp2 / sum / rescale of O / pack of P(i)     (no MFMA, no LDS)
// separated by workgroup barriers; group B runs ONE segment behind group A, so that in every phase each SIMD holds one wave in its MFMA segment
// and one in its VALU segment (the gemm256 ping-pong applied to attention).  K(i + 1) and V(i) are fetched by group A's waves at the start of X(i)
// (LDS-DMA) into the halves nobody reads during phases 2 i and 2 i + 1 and are waited for at the end of Y(i).
template <int HDP, bool PRIO, bool GSEL, bool STAMP>
__device__ __forceinline__ void attn32pp_fwd_body(
    const bf16_t* __restrict__ q, long qsb, long qsl, long qsh, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, long sb, long sl, long sh,
    bf16_t* __restrict__ out, long ob, long ol, long oh, float* __restrict__ lse, int H, int Lq, int Lk_max, int hd, float scale,
    const int32_t* __restrict__ kv_len, const int32_t* __restrict__ nb_dev, unsigned long long* __restrict__ stamps) {
  using C = A32<HDP>;
  __shared__ __attribute__((aligned(16))) char lds[4 * C::TILE];            // K[0] K[1] V[0] V[1]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = GSEL ? (wave & 1) : (wave >> 2), w4 = GSEL ? (wave >> 1) : (wave & 3);       // GSEL: the other guess at which waves share a SIMD
  const int hi = lane >> 5;
  // STAMP (measurement aid): workgroup 0's waves 0 and 4 write s_memtime at the start and end of every segment: stamps[(wave >> 2) * 256 + n]
  int stamp_n = 0;
  auto stamp = [&]() __attribute__((always_inline)) {
    if constexpr (STAMP) {
      if (blockIdx.x == 0 && (wave & 3) == 0) {
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (lane == 0 && stamp_n < 256) stamps[(wave >> 2) * 256 + stamp_n] = t;
        ++stamp_n;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  const int npass = (Lq + 255) >> 8;
  int nwg = gridDim.x;
  if (nb_dev) {
    nwg = min(nwg, npass * H * max(0, *nb_dev));
    if ((int)blockIdx.x >= nwg) return;
  }
  const int wid = xcd_remap(blockIdx.x, nwg);
  const int bh = wid / npass;
  const int b = bh / H, h = bh - b * H;
  const int q0 = (wid - bh * npass) * 256 + wave * 32;
  const int Lk = kv_len ? max(1, min(kv_len[b], Lk_max)) : Lk_max;
  const bf16_t* qb = q + (long)b * qsb + (long)h * qsh;
  const bf16_t* kb = k + (long)b * sb + (long)h * sh;
  const bf16_t* vb = v + (long)b * sb + (long)h * sh;
  const int range = (int)((((long)Lk - 1) * sl + hd) * 2);
  const u32x4 rs_k = a32_rsrc(kb, range);
  const u32x4 rs_v = a32_rsrc(vb, range);
  unsigned voff[C::RPW];
  a32_dma_offsets<HDP>(lane, w4, sl, hd, voff);
  A32Lane<HDP> ln;
  ln.init(lane);
  const unsigned tstep = (unsigned)(64 * sl * 2);

  if (grp == 0) a32_dma_tile<HDP>(rs_k, voff, 0u, 0u, w4);

  const bool active = q0 < Lq;
  const int qrow = q0 + (lane & 31);
  u32x4 qf[C::KS];
  a32_row_frags_global<HDP>(qb, qsl, qrow, Lq, hd, qf, lane);
  f32x16 o[C::MT];
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[mt][r] = 0.f;
  float m = -INFINITY, l = 0.f;
  const float c2 = scale * A32_LOG2E;
  const int nt = (Lk + 63) >> 6;
  f32x16 s[2];
  u32x4 pf[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) pf[i] = u32x4{0u, 0u, 0u, 0u};

#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) asm volatile("" : "+v"(qf[ks]));
  A32_WAIT_DMA();
  __builtin_amdgcn_s_barrier();
  if (grp == 1) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }   // B: one phase behind

  // O += V(i-1)^T P(i-1) from the V half of parity VPAR
  auto pv = [&](auto vpar_tag) __attribute__((always_inline)) {
    constexpr int VPAR = decltype(vpar_tag)::value;
    const char* Vt = lds + (2 + VPAR) * C::TILE;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt) o[mt] = mfma32(a32_tr_frag<HDP>(Vt, ln, j, c, mt), pf[2 * j + c], o[mt]);
  };
  auto xy = [&](const int i, auto par_tag, auto first_tag, auto last_tag) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_tag)::value;
    constexpr bool FIRST = decltype(first_tag)::value;
    constexpr bool LAST = decltype(last_tag)::value;                      // the last tile: ragged, nothing left to fetch
    // ---------------- X(i): MFMA segment
    stamp();
    if (grp == 0) {
      if (!LAST) a32_dma_tile<HDP>(rs_k, voff, (unsigned)(i + 1) * tstep, (unsigned)((PAR ^ 1) * C::TILE), w4);
      a32_dma_tile<HDP>(rs_v, voff, (unsigned)i * tstep, (unsigned)((2 + PAR) * C::TILE), w4);
    }
    if (active) {
      if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
      const char* Kt = lds + PAR * C::TILE;
      u32x4 kfr[2 * C::KS];
#pragma unroll
      for (int n = 0; n < 2 * C::KS; ++n) kfr[n] = a32_row_frag<HDP>(Kt, ln, n & 1, n >> 1);
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[0][r] = 0.f; s[1][r] = 0.f; }
#pragma unroll
      for (int n = 0; n < 2 * C::KS; ++n) s[n & 1] = mfma32(kfr[n], qf[n >> 1], s[n & 1]);
      a32_sched_pipeline<2 * C::KS, 1, 4>();
      if constexpr (!FIRST) {
        pv(std::integral_constant<int, PAR ^ 1>{});
        a32_sched_pipeline<4 * C::MT, 2, 3>();
      }
      if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    }
    if constexpr (STAMP) { asm volatile("s_nop 0" : "+v"(s[0]), "+v"(s[1])); asm volatile("" : "+v"(o[0])); }   // the segment's MFMAs have retired
    stamp();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---------------- Y(i): VALU segment
    stamp();
    if (active) {
      float mt_ = -INFINITY;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if constexpr (LAST) {
            const int key = i * 64 + 32 * j + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (key >= Lk) s[j][r] = -INFINITY;
          }
          mt_ = fmaxf(mt_, s[j][r]);
        }
      mt_ = a32_max_halves(mt_);
      const float mn = fmaxf(m, mt_ * c2);
      const float alpha = a32_exp2(m - mn);
      m = mn;
      const float ps = a32_exp_rows(s[0], c2, mn) + a32_exp_rows(s[1], c2, mn);
      l = l * alpha + ps;
      if constexpr (!FIRST) {
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[mt][r] *= alpha;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int c = 0; c < 2; ++c) pf[2 * j + c] = a32_pack8(s[j], c);
    }
    if constexpr (STAMP) { asm volatile("" : "+v"(pf[0]), "+v"(pf[3])); }
    stamp();
    if (grp == 0) A32_WAIT_DMA();
    stamp();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  {
    constexpr std::integral_constant<int, 0> P0{};
    constexpr std::integral_constant<int, 1> P1{};
    constexpr std::true_type T{};
    constexpr std::false_type F{};
    int lastpar;
    if (nt == 1) { xy(0, P0, T, T); lastpar = 0; }
    else {
      xy(0, P0, T, F);
      int i = 1;
      for (; i + 2 < nt; i += 2) { xy(i, P1, F, F); xy(i + 1, P0, F, F); }
      if (nt - i == 2) { xy(i, P1, F, F); xy(i + 1, P0, F, T); lastpar = 0; }
      else { xy(i, P1, F, T); lastpar = 1; }
    }
    // X(nt): the last P V product
    if (active) {
      if (lastpar == 0) { pv(P0); a32_sched_pipeline<4 * C::MT, 2, 3>(); }
      else { pv(P1); a32_sched_pipeline<4 * C::MT, 2, 3>(); }
    }
    if (grp == 0) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }        // pairs with B's delayed start
  }

  if (active) {
    const float lt = a32_sum_halves(l);
    const float inv = 1.0f / lt;
    const bool row_ok = qrow < Lq;
    if (row_ok && hi == 0 && lse) lse[((long)b * H + h) * Lq + qrow] = m * A32_LN2 + logf(lt);
    a32_store_rows<HDP>(o, inv, out + (long)b * ob + (long)qrow * ol + (long)h * oh, row_ok, hd, lane);
  }
}

// measurement aid (ivh_probe_attn32_pingpong 7): the shipped forward at TWO waves per SIMD (256 VGPRs: no spills, a third fewer waves to hide latency)
template <int HDP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2), target("no-packed-fp32-ops"))) void attn32_fwd_np2_kernel(A32_FWD_PARAMS) {
  attn32_fwd_body<HDP, false, false>(A32_FWD_ARGS);
}

#define A32PP_PARAMS                                                                                                                                \
    const bf16_t* __restrict__ q, long qsb, long qsl, long qsh, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, long sb, long sl, long sh, \
    bf16_t* __restrict__ out, long ob, long ol, long oh, float* __restrict__ lse, int H, int Lq, int Lk_max, int hd, float scale,                       \
    const int32_t* __restrict__ kv_len, const int32_t* __restrict__ nb_dev, unsigned long long* __restrict__ stamps = nullptr
#define A32PP_ARGS q, qsb, qsl, qsh, k, v, sb, sl, sh, out, ob, ol, oh, lse, H, Lq, Lk_max, hd, scale, kv_len, nb_dev, stamps
template <int HDP, bool PRIO, bool GSEL = false, bool STAMP = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void attn32pp_fwd_kernel(A32PP_PARAMS) {
  attn32pp_fwd_body<HDP, PRIO, GSEL, STAMP>(A32PP_ARGS);
}
template <int HDP, bool PRIO, bool STAMP = false>      // without packed fp32 instructions (see attn32_fwd_np_kernel)
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2), target("no-packed-fp32-ops"))) void attn32pp_fwd_np_kernel(A32PP_PARAMS) {
  attn32pp_fwd_body<HDP, PRIO, false, STAMP>(A32PP_ARGS);
}

// =========================================================================================================
// dQ for 128 queries per workgroup (32 per wave), looping over the key tiles; also writes delta = <dO, O> per query.
#define A32_DQ_PARAMS                                                                                                                               \
    const bf16_t* __restrict__ q, long qsb, long qsl, long qsh, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, long sb, long sl, long sh, \
    const bf16_t* __restrict__ out, const bf16_t* __restrict__ dout, long ob, long ol, long oh, const float* __restrict__ lse,                        \
    float* __restrict__ delta, bf16_t* __restrict__ dq, long dqb, long dql, long dqh, int H, int Lq, int Lk_max, int hd, float scale,                 \
    const int32_t* __restrict__ kv_len, const int32_t* __restrict__ nb_dev
#define A32_DQ_ARGS q, qsb, qsl, qsh, k, v, sb, sl, sh, out, dout, ob, ol, oh, lse, delta, dq, dqb, dql, dqh, H, Lq, Lk_max, hd, scale, kv_len, nb_dev
template <int HDP>
__device__ __forceinline__ void attn32_bwd_dq_body(A32_DQ_PARAMS) {
  using C = A32<HDP>;
  __shared__ __attribute__((aligned(16))) char lds[4 * C::TILE];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hi = lane >> 5;
  const int npass = (Lq + 127) >> 7;
  // device-side clip count (DropPath skipping): only the first *nb_dev clips exist.  The launch is sized for all of them; the workgroups
  // BEHIND the real work in dispatch order leave at once and the XCD-contiguous remap runs over the real grid -- every XCD loses the same
  // share (dropping the tail of the remapped ids instead would idle one XCD: they are the last XCD's chunk)
  int nwg = gridDim.x;
  if (nb_dev) {
    nwg = min(nwg, npass * H * max(0, *nb_dev));
    if ((int)blockIdx.x >= nwg) return;
  }
  const int wid = xcd_remap(blockIdx.x, nwg);
  const int bh = wid / npass;
  const int b = bh / H, h = bh - b * H;
  const int q0 = (wid - bh * npass) * 128 + wave * 32;
  const int Lk = kv_len ? max(1, min(kv_len[b], Lk_max)) : Lk_max;
  const bf16_t* qb = q + (long)b * qsb + (long)h * qsh;
  const bf16_t* kb = k + (long)b * sb + (long)h * sh;
  const bf16_t* vb = v + (long)b * sb + (long)h * sh;
  const bf16_t* dob = dout + (long)b * ob + (long)h * oh;
  const int range = (int)((((long)Lk - 1) * sl + hd) * 2);
  const u32x4 rs_k = a32_rsrc(kb, range);
  const u32x4 rs_v = a32_rsrc(vb, range);
  unsigned voff[C::RPW];
  a32_dma_offsets<HDP>(lane, wave, sl, hd, voff);
  A32Lane<HDP> ln;
  ln.init(lane);
  const unsigned tstep = (unsigned)(64 * sl * 2);

  a32_dma_tile<HDP>(rs_k, voff, 0u, 0u, wave);
  a32_dma_tile<HDP>(rs_v, voff, 0u, (unsigned)C::TILE, wave);

  const bool active = q0 < Lq;
  const int qrow = q0 + (lane & 31);
  u32x4 qf[C::KS], dof[C::KS];
  a32_row_frags_global<HDP>(qb, qsl, qrow, Lq, hd, qf, lane);
  a32_row_frags_global<HDP>(dob, ol, qrow, Lq, hd, dof, lane);
  const float lse2 = qrow < Lq ? lse[((long)b * H + h) * Lq + qrow] * A32_LOG2E : INFINITY;   // +inf -> P = 0 for padded queries
  // delta[q] = <dO[q], O[q]>: the lane holds half of its query row (the other lane half holds the rest)
  float del = 0.f;
  {
    u32x4 of[C::KS];
    a32_row_frags_global<HDP>(out + (long)b * ob + (long)h * oh, ol, qrow, Lq, hd, of, lane);
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      float a[8], g8[8];
      unpack8(of[ks], a);
      unpack8(dof[ks], g8);
#pragma unroll
      for (int e = 0; e < 8; ++e) del += a[e] * g8[e];
    }
    del = a32_sum_halves(del);
    if (hi == 0 && qrow < Lq) delta[((long)b * H + h) * Lq + qrow] = del;
  }
  f32x16 dqa[C::MT];
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dqa[mt][r] = 0.f;
  const float c2 = scale * A32_LOG2E;
  const int nt = (Lk + 63) >> 6;

#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) asm volatile("" : "+v"(qf[ks]), "+v"(dof[ks]));   // ordinary loads are waited for before the loop
  asm volatile("" : "+v"(del));
  A32_WAIT_DMA();
  __builtin_amdgcn_s_barrier();

  auto tile = [&](const int t, auto par_tag, auto ragged_tag) __attribute__((always_inline)) {
    constexpr bool RAGGED = decltype(ragged_tag)::value;
    constexpr int PAR = decltype(par_tag)::value;
    const char* Kt = lds + PAR * 2 * C::TILE;
    const char* Vt = Kt + C::TILE;
    if (!RAGGED) {
      constexpr unsigned nxt = (unsigned)((PAR ^ 1) * 2 * C::TILE);
      a32_dma_tile<HDP>(rs_k, voff, (unsigned)(t + 1) * tstep, nxt, wave);
      a32_dma_tile<HDP>(rs_v, voff, (unsigned)(t + 1) * tstep, nxt + (unsigned)C::TILE, wave);
    }
    if (active) {
      const a32_f2 c2v = {c2, c2}, lsev = {lse2, lse2}, delv = {del, del};
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
          s = mfma32(a32_row_frag<HDP>(Kt, ln, j, ks), qf[ks], s);
          dp = mfma32(a32_row_frag<HDP>(Vt, ln, j, ks), dof[ks], dp);
        }
        a32_sched_pipeline<2 * C::KS, 1, 6>();
#pragma unroll
        for (int i = 0; i < 8; ++i) {                     // two scores per packed instruction: P = exp2(s c2 - lse2), dS = P (dP - delta)
          a32_f2 e = a32_f2{s[2 * i], s[2 * i + 1]} * c2v - lsev;
          e[0] = a32_exp2(e[0]); e[1] = a32_exp2(e[1]);
          if constexpr (RAGGED) {
            const int key = t * 64 + 32 * j + ((2 * i) & 3) + 8 * ((2 * i) >> 2) + 4 * hi;
            if (key >= Lk) e[0] = 0.f;
            if (key + 1 >= Lk) e[1] = 0.f;
          }
          const a32_f2 d = e * (a32_f2{dp[2 * i], dp[2 * i + 1]} - delv);
          s[2 * i] = d[0]; s[2 * i + 1] = d[1];
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const u32x4 dsf = a32_pack8(s, c);
#pragma unroll
          for (int mt = 0; mt < C::MT; ++mt) dqa[mt] = mfma32(a32_tr_frag<HDP>(Kt, ln, j, c, mt), dsf, dqa[mt]);
        }
        a32_sched_pipeline<2 * C::MT, 2, 3>();
      }
    }
    A32_WAIT_DMA();
    __builtin_amdgcn_s_barrier();
  };
  {
    constexpr std::integral_constant<int, 0> P0{};
    constexpr std::integral_constant<int, 1> P1{};
    int t = 0;
    for (; t + 2 < nt; t += 2) { tile(t, P0, std::false_type{}); tile(t + 1, P1, std::false_type{}); }
    if (nt - t == 2) { tile(t, P0, std::false_type{}); tile(t + 1, P1, std::true_type{}); }
    else tile(t, P0, std::true_type{});
  }

  if (active)
    a32_store_rows<HDP>(dqa, scale, dq + (long)b * dqb + (long)qrow * dql + (long)h * dqh, qrow < Lq, hd, lane);
}

// NP: compiled without the packed fp32 instructions (see attn32_fwd_np_kernel)
template <int HDP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void attn32_bwd_dq_kernel(A32_DQ_PARAMS = nullptr) { attn32_bwd_dq_body<HDP>(A32_DQ_ARGS); }
template <int HDP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2), target("no-packed-fp32-ops"))) void attn32_bwd_dq_np_kernel(A32_DQ_PARAMS = nullptr) {
  attn32_bwd_dq_body<HDP>(A32_DQ_ARGS);
}

// =========================================================================================================
// dK, dV for 128 keys per workgroup (32 per wave), looping over 64-query tiles of Q and dO (LDS-DMA, double buffered).  The
// per-query statistics (lse * log2 e, delta) of the whole sequence are staged in LDS once, before the loop (+inf / 0 for padded
// queries -> P = 0 there): the loop itself contains no ordinary global load, so hipcc has no reason to touch vmcnt inside it.
// Dynamic LDS: 4 tiles + 2 * 64 * ceil(Lq / 64) floats.
#define A32_DKDV_PARAMS                                                                                                                             \
    const bf16_t* __restrict__ q, long qsb, long qsl, long qsh, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, long sb, long sl, long sh, \
    const bf16_t* __restrict__ dout, long ob, long ol, long oh, const float* __restrict__ lse, const float* __restrict__ delta,                       \
    bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, long dsb, long dsl, long dsh,                                                                    \
    int H, int Lq, int Lk, int hd, float scale, const int32_t* __restrict__ kv_len, const int32_t* __restrict__ nb_dev
#define A32_DKDV_ARGS q, qsb, qsl, qsh, k, v, sb, sl, sh, dout, ob, ol, oh, lse, delta, dk, dv, dsb, dsl, dsh, H, Lq, Lk, hd, scale, kv_len, nb_dev
template <int HDP>
__device__ __forceinline__ void attn32_bwd_dkdv_body(A32_DKDV_PARAMS) {
  using C = A32<HDP>;
  constexpr int BUF = 2 * C::TILE;                                            // Q tile, dO tile
  extern __shared__ __attribute__((aligned(16))) char lds[];                  // [2 * BUF] tiles, then lse2[nt * 64], delta[nt * 64]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hi = lane >> 5;
  const int npass = (Lk + 127) >> 7;
  int nwg = gridDim.x;                                                        // (see attn32_fwd_kernel)
  if (nb_dev) {
    nwg = min(nwg, npass * H * max(0, *nb_dev));
    if ((int)blockIdx.x >= nwg) return;
  }
  const int wid = xcd_remap(blockIdx.x, nwg);
  const int bh = wid / npass;
  const int b = bh / H, h = bh - b * H;
  const int k0 = (wid - bh * npass) * 128 + wave * 32;
  const int Lk_b = kv_len ? max(1, min(kv_len[b], Lk)) : Lk;                  // keys >= Lk_b are padding: their dK / dV rows are written as zeros
  const bf16_t* qb = q + (long)b * qsb + (long)h * qsh;
  const bf16_t* kb = k + (long)b * sb + (long)h * sh;
  const bf16_t* vb = v + (long)b * sb + (long)h * sh;
  const bf16_t* dob = dout + (long)b * ob + (long)h * oh;
  const float* lseb = lse + ((long)b * H + h) * Lq;
  const float* delb = delta + ((long)b * H + h) * Lq;
  const u32x4 rs_q = a32_rsrc(qb, (int)((((long)Lq - 1) * qsl + hd) * 2));
  const u32x4 rs_do = a32_rsrc(dob, (int)((((long)Lq - 1) * ol + hd) * 2));
  unsigned voff_q[C::RPW], voff_do[C::RPW];
  a32_dma_offsets<HDP>(lane, wave, qsl, hd, voff_q);
  a32_dma_offsets<HDP>(lane, wave, ol, hd, voff_do);
  A32Lane<HDP> ln;
  ln.init(lane);
  const unsigned tstep_q = (unsigned)(64 * qsl * 2), tstep_do = (unsigned)(64 * ol * 2);
  auto issue = [&](int t, unsigned buf) {
    a32_dma_tile<HDP>(rs_q, voff_q, (unsigned)t * tstep_q, buf, wave);
    a32_dma_tile<HDP>(rs_do, voff_do, (unsigned)t * tstep_do, buf + (unsigned)C::TILE, wave);
  };
  issue(0, 0u);

  const int nt = (Lq + 63) >> 6;
  float* lse_s = reinterpret_cast<float*>(lds + 2 * BUF);
  float* del_s = lse_s + nt * 64;
  for (int i = threadIdx.x; i < nt * 64; i += 256) {
    lse_s[i] = i < Lq ? lseb[i] * A32_LOG2E : INFINITY;
    del_s[i] = i < Lq ? delb[i] : 0.f;
  }
  const bool active = k0 < Lk;
  const int key = k0 + (lane & 31);
  u32x4 kf[C::KS], vf[C::KS];
  a32_row_frags_global<HDP>(kb, sl, key, Lk_b, hd, kf, lane);
  a32_row_frags_global<HDP>(vb, sl, key, Lk_b, hd, vf, lane);
  f32x16 dka[C::MT], dva[C::MT];
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dka[mt][r] = 0.f; dva[mt][r] = 0.f; }
  const float c2 = scale * A32_LOG2E;

#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) asm volatile("" : "+v"(kf[ks]), "+v"(vf[ks]));   // ordinary loads are waited for before the loop
  A32_WAIT_DMA();
  __syncthreads();

  auto tile = [&](const int t, auto par_tag) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_tag)::value;
    const char* Qt = lds + PAR * BUF;
    const char* Dt = Qt + C::TILE;
    if (t + 1 < nt) issue(t + 1, (unsigned)((PAR ^ 1) * BUF));
    if (active) {
      const a32_f2 c2v = {c2, c2};
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        // S and dP: rows = queries 32 j + (r & 3) + 8 (r >> 2) + 4 hi, col = this lane's key
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
          s = mfma32(a32_row_frag<HDP>(Qt, ln, j, ks), kf[ks], s);
          dp = mfma32(a32_row_frag<HDP>(Dt, ln, j, ks), vf[ks], dp);
        }
        a32_sched_pipeline<2 * C::KS, 1, 6>();
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const f32x4 lv = *reinterpret_cast<const f32x4*>(lse_s + t * 64 + 32 * j + 8 * g4 + 4 * hi);
          const f32x4 dl = *reinterpret_cast<const f32x4*>(del_s + t * 64 + 32 * j + 8 * g4 + 4 * hi);
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {                // two scores per packed instruction
            const int r = 4 * g4 + 2 * h2;
            a32_f2 e = a32_f2{s[r], s[r + 1]} * c2v - a32_f2{lv[2 * h2], lv[2 * h2 + 1]};
            e[0] = a32_exp2(e[0]); e[1] = a32_exp2(e[1]);
            const a32_f2 d = e * (a32_f2{dp[r], dp[r + 1]} - a32_f2{dl[2 * h2], dl[2 * h2 + 1]});
            s[r] = e[0]; s[r + 1] = e[1];
            dp[r] = d[0]; dp[r + 1] = d[1];
          }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const u32x4 pf = a32_pack8(s, c);
          const u32x4 dsf = a32_pack8(dp, c);
#pragma unroll
          for (int mt = 0; mt < C::MT; ++mt) {
            dva[mt] = mfma32(a32_tr_frag<HDP>(Dt, ln, j, c, mt), pf, dva[mt]);
            dka[mt] = mfma32(a32_tr_frag<HDP>(Qt, ln, j, c, mt), dsf, dka[mt]);
          }
        }
        a32_sched_pipeline<4 * C::MT, 2, 3>();
      }
    }
    A32_WAIT_DMA();
    __syncthreads();
  };
  {
    constexpr std::integral_constant<int, 0> P0{};
    constexpr std::integral_constant<int, 1> P1{};
    int t = 0;
    for (; t + 1 < nt; t += 2) { tile(t, P0); tile(t + 1, P1); }
    if (t < nt) tile(t, P0);
  }
  if (active) {
    const float live = key < Lk_b ? 1.0f : 0.0f;
    const bool row_ok = key < Lk;
    a32_store_rows<HDP>(dka, scale * live, dk + (long)b * dsb + (long)key * dsl + (long)h * dsh, row_ok, hd, lane);
    a32_store_rows<HDP>(dva, live, dv + (long)b * dsb + (long)key * dsl + (long)h * dsh, row_ok, hd, lane);
  }
}
template <int HDP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void attn32_bwd_dkdv_kernel(A32_DKDV_PARAMS = nullptr) { attn32_bwd_dkdv_body<HDP>(A32_DKDV_ARGS); }
template <int HDP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2), target("no-packed-fp32-ops"))) void attn32_bwd_dkdv_np_kernel(A32_DKDV_PARAMS = nullptr) {
  attn32_bwd_dkdv_body<HDP>(A32_DKDV_ARGS);
}

}  // namespace ivh

using namespace ivh;

// Can the 32x32 kernels take this problem?  16-byte stores / DMA chunks: every stride a multiple of 8 elements, every (b, h)
// slice addressable with 31-bit byte offsets.
extern "C" int ivh_attn32_supported(int64_t qsb, int64_t qsl, int64_t qsh, int64_t sb, int64_t sl, int64_t sh,
                                    int64_t ob, int64_t ol, int64_t oh, int Lq, int Lk, int hd) {
  (void)qsb; (void)sb; (void)ob;
  if (hd % 8 || hd > 128 || hd <= 0) return 0;
  if ((qsl % 8) || (qsh % 8) || (sl % 8) || (sh % 8) || (ol % 8) || (oh % 8) || (qsb % 8) || (sb % 8) || (ob % 8)) return 0;
  const int64_t lim = (1LL << 31) - (1 << 20);
  if (((int64_t)Lk + 64) * sl * 2 >= lim || ((int64_t)Lq + 64) * qsl * 2 >= lim || ((int64_t)Lq + 64) * ol * 2 >= lim) return 0;
  return 1;
}

#define IVH_ATTN32_DISPATCH(hd, KERNEL, grid, s, ...)                                                   \
  if ((hd) <= 64) hipLaunchKernelGGL((KERNEL<64>), grid, dim3(256), 0, s, __VA_ARGS__);                 \
  else if ((hd) <= 96) hipLaunchKernelGGL((KERNEL<96>), grid, dim3(256), 0, s, __VA_ARGS__);            \
  else hipLaunchKernelGGL((KERNEL<128>), grid, dim3(256), 0, s, __VA_ARGS__);

// measurement aid: a device buffer of [rows][4] uint64 receives wave 0's shader-clock stamps (entry, loop start, loop end, exit) of every
// forward workgroup launched while it is set; NULL switches it off (the default).  Not thread-safe: a debugging facility.
static unsigned long long* g_a32_stamps = nullptr;
static long g_a32_stamp_rows = 0;
extern "C" int ivh_attn32_debug_stamps(void* buf, int64_t rows) {
  g_a32_stamps = reinterpret_cast<unsigned long long*>(buf);
  g_a32_stamp_rows = buf ? (long)rows : 0;
  return 0;
}

// unpacked fp32 arithmetic in the three 32x32 attention kernels: -1 = read IVH_ATTN_NOPK once (default on), 0 / 1 = ivh_probe_attn32_unpacked
static int g_a32_np = -1;
static bool a32_np() {
  if (g_a32_np < 0) { const char* e = getenv("IVH_ATTN_NOPK"); g_a32_np = (e && e[0] == '0') ? 0 : 1; }
  return g_a32_np == 1;
}
extern "C" int ivh_probe_attn32_unpacked(int on) { g_a32_np = on ? 1 : 0; return 0; }

// measurement switch (internvideo_hip_debug.h): 0 = the one-group forward kernel, 1 / 2 = attn32pp_fwd_kernel; -1 = read IVH_ATTN_PP once
static int g_a32_pingpong = -1;
extern "C" int ivh_probe_attn32_pingpong(int mode) {
  IVH_REQUIRE(mode >= 0 && mode <= 7, "ivh_probe_attn32_pingpong: 0 (off), 1 (two wave groups), 2 (+ raised priority), 3 (groups = even / odd waves), 4 / 5 (1 / 2 with unpacked softmax), 6 (one group, unpacked), 7 (6 at two waves per SIMD)");
  g_a32_pingpong = mode;
  return 0;
}

extern "C" int ivh_attn32_fwd_launch(const uint16_t* q, int64_t qsb, int64_t qsl, int64_t qsh,
                                     const uint16_t* k, const uint16_t* v, int64_t sb, int64_t sl, int64_t sh,
                                     uint16_t* out, int64_t ob, int64_t ol, int64_t oh, float* lse,
                                     int B, int H, int Lq, int Lk, int hd, float scale, const int32_t* kv_len, const int32_t* nb_dev, void* stream) {
  IVH_REQUIRE(((uintptr_t)out % 16) == 0, "flash_attn_fwd: out must be 16-byte aligned");
  if (g_a32_pingpong < 0) { const char* e = getenv("IVH_ATTN_PP"); g_a32_pingpong = e ? atoi(e) : 0; }
  IVH_REQUIRE(!g_a32_stamps || g_a32_pingpong > 0 || (long)((Lq + 127) / 128) * H * B <= g_a32_stamp_rows, "flash_attn_fwd: the stamp buffer holds %ld workgroups", g_a32_stamp_rows);
  static int defer = -1;
  if (defer < 0) { const char* e = getenv("IVH_ATTN_DEFER"); defer = (e && e[0] == '1') ? 1 : 0; }
  hipStream_t s = (hipStream_t)stream;
  if (g_a32_pingpong > 0 && g_a32_pingpong < 6 && g_a32_stamps && hd > 64 && hd <= 96) {     // segment stamps of workgroup 0, waves 0 and 4: [2][256] (tools/probes/attn_pp_stamps.py)
    IVH_REQUIRE(g_a32_stamp_rows >= 128, "flash_attn_fwd: the two-group stamp buffer holds 2 x 256 uint64");
    dim3 grid2((unsigned)((long)((Lq + 255) / 256) * H * B), 1, 1);
    if (g_a32_pingpong >= 4)
      hipLaunchKernelGGL((attn32pp_fwd_np_kernel<96, false, true>), grid2, dim3(512), 0, s, q, (long)qsb, (long)qsl, (long)qsh, k, v, (long)sb, (long)sl,
                         (long)sh, out, (long)ob, (long)ol, (long)oh, lse, H, Lq, Lk, hd, scale, kv_len, nb_dev, g_a32_stamps);
    else
    hipLaunchKernelGGL((attn32pp_fwd_kernel<96, false, false, true>), grid2, dim3(512), 0, s, q, (long)qsb, (long)qsl, (long)qsh, k, v, (long)sb, (long)sl,
                       (long)sh, out, (long)ob, (long)ol, (long)oh, lse, H, Lq, Lk, hd, scale, kv_len, nb_dev, g_a32_stamps);
    return ivh_host::check_launch("flash_attn_fwd (32x32, two wave groups, stamps)");
  }
  if (g_a32_pingpong == 7 && !g_a32_stamps && hd > 64 && hd <= 96) {                  // the shipped kernel at two waves per SIMD
    dim3 grid1((unsigned)((long)((Lq + 127) / 128) * H * B), 1, 1);
    hipLaunchKernelGGL((attn32_fwd_np2_kernel<96>), grid1, dim3(256), 0, s, q, (long)qsb, (long)qsl, (long)qsh, k, v, (long)sb, (long)sl,
                       (long)sh, out, (long)ob, (long)ol, (long)oh, lse, H, Lq, Lk, hd, scale, kv_len, (unsigned long long*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, nb_dev);
    return ivh_host::check_launch("flash_attn_fwd (32x32, two waves per SIMD)");
  }
  if (g_a32_pingpong == 6 && !g_a32_stamps && hd > 64 && hd <= 96) {                  // the one-group kernel with unpacked softmax arithmetic
    dim3 grid1((unsigned)((long)((Lq + 127) / 128) * H * B), 1, 1);
    hipLaunchKernelGGL((attn32_fwd_np_kernel<96>), grid1, dim3(256), 0, s, q, (long)qsb, (long)qsl, (long)qsh, k, v, (long)sb, (long)sl,
                       (long)sh, out, (long)ob, (long)ol, (long)oh, lse, H, Lq, Lk, hd, scale, kv_len, (unsigned long long*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, nb_dev);
    return ivh_host::check_launch("flash_attn_fwd (32x32, unpacked softmax)");
  }
  if (g_a32_pingpong > 0 && g_a32_pingpong < 6 && !g_a32_stamps) {                   // two wave groups one phase apart (1: plain, 2: MFMA segments at raised priority)
    dim3 grid2((unsigned)((long)((Lq + 255) / 256) * H * B), 1, 1);
#define IVH_A32_PP(HDP, PR) hipLaunchKernelGGL((attn32pp_fwd_kernel<HDP, PR>), grid2, dim3(512), 0, s, q, (long)qsb, (long)qsl, (long)qsh, k, v, (long)sb, (long)sl, \
                                               (long)sh, out, (long)ob, (long)ol, (long)oh, lse, H, Lq, Lk, hd, scale, kv_len, nb_dev)
    const bool pr = g_a32_pingpong == 2;
    if ((g_a32_pingpong == 4 || g_a32_pingpong == 5) && hd > 64 && hd <= 96) {      // unpacked softmax arithmetic (5: + raised priority)
      if (g_a32_pingpong == 4)
        hipLaunchKernelGGL((attn32pp_fwd_np_kernel<96, false>), grid2, dim3(512), 0, s, q, (long)qsb, (long)qsl, (long)qsh, k, v, (long)sb, (long)sl,
                           (long)sh, out, (long)ob, (long)ol, (long)oh, lse, H, Lq, Lk, hd, scale, kv_len, nb_dev);
      else
        hipLaunchKernelGGL((attn32pp_fwd_np_kernel<96, true>), grid2, dim3(512), 0, s, q, (long)qsb, (long)qsl, (long)qsh, k, v, (long)sb, (long)sl,
                           (long)sh, out, (long)ob, (long)ol, (long)oh, lse, H, Lq, Lk, hd, scale, kv_len, nb_dev);
      return ivh_host::check_launch("flash_attn_fwd (32x32, two wave groups, unpacked softmax)");
    }
    if (g_a32_pingpong == 3 && hd > 64 && hd <= 96) {
      hipLaunchKernelGGL((attn32pp_fwd_kernel<96, false, true>), grid2, dim3(512), 0, s, q, (long)qsb, (long)qsl, (long)qsh, k, v, (long)sb, (long)sl,
                         (long)sh, out, (long)ob, (long)ol, (long)oh, lse, H, Lq, Lk, hd, scale, kv_len, nb_dev);
      return ivh_host::check_launch("flash_attn_fwd (32x32, two wave groups, alternating)");
    }
    if (hd <= 64) { if (pr) IVH_A32_PP(64, true); else IVH_A32_PP(64, false); }
    else if (hd <= 96) { if (pr) IVH_A32_PP(96, true); else IVH_A32_PP(96, false); }
    else { if (pr) IVH_A32_PP(128, true); else IVH_A32_PP(128, false); }
#undef IVH_A32_PP
    return ivh_host::check_launch("flash_attn_fwd (32x32, two wave groups)");
  }
  dim3 grid((unsigned)((long)((Lq + 127) / 128) * H * B), 1, 1);
#define IVH_A32_FWD(HDP, DF) hipLaunchKernelGGL((attn32_fwd_kernel<HDP, DF>), grid, dim3(256), 0, s, q, (long)qsb, (long)qsl, (long)qsh, k, v, (long)sb, (long)sl, \
                                                (long)sh, out, (long)ob, (long)ol, (long)oh, lse, H, Lq, Lk, hd, scale, kv_len, g_a32_stamps, \
                                                (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, nb_dev)
#define IVH_A32_FWD_NP(HDP) hipLaunchKernelGGL((attn32_fwd_np_kernel<HDP>), grid, dim3(256), 0, s, q, (long)qsb, (long)qsl, (long)qsh, k, v, (long)sb, (long)sl, \
                                               (long)sh, out, (long)ob, (long)ol, (long)oh, lse, H, Lq, Lk, hd, scale, kv_len, g_a32_stamps, \
                                               (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, nb_dev)
  if (defer && a32_np() && hd > 64 && hd <= 96) {              // measurement aid: deferred rescale without packed fp32 (IVH_ATTN_DEFER=1)
    hipLaunchKernelGGL((attn32_fwd_np_kernel<96, true>), grid, dim3(256), 0, s, q, (long)qsb, (long)qsl, (long)qsh, k, v, (long)sb, (long)sl,
                       (long)sh, out, (long)ob, (long)ol, (long)oh, lse, H, Lq, Lk, hd, scale, kv_len, g_a32_stamps,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, nb_dev);
    return ivh_host::check_launch("flash_attn_fwd (32x32, deferred rescale)");
  }
  const bool np = a32_np() && !defer;
  if (hd <= 64) { if (np) IVH_A32_FWD_NP(64); else if (defer) IVH_A32_FWD(64, true); else IVH_A32_FWD(64, false); }
  else if (hd <= 96) { if (np) IVH_A32_FWD_NP(96); else if (defer) IVH_A32_FWD(96, true); else IVH_A32_FWD(96, false); }
  else { if (np) IVH_A32_FWD_NP(128); else if (defer) IVH_A32_FWD(128, true); else IVH_A32_FWD(128, false); }
#undef IVH_A32_FWD
#undef IVH_A32_FWD_NP
  return ivh_host::check_launch("flash_attn_fwd (32x32)");
}

// round-5 prototype (internvideo_hip_debug.h): the forward kernel with the q/k RMSNorm applied on the fly -- q, k un-normalised, rq / rk fp32 [B * L]
// (rstd over the concatenated heads), wqk fp32 [H * hd] = q_norm.weight * k_norm.weight.  hd <= 96, Lq == Lk <= 512, no kv_len.
extern "C" int ivh_probe_attn32_fwd_qkn(const uint16_t* q, int64_t qsb, int64_t qsl, int64_t qsh,
                                        const uint16_t* k, const uint16_t* v, int64_t sb, int64_t sl, int64_t sh,
                                        uint16_t* out, int64_t ob, int64_t ol, int64_t oh, float* lse,
                                        int B, int H, int Lq, int Lk, int hd, float scale, const float* rq, const float* rk, const float* wqk, void* stream) {
  IVH_REQUIRE(rq && rk && wqk && hd > 64 && hd <= 96 && hd % 8 == 0 && Lk <= 512 && Lq == Lk, "probe_attn32_fwd_qkn: prototype for 64 < hd <= 96, Lq == Lk <= 512");
  IVH_REQUIRE(((uintptr_t)out % 16) == 0 && ((uintptr_t)wqk % 16) == 0 && ivh_attn32_supported(qsb, qsl, qsh, sb, sl, sh, ob, ol, oh, Lq, Lk, hd), "probe_attn32_fwd_qkn: layout");
  dim3 grid((unsigned)((long)((Lq + 127) / 128) * H * B), 1, 1);
  hipLaunchKernelGGL((attn32_fwd_kernel<96, false, true>), grid, dim3(256), 0, (hipStream_t)stream, q, (long)qsb, (long)qsl, (long)qsh, k, v, (long)sb, (long)sl,
                     (long)sh, out, (long)ob, (long)ol, (long)oh, lse, H, Lq, Lk, hd, scale, (const int32_t*)nullptr, (unsigned long long*)nullptr, rq, rk, wqk);
  return ivh_host::check_launch("probe_attn32_fwd_qkn");
}

// dQ (+ delta) part of the backward
extern "C" int ivh_attn32_bwd_dq_launch(const uint16_t* q, int64_t qsb, int64_t qsl, int64_t qsh,
                                        const uint16_t* k, const uint16_t* v, int64_t sb, int64_t sl, int64_t sh,
                                        const uint16_t* out, const uint16_t* dout, int64_t ob, int64_t ol, int64_t oh,
                                        const float* lse, float* delta, uint16_t* dq, int64_t dqb, int64_t dql, int64_t dqh,
                                        int B, int H, int Lq, int Lk, int hd, float scale, const int32_t* kv_len, const int32_t* nb_dev, void* stream) {
  IVH_REQUIRE(((uintptr_t)dq % 16) == 0 && dqb % 8 == 0 && dql % 8 == 0 && dqh % 8 == 0, "flash_attn_bwd: dq must be 16-byte aligned with strides that are multiples of 8");
  dim3 gq((unsigned)((long)((Lq + 127) / 128) * H * B), 1, 1);
  if (a32_np()) {
    IVH_ATTN32_DISPATCH(hd, attn32_bwd_dq_np_kernel, gq, (hipStream_t)stream, q, (long)qsb, (long)qsl, (long)qsh, k, v, (long)sb, (long)sl, (long)sh, out, dout,
                        (long)ob, (long)ol, (long)oh, lse, delta, dq, (long)dqb, (long)dql, (long)dqh, H, Lq, Lk, hd, scale, kv_len, nb_dev);
  } else {
    IVH_ATTN32_DISPATCH(hd, attn32_bwd_dq_kernel, gq, (hipStream_t)stream, q, (long)qsb, (long)qsl, (long)qsh, k, v, (long)sb, (long)sl, (long)sh, out, dout,
                        (long)ob, (long)ol, (long)oh, lse, delta, dq, (long)dqb, (long)dql, (long)dqh, H, Lq, Lk, hd, scale, kv_len, nb_dev);
  }
  return ivh_host::check_launch("flash_attn_bwd dq (32x32)");
}

// dynamic LDS of the dK/dV kernel; 0 = this problem stays on the 16x16 kernel (head dims above 96 spill there; very long sequences)
extern "C" int ivh_attn32_dkdv_lds_bytes(int Lq, int hd) {
  if (hd > 96) return 0;
  const int hdp = hd <= 64 ? 64 : 96;
  const long bytes = 4L * 64 * hdp * 2 + (long)((Lq + 63) / 64) * 64 * 8;
  return bytes <= 80 * 1024 ? (int)bytes : 0;            // two workgroups per CU
}

extern "C" int ivh_attn32_bwd_dkdv_launch(const uint16_t* q, int64_t qsb, int64_t qsl, int64_t qsh,
                                          const uint16_t* k, const uint16_t* v, int64_t sb, int64_t sl, int64_t sh,
                                          const uint16_t* dout, int64_t ob, int64_t ol, int64_t oh, const float* lse, const float* delta,
                                          uint16_t* dk, uint16_t* dv, int64_t dsb, int64_t dsl, int64_t dsh,
                                          int B, int H, int Lq, int Lk, int hd, float scale, const int32_t* kv_len, const int32_t* nb_dev, void* stream) {
  const int lds_bytes = ivh_attn32_dkdv_lds_bytes(Lq, hd);
  IVH_REQUIRE(lds_bytes > 0, "flash_attn_bwd dkdv (32x32): unsupported head dim / sequence length");
  IVH_REQUIRE(((uintptr_t)dk % 16) == 0 && ((uintptr_t)dv % 16) == 0 && dsb % 8 == 0 && dsl % 8 == 0 && dsh % 8 == 0,
              "flash_attn_bwd: dk / dv must be 16-byte aligned with strides that are multiples of 8");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)attn32_bwd_dkdv_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)attn32_bwd_dkdv_kernel<96>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)attn32_bwd_dkdv_np_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)attn32_bwd_dkdv_np_kernel<96>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    attr_set = true;
  }
  dim3 gk((unsigned)((long)((Lk + 127) / 128) * H * B), 1, 1);
  hipStream_t s = (hipStream_t)stream;
#define IVH_A32_DKDV(KERNEL) hipLaunchKernelGGL((KERNEL), gk, dim3(256), lds_bytes, s, q, (long)qsb, (long)qsl, (long)qsh, k, v, (long)sb, (long)sl, (long)sh, dout, \
                                                (long)ob, (long)ol, (long)oh, lse, delta, dk, dv, (long)dsb, (long)dsl, (long)dsh, H, Lq, Lk, hd, scale, kv_len, nb_dev)
  if (a32_np()) { if (hd <= 64) IVH_A32_DKDV(attn32_bwd_dkdv_np_kernel<64>); else IVH_A32_DKDV(attn32_bwd_dkdv_np_kernel<96>); }
  else { if (hd <= 64) IVH_A32_DKDV(attn32_bwd_dkdv_kernel<64>); else IVH_A32_DKDV(attn32_bwd_dkdv_kernel<96>); }
#undef IVH_A32_DKDV
  return ivh_host::check_launch("flash_attn_bwd dkdv (32x32)");
}
