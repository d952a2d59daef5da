// Attention-forward laboratory (measurement aid, NOT part of the library): a standalone hipcc program that includes the shipped
// flash_attn32.hip, runs its forward kernel and an experimental software-pipelined variant on the 1B step's shape, compares the outputs and
// times both.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -w -o /tmp/attn_fwd_lab tools/probes/attn_fwd_lab.hip && /tmp/attn_fwd_lab
//
// Why: tools/probes/mfma_valu_overlap.hip shows that a SIMD runs an MFMA stream and a VALU stream in 0.68 of the sum of their times, also
// when both are interleaved inside ONE wave.  The shipped forward kernel runs QK^T (MFMA), softmax (VALU) and PV (MFMA) of a key tile one
// after the other inside a wave and relies on the other waves of the SIMD for overlap (measured: ~2080 cycles per wave and 64-key tile
// against 768 matrix + ~990 VALU issue cycles).  The variant computes S(t + 1) = K(t + 1) Q^T WHILE it runs the softmax of tile t: the
// 12 QK^T MFMAs are interleaved with the ~175 softmax VALU instructions by sched_group_barrier, PV(t) follows.  K needs 2 LDS buffers,
// V needs 3 (V(t) is read while K / V (t + 2) are in flight and V(t + 1) waits): 60 KiB per workgroup, 2 workgroups per CU.
// The lab fixes the tile count at compile time (NT = ceil(L / 64)) so that every LDS offset stays an immediate; a production version
// would unroll the period-6 buffer rotation instead.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <utility>
#include <vector>

#include "../../internvideo_amd/csrc/flash_attn32.hip"

namespace ivh_host {
void set_error(const char*, ...) {}
int check_launch(const char*) { return hipGetLastError() != hipSuccess ? -1 : 0; }
}  // namespace ivh_host

namespace ivh {

template <int HDP, int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void attn32_fwd_lab_kernel(
    const bf16_t* __restrict__ q, long qsb, long qsl, long qsh, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, long sb, long sl, long sh,
    bf16_t* __restrict__ out, long ob, long ol, long oh, float* __restrict__ lse, int H, int Lq, int Lk, int hd, float scale) {
  using C = A32<HDP>;
  __shared__ __attribute__((aligned(16))) char lds[5 * C::TILE];            // K buffers 0 / 1, V buffers 2 / 3 / 4
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hi = lane >> 5;
  const int npass = (Lq + 127) >> 7;
  const int wid = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = wid / npass;
  const int b = bh / H, h = bh - b * H;
  const int q0 = (wid - bh * npass) * 128 + wave * 32;
  const bf16_t* qb = q + (long)b * qsb + (long)h * qsh;
  const bf16_t* kb = k + (long)b * sb + (long)h * sh;
  const bf16_t* vb = v + (long)b * sb + (long)h * sh;
  const int range = (int)((((long)Lk - 1) * sl + hd) * 2);
  const u32x4 rs_k = a32_rsrc(kb, range);
  const u32x4 rs_v = a32_rsrc(vb, range);
  unsigned voff[C::RPW];
  a32_dma_offsets<HDP>(lane, wave, sl, hd, voff);
  A32Lane<HDP> ln;
  ln.init(lane);
  const unsigned tstep = (unsigned)(64 * sl * 2);

  // tiles 0 and 1 (a tile past the end reads zeros through the descriptor's bounds check)
  a32_dma_tile<HDP>(rs_k, voff, 0u, 0u, wave);
  a32_dma_tile<HDP>(rs_v, voff, 0u, (unsigned)(2 * C::TILE), wave);
  a32_dma_tile<HDP>(rs_k, voff, tstep, (unsigned)C::TILE, wave);
  a32_dma_tile<HDP>(rs_v, voff, tstep, (unsigned)(3 * C::TILE), wave);

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

#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) asm volatile("" : "+v"(qf[ks]));
  A32_WAIT_DMA();
  __builtin_amdgcn_s_barrier();

  f32x16 sa[2], sb_[2];                                  // scores of the even / odd tiles
  // S(0), not overlapped with anything
  if (active) {
    const char* Kt = lds;
    u32x4 kfr[2 * C::KS];
#pragma unroll
    for (int i = 0; i < 2 * C::KS; ++i) kfr[i] = a32_row_frag<HDP>(Kt, ln, i & 1, i >> 1);
#pragma unroll
    for (int r = 0; r < 16; ++r) { sa[0][r] = 0.f; sa[1][r] = 0.f; }
#pragma unroll
    for (int i = 0; i < 2 * C::KS; ++i) sa[i & 1] = mfma32(kfr[i], qf[i >> 1], sa[i & 1]);
    a32_sched_pipeline<2 * C::KS, 1, 4>();
  }
  __builtin_amdgcn_s_barrier();                          // iteration 0 overwrites K(0)'s buffer: every wave must be done with S(0)

  auto tile = [&](auto t_tag) __attribute__((always_inline)) {
    constexpr int t = decltype(t_tag)::value;
    constexpr bool LAST = t == NT - 1;
    constexpr int KNEXT = ((t + 1) & 1) * C::TILE;                       // K(t + 1)
    constexpr int VCUR = (2 + t % 3) * C::TILE;                          // V(t)
    f32x16 (&s)[2] = (t & 1) ? sb_ : sa;                                 // S(t), computed one iteration ago
    f32x16 (&sn)[2] = (t & 1) ? sa : sb_;                                // S(t + 1), computed now
    if constexpr (t + 2 < NT) {                                          // K(t) and V(t - 1) are dead since the last barrier
      a32_dma_tile<HDP>(rs_k, voff, (unsigned)(t + 2) * tstep, (unsigned)((t & 1) * C::TILE), wave);
      a32_dma_tile<HDP>(rs_v, voff, (unsigned)(t + 2) * tstep, (unsigned)((2 + (t + 2) % 3) * C::TILE), wave);
    }
    if (active) {
      // ---- S(t + 1) = K(t + 1) Q^T: 12 MFMAs, issued in program order BEFORE the softmax of tile t and spread over it by the directives below
      if constexpr (!LAST) {
        const char* Kt = lds + KNEXT;
        u32x4 kfr[2 * C::KS];
#pragma unroll
        for (int i = 0; i < 2 * C::KS; ++i) kfr[i] = a32_row_frag<HDP>(Kt, ln, i & 1, i >> 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) { sn[0][r] = 0.f; sn[1][r] = 0.f; }
#pragma unroll
        for (int i = 0; i < 2 * C::KS; ++i) sn[i & 1] = mfma32(kfr[i], qf[i >> 1], sn[i & 1]);
      }
      // ---- softmax of tile t
      float mt_ = -INFINITY;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if constexpr (LAST) {
            const int key = t * 64 + 32 * j + (r & 3) + 8 * (r >> 2) + 4 * hi;
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
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[mt][r] *= alpha;
      if constexpr (!LAST) {
        // 4 fragment reads ahead, then per MFMA: 1 MFMA, 1 read (while any are left), ~13 VALU
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int i = 0; i < 2 * C::KS; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (i < 2 * C::KS - 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 13, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- O^T += V(t)^T P(t)^T
      const char* Vt = lds + VCUR;
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
    A32_WAIT_DMA();
    __builtin_amdgcn_s_barrier();
  };
  [&]<int... T>(std::integer_sequence<int, T...>) { (tile(std::integral_constant<int, T>{}), ...); }(std::make_integer_sequence<int, NT>{});

  if (active) {
    const float lt = a32_sum_halves(l);
    const float inv = 1.0f / lt;
    const bool row_ok = qrow < Lq;
    if (row_ok && hi == 0 && lse) lse[((long)b * H + h) * Lq + qrow] = m * A32_LN2 + logf(lt);
    a32_store_rows<HDP>(o, inv, out + (long)b * ob + (long)qrow * ol + (long)h * oh, row_ok, hd, lane);
  }
}

// 64 queries per wave: two Q fragment sets, every K / V^T fragment read from LDS feeds TWO MFMAs (half the LDS bytes per FLOP of the shipped
// kernel).  Workgroup = 4 waves = 256 queries; otherwise the shipped structure (double-buffered K / V tiles, one barrier per tile).
template <int HDP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn32_fwd_q64_kernel(
    const bf16_t* __restrict__ q, long qsb, long qsl, long qsh, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, long sb, long sl, long sh,
    bf16_t* __restrict__ out, long ob, long ol, long oh, float* __restrict__ lse, int H, int Lq, int Lk, int hd, float scale) {
  using C = A32<HDP>;
  __shared__ __attribute__((aligned(16))) char lds[4 * C::TILE];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hi = lane >> 5;
  const int npass = (Lq + 255) >> 8;
  const int wid = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = wid / npass;
  const int b = bh / H, h = bh - b * H;
  const int q0 = (wid - bh * npass) * 256 + wave * 64;
  const bf16_t* qb = q + (long)b * qsb + (long)h * qsh;
  const bf16_t* kb = k + (long)b * sb + (long)h * sh;
  const bf16_t* vb = v + (long)b * sb + (long)h * sh;
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
  u32x4 qf[2][C::KS];
  f32x16 o[2][C::MT];
  float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    a32_row_frags_global<HDP>(qb, qsl, q0 + 32 * u + (lane & 31), Lq, hd, qf[u], lane);
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[u][mt][r] = 0.f;
  }
  const float c2 = scale * A32_LOG2E;
  const int nt = (Lk + 63) >> 6;
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) asm volatile("" : "+v"(qf[u][ks]));
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
      f32x16 s[2][2];                                    // [query set][key half]
      {
        u32x4 kfr[2 * C::KS];
#pragma unroll
        for (int i = 0; i < 2 * C::KS; ++i) kfr[i] = a32_row_frag<HDP>(Kt, ln, i & 1, i >> 1);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int r = 0; r < 16; ++r) { s[u][0][r] = 0.f; s[u][1][r] = 0.f; }
#pragma unroll
        for (int i = 0; i < 2 * C::KS; ++i)
#pragma unroll
          for (int u = 0; u < 2; ++u) s[u][i & 1] = mfma32(kfr[i], qf[u][i >> 1], s[u][i & 1]);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int i = 0; i < 2 * C::KS; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          if (i < 2 * C::KS - 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      }
      float alpha[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float mt_ = -INFINITY;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if constexpr (RAGGED) {
              const int key = t * 64 + 32 * j + (r & 3) + 8 * (r >> 2) + 4 * hi;
              if (key >= Lk) s[u][j][r] = -INFINITY;
            }
            mt_ = fmaxf(mt_, s[u][j][r]);
          }
        mt_ = a32_max_halves(mt_);
        const float mn = fmaxf(m[u], mt_ * c2);
        alpha[u] = a32_exp2(m[u] - mn);
        m[u] = mn;
        const float ps = a32_exp_rows(s[u][0], c2, mn) + a32_exp_rows(s[u][1], c2, mn);
        l[u] = l[u] * alpha[u] + ps;
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[u][mt][r] *= alpha[u];
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const u32x4 pf0 = a32_pack8(s[0][j], c), pf1 = a32_pack8(s[1][j], c);
#pragma unroll
          for (int mt = 0; mt < C::MT; ++mt) {
            const u32x4 vf = a32_tr_frag<HDP>(Vt, ln, j, c, mt);
            o[0][mt] = mfma32(vf, pf0, o[0][mt]);
            o[1][mt] = mfma32(vf, pf1, o[1][mt]);
          }
        }
      // 3 fragments (6 transposing reads) ahead, then 2 MFMAs per fragment
      __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
      for (int i = 0; i < 4 * C::MT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        if (i < 4 * C::MT - 3) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
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
  if (active) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int qrow = q0 + 32 * u + (lane & 31);
      const float lt = a32_sum_halves(l[u]);
      const bool row_ok = qrow < Lq;
      if (row_ok && hi == 0 && lse) lse[((long)b * H + h) * Lq + qrow] = m[u] * A32_LN2 + logf(lt);
      a32_store_rows<HDP>(o[u], 1.0f / lt, out + (long)b * ob + (long)qrow * ol + (long)h * oh, row_ok, hd, lane);
    }
  }
}

// The shipped forward structure with parts compiled out (results are garbage): ABL 1 = no softmax arithmetic (P = S), 2 = no MFMAs,
// 3 = no LDS fragment reads (constant fragments), 4 = no LDS-DMA and no per-tile barrier (the first tile pair is reused), 0 = everything.
// AHK / AHV: how many fragment reads run ahead of the MFMA that consumes them in the QK^T / PV runs (shipped: 4 / 3); WPE = waves per SIMD
template <int HDP, int ABL, int AHK = 4, int AHV = 3, int WPE = 3>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE))) void attn32_fwd_ablate_kernel(
    const bf16_t* __restrict__ q, long qsb, long qsl, long qsh, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, long sb, long sl, long sh,
    bf16_t* __restrict__ out, long ob, long ol, long oh, int H, int Lq, int Lk, int hd, float scale) {
  using C = A32<HDP>;
  __shared__ __attribute__((aligned(16))) char lds[4 * C::TILE];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int npass = (Lq + 127) >> 7;
  const int wid = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = wid / npass;
  const int b = bh / H, h = bh - b * H;
  const int q0 = (wid - bh * npass) * 128 + wave * 32;
  const bf16_t* qb = q + (long)b * qsb + (long)h * qsh;
  const bf16_t* kb = k + (long)b * sb + (long)h * sh;
  const bf16_t* vb = v + (long)b * sb + (long)h * sh;
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
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) asm volatile("" : "+v"(qf[ks]));
  A32_WAIT_DMA();
  __builtin_amdgcn_s_barrier();
  u32x4 cfr = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  asm volatile("" : "+v"(cfr));
  auto tile = [&](const int t, auto par_tag) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_tag)::value;
    const char* Kt = lds + PAR * 2 * C::TILE;
    const char* Vt = Kt + C::TILE;
    if constexpr (ABL != 4) {
      constexpr unsigned nxt = (unsigned)((PAR ^ 1) * 2 * C::TILE);
      a32_dma_tile<HDP>(rs_k, voff, (unsigned)(t + 1) * tstep, nxt, wave);
      a32_dma_tile<HDP>(rs_v, voff, (unsigned)(t + 1) * tstep, nxt + (unsigned)C::TILE, wave);
    }
    if (active) {
      f32x16 s[2];
      {
        u32x4 kfr[2 * C::KS];
#pragma unroll
        for (int i = 0; i < 2 * C::KS; ++i) kfr[i] = (ABL == 3) ? cfr : a32_row_frag<HDP>(Kt, ln, i & 1, i >> 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[0][r] = 0.f; s[1][r] = 0.f; }
        if constexpr (ABL != 2) {
#pragma unroll
          for (int i = 0; i < 2 * C::KS; ++i) s[i & 1] = mfma32(kfr[i], qf[i >> 1], s[i & 1]);
          if constexpr (ABL != 3) a32_sched_pipeline<2 * C::KS, 1, AHK>();
        } else {
#pragma unroll
          for (int i = 0; i < 2 * C::KS; ++i) { asm volatile("" :: "v"(kfr[i])); }
#pragma unroll
          for (int r = 0; r < 16; ++r) { s[0][r] = __uint_as_float(kfr[r & 7][r & 3]) * 1e-3f; s[1][r] = __uint_as_float(kfr[(r + 3) & 7][r & 3]) * 1e-3f; }
        }
      }
      if constexpr (ABL != 1) {
        float mt_ = -INFINITY;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) mt_ = fmaxf(mt_, s[j][r]);
        mt_ = a32_max_halves(mt_);
        const float mn = fmaxf(m, mt_ * c2);
        const float alpha = a32_exp2(m - mn);
        m = mn;
        const float ps = a32_exp_rows(s[0], c2, mn) + a32_exp_rows(s[1], c2, mn);
        l = l * alpha + ps;
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[mt][r] *= alpha;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const u32x4 pf = a32_pack8(s[j], c);
#pragma unroll
          for (int mt = 0; mt < C::MT; ++mt) {
            const u32x4 vf = (ABL == 3) ? cfr : a32_tr_frag<HDP>(Vt, ln, j, c, mt);
            if constexpr (ABL != 2) o[mt] = mfma32(vf, pf, o[mt]);
            else { asm volatile("" :: "v"(vf), "v"(pf)); }
          }
        }
      if constexpr (ABL != 2 && ABL != 3) a32_sched_pipeline<4 * C::MT, 2, AHV>();
    }
    if constexpr (ABL != 4) {
      A32_WAIT_DMA();
      __builtin_amdgcn_s_barrier();
    }
  };
  {
    constexpr std::integral_constant<int, 0> P0{};
    constexpr std::integral_constant<int, 1> P1{};
    int t = 0;
    for (; t + 2 <= nt; t += 2) {
      tile(t, P0);
      if constexpr (ABL == 4) tile(t + 1, P0); else tile(t + 1, P1);
    }
    if (t < nt) tile(t, P0);
  }
  if (active) {
    const float lt = a32_sum_halves(l) + 1.0f;
    a32_store_rows<HDP>(o, 1.0f / lt, out + (long)b * ob + (long)qrow * ol + (long)h * oh, qrow < Lq, hd, lane);
  }
}

}  // namespace ivh

static float bf2f(uint16_t x) { unsigned u = (unsigned)x << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }

int main() {
  const int B = 128, H = 16, L = 417, hd = 88;
  constexpr int HDP = 96, NT = 7;
  const long D = (long)H * hd, qsl = 3 * D, qsb = (long)L * qsl, qsh = hd;
  const size_t n_qkv = (size_t)B * L * 3 * D, n_out = (size_t)B * L * D;
  std::vector<uint16_t> hq(n_qkv);
  unsigned st = 12345u;
  for (size_t i = 0; i < n_qkv; ++i) { st = st * 1664525u + 1013904223u; hq[i] = f2bf(((int)(st >> 9) % 2001 - 1000) * 1e-3f); }
  uint16_t *dq, *o_ref, *o_lab;
  float *lse_ref, *lse_lab;
  hipMalloc(&dq, n_qkv * 2); hipMalloc(&o_ref, n_out * 2); hipMalloc(&o_lab, n_out * 2);
  hipMalloc(&lse_ref, (size_t)B * H * L * 4); hipMalloc(&lse_lab, (size_t)B * H * L * 4);
  hipMemcpy(dq, hq.data(), n_qkv * 2, hipMemcpyHostToDevice);
  const float scale = 1.0f / sqrtf((float)hd);
  const int npass = (L + 127) / 128;
  dim3 grid(B * H * npass), block(256);
  auto run_ref = [&]() {
    hipLaunchKernelGGL((ivh::attn32_fwd_kernel<HDP, false>), grid, block, 0, 0, dq, qsb, qsl, qsh, dq + D, dq + 2 * D, qsb, qsl, qsh, o_ref, (long)L * D, D, (long)hd,
                       lse_ref, H, L, L, hd, scale, (const int32_t*)nullptr);
  };
  auto run_lab = [&]() {
    hipLaunchKernelGGL((ivh::attn32_fwd_lab_kernel<HDP, NT>), grid, block, 0, 0, dq, qsb, qsl, qsh, dq + D, dq + 2 * D, qsb, qsl, qsh, o_lab, (long)L * D, D, (long)hd,
                       lse_lab, H, L, L, hd, scale);
  };
  auto time_of = [&](auto&& fn) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) fn();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 40; ++i) fn();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 40 * 1e3f;
  };
  uint16_t* o_q64;
  float* lse_q64;
  hipMalloc(&o_q64, n_out * 2); hipMalloc(&lse_q64, (size_t)B * H * L * 4);
  dim3 grid64(B * H * ((L + 255) / 256));
  auto run_q64 = [&]() {
    hipLaunchKernelGGL((ivh::attn32_fwd_q64_kernel<HDP>), grid64, block, 0, 0, dq, qsb, qsl, qsh, dq + D, dq + 2 * D, qsb, qsl, qsh, o_q64, (long)L * D, D, (long)hd,
                       lse_q64, H, L, L, hd, scale);
  };
  run_ref(); run_lab(); run_q64();
  hipDeviceSynchronize();
  const hipError_t err = hipGetLastError();
  std::vector<uint16_t> a(n_out), bb(n_out);
  hipMemcpy(a.data(), o_ref, n_out * 2, hipMemcpyDeviceToHost);
  hipMemcpy(bb.data(), o_lab, n_out * 2, hipMemcpyDeviceToHost);
  double num = 0, den = 0, mx = 0;
  for (size_t i = 0; i < n_out; ++i) { const double x = bf2f(a[i]), y = bf2f(bb[i]); num += (x - y) * (x - y); den += x * x; mx = fmax(mx, fabs(x - y)); }
  std::vector<float> la((size_t)B * H * L), lb((size_t)B * H * L);
  hipMemcpy(la.data(), lse_ref, la.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(lb.data(), lse_lab, lb.size() * 4, hipMemcpyDeviceToHost);
  double lmx = 0;
  for (size_t i = 0; i < la.size(); ++i) lmx = fmax(lmx, fabs((double)la[i] - lb[i]));
  std::vector<uint16_t> cc(n_out);
  hipMemcpy(cc.data(), o_q64, n_out * 2, hipMemcpyDeviceToHost);
  double num64 = 0, mx64 = 0;
  for (size_t i = 0; i < n_out; ++i) { const double x = bf2f(a[i]), y = bf2f(cc[i]); num64 += (x - y) * (x - y); mx64 = fmax(mx64, fabs(x - y)); }
  const float t_q64 = time_of(run_q64);
  printf("{\"q64_rel_l2_out\": %.3e, \"q64_max_abs_out\": %.3e, \"q64_us\": %.1f, \"q64_tflops\": %.1f}\n", sqrt(num64 / fmax(den, 1e-30)), mx64, t_q64,
         4.0 * B * H * (double)L * L * hd / t_q64 / 1e6);
  float t_abl[5];
#define RUN_ABL(A)                                                                                                                                      \
  t_abl[A] = time_of([&]() {                                                                                                                            \
    hipLaunchKernelGGL((ivh::attn32_fwd_ablate_kernel<HDP, A>), grid, block, 0, 0, dq, qsb, qsl, qsh, dq + D, dq + 2 * D, qsb, qsl, qsh, o_lab, (long)L * D, D, \
                       (long)hd, H, L, L, hd, scale);                                                                                                   \
  })
  const float t_ref = time_of(run_ref), t_lab = time_of(run_lab);
  RUN_ABL(0); RUN_ABL(1); RUN_ABL(2); RUN_ABL(3); RUN_ABL(4);
  float t_ah[5];
#define RUN_AH(I, AK, AV, W)                                                                                                                            \
  t_ah[I] = time_of([&]() {                                                                                                                             \
    hipLaunchKernelGGL((ivh::attn32_fwd_ablate_kernel<HDP, 0, AK, AV, W>), grid, block, 0, 0, dq, qsb, qsl, qsh, dq + D, dq + 2 * D, qsb, qsl, qsh, o_lab,  \
                       (long)L * D, D, (long)hd, H, L, L, hd, scale);                                                                                   \
  })
  RUN_AH(0, 4, 3, 2); RUN_AH(1, 8, 6, 2); RUN_AH(2, 12, 12, 2); RUN_AH(3, 6, 4, 3); RUN_AH(4, 2, 1, 3);
  printf("{\"ahead_4_3_wpe2_us\": %.1f, \"ahead_8_6_wpe2_us\": %.1f, \"ahead_12_12_wpe2_us\": %.1f, \"ahead_6_4_wpe3_us\": %.1f, \"ahead_2_1_wpe3_us\": %.1f}\n",
         t_ah[0], t_ah[1], t_ah[2], t_ah[3], t_ah[4]);
  const double flop = 4.0 * B * H * (double)L * L * hd;
  printf("{\"hip_error\": %d, \"rel_l2_out\": %.3e, \"max_abs_out\": %.3e, \"max_abs_lse\": %.3e, \"ref_us\": %.1f, \"lab_us\": %.1f, \"ref_tflops\": %.1f, "
         "\"lab_tflops\": %.1f, \"speedup\": %.3f, \"ablate_all_us\": %.1f, \"no_softmax_us\": %.1f, \"no_mfma_us\": %.1f, \"no_lds_reads_us\": %.1f, "
         "\"no_dma_no_barrier_us\": %.1f}\n", (int)err, sqrt(num / fmax(den, 1e-30)), mx, lmx, t_ref, t_lab, flop / t_ref / 1e6, flop / t_lab / 1e6, t_ref / t_lab,
         t_abl[0], t_abl[1], t_abl[2], t_abl[3], t_abl[4]);
  return 0;
}
