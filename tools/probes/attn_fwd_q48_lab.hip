// Attention forward candidate for the next round (measurement aid, NOT part of the library): the LDS-DMA structure of flash_attn32.hip
// (double-buffered K / V tiles filled by buffer_load ... lds, one barrier per tile, unpadded rows with the per-row chunk permutation) with the
// 16x16x32 MFMA plan of flash_attn.hip and QW = 3 query blocks per wave: 48 queries per wave, so every 1 KiB K fragment (ds_read_b128) and
// every transposed V fragment (2 x ds_read_b64_tr_b16) feeds THREE 16-cycle MFMAs -- a fragment per 48 matrix cycles instead of per 32.
// Workgroup = NW = 3 waves = 144 queries: 3 workgroups per (b, h) = 432 slots for the 417 queries of the 1B shape (3.5 % padding).
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -w -o /tmp/attn_fwd_q48_lab tools/probes/attn_fwd_q48_lab.hip && /tmp/attn_fwd_q48_lab
// Prints the difference to the shipped 32x32x16 forward kernel (same fp32 arithmetic per query up to the MFMA's internal summation order) and
// both timings.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>

#include "../../internvideo_amd/csrc/flash_attn32.hip"
#include "../../internvideo_amd/csrc/flash_attn.hip"

namespace ivh_host {
void set_error(const char*, ...) {}
int check_launch(const char*) { return hipGetLastError() != hipSuccess ? -1 : 0; }
}  // namespace ivh_host

namespace ivh {

// Chunk permutation of the LDS image for THIS kernel's read patterns (HDP = 96: 12 chunks per row): rotation by 2 * ((r >> 2) & 3) instead of
// flash_attn32.hip's ((r >> 2) & 3).  With the shipped rotation every K fragment read (ds_read_b128, 16 rows x one chunk per 16-lane group) and
// every transposing V read costs twice its conflict-free cycles under the gfx950 bank map; this one is conflict-free for both (searched with
// the bank model of tools/attn32_layout_check.py: rotations 2, 6 or 10 per 4-row group work).
template <int HDP> __device__ __forceinline__ int q48_phys(int r, int c) {
  static_assert(HDP == 96, "the lab kernel is written for the 1B head dim (88 padded to 96)");
  const int x = c + 2 * ((r >> 2) & 3);
  return x >= 12 ? x - 12 : x;
}
template <int HDP> __device__ __forceinline__ int q48_logical(int r, int x) {
  const int c = x - 2 * ((r >> 2) & 3);
  return c < 0 ? c + 12 : c;
}

// per-lane DMA source offsets of the wave's requests of a 64-row tile, NW waves sharing its TILE / 1024 requests (cf. a32_dma_offsets)
template <int HDP, int NW>
__device__ __forceinline__ void q48_dma_offsets(int lane, int wave, long sl, int hd, unsigned* voff) {
  using C = A32<HDP>;
  constexpr int RPW = C::TILE / 1024 / NW;
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int n = (wave * RPW + i) * 64 + lane;
    const int r = n / C::CPR, x = n - r * C::CPR;
    const int c = q48_logical<HDP>(r, x);
    voff[i] = (c * 8 < hd) ? (unsigned)(((long)r * sl + c * 8) * 2) : A32_OOB;
  }
}
template <int HDP, int NW>
__device__ __forceinline__ void q48_dma_tile(u32x4 rs, const unsigned* voff, unsigned toff, unsigned tile, int wave) {
  using C = A32<HDP>;
  constexpr int RPW = C::TILE / 1024 / NW;
#pragma unroll
  for (int i = 0; i < RPW; ++i) a32_dma16(rs, tile + (unsigned)((wave * RPW + i) * 1024), voff[i] + toff);
}

// max / sum over the four 16-lane rows of a wave (the 4 x 4 keys a query's column is spread over), result in every row: two swap instructions
// (v_permlane16_swap, v_permlane32_swap) instead of two LDS permutes in the middle of the softmax's dependency chain
__device__ __forceinline__ float q48_max_rows(float x) {
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return a32_max_halves(fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1])));
}
__device__ __forceinline__ float q48_sum_rows(float x) {
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return a32_sum_halves(__uint_as_float(r[0]) + __uint_as_float(r[1]));
}

// SCHED = 1: read-ahead directives for the two MFMA runs of a tile (as a32_sched_pipeline does for the shipped kernel): the fragment reads run 3 / 2
// fragments ahead of the QW MFMAs that consume each of them; 0 = the machine scheduler's order
template <int HDP, int QW, int NW, int SCHED = 0>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2))) void attn_fwd_q48_kernel(
    const bf16_t* __restrict__ q, long qsb, long qsl, long qsh, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, long sb, long sl, long sh,
    bf16_t* __restrict__ out, long ob, long ol, long oh, float* __restrict__ lse, int H, int Lq, int Lk, int hd, float scale) {
  using C = A32<HDP>;
  constexpr int KS = HDP / 32, DT = HDP / 16, QPW = NW * 16 * QW;
  static_assert((C::TILE / 1024) % NW == 0, "the tile's DMA requests must divide among the waves");
  __shared__ __attribute__((aligned(16))) char lds[4 * C::TILE];            // [buffer][K, V]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i16 = lane & 15, g = lane >> 4;
  const int ntq = (Lq + QPW - 1) / QPW;
  const int wid = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = wid / ntq;
  const int b = bh / H, h = bh - b * H;
  const int q0 = (wid - bh * ntq) * QPW + wave * 16 * QW;
  const bf16_t* qb = q + (long)b * qsb + (long)h * qsh;
  const bf16_t* kb = k + (long)b * sb + (long)h * sh;
  const bf16_t* vb = v + (long)b * sb + (long)h * sh;
  const int range = (int)((((long)Lk - 1) * sl + hd) * 2);
  const u32x4 rs_k = a32_rsrc(kb, range);
  const u32x4 rs_v = a32_rsrc(vb, range);
  unsigned voff[C::TILE / 1024 / NW];
  q48_dma_offsets<HDP, NW>(lane, wave, sl, hd, voff);
  const unsigned tstep = (unsigned)(64 * sl * 2);
  q48_dma_tile<HDP, NW>(rs_k, voff, 0u, 0u, wave);
  q48_dma_tile<HDP, NW>(rs_v, voff, 0u, (unsigned)C::TILE, wave);

  // fragment read bases.  K (A operand, 16 keys x 32 head-dim): row 16 j + i16, logical chunk 4 ks + g.  The chunk permutation depends on the
  // row only through bits that 16 j leaves alone (see flash_attn32.hip), so the 16 j * RS term is an immediate.
  unsigned krow[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) krow[ks] = (unsigned)(i16 * C::RS + q48_phys<HDP>(i16, 4 * ks + g) * 16);
  // V^T (A operand, 16 head-dim x 32 keys): lane i16 of group g points at row 32 c + 4 g + (i16 >> 2) (+ 16 for the second 4-key group),
  // columns 16 dt + 4 (i16 & 3) .. + 3 = chunk 2 dt + ((i16 & 3) >> 1), byte (i16 & 1) * 8 inside it
  unsigned vtr[DT][2];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int sec = 0; sec < 2; ++sec) {
      const int r = 4 * g + (i16 >> 2) + 16 * sec;
      vtr[dt][sec] = (unsigned)(r * C::RS + q48_phys<HDP>(r, 2 * dt + ((i16 & 3) >> 1)) * 16 + (i16 & 1) * 8);
    }

  const bool active = q0 < Lq;
  s16x8 qf[QW][KS];
  f32x4 o[QW][DT];
  float m[QW], l[QW];
  int qrow[QW];
#pragma unroll
  for (int w = 0; w < QW; ++w) {
    qrow[w] = q0 + 16 * w + i16;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int d = 32 * ks + 8 * g;
      if (qrow[w] < Lq && d < hd) qf[w][ks] = *reinterpret_cast<const s16x8*>(qb + (long)qrow[w] * qsl + d);
      else qf[w][ks] = s16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[w][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    m[w] = -INFINITY; l[w] = 0.f;
  }
  const float c2 = scale * A32_LOG2E;
  const int nt = (Lk + 63) >> 6;
#pragma unroll
  for (int w = 0; w < QW; ++w)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[w][ks]));
  A32_WAIT_DMA();
  __builtin_amdgcn_s_barrier();

  auto tile = [&](const int t, auto par_tag, auto ragged_tag) __attribute__((always_inline)) {
    constexpr bool RAGGED = decltype(ragged_tag)::value;
    constexpr int PAR = decltype(par_tag)::value;
    const char* Kt = lds + PAR * 2 * C::TILE;
    const char* Vt = Kt + C::TILE;
    if (!RAGGED) {
      constexpr unsigned nxt = (unsigned)((PAR ^ 1) * 2 * C::TILE);
      q48_dma_tile<HDP, NW>(rs_k, voff, (unsigned)(t + 1) * tstep, nxt, wave);
      q48_dma_tile<HDP, NW>(rs_v, voff, (unsigned)(t + 1) * tstep, nxt + (unsigned)C::TILE, wave);
    }
    if (active) {
      // S^T tiles: rows = keys 16 j + 4 g + r, col = this lane's query of each block
      f32x4 s[QW][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int w = 0; w < QW; ++w) s[w][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const s16x8 kfrag = *reinterpret_cast<const s16x8*>(Kt + krow[ks] + 16 * j * C::RS);
#pragma unroll
          for (int w = 0; w < QW; ++w) s[w][j] = mfma16(kfrag, qf[w][ks], s[w][j]);
        }
      }
      if constexpr (SCHED == 1) {
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
        for (int i = 0; i < 4 * KS; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, QW, 0);
          if (i < 4 * KS - 3) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      }
      s16x8 pf[QW][2];
#pragma unroll
      for (int w = 0; w < QW; ++w) {
        float mt = -INFINITY;
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
        mt = q48_max_rows(mt);
        const float mn = fmaxf(m[w], mt * c2);
        const float alpha = a32_exp2(m[w] - mn);
        m[w] = mn;
        float ps = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) { s[w][j][r] = a32_exp2(fmaf(s[w][j][r], c2, -mn)); ps += s[w][j][r]; }
        l[w] = l[w] * alpha + ps;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r) o[w][dt][r] *= alpha;
        pf[w][0] = pack_frag(s[w][0], s[w][1]);
        pf[w][1] = pack_frag(s[w][2], s[w][3]);
      }
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const s16x4 t0 = lds_tr16(Vt + vtr[dt][0] + 32 * c * C::RS);
          const s16x4 t1 = lds_tr16(Vt + vtr[dt][1] + 32 * c * C::RS);
          s16x8 vfrag;
          vfrag[0] = t0[0]; vfrag[1] = t0[1]; vfrag[2] = t0[2]; vfrag[3] = t0[3];
          vfrag[4] = t1[0]; vfrag[5] = t1[1]; vfrag[6] = t1[2]; vfrag[7] = t1[3];
#pragma unroll
          for (int w = 0; w < QW; ++w) o[w][dt] = mfma16(vfrag, pf[w][c], o[w][dt]);
        }
      if constexpr (SCHED == 1) {
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int i = 0; i < 2 * DT; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, QW, 0);
          if (i < 2 * DT - 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
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
    for (int w = 0; w < QW; ++w) {
      const float lw = q48_sum_rows(l[w]);
      const float inv = 1.0f / lw;
      if (qrow[w] < Lq) {
        if (g == 0 && lse) lse[((long)b * H + h) * Lq + qrow[w]] = m[w] * A32_LN2 + logf(lw);
        bf16_t* op = out + (long)b * ob + (long)qrow[w] * ol + (long)h * oh;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const int d = 16 * dt + 4 * g;
          if (d < hd) *reinterpret_cast<u32x2*>(op + d) = pack4(o[w][dt][0] * inv, o[w][dt][1] * inv, o[w][dt][2] * inv, o[w][dt][3] * inv);
        }
      }
    }
  }
}

}  // namespace ivh

static float bf2f(uint16_t x) { unsigned u = (unsigned)x << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }

template <typename F>
static double time_us(F&& fn) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) fn();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 40; ++i) fn();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  return ms / 40 * 1e3;
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int B = 128, H = 16, L = 417, hd = 88;
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
  hipMemset(o_lab, 0, n_out * 2);
  const float scale = 1.0f / sqrtf((float)hd);
  auto run_ref = [&]() {
    ivh::attn32_fwd_kernel<96, false><<<dim3(B * H * ((L + 127) / 128)), dim3(256), 0, 0>>>(dq, qsb, qsl, qsh, dq + D, dq + 2 * D, qsb, qsl, qsh, o_ref, (long)L * D, D, (long)hd,
                                                                                           lse_ref, H, L, L, hd, scale, (const int32_t*)nullptr);
  };
  auto run_lab = [&]() {
    ivh::attn_fwd_q48_kernel<96, 3, 3><<<dim3(B * H * ((L + 143) / 144)), dim3(192), 0, 0>>>(dq, qsb, qsl, qsh, dq + D, dq + 2 * D, qsb, qsl, qsh, o_lab, (long)L * D, D, (long)hd,
                                                                                           lse_lab, H, L, L, hd, scale);
  };
  auto run_v1 = [&]() {   // read-ahead directives
    ivh::attn_fwd_q48_kernel<96, 3, 3, 1><<<dim3(B * H * ((L + 143) / 144)), dim3(192), 0, 0>>>(dq, qsb, qsl, qsh, dq + D, dq + 2 * D, qsb, qsl, qsh, o_lab, (long)L * D, D,
                                                                                              (long)hd, lse_lab, H, L, L, hd, scale);
  };
  auto run_v2 = [&]() {   // four waves per workgroup (192 queries: all eight wave slots of a CU, 28 % query padding at L = 417)
    ivh::attn_fwd_q48_kernel<96, 3, 4><<<dim3(B * H * ((L + 191) / 192)), dim3(256), 0, 0>>>(dq, qsb, qsl, qsh, dq + D, dq + 2 * D, qsb, qsl, qsh, o_lab, (long)L * D, D, (long)hd,
                                                                                           lse_lab, H, L, L, hd, scale);
  };
  auto run_v3 = [&]() {
    ivh::attn_fwd_q48_kernel<96, 3, 4, 1><<<dim3(B * H * ((L + 191) / 192)), dim3(256), 0, 0>>>(dq, qsb, qsl, qsh, dq + D, dq + 2 * D, qsb, qsl, qsh, o_lab, (long)L * D, D,
                                                                                              (long)hd, lse_lab, H, L, L, hd, scale);
  };
  run_ref();
  printf("{\"shipped_sync\": %d}\n", (int)hipDeviceSynchronize());
  run_lab();
  printf("{\"q48_sync\": %d}\n", (int)hipDeviceSynchronize());
  std::vector<uint16_t> a(n_out), c(n_out);
  hipMemcpy(a.data(), o_ref, n_out * 2, hipMemcpyDeviceToHost);
  hipMemcpy(c.data(), o_lab, n_out * 2, hipMemcpyDeviceToHost);
  double num = 0, den = 0, mx = 0;
  for (size_t i = 0; i < n_out; ++i) { const double x = bf2f(a[i]), y = bf2f(c[i]); num += (x - y) * (x - y); den += x * x; mx = fmax(mx, fabs(x - y)); }
  std::vector<float> la((size_t)B * H * L), lb((size_t)B * H * L);
  hipMemcpy(la.data(), lse_ref, la.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(lb.data(), lse_lab, lb.size() * 4, hipMemcpyDeviceToHost);
  double lmx = 0;
  for (size_t i = 0; i < la.size(); ++i) lmx = fmax(lmx, fabs((double)la[i] - lb[i]));
  printf("{\"rel_l2_vs_shipped\": %.3e, \"max_abs_out\": %.3e, \"max_abs_lse\": %.3e}\n", sqrt(num / fmax(den, 1e-30)), mx, lmx);
  auto check = [&](const char* name, auto&& fn) {        // every variant against the shipped kernel, then its time
    hipMemset(o_lab, 0, n_out * 2);
    fn();
    const int e = (int)hipDeviceSynchronize();
    hipMemcpy(c.data(), o_lab, n_out * 2, hipMemcpyDeviceToHost);
    double nn = 0, mm = 0;
    for (size_t i = 0; i < n_out; ++i) { const double x = bf2f(a[i]), y = bf2f(c[i]); nn += (x - y) * (x - y); mm = fmax(mm, fabs(x - y)); }
    const double t = time_us(fn);
    printf("{\"variant\": \"%s\", \"sync\": %d, \"rel_l2_vs_shipped\": %.3e, \"max_abs\": %.3e, \"us\": %.1f}\n", name, e, sqrt(nn / fmax(den, 1e-30)), mm, t);
  };
  check("3 waves + read-ahead directives", run_v1);
  check("4 waves", run_v2);
  check("4 waves + read-ahead directives", run_v3);
  const double t_ref = time_us(run_ref), t_lab = time_us(run_lab);
  const double flop = 4.0 * B * H * (double)L * L * hd;
  printf("{\"shipped_us\": %.1f, \"q48_us\": %.1f, \"shipped_tflops\": %.1f, \"q48_tflops\": %.1f, \"speedup\": %.3f}\n", t_ref, t_lab, flop / t_ref / 1e6, flop / t_lab / 1e6,
         t_ref / t_lab);
  return 0;
}
