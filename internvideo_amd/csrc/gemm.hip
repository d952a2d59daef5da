// bf16 MFMA GEMM for gfx950:  C[m,n] = epi( alpha * sum_k A(m,k) B(n,k) ),  fp32 accumulate.
//
// One kernel serves the forward Linear (A, B both K-contiguous), dgrad (B stored [K][N]) and wgrad
// (A and B stored [K][rows]) of SURVEY.md 8(a) rows a5, a8, a9, a13, a14: the non-K-contiguous operands
// are staged as they lie in HBM (coalesced 16-byte LDS-DMA) and transposed on the way to the MFMA by
// ds_read_b64_tr_b16, so no transposed copy of an activation or a weight is ever materialised.
//
// Tile 128 x 128 x 64, 256 threads = 4 waves (2 x 2), each wave 64 x 64 = 4 x 4 MFMA 16x16x32 tiles.
// LDS: 2 stages x (16 KiB A + 16 KiB B), filled by global_load_lds (16 B / lane, 1 KiB / wave-instruction),
// stage t+1 in flight while stage t is multiplied; one barrier per K step.
// LDS image, K-contiguous operand  : [128 rows][64 k]  (128 B rows), 16-B chunk c of row r stored at chunk
//                                    slot c ^ ((r >> 1) & 7): ds_read_b128 of 16 rows x one chunk is conflict free.
// LDS image, rows-contiguous operand: [64 k][128 rows] (256 B rows), chunk c of k-row kr stored at slot
//                                    c ^ (((kr & 3) << 1) | (((kr >> 3) & 1) << 3)) for the transposing read.
// LDS-DMA writes lane-linear, so the swizzle is applied to the per-lane SOURCE address and to the read.
// MFMA orientation: D = mfma(Bfrag, Afrag): D rows = n (4 consecutive per lane), cols = m (lane & 15), so
// a lane owns 4 consecutive n of one output row: 8-byte bf16x4 / 16-byte fp32x4 row-major stores.
#include "common.h"
#include "../../include/internvideo_hip.h"

namespace ivh {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 64 * 2;  // 16 KiB per operand per stage

struct GemmParams {
  const bf16_t* A; const bf16_t* B;
  long lda, ldb;
  int M, N, K;
  void* C; long ldc; int c_fp32;
  const float* bias;
  int act;
  bf16_t* preact; long ldp;
  const bf16_t* dact_in; long ldd;
  float alpha;
  int tiles_m, tiles_n;
  long strideA, strideB, strideC, stride_bias, stride_preact, stride_dact;
};

template <bool KC>
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ base, long ld, int rows_total, int row0,
                                           int K, int k0, char* lds_tile, int wave, int lane) {
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page);
  if constexpr (KC) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = wave * 4 + j;        // 1 KiB piece = 8 rows x 128 B
      const int r = q * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      int grow = row0 + r;
      grow = grow < rows_total ? grow : rows_total - 1;
      const int k = k0 + c * 8;
      const bf16_t* src = (k < K) ? base + (long)grow * ld + k : zero;
      glds16(src, lds_tile + q * 1024);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = wave * 4 + j;        // 1 KiB piece = 4 k-rows x 256 B
      const int kr = q * 4 + (lane >> 4);
      const int s = ((kr & 3) << 1) | (((kr >> 3) & 1) << 3);
      const int c = (lane & 15) ^ s;
      const int k = k0 + kr;
      const int rr = row0 + c * 8;
      const bf16_t* src = (k < K && rr < rows_total) ? base + (long)k * ld + rr : zero;
      glds16(src, lds_tile + q * 1024);
    }
  }
}

template <bool KC>
__device__ __forceinline__ s16x8 load_frag(const char* lds_tile, int rbase, int kk, int lane) {
  if constexpr (KC) {
    const int r = rbase + (lane & 15);
    const int c = kk * 4 + (lane >> 4);
    return *reinterpret_cast<const s16x8*>(lds_tile + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
  } else {
    const int i = lane & 15, g = lane >> 4;
    const int chunk = (rbase >> 3) + ((i & 3) >> 1);
    const int within = (i & 1) * 8;
    const int kr0 = kk * 32 + 8 * g + (i >> 2);
    const int kr1 = kr0 + 4;
    const int s0 = ((kr0 & 3) << 1) | (((kr0 >> 3) & 1) << 3);
    const int s1 = ((kr1 & 3) << 1) | (((kr1 >> 3) & 1) << 3);
    const s16x4 t0 = lds_tr16(lds_tile + kr0 * 256 + ((chunk ^ s0) << 4) + within);
    const s16x4 t1 = lds_tr16(lds_tile + kr1 * 256 + ((chunk ^ s1) << 4) + within);
    s16x8 r;
    r[0] = t0[0]; r[1] = t0[1]; r[2] = t0[2]; r[3] = t0[3];
    r[4] = t1[0]; r[5] = t1[1]; r[6] = t1[2]; r[7] = t1[3];
    return r;
  }
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) char lds[4 * TILE_BYTES];  // [stage][A|B]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // tile id: XCD-contiguous remap, then groups of 8 m-tiles walked n-major for L2 reuse of both panels
  const int nwg = p.tiles_m * p.tiles_n;
  const int id = xcd_remap(blockIdx.x, nwg);
  constexpr int G = 8;
  const int per_group = G * p.tiles_n;
  const int grp = id / per_group;
  const int first_m = grp * G;
  const int gsz = min(G, p.tiles_m - first_m);
  const int in_grp = id - grp * per_group;
  const int tile_m = first_m + in_grp % gsz;
  const int tile_n = in_grp / gsz;
  const int z = blockIdx.z;

  const bf16_t* A = p.A + (long)z * p.strideA;
  const bf16_t* B = p.B + (long)z * p.strideB;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (p.K + BK - 1) / BK;
  stage_tile<A_KC>(A, p.lda, p.M, m0, p.K, 0, lds, wave, lane);
  stage_tile<B_KC>(B, p.ldb, p.N, n0, p.K, 0, lds + TILE_BYTES, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    char* cur = lds + (kt & 1) * (2 * TILE_BYTES);
    if (kt + 1 < nk) {
      char* nxt = lds + ((kt + 1) & 1) * (2 * TILE_BYTES);
      stage_tile<A_KC>(A, p.lda, p.M, m0, p.K, (kt + 1) * BK, nxt, wave, lane);
      stage_tile<B_KC>(B, p.ldb, p.N, n0, p.K, (kt + 1) * BK, nxt + TILE_BYTES, wave, lane);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      s16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = load_frag<A_KC>(cur, wm * 64 + i * 16, kk, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = load_frag<B_KC>(cur + TILE_BYTES, wn * 64 + j * 16, kk, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = mfma16(bfr[j], af[i], acc[j][i]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // epilogue: lane owns row m = .. + (lane & 15) and the 4 consecutive columns n = .. + 4 * (lane >> 4) + {0..3}
  const int g = lane >> 4;
  const float* bias = p.bias ? p.bias + (long)z * p.stride_bias : nullptr;
  bf16_t* preact = p.preact ? p.preact + (long)z * p.stride_preact : nullptr;
  const bf16_t* dact = p.dact_in ? p.dact_in + (long)z * p.stride_dact : nullptr;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + (lane & 15);
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + 4 * g;
      if (n >= p.N) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[j][i][r] * p.alpha;
      if (bias) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bv[r];
      }
      if (preact) {
        if (p.act == 3) *reinterpret_cast<u32x2*>(preact + (long)m * p.ldp + n) = pack4(dgelu_erf(v[0]), dgelu_erf(v[1]), dgelu_erf(v[2]), dgelu_erf(v[3]));
        else *reinterpret_cast<u32x2*>(preact + (long)m * p.ldp + n) = pack4(v[0], v[1], v[2], v[3]);
      }
      if (dact) {
        const u32x2 uu = *reinterpret_cast<const u32x2*>(dact + (long)m * p.ldd + n);
        float u[4] = {__uint_as_float(uu[0] << 16), __uint_as_float(uu[0] & 0xffff0000u),
                      __uint_as_float(uu[1] << 16), __uint_as_float(uu[1] & 0xffff0000u)};
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= (p.act == 3) ? u[r] : ((p.act == 2) ? dgelu_tanh(u[r]) : dgelu_erf(u[r]));
      } else if (p.act == 1 || p.act == 3) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
      } else if (p.act == 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_tanh(v[r]);
      }
      if (p.c_fp32) {
        float* C = reinterpret_cast<float*>(p.C) + (long)z * p.strideC;
        *reinterpret_cast<f32x4*>(C + (long)m * p.ldc + n) = f32x4{v[0], v[1], v[2], v[3]};
      } else {
        bf16_t* C = reinterpret_cast<bf16_t*>(p.C) + (long)z * p.strideC;
        *reinterpret_cast<u32x2*>(C + (long)m * p.ldc + n) = pack4(v[0], v[1], v[2], v[3]);
      }
    }
  }
}

}  // namespace ivh

extern "C" int ivh_gemm256_launch(const ivh_gemm_desc* d, void* stream);   // gemm256.hip
extern "C" int ivh_gemm256_supported(const ivh_gemm_desc* d);

static int g_gemm_kernel_choice = 0;   // 0 = heuristic, 1 = 128^2 / 4-wave kernel, 2 = 256^2 / 8-wave ping-pong kernel
extern "C" int ivh_set_gemm_kernel(int choice) {
  IVH_REQUIRE(choice >= 0 && choice <= 2, "set_gemm_kernel: choice must be 0 (auto), 1 (128^2) or 2 (256^2)");
  g_gemm_kernel_choice = choice;
  return 0;
}

// Launch-time model (microseconds on MI355X, fitted to tools/bench_gemm.py / bench_gemm_ksweep.py on the 1B block shapes,
// profiles/r1_gemm_*): a kernel runs ceil(tiles / slots) rounds of (a * k_steps + b).
//   256^2 / 8 waves, persistent, one workgroup per CU: a = 1.45 (any operand layout), b = 4 (epilogue of both wave groups);
//   128^2 / 4 waves, two workgroups per CU:            a = 0.80, b = 6, x 1.2 with a rows-contiguous (transposed-read) operand.
//   256^2 with the tail round split along K (gemm256.hip, SPLIT): the last round runs `tail_frac` of the K steps + ~35 us of exchange.
static double gemm_time_model(int M, int N, int K, int batch, bool any_tr, bool big, double tail_frac = 1.0) {
  const int bm = big ? 256 : 128;
  const long tiles = (long)((M + bm - 1) / bm) * ((N + bm - 1) / bm) * batch;
  const long slots = big ? 256 : 512;
  const long rounds = (tiles + slots - 1) / slots;
  const double nk = (K + 63) / 64;
  const double per_round = big ? (1.45 * nk + 4.0) : (0.80 * nk + 6.0) * (any_tr ? 1.2 : 1.0);
  if (big && tail_frac < 1.0) return (rounds - 1) * per_round + (1.45 * nk * tail_frac + 40.0);
  return rounds * per_round;
}

// which kernel ivh_gemm_bf16 would launch for this problem: 1 = 128^2, 2 = 256^2
extern "C" int ivh_gemm256_fits(const ivh_gemm_desc* d);   // gemm256.hip

extern "C" int64_t ivh_gemm256_split_ws_bytes(const ivh_gemm_desc* d, int fp8);      // gemm256.hip
extern "C" double ivh_gemm256_split_tail_frac(const ivh_gemm_desc* d, int fp8);
extern "C" double ivh_gemm256_half_rounds(const ivh_gemm_desc* d);                    // half-width tiles: launch length in rounds, or -1

static int gemm_select_impl(const ivh_gemm_desc* d, bool assume_ws) {
  if (!ivh_gemm256_supported(d)) return 1;
  if (d->m_dev || d->k_dev) return 2;                    // device-side row counts exist in the 256^2 kernel only
  if (g_gemm_kernel_choice) return g_gemm_kernel_choice;
  const int batch = d->batch > 0 ? d->batch : 1;
  const bool any_tr = !d->a_kc || !d->b_kc;
  double frac = 1.0;
  if (assume_ws || (d->split_ws && d->split_ws_bytes >= ivh_gemm256_split_ws_bytes(d, 0))) frac = ivh_gemm256_split_tail_frac(d, 0);
  double t256 = gemm_time_model(d->M, d->N, d->K, batch, any_tr, true, frac);
  if (frac >= 1.0) {                                     // (the K split of the tail round keeps precedence in ivh_gemm256_launch)
    const double hr = ivh_gemm256_half_rounds(d);
    if (hr > 0.0) t256 = hr * (1.45 * ((d->K + 63) / 64) + 4.0);
  }
  const double t128 = gemm_time_model(d->M, d->N, d->K, batch, any_tr, false);
  return (t256 <= t128 * 1.02) ? 2 : 1;
}

extern "C" int ivh_gemm_select(const ivh_gemm_desc* d) {
  IVH_REQUIRE(d, "gemm_select: null descriptor");
  return gemm_select_impl(d, false);
}

// bytes of scratch the caller may pass in d->split_ws so that the 256^2 kernel cuts the tail round of this problem along K (0 = no use)
extern "C" int64_t ivh_gemm_split_workspace(const ivh_gemm_desc* d) {
  if (d && d->m_dev) return (ivh_gemm256_supported(d) && ivh_gemm256_fits(d)) ? ivh_gemm256_split_ws_bytes(d, 0) : 0;   // the 256^2 kernel is the only one
  if (!d || g_gemm_kernel_choice == 1) return 0;
  const int64_t need = ivh_gemm256_split_ws_bytes(d, 0);
  if (need <= 0 || !ivh_gemm256_fits(d)) return 0;
  return gemm_select_impl(d, true) == 2 ? need : 0;
}

extern "C" int ivh_gemm_bf16(const ivh_gemm_desc* d, void* stream) {
  using namespace ivh;
  IVH_REQUIRE(d && d->A && d->B && d->C, "gemm: null operand");
  IVH_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "gemm: empty problem M=%d N=%d K=%d", d->M, d->N, d->K);
  IVH_REQUIRE(d->N % 8 == 0, "gemm: N=%d must be a multiple of 8", d->N);
  IVH_REQUIRE(d->lda % 8 == 0 && d->ldb % 8 == 0 && d->ldc % 4 == 0, "gemm: leading dims must be multiples of 8");
  if (d->a_kc) IVH_REQUIRE(d->K % 8 == 0, "gemm: K=%d must be a multiple of 8 for a K-contiguous A", d->K);
  else IVH_REQUIRE(d->M % 8 == 0, "gemm: M=%d must be a multiple of 8 for a rows-contiguous A", d->M);
  if (d->b_kc) IVH_REQUIRE(d->K % 8 == 0, "gemm: K=%d must be a multiple of 8 for a K-contiguous B", d->K);
  IVH_REQUIRE(((uintptr_t)d->A % 16) == 0 && ((uintptr_t)d->B % 16) == 0 && ((uintptr_t)d->C % 16) == 0,
              "gemm: base pointers must be 16-byte aligned");
  IVH_REQUIRE(d->act >= 0 && d->act <= 3, "gemm: unknown activation %d", d->act);
  IVH_REQUIRE(!d->colsum_part || ivh_gemm_select(d) == 2, "gemm: colsum_part is produced by the 256x256 dgrad epilogue only (check ivh_gemm_select)");
  if (d->m_dev || d->k_dev) {
    IVH_REQUIRE(ivh_gemm256_supported(d) && ivh_gemm256_fits(d) && (d->batch <= 1) && !d->c_fp32,
                "gemm: device-side row counts (m_dev / k_dev) need the 256x256 kernel: bf16 output, batch 1, operands below 2 GiB, erf-GELU or plain epilogue");
    IVH_REQUIRE(((uintptr_t)d->m_dev % 4) == 0 && ((uintptr_t)d->k_dev % 4) == 0, "gemm: m_dev / k_dev must be 4-byte aligned");
    return ivh_gemm256_launch(d, stream);
  }
  if (ivh_gemm_select(d) == 2) {
    if (ivh_gemm256_fits(d)) return ivh_gemm256_launch(d, stream);
    // an operand or output of 2 GiB or more.  K-contiguous A (forward / dgrad): row blocks of A, C (and the epilogue operands) are
    // independent problems -- split M into the fewest blocks of whole 256-row tiles that fit.  Anything else (a weight gradient whose
    // token dimension is K, batched operands) runs on the 128^2 kernel below, which addresses with 64-bit pointers.
    const int batch1 = d->batch > 0 ? d->batch : 1;
    if (d->a_kc && batch1 == 1) {
      const long lim = (1L << 31) - (1L << 24);
      long row_bytes = d->lda * 2;
      if (d->ldc * (d->c_fp32 ? 4 : 2) > row_bytes) row_bytes = d->ldc * (d->c_fp32 ? 4 : 2);
      if (d->preact && d->ldp * 2 > row_bytes) row_bytes = d->ldp * 2;
      if (d->dact_in && d->ldd * 2 > row_bytes) row_bytes = d->ldd * 2;
      long rows = (lim / row_bytes) / 256 * 256;
      ivh_gemm_desc probe = *d;
      probe.M = rows > 0 ? (int)(rows < d->M ? rows : d->M) : 0;
      if (rows >= 256 && ivh_gemm256_fits(&probe)) {
        for (long m0 = 0; m0 < d->M; m0 += rows) {
          ivh_gemm_desc q = *d;
          q.M = (int)((d->M - m0) < rows ? (d->M - m0) : rows);
          q.A = d->A + m0 * d->lda;
          q.C = d->c_fp32 ? (void*)((float*)d->C + m0 * d->ldc) : (void*)((uint16_t*)d->C + m0 * d->ldc);
          if (d->preact) q.preact = d->preact + m0 * d->ldp;
          if (d->dact_in) q.dact_in = d->dact_in + m0 * d->ldd;
          if (d->colsum_part) q.colsum_part = d->colsum_part + 2 * (m0 / 256) * d->N;
          const int rc = ivh_gemm256_launch(&q, stream);
          if (rc) return rc;
        }
        return 0;
      }
    }
    IVH_REQUIRE(!d->colsum_part, "gemm: colsum_part needs the 256x256 kernel, which cannot address this problem (an operand of 2 GiB or more)");
  }
  GemmParams p;
  p.A = d->A; p.B = d->B; p.lda = d->lda; p.ldb = d->ldb; p.M = d->M; p.N = d->N; p.K = d->K;
  p.C = d->C; p.ldc = d->ldc; p.c_fp32 = d->c_fp32; p.bias = d->bias; p.act = d->act;
  p.preact = d->preact; p.ldp = d->ldp; p.dact_in = d->dact_in; p.ldd = d->ldd;
  p.alpha = d->alpha; p.tiles_m = (d->M + BM - 1) / BM; p.tiles_n = (d->N + BN - 1) / BN;
  p.strideA = d->strideA; p.strideB = d->strideB; p.strideC = d->strideC; p.stride_bias = d->stride_bias;
  p.stride_preact = d->stride_preact; p.stride_dact = d->stride_dact;
  const int batch = d->batch > 0 ? d->batch : 1;
  dim3 grid(p.tiles_m * p.tiles_n, 1, batch), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (d->a_kc && d->b_kc) hipLaunchKernelGGL((gemm_bf16_kernel<true, true>), grid, block, 0, s, p);
  else if (d->a_kc && !d->b_kc) hipLaunchKernelGGL((gemm_bf16_kernel<true, false>), grid, block, 0, s, p);
  else if (!d->a_kc && d->b_kc) hipLaunchKernelGGL((gemm_bf16_kernel<false, true>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((gemm_bf16_kernel<false, false>), grid, block, 0, s, p);
  return ivh_host::check_launch("gemm_bf16");
}


extern "C" int ivh_gemm256_grouped_launch(const ivh_gemm_desc* d, int n, void* stream);   // gemm256.hip

extern "C" int ivh_gemm_grouped_bf16(const ivh_gemm_desc* d, int n, void* stream) {
  IVH_REQUIRE(d && n > 0, "gemm_grouped: empty problem list");
  if (g_gemm_kernel_choice != 1) {
    int done = 0;                                        // groups of <= 32 problems
    bool grouped_all = true;
    while (done < n && grouped_all) {
      const int m = (n - done) > 32 ? 32 : (n - done);
      const int rc = (m >= 2) ? ivh_gemm256_grouped_launch(d + done, m, stream) : 1;
      if (rc < 0) return rc;
      if (rc == 1) { grouped_all = false; break; }
      done += m;
    }
    if (done == n) return 0;
    d += done; n -= done;                                // the rest is not groupable: one launch per problem
  }
  for (int i = 0; i < n; ++i) {                          // not groupable: one launch per problem
    const int rc = ivh_gemm_bf16(d + i, stream);
    if (rc) return rc;
  }
  return 0;
}
