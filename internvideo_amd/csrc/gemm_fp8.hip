// fp8 (OCP e4m3fn) MFMA GEMM for gfx950 -- BASELINE configs[4] ("InternVideo2-6B encoder ... fp8 MFMA"), SURVEY.md 8(d) "(5)".
//   C[m,n] = epi( alpha * scale_a * scale_b * sum_k A(m,k) B(n,k) ),   A, B fp8 e4m3 bytes, fp32 accumulate, bf16 / fp32 out.
//
// gfx950 has no large-K fp8 MFMA except the block-scaled MX form (guide, MFMA section: the non-scaled 16x16x32 fp8 MFMA runs at
// the bf16 rate; v_mfma_scale_f32_16x16x128_f8f6f4 is the 2x path).  Here every 32-element block scale is 2^0 (e8m0 127), so the
// instruction is a plain e4m3 x e4m3 -> fp32 MFMA with K = 128 at twice the bf16 rate, and the per-TENSOR scales produced by
// ivh_fp8_quantize multiply the accumulator once, in the epilogue (they live in device memory: no host round trip).
//
// Structure = the 128 x 128 kernel of gemm.hip with one byte per element: a K step is 128 bytes of every row, i.e. exactly the LDS
// image of the bf16 kernel's 64-element K step (128-byte rows, 16-byte chunk c of row r at slot c ^ ((r >> 1) & 7), LDS-DMA with
// the swizzle on the source address).  A lane feeds the MFMA 32 contiguous k of its row (two swizzled ds_read_b128); A and B use the
// same k -> (lane, byte) pattern, so the instruction's internal k order is irrelevant.  Both operands must be K-contiguous: dgrad /
// wgrad run on the TRANSPOSED fp8 copies that ivh_fp8_quantize writes alongside the plain ones (the usual fp8-training recipe;
// an 8-bit transposing LDS read would save those copies and is the next step).
// Epilogues as gemm.hip (bias, GELU erf / tanh, pre-activation copy, gelu' multiply), so an fp8 Linear drops into the same call sites.
#include "common.h"
#include "../../include/internvideo_hip.h"

namespace ivh {

typedef __attribute__((ext_vector_type(8))) int i32x8;
constexpr int F8_BM = 128, F8_BN = 128, F8_BK = 128;      // BK in elements = bytes
constexpr int F8_TILE = 128 * 128;                         // 16 KiB per operand per stage
constexpr float F8_MAX = 448.0f;                           // largest finite e4m3fn

struct GemmF8Params {
  const uint8_t* A; const uint8_t* B;
  long lda, ldb;
  int M, N, K;
  void* C; long ldc; int c_fp32;
  const float* bias;
  int act;
  bf16_t* preact; long ldp;
  const bf16_t* dact_in; long ldd;
  float alpha;
  const float* scale_a; const float* scale_b;
  int scale_b_vec;                                         // 1: scale_b is a vector, one scale per row of B = per output column
  int tiles_m, tiles_n;
};

__device__ __forceinline__ void f8_stage_tile(const uint8_t* __restrict__ base, long ld, int rows_total, int row0, int K, int k0,
                                              char* lds_tile, int wave, int lane) {
  const uint8_t* zero = reinterpret_cast<const uint8_t*>(g_zero_page);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = wave * 4 + j;        // 1 KiB piece = 8 rows x 128 B
    const int r = q * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    int grow = row0 + r;
    grow = grow < rows_total ? grow : rows_total - 1;
    const int k = k0 + c * 16;
    const uint8_t* src = (k < K) ? base + (long)grow * ld + k : zero;
    glds16(src, lds_tile + q * 1024);
  }
}
// 32 contiguous k (bytes) of row rbase + (lane & 15): k = 32 * (lane >> 4) ..
__device__ __forceinline__ i32x8 f8_load_frag(const char* lds_tile, int rbase, int lane) {
  const int r = rbase + (lane & 15);
  const int sw = (r >> 1) & 7;
  const int c0 = 2 * (lane >> 4);
  const u32x4 lo = *reinterpret_cast<const u32x4*>(lds_tile + r * 128 + ((c0 ^ sw) << 4));
  const u32x4 hi = *reinterpret_cast<const u32x4*>(lds_tile + r * 128 + (((c0 + 1) ^ sw) << 4));
  return i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
}

__global__ __launch_bounds__(256) void gemm_fp8_kernel(GemmF8Params p) {
  __shared__ __attribute__((aligned(16))) char lds[4 * F8_TILE];  // [stage][A|B]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nwg = p.tiles_m * p.tiles_n;
  const int id = xcd_remap(blockIdx.x, nwg);
  constexpr int G = 8;
  const int per_group = G * p.tiles_n;
  const int grp = id / per_group;
  const int first_m = grp * G;
  const int gsz = min(G, p.tiles_m - first_m);
  const int in_grp = id - grp * per_group;
  const int m0 = (first_m + in_grp % gsz) * F8_BM, n0 = (in_grp / gsz) * F8_BN;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (p.K + F8_BK - 1) / F8_BK;
  f8_stage_tile(p.A, p.lda, p.M, m0, p.K, 0, lds, wave, lane);
  f8_stage_tile(p.B, p.ldb, p.N, n0, p.K, 0, lds + F8_TILE, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    char* cur = lds + (kt & 1) * (2 * F8_TILE);
    if (kt + 1 < nk) {
      char* nxt = lds + ((kt + 1) & 1) * (2 * F8_TILE);
      f8_stage_tile(p.A, p.lda, p.M, m0, p.K, (kt + 1) * F8_BK, nxt, wave, lane);
      f8_stage_tile(p.B, p.ldb, p.N, n0, p.K, (kt + 1) * F8_BK, nxt + F8_TILE, wave, lane);
    }
    i32x8 af[4], bfr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = f8_load_frag(cur, wm * 64 + i * 16, lane);
#pragma unroll
    for (int j = 0; j < 4; ++j) bfr[j] = f8_load_frag(cur + F8_TILE, wn * 64 + j * 16, lane);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)     // format 0 = e4m3 for both operands; block scales 2^0
        acc[j][i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bfr[j], af[i], acc[j][i], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // epilogue: lane owns row m = .. + (lane & 15) and the 4 consecutive columns n = .. + 4 * (lane >> 4) + {0..3}
  const int g = lane >> 4;
  const float alpha = p.alpha * (p.scale_a ? p.scale_a[0] : 1.0f) * ((p.scale_b && !p.scale_b_vec) ? p.scale_b[0] : 1.0f);
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
      for (int r = 0; r < 4; ++r) v[r] = acc[j][i][r] * alpha;
      if (p.scale_b_vec) {
        const f32x4 sv = *reinterpret_cast<const f32x4*>(p.scale_b + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= sv[r];
      }
      if (p.bias) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bv[r];
      }
      if (p.preact) {
        if (p.act == 3) *reinterpret_cast<u32x2*>(p.preact + (long)m * p.ldp + n) = pack4(dgelu_erf(v[0]), dgelu_erf(v[1]), dgelu_erf(v[2]), dgelu_erf(v[3]));
        else *reinterpret_cast<u32x2*>(p.preact + (long)m * p.ldp + n) = pack4(v[0], v[1], v[2], v[3]);
      }
      if (p.dact_in) {
        const u32x2 uu = *reinterpret_cast<const u32x2*>(p.dact_in + (long)m * p.ldd + n);
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
      if (p.c_fp32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n) = f32x4{v[0], v[1], v[2], v[3]};
      else *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(p.C) + (long)m * p.ldc + n) = pack4(v[0], v[1], v[2], v[3]);
    }
  }
}

// ---- quantisation: bf16 [M][K] -> e4m3 [M][K] (+ transposed copy [K][ldt], ldt >= M, pad columns zero) with ONE per-tensor scale ------------
__global__ __launch_bounds__(256) void f8_amax_kernel(const bf16_t* __restrict__ x, long ld, int M, int K, unsigned* __restrict__ amax_bits) {
  // NaN / Inf must survive: fmaxf drops NaN operands, and a finite amax would quantise a NaN element to a finite value -- every fp8
  // GEMM would then launder a diverged activation and the per-step NaN guard of the engine (engine_for_pretraining.py:151-161) could
  // never fire.  |x| is tracked so that NaN sticks, and reduced on the bit patterns (non-negative floats, Inf and NaN order like
  // unsigned integers): a NaN anywhere makes amax NaN, hence the scale, hence every output of the GEMMs that use it.
  __shared__ unsigned red[256];
  float mx = 0.f;
  const int kv = K >> 3;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)M * kv; i += (long)gridDim.x * 256) {
    const long r = i / kv;
    const int c = (int)(i - r * kv);
    float f[8];
    unpack8(*reinterpret_cast<const u32x4*>(x + r * ld + c * 8), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float a = fabsf(f[e]); mx = (a > mx || a != a) ? a : mx; }     // once NaN, mx stays NaN
  }
  red[threadIdx.x] = __float_as_uint(mx) & 0x7fffffffu;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] = max(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicMax(amax_bits, red[0]);     // non-negative floats (and Inf < NaN) order like their bit patterns
}

__device__ __forceinline__ unsigned f8_pack4(float a, float b, float c, float d) {
  unsigned w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return w;
}

// one 64 x 64 tile per workgroup: rows written straight (16 bytes per lane), the transposed copy through LDS
// amax_next != NULL (delayed scaling): amax_bits is LAST step's amax of this call site; this call's own max|x| is collected on the way
// (one atomicMax per wave) for the next step, so the tensor is read once instead of twice.  Values beyond the stale range saturate at +-448.
__global__ __launch_bounds__(256) void f8_quantize_kernel(const bf16_t* __restrict__ x, long ld, int M, int K, const unsigned* __restrict__ amax_bits,
                                                          uint8_t* __restrict__ q, long ldq, uint8_t* __restrict__ qt, long ldt, int Mt,
                                                          float* __restrict__ scale_out, unsigned* __restrict__ amax_next) {
  __shared__ __attribute__((aligned(16))) uint8_t tile[64][80];           // [row][k] e4m3, 16-byte aligned rows
  const float amax_raw = __uint_as_float(amax_bits[0]);
  const float amax = (amax_raw != amax_raw) ? amax_raw : fmaxf(amax_raw, 1e-12f);      // NaN stays NaN (Inf stays Inf): the scale poisons the product
  const float scale = amax / F8_MAX;                   // dequantisation multiplier
  const float inv = F8_MAX / amax;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) scale_out[0] = scale;
  const int r0 = blockIdx.y * 64, k0 = blockIdx.x * 64;
  const int tr = threadIdx.x >> 2, tc = (threadIdx.x & 3) * 16;        // row 0..63, 16 consecutive k
  const int row = r0 + tr, kk = k0 + tc;
  unsigned w[4] = {0u, 0u, 0u, 0u};
  float mx = 0.f;
  if (row < M) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (kk + 8 * h < K) {
        float f[8];
        unpack8(*reinterpret_cast<const u32x4*>(x + (long)row * ld + kk + 8 * h), f);
        if (amax_next) {
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float a = fabsf(f[e]); mx = (a > mx || a != a) ? a : mx; }     // NaN sticks (see f8_amax_kernel)
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = fminf(fmaxf(f[e] * inv, -F8_MAX), F8_MAX);
        w[2 * h] = f8_pack4(f[0], f[1], f[2], f[3]);
        w[2 * h + 1] = f8_pack4(f[4], f[5], f[6], f[7]);
      }
    }
    if (kk + 16 <= K) *reinterpret_cast<u32x4*>(q + (long)row * ldq + kk) = u32x4{w[0], w[1], w[2], w[3]};
    else if (kk + 8 <= K) *reinterpret_cast<u32x2*>(q + (long)row * ldq + kk) = u32x2{w[0], w[1]};
  }
  if (amax_next) {                                                       // wave max on the bit patterns, one atomic per wave
    unsigned mb = __float_as_uint(mx) & 0x7fffffffu;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mb = max(mb, (unsigned)__shfl_xor((int)mb, o, 64));
    // tens of thousands of waves would otherwise queue on ONE address (measured: the 6B step 390 -> 686 ms): look first (a relaxed
    // device-scope load; a stale value only costs a redundant atomic), and after the first few waves almost nobody needs to write
    if ((threadIdx.x & 63) == 0 && mb > __hip_atomic_load(amax_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(amax_next, mb);
  }
  if (qt) {
    *reinterpret_cast<u32x4*>(&tile[tr][tc]) = u32x4{w[0], w[1], w[2], w[3]};
    __syncthreads();
    // transposed write: thread -> k column tk, 16 consecutive rows
    const int tk = threadIdx.x >> 2, tm = (threadIdx.x & 3) * 16;
    if (k0 + tk < K && r0 + tm < Mt) {
      unsigned o[4];
#pragma unroll
      for (int h = 0; h < 4; ++h)
        o[h] = (unsigned)tile[tm + 4 * h][tk] | ((unsigned)tile[tm + 4 * h + 1][tk] << 8) | ((unsigned)tile[tm + 4 * h + 2][tk] << 16) |
               ((unsigned)tile[tm + 4 * h + 3][tk] << 24);
      *reinterpret_cast<u32x4*>(qt + (long)(k0 + tk) * ldt + r0 + tm) = u32x4{o[0], o[1], o[2], o[3]};   // rows >= M of the tile are zeros
    }
  }
}

// ---- weights: one scale per OUTPUT channel of each GEMM that reads them ----------------------------------------------------------------
// W bf16 [N][K] feeds two GEMMs: forward y = x W^T (B operand = W, output column n) and dgrad dx = dy W (B operand = W^T, output column
// k).  A scale can leave the contraction only along the OUTPUT dimension, so the plain copy is scaled per row n and the transposed copy per
// column k of W: two amax vectors from one pass, two independently rounded e4m3 images from the second.
__global__ __launch_bounds__(256) void f8_rowcol_amax_kernel(const bf16_t* __restrict__ x, long ld, int M, int K, unsigned* __restrict__ amax_rows,
                                                             unsigned* __restrict__ amax_cols) {
  __shared__ unsigned colmx[4][64];                                       // [16-row group][column of the tile]
  const int r0 = blockIdx.y * 64, k0 = blockIdx.x * 64;
  const int tr = threadIdx.x >> 2, tc = (threadIdx.x & 3) * 16;
  const int row = r0 + tr, kk = k0 + tc;
  float f[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) f[e] = 0.f;
  if (row < M) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
      if (kk + 8 * h < K) unpack8(*reinterpret_cast<const u32x4*>(x + (long)row * ld + kk + 8 * h), f + 8 * h);
  }
  unsigned rb = 0u, cb[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) { cb[e] = __float_as_uint(f[e]) & 0x7fffffffu; rb = max(rb, cb[e]); }   // bit patterns: NaN / Inf stick
  rb = max(rb, (unsigned)__shfl_xor((int)rb, 1, 64));                    // the 4 threads of a row are neighbouring lanes
  rb = max(rb, (unsigned)__shfl_xor((int)rb, 2, 64));
  if ((threadIdx.x & 3) == 0 && row < M && rb) atomicMax(amax_rows + row, rb);
#pragma unroll
  for (int e = 0; e < 16; ++e) {                                          // column max over the 16 rows of this wave (lanes 4 apart)
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) cb[e] = max(cb[e], (unsigned)__shfl_xor((int)cb[e], o, 64));
  }
  if ((threadIdx.x & 63) < 4) {
#pragma unroll
    for (int e = 0; e < 16; ++e) colmx[threadIdx.x >> 6][tc + e] = cb[e];
  }
  __syncthreads();
  if (threadIdx.x < 64 && k0 + threadIdx.x < K) {
    const unsigned m4 = max(max(colmx[0][threadIdx.x], colmx[1][threadIdx.x]), max(colmx[2][threadIdx.x], colmx[3][threadIdx.x]));
    if (m4) atomicMax(amax_cols + k0 + threadIdx.x, m4);
  }
}

__device__ __forceinline__ void f8_scale_pair(unsigned bits, float& scale, float& inv) {
  const float raw = __uint_as_float(bits);
  const float amax = (raw != raw) ? raw : fmaxf(raw, 1e-12f);            // an all-zero channel quantises to zeros with a tiny scale
  scale = amax / F8_MAX; inv = F8_MAX / amax;
}

__global__ __launch_bounds__(256) void f8_quantize_rowcol_kernel(const bf16_t* __restrict__ x, long ld, int M, int K, const unsigned* __restrict__ amax_rows,
                                                                 const unsigned* __restrict__ amax_cols, uint8_t* __restrict__ q, long ldq,
                                                                 uint8_t* __restrict__ qt, long ldt, int Mt, float* __restrict__ scale_rows,
                                                                 float* __restrict__ scale_cols) {
  __shared__ __attribute__((aligned(16))) uint8_t tile[64][80];
  const int r0 = blockIdx.y * 64, k0 = blockIdx.x * 64;
  const int tr = threadIdx.x >> 2, tc = (threadIdx.x & 3) * 16;
  const int row = r0 + tr, kk = k0 + tc;
  float srow = 0.f, irow = 0.f;
  if (row < M) f8_scale_pair(amax_rows[row], srow, irow);
  if (blockIdx.x == 0 && (threadIdx.x & 3) == 0 && row < M) scale_rows[row] = srow;
  if (blockIdx.y == 0 && threadIdx.x < 64 && k0 + threadIdx.x < K) { float sc, ic; f8_scale_pair(amax_cols[k0 + threadIdx.x], sc, ic); scale_cols[k0 + threadIdx.x] = sc; }
  unsigned w[4] = {0u, 0u, 0u, 0u}, wt[4] = {0u, 0u, 0u, 0u};
  if (row < M) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (kk + 8 * h < K) {
        float f[8], g[8];
        unpack8(*reinterpret_cast<const u32x4*>(x + (long)row * ld + kk + 8 * h), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float sc, ic;
          f8_scale_pair(amax_cols[kk + 8 * h + e], sc, ic);
          g[e] = fminf(fmaxf(f[e] * ic, -F8_MAX), F8_MAX);
          f[e] = fminf(fmaxf(f[e] * irow, -F8_MAX), F8_MAX);
        }
        w[2 * h] = f8_pack4(f[0], f[1], f[2], f[3]);  w[2 * h + 1] = f8_pack4(f[4], f[5], f[6], f[7]);
        wt[2 * h] = f8_pack4(g[0], g[1], g[2], g[3]); wt[2 * h + 1] = f8_pack4(g[4], g[5], g[6], g[7]);
      }
    }
    if (kk + 16 <= K) *reinterpret_cast<u32x4*>(q + (long)row * ldq + kk) = u32x4{w[0], w[1], w[2], w[3]};
    else if (kk + 8 <= K) *reinterpret_cast<u32x2*>(q + (long)row * ldq + kk) = u32x2{w[0], w[1]};
  }
  *reinterpret_cast<u32x4*>(&tile[tr][tc]) = u32x4{wt[0], wt[1], wt[2], wt[3]};
  __syncthreads();
  const int tk = threadIdx.x >> 2, tm = (threadIdx.x & 3) * 16;
  if (k0 + tk < K && r0 + tm < Mt) {
    unsigned o[4];
#pragma unroll
    for (int h = 0; h < 4; ++h)
      o[h] = (unsigned)tile[tm + 4 * h][tk] | ((unsigned)tile[tm + 4 * h + 1][tk] << 8) | ((unsigned)tile[tm + 4 * h + 2][tk] << 16) |
             ((unsigned)tile[tm + 4 * h + 3][tk] << 24);
    *reinterpret_cast<u32x4*>(qt + (long)(k0 + tk) * ldt + r0 + tm) = u32x4{o[0], o[1], o[2], o[3]};
  }
}

}  // namespace ivh

using namespace ivh;

extern "C" int ivh_fp8_quantize(const uint16_t* x, int64_t ld, int M, int K, uint8_t* q, int64_t ldq, uint8_t* qt, int64_t ldt,
                                float* scale_out, uint32_t* amax_scratch, void* stream) {
  IVH_REQUIRE(x && q && scale_out && amax_scratch && M > 0 && K > 0, "fp8_quantize: bad args");
  IVH_REQUIRE(K % 8 == 0 && ld % 8 == 0 && ldq % 16 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)q % 16) == 0,
              "fp8_quantize: K, ld multiples of 8, ldq multiple of 16, 16-byte aligned buffers");
  IVH_REQUIRE(!qt || (ldt % 16 == 0 && ldt >= ((M + 15) / 16) * 16 && ((uintptr_t)qt % 16) == 0),
              "fp8_quantize: transposed copy needs ldt >= M rounded up to 16 (pad columns are zero-filled) and 16-byte alignment");
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(amax_scratch, 0, 4, s) != hipSuccess) { ivh_host::set_error("fp8_quantize: memset failed"); return -2; }
  long blocks = ((long)M * (K / 8) + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(f8_amax_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, (long)ld, M, K, amax_scratch);
  const int Mt = qt ? (int)(((M + 15) / 16) * 16) : 0;
  dim3 grid((K + 63) / 64, (M + 63) / 64, 1);
  hipLaunchKernelGGL(f8_quantize_kernel, grid, dim3(256), 0, s, x, (long)ld, M, K, amax_scratch, q, (long)ldq, qt, (long)ldt, Mt, scale_out, (unsigned*)nullptr);
  return ivh_host::check_launch("fp8_quantize");
}

extern "C" int ivh_fp8_quantize_delayed(const uint16_t* x, int64_t ld, int M, int K, uint8_t* q, int64_t ldq, uint8_t* qt, int64_t ldt,
                                        const float* amax_prev, float* scale_out, uint32_t* amax_next, void* stream) {
  IVH_REQUIRE(x && q && scale_out && amax_prev && amax_next && M > 0 && K > 0, "fp8_quantize_delayed: bad args");
  IVH_REQUIRE(K % 8 == 0 && ld % 8 == 0 && ldq % 16 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)q % 16) == 0,
              "fp8_quantize_delayed: K, ld multiples of 8, ldq multiple of 16, 16-byte aligned buffers");
  IVH_REQUIRE(!qt || (ldt % 16 == 0 && ldt >= ((M + 15) / 16) * 16 && ((uintptr_t)qt % 16) == 0),
              "fp8_quantize_delayed: transposed copy needs ldt >= M rounded up to 16 (pad columns are zero-filled) and 16-byte alignment");
  const int Mt = qt ? (int)(((M + 15) / 16) * 16) : 0;
  dim3 grid((K + 63) / 64, (M + 63) / 64, 1);
  hipLaunchKernelGGL(f8_quantize_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, (long)ld, M, K, reinterpret_cast<const unsigned*>(amax_prev), q, (long)ldq, qt,
                     (long)ldt, Mt, scale_out, amax_next);
  return ivh_host::check_launch("fp8_quantize_delayed");
}

extern "C" int ivh_fp8_quantize_weight(const uint16_t* w, int64_t ld, int N, int K, uint8_t* q, int64_t ldq, uint8_t* qt, int64_t ldt,
                                       float* scale_rows, float* scale_cols, uint32_t* amax_scratch, void* stream) {
  IVH_REQUIRE(w && q && qt && scale_rows && scale_cols && amax_scratch && N > 0 && K > 0, "fp8_quantize_weight: bad args");
  IVH_REQUIRE(K % 8 == 0 && ld % 8 == 0 && ldq % 16 == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)q % 16) == 0,
              "fp8_quantize_weight: K, ld multiples of 8, ldq multiple of 16, 16-byte aligned buffers");
  IVH_REQUIRE(ldt % 16 == 0 && ldt >= ((N + 15) / 16) * 16 && ((uintptr_t)qt % 16) == 0,
              "fp8_quantize_weight: transposed copy needs ldt >= N rounded up to 16 (pad columns are zero-filled) and 16-byte alignment");
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(amax_scratch, 0, (size_t)(N + K) * 4, s) != hipSuccess) { ivh_host::set_error("fp8_quantize_weight: memset failed"); return -2; }
  dim3 grid((K + 63) / 64, (N + 63) / 64, 1);
  hipLaunchKernelGGL(f8_rowcol_amax_kernel, grid, dim3(256), 0, s, w, (long)ld, N, K, amax_scratch, amax_scratch + N);
  hipLaunchKernelGGL(f8_quantize_rowcol_kernel, grid, dim3(256), 0, s, w, (long)ld, N, K, amax_scratch, amax_scratch + N, q, (long)ldq, qt, (long)ldt,
                     (int)(((N + 15) / 16) * 16), scale_rows, scale_cols);
  return ivh_host::check_launch("fp8_quantize_weight");
}

extern "C" int ivh_gemm256_fp8_launch(const ivh_gemm_desc* d, const float* scale_a, const float* scale_b, int scale_b_vec, void* stream);   // gemm256.hip
static int g_f8_kernel = 0;                               // 0 = per problem (256^2 when it applies), 1 = always the 128^2 kernel (A/B, tests)
extern "C" int ivh_set_gemm_fp8_kernel(int choice) { g_f8_kernel = choice == 1 ? 1 : 0; return 0; }

extern "C" int64_t ivh_gemm256_split_ws_bytes(const ivh_gemm_desc* d, int fp8);      // gemm256.hip
// bytes of d->split_ws with which ivh_gemm_fp8 would cut the tiles of a mostly empty last round into K slices (0 = no use)
extern "C" int64_t ivh_gemm_fp8_split_workspace(const ivh_gemm_desc* d) {
  if (!d || g_f8_kernel == 1 || !d->a_kc || !d->b_kc || (long)d->M * d->N < 512L * 512 || d->K < 512) return 0;
  return ivh_gemm256_split_ws_bytes(d, 1);
}

static int gemm_fp8_impl(const ivh_gemm_desc* d, const float* scale_a, const float* scale_b, int scale_b_vec, void* stream);
extern "C" int ivh_gemm_fp8(const ivh_gemm_desc* d, const float* scale_a, const float* scale_b, void* stream) {
  return gemm_fp8_impl(d, scale_a, scale_b, 0, stream);
}
extern "C" int ivh_gemm_fp8_cs(const ivh_gemm_desc* d, const float* scale_a, const float* scale_b_cols, void* stream) {
  IVH_REQUIRE(scale_a && scale_b_cols && ((uintptr_t)scale_b_cols % 16) == 0, "gemm_fp8_cs: scale_a and a 16-byte aligned vector of d->N column scales");
  return gemm_fp8_impl(d, scale_a, scale_b_cols, 1, stream);
}
static int gemm_fp8_impl(const ivh_gemm_desc* d, const float* scale_a, const float* scale_b, int scale_b_vec, void* stream) {
  IVH_REQUIRE(d && d->A && d->B && d->C, "gemm_fp8: null operand");
  IVH_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "gemm_fp8: empty problem M=%d N=%d K=%d", d->M, d->N, d->K);
  IVH_REQUIRE(d->a_kc && d->b_kc, "gemm_fp8: both operands must be K-contiguous (use the transposed copies of ivh_fp8_quantize for dgrad / wgrad)");
  IVH_REQUIRE(d->K % 16 == 0 && d->lda % 16 == 0 && d->ldb % 16 == 0, "gemm_fp8: K, lda, ldb must be multiples of 16 (bytes)");
  IVH_REQUIRE(d->N % 8 == 0 && d->ldc % 4 == 0, "gemm_fp8: N multiple of 8, ldc multiple of 4");
  IVH_REQUIRE(((uintptr_t)d->A % 16) == 0 && ((uintptr_t)d->B % 16) == 0 && ((uintptr_t)d->C % 16) == 0, "gemm_fp8: base pointers must be 16-byte aligned");
  IVH_REQUIRE(d->act >= 0 && d->act <= 3 && (d->batch <= 1) && !d->colsum_part, "gemm_fp8: unsupported activation / batch / colsum request");
  if (g_f8_kernel != 1) {                                  // large problems: the persistent 256 x 256 ping-pong kernel (gemm256.hip, FP8 flavour)
    IVH_REQUIRE(scale_a && scale_b, "gemm_fp8: null scale");
    const int rc = ivh_gemm256_fp8_launch(d, scale_a, scale_b, scale_b_vec, stream);
    if (rc <= 0) return rc;
  }
  GemmF8Params p;
  p.A = reinterpret_cast<const uint8_t*>(d->A); p.B = reinterpret_cast<const uint8_t*>(d->B);
  p.lda = d->lda; p.ldb = d->ldb; p.M = d->M; p.N = d->N; p.K = d->K;
  p.C = d->C; p.ldc = d->ldc; p.c_fp32 = d->c_fp32; p.bias = d->bias; p.act = d->act;
  p.preact = d->preact; p.ldp = d->ldp; p.dact_in = d->dact_in; p.ldd = d->ldd;
  p.alpha = d->alpha; p.scale_a = scale_a; p.scale_b = scale_b; p.scale_b_vec = scale_b_vec;
  p.tiles_m = (d->M + F8_BM - 1) / F8_BM; p.tiles_n = (d->N + F8_BN - 1) / F8_BN;
  dim3 grid(p.tiles_m * p.tiles_n, 1, 1), block(256);
  hipLaunchKernelGGL(gemm_fp8_kernel, grid, block, 0, (hipStream_t)stream, p);
  return ivh_host::check_launch("gemm_fp8");
}
