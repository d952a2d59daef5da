// bf16 MFMA GEMM for gfx950, 256 x 256 x 64 tile, 8 waves, LDS-DMA ring with counted vmcnt and a two-group ping-pong.
//   C[m,n] = epi( alpha * sum_k A(m,k) B(n,k) ),  fp32 accumulate          (same contract as gemm.hip / ivh_gemm_desc)
//
// Why a second GEMM: the 128^2 / 4-wave kernel of gemm.hip drains its LDS-DMA queue (vmcnt(0)) at the one barrier of every
// K step and sits at 430-660 TFLOP/s inside the 1B training step; 75 % of the step is GEMM.  This kernel keeps the DMA queue
// 4 pieces deep across barriers and alternates two wave groups between "matrix" and "memory" segments so that every SIMD
// always has one wave issuing MFMAs.
//
// Geometry.  512 threads = 8 waves = 2 (M) x 4 (N); wave (wm, wn) owns C rows wm*128..+128, cols wn*64..+64 = 8 x 4 MFMA
// 16x16x32 tiles (128 accumulator VGPRs).  A K step (64) is consumed in four phases, one C quadrant (64 x 32 per wave, 16 MFMAs)
// per phase, in the order (mh,nh) = (0,0) (0,1) (1,1) (1,0) so that each phase needs exactly one fresh operand piece:
//     piece type 0  Alo : rows {wm*128 + 0..63}   of A   (both groups: 128 rows x 64 k = 16 KiB)     first read in phase 0
//                1  Blo : cols {wn*64 + 0..31}    of B   (all four wn)                              phase 0 (kept in VGPRs for phase 3)
//                2  Bhi : cols {wn*64 + 32..63}                                                      phase 1
//                3  Ahi : rows {wm*128 + 64..127}                                                     phase 2
// Piece s = 4*ktile + type lives in LDS slot s & 7 (8 x 16 KiB = 128 KiB, one workgroup per CU).
//
// Pipeline (p = global phase counter = 4*ktile + j).  Phase p:   [ds_read fragments of phase p] [issue LDS-DMA of piece p+6]
//   [s_waitcnt vmcnt(8): pieces <= p+2 have landed] [s_barrier] [16 MFMA behind the compiler's counted lgkmcnt] [s_barrier].
//   RAW: the reads of phase p touch pieces <= p+1, waited for (by every wave) in phase p-1 before a barrier.
//   WAR: piece p+6 overwrites the slot of piece p-2, last read in phase <= p-2, i.e. two barriers earlier for both groups.
//   The DMA queue is never drained inside the loop: 8-10 x 1 KiB requests per wave stay in flight across the barriers.
// Ping-pong: group 1 (waves 4-7, wm = 1) executes one extra s_barrier up front, so between any two barriers one group is in
// its MFMA segment while the other issues its ds_reads / DMA; the two waves sharing a SIMD are one of each.
// K tails / ghost K steps / rows past the matrix read zeros through the buffer descriptor's bounds check (or an explicit
// out-of-range offset for a K-contiguous operand), so the loop has no tail code.
//
// Operand layouts as in gemm.hip: K-contiguous operands are staged [128 rows][64 k] with the 16-byte chunk XOR swizzle applied
// on the DMA source address and on the ds_read_b128; rows-contiguous operands (dgrad B, wgrad A and B) are staged as they lie,
// [64 k][128 rows], and transposed by ds_read_b64_tr_b16.
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "../../include/internvideo_hip.h"
#include "../../include/internvideo_hip_debug.h"

namespace ivh {

typedef __attribute__((address_space(3))) void lds_void_t;

constexpr int G2_BM = 256, G2_BN = 256, G2_BK = 64;
constexpr int G2_PIECE = 16384;
constexpr unsigned G2_OOB = 0x80000000u;      // byte offset beyond any descriptor range (buffers are < 2 GiB): reads 0

struct G2Prob {                                // one problem of a grouped launch (same K, layouts, plain bf16 epilogue)
  const bf16_t* A; const bf16_t* B; void* C;
  long a_bytes, b_bytes, c_bytes;
  int lda, ldb, ldc, M, N, tiles_m, tiles_n, tile_begin;
  const int* k_dev;                            // DYN kernels: this problem's contraction length lives in device memory (<= its K); NULL = K
  int K;                                       // DYN kernels: this problem's own contraction length (problems of one launch may differ: the
};                                             // decoders' weight gradients with and without the cls row); the launch's K is the largest

struct Gemm256Params {
  const bf16_t* A; const bf16_t* B;
  int lda, ldb;                                // leading dimensions fit 31 bits (operands are < 2 GiB): keeps SGPR pressure down
  int M, N, K;
  void* C; int ldc; int c_fp32;
  const float* bias;
  int act;
  bf16_t* preact; int ldp;
  const bf16_t* dact_in; int ldd;
  float alpha;
  int tiles_m, tiles_n, batch;
  long a_bytes, b_bytes;                       // extent of the whole (batched) operand: descriptor range
  long c_bytes, p_bytes, d_bytes, bias_bytes;  // same for C, preact, dact_in, bias (0 when absent)
  int stagger;                                 // start-up skew: workgroup w sleeps (w % 16) * stagger * ~0.5 us (0 = off)
  int debug_skip_stores;                       // measurement aid (tools/bench_gemm.py): drop every C / preact store
  unsigned long long* debug_stamps;            // measurement aid: s_memtime stamps of one workgroup / waves 0 and 4 (or NULL)
  int debug_stamp_wg;                          // ... which workgroup (blockIdx.x; env IVH_G2_STAMP_WG, default 0)
  float* colsum_part;                          // EPI 3: fp32 [2 * tiles_m][N] column sums of C per 128-row block (or NULL)
  const float* scale_a; const float* scale_b;  // FP8: per-tensor scales of the e4m3 operands (device scalars), folded into alpha
  int scale_b_vec;                             // FP8: 1 = scale_b is a vector of N scales, one per row of B (= output column)
  int total_tiles;                             // tiles_m * tiles_n * batch, or the sum over the problems of a grouped launch
  int nprob;                                   // GROUPED kernels: number of valid entries of prob[]
  G2Prob prob[32];
  long strideA, strideB, strideC, stride_bias, stride_preact, stride_dact;
  // SPLIT kernels (tail split along K, see gemm256_kernel): linear ids [0, split_main) are whole tiles, id split_main + u is K slice
  // u % split_s of tile split_main + u / split_s; total_tiles counts ids.  split_nk2 = loop trips (2 K steps each) per slice.
  int split_main, split_s, split_nk2;
  // HALF kernels (half-width tiles, see gemm256_kernel): linear ids [0, half_begin) are whole tiles of the tiles_m x tiles_nf grid of
  // full-width column tiles; the next half_split ids are the two 128-column halves of the whole tiles half_begin + j / 2 of that grid
  // (the leftover tiles of the last round, cut so that no workgroup carries a whole extra tile); the rest, one per row tile, are the
  // N-edge tiles of an output whose last column tile is at most 128 wide.  total_tiles counts ids.
  int half_begin, half_split, tiles_nf;
  int half_interleave;                         // 1 = a workgroup runs its half tile between its whole tiles (position by XCD block), 0 = last
  float* split_ws;                             // [units][32][512] f32x4: the accumulators of every slice, fragment layout
  long split_ws_bytes;                         // extent of split_ws (descriptor range)
  unsigned* split_cnt;                         // [tail tiles] arrival counters, zero at launch
  // DYN kernels (row counts known only on the device: the kept samples of a DropPath branch, see ivh_gemm_desc.m_dev / k_dev):
  // *m_dev <= M replaces M (rows of a K-contiguous A, of C and of the epilogue operands), *k_dev <= K replaces K (rows-contiguous operands:
  // the token axis of a weight gradient).  The launch is sized for M / K; tiles past the device count are never started.
  const int* m_dev; const int* k_dev;
};

__device__ __forceinline__ int g2_swz(int kr) { return ((kr & 3) << 1) | (((kr >> 3) & 1) << 3); }

// Per-lane byte offset (relative to the operand base, K offset excluded) of the 16 bytes this lane feeds to LDS-DMA
// request j (0/1) of a piece; `hi` selects the lo/hi piece.  KC: also returns the lane's k chunk start for the K mask.
// IDENT: piece row r is operand row r (a half-width tile's one B piece holds 128 CONSECUTIVE columns, 32 per wave column).
template <bool KC, bool IS_A, int ES = 2, bool IDENT = false>   // ES = bytes per element (2: bf16, 1: e4m3; a K step is 128 bytes of a row either way)
__device__ __forceinline__ unsigned g2_piece_voff(int lane, int wave, int j, int hi, int ld, int row0, int& kchunk) {
  const int q = wave * 2 + j;                       // 1 KiB request index inside the 16 KiB piece
  if constexpr (KC) {
    const int r = q * 8 + (lane >> 3);              // piece row 0..127
    const int c = (lane & 7) ^ ((r >> 1) & 7);      // source chunk that lands in slot (lane & 7) of that row
    const int trow = IDENT ? r : (IS_A ? ((r >> 6) * 128 + (r & 63) + hi * 64) : ((r >> 5) * 64 + (r & 31) + hi * 32));
    kchunk = c * (16 / ES);
    return (unsigned)((long)(row0 + trow) * ld * ES + c * 16);
  } else {
    const int kr = q * 4 + (lane >> 4);             // k row 0..63 of the piece
    const int c = (lane & 15) ^ g2_swz(kr);
    const int pr = c * 8;                           // piece row of the first of 8 consecutive rows
    const int trow = IDENT ? pr : (IS_A ? ((pr >> 6) * 128 + (pr & 63) + hi * 64) : ((pr >> 5) * 64 + (pr & 31) + hi * 32));
    kchunk = kr;
    return (unsigned)(((long)kr * ld + row0 + trow) * 2);
  }
}

struct G2Stage {                                    // everything a wave needs to issue its 2 DMA requests of any piece
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned voff[2];                                 // [j] per-lane byte offset inside the lo piece of tile (0, 0)
  unsigned hi_off;                                  // scalar: lo piece -> hi piece
  unsigned toff;                                    // scalar: byte offset of the tile being issued
  int kchunk[2];                                    // [j] first k index of the lane's 16 bytes (KC: chunk start, else k row)
};

// `wave_lds` = LDS address of the wave's first 1 KiB request slot inside piece slot 0 (scalar, re-derived per loop trip so
// that the 16 distinct M0 values are one s_add each instead of 16 live SGPRs).
template <bool KC>
__device__ __forceinline__ void g2_issue(const G2Stage& st, int hi, unsigned kbyte, int krem, char* wave_lds, int slot) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    unsigned v = st.voff[j] + (kbyte + st.toff + (hi ? st.hi_off : 0u));
    v = (st.kchunk[j] < krem) ? v : G2_OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(st.rsrc, (lds_void_t*)(wave_lds + slot * G2_PIECE + j * 1024), 16, v, 0, 0, 0);
  }
}

// One LDS-DMA request outside the compiler's view (half-width tiles): 64 lanes x 16 bytes -> LDS [lds_dst, + 1 KiB).  hipcc neither counts it
// nor orders LDS reads against it -- the half-tile loop addresses its ring slots at run time, which the builtin's alias analysis would
// answer with s_waitcnt vmcnt(0) in front of every fragment read -- so every wait of that loop is written by hand.
__device__ __forceinline__ void g2_dma16_asm(u32x4 rs, unsigned lds_dst, unsigned voff) {
  // M0 (the DMA's LDS base) is handed over as a "{m0}"-constrained INPUT: the compiler writes it itself and therefore knows it changed --
  // this kernel also uses the compiler-managed LDS-DMA builtins, which share M0 (an undeclared write could be merged away)
  asm volatile("s_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(voff), "s"(rs), "{m0}"(lds_dst) : "memory");
}
__device__ __forceinline__ u32x4 g2_rsrc4(const void* base, long bytes) {
  const unsigned long long a = (unsigned long long)base;
  return u32x4{(unsigned)a, (unsigned)(a >> 32) & 0xffffu, (unsigned)bytes, 0x00020000u};
}

// MFMA operand fragment (16 rows x 32 k) of piece rows rbase..rbase+15, k half kk, out of an LDS piece.
// MFMA operand fragment (16 rows x 32 k).  `lane_base` = LDS byte address of the lane's first 16 bytes for (k half, wave row
// base) -- an opaque per-lane value -- and `cst` = slot * 16 KiB + tile row offset, a compile-time constant that folds into the
// ds_read offset field.  Transposing reads: the lane's other 4-row group is + 1024 bytes and the other k half + 8192 (the
// swizzle term only depends on k & 3 and (k >> 3) & 1, which + 4 and + 32 preserve); the tile's row-chunk bits are XORed into
// the lane base by the caller (one base register per 16-row tile).
template <bool KC>
__device__ __forceinline__ s16x8 g2_frag(const char* lds, unsigned lane_base, int cst) {
  if constexpr (KC) {
    return *reinterpret_cast<const s16x8*>(lds + lane_base + cst);
  } else {
    // inline asm, not the builtin: the compiler cannot prove that a transposing read does not alias the LDS-DMA writes in
    // flight and would drain them (s_waitcnt vmcnt(0)) before every such read.  The asm is invisible to its waitcnt pass, so
    // the phase code waits lgkmcnt(0) itself before the MFMAs (G2_SEG_BEGIN).
    (void)lds;
    s16x4 t0, t1;
    const unsigned base = lane_base + (unsigned)(cst & ~0xFFFF);        // the DS offset field holds 16 bits: slots 4..7 use base + 64 KiB
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(t0) : "v"(base), "n"(cst & 0xFFFF));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(t1) : "v"(base), "n"((cst & 0xFFFF) + 1024));
    s16x8 r;
    r[0] = t0[0]; r[1] = t0[1]; r[2] = t0[2]; r[3] = t0[3];
    r[4] = t1[0]; r[5] = t1[1]; r[6] = t1[2]; r[7] = t1[3];
    return r;
  }
}

typedef __attribute__((ext_vector_type(8))) int g2_i32x8;
// e4m3 fragment: 32 consecutive bytes of the lane's row = the two 16-byte chunks whose (swizzled) addresses differ in bit 4
__device__ __forceinline__ g2_i32x8 g2_frag8(const char* lds, unsigned base_lo, unsigned base_hi, int cst) {
  const u32x4 lo = *reinterpret_cast<const u32x4*>(lds + base_lo + cst);
  const u32x4 hi = *reinterpret_cast<const u32x4*>(lds + base_hi + cst);
  return g2_i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
}
template <bool KC, typename T>
__device__ __forceinline__ T g2_frag_t(const char* lds, unsigned lane_base, int cst) {
  if constexpr (std::is_same_v<T, s16x8>) return g2_frag<KC>(lds, lane_base, cst);
  else return T{};
}
__device__ __forceinline__ f32x4 g2_mma(s16x8 a, s16x8 b, f32x4 c) { return mfma16(a, b, c); }
__device__ __forceinline__ f32x4 g2_mma(g2_i32x8 a, g2_i32x8 b, f32x4 c) {     // format 0 = e4m3 for both operands, block scales 2^0
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
}

#define G2_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// erf via Abramowitz-Stegun 7.1.26 (|error| < 1.5e-7, far below the bf16 output step): 1 rcp + 1 exp + 7 fma instead of libm's
// ~40-instruction erff.  The epilogue evaluates it 128 times per lane and tile, so it is both the VALU time and -- with libm's
// version inlined 128 times -- the instruction-cache footprint of the tile boundary.
__device__ __forceinline__ float g2_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __expf(-ax * ax);
  const float r = fmaf(-p * t, e, 1.0f);
  return copysignf(r, x);
}
// gelu(x) and gelu'(x) from ONE erf / exp evaluation: Phi = (1 + erf(x / sqrt 2)) / 2, e = exp(-x^2 / 2);  g = x Phi,  d = Phi + x e / sqrt(2 pi)
__device__ __forceinline__ void g2_gelu_pair(float x, float& g, float& d) {
  const float ax = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __expf(-ax * ax);                      // = exp(-x^2 / 2)
  const float erfv = copysignf(fmaf(-p * t, e, 1.0f), x);
  const float phi = fmaf(0.5f, erfv, 0.5f);
  g = x * phi;
  d = fmaf(x * 0.3989422804014327f, e, phi);
}
__device__ __forceinline__ float g2_gelu(float x) { return 0.5f * x * (1.0f + g2_erf(x * 0.70710678118654752f)); }
__device__ __forceinline__ float g2_dgelu(float x) {
  const float cdf = 0.5f * (1.0f + g2_erf(x * 0.70710678118654752f));
  return fmaf(x * 0.3989422804014327f, __expf(-0.5f * x * x), cdf);
}

// sum over the 16 lanes of a DPP row (lanes with the same lane >> 4): four row rotations (by 8, 4, 2, 1) folded into the adds as DPP modifiers --
// no LDS round trip; every lane ends with the total (a fixed order per lane: bitwise reproducible)
__device__ __forceinline__ float g2_row16_sum(float x) {
#define G2_ROR(v, n) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + (n), 0xf, 0xf, false))
  x += G2_ROR(x, 8);
  x += G2_ROR(x, 4);
  x += G2_ROR(x, 2);
  x += G2_ROR(x, 1);
#undef G2_ROR
  return x;
}

struct G2Tile { int z, m0, n0; };

// linear tile id -> (batch, m0, n0): groups of 8 m-tiles walked n-major so that 8 x 4 neighbouring tiles share an XCD's L2
__device__ __forceinline__ G2Tile g2_decode(int lin, int tiles_m, int tiles_n) {
  const int per_z = tiles_m * tiles_n;
  const int z = lin / per_z;
  const int id = lin - z * per_z;
  constexpr int G = 8;
  const int per_group = G * tiles_n;
  const int grp = id / per_group;
  const int first_m = grp * G;
  const int gsz = min(G, tiles_m - first_m);
  const int in_grp = id - grp * per_group;
  return G2Tile{z, (first_m + in_grp % gsz) * G2_BM, (in_grp / gsz) * G2_BN};
}

// Persistent kernel: gridDim.x = min(#tiles, #CUs) workgroups; workgroup w computes tiles w', w' + grid, ... (w' = XCD-contiguous
// remap of w).  The LDS-DMA stream never stops at a tile boundary: the last six phases of a tile already fetch the first six
// pieces of the next one, the C tile is written with fire-and-forget stores, and the first three phases of the next tile skip
// their vmcnt wait (their pieces were waited for at the end of the previous tile), so the stores have ~3 phases to retire
// before a counted vmcnt can see them.
// EPI = 0: C = alpha * acc + bias;  EPI = 2: C = gelu_erf(alpha * acc + bias) with optional pre-activation copy;
// EPI = 1: C = (alpha * acc + bias) * gelu_erf'(dact_in);  EPI = 3: C = (alpha * acc + bias) * dact_in (act = 3: the forward
// stored the derivative).  With act = 3, EPI = 2 writes gelu'(pre-activation) into `preact`.  bf16 output only; tanh-GELU and
// fp32 outputs use the 128^2 kernel.
// (One epilogue flavour per kernel: with all of them behind run-time flags the tile boundary was ~120 KB of code, and the
// instruction-cache misses of hopping over the dead flavours cost more than the K loop of a 22-step tile.)
// GROUPED: up to 32 independent problems (e.g. the weight-gradient GEMMs of three transformer blocks: 3 x (102 + 36 + 144 + 144)
// = 1278 tiles = 4.99 rounds of 256 CUs) share one persistent launch instead of leaving 112-220 CUs idle in each of twelve.  The tile -> problem lookup and the per-problem descriptors / leading dimensions are re-read
// from the kernel arguments (scalar loads) whenever the issue stream or the epilogue moves to a tile.
// DBG (measurement aid, tools/bench_gemm_bound.py; results are garbage): 1 = no MFMAs, 2 = no LDS-DMA in the K loop, 3 = no fragment
// ds_reads, 4 = no B-fragment ds_reads (a third of them: the LDS traffic of 128 x 128 per-wave tiles; constant B operands), 5 = the same with the B operands copied from A fragments (random data, no LDS read) -- what the K loop's time is made of;
// 6 = every tile loads the same 4 A and 2 B panels (2.2 MB at K = 1408: L2 hits; same instruction stream, same C stores) -- what the operand traffic beyond the L2 costs.
// SCHED = 1: the "rolling" K loop (see the comment in front of `trip_roll`): no ping-pong, one barrier per phase, every fragment is
// read half a phase before the MFMAs that consume it.
// FP8 = true: the operands are e4m3 bytes, both K-contiguous.  A K step is still 128 bytes of every row -- now 128 values -- so the LDS
// image, the DMA requests, the ring and the phase structure are unchanged; a lane's MFMA fragment is 32 consecutive bytes of its row
// (two swizzled ds_read_b128), one v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales: twice the bf16 rate) replaces the two
// 16x16x32 bf16 MFMAs of a tile and K step, and the per-tensor scales multiply alpha in the epilogue.
// HALF = true: HALF-WIDTH TILES.  A 1408- or 4224-wide output is 5.5 / 16.5 column tiles; computed as a whole tile the last one
// multiplies 128 columns of zeros (8.3 % of the MFMAs of proj / fc2 / dgrad-qkv / dgrad-fc1), and a persistent launch whose last round
// holds a few leftover tiles is as long as if it were full.  A half-width tile keeps the 256 rows and the wave layout but owns 128 real
// columns: wave column wn takes columns n0 + 32 wn .. + 31 as its "Blo" half (the one B piece of a K step, W, is loaded with the identity
// row map), "Bhi" does not exist and the two phases that would multiply it are not executed: a K step is TWO phases
//     A(u): read Alo(u) + W(u), quadrant (0,0)            B(u): read Ahi(u) (W stays in registers), quadrant (1,0)
// over three 16 KiB pieces.  Half the MFMAs per byte staged means half the ring depth in TIME, and a half tile's A panel is its own (the
// whole tiles of a row block share theirs through the L2): the A pieces get the ring's depth -- six slots {0,3,4,7,2,6}, three K steps,
// requested two steps ahead (2-3 phases of flight) -- W, which every workgroup reads and the L2 holds, two slots {1,5}, one step ahead:
//     A(u) issues W(u+1), Alo(u+2) [in this order: the counted wait for W must not cover the younger A piece]   wait vmcnt(10) -> Ahi(u)
//     B(u) issues Ahi(u+2)                                                                                       wait vmcnt(4)  -> W(u+1), Alo(u+1)
// Same ping-pong and the same RAW / WAR rules as the four-phase loop (a piece is waited for one phase before it is read, by every wave,
// in front of a barrier; a slot is refilled at the earliest two phases after its last read).  Ring slots are run-time values here (no
// 6-step unrolling), so the DMA requests are inline asm (g2_dma16_asm) and every wait is explicit.
// Scheduling (host: g2_half_plan; ids: whole tiles, then half tiles).  Measured first (profiles/r4_gemm_half_width_v1_halves_last.jsonl): with
// ALL half tiles in a last round, every workgroup streams an A panel of its own at once -- 656 MB for K = 6144, no MFMA work to hide it
// behind: that round was as long as a round of whole tiles.  So a workgroup runs its half tile BETWEEN its whole tiles, at a position
// that depends on its block of 32 ids (an XCD's share): at any time about one workgroup in five is on a half tile, the others keep
// sharing panels in lock step.  Each switch drains and refills the pipeline (the whole-tile stream ends like a launch's last tile).
// DYN = true: DEVICE-SIDE ROW COUNTS.  DropPath (P:264,274) zeroes whole samples of a branch; the block stack then runs that branch on the
// kept samples only, compacted to the front of its buffers, and how many were kept is a draw that lives in device memory (the step is a
// replayed HIP graph: no launch argument may depend on it).  The launch is sized for the full row count; the kernel reads the real one --
// *m_dev for forward / dgrad launches (fewer row tiles: workgroups whose first tile does not exist leave at once, descriptors end at the
// last real row so that stale rows behind it are never read or written), *k_dev for weight gradients (shorter K loops; per problem in a
// grouped launch).  A count of 0 is legal (every sample of the branch dropped): no tile runs / the weight gradient is written as zeros.
template <bool A_KC, bool B_KC, int EPI, bool GROUPED = false, int DBG = 0, int SCHED = 0, bool FP8 = false, bool SPLIT = false, bool HALF = false, bool DYN = false>
__global__ __launch_bounds__(512) void gemm256_kernel(Gemm256Params p) {
  static_assert(!DYN || (!FP8 && !HALF && SCHED == 0 && DBG == 0), "device-side row counts: plain bf16 kernels (single problem, with or without the tail split, or grouped)");
  static_assert(!FP8 || (A_KC && B_KC && !GROUPED && SCHED == 0 && DBG == 0), "the e4m3 flavour is built for K-contiguous operands only");
  static_assert(!SPLIT || (!GROUPED && SCHED == 0 && DBG == 0), "the tail split is built for the plain single-problem kernels");
  static_assert(!HALF || (A_KC && !GROUPED && !SPLIT && !FP8 && SCHED == 0 && DBG == 0), "half-width tiles: bf16, K-contiguous A, plain single-problem kernels");
  constexpr int ES = FP8 ? 1 : 2;                    // bytes per operand element
  constexpr int BKE = FP8 ? 128 : 64;                // operand elements per K step
  __shared__ __attribute__((aligned(16))) char lds[8 * G2_PIECE + 8 * 4096];     // ring + 8 wave-private epilogue windows = 160 KiB
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int nprog = gridDim.x;
  int tiles_m = p.tiles_m, K = p.K, total = p.total_tiles, M_rt = p.M;
  const int tiles_n = p.tiles_n;
  int a_bytes = (int)p.a_bytes, b_bytes = (int)p.b_bytes, c_bytes = (int)p.c_bytes, p_bytes = (int)p.p_bytes, d_bytes = (int)p.d_bytes;
  if constexpr (DYN && !GROUPED) {
    if (p.m_dev) {                                     // fewer rows (A_KC launches: A, C, preact, dact_in are [rows][...])
      M_rt = max(0, min(__builtin_amdgcn_readfirstlane(*p.m_dev), p.M));
      tiles_m = (M_rt + G2_BM - 1) / G2_BM;
      total = tiles_m * tiles_n;
      const int last = M_rt - 1;
      a_bytes = M_rt > 0 ? (last * p.lda + K) * 2 : 0;
      c_bytes = M_rt > 0 ? (last * p.ldc + p.N) * 2 : 0;
      p_bytes = (M_rt > 0 && p.preact) ? (last * p.ldp + p.N) * 2 : 0;
      d_bytes = (M_rt > 0 && p.dact_in) ? (last * p.ldd + p.N) * 2 : 0;
    }
    if (p.k_dev) {                                     // shorter contraction (rows-contiguous operands: [k][rows])
      K = max(0, min(__builtin_amdgcn_readfirstlane(*p.k_dev), p.K));
      a_bytes = K > 0 ? ((K - 1) * p.lda + p.M) * 2 : 0;
      b_bytes = K > 0 ? ((K - 1) * p.ldb + p.N) * 2 : 0;
    }
  }
  const int nk = (K + BKE - 1) / BKE;
  // loop trips: 2 K steps each (a ghost step multiplies zeros).  DYN: at least two trips -- a device-side K of 0 ... 128 still runs the
  // first-trip / last-trip pair the pipeline is built around, on zeros
  const int nk2 = DYN ? max((nk + 1) >> 1, 2) : ((nk + 1) >> 1);
  // tail split (SPLIT kernels): the plan comes from the host (g2_split_plan) -- or, under a device-side row count, is made HERE from the
  // real tile count by the same arithmetic (every workgroup computes the same scalars): the last round's `rem` tiles are cut into sp_s K
  // slices of sp_nk2 loop trips, one slice per workgroup (rem * sp_s <= grid: the launch is sized for the full row count)
  int sp_main = SPLIT ? p.split_main : total, sp_s = SPLIT ? p.split_s : 0, sp_nk2 = SPLIT ? p.split_nk2 : 0;
  if constexpr (DYN && SPLIT) {
    sp_main = total; sp_s = 0; sp_nk2 = 0;
    const int R = total / nprog, rem = total - R * nprog;
    if (rem > 0 && p.split_ws) {
      int sl = min(nprog / rem, 4);
      if (sl >= 2) {
        const int per = max((nk2 + sl - 1) / sl, 2);
        sl = (nk2 + per - 1) / per;
        if (sl >= 2 && nk - 2 * per >= 40) { sp_main = R * nprog; sp_s = sl; sp_nk2 = per; total = sp_main + rem * sl; }
      }
    }
  }
  const int smain = SPLIT ? sp_main : total;           // linear ids >= smain are K slices of the tail tiles (SPLIT only)
  const int full_end = HALF ? p.half_begin : total;    // linear ids >= full_end are half-width tiles (HALF only)
  const int tn_grid = HALF ? p.tiles_nf : tiles_n;     // column tiles of the grid the whole-tile ids are decoded in
  int kiss = K;                                        // K extent of the tile / slice whose pieces are being issued
  unsigned a_kstep = A_KC ? 128u : (unsigned)(64 * p.lda * 2);         // bytes per K step
  unsigned b_kstep = B_KC ? 128u : (unsigned)(64 * p.ldb * 2);
  const float alpha = FP8 ? p.alpha * p.scale_a[0] * (p.scale_b_vec ? 1.0f : p.scale_b[0]) : p.alpha;
  auto find_prob = [&](int l) {                                          // grouped launch: which problem owns linear tile l
    int pi = 0;
#pragma unroll
    for (int q = 1; q < 32; ++q) pi += (q < p.nprob && l >= p.prob[q].tile_begin) ? 1 : 0;
    return pi;
  };
  int lin = xcd_remap(blockIdx.x, nprog);
  if constexpr (DYN && !GROUPED) {
    if (lin >= total) return;                          // no tile (or tail slice) for this workgroup: before any barrier, any DMA request
  }

  // ---- staging state of the tile whose pieces are being ISSUED (runs ahead of the tile being multiplied) -------------
  // One buffer descriptor per operand for the whole launch (batch entries are reached through the per-lane offset), so the
  // descriptors are provably wave-uniform SGPRs and the LDS-DMA requests need no waterfall loop.  K tails and ghost K steps
  // are masked per lane (k index >= K -> out-of-range offset -> zeros); rows past M / N read finite neighbours or zeros and
  // only ever feed C rows / columns that are not stored.
  // The per-lane offsets are tile independent; the tile (and batch entry) enters as one scalar byte offset per operand, so
  // moving the issue stream to the next tile costs two s_add.
  G2Stage sa, sb;
  sa.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, a_bytes, 0x00020000);
  sb.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, b_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, c_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_pre = __builtin_amdgcn_make_buffer_rsrc((void*)p.preact, 0, p_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dact = __builtin_amdgcn_make_buffer_rsrc((void*)p.dact_in, 0, d_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_bias = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, (int)p.bias_bytes, 0x00020000);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    sa.voff[j] = g2_piece_voff<A_KC, true, ES>(lane, wave, j, 0, p.lda, 0, sa.kchunk[j]);
    sb.voff[j] = g2_piece_voff<B_KC, false, ES>(lane, wave, j, 0, p.ldb, 0, sb.kchunk[j]);
  }
  sa.hi_off = A_KC ? (unsigned)(64 * p.lda * ES) : 128u;         // + 64 rows
  sb.hi_off = B_KC ? (unsigned)(32 * p.ldb * ES) : 64u;          // + 32 rows
  auto stage_setup = [&](int l) {
    if (l >= full_end) { sa.toff = G2_OOB; sb.toff = G2_OOB; return; }  // no next (whole) tile: ghost requests read zeros
    if constexpr (GROUPED) {
      const G2Prob& q = p.prob[find_prob(l)];
      int qa_bytes = (int)q.a_bytes, qb_bytes = (int)q.b_bytes;
      if constexpr (DYN) {                                               // this problem's own contraction length
        kiss = q.k_dev ? max(0, min(__builtin_amdgcn_readfirstlane(*q.k_dev), q.K)) : q.K;
        qa_bytes = kiss > 0 ? ((kiss - 1) * q.lda + q.M) * 2 : 0;
        qb_bytes = kiss > 0 ? ((kiss - 1) * q.ldb + q.N) * 2 : 0;
      }
      sa.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)q.A, 0, qa_bytes, 0x00020000);
      sb.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)q.B, 0, qb_bytes, 0x00020000);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        sa.voff[j] = g2_piece_voff<A_KC, true>(lane, wave, j, 0, q.lda, 0, sa.kchunk[j]);
        sb.voff[j] = g2_piece_voff<B_KC, false>(lane, wave, j, 0, q.ldb, 0, sb.kchunk[j]);
      }
      sa.hi_off = A_KC ? (unsigned)(64 * q.lda * 2) : 128u;
      sb.hi_off = B_KC ? (unsigned)(32 * q.ldb * 2) : 64u;
      a_kstep = A_KC ? 128u : (unsigned)(64 * q.lda * 2);
      b_kstep = B_KC ? 128u : (unsigned)(64 * q.ldb * 2);
      const G2Tile t = g2_decode(l - q.tile_begin, q.tiles_m, q.tiles_n);
      sa.toff = (unsigned)((A_KC ? (long)t.m0 * q.lda : (long)t.m0) * 2);
      sb.toff = (unsigned)((B_KC ? (long)t.n0 * q.ldb : (long)t.n0) * 2);
      return;
    }
    int tile_id = l;
    long k0 = 0;
    if constexpr (SPLIT) {
      if (l >= smain) {
        const int u = l - smain, tl = u / sp_s;
        tile_id = smain + tl;
        k0 = (long)(u - tl * sp_s) * sp_nk2 * 2 * BKE;
        kiss = min(K - (int)k0, sp_nk2 * 2 * BKE);
      } else {
        kiss = K;
      }
    }
    G2Tile t = g2_decode(tile_id, tiles_m, tn_grid);
    if constexpr (DBG == 6) { t.m0 = ((t.m0 / G2_BM) & 3) * G2_BM; t.n0 = ((t.n0 / G2_BN) & 1) * G2_BN; }   // every tile reads the same 4 + 2 panels
    sa.toff = (unsigned)((t.z * p.strideA + (A_KC ? (long)t.m0 * p.lda + k0 : k0 * p.lda + (long)t.m0)) * ES);
    sb.toff = (unsigned)((t.z * p.strideB + (B_KC ? (long)t.n0 * p.ldb + k0 : k0 * p.ldb + (long)t.n0)) * ES);
  };

  // ---- fragment read addresses: two opaque per-lane bases per operand (k half 0 / 1) ---------------------------------------
  const int i16 = lane & 15, g4 = lane >> 4;
  // KC operand: base[kk] (+ it * 2048 as an immediate).  Transposed operand: base[tile] (+ kk * 8192 as an immediate).
  unsigned la[4], lb[2];
  if constexpr (FP8) {                                  // 32 bytes of the lane's row: chunks 2 g4 and 2 g4 + 1 (swizzled: they differ in bit 4)
    const unsigned kc = (unsigned)(i16 * 128 + (((2 * g4) ^ ((i16 >> 1) & 7)) << 4));
    la[0] = kc + wm * 64 * 128; la[1] = (kc ^ 16u) + wm * 64 * 128;
    la[2] = la[3] = 0;
  } else if constexpr (A_KC) {
    const unsigned kc = (unsigned)(i16 * 128 + ((g4 ^ ((i16 >> 1) & 7)) << 4));
    la[0] = kc + wm * 64 * 128; la[1] = (kc ^ 64u) + wm * 64 * 128;             // piece rows wm*64 + it*16
    la[2] = la[3] = 0;
  } else {
    const int kr = 8 * g4 + (i16 >> 2);
    const unsigned t = (unsigned)(kr * 256 + (g2_swz(kr) << 4) + (((i16 & 3) >> 1) << 4) + (i16 & 1) * 8);
#pragma unroll
    for (int it = 0; it < 4; ++it) la[it] = t ^ ((unsigned)(wm * 8 + it * 2) << 4);   // row chunk wm*8 + it*2
  }
  if constexpr (FP8) {
    const unsigned kc = (unsigned)(i16 * 128 + (((2 * g4) ^ ((i16 >> 1) & 7)) << 4));
    lb[0] = kc + wn * 32 * 128; lb[1] = (kc ^ 16u) + wn * 32 * 128;
  } else if constexpr (B_KC) {
    const unsigned kc = (unsigned)(i16 * 128 + ((g4 ^ ((i16 >> 1) & 7)) << 4));
    lb[0] = kc + wn * 32 * 128; lb[1] = (kc ^ 64u) + wn * 32 * 128;             // piece rows wn*32 + jt*16
  } else {
    const int kr = 8 * g4 + (i16 >> 2);
    const unsigned t = (unsigned)(kr * 256 + (g2_swz(kr) << 4) + (((i16 & 3) >> 1) << 4) + (i16 & 1) * 8);
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) lb[jt] = t ^ ((unsigned)(wn * 4 + jt * 2) << 4);
  }
  // NOTE: these (and the DMA destinations) must stay transparent to the compiler: its waitcnt pass proves from their known bits
  // that a ds_read of one 16 KiB slot cannot alias the LDS-DMA writes in flight to the other slots; an opaque address makes it
  // insert s_waitcnt vmcnt(0) in front of every ds_read, which drains the DMA queue and serialises the pipeline.

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  using FragT = std::conditional_t<FP8, g2_i32x8, s16x8>;
  constexpr int KKN = FP8 ? 1 : 2;                       // MFMA k slices per K step
  FragT af[KKN][4], blo[KKN][2], bhi[KKN][2];
  if constexpr ((DBG == 3 || DBG == 4) && !FP8) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int it = 0; it < 4; ++it) { af[kk][it] = s16x8{1, 2, 3, 4, 5, 6, 7, 8}; asm volatile("" : "+v"(af[kk][it])); }
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) { blo[kk][jt] = s16x8{1, 2, 3, 4, 5, 6, 7, 8}; bhi[kk][jt] = blo[kk][jt]; asm volatile("" : "+v"(blo[kk][jt]), "+v"(bhi[kk][jt])); }
    }
  }
  float zero1 = 0.f;
  asm volatile("" : "+v"(zero1));                        // not a compile-time constant for the tile loop (see epilogue)

  const int nk_e = 2 * nk2;
  int nk2_cur = nk2, nk_e_cur = nk_e;                     // of the tile being multiplied (SPLIT: a K slice runs split_nk2 trips)

  // issue piece (type, K step u of the issue tile) into LDS slot `slot`
  unsigned wave_off = (unsigned)wave * 2048u;
  auto issue = [&](int type, int slot, int u) {
    if constexpr (DBG == 2) return;
    char* base = lds + wave_off;
    if (type == 0) g2_issue<A_KC>(sa, 0, (unsigned)u * a_kstep, kiss - u * BKE, base, slot);
    else if (type == 3) g2_issue<A_KC>(sa, 1, (unsigned)u * a_kstep, kiss - u * BKE, base, slot);
    else {
      if constexpr (A_KC == B_KC) { sb.kchunk[0] = sa.kchunk[0]; sb.kchunk[1] = sa.kchunk[1]; }   // same values: one register pair
      if (type == 1) g2_issue<B_KC>(sb, 0, (unsigned)u * b_kstep, kiss - u * BKE, base, slot);
      else g2_issue<B_KC>(sb, 1, (unsigned)u * b_kstep, kiss - u * BKE, base, slot);
    }
  };

  // ---- optional start-up skew (measurement aid, ivh_gemm256_debug): workgroup w sleeps (w % 16) * stagger * ~0.5 us.  It was
  // tried as a way to spread the C store bursts of the lock-stepped workgroups; measured, it only adds its own delay.
  if (p.stagger > 0) {
    const int n = (lin & 15) * p.stagger;
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(16);          // ~1024 cycles ~ 0.5 us
  }

  // ---- prologue of the first tile: pieces 0..5 (K step 0 complete, Alo/Blo of K step 1) --------------------------------
  if constexpr (SCHED == 1) {
    stage_setup(lin);
    // all eight pieces of K steps 0 and 1, in the order they are needed: Blo, Alo, Bhi, Ahi
    issue(1, 1, 0); issue(0, 0, 0); issue(2, 2, 0); issue(3, 3, 0); issue(1, 5, 1); issue(0, 4, 1); issue(2, 6, 1); issue(3, 7, 1);
    G2_WAIT_VM(6);                                       // pieces 0..4 landed (this wave's share); the tile prologue has the barrier
  } else if constexpr (!HALF) {                          // (HALF: every run of whole tiles has its prologue inside the tile loop)
    stage_setup(lin);
    issue(0, 0, 0); issue(1, 1, 0); issue(2, 2, 0); issue(3, 3, 0); issue(0, 4, 1); issue(1, 5, 1);
    G2_WAIT_VM(2);                                       // pieces 0..4 landed (this wave's share)
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();           // group 1 runs one barrier behind group 0
  }

  // FP8: the scaled-MFMA intrinsic is a pure IR value that the optimiser sinks out of its segment (all 64 of a loop trip ended up in one
  // clump, 180 VGPRs spilled); empty volatile asm "uses" of the operands after the segment's barrier and of the results before its end
  // pin the MFMAs between them without emitting an instruction.
#define G2_PIN_IN(BF)                                                                           \
  if constexpr (FP8) {                                                                          \
    _Pragma("unroll") for (int it = 0; it < 4; ++it) asm volatile("" : "+v"(af[0][it]));        \
    _Pragma("unroll") for (int jt = 0; jt < 2; ++jt) asm volatile("" : "+v"(BF[0][jt]));        \
  }
#define G2_PIN_OUT(MH, NH)                                                                      \
  if constexpr (FP8) {                                                                          \
    _Pragma("unroll") for (int it = 0; it < 4; ++it)                                            \
    _Pragma("unroll") for (int jt = 0; jt < 2; ++jt) asm volatile("" : "+v"(acc[(MH) * 4 + it][(NH) * 2 + jt])); \
  }
#define G2_MMA(MH, BF, NH)                                                                      \
  G2_PIN_IN(BF)                                                                                 \
  if constexpr (DBG != 1)                                                                       \
  _Pragma("unroll") for (int kk = 0; kk < KKN; ++kk)                                            \
  _Pragma("unroll") for (int it = 0; it < 4; ++it)                                              \
  _Pragma("unroll") for (int jt = 0; jt < 2; ++jt)                                              \
      acc[(MH) * 4 + it][(NH) * 2 + jt] = g2_mma(BF[kk][jt], af[kk][it], acc[(MH) * 4 + it][(NH) * 2 + jt]); \
  G2_PIN_OUT(MH, NH)

#define G2_SEG_BEGIN(NOWAIT)                                                             \
  __builtin_amdgcn_sched_barrier(0);                                                     \
  if (!(NOWAIT)) G2_WAIT_VM(8);                                                          \
  __builtin_amdgcn_s_barrier();                                                          \
  if constexpr (!A_KC || !B_KC) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
  __builtin_amdgcn_sched_barrier(0);                                                     \
  __builtin_amdgcn_s_setprio(1);
#define G2_SEG_BEGIN_N(N)                                                                \
  __builtin_amdgcn_sched_barrier(0);                                                     \
  G2_WAIT_VM(N);                                                                         \
  __builtin_amdgcn_s_barrier();                                                          \
  if constexpr (!A_KC || !B_KC) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
  __builtin_amdgcn_sched_barrier(0);                                                     \
  __builtin_amdgcn_s_setprio(1);
#define G2_SEG_END(SKIP)                                     \
  __builtin_amdgcn_s_setprio(0);                             \
  __builtin_amdgcn_sched_barrier(0);                         \
  if (!(SKIP)) __builtin_amdgcn_s_barrier();                 \
  __builtin_amdgcn_sched_barrier(0);

  // fragment reads of one piece (all k slices): the A rows / B columns of this wave's quadrant half
  auto rd_a = [&](int slot) {
    if constexpr (DBG == 3) return;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      if constexpr (FP8) af[0][it] = g2_frag8(lds, la[0], la[1], slot * G2_PIECE + it * 2048);
      else {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) af[kk][it] = g2_frag_t<A_KC, FragT>(lds, la[A_KC ? kk : it], slot * G2_PIECE + (A_KC ? it * 2048 : kk * 8192));
      }
    }
  };
  auto rd_b = [&](FragT (&dst)[KKN][2], int slot) {
    if constexpr (DBG == 3 || DBG == 4) return;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
      if constexpr (FP8) dst[0][jt] = g2_frag8(lds, lb[0], lb[1], slot * G2_PIECE + jt * 2048);
      else if constexpr (DBG == 5) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) { dst[kk][jt] = af[kk][jt + 1]; asm volatile("" : "+v"(dst[kk][jt])); }
      } else {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) dst[kk][jt] = g2_frag_t<B_KC, FragT>(lds, lb[B_KC ? kk : jt], slot * G2_PIECE + (B_KC ? jt * 2048 : kk * 8192));
      }
    }
  };

  // one loop trip = K steps u0 (slots 0..3) and u0 + 1 (slots 4..7) of the current tile.
  //   FIRST: first trip of a tile (phases 0..2 skip the vmcnt wait);  LAST: last trip (the issue stream moves to `lin_next`).
  auto trip = [&](bool FIRST, bool LAST, int u0, int lin_next, bool final_tile) {
    int kshift = 0;                                      // K steps to subtract once the issue stream is in the next tile
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int u = u0 + half;
      const bool nowait = FIRST && half == 0;
      // ---- phase 0: quadrant (0,0): read Alo + Blo; issue piece p+6 = Bhi(u+1) into slot ((half^1)*4 + 2)
      rd_b(blo, half * 4 + 1);
      rd_a(half * 4 + 0);
      issue(2, (half ^ 1) * 4 + 2, u + 1 - kshift);
      G2_SEG_BEGIN(nowait);
      G2_MMA(0, blo, 0);
      G2_SEG_END(false);
      // ---- phase 1: quadrant (0,1): read Bhi; issue Ahi(u+1)
      rd_b(bhi, half * 4 + 2);
      issue(3, (half ^ 1) * 4 + 3, u + 1 - kshift);
      G2_SEG_BEGIN(nowait);
      G2_MMA(0, bhi, 1);
      G2_SEG_END(false);
      // ---- phase 2: quadrant (1,1): read Ahi; issue Alo(u+2) (slot of Alo(u), dead since phase 0)
      rd_a(half * 4 + 3);
      if (LAST && half == 0) { stage_setup(lin_next); kshift = nk_e_cur; }   // K step u0 + 2 = nk_e is step 0 of the next tile
      issue(0, half * 4 + 0, u + 2 - kshift);
      G2_SEG_BEGIN(nowait);
      G2_MMA(1, bhi, 1);
      G2_SEG_END(false);
      // ---- phase 3: quadrant (1,0): nothing to read (Blo still in registers); issue Blo(u+2)
      issue(1, half * 4 + 1, u + 2 - kshift);
      G2_SEG_BEGIN(false);
      G2_MMA(1, blo, 0);
      G2_SEG_END(LAST && half == 1 && final_tile && wm == 1);
    }
  };
  // ---- SCHED = 1: rolling K loop ------------------------------------------------------------------------------------------
  // The ping-pong above serialises the two wave groups: a phase lasts (memory segment of one group) + (memory segment of the
  // other), each hidden behind the partner's 16 MFMAs only if it is shorter than them -- measured, the loop without any MFMA
  // still takes 70 % of its time (tools/bench_gemm_bound.py).  Here every wave runs the same straight-line schedule and the two
  // waves of a SIMD interleave freely: a phase is [s_waitcnt vmcnt(10)] [s_barrier] [LDS-DMA of the piece 8 phases ahead] and two
  // halves of [2-4 fragment ds_reads for the NEXT half] [8 MFMA].  Fragment registers are recycled half a phase after their last
  // use, so the 64 fragment VGPRs of the ping-pong version suffice:
  //     half-phase   MFMAs (quadrant, k half)      reads issued (consumed one half later)
  //     p0.h0        (0,0) kk0   Alo Blo           Alo[kk1]                    -> af[1]
  //     p0.h1        (0,0) kk1                     Bhi[kk0]                    -> BH[0]
  //     p1.h0        (0,1) kk0   Alo Bhi           Bhi[kk1]                    -> BH[1]
  //     p1.h1        (0,1) kk1                     Ahi[kk0]                    -> af[0]
  //     p2.h0        (1,1) kk0   Ahi Bhi           Ahi[kk1]                    -> af[1]
  //     p2.h1        (1,1) kk1                     Blo(u+1)[kk0]               -> BH[0]   (Bhi's registers: the two B buffers swap roles
  //     p3.h0        (1,0) kk0   Ahi Blo           Blo(u+1)[kk1]               -> BH[1]    every K step; the loop body is two K steps)
  //     p3.h1        (1,0) kk1                     Alo(u+1)[kk0]               -> af[0]
  // Pieces are numbered in the order they are needed, S = 4 * kstep + (Blo 0, Alo 1, Bhi 2, Ahi 3): piece S is read in the second
  // half of phase S-2 and the first half of phase S-1, so it must have landed at the barrier of phase S-2 (RAW) and its slot is free
  // from the barrier of phase S on (WAR); phase P issues piece P+8 into that slot: 6 phases of DMA lead, 5 pieces (10 requests per
  // wave) may be in flight across the wait.  The last K step of a tile does not prefetch into the next tile (the registers
  // would be live across the epilogue): the tile prologue reads Alo[kk0] and Blo after its barrier instead.
  s16x8 bx[2][2], by[2][2];
  auto trip_roll = [&](bool FIRST, bool LAST, int u0, int lin_next) {
    if constexpr (SCHED == 1) {
    int kshift = 0;
    if (LAST) { stage_setup(lin_next); kshift = nk_e; }  // everything issued in the last trip (K steps u0+2, u0+3) is the next tile's
#define G2R_BEGIN(NOWAIT)                                 \
  __builtin_amdgcn_sched_barrier(0);                      \
  if (!(NOWAIT)) G2_WAIT_VM(10);                          \
  __builtin_amdgcn_s_barrier();                           \
  __builtin_amdgcn_sched_barrier(0);
#define G2R_LGKM(NA, NB)                                                                                             \
  if constexpr (!A_KC || !B_KC) {                                                                                    \
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"((NA) * (A_KC ? 1 : 2) + (NB) * (B_KC ? 1 : 2)) : "memory");          \
  }                                                                                                                  \
  __builtin_amdgcn_sched_barrier(0);
#define G2R_MMA(MH, BF, NH, KK)                                                                  \
  if constexpr (DBG != 1)                                                                        \
  _Pragma("unroll") for (int it = 0; it < 4; ++it)                                               \
  _Pragma("unroll") for (int jt = 0; jt < 2; ++jt)                                               \
      acc[(MH) * 4 + it][(NH) * 2 + jt] = mfma16(BF[KK][jt], af[KK][it], acc[(MH) * 4 + it][(NH) * 2 + jt]); \
  __builtin_amdgcn_sched_barrier(0);
#define G2R_READ_A(SLOT, KK)                                                                     \
  if constexpr (DBG != 3)                                                                        \
  _Pragma("unroll") for (int it = 0; it < 4; ++it)                                               \
      af[KK][it] = g2_frag<A_KC>(lds, la[A_KC ? (KK) : it], (SLOT) * G2_PIECE + (A_KC ? it * 2048 : (KK) * 8192));
#define G2R_READ_B(DST, SLOT, KK)                                                                \
  if constexpr (DBG != 3)                                                                        \
  _Pragma("unroll") for (int jt = 0; jt < 2; ++jt)                                               \
      DST[KK][jt] = g2_frag<B_KC>(lds, lb[B_KC ? (KK) : jt], (SLOT) * G2_PIECE + (B_KC ? jt * 2048 : (KK) * 8192));
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int u = u0 + half;
      const bool nowait = FIRST && half == 0;
      const bool tail = LAST && half == 1;               // last K step of the tile: no prefetch into the next tile
      auto& BL = half ? by : bx;                         // Blo of this K step
      auto& BH = half ? bx : by;                         // Bhi of this K step, then Blo of the next one
      const int s0 = half * 4, n0 = (half ^ 1) * 4;
      // ---- phase 0: (0,0) = Alo x Blo; issue Blo(u+2)
      G2R_BEGIN(nowait);
      issue(1, s0 + 1, u + 2 - kshift);
      G2R_READ_A(s0 + 0, 1);
      G2R_LGKM(4, 0);
      G2R_MMA(0, BL, 0, 0);
      G2R_READ_B(BH, s0 + 2, 0);
      G2R_LGKM(0, 2);
      G2R_MMA(0, BL, 0, 1);
      // ---- phase 1: (0,1) = Alo x Bhi; issue Alo(u+2)
      G2R_BEGIN(nowait);
      issue(0, s0 + 0, u + 2 - kshift);
      G2R_READ_B(BH, s0 + 2, 1);
      G2R_LGKM(0, 2);
      G2R_MMA(0, BH, 1, 0);
      G2R_READ_A(s0 + 3, 0);
      G2R_LGKM(4, 0);
      G2R_MMA(0, BH, 1, 1);
      // ---- phase 2: (1,1) = Ahi x Bhi; issue Bhi(u+2)
      G2R_BEGIN(nowait);
      issue(2, s0 + 2, u + 2 - kshift);
      G2R_READ_A(s0 + 3, 1);
      G2R_LGKM(4, 0);
      G2R_MMA(1, BH, 1, 0);
      if (!tail) { G2R_READ_B(BH, n0 + 1, 0); }
      G2R_LGKM(0, 2);
      G2R_MMA(1, BH, 1, 1);
      // ---- phase 3: (1,0) = Ahi x Blo; issue Ahi(u+2)
      G2R_BEGIN(false);
      issue(3, s0 + 3, u + 2 - kshift);
      if (!tail) { G2R_READ_B(BH, n0 + 1, 1); }
      G2R_LGKM(0, 2);
      G2R_MMA(1, BL, 0, 0);
      if (!tail) { G2R_READ_A(n0 + 0, 0); }
      G2R_LGKM(4, 0);
      G2R_MMA(1, BL, 0, 1);
    }
    }
  };
  // ---- HALF: the K loop of a half-width tile (see the kernel comment) -------------------------------------------------------------------
  auto half_tile = [&](int m0, int n0) {
    if constexpr (HALF) {
      const unsigned lds_base = (unsigned)(unsigned long)(lds_void_t*)lds + wave_off;
      const u32x4 ra4 = g2_rsrc4(p.A, p.a_bytes), rb4 = g2_rsrc4(p.B, p.b_bytes);
      unsigned vb[2]; int kb[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) vb[j] = g2_piece_voff<B_KC, false, ES, true>(lane, wave, j, 0, p.ldb, 0, kb[j]);
      const unsigned a_t = (unsigned)((long)m0 * p.lda * ES), b_t = (unsigned)((B_KC ? (long)n0 * p.ldb : (long)n0) * ES);
      auto dma = [&](const u32x4& rs, const unsigned (&v)[2], const int (&kc)[2], unsigned add, int krem, unsigned slot) {
#pragma unroll
        for (int j = 0; j < 2; ++j) g2_dma16_asm(rs, lds_base + slot * (unsigned)G2_PIECE + (unsigned)(j * 1024), (kc[j] < krem) ? v[j] + add : G2_OOB);
      };
      auto a_slot = [](int idx) { return (unsigned)((0x627430u >> (4 * idx)) & 15u); };          // A ring position 0..5 -> slot {0,3,4,7,2,6}
      auto issue_a = [&](int u, int hi, int idx) { dma(ra4, sa.voff, sa.kchunk, a_t + (hi ? sa.hi_off : 0u) + (unsigned)u * a_kstep, K - u * BKE, a_slot(idx)); };
      auto issue_w = [&](int u) { dma(rb4, vb, kb, b_t + (unsigned)u * b_kstep, K - u * BKE, (u & 1) ? 5u : 1u); };
      auto rd_a_rt = [&](unsigned off) {
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) af[kk][it] = g2_frag_t<A_KC, FragT>(lds, la[A_KC ? kk : it] + off, A_KC ? it * 2048 : kk * 8192);
      };
      auto rd_w_rt = [&](unsigned off) {
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) blo[kk][jt] = g2_frag_t<B_KC, FragT>(lds, lb[B_KC ? kk : jt] + off, B_KC ? jt * 2048 : kk * 8192);
      };
      // prologue = the issue order of the steady state from two steps back: [Alo(0)] [Ahi(0)] [W(0), Alo(1)] [Ahi(1)]
      issue_a(0, 0, 0); issue_a(0, 1, 1); issue_w(0); issue_a(1, 0, 2); issue_a(1, 1, 3);
      G2_WAIT_VM(4);                                       // W(0), Alo(0), Ahi(0) landed (this wave's share)
      __builtin_amdgcn_s_barrier();
      if (wm == 1) __builtin_amdgcn_s_barrier();
      int ia = 0;                                          // A ring position of Alo(u)
      for (int u = 0; u < nk; ++u) {
        int uo = u;
        asm volatile("" : "+s"(uo));                       // opaque: one copy of the two-phase body
        const int ia1 = ia + 1, ia4 = ia + 4 >= 6 ? ia - 2 : ia + 4, ia5 = ia + 5 >= 6 ? ia - 1 : ia + 5;
        // ---- phase A: quadrant (0,0)
        rd_w_rt(((uo & 1) ? 5u : 1u) * (unsigned)G2_PIECE);
        rd_a_rt(a_slot(ia) * (unsigned)G2_PIECE);
        issue_w(uo + 1);
        issue_a(uo + 2, 0, ia4);
        G2_SEG_BEGIN_N(10);
        G2_MMA(0, blo, 0);
        G2_SEG_END(false);
        // ---- phase B: quadrant (1,0)
        rd_a_rt(a_slot(ia1) * (unsigned)G2_PIECE);
        issue_a(uo + 2, 1, ia5);
        G2_SEG_BEGIN_N(4);
        G2_MMA(1, blo, 0);
        G2_SEG_END(uo == nk - 1 && wm == 1);               // the tile's last barrier is group 1's to skip: both groups leave aligned
        ia = ia + 2 >= 6 ? ia - 4 : ia + 2;
      }
      G2_WAIT_VM(0);                                       // the ghost requests of steps >= nk must not outlive the tile
    }
  };
  // linear id -> (m0, n0) of a half-width tile
  auto half_decode = [&](int l) {
    const int j = l - full_end;
    if (j < p.half_split) {                              // column half (j & 1) of whole tile full_end + j / 2
      G2Tile t = g2_decode(full_end + (j >> 1), tiles_m, tn_grid);
      t.n0 += (j & 1) * 128;
      return t;
    }
    return G2Tile{0, (j - p.half_split) * G2_BM, tn_grid * G2_BN};   // N-edge tile of row tile j - half_split
  };
  int stamp_i = 0;
  auto stamp = [&]() {                                   // 4 stamps per tile: K loop start, K loop end, DMA wait done, epilogue end
    if (p.debug_stamps && (int)blockIdx.x == p.debug_stamp_wg && (wave & 3) == 0 && lane == 0 && stamp_i < 64)
      p.debug_stamps[(wave >> 2) * 64 + stamp_i] = __builtin_amdgcn_s_memtime();
    ++stamp_i;
  };
  // HALF: the workgroup's sequence = its whole tiles lin0, lin0 + nprog, ... < full_end and its half tiles (the following ids < total);
  // the first half tile runs before whole tile number `hpos`, further ones (rare plans) after the last whole tile.
  int lin_f = lin, lin_h = total, hpos = 0x7fffffff, fulls_done = 0;
  bool need_pro = true;
  if constexpr (HALF) {
    const int nf_w = lin < full_end ? (full_end - lin + nprog - 1) / nprog : 0;
    lin_h = lin + nf_w * nprog;
    hpos = nf_w;
    if (p.half_interleave && lin_h < total) {
      // an N-edge tile runs in the round in which its row block's whole tiles run (they share its A panel: same time window -> the
      // memory-side cache still holds it; alone at the end of the launch the panel comes from HBM again); cut halves run last
      const int j = lin_h - full_end;
      if (p.half_interleave == 2) hpos = (lin >> 5) % (nf_w + 1);
      else if (j >= p.half_split) {
        const int e = j - p.half_split;
        hpos = min(((e >> 3) * (8 * tn_grid) + (e & 7)) / nprog, nf_w);
      }
    }
  }
  while (true) {
    bool is_half = false;
    if constexpr (HALF) {
      is_half = lin_h < total && (fulls_done == hpos || lin_f >= full_end);
      if (is_half) { lin = lin_h; lin_h += nprog; hpos = 0x7fffffff; }
      else { lin = lin_f; lin_f += nprog; ++fulls_done; }
    }
    const int lin_next = HALF ? lin_f : lin + nprog;
    // the last tile of a run of whole tiles ends like a launch's last tile: nothing prefetched, queue drained, wave groups aligned
    const bool final_tile = HALF ? (lin_f >= full_end || (lin_h < total && fulls_done == hpos)) : (lin_next >= full_end);
    stamp();
    if (is_half) {
      if constexpr (HALF) {
        // every request of the previous tile has been waited for (a run's final wait is vmcnt(0)); its C stores may still fly and the
        // counted waits of the half loop see them as older entries -- they only ever make a wait longer.  Both groups arrive aligned.
        const G2Tile t = half_decode(lin);
        half_tile(t.m0, t.n0);
        need_pro = true;
        stamp();                                           // (four stamps per tile on either path)
      }
    } else {
    if constexpr (HALF) {
      if (need_pro) {                                    // first whole tile of a run: pieces 0..5 (K step 0 complete, Alo / Blo of K step 1)
        kiss = K;
        stage_setup(lin);
        issue(0, 0, 0); issue(1, 1, 0); issue(2, 2, 0); issue(3, 3, 0); issue(0, 4, 1); issue(1, 5, 1);
        G2_WAIT_VM(2);
        __builtin_amdgcn_s_barrier();
        if (wm == 1) __builtin_amdgcn_s_barrier();
      }
      need_pro = final_tile;
    }
    if constexpr (SCHED == 1) {
      // tile prologue: pieces 0..4 of this tile were waited for (prologue / end of the previous tile) by every wave before this barrier
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      G2R_READ_A(0, 0);
      G2R_READ_B(bx, 1, 0);
      G2R_READ_B(bx, 1, 1);
      if constexpr (!A_KC || !B_KC) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (SPLIT) { nk2_cur = (lin >= smain) ? sp_nk2 : nk2; nk_e_cur = 2 * nk2_cur; }
    if constexpr (GROUPED && DYN) {                      // the K loop of THIS tile's problem (the issue stream may already be in another one)
      const G2Prob& q = p.prob[find_prob(lin)];
      const int kq = q.k_dev ? max(0, min(__builtin_amdgcn_readfirstlane(*q.k_dev), q.K)) : q.K;
      nk2_cur = max((((kq + BKE - 1) / BKE) + 1) >> 1, 2);
      nk_e_cur = 2 * nk2_cur;
    }
    for (int t2 = 0; t2 < nk2_cur; ++t2) {
      int t2o = t2;
      asm volatile("" : "+s"(t2o));                      // opaque: no peeled first / last copies of the 8-phase body (they spill)
      if constexpr (SCHED == 1) trip_roll(t2o == 0, t2o == nk2 - 1, 2 * t2, lin_next);
      else trip(t2o == 0, t2o == nk2_cur - 1, 2 * t2, (HALF && final_tile) ? full_end : lin_next, final_tile);
    }
    // pieces 0..4 of the next tile must have landed before its first three phases (which do not wait); the ghost requests
    // of the final tile must not outlive the workgroup's LDS
    stamp();
    if (final_tile) { G2_WAIT_VM(0); } else if constexpr (SCHED == 1) { G2_WAIT_VM(6); } else { G2_WAIT_VM(2); }
    }
    stamp();

    // ---- SPLIT: a K slice of a tail tile.  Every slice publishes its accumulators in the workspace (fragment layout, write-through
    // sc1 stores: 32 coalesced 8 KiB stores per workgroup, no L2 write-back fence), drains them, and one lane takes a ticket from the
    // tile's counter; the slice that draws the last ticket reads the other slabs (16 x 16 bytes in flight per lane -- a dependent
    // cross-XCD read is microseconds, the rate is the number of loads in flight) and adds all of them IN SLICE ORDER, its own from
    // the registers: the result does not depend on which slice arrives last.  It then runs the ordinary epilogue; the others run it
    // with their stores masked.  A slice is always its workgroup's final tile (tail tiles x slices <= grid), where both wave groups
    // have passed the same number of barriers, so __syncthreads() is safe here.  No workgroup ever waits for another one.
    bool unit_live = true;
    if constexpr (SPLIT) {
      if (lin >= smain) {
        const int u = lin - smain, tl = u / sp_s, me = u - tl * sp_s;
        const __amdgpu_buffer_rsrc_t rs_ws = __builtin_amdgcn_make_buffer_rsrc((void*)p.split_ws, 0, (int)p.split_ws_bytes, 0x00020000);
        const unsigned my_off = ((unsigned)u * 16384u + threadIdx.x) * 16u;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rs_ws, my_off + (unsigned)((i * 4 + j) * 8192), 0, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                // every wave drains its own write-through stores
        __syncthreads();
        volatile unsigned* flag = reinterpret_cast<volatile unsigned*>(lds);            // ring slot 0: no DMA, no fragment read is left
        if (threadIdx.x == 0) {
          const unsigned ticket = __hip_atomic_fetch_add(p.split_cnt + tl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const bool last = ticket == (unsigned)(sp_s - 1);
          if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          *flag = last ? 1u : 0u;
        }
        __syncthreads();
        unit_live = *flag != 0u;
        if (unit_live) {
          unsigned qoff[4];                                                             // slab of slice q; own / absent slices read zeros
#pragma unroll
          for (int q = 0; q < 4; ++q)
            qoff[q] = (q < sp_s && q != me) ? ((unsigned)(tl * sp_s + q) * 16384u + threadIdx.x) * 16u : G2_OOB;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            f32x4 part[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int j = 0; j < 4; ++j)
                part[q][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_ws, qoff[q] == G2_OOB ? G2_OOB : qoff[q] + (unsigned)((i * 4 + j) * 8192), 0, 16));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              f32x4 sum = (me == 0) ? acc[i][j] : part[0][j];
#pragma unroll
              for (int q = 1; q < 4; ++q) sum += (me == q) ? acc[i][j] : part[q][j];
              acc[i][j] = sum;
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }

    // ---- epilogue: lane owns row m = .. + (lane & 15) and the 4 consecutive columns n = .. + 4 * (lane >> 4) + {0..3} -----
    // Straight-line code: every global access goes through a buffer descriptor and an element outside C gets an out-of-range
    // offset (loads return 0, stores are dropped), so there is no divergent branch.  All LOADS (bias, gelu' input) are issued
    // and consumed before the first store: when the tile loop comes round only stores are in flight, which the compiler lets
    // ride under the next K loop; a load that might still be pending on some path would make it drain everything there.
    {
      int eM = M_rt, eN = p.N, eldc = p.ldc;
      __amdgpu_buffer_rsrc_t rs_ct = rs_c;
      G2Tile t;
      if constexpr (GROUPED) {
        const G2Prob& q = p.prob[find_prob(lin)];
        eM = q.M; eN = q.N; eldc = q.ldc;
        rs_ct = __builtin_amdgcn_make_buffer_rsrc(q.C, 0, (int)q.c_bytes, 0x00020000);
        t = g2_decode(lin - q.tile_begin, q.tiles_m, q.tiles_n);
      } else {
        int tile_id = lin;
        if constexpr (SPLIT) { if (lin >= smain) tile_id = smain + (lin - smain) / sp_s; }
        if (is_half) t = half_decode(lin); else t = g2_decode(tile_id, tiles_m, tn_grid);
      }
      // half-width tile: wave column wn owns 32 columns (its nt = 0, 1 tiles); the nt = 2, 3 accumulators are zero and never stored
      const int wcols = is_half ? 32 : 64;
      int i16e = lane & 15, g4e = lane >> 4;
      asm volatile("" : "+v"(i16e), "+v"(g4e));          // opaque: nothing of the address math is hoisted across the K loop
      const int mrow = t.m0 + wm * 128 + i16e;           // + mt * 16
      const int ncol = t.n0 + wn * wcols + 4 * g4e;      // + nt * 16
      const bool has_bias = p.bias != nullptr, has_pre = (EPI == 2) && p.preact != nullptr, live = p.debug_skip_stores == 0 && unit_live;
      const unsigned b_lane = (unsigned)((t.z * p.stride_bias + ncol) * 4);
      if constexpr (FP8) {
        if (p.scale_b_vec) {                                // per-channel weight scales: the accumulators take their column's scale in place
          const __amdgpu_buffer_rsrc_t rs_sb = __builtin_amdgcn_make_buffer_rsrc((void*)p.scale_b, 0, eN * 4, 0x00020000);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            const f32x4 sv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_sb, (ncol + nt * 16 < eN) ? (unsigned)((ncol + nt * 16) * 4) : G2_OOB, 0, 0));
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) acc[mt][nt] *= sv;
          }
        }
      }
      f32x4 bv[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const bool n_ok = ncol + nt * 16 < eN && (!is_half || nt < 2);
        bv[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (has_bias) bv[nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_bias, n_ok ? b_lane + nt * 64 : G2_OOB, 0, 0));
      }
      // C leaves through a wave-private 4 KiB LDS window (the 32 KiB above the ring): the MFMA fragment layout gives a lane 4
      // columns of 16 different rows, and storing that directly (16 x 32-byte pieces per instruction) runs at ~10 B/clk per CU --
      // the store ISSUE, not HBM, was then a quarter of a 22-step tile.  Two 16-row tiles at a time are written as fragments
      // (ds_write_b64, 16-byte chunks XOR-swizzled by row) and read back row-major: one buffer_store_dwordx4 = 8 rows x 128 B.
      char* win = lds + 8 * G2_PIECE + wave * 4096;
      const int wrow = i16e, wcol = g4e;                                  // fragment coordinates of this lane
      const int rrow = lane >> 3, rchunk = lane & 7;                      // row-major coordinates: row (of 8), 16-byte chunk
      const unsigned w_off = (unsigned)(wrow * 128 + (wcol & 1) * 8);     // + mt2 * 2048, chunk (nt * 2 + (wcol >> 1)) ^ (row & 7)
      const unsigned r_off = (unsigned)(rrow * 128 + ((rchunk ^ (rrow & 7)) << 4));   // + j * 1024 (8 rows; (row & 7) unchanged)
      const int srow = t.m0 + wm * 128 + rrow;                            // + pr * 32 + j * 8
      const int scol = t.n0 + wn * wcols + rchunk * 8;
      const bool col_in = scol < eN && (!is_half || rchunk < 4);          // (half-width: chunks 4..7 of the window row hold the zero tiles)
      const bool scol_ok = live && col_in;
      const unsigned c_st = (unsigned)((t.z * p.strideC + (long)srow * eldc + scol) * 2);
      const unsigned p_st = (unsigned)((t.z * p.stride_preact + (long)srow * p.ldp + scol) * 2);
      auto flush = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned lane_off, int ld, int pr) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const u32x4 row = *reinterpret_cast<const u32x4*>(win + r_off + j * 1024);
          const bool ok = scol_ok && (srow + pr * 32 + j * 8 < eM);
          __builtin_amdgcn_raw_buffer_store_b128(row, rs, ok ? lane_off + (unsigned)((pr * 32 + j * 8) * ld) * 2u : G2_OOB, 0, 0);
        }
      };
      // EPI = 1 / 3: the row-major loads of the gelu' input ([8 rows x 128 B] each) run two passes ahead of their use in two register
      // sets (32 VGPRs; all 16 rows at once spill): passes 0 and 1 are requested up front, the set a pass has copied into the window is
      // refilled with the rows of pass + 2 at once -- those loads are OLDER than the pass's stores, so the wait in front of pass + 2
      // (vmcnt counts loads and stores in issue order) leaves the stores in flight and finds the rows already there
      // (fc2 dgrad at B = 128: 887-892 -> 875-876 us, profiles/r4_gemm_epi3_rolling_prefetch_ab_v1.jsonl).
      u32x4 urows[2][4];
      const unsigned d_st = (unsigned)((t.z * p.stride_dact + (long)srow * p.ldd + scol) * 2);
      auto load_u = [&](int set, int pr) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool ok = col_in && (srow + pr * 32 + j * 8 < eM);
          urows[set][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dact, ok ? d_st + (unsigned)((pr * 32 + j * 8) * p.ldd) * 2u : G2_OOB, 0, 0));
        }
      };
      float csum[4][4];                                                   // EPI 1 / 3: column sums of this lane's C values
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) csum[nt][r] = 0.f;
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) {                                    // pairs of 16-row tiles
        if constexpr (EPI == 1 || EPI == 3) { if (pr == 0) { load_u(0, 0); load_u(1, 1); } }
        u32x2 du[2][4];
        if constexpr (EPI == 1 || EPI == 3) {             // gelu' inputs: row-major rows (loaded above) -> window -> fragment layout
#pragma unroll
          for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4*>(win + r_off + j * 1024) = urows[pr & 1][j];
          if (pr + 2 < 4) load_u(pr & 1, pr + 2);
#pragma unroll
          for (int mt2 = 0; mt2 < 2; ++mt2)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
              const unsigned ch = (unsigned)(nt * 2 + (wcol >> 1)) ^ (unsigned)(wrow & 7);
              du[mt2][nt] = *reinterpret_cast<const u32x2*>(win + w_off + mt2 * 2048 + (ch << 4));
            }
        }
        // one pass = the 8 fragments of this tile pair -> LDS window -> 4 row-major stores.  MODE 0: alpha * acc + bias,
        // 1: (..) * gelu'(u), 2: gelu(..).  The affine part is recomputed per pass instead of holding 32 floats across passes.
        auto pass = [&](auto mode_c, bool last, const __amdgpu_buffer_rsrc_t& rs, unsigned lane_off, int ld) {
          constexpr int MODE = decltype(mode_c)::value;
#pragma unroll
          for (int mt2 = 0; mt2 < 2; ++mt2)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
              const int mt = pr * 2 + mt2;
              float v[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = acc[mt][nt][r] * alpha + bv[nt][r];
              if (last) acc[mt][nt] = f32x4{zero1, zero1, zero1, zero1};  // opaque zero: the accumulators stay in place across tiles
              if constexpr (MODE == 1 || MODE == 3) {
                const u32x2 uu = du[mt2][nt];
                const float u[4] = {__uint_as_float(uu[0] << 16), __uint_as_float(uu[0] & 0xffff0000u),
                                    __uint_as_float(uu[1] << 16), __uint_as_float(uu[1] & 0xffff0000u)};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  v[r] *= (MODE == 3) ? u[r] : g2_dgelu(u[r]);
                  if constexpr (MODE == 3) csum[nt][r] += v[r];
                }
              } else if constexpr (MODE == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = g2_gelu(v[r]);
              }
              const unsigned ch = (unsigned)(nt * 2 + (wcol >> 1)) ^ (unsigned)(wrow & 7);
              *reinterpret_cast<u32x2*>(win + w_off + mt2 * 2048 + (ch << 4)) = pack4(v[0], v[1], v[2], v[3]);
            }
          flush(rs, lane_off, ld, pr);
        };
        if constexpr (EPI == 2) {
          if (has_pre) {
            // fc1 of the training step (act = 3; a pre-activation copy with act = 1 runs on the 128^2 kernel): gelu and gelu' share
            // one erf / exp; the derivative goes through the window first while the packed activations wait in 16 VGPRs
            u32x2 gpk[2][4];
#pragma unroll
            for (int mt2 = 0; mt2 < 2; ++mt2)
#pragma unroll
              for (int nt = 0; nt < 4; ++nt) {
                const int mt = pr * 2 + mt2;
                float gv[4], dv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) g2_gelu_pair(acc[mt][nt][r] * alpha + bv[nt][r], gv[r], dv[r]);
                acc[mt][nt] = f32x4{zero1, zero1, zero1, zero1};
                const unsigned ch = (unsigned)(nt * 2 + (wcol >> 1)) ^ (unsigned)(wrow & 7);
                *reinterpret_cast<u32x2*>(win + w_off + mt2 * 2048 + (ch << 4)) = pack4(dv[0], dv[1], dv[2], dv[3]);
                gpk[mt2][nt] = pack4(gv[0], gv[1], gv[2], gv[3]);
              }
            flush(rs_pre, p_st, p.ldp, pr);
#pragma unroll
            for (int mt2 = 0; mt2 < 2; ++mt2)
#pragma unroll
              for (int nt = 0; nt < 4; ++nt) {
                const unsigned ch = (unsigned)(nt * 2 + (wcol >> 1)) ^ (unsigned)(wrow & 7);
                *reinterpret_cast<u32x2*>(win + w_off + mt2 * 2048 + (ch << 4)) = gpk[mt2][nt];
              }
            flush(rs_ct, c_st, eldc, pr);
          } else {
            pass(std::integral_constant<int, 2>{}, true, rs_ct, c_st, eldc);
          }
        } else {
          pass(std::integral_constant<int, EPI>{}, true, rs_ct, c_st, eldc);
        }
      }
      if constexpr (EPI == 3) {
        // bias gradient of the layer in front (fc1): column sums of this C tile's rows.  Rows past M hold exact zeros (their A rows
        // and gelu' inputs were read as zeros).  16 rows (lanes of one 16-lane group) meet by xor shuffles, lane 0 of each group
        // stores 16 floats; rows [2 * tile_m + wm] of colsum_part, reduced later by ivh_colsum_finish: deterministic.  (Round 5: the 64 xor
        // shuffles per lane and tile were ds_bpermute round trips through the LDS crossbar -- 36 of the 130 us this epilogue costs over a plain
        // dgrad launch, profiles/r5_gemm_epilogue_decomposition_v1.jsonl; the DPP row rotations ride on the adds.)
        if (p.colsum_part && unit_live) {
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              csum[nt][r] = g2_row16_sum(csum[nt][r]);
            }
          if ((lane & 15) == 0) {
            float* dst = p.colsum_part + (long)(2 * (t.m0 / G2_BM) + wm) * eN + t.n0 + wn * wcols + 4 * (lane >> 4);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
              if (t.n0 + wn * wcols + 4 * (lane >> 4) + nt * 16 < eN && (!is_half || nt < 2))
                *reinterpret_cast<f32x4*>(dst + nt * 16) = f32x4{csum[nt][0], csum[nt][1], csum[nt][2], csum[nt][3]};
          }
        }
      }
    }
    stamp();
    if constexpr (HALF) {
      if (lin_f >= full_end && lin_h >= total) break;
    } else {
      if (lin_next >= total) break;
      lin = lin_next;
    }
  }
}

}  // namespace ivh

// Launcher used by ivh_gemm_bf16 (gemm.hip) when the 256^2 kernel is selected.  Arguments were validated there.
static int g_g2_stagger = -1, g_g2_skip_stores = 0;      // -1 = choose per launch
static unsigned long long* g_g2_stamps = nullptr;
static int g_g2_max_wg = 0;                               // measurement aid: cap the number of workgroups (0 = #CUs)
static int g_g2_sched = 0;                                // 0 = ping-pong K loop, 1 = rolling K loop (ivh_gemm256_debug_sched; A/B)
extern "C" int ivh_gemm256_debug_sched(int sched) { g_g2_sched = sched == 1 ? 1 : 0; return 0; }
static int g_g2_dbg = 0;                                  // measurement aid: K-loop ablation of the plain NT kernel (see DBG above)
extern "C" int ivh_gemm256_debug_ablate(int mode) { g_g2_dbg = (mode >= 0 && mode <= 6) ? mode : 0; return 0; }
extern "C" int ivh_gemm256_debug_max_wg(int n) { g_g2_max_wg = n; return 0; }
extern "C" int ivh_gemm256_debug(int stagger, int skip_stores) {
  g_g2_stagger = stagger; g_g2_skip_stores = skip_stores;
  return 0;
}
extern "C" int ivh_gemm256_debug_stamps(void* buf_128_u64) {
  g_g2_stamps = (unsigned long long*)buf_128_u64;
  return 0;
}

extern "C" int ivh_gemm256_supported(const ivh_gemm_desc* d);
static int g2_n_cu() {
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) n_cu = 256;
    else n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return n_cu;
}

// Tail split along K.  A persistent launch runs ceil(tiles / grid) rounds and the last one is as long as the others however few
// tiles it holds (B = 32 rows of the 1B model, N = 1408: 318 tiles = one full round + 62 tiles on 256 CUs).  When the tail round is at
// most half full, its `rem` tiles are cut into S = min(grid / rem, 4) K slices of whole double K steps, one slice per workgroup
// (rem * S <= grid), which meet through a workspace (gemm256_kernel, SPLIT).  The tail then costs per / nk2 of a round plus the exchange.
// Returns 1 and the plan when that pays: long K only (fc2 forward, the dgrads of qkv / fc1 at 13k rows: -15 ... -23 %).
static int g2_split_plan(long total, long cap, int nk, int* main_tiles, int* S, int* nk2s, int* tail_tiles) {
  if (cap <= 0 || total <= 0) return 0;
  const long R = total / cap, rem = total - R * cap;
  if (rem == 0) return 0;
  long s = cap / rem;
  if (s > 4) s = 4;                                      // the reducer holds one 16-byte load per slice and accumulator column in flight
  const int nk2 = (nk + 1) / 2;
  if (s < 2) return 0;
  int per = (int)((nk2 + s - 1) / s);
  if (per < 2) per = 2;                                  // a slice runs at least two loop trips (first / last trip are distinct code paths)
  s = (nk2 + per - 1) / per;
  // measured (profiles/r2_gemm_tail_split_v1.jsonl): the exchange -- rem * S slabs of 256 KiB written through to memory, rem workgroups
  // reading S - 1 of them at the cross-XCD rate, the extra pipeline fill -- costs 25-45 us; 40 K steps are ~58 us of a round
  if (s < 2 || nk - 2 * per < 40) return 0;
  *main_tiles = (int)(R * cap); *S = (int)s; *nk2s = per; *tail_tiles = (int)rem;
  return 1;
}

static int g_g2_split = [] { const char* e = getenv("IVH_NO_SPLIT"); return (e && e[0] == '1') ? 0 : 1; }();   // 0 = never split (A/B, tests; env IVH_NO_SPLIT=1)
extern "C" int ivh_gemm256_debug_split(int on) { g_g2_split = on ? 1 : 0; return 0; }

// bytes of workspace (d->split_ws) with which ivh_gemm256_launch / ivh_gemm256_fp8_launch would split the tail of this problem along K;
// 0 = no split (not built for this flavour, nothing to gain, or switched off).  `fp8`: e4m3 operands (a K step is 128 values).
extern "C" int64_t ivh_gemm256_split_ws_bytes(const ivh_gemm_desc* d, int fp8) {
  if (!g_g2_split || !d || !d->a_kc || d->batch > 1 || d->c_fp32 || g_g2_dbg || g_g2_sched || g_g2_stamps || g_g2_stagger > 0) return 0;
  if (d->k_dev) return 0;
  if (d->m_dev) {
    // device-side row count: the plan is made inside the kernel from the real tile count (gemm256_kernel, DYN && SPLIT); the workspace is
    // sized for the largest plan a launch on `cap` workgroups can make (one 256 KiB slab per workgroup).  Long K only: a slice needs
    // nk - 2 * per >= 40 with per >= 2 loop trips.
    if (fp8) return 0;
    const int epi_d = d->dact_in ? (d->act == 3 ? 3 : 1) : (d->act ? 2 : 0);
    if (epi_d == 1 || (epi_d == 2 && !d->b_kc) || (epi_d == 3 && d->b_kc)) return 0;
    if ((d->K + 63) / 64 < 44) return 0;
    const long capd = g_g2_max_wg > 0 ? g_g2_max_wg : g2_n_cu();
    return 4096 + (int64_t)capd * 262144;
  }
  if (fp8 && !d->b_kc) return 0;
  const int epi = d->dact_in ? (d->act == 3 ? 3 : 1) : (d->act ? 2 : 0);
  if (epi == 1 || (epi == 2 && !d->b_kc) || (epi == 3 && (fp8 ? false : d->b_kc))) return 0;
  const long total = (long)((d->M + ivh::G2_BM - 1) / ivh::G2_BM) * ((d->N + ivh::G2_BN - 1) / ivh::G2_BN);
  const long cap = g_g2_max_wg > 0 ? g_g2_max_wg : g2_n_cu();
  const int bke = fp8 ? 128 : 64;
  int mt, S, per, tail;
  if (!g2_split_plan(total, cap, (d->K + bke - 1) / bke, &mt, &S, &per, &tail)) return 0;
  return 4096 + (int64_t)tail * S * 262144;
}

// K steps of the split tail round relative to a whole round (1.0 = no split): the launch-time model of gemm.hip
extern "C" double ivh_gemm256_split_tail_frac(const ivh_gemm_desc* d, int fp8) {
  if (ivh_gemm256_split_ws_bytes(d, fp8) <= 0) return 1.0;
  const long total = (long)((d->M + ivh::G2_BM - 1) / ivh::G2_BM) * ((d->N + ivh::G2_BN - 1) / ivh::G2_BN);
  const long cap = g_g2_max_wg > 0 ? g_g2_max_wg : g2_n_cu();
  const int bke = fp8 ? 128 : 64;
  const int nk = (d->K + bke - 1) / bke;
  int mt, S, per, tail;
  if (!g2_split_plan(total, cap, nk, &mt, &S, &per, &tail)) return 1.0;
  return (double)(2 * per) / (double)nk;
}

// fills the split fields of p (total_tiles becomes the number of linear ids) and clears the counters; returns 1 if the launch is a SPLIT one
static int g2_apply_split(const ivh_gemm_desc* d, int fp8, ivh::Gemm256Params& p, long cap, hipStream_t s) {
  p.split_main = p.total_tiles; p.split_s = 0; p.split_nk2 = 0; p.split_ws = nullptr; p.split_cnt = nullptr; p.split_ws_bytes = 0;
  if (!d->split_ws) return 0;
  const int64_t need = ivh_gemm256_split_ws_bytes(d, fp8);
  if (need <= 0 || d->split_ws_bytes < need || ((uintptr_t)d->split_ws % 16) != 0) return 0;
  const int bke = fp8 ? 128 : 64;
  int mt, S, per, tail;
  if (!g2_split_plan(p.total_tiles, cap, (d->K + bke - 1) / bke, &mt, &S, &per, &tail)) return 0;
  if (hipMemsetAsync(d->split_ws, 0, 4096, s) != hipSuccess) return 0;
  p.split_main = mt; p.split_s = S; p.split_nk2 = per;
  p.split_cnt = reinterpret_cast<unsigned*>(d->split_ws);
  p.split_ws = reinterpret_cast<float*>(reinterpret_cast<char*>(d->split_ws) + 4096);
  p.split_ws_bytes = (long)tail * S * 262144;
  p.total_tiles = mt + tail * S;
  return 1;
}

// Half-width tiles (gemm256_kernel, HALF).  Static round-robin cost of a launch in whole-tile units: `nfull` ids of cost 1 followed by
// `nhalf` ids of cost G2_HALF_COST on `cap` workgroups (workgroup w runs ids w, w + cap, ...; the XCD remap is a bijection).
static const double G2_HALF_COST = 0.6;                  // a half-width tile: half the MFMAs, 3/4 of the operand traffic, its own prologue
static double g2_rr_cost(long nfull, long nhalf, long cap) {
  const long R = nfull / cap, r = nfull % cap, q = nhalf / cap, rh = nhalf % cap;
  double best = (double)R + G2_HALF_COST * (double)(q + (rh > 0 ? 1 : 0));      // a workgroup with R whole tiles, first in line for a half
  if (r > 0) {
    const double c = (double)(R + 1) + G2_HALF_COST * (double)(q + ((r + rh > cap) ? 1 : 0));
    if (c > best) best = c;
  }
  return best;
}
// The plan: tiles_nf = column tiles that are computed whole; ids [0, half_begin) whole tiles, then half_split ids = column halves of the
// leftover whole tiles of the last round, then one id per row tile for an N edge of at most 128 columns.  Returns 1 when the launch gets
// shorter by more than 2 % (model), 0 when the plain launch is as good.
static int g2_half_plan(int M, int N, long cap, int* tiles_nf, int* half_begin, int* half_split, int* total_ids) {
  if (cap <= 0 || M <= 0 || N <= 0) return 0;
  const long tm = (M + ivh::G2_BM - 1) / ivh::G2_BM, tn = (N + ivh::G2_BN - 1) / ivh::G2_BN;
  const int n_rem = N % ivh::G2_BN;
  const bool edge = n_rem > 0 && n_rem <= 128;
  const long tnf = edge ? tn - 1 : tn, F = tm * tnf, H = edge ? tm : 0;
  const double old_cost = (double)((tm * tn + cap - 1) / cap);
  const long rem = F % cap;
  // at most ONE half tile per workgroup: only the first one runs in the round of its row block (its A panel still cached); further ones
  // would run after the last whole tile, each streaming a panel of its own from HBM (measured on the frozen teachers' 263168-row GEMMs:
  // four half tiles per workgroup, the recipe step 3.5 % slower).  Launches that large gain <= 1 % from half tiles anyway.
  const double a = (edge && H <= cap) ? g2_rr_cost(F, H, cap) : 1e30;                   // edge tiles as half tiles, leftovers whole
  const double b = (rem > 0 && 2 * rem + H <= cap) ? g2_rr_cost(F - rem, 2 * rem + H, cap) : 1e30;   // ... and the leftovers cut in two
  const bool cut = b < a;
  const double best = cut ? b : a;
  if (!(best < 0.98 * old_cost)) return 0;
  if (F + 2 * rem + H >= (1L << 30)) return 0;
  *tiles_nf = (int)tnf;
  *half_begin = (int)(cut ? F - rem : F);
  *half_split = (int)(cut ? 2 * rem : 0);
  *total_ids = *half_begin + *half_split + (int)H;
  return 1;
}
// 0 = never (A/B, tests; env IVH_NO_HALF=1), 1 = half tiles interleaved with a workgroup's whole tiles (default), 2 = half tiles last
static int g_g2_half = [] { const char* e = getenv("IVH_NO_HALF"); return (e && e[0] == '1') ? 0 : 1; }();
extern "C" int ivh_gemm256_debug_half(int mode) { g_g2_half = (mode >= 0 && mode <= 3) ? mode : 1; return 0; }   // 3 = interleaved by XCD block (A/B)
static int g2_half_flavour(const ivh_gemm_desc* d) {     // the epilogue / layout combinations the HALF kernels are instantiated for
  if (!g_g2_half || !d->a_kc || d->c_fp32 || d->batch > 1 || d->m_dev || d->k_dev) return 0;
  const int epi = d->dact_in ? (d->act == 3 ? 3 : 1) : (d->act ? 2 : 0);
  return epi == 0 || (epi == 2 && d->b_kc) || (epi == 3 && !d->b_kc);
}
// host-side view of the plan ivh_gemm256_launch would use for `d` on `cap` workgroups (0 = the device's CUs):
// out = {tiles_nf, half_begin, half_split, total_ids}; returns 1 when half-width tiles are used, else 0.
extern "C" int ivh_gemm256_half_plan(const ivh_gemm_desc* d, int cap, int* out4) {
  if (!d || !out4 || !g2_half_flavour(d) || !ivh_gemm256_supported(d)) return 0;
  const long c = cap > 0 ? cap : (g_g2_max_wg > 0 ? g_g2_max_wg : g2_n_cu());
  return g2_half_plan(d->M, d->N, c, out4, out4 + 1, out4 + 2, out4 + 3);
}

// modelled length of the launch in whole-tile rounds when half-width tiles apply, else -1 (the launch-time model of gemm.hip)
extern "C" double ivh_gemm256_half_rounds(const ivh_gemm_desc* d) {
  int o[4];
  if (!ivh_gemm256_half_plan(d, 0, o)) return -1.0;
  return g2_rr_cost(o[1], (long)o[3] - o[1], g_g2_max_wg > 0 ? g_g2_max_wg : g2_n_cu());
}

// The combinations the 256x256 kernel is built for (everything else runs on the 128x128 kernel of gemm.hip).
extern "C" int ivh_gemm256_supported(const ivh_gemm_desc* d) {
  if (d->c_fp32 || d->act == 2) return 0;
  if (d->dact_in) return (d->act == 1 || d->act == 3) && d->a_kc && !d->b_kc && !d->preact;   // fc2 dgrad: dy W2 * gelu'(u)
  if (d->act == 1 && d->preact) return 0;                                         // u copy + erf GELU: 128^2 kernel
  if (d->act == 1 || d->act == 3) return d->a_kc && d->b_kc;                     // fc1 forward: gelu(x W1^T + b) (+ gelu' copy)
  return d->preact == nullptr;
}

// The 256^2 kernel addresses every operand through one buffer descriptor with 32-bit byte offsets: each (batched) operand / output
// extent must stay below 2 GiB - 16 MiB.  ivh_gemm_bf16 splits larger problems along M (K-contiguous A) or falls back to the 128^2
// kernel, which uses 64-bit pointers.
extern "C" int ivh_gemm256_fits(const ivh_gemm_desc* d) {
  const long lim = (1L << 31) - (1L << 24);
  const int nb = d->batch > 0 ? d->batch : 1;
  const long a_elems = d->a_kc ? ((long)d->M - 1) * d->lda + d->K : ((long)d->K - 1) * d->lda + d->M;
  const long b_elems = d->b_kc ? ((long)d->N - 1) * d->ldb + d->K : ((long)d->K - 1) * d->ldb + d->N;
  if (((long)(nb - 1) * d->strideA + a_elems) * 2 >= lim || ((long)(nb - 1) * d->strideB + b_elems) * 2 >= lim) return 0;
  if (((long)(nb - 1) * d->strideC + ((long)d->M - 1) * d->ldc + d->N) * (d->c_fp32 ? 4 : 2) >= lim) return 0;
  if (d->preact && ((long)(nb - 1) * d->stride_preact + ((long)d->M - 1) * d->ldp + d->N) * 2 >= lim) return 0;
  if (d->dact_in && ((long)(nb - 1) * d->stride_dact + ((long)d->M - 1) * d->ldd + d->N) * 2 >= lim) return 0;
  return 1;
}

extern "C" int ivh_gemm256_launch(const ivh_gemm_desc* d, void* stream) {
  using namespace ivh;
  IVH_REQUIRE(ivh_gemm256_supported(d), "gemm256: unsupported epilogue / layout combination");
  const long a_elems = d->a_kc ? ((long)d->M - 1) * d->lda + d->K : ((long)d->K - 1) * d->lda + d->M;
  const long b_elems = d->b_kc ? ((long)d->N - 1) * d->ldb + d->K : ((long)d->K - 1) * d->ldb + d->N;
  const int nb = d->batch > 0 ? d->batch : 1;
  const long a_bytes = ((long)(nb - 1) * d->strideA + a_elems) * 2, b_bytes = ((long)(nb - 1) * d->strideB + b_elems) * 2;
  IVH_REQUIRE(a_bytes < (1L << 31) - (1L << 24) && b_bytes < (1L << 31) - (1L << 24), "gemm256: (batched) operand larger than 2 GiB");
  IVH_REQUIRE(d->strideA >= 0 && d->strideB >= 0, "gemm256: negative batch stride");
  Gemm256Params p;
  IVH_REQUIRE(d->lda < (1L << 31) && d->ldb < (1L << 31) && d->ldc < (1L << 31) && d->ldp < (1L << 31) && d->ldd < (1L << 31),
              "gemm256: leading dimension does not fit 31 bits");
  p.A = d->A; p.B = d->B; p.lda = (int)d->lda; p.ldb = (int)d->ldb; p.M = d->M; p.N = d->N; p.K = d->K;
  p.C = d->C; p.ldc = (int)d->ldc; p.c_fp32 = d->c_fp32; p.bias = d->bias; p.act = d->act;
  p.preact = d->preact; p.ldp = (int)d->ldp; p.dact_in = d->dact_in; p.ldd = (int)d->ldd;
  p.alpha = d->alpha; p.tiles_m = (d->M + G2_BM - 1) / G2_BM; p.tiles_n = (d->N + G2_BN - 1) / G2_BN;
  p.strideA = d->strideA; p.strideB = d->strideB; p.strideC = d->strideC; p.stride_bias = d->stride_bias;
  p.stride_preact = d->stride_preact; p.stride_dact = d->stride_dact;
  p.batch = nb; p.a_bytes = a_bytes; p.b_bytes = b_bytes;
  p.colsum_part = (d->dact_in && d->act == 3 && nb == 1) ? d->colsum_part : nullptr;
  const long lim = (1L << 31) - (1L << 24);
  p.c_bytes = ((long)(nb - 1) * d->strideC + ((long)d->M - 1) * d->ldc + d->N) * (d->c_fp32 ? 4 : 2);
  p.p_bytes = d->preact ? ((long)(nb - 1) * d->stride_preact + ((long)d->M - 1) * d->ldp + d->N) * 2 : 0;
  p.d_bytes = d->dact_in ? ((long)(nb - 1) * d->stride_dact + ((long)d->M - 1) * d->ldd + d->N) * 2 : 0;
  p.bias_bytes = d->bias ? ((long)(nb - 1) * d->stride_bias + d->N) * 4 : 0;
  IVH_REQUIRE(p.c_bytes < lim && p.p_bytes < lim && p.d_bytes < lim, "gemm256: (batched) output larger than 2 GiB");
  IVH_REQUIRE(d->strideC >= 0 && d->stride_preact >= 0 && d->stride_dact >= 0 && d->stride_bias >= 0, "gemm256: negative batch stride");
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) n_cu = 256;
    else n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const long total = (long)p.tiles_m * p.tiles_n * p.batch;
  p.total_tiles = (int)total; p.nprob = 0;
  p.stagger = g_g2_stagger > 0 ? g_g2_stagger : 0;       // measured: the skew never pays once the epilogue stores are row-major
  p.debug_skip_stores = g_g2_skip_stores;
  p.debug_stamps = g_g2_stamps;
  { const char* e = g_g2_stamps ? getenv("IVH_G2_STAMP_WG") : nullptr; p.debug_stamp_wg = e ? atoi(e) : 0; }
  const long cap = g_g2_max_wg > 0 ? g_g2_max_wg : n_cu;
  hipStream_t s = (hipStream_t)stream;
  const int epi = d->dact_in ? (d->act == 3 ? 3 : 1) : (d->act ? 2 : 0);
  p.m_dev = d->m_dev; p.k_dev = d->k_dev;
  if (d->m_dev || d->k_dev) {                             // device-side row counts (DYN kernels): no tail split, no half-width tiles
    IVH_REQUIRE(nb == 1 && !g_g2_dbg && !g_g2_sched, "gemm256: device-side row counts (m_dev / k_dev) are built for single, unbatched problems");
    IVH_REQUIRE(!(d->m_dev && d->k_dev), "gemm256: m_dev and k_dev are mutually exclusive");
    IVH_REQUIRE(!d->m_dev || d->a_kc, "gemm256: m_dev counts the rows of a K-contiguous A (forward / dgrad launches)");
    IVH_REQUIRE(!d->k_dev || (!d->a_kc && !d->b_kc && epi == 0), "gemm256: k_dev counts the rows of rows-contiguous operands (weight gradients, plain epilogue)");
    IVH_REQUIRE(epi != 1, "gemm256: m_dev with the erf-recomputing dgrad epilogue (act = 1) is not built; use act = 3");
    p.half_begin = p.total_tiles; p.half_split = 0; p.tiles_nf = p.tiles_n; p.half_interleave = 0;
    p.split_main = p.total_tiles; p.split_s = 0; p.split_nk2 = 0; p.split_ws = nullptr; p.split_cnt = nullptr; p.split_ws_bytes = 0;
    dim3 grid((unsigned)(total < cap ? total : cap), 1, 1), block(512);
    if (d->m_dev && d->split_ws && total >= cap) {        // tail split planned on the device (long K, workspace for the largest plan supplied)
      const int64_t need = ivh_gemm256_split_ws_bytes(d, 0);
      if (need > 0 && d->split_ws_bytes >= need && ((uintptr_t)d->split_ws % 16) == 0 && hipMemsetAsync(d->split_ws, 0, 4096, s) == hipSuccess) {
        p.split_cnt = reinterpret_cast<unsigned*>(d->split_ws);
        p.split_ws = reinterpret_cast<float*>(reinterpret_cast<char*>(d->split_ws) + 4096);
        p.split_ws_bytes = (long)cap * 262144;
        if (epi == 0 && d->b_kc) hipLaunchKernelGGL((gemm256_kernel<true, true, 0, false, 0, 0, false, true, false, true>), grid, block, 0, s, p);
        else if (epi == 0) hipLaunchKernelGGL((gemm256_kernel<true, false, 0, false, 0, 0, false, true, false, true>), grid, block, 0, s, p);
        else if (epi == 2) hipLaunchKernelGGL((gemm256_kernel<true, true, 2, false, 0, 0, false, true, false, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((gemm256_kernel<true, false, 3, false, 0, 0, false, true, false, true>), grid, block, 0, s, p);
        return ivh_host::check_launch("gemm256_bf16 (device-side row count, tail split)");
      }
    }
    if (d->k_dev) hipLaunchKernelGGL((gemm256_kernel<false, false, 0, false, 0, 0, false, false, false, true>), grid, block, 0, s, p);
    else if (epi == 0 && d->b_kc) hipLaunchKernelGGL((gemm256_kernel<true, true, 0, false, 0, 0, false, false, false, true>), grid, block, 0, s, p);
    else if (epi == 0) hipLaunchKernelGGL((gemm256_kernel<true, false, 0, false, 0, 0, false, false, false, true>), grid, block, 0, s, p);
    else if (epi == 2) hipLaunchKernelGGL((gemm256_kernel<true, true, 2, false, 0, 0, false, false, false, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((gemm256_kernel<true, false, 3, false, 0, 0, false, false, false, true>), grid, block, 0, s, p);
    return ivh_host::check_launch("gemm256_bf16 (device-side row count)");
  }
  if (g2_apply_split(d, 0, p, cap, s)) {                  // tail tiles cut into K slices (SPLIT kernels)
    dim3 grid((unsigned)(p.total_tiles < cap ? p.total_tiles : cap), 1, 1), block(512);
    if (epi == 0 && d->b_kc) hipLaunchKernelGGL((gemm256_kernel<true, true, 0, false, 0, 0, false, true>), grid, block, 0, s, p);
    else if (epi == 0) hipLaunchKernelGGL((gemm256_kernel<true, false, 0, false, 0, 0, false, true>), grid, block, 0, s, p);
    else if (epi == 2) hipLaunchKernelGGL((gemm256_kernel<true, true, 2, false, 0, 0, false, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((gemm256_kernel<true, false, 3, false, 0, 0, false, true>), grid, block, 0, s, p);
    return ivh_host::check_launch("gemm256_bf16 (tail split)");
  }
  p.half_begin = p.total_tiles; p.half_split = 0; p.tiles_nf = p.tiles_n; p.half_interleave = 0;
  if (g2_half_flavour(d) && !g_g2_dbg && !g_g2_sched && g_g2_stagger <= 0) {
    int tnf, hb, hs, ids;
    if (g2_half_plan(d->M, d->N, cap, &tnf, &hb, &hs, &ids)) {                      // half-width tiles (HALF kernels)
      p.tiles_nf = tnf; p.half_begin = hb; p.half_split = hs; p.total_tiles = ids; p.half_interleave = g_g2_half == 1 ? 1 : (g_g2_half == 3 ? 2 : 0);
      dim3 grid((unsigned)(ids < cap ? ids : cap), 1, 1), block(512);
      if (epi == 0 && d->b_kc) hipLaunchKernelGGL((gemm256_kernel<true, true, 0, false, 0, 0, false, false, true>), grid, block, 0, s, p);
      else if (epi == 0) hipLaunchKernelGGL((gemm256_kernel<true, false, 0, false, 0, 0, false, false, true>), grid, block, 0, s, p);
      else if (epi == 2) hipLaunchKernelGGL((gemm256_kernel<true, true, 2, false, 0, 0, false, false, true>), grid, block, 0, s, p);
      else hipLaunchKernelGGL((gemm256_kernel<true, false, 3, false, 0, 0, false, false, true>), grid, block, 0, s, p);
      return ivh_host::check_launch("gemm256_bf16 (half-width tiles)");
    }
  }
  dim3 grid((unsigned)(total < cap ? total : cap), 1, 1), block(512);
  if (epi == 0) {
    if (d->a_kc && d->b_kc && g_g2_dbg == 1) hipLaunchKernelGGL((gemm256_kernel<true, true, 0, false, 1>), grid, block, 0, s, p);
    else if (d->a_kc && d->b_kc && g_g2_dbg == 2) hipLaunchKernelGGL((gemm256_kernel<true, true, 0, false, 2>), grid, block, 0, s, p);
    else if (d->a_kc && d->b_kc && g_g2_dbg == 3) hipLaunchKernelGGL((gemm256_kernel<true, true, 0, false, 3>), grid, block, 0, s, p);
    else if (d->a_kc && d->b_kc && g_g2_dbg == 4) hipLaunchKernelGGL((gemm256_kernel<true, true, 0, false, 4>), grid, block, 0, s, p);
    else if (d->a_kc && d->b_kc && g_g2_dbg == 5) hipLaunchKernelGGL((gemm256_kernel<true, true, 0, false, 5>), grid, block, 0, s, p);
    else if (d->a_kc && d->b_kc && g_g2_dbg == 6) hipLaunchKernelGGL((gemm256_kernel<true, true, 0, false, 6>), grid, block, 0, s, p);
    else if (d->a_kc && d->b_kc && g_g2_sched == 1) hipLaunchKernelGGL((gemm256_kernel<true, true, 0, false, 0, 1>), grid, block, 0, s, p);
    else if (d->a_kc && d->b_kc) hipLaunchKernelGGL((gemm256_kernel<true, true, 0>), grid, block, 0, s, p);
    else if (d->a_kc && !d->b_kc) hipLaunchKernelGGL((gemm256_kernel<true, false, 0>), grid, block, 0, s, p);
    else if (!d->a_kc && d->b_kc) hipLaunchKernelGGL((gemm256_kernel<false, true, 0>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((gemm256_kernel<false, false, 0>), grid, block, 0, s, p);
  } else if (epi == 2) {
    hipLaunchKernelGGL((gemm256_kernel<true, true, 2>), grid, block, 0, s, p);
  } else if (epi == 3) {
    hipLaunchKernelGGL((gemm256_kernel<true, false, 3>), grid, block, 0, s, p);
  } else {
    hipLaunchKernelGGL((gemm256_kernel<true, false, 1>), grid, block, 0, s, p);
  }
  return ivh_host::check_launch("gemm256_bf16");
}


// e4m3 operands (ivh_gemm_fp8, gemm_fp8.hip): bf16 output, K-contiguous operands, plain / GELU(+gelu') / x gelu' epilogues, problems
// large enough for 256 x 256 tiles.  Returns 1 when the problem is not for this kernel (the caller uses the 128^2 e4m3 kernel).
extern "C" int ivh_gemm256_fp8_launch(const ivh_gemm_desc* d, const float* scale_a, const float* scale_b, int scale_b_vec, void* stream) {
  using namespace ivh;
  if (!d->a_kc || !d->b_kc || d->c_fp32 || d->batch > 1 || d->colsum_part || d->act == 2 || d->m_dev || d->k_dev) return 1;
  if (d->dact_in && (d->act != 3 || d->preact)) return 1;
  if (!d->dact_in && d->act == 1 && d->preact) return 1;
  if (!d->dact_in && d->act == 0 && d->preact) return 1;
  if ((long)d->M * d->N < 512L * 512 || d->K < 512) return 1;
  const long lim = (1L << 31) - (1L << 24);
  const long a_bytes = ((long)d->M - 1) * d->lda + d->K, b_bytes = ((long)d->N - 1) * d->ldb + d->K;
  Gemm256Params p{};
  p.c_bytes = (((long)d->M - 1) * d->ldc + d->N) * 2;
  p.p_bytes = d->preact ? (((long)d->M - 1) * d->ldp + d->N) * 2 : 0;
  p.d_bytes = d->dact_in ? (((long)d->M - 1) * d->ldd + d->N) * 2 : 0;
  if (a_bytes >= lim || b_bytes >= lim || p.c_bytes >= lim || p.p_bytes >= lim || p.d_bytes >= lim) return 1;
  if (d->lda >= (1L << 31) || d->ldb >= (1L << 31) || d->ldc >= (1L << 31) || d->ldp >= (1L << 31) || d->ldd >= (1L << 31)) return 1;
  p.A = d->A; p.B = d->B; p.lda = (int)d->lda; p.ldb = (int)d->ldb; p.M = d->M; p.N = d->N; p.K = d->K;
  p.C = d->C; p.ldc = (int)d->ldc; p.c_fp32 = 0; p.bias = d->bias; p.act = d->act;
  p.preact = d->preact; p.ldp = (int)d->ldp; p.dact_in = d->dact_in; p.ldd = (int)d->ldd;
  p.alpha = d->alpha; p.tiles_m = (d->M + G2_BM - 1) / G2_BM; p.tiles_n = (d->N + G2_BN - 1) / G2_BN;
  p.batch = 1; p.a_bytes = a_bytes; p.b_bytes = b_bytes;
  p.bias_bytes = d->bias ? (long)d->N * 4 : 0;
  p.scale_a = scale_a; p.scale_b = scale_b; p.scale_b_vec = scale_b_vec;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) n_cu = 256;
    else n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const long total = (long)p.tiles_m * p.tiles_n;
  p.total_tiles = (int)total; p.nprob = 0; p.stagger = 0; p.debug_skip_stores = 0; p.debug_stamps = nullptr;
  const long cap = g_g2_max_wg > 0 ? g_g2_max_wg : n_cu;
  hipStream_t s = (hipStream_t)stream;
  const int epi = d->dact_in ? 3 : (d->act ? 2 : 0);
  if (g2_apply_split(d, 1, p, cap, s)) {
    dim3 grid((unsigned)(p.total_tiles < cap ? p.total_tiles : cap), 1, 1), block(512);
    if (epi == 0) hipLaunchKernelGGL((gemm256_kernel<true, true, 0, false, 0, 0, true, true>), grid, block, 0, s, p);
    else if (epi == 2) hipLaunchKernelGGL((gemm256_kernel<true, true, 2, false, 0, 0, true, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((gemm256_kernel<true, true, 3, false, 0, 0, true, true>), grid, block, 0, s, p);
    return ivh_host::check_launch("gemm256_fp8 (tail split)") ? -1 : 0;
  }
  dim3 grid((unsigned)(total < cap ? total : cap), 1, 1), block(512);
  if (epi == 0) hipLaunchKernelGGL((gemm256_kernel<true, true, 0, false, 0, 0, true>), grid, block, 0, s, p);
  else if (epi == 2) hipLaunchKernelGGL((gemm256_kernel<true, true, 2, false, 0, 0, true>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((gemm256_kernel<true, true, 3, false, 0, 0, true>), grid, block, 0, s, p);
  return ivh_host::check_launch("gemm256_fp8") ? -1 : 0;
}


// Grouped launch: n <= 32 problems with the same K and operand layouts, plain bf16 output (no bias / activation / batch), one
// persistent kernel over the concatenated tile lists.  Returns 1 if this set cannot be grouped (the caller launches them one
// by one), 0 when enqueued, < 0 on error.
extern "C" int ivh_gemm256_grouped_launch(const ivh_gemm_desc* d, int n, void* stream) {
  using namespace ivh;
  if (n < 2 || n > 32) return 1;
  for (int i = 0; i < n; ++i) {
    const ivh_gemm_desc& q = d[i];
    if (q.a_kc != d[0].a_kc || q.b_kc != d[0].b_kc || q.a_kc || q.b_kc) return 1;                       // built for the wgrad layout
    if (q.bias || q.act || q.preact || q.dact_in || q.c_fp32 || q.alpha != 1.0f || (q.batch > 1)) return 1;
    if (!ivh_gemm256_supported(&q)) return 1;
  }
  Gemm256Params p{};
  int kmax = 0;
  for (int i = 0; i < n; ++i) kmax = d[i].K > kmax ? d[i].K : kmax;
  p.K = kmax; p.alpha = 1.0f; p.batch = 1; p.nprob = n;                  // problems may differ in K (per-problem K: the DYN grouped kernel)
  const long lim = (1L << 31) - (1L << 24);
  int tile = 0;
  bool dyn = false;                                      // some problem's contraction length lives in device memory (ivh_gemm_desc.k_dev)
  for (int i = 0; i < n; ++i) {
    const ivh_gemm_desc& q = d[i];
    G2Prob& g = p.prob[i];
    g.A = q.A; g.B = q.B; g.C = q.C; g.lda = (int)q.lda; g.ldb = (int)q.ldb; g.ldc = (int)q.ldc; g.M = q.M; g.N = q.N;
    g.a_bytes = (((long)q.K - 1) * q.lda + q.M) * 2; g.b_bytes = (((long)q.K - 1) * q.ldb + q.N) * 2;
    g.c_bytes = (((long)q.M - 1) * q.ldc + q.N) * 2;
    if (g.a_bytes >= lim || g.b_bytes >= lim || g.c_bytes >= lim) return 1;          // too large for 32-bit offsets: launched one by one
    IVH_REQUIRE(q.lda < (1L << 31) && q.ldb < (1L << 31) && q.ldc < (1L << 31), "gemm256 grouped: leading dimension does not fit 31 bits");
    g.tiles_m = (q.M + G2_BM - 1) / G2_BM; g.tiles_n = (q.N + G2_BN - 1) / G2_BN; g.tile_begin = tile;
    g.k_dev = q.k_dev; g.K = q.K; dyn = dyn || q.k_dev != nullptr || q.K != d[0].K;
    if (q.m_dev) return 1;                               // (a row count on the output side: not a weight gradient)
    tile += g.tiles_m * g.tiles_n;
  }
  // the non-grouped fields the kernel still reads
  p.A = d[0].A; p.B = d[0].B; p.C = d[0].C; p.lda = (int)d[0].lda; p.ldb = (int)d[0].ldb; p.ldc = (int)d[0].ldc;
  p.M = d[0].M; p.N = d[0].N; p.tiles_m = p.prob[0].tiles_m; p.tiles_n = p.prob[0].tiles_n;
  p.a_bytes = p.prob[0].a_bytes; p.b_bytes = p.prob[0].b_bytes; p.c_bytes = p.prob[0].c_bytes;
  p.total_tiles = tile;
  p.stagger = 0; p.debug_skip_stores = g_g2_skip_stores; p.debug_stamps = nullptr;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) n_cu = 256;
    else n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const long cap = g_g2_max_wg > 0 ? g_g2_max_wg : n_cu;
  dim3 grid((unsigned)(tile < cap ? tile : cap), 1, 1), block(512);
  if (dyn) hipLaunchKernelGGL((gemm256_kernel<false, false, 0, true, 0, 0, false, false, false, true>), grid, block, 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL((gemm256_kernel<false, false, 0, true>), grid, block, 0, (hipStream_t)stream, p);
  return ivh_host::check_launch("gemm256_grouped");
}
