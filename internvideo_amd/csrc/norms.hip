// Row-wise HBM-bound kernels of the block: residual add + LayerScale + DropPath + RMSNorm (fwd / bwd),
// q/k RMSNorm over the concatenated-heads axis (fwd / bwd), LayerNorm + l2-normalise decoder tail with the
// fused cosine distillation loss (fwd / bwd), and the deterministic column reductions (bias / weight grads).
//
// Layout: one 64-lane wave owns one row; a lane owns the 16-byte (8 x bf16) / 32-byte (8 x fp32) chunks
// lane, lane + 64, ... of the row, so every wave-level load is 1 KiB (bf16) / 2 KiB (fp32) contiguous.  The row
// stays in registers between the reduction and the scaling pass: each byte is read from HBM once.
#include <stdlib.h>
#include "common.h"
#include "../../include/internvideo_hip.h"

namespace ivh {

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

// the residual stream is fp32 (parity default) or bf16 (what the reference's bf16 recipe carries: DropoutAddRMSNorm(prenorm=True,
// residual_in_fp32=False), internvideo2_pretrain.py:283-286, 467): 8 elements of either
__device__ __forceinline__ void ld8r(const float* p, float* o) { ld8f(p, o); }
__device__ __forceinline__ void ld8r(const bf16_t* p, float* o) { ld8b(p, o); }
__device__ __forceinline__ void st8r(float* p, const float* v) { st8f(p, v); }
__device__ __forceinline__ void st8r(bf16_t* p, const float* v) { st8b(p, v); }

// Sum over a whole row.  WPR = 1: the row lives in one wave.  WPR = 4 (rows wider than 2048 elements): the four waves of the
// workgroup share the row -- wave w owns the chunks {lane + 64 * (4 i + w)} -- so that the per-lane state is that of a row a
// quarter as wide (the one-wave form of the backward kernels needs > 256 VGPRs at D = 3200 and spills); partial sums meet in LDS,
// one barrier per reduction, two slots alternating so that a wave that races ahead cannot overwrite what a slower one still reads.
template <int WPR>
__device__ __forceinline__ float row_sum(float* xch, int& par, float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if constexpr (WPR == 1) {
    return v;
  } else {
    static_assert(WPR == 4, "a row is owned by one wave or by the whole workgroup");
    if ((threadIdx.x & 63) == 0) xch[par * 4 + (threadIdx.x >> 6)] = v;
    __syncthreads();
    const float r = (xch[par * 4] + xch[par * 4 + 1]) + (xch[par * 4 + 2] + xch[par * 4 + 3]);
    par ^= 1;
    return r;
  }
}

// two sums in one exchange (LayerNorm backward's mean(g) and mean(g * xhat))
template <int WPR>
__device__ __forceinline__ void row_sum2(float* xch2, int& par, float& a, float& b) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
  if constexpr (WPR > 1) {
    if ((threadIdx.x & 63) == 0) { xch2[(par * 4 + (threadIdx.x >> 6)) * 2] = a; xch2[(par * 4 + (threadIdx.x >> 6)) * 2 + 1] = b; }
    __syncthreads();
    const float* q = xch2 + par * 8;
    a = (q[0] + q[2]) + (q[4] + q[6]);
    b = (q[1] + q[3]) + (q[5] + q[7]);
    par ^= 1;
  }
}

// ---------------------------------------------------------------------------------------------------------
// res_out = res_in + rowscale * gamma * branch ;  y = rmsnorm(res_out) * w
// SKIP (DropPath sample skipping, ivh_droppath_plan): `branch` holds only the samples its DropPath kept, compacted in sample order --
// sample s lives at rows branch_slot[s] * rows_per_sample ... (-1: dropped, the stream passes through unchanged and nothing of `branch` is
// read: the rows a skipped GEMM never wrote may hold anything) -- and y is written compacted by y_slot, the keep map of the branch that
// CONSUMES it (a dropped sample's y is not computed).  res_out and rstd keep the stream's own row numbering.  NULL map = identity.
template <int NCH, int WPR = 1, typename TR = float, bool SKIP = false>
__global__ __launch_bounds__(256) void rmsnorm_add_fwd_kernel(
    const TR* __restrict__ res_in, const bf16_t* __restrict__ branch, const float* __restrict__ gamma,
    const float* __restrict__ rowscale, int rows_per_sample, const float* __restrict__ w, float eps, int M, int D,
    TR* __restrict__ res_out, bf16_t* __restrict__ y, float* __restrict__ rstd_out,
    const int* __restrict__ branch_slot = nullptr, const int* __restrict__ y_slot = nullptr) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ float rs_xch[8], rs_xch2[16];
  int rs_par = 0, rs_par2 = 0;
  (void)rs_xch; (void)rs_par; (void)rs_xch2; (void)rs_par2;
  const int nch = D >> 3;
  for (int row = blockIdx.x * (4 / WPR) + wave / WPR; row < M; row += gridDim.x * (4 / WPR)) {
    const float rs = rowscale ? rowscale[row / rows_per_sample] : 1.0f;
    long brow = row, yrow = row;                               // rows of this token in `branch` / in y
    bool has_b = branch != nullptr, has_y = y != nullptr;
    if constexpr (SKIP) {
      const int smp = row / rows_per_sample, tok = row - smp * rows_per_sample;
      if (branch_slot) { const int sl = branch_slot[smp]; has_b = has_b && sl >= 0; brow = (long)sl * rows_per_sample + tok; }
      if (y_slot) { const int sl = y_slot[smp]; has_y = has_y && sl >= 0; yrow = (long)sl * rows_per_sample + tok; }
    }
    float x[NCH][8];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * (i * WPR + wave % WPR);
      if (c < nch) {
        const long off = (long)row * D + c * 8;
        float r[8], b[8];
        if (res_in) ld8r(res_in + off, r);
        else {
#pragma unroll
          for (int e = 0; e < 8; ++e) r[e] = 0.f;
        }
        if (has_b) {
          ld8b(branch + (SKIP ? brow * D + c * 8 : off), b);
          if (gamma) {
            float gm[8];
            ld8f(gamma + c * 8, gm);
#pragma unroll
            for (int e = 0; e < 8; ++e) b[e] *= gm[e];
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) r[e] += rs * b[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { x[i][e] = r[e]; ss += r[e] * r[e]; }     // the norm sees the sum before it is rounded to the stream's type
        if (res_out) st8r(res_out + off, r);
      }
    }
    if (has_y) {                                               // (uniform over the waves that share a row: the barrier inside row_sum is safe)
      ss = row_sum<WPR>(rs_xch, rs_par, ss);
      const float rstd = rsqrtf(ss / (float)D + eps);
      if (rstd_out && lane == 0) rstd_out[row] = rstd;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * (i * WPR + wave % WPR);
        if (c < nch) {
          float wv[8], o[8];
          ld8f(w + c * 8, wv);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = x[i][e] * rstd * wv[e];
          st8b(y + yrow * D + c * 8, o);
        }
      }
    }
  }
}

// backward.  dres = dres_out + rmsnorm_bwd(dy);  dres_in = dres;  dbranch = rowscale*gamma*dres;
// dw += dy * xhat ; dgamma += rowscale * branch * dres   (column sums -> per-block partials)
// dres_extra (optional, the stream's type): a second gradient of the same rows that joins dres_out on load -- the gradient of a feature tap
// (the decoder that read this block's input), which would otherwise cost a read-modify-write pass of its own over dres_out.
// SKIP: dy is compacted by y_slot, branch / dbranch by branch_slot (see rmsnorm_add_fwd_kernel): a sample whose y was dropped has no dy (the
// norm contributes nothing to its dres or to dw), a sample whose branch was dropped gets no dbranch row and adds nothing to dgamma / dbias.
template <int NCH, int WPR = 1, typename TR = float, bool SKIP = false>
__global__ __launch_bounds__(256) void rmsnorm_add_bwd_kernel(
    const bf16_t* __restrict__ dy, const TR* __restrict__ dres_out, const TR* __restrict__ res_out,
    const float* __restrict__ rstd_in, const float* __restrict__ w, const bf16_t* __restrict__ branch,
    const float* __restrict__ gamma, const float* __restrict__ rowscale, int rows_per_sample, int M, int D,
    TR* __restrict__ dres_in, bf16_t* __restrict__ dbranch, float* __restrict__ dw_part, float* __restrict__ dgamma_part,
    float* __restrict__ dbias_part, const TR* __restrict__ dres_extra = nullptr,
    const int* __restrict__ y_slot = nullptr, const int* __restrict__ branch_slot = nullptr) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [4][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ float rs_xch[8], rs_xch2[16];
  int rs_par = 0, rs_par2 = 0;
  (void)rs_xch; (void)rs_par; (void)rs_xch2; (void)rs_par2;
  const int nch = D >> 3;
  float aw[NCH][8], ag[NCH][8], ab[NCH][8];                     // column sums: dy*xhat (dw), rs*branch*dres (dgamma), dbranch (bias of the
#pragma unroll                                                  // Linear that produced `branch`: its gradient is colsum(dbranch))
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) { aw[i][e] = 0.f; ag[i][e] = 0.f; ab[i][e] = 0.f; }

  for (int row = blockIdx.x * (4 / WPR) + wave / WPR; row < M; row += gridDim.x * (4 / WPR)) {
    const float rs = rowscale ? rowscale[row / rows_per_sample] : 1.0f;
    long yrow = row, brow = row;
    bool has_dy = dy != nullptr, has_br = dbranch != nullptr;
    if constexpr (SKIP) {
      const int smp = row / rows_per_sample, tok = row - smp * rows_per_sample;
      if (y_slot) { const int sl = y_slot[smp]; has_dy = has_dy && sl >= 0; yrow = (long)sl * rows_per_sample + tok; }
      if (branch_slot) { const int sl = branch_slot[smp]; has_br = has_br && sl >= 0; brow = (long)sl * rows_per_sample + tok; }
    }
    float dr[NCH][8];     // running dres
    float xh[NCH][8], wdy[NCH][8];
    float dot = 0.f;
    const float rstd = has_dy ? rstd_in[row] : 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * (i * WPR + wave % WPR);
      if (c < nch) {
        const long off = (long)row * D + c * 8;
        if (dres_out) ld8r(dres_out + off, dr[i]);
        else {
#pragma unroll
          for (int e = 0; e < 8; ++e) dr[i][e] = 0.f;
        }
        if (dres_extra) {
          float ex[8];
          ld8r(dres_extra + off, ex);
#pragma unroll
          for (int e = 0; e < 8; ++e) dr[i][e] += ex[e];
        }
        if (has_dy) {
          float xv[8], dv[8], wv[8];
          ld8r(res_out + off, xv);
          ld8b(dy + (SKIP ? yrow * D + c * 8 : off), dv);
          ld8f(w + c * 8, wv);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            xh[i][e] = xv[e] * rstd;
            wdy[i][e] = wv[e] * dv[e];
            dot += wdy[i][e] * xh[i][e];
            aw[i][e] += dv[e] * xh[i][e];
          }
        }
      }
    }
    if (has_dy) {
      dot = row_sum<WPR>(rs_xch, rs_par, dot) / (float)D;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * (i * WPR + wave % WPR);
        if (c < nch) {
#pragma unroll
          for (int e = 0; e < 8; ++e) dr[i][e] += rstd * (wdy[i][e] - xh[i][e] * dot);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * (i * WPR + wave % WPR);
      if (c < nch) {
        const long off = (long)row * D + c * 8;
        if (dres_in) st8r(dres_in + off, dr[i]);
        if (has_br) {
          const long boff = SKIP ? brow * D + c * 8 : off;
          float o[8], gm[8];
          if (gamma) ld8f(gamma + c * 8, gm);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = rs * (gamma ? gm[e] : 1.0f) * dr[i][e];
          st8b(dbranch + boff, o);
          if (dbias_part) {
#pragma unroll
            for (int e = 0; e < 8; ++e) ab[i][e] += o[e];
          }
          if (dgamma_part && branch) {
            float b[8];
            ld8b(branch + boff, b);
#pragma unroll
            for (int e = 0; e < 8; ++e) ag[i][e] += rs * b[e] * dr[i][e];
          }
        }
      }
    }
  }
  // block reduction of the column accumulators: 4 waves -> 1 partial row per block
  for (int pass = 0; pass < 3; ++pass) {
    float* dst = pass == 0 ? dw_part : (pass == 1 ? dgamma_part : dbias_part);
    if (!dst) continue;
    __syncthreads();
    if constexpr (WPR > 1) {                                  // a wave owns a quarter of the columns: the rest of its region is zero
      for (int d = threadIdx.x; d < 4 * D; d += 256) red[d] = 0.f;
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * (i * WPR + wave % WPR);
      if (c < nch) st8f(red + wave * D + c * 8, pass == 0 ? aw[i] : (pass == 1 ? ag[i] : ab[i]));
    }
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += 256)
      dst[(long)blockIdx.x * D + d] = red[d] + red[D + d] + red[2 * D + d] + red[3 * D + d];
  }
}

// gfx950 STORE-DATA HAZARD (found in round 5 with rmsnorm_add_bwd_b16_kernel<2, 1, true>; tools/isa_store_hazard_scan.py,
// profiles/r5_store_data_hazard_gfx950.txt).  A 128-bit MUBUF store with an SGPR soffset keeps reading its data registers for a few issue
// slots after it issues (dword k about k slots later); `buffer_store_dwordx4 v[82:85], v195, s[48:51], s63 offen` followed DIRECTLY by
// `v_pk_mul_f32 v[82:83], ...` stored garbage in dword 1 of lanes 12-15 of every 16.  LLVM's hazard recognizer covers this only for stores
// WITHOUT a register soffset (GCNHazardRecognizer::createsVALUHazard), so nothing is inserted for the scalar-row-offset stores of the
// bytes-in-flight kernels below.  Every such store is therefore followed by four wait states, fenced so that no VALU instruction can be
// scheduled between the store and the nop.  (Whether a variant was hit was a matter of register allocation: the default instantiations
// were clean, ROWS = 2 / 4 and the EXTRA variant were not -- the "wrong rows" of the first bytes-in-flight q/k backward in round 3 fit.)
__device__ __forceinline__ void store_data_settle() {
#ifndef IVH_NO_STORE_SETTLE                                      // (the A/B build of the measurement in profiles/r5_store_data_hazard_gfx950.txt)
  asm volatile("s_nop 3");
  __builtin_amdgcn_sched_barrier(0);
#endif
}

// The same backward for the bf16 residual stream, organised for bytes in flight -- the configuration every block interior runs (dy, the
// incoming stream gradient, LayerScale's gamma and the branch all present, all three column sums wanted; anything else takes the generic
// kernel above).  With 2-byte rows the generic kernel (238 VGPRs: two waves per SIMD, one row each) keeps only ~70 KB of loads in flight per
// CU and stops at 3.8 TB/s.  Here the four waves of a workgroup SHARE ROWS rows per trip -- wave w owns the 16-byte chunks lane + 64 w
// (+ 256 i) of every row, so a lane carries 8 NCH columns of the three column sums (24 NCH registers instead of 72 at D = 1408) -- and
// every operand of all ROWS rows (dres, x, dy, branch) is requested before anything is computed and held as RAW bf16 (4 registers per 8
// elements); x_hat and w * dy are recomputed in the second pass instead of kept in fp32.  The per-row dot products meet in LDS: ONE
// barrier per trip of ROWS rows (two slot sets alternate, so a wave that races ahead cannot overwrite what a slower one still reads).
// No cross-wave reduction of the column sums at the end: every column belongs to exactly one lane of the workgroup.
// Row operands go through buffer descriptors: per-lane byte offset (constant over the kernel; out-of-row chunks point past the buffer:
// they read zeros and their stores are dropped) + a SCALAR row offset -- no 64-bit per-row addresses in vector registers, no predicates.
// The scalar offset is NOT part of the hardware's range check (gfx9 raw buffers check the vector offset only), so a row at or past M
// swaps every lane's offset for the out-of-range marker: it reads zeros and stores nothing.  (The launcher keeps M * D * 2 below 2 GiB.)
// EXTRA: a fifth row operand, dres_extra (see rmsnorm_add_bwd_kernel), requested with the others and added to dres in fp32.
// SKIP: dy compacted by y_slot, branch / dbranch by branch_slot (see rmsnorm_add_fwd_kernel).  Only the SCALAR row offsets of those three
// operands change; a dropped sample's operand takes the out-of-range marker instead (dy / branch read as zeros: exactly what they contribute,
// its dbranch store is dropped) and its rstd -- never written by the forward -- is replaced by 0.
template <int NCH, int ROWS, bool EXTRA = false, bool SKIP = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void rmsnorm_add_bwd_b16_kernel(
    const bf16_t* __restrict__ dy, const bf16_t* __restrict__ dres_out, const bf16_t* __restrict__ res_out,
    const float* __restrict__ rstd_in, const float* __restrict__ w, const bf16_t* __restrict__ branch,
    const float* __restrict__ gamma, const float* __restrict__ rowscale, int rows_per_sample, int M, int D,
    bf16_t* __restrict__ dres_in, bf16_t* __restrict__ dbranch, float* __restrict__ dw_part, float* __restrict__ dgamma_part,
    float* __restrict__ dbias_part, const bf16_t* __restrict__ dres_extra = nullptr,
    const int* __restrict__ y_slot = nullptr, const int* __restrict__ branch_slot = nullptr) {
  __shared__ float xch[2][4][ROWS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = D >> 3;
  const int bytes = M * D * 2;
  const __amdgpu_buffer_rsrc_t rs_dres = __builtin_amdgcn_make_buffer_rsrc((void*)dres_out, 0, bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)res_out, 0, bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_br = __builtin_amdgcn_make_buffer_rsrc((void*)branch, 0, bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_din = __builtin_amdgcn_make_buffer_rsrc((void*)dres_in, 0, bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dbr = __builtin_amdgcn_make_buffer_rsrc((void*)dbranch, 0, bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_ex = __builtin_amdgcn_make_buffer_rsrc((void*)(EXTRA ? dres_extra : dres_out), 0, bytes, 0x00020000);
  unsigned voff[NCH];
  float aw[NCH][8], ag[NCH][8], ab[NCH][8];
  float wv[NCH][8], gm[NCH][8];                                 // this lane's columns of the norm weight and of LayerScale's gamma
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * wave + 256 * i;
    voff[i] = c < nch ? (unsigned)(c * 16) : 0x80000000u;
#pragma unroll
    for (int e = 0; e < 8; ++e) { aw[i][e] = 0.f; ag[i][e] = 0.f; ab[i][e] = 0.f; wv[i][e] = 0.f; gm[i][e] = 0.f; }
    if (c < nch) { ld8f(w + c * 8, wv[i]); ld8f(gamma + c * 8, gm[i]); }
  }
  int par = 0;
  const int row_bytes = D * 2;
  const float inv_d = 1.0f / (float)D;
  // The operands of trip t + 1 are requested before trip t is computed (a second raw register set): a trip's ~5 us of load latency then
  // overlaps the previous trip's arithmetic, barrier and stores instead of following them.
  u32x4 rdr[ROWS][NCH], rx[ROWS][NCH], rdy[ROWS][NCH], rbr[ROWS][NCH];
  u32x4 ndr[ROWS][NCH], nx[ROWS][NCH], ndy[ROWS][NCH], nbr[ROWS][NCH];
  u32x4 rex[EXTRA ? ROWS : 1][EXTRA ? NCH : 1], nex[EXTRA ? ROWS : 1][EXTRA ? NCH : 1];
  // SKIP: where stream row `row` lives in dy (through y_slot) and in branch / dbranch (through branch_slot): presence + scalar byte offsets.
  // Worked out ONCE per row, when its operands are requested (one trip ahead of their use), and carried to the trip that computes and stores
  // the row: the two slot loads and the sample index then sit a whole trip of latency away from the stores that need them.  The sample index
  // is a multiply-high with a precomputed reciprocal (exact for row < 2^40 / rows_per_sample; the launcher keeps M * D * 2 below 2^31).
  struct RowMap { bool ok_dy, ok_br; int so_dy, so_br; };
  RowMap cmap[ROWS], nmap[ROWS];
  const unsigned long long rps_magic = SKIP ? ((1ull << 40) + (unsigned long long)rows_per_sample - 1ull) / (unsigned long long)rows_per_sample : 0ull;
  auto map_row = [&](int row, bool ok) __attribute__((always_inline)) {
    RowMap m{ok, ok, ok ? row * row_bytes : 0, ok ? row * row_bytes : 0};
    if (ok) {
      const int smp = (int)(((unsigned long long)(unsigned)row * rps_magic) >> 40);
      const int tok = row - smp * rows_per_sample;
      if (y_slot) { const int sl = y_slot[smp]; m.ok_dy = sl >= 0; m.so_dy = m.ok_dy ? (sl * rows_per_sample + tok) * row_bytes : 0; }
      if (branch_slot) { const int sl = branch_slot[smp]; m.ok_br = sl >= 0; m.so_br = m.ok_br ? (sl * rows_per_sample + tok) * row_bytes : 0; }
    }
    return m;
  };
  auto fetch = [&](int r0, u32x4 (&fdr)[ROWS][NCH], u32x4 (&fx)[ROWS][NCH], u32x4 (&fdy)[ROWS][NCH], u32x4 (&fbr)[ROWS][NCH],
                   u32x4 (&fex)[EXTRA ? ROWS : 1][EXTRA ? NCH : 1], RowMap (&fmap)[ROWS]) __attribute__((always_inline)) {
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) {
      const bool ok = r0 + rr < M;                               // scalar
      const int so = ok ? (r0 + rr) * row_bytes : 0;
      bool ok_dy = ok; int so_dy = so;
      if constexpr (SKIP) { fmap[rr] = map_row(r0 + rr, ok); ok_dy = fmap[rr].ok_dy; so_dy = fmap[rr].so_dy; }
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const unsigned vo = ok ? voff[i] : 0x80000000u;
        fx[rr][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, vo, so, 0);
        fdy[rr][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_dy, SKIP ? (ok_dy ? voff[i] : 0x80000000u) : vo, so_dy, 0);
      }
    }
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) {
      const bool ok = r0 + rr < M;
      const int so = ok ? (r0 + rr) * row_bytes : 0;
      bool ok_br = ok; int so_br = so;
      if constexpr (SKIP) { ok_br = fmap[rr].ok_br; so_br = fmap[rr].so_br; }
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const unsigned vo = ok ? voff[i] : 0x80000000u;
        fdr[rr][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_dres, vo, so, 0);
        fbr[rr][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_br, SKIP ? (ok_br ? voff[i] : 0x80000000u) : vo, so_br, 0);
        if constexpr (EXTRA) fex[rr][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_ex, vo, so, 0);
      }
    }
  };
  const int stride = gridDim.x * ROWS;
  fetch(blockIdx.x * ROWS, rdr, rx, rdy, rbr, rex, cmap);
  for (int row0 = blockIdx.x * ROWS; row0 < M; row0 += stride) {
    const int nrow = row0 + stride < M ? row0 + stride : M;      // nothing left: a fetch of rows past M returns zeros and moves no data
    fetch(nrow, ndr, nx, ndy, nbr, nex, nmap);
    float rstd[ROWS], k2[ROWS], rsc[ROWS];
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) {
      const int row = row0 + rr < M ? row0 + rr : M - 1;         // scalar loads; a row past M contributes zeros whatever they return
      rstd[rr] = rstd_in[row];
      rsc[rr] = rowscale ? rowscale[row / rows_per_sample] : 1.0f;
      if constexpr (SKIP) {                                      // a sample without y has no rstd (the forward never wrote it): 0, not garbage
        if (!cmap[rr].ok_dy) rstd[rr] = 0.f;
      }
    }
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) {
      float d = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {                            // chunks past the row and rows past M hold zeros: they add nothing
        float xv[8], dv[8];
        asm volatile("" : "+v"(rx[rr][i]), "+v"(rdy[rr][i]));    // stay packed until here: an early unpack doubles the registers of
        unpack8(rx[rr][i], xv); unpack8(rdy[rr][i], dv);         // every row in flight
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = xv[e] * rstd[rr];
          d += wv[i][e] * dv[e] * xh;
          aw[i][e] += dv[e] * xh;
        }
      }
      d = wave_sum(d);
      if (lane == 0) xch[par][wave][rr] = d;
      __builtin_amdgcn_sched_barrier(0);                         // one row at a time: interleaving the rows multiplies the fp32 temporaries
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr)
      k2[rr] = ((xch[par][0][rr] + xch[par][1][rr]) + (xch[par][2][rr] + xch[par][3][rr])) * inv_d * rstd[rr] * rstd[rr];
    par ^= 1;
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) {
      const bool ok = row0 + rr < M;
      const int so = ok ? (row0 + rr) * row_bytes : 0;
      bool ok_br = ok; int so_br = so;
      if constexpr (SKIP) { ok_br = cmap[rr].ok_br; so_br = cmap[rr].so_br; }
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const unsigned vo = ok ? voff[i] : 0x80000000u;
        float dr[8], xv[8], dv[8], o[8], bb[8];
        asm volatile("" : "+v"(rdr[rr][i]), "+v"(rx[rr][i]), "+v"(rdy[rr][i]), "+v"(rbr[rr][i]));   // unpacked again, not kept in fp32 across the barrier
        unpack8(rdr[rr][i], dr); unpack8(rx[rr][i], xv); unpack8(rdy[rr][i], dv); unpack8(rbr[rr][i], bb);
        if constexpr (EXTRA) {
          float ex[8];
          asm volatile("" : "+v"(rex[rr][i]));
          unpack8(rex[rr][i], ex);
#pragma unroll
          for (int e = 0; e < 8; ++e) dr[e] += ex[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          dr[e] += rstd[rr] * wv[i][e] * dv[e] - xv[e] * k2[rr];
          o[e] = rsc[rr] * gm[i][e] * dr[e];
          ab[i][e] += o[e];
          ag[i][e] += rsc[rr] * bb[e] * dr[e];
        }
        u32x4 pdr = pack8(dr), po = pack8(o);
        asm volatile("" : "+v"(pdr), "+v"(po));                  // both packed before the first store: nothing rewrites store data in between
        __builtin_amdgcn_raw_buffer_store_b128(pdr, rs_din, vo, so, 0);
        __builtin_amdgcn_raw_buffer_store_b128(po, rs_dbr, SKIP ? (ok_br ? voff[i] : 0x80000000u) : vo, so_br, 0);
        store_data_settle();
      }
    }
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr)
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        rdr[rr][i] = ndr[rr][i]; rx[rr][i] = nx[rr][i]; rdy[rr][i] = ndy[rr][i]; rbr[rr][i] = nbr[rr][i];
        if constexpr (EXTRA) rex[rr][i] = nex[rr][i];
      }
    if constexpr (SKIP) {
#pragma unroll
      for (int rr = 0; rr < ROWS; ++rr) cmap[rr] = nmap[rr];
    }
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * wave + 256 * i;
    if (c < nch) {
      st8f(dw_part + (long)blockIdx.x * D + c * 8, aw[i]);
      st8f(dgamma_part + (long)blockIdx.x * D + c * 8, ag[i]);
      st8f(dbias_part + (long)blockIdx.x * D + c * 8, ab[i]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// q/k RMSNorm over the full D axis, in place on packed qkv [M][3][D]
template <int NCH, int WPR = 1>
__global__ __launch_bounds__(256) void qk_rmsnorm_fwd_kernel(bf16_t* __restrict__ qkv, const float* __restrict__ wq,
                                                             const float* __restrict__ wk, float eps, int M, int D,
                                                             float* __restrict__ rstd_q, float* __restrict__ rstd_k,
                                                             const int* __restrict__ m_dev = nullptr) {
  if (m_dev) M = max(0, min(M, *m_dev));                       // device-side row count (ivh_qk_rmsnorm_fwd_dyn): rows past it are not touched
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ float rs_xch[8], rs_xch2[16];
  int rs_par = 0, rs_par2 = 0;
  (void)rs_xch; (void)rs_par; (void)rs_xch2; (void)rs_par2;
  const int nch = D >> 3;
  for (int row = blockIdx.x * (4 / WPR) + wave / WPR; row < M; row += gridDim.x * (4 / WPR)) {
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      bf16_t* base = qkv + (long)row * 3 * D + which * D;
      const float* wv_ = which == 0 ? wq : wk;
      float x[NCH][8];
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * (i * WPR + wave % WPR);
        if (c < nch) {
          ld8b(base + c * 8, x[i]);
#pragma unroll
          for (int e = 0; e < 8; ++e) ss += x[i][e] * x[i][e];
        }
      }
      ss = row_sum<WPR>(rs_xch, rs_par, ss);
      const float rstd = rsqrtf(ss / (float)D + eps);
      if (lane == 0) (which == 0 ? rstd_q : rstd_k)[row] = rstd;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * (i * WPR + wave % WPR);
        if (c < nch) {
          float wv[8], o[8];
          ld8f(wv_ + c * 8, wv);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = x[i][e] * rstd * wv[e];
          st8b(base + c * 8, o);
        }
      }
    }
  }
}

// backward: qkv holds q_hat*w (normalised, weighted); dqkv holds d(q_hat w) -> rewritten to dq.  xhat = y / w.
template <int NCH, int WPR = 1>
__global__ __launch_bounds__(256) void qk_rmsnorm_bwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ dqkv,
                                                             const float* __restrict__ wq, const float* __restrict__ wk,
                                                             const float* __restrict__ rstd_q, const float* __restrict__ rstd_k,
                                                             int M, int D, float* __restrict__ dwq_part, float* __restrict__ dwk_part,
                                                             const int* __restrict__ m_dev = nullptr) {
  if (m_dev) M = max(0, min(M, *m_dev));
  extern __shared__ __attribute__((aligned(16))) float red[];   // [4][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ float rs_xch[8], rs_xch2[16];
  int rs_par = 0, rs_par2 = 0;
  (void)rs_xch; (void)rs_par; (void)rs_xch2; (void)rs_par2;
  const int nch = D >> 3;
  float acc[2][NCH][8];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[a][i][e] = 0.f;
  for (int row = blockIdx.x * (4 / WPR) + wave / WPR; row < M; row += gridDim.x * (4 / WPR)) {
    // all four row segments (q, k and their gradients) are requested before anything is computed: the in-place store of dq would
    // otherwise order the k loads behind it (same buffer), leaving one 2.8 KB segment pair in flight per wave at 2 waves / SIMD
    u32x4 yraw[2][NCH], draw[2][NCH];
#pragma unroll
    for (int which = 0; which < 2; ++which)
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * (i * WPR + wave % WPR);
        if (c < nch) {
          yraw[which][i] = *reinterpret_cast<const u32x4*>(qkv + (long)row * 3 * D + which * D + c * 8);
          draw[which][i] = *reinterpret_cast<const u32x4*>(dqkv + (long)row * 3 * D + which * D + c * 8);
        }
      }
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      bf16_t* dbase = dqkv + (long)row * 3 * D + which * D;
      const float* wv_ = which == 0 ? wq : wk;
      const float rstd = (which == 0 ? rstd_q : rstd_k)[row];
      float xh[NCH][8], wdy[NCH][8];
      float dot = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * (i * WPR + wave % WPR);
        if (c < nch) {
          float yv[8], dv[8], wv[8];
          unpack8(yraw[which][i], yv);
          unpack8(draw[which][i], dv);
          ld8f(wv_ + c * 8, wv);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            xh[i][e] = wv[e] != 0.f ? yv[e] * __builtin_amdgcn_rcpf(wv[e]) : 0.f;     // xhat = y / w (1 ulp rcp: far inside bf16)
            wdy[i][e] = wv[e] * dv[e];
            dot += wdy[i][e] * xh[i][e];
            acc[which][i][e] += dv[e] * xh[i][e];
          }
        }
      }
      dot = row_sum<WPR>(rs_xch, rs_par, dot) / (float)D;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * (i * WPR + wave % WPR);
        if (c < nch) {
          float o[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = rstd * (wdy[i][e] - xh[i][e] * dot);
          st8b(dbase + c * 8, o);
        }
      }
    }
  }
  for (int pass = 0; pass < 2; ++pass) {
    float* dst = pass == 0 ? dwq_part : dwk_part;
    __syncthreads();
    if constexpr (WPR > 1) {                                  // a wave owns a quarter of the columns: the rest of its region is zero
      for (int d = threadIdx.x; d < 4 * D; d += 256) red[d] = 0.f;
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * (i * WPR + wave % WPR);
      if (c < nch) st8f(red + wave * D + c * 8, acc[pass][i]);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += 256)
      dst[(long)blockIdx.x * D + d] = red[d] + red[D + d] + red[2 * D + d] + red[3 * D + d];
  }
}

// The same backward organised for bytes in flight, in the image of rmsnorm_add_bwd_b16_kernel (round 5; the r3 attempt at this produced wrong
// rows and was dropped -- this one is a line-by-line sibling of the residual kernel, whose addressing is covered by guard-row tests).  The four
// waves of a workgroup share a TOKEN: wave w owns the 16-byte chunks lane + 64 w (+ 256 i) of BOTH its q and its k segment, so a lane carries
// 8 NCH columns of each of the two weight-gradient sums (16 NCH registers instead of 48 at D = 1408), its own columns of w and 1 / w stay in
// registers, the four row segments (y_q, y_k, dy_q, dy_k) stay packed bf16 until they are used, and the NEXT token's segments are requested
// before the current one is computed.  One barrier per token (the two dot products meet in LDS; two slot sets alternate).  The row dot product
// <w dy, xhat> is computed as <dy, y>: y = xhat w was stored by the forward (for w = 0 both forms are 0).  In place: every chunk is read and
// rewritten by the same lane.  Addressing as in the residual kernel: per-lane byte offsets through buffer descriptors + a SCALAR token offset
// (not range-checked by the hardware: a token at or past M swaps the vector offsets for the out-of-range marker).
template <int NCH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void qk_rmsnorm_bwd_b16_kernel(
    const bf16_t* __restrict__ qkv, bf16_t* __restrict__ dqkv, const float* __restrict__ wq, const float* __restrict__ wk,
    const float* __restrict__ rstd_q, const float* __restrict__ rstd_k, int M, int D, float* __restrict__ dwq_part, float* __restrict__ dwk_part,
    const int* __restrict__ m_dev = nullptr) {
  if (m_dev) M = max(0, min(M, __builtin_amdgcn_readfirstlane(*m_dev)));
  __shared__ float xch[2][4][2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = D >> 3;
  const int bytes = M * D * 6;                                  // M tokens x 3 D bf16
  const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)qkv, 0, bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc((void*)dqkv, 0, bytes, 0x00020000);
  unsigned voff[2][NCH];
  float acc[2][NCH][8], wv[2][NCH][8], iw[2][NCH][8];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * wave + 256 * i;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      voff[a][i] = c < nch ? (unsigned)((a * D + c * 8) * 2) : 0x80000000u;
#pragma unroll
      for (int e = 0; e < 8; ++e) { acc[a][i][e] = 0.f; wv[a][i][e] = 0.f; iw[a][i][e] = 0.f; }
      if (c < nch) {
        ld8f((a == 0 ? wq : wk) + c * 8, wv[a][i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) iw[a][i][e] = wv[a][i][e] != 0.f ? __builtin_amdgcn_rcpf(wv[a][i][e]) : 0.f;
      }
    }
  }
  const int row_bytes = D * 6;
  const float inv_d = 1.0f / (float)D;
  u32x4 ry[2][NCH], rd[2][NCH], ny[2][NCH], nd[2][NCH];
  auto fetch = [&](int row, u32x4 (&fy)[2][NCH], u32x4 (&fd)[2][NCH]) __attribute__((always_inline)) {
    const bool ok = row < M;                                     // scalar
    const int so = ok ? row * row_bytes : 0;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const unsigned vo = ok ? voff[a][i] : 0x80000000u;
        fy[a][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, vo, so, 0);
        fd[a][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_d, vo, so, 0);
      }
  };
  int par = 0;
  fetch(blockIdx.x, ry, rd);
  for (int row = blockIdx.x; row < M; row += gridDim.x) {
    const int nrow = row + gridDim.x < M ? row + gridDim.x : M;   // nothing left: a fetch past M returns zeros and moves no data
    fetch(nrow, ny, nd);
    const float rstd[2] = {rstd_q[row], rstd_k[row]};             // scalar loads
    float dot[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      float d = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        float yv[8], dv[8];
        asm volatile("" : "+v"(ry[a][i]), "+v"(rd[a][i]));       // stay packed until here
        unpack8(ry[a][i], yv); unpack8(rd[a][i], dv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          d += dv[e] * yv[e];
          acc[a][i][e] += dv[e] * (yv[e] * iw[a][i][e]);
        }
      }
      d = wave_sum(d);
      if (lane == 0) xch[par][wave][a] = d;
      dot[a] = 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 2; ++a) dot[a] = ((xch[par][0][a] + xch[par][1][a]) + (xch[par][2][a] + xch[par][3][a])) * inv_d;
    par ^= 1;
    const int so = row * row_bytes;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        float yv[8], dv[8], o[8];
        asm volatile("" : "+v"(ry[a][i]), "+v"(rd[a][i]));       // unpacked again, not kept in fp32 across the barrier
        unpack8(ry[a][i], yv); unpack8(rd[a][i], dv);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = rstd[a] * (wv[a][i][e] * dv[e] - yv[e] * iw[a][i][e] * dot[a]);
        __builtin_amdgcn_raw_buffer_store_b128(pack8(o), rs_d, voff[a][i], so, 0);
        store_data_settle();
      }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int i = 0; i < NCH; ++i) { ry[a][i] = ny[a][i]; rd[a][i] = nd[a][i]; }
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * wave + 256 * i;
    if (c < nch) {
      st8f(dwq_part + (long)blockIdx.x * D + c * 8, acc[0][i]);
      st8f(dwk_part + (long)blockIdx.x * D + c * 8, acc[1][i]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// decoder tail: LayerNorm -> l2 normalise [-> cosine loss row terms]
template <int NCH, int WPR = 1>
__global__ __launch_bounds__(256) void ln_l2_fwd_kernel(const bf16_t* __restrict__ y, const float* __restrict__ w,
                                                        const float* __restrict__ b, float eps, int M, int C,
                                                        bf16_t* __restrict__ out, float* __restrict__ stats,
                                                        const void* __restrict__ target, int target_bf16,
                                                        float* __restrict__ loss_rows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ float rs_xch[8], rs_xch2[16];
  int rs_par = 0, rs_par2 = 0;
  (void)rs_xch; (void)rs_par; (void)rs_xch2; (void)rs_par2;
  const int nch = C >> 3;
  for (int row = blockIdx.x * (4 / WPR) + wave / WPR; row < M; row += gridDim.x * (4 / WPR)) {
    float x[NCH][8];
    u32x4 traw[NCH];                                     // a bf16 target row is requested WITH y (round 5): loaded after the three reductions it
    float s = 0.f;                                       // cost a second HBM latency per row (192 us per 53376 x 3200 launch = 3.5 TB/s)
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * (i * WPR + wave % WPR);
      traw[i] = u32x4{0u, 0u, 0u, 0u};
      if (c < nch) {
        const u32x4 yraw = *reinterpret_cast<const u32x4*>(y + (long)row * C + c * 8);
        if (target && target_bf16) traw[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(target) + (long)row * C + c * 8);
        unpack8(yraw, x[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += x[i][e];
      }
    }
    const float mu = row_sum<WPR>(rs_xch, rs_par, s) / (float)C;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * (i * WPR + wave % WPR);
      if (c < nch) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = x[i][e] - mu; v += d * d; }
      }
    }
    const float rstd = rsqrtf(row_sum<WPR>(rs_xch, rs_par, v) / (float)C + eps);
    float n2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * (i * WPR + wave % WPR);
      if (c < nch) {
        float wv[8], bv[8];
        ld8f(w + c * 8, wv);
        ld8f(b + c * 8, bv);
#pragma unroll
        for (int e = 0; e < 8; ++e) { x[i][e] = (x[i][e] - mu) * rstd * wv[e] + bv[e]; n2 += x[i][e] * x[i][e]; }
      }
    }
    const float inv = rsqrtf(row_sum<WPR>(rs_xch, rs_par, n2));   // no epsilon, as the reference (P:359)
    if (stats && lane == 0) { stats[row * 3] = mu; stats[row * 3 + 1] = rstd; stats[row * 3 + 2] = inv; }
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * (i * WPR + wave % WPR);
      if (c < nch) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = x[i][e] * inv;
        if (out) st8b(out + (long)row * C + c * 8, o);
        if (target) {
          float t[8];
          if (target_bf16) unpack8(traw[i], t);
          else ld8f(reinterpret_cast<const float*>(target) + (long)row * C + c * 8, t);
#pragma unroll
          for (int e = 0; e < 8; ++e) dot += o[e] * t[e];
        }
      }
    }
    if (target && loss_rows) {
      dot = row_sum<WPR>(rs_xch, rs_par, dot);
      if (lane == 0) loss_rows[row] = 2.f - 2.f * dot;
    }
  }
}

// backward.  d_o = dout (fp32) or dscale * target ; d_ln = (d_o - o <o, d_o>) * inv ; LayerNorm backward.
template <int NCH, int WPR = 1>
__global__ __launch_bounds__(256) void ln_l2_bwd_kernel(const bf16_t* __restrict__ y, const float* __restrict__ w,
                                                        const float* __restrict__ b, const float* __restrict__ stats,
                                                        const void* __restrict__ dout, int dout_bf16,
                                                        const void* __restrict__ target, int target_bf16, float dscale,
                                                        const float* __restrict__ dscale_dev,
                                                        int M, int C, bf16_t* __restrict__ dy,
                                                        float* __restrict__ dw_part, float* __restrict__ db_part) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [4][C]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ float rs_xch[8], rs_xch2[16];
  int rs_par = 0, rs_par2 = 0;
  (void)rs_xch; (void)rs_par; (void)rs_xch2; (void)rs_par2;
  const int nch = C >> 3;
  if (dscale_dev) dscale *= dscale_dev[0];
  float aw[NCH][8], ab[NCH][8];
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) { aw[i][e] = 0.f; ab[i][e] = 0.f; }
  for (int row = blockIdx.x * (4 / WPR) + wave / WPR; row < M; row += gridDim.x * (4 / WPR)) {
    const float mu = stats[row * 3], rstd = stats[row * 3 + 1], inv = stats[row * 3 + 2];
    float xh[NCH][8], o[NCH][8], g[NCH][8];
    float od = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * (i * WPR + wave % WPR);
      if (c < nch) {
        float yv[8], wv[8], bv[8];
        ld8b(y + (long)row * C + c * 8, yv);
        ld8f(w + c * 8, wv);
        ld8f(b + c * 8, bv);
        if (dout) {
          if (dout_bf16) ld8b(reinterpret_cast<const bf16_t*>(dout) + (long)row * C + c * 8, g[i]);
          else ld8f(reinterpret_cast<const float*>(dout) + (long)row * C + c * 8, g[i]);
        } else {
          if (target_bf16) ld8b(reinterpret_cast<const bf16_t*>(target) + (long)row * C + c * 8, g[i]);
          else ld8f(reinterpret_cast<const float*>(target) + (long)row * C + c * 8, g[i]);
#pragma unroll
          for (int e = 0; e < 8; ++e) g[i][e] *= dscale;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          xh[i][e] = (yv[e] - mu) * rstd;
          o[i][e] = (xh[i][e] * wv[e] + bv[e]) * inv;
          od += o[i][e] * g[i][e];
        }
      }
    }
    od = row_sum<WPR>(rs_xch, rs_par, od);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * (i * WPR + wave % WPR);
      if (c < nch) {
        float wv[8];
        ld8f(w + c * 8, wv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float dln = (g[i][e] - o[i][e] * od) * inv;
          aw[i][e] += dln * xh[i][e];
          ab[i][e] += dln;
          const float dxh = dln * wv[e];
          g[i][e] = dxh;
          s1 += dxh;
          s2 += dxh * xh[i][e];
        }
      }
    }
    row_sum2<WPR>(rs_xch2, rs_par2, s1, s2);
    s1 /= (float)C; s2 /= (float)C;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * (i * WPR + wave % WPR);
      if (c < nch) {
        float r[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = rstd * (g[i][e] - s1 - xh[i][e] * s2);
        st8b(dy + (long)row * C + c * 8, r);
      }
    }
  }
  for (int pass = 0; pass < 2; ++pass) {
    float* dst = pass == 0 ? dw_part : db_part;
    __syncthreads();
    if constexpr (WPR > 1) {                                  // a wave owns a quarter of the columns: the rest of its region is zero
      for (int d = threadIdx.x; d < 4 * C; d += 256) red[d] = 0.f;
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * (i * WPR + wave % WPR);
      if (c < nch) st8f(red + wave * C + c * 8, pass == 0 ? aw[i] : ab[i]);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < C; d += 256)
      dst[(long)blockIdx.x * C + d] = red[d] + red[C + d] + red[2 * C + d] + red[3 * C + d];
  }
}

// The distillation configuration of the backward above (no upstream tensor: d_o = dscale * target, target bf16) organised like
// rmsnorm_add_bwd_b16_kernel: the four waves share a row (lane -> chunks lane + 64 w + 256 i), operands stay raw bf16, the next row is
// requested before this one is computed, every column of the two column sums belongs to one lane.  Two LDS exchanges per row (<o, d_o>,
// then the two LayerNorm sums), slot sets alternating.  53376 x 3200 (tools/bench_decoder_tail.py, every row checked against autograd):
// 316 us = 3.2 TB/s; the generic kernel (one row per workgroup and trip, no prefetch) 504 us = 2.0 TB/s.
#ifndef LNL2_PF_WAVES
#define LNL2_PF_WAVES 2
#endif
template <int NCH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NCH == 1 ? 4 : LNL2_PF_WAVES))) void ln_l2_bwd_pf_kernel(
    const bf16_t* __restrict__ y, const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ stats,
    const bf16_t* __restrict__ target, float dscale, const float* __restrict__ dscale_dev, int M, int C, bf16_t* __restrict__ dy,
    float* __restrict__ dw_part, float* __restrict__ db_part) {
  __shared__ float xa[2][4], xb[2][4][2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = C >> 3;
  const int bytes = M * C * 2;
  const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc((void*)target, 0, bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, bytes, 0x00020000);
  if (dscale_dev) dscale *= dscale_dev[0];
  unsigned voff[NCH];
  float aw[NCH][8], ab[NCH][8], wv[NCH][8], bv[NCH][8];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * wave + 256 * i;
    voff[i] = c < nch ? (unsigned)(c * 16) : 0x80000000u;
#pragma unroll
    for (int e = 0; e < 8; ++e) { aw[i][e] = 0.f; ab[i][e] = 0.f; wv[i][e] = 0.f; bv[i][e] = 0.f; }
    if (c < nch) { ld8f(w + c * 8, wv[i]); ld8f(b + c * 8, bv[i]); }
  }
  const int row_bytes = C * 2;
  const float inv_c = 1.0f / (float)C;
  u32x4 ry[NCH], rt[NCH], ny[NCH], nt[NCH];
  auto fetch = [&](int row, u32x4 (&fy)[NCH], u32x4 (&ft)[NCH]) __attribute__((always_inline)) {
    const bool ok = row < M;                                    // scalar; the scalar offset is not range-checked by the hardware
    const int so = ok ? row * row_bytes : 0;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const unsigned vo = ok ? voff[i] : 0x80000000u;
      fy[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, vo, so, 0);
      ft[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_t, vo, so, 0);
    }
  };
  int par = 0;
  fetch(blockIdx.x, ry, rt);
  for (int row = blockIdx.x; row < M; row += gridDim.x) {
    fetch(row + gridDim.x, ny, nt);
    const float mu = stats[row * 3], rstd = stats[row * 3 + 1], inv = stats[row * 3 + 2];
    float od = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      float yv[8], tv[8];
      asm volatile("" : "+v"(ry[i]), "+v"(rt[i]));
      unpack8(ry[i], yv); unpack8(rt[i], tv);
      const bool live = voff[i] != 0x80000000u;                 // a chunk past the row: y reads 0 but (0 - mu) * rstd * w + b is not 0
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (yv[e] - mu) * rstd;
        const float o = (xh * wv[i][e] + bv[i][e]) * inv;
        od += live ? o * tv[e] * dscale : 0.f;
      }
    }
    od = wave_sum(od);
    if (lane == 0) xa[par][wave] = od;
    __syncthreads();
    od = (xa[par][0] + xa[par][1]) + (xa[par][2] + xa[par][3]);
    float s1 = 0.f, s2 = 0.f;
    float dxh[NCH][8];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      float yv[8], tv[8];
      asm volatile("" : "+v"(ry[i]), "+v"(rt[i]));
      unpack8(ry[i], yv); unpack8(rt[i], tv);
      const bool live = voff[i] != 0x80000000u;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (yv[e] - mu) * rstd;
        const float o = (xh * wv[i][e] + bv[i][e]) * inv;
        const float dln = live ? (tv[e] * dscale - o * od) * inv : 0.f;
        aw[i][e] += dln * xh;
        ab[i][e] += dln;
        dxh[i][e] = dln * wv[i][e];
        s1 += dxh[i][e];
        s2 += dxh[i][e] * xh;
      }
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (lane == 0) { xb[par][wave][0] = s1; xb[par][wave][1] = s2; }
    __syncthreads();
    s1 = ((xb[par][0][0] + xb[par][1][0]) + (xb[par][2][0] + xb[par][3][0])) * inv_c;
    s2 = ((xb[par][0][1] + xb[par][1][1]) + (xb[par][2][1] + xb[par][3][1])) * inv_c;
    par ^= 1;
    const int so = row * row_bytes;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      float yv[8], r[8];
      asm volatile("" : "+v"(ry[i]));
      unpack8(ry[i], yv);
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] = rstd * (dxh[i][e] - s1 - (yv[e] - mu) * rstd * s2);
      __builtin_amdgcn_raw_buffer_store_b128(pack8(r), rs_o, voff[i], so, 0);
      store_data_settle();
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) { ry[i] = ny[i]; rt[i] = nt[i]; }
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * wave + 256 * i;
    if (c < nch) {
      st8f(dw_part + (long)blockIdx.x * C + c * 8, aw[i]);
      st8f(db_part + (long)blockIdx.x * C + c * 8, ab[i]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// LayerNorm with one or two affine heads sharing the statistics (attention-pooling projector: norm1_k / norm1_v
// read the same tokens, P:99-101).  x fp32 or bf16; y, y2 bf16; stats [M][2] = (mean, rstd).
template <int NCH, typename TX, int WPR = 1>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const TX* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                            const float* __restrict__ w2, const float* __restrict__ b2, float eps, int M, int C,
                                                            bf16_t* __restrict__ y, bf16_t* __restrict__ y2, float* __restrict__ stats) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ float rs_xch[8], rs_xch2[16];
  int rs_par = 0, rs_par2 = 0;
  (void)rs_xch; (void)rs_par; (void)rs_xch2; (void)rs_par2;
  const int nch = C >> 3;
  for (int row = blockIdx.x * (4 / WPR) + wave / WPR; row < M; row += gridDim.x * (4 / WPR)) {
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * (i * WPR + wave % WPR);
      if (c < nch) {
        if constexpr (sizeof(TX) == 4) ld8f(reinterpret_cast<const float*>(x) + (long)row * C + c * 8, v[i]);
        else ld8b(reinterpret_cast<const bf16_t*>(x) + (long)row * C + c * 8, v[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[i][e];
      }
    }
    const float mu = row_sum<WPR>(rs_xch, rs_par, s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * (i * WPR + wave % WPR);
      if (c < nch) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mu; q += d * d; }
      }
    }
    const float rstd = rsqrtf(row_sum<WPR>(rs_xch, rs_par, q) / (float)C + eps);
    if (stats && lane == 0) { stats[row * 2] = mu; stats[row * 2 + 1] = rstd; }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * (i * WPR + wave % WPR);
      if (c < nch) {
        float wv[8], bv[8], o[8];
        ld8f(w + c * 8, wv); ld8f(b + c * 8, bv);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mu) * rstd * wv[e] + bv[e];
        st8b(y + (long)row * C + c * 8, o);
        if (y2) {
          ld8f(w2 + c * 8, wv); ld8f(b2 + c * 8, bv);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mu) * rstd * wv[e] + bv[e];
          st8b(y2 + (long)row * C + c * 8, o);
        }
      }
    }
  }
}

// dx (fp32, = or +=) from dy (bf16) [and dy2]; partial column sums of dw, db [, dw2, db2]
template <int NCH, typename TX, int WPR = 1>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const TX* __restrict__ x, const float* __restrict__ w, const float* __restrict__ w2,
                                                            const float* __restrict__ stats, const bf16_t* __restrict__ dy,
                                                            const bf16_t* __restrict__ dy2, int M, int C, float* __restrict__ dx, int accumulate,
                                                            float* __restrict__ dw_part, float* __restrict__ db_part,
                                                            float* __restrict__ dw2_part, float* __restrict__ db2_part) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [4][C]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ float rs_xch[8], rs_xch2[16];
  int rs_par = 0, rs_par2 = 0;
  (void)rs_xch; (void)rs_par; (void)rs_xch2; (void)rs_par2;
  const int nch = C >> 3;
  float aw[NCH][8], ab[NCH][8], aw2[NCH][8], ab2[NCH][8];
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) { aw[i][e] = 0.f; ab[i][e] = 0.f; aw2[i][e] = 0.f; ab2[i][e] = 0.f; }
  for (int row = blockIdx.x * (4 / WPR) + wave / WPR; row < M; row += gridDim.x * (4 / WPR)) {
    const float mu = stats[row * 2], rstd = stats[row * 2 + 1];
    float xh[NCH][8], g[NCH][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * (i * WPR + wave % WPR);
      if (c < nch) {
        float xv[8], d1[8], wv[8];
        if constexpr (sizeof(TX) == 4) ld8f(reinterpret_cast<const float*>(x) + (long)row * C + c * 8, xv);
        else ld8b(reinterpret_cast<const bf16_t*>(x) + (long)row * C + c * 8, xv);
        ld8b(dy + (long)row * C + c * 8, d1);
        ld8f(w + c * 8, wv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          xh[i][e] = (xv[e] - mu) * rstd;
          g[i][e] = d1[e] * wv[e];
          aw[i][e] += d1[e] * xh[i][e];
          ab[i][e] += d1[e];
        }
        if (dy2) {
          ld8b(dy2 + (long)row * C + c * 8, d1);
          ld8f(w2 + c * 8, wv);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            g[i][e] += d1[e] * wv[e];
            aw2[i][e] += d1[e] * xh[i][e];
            ab2[i][e] += d1[e];
          }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1 += g[i][e]; s2 += g[i][e] * xh[i][e]; }
      }
    }
    row_sum2<WPR>(rs_xch2, rs_par2, s1, s2);
    s1 /= (float)C; s2 /= (float)C;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * (i * WPR + wave % WPR);
      if (c < nch) {
        float r[8];
        float* dp = dx + (long)row * C + c * 8;
        if (accumulate) ld8f(dp, r);
        else {
#pragma unroll
          for (int e = 0; e < 8; ++e) r[e] = 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] += rstd * (g[i][e] - s1 - xh[i][e] * s2);
        st8f(dp, r);
      }
    }
  }
  for (int pass = 0; pass < 4; ++pass) {
    float* dst = pass == 0 ? dw_part : pass == 1 ? db_part : pass == 2 ? dw2_part : db2_part;
    if (!dst) continue;
    __syncthreads();
    if constexpr (WPR > 1) {                                  // a wave owns a quarter of the columns: the rest of its region is zero
      for (int d = threadIdx.x; d < 4 * C; d += 256) red[d] = 0.f;
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * (i * WPR + wave % WPR);
      if (c < nch) st8f(red + wave * C + c * 8, pass == 0 ? aw[i] : pass == 1 ? ab[i] : pass == 2 ? aw2[i] : ab2[i]);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < C; d += 256)
      dst[(long)blockIdx.x * C + d] = red[d] + red[C + d] + red[2 * C + d] + red[3 * C + d];
  }
}

// mean over the L tokens of each clip: x fp32 [B][L][D] -> out fp32 [B][D]
__global__ __launch_bounds__(256) void token_mean_fwd_kernel(const float* __restrict__ x, int B, int L, int D, float* __restrict__ out) {
  const int nch = D >> 3;
  const long id = (long)blockIdx.x * 256 + threadIdx.x;
  if (id >= (long)B * nch) return;
  const int c = id % nch, b = id / nch;
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int l = 0;
  for (; l + 3 < L; l += 4) {                          // four rows in flight, added in row order (same bits as one at a time)
    float v[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u) ld8f(x + ((long)b * L + l + u) * D + c * 8, v[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += v[u][e];
  }
  for (; l < L; ++l) {
    float v[8];
    ld8f(x + ((long)b * L + l) * D + c * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] += v[e];
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) a[e] /= (float)L;
  st8f(out + (long)b * D + c * 8, a);
}
// dx[b][l][:] += dmean[b][:] / L
__global__ __launch_bounds__(256) void token_mean_bwd_kernel(const float* __restrict__ dmean, int B, int L, int D, float* __restrict__ dx) {
  const int nch = D >> 3;
  const long id = (long)blockIdx.x * 256 + threadIdx.x;
  if (id >= (long)B * L * nch) return;
  const int c = id % nch;
  const long bl = id / nch;
  const int b = bl / L;
  float g[8], v[8];
  ld8f(dmean + (long)b * D + c * 8, g);
  ld8f(dx + bl * D + c * 8, v);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] += g[e] / (float)L;
  st8f(dx + bl * D + c * 8, v);
}

// ---------------------------------------------------------------------------------------------------------
// column reductions
// A workgroup owns 16 columns; 16 lanes per column walk the partials 16 apart with eight loads in flight each.  (The first version gave a
// workgroup 64 columns and 4 lanes per column: 22 workgroups for D = 1408 and a chain of 32 dependent trips -- 12 us of latency per
// launch, 130 launches per step.)  The order of the sum is fixed: results do not depend on scheduling.
__device__ __forceinline__ void colsum_finish_body(const float* __restrict__ part, int n_part, int D, float* __restrict__ out, int accumulate,
                                                   int col_block, float (*red)[16]) {
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int d = col_block * 16 + tx;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (d < D) {
    int p = ty;
    for (; p + 112 < n_part; p += 128) {
#pragma unroll
      for (int u = 0; u < 8; ++u) s[u] += part[(long)(p + 16 * u) * D + d];
    }
    for (; p < n_part; p += 16) s[0] += part[(long)p * D + d];
  }
  red[ty][tx] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  if (ty == 0 && d < D) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) t += red[j][tx];
    out[d] = accumulate ? out[d] + t : t;
  }
}
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ part, int n_part, int D,
                                                            float* __restrict__ out, int accumulate) {
  __shared__ float red[16][16];
  colsum_finish_body(part, n_part, D, out, accumulate, blockIdx.x, red);
}
// the partial rows were produced per `rows_per_part` rows of a matrix whose row count lives in device memory (the 256^2 dgrad epilogue under
// ivh_gemm_desc.m_dev: two partial rows per started 256-row tile): only the first parts_per_unit * ceil(*m_dev / rows_per_unit) are valid
__global__ __launch_bounds__(256) void colsum_finish_dyn_kernel(const float* __restrict__ part, int n_part, int D, float* __restrict__ out, int accumulate,
                                                                const int* __restrict__ m_dev, int rows_per_unit, int parts_per_unit) {
  __shared__ float red[16][16];
  const int m = max(0, *m_dev);
  const int valid = parts_per_unit * ((m + rows_per_unit - 1) / rows_per_unit);
  colsum_finish_body(part, min(n_part, valid), D, out, accumulate, blockIdx.x, red);
}

// DropPath keep maps (ivh_droppath_plan): one wave per (block, branch) set walks its B per-sample scales in chunks of 64 -- a sample is kept
// when its scale is non-zero; slot = number of kept samples in front of it (ballot + popcount: exact, order preserving)
__global__ __launch_bounds__(64) void droppath_plan_kernel(const float* __restrict__ rowscale, int B, int rows_per_sample,
                                                           int* __restrict__ slot, int* __restrict__ count) {
  const int set = blockIdx.x, lane = threadIdx.x;
  const float* rs = rowscale + (long)set * B;
  int* sl = slot + (long)set * B;
  int base = 0;
  for (int b0 = 0; b0 < B; b0 += 64) {
    const int b = b0 + lane;
    const bool keep = b < B && rs[b] != 0.0f;
    const unsigned long long mask = __ballot(keep);
    const int before = __popcll(mask & ((1ull << lane) - 1ull));
    if (b < B) sl[b] = keep ? base + before : -1;
    base += __popcll(mask);
  }
  if (lane == 0) { count[set * 2] = base; count[set * 2 + 1] = base * rows_per_sample; }
}

// x [M][N] bf16 -> part[blockIdx.y][N];  block = 64 chunk-columns x 4 row lanes
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* __restrict__ x, long ld, int M, int N,
                                                          float* __restrict__ part) {
  __shared__ float red[4][64 * 8];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;   // 8-column chunk
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c * 8 < N) {
    for (int r = blockIdx.y * 4 + ty; r < M; r += gridDim.y * 4) {
      float v[8];
      ld8b(x + (long)r * ld + c * 8, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[ty][tx * 8 + e] = a[e];
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 256) {
    const int n = blockIdx.x * 512 + i;
    if (n < N) part[(long)blockIdx.y * N + n] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
  }
}

// out[0] = scale * sum x[0..n)   (single block, fixed order: deterministic)
__global__ __launch_bounds__(256) void sum_rows_kernel(const float* __restrict__ x, int n, float scale, float* __restrict__ out) {
  __shared__ float red[256];
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};    // eight loads in flight per thread (one at a time: 208 dependent trips = 54 us for 53376 rows)
  int i = threadIdx.x;
  for (; i + 7 * 256 < n; i += 8 * 256) {
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] += x[i + u * 256];
  }
  for (; i < n; i += 256) a[0] += x[i];
  red[threadIdx.x] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0] * scale;
}

static inline int nch_for(int D) { return (D / 8 + 63) / 64; }
static inline int row_grid(int M, int cap) { int g = (M + 3) / 4; return g < cap ? (g < 1 ? 1 : g) : cap; }
// workgroups (= partial-sum rows) of the backward row kernels.  IVH_BWD_PARTS overrides it for occupancy experiments (tools/bench_rows.py)
static int bwd_parts_cap() {
  static const int v = [] { const char* e = getenv("IVH_BWD_PARTS"); const int n = e ? atoi(e) : 0; return n >= 64 && n <= 8192 ? n : 512; }();
  return v;
}
#define BWD_PARTS_CAP bwd_parts_cap()
// rows per workgroup and trip of the bf16-stream backward (rmsnorm_add_bwd_b16_kernel): 1 by default -- measured on M = 53376, D = 1408
// (profiles/r3_rows_bf16_stream_sweep.jsonl): generic kernel 247 us, 1 row 174 us, 2 rows 182 us, 4 rows 188 us at 512 workgroups; more
// workgroups are slower for every shape.  IVH_BWD_ROWS = 1 / 2 / 4 selects, 0 = the generic kernel
static int bwd_rows() {
  static const int v = [] { const char* e = getenv("IVH_BWD_ROWS"); const int n = e ? atoi(e) : 1; return n >= 0 && n <= 4 ? n : 1; }();
  return v;
}

}  // namespace ivh

#define IVH_DISPATCH_NCH(nch, KERNEL, grid, block, shmem, s, ...)                                   \
  switch (nch) {                                                                                    \
    case 1: hipLaunchKernelGGL((KERNEL<1>), grid, block, shmem, s, __VA_ARGS__); break;             \
    case 2: hipLaunchKernelGGL((KERNEL<2>), grid, block, shmem, s, __VA_ARGS__); break;             \
    case 3: hipLaunchKernelGGL((KERNEL<3>), grid, block, shmem, s, __VA_ARGS__); break;             \
    case 4: hipLaunchKernelGGL((KERNEL<4>), grid, block, shmem, s, __VA_ARGS__); break;             \
    case 5: case 6: case 7: case 8: hipLaunchKernelGGL((KERNEL<2, 4>), grid, block, shmem, s, __VA_ARGS__); break; \
    default: ivh_host::set_error("row width %d not supported (max 4096)", (nch) * 512); return -1;  \
  }

#define IVH_DISPATCH_NCH_R(nch, KERNEL, TR, grid, block, shmem, s, ...)                             \
  switch (nch) {                                                                                    \
    case 1: hipLaunchKernelGGL((KERNEL<1, 1, TR>), grid, block, shmem, s, __VA_ARGS__); break;      \
    case 2: hipLaunchKernelGGL((KERNEL<2, 1, TR>), grid, block, shmem, s, __VA_ARGS__); break;      \
    case 3: hipLaunchKernelGGL((KERNEL<3, 1, TR>), grid, block, shmem, s, __VA_ARGS__); break;      \
    case 4: hipLaunchKernelGGL((KERNEL<4, 1, TR>), grid, block, shmem, s, __VA_ARGS__); break;      \
    case 5: case 6: case 7: case 8: hipLaunchKernelGGL((KERNEL<2, 4, TR>), grid, block, shmem, s, __VA_ARGS__); break; \
    default: ivh_host::set_error("row width %d not supported (max 4096)", (nch) * 512); return -1;  \
  }

using namespace ivh;

extern "C" int ivh_rmsnorm_add_fwd(const float* res_in, const uint16_t* branch, const float* gamma, const float* rowscale,
                                   int rows_per_sample, const float* w, float eps, int M, int D,
                                   float* res_out, uint16_t* y, float* rstd, void* stream) {
  IVH_REQUIRE(M > 0 && D > 0 && D % 8 == 0, "rmsnorm_add_fwd: bad shape M=%d D=%d", M, D);
  IVH_REQUIRE(res_in || branch, "rmsnorm_add_fwd: need res_in or branch");
  IVH_REQUIRE(!y || w, "rmsnorm_add_fwd: y requested without weight");
  IVH_REQUIRE(!rowscale || rows_per_sample > 0, "rmsnorm_add_fwd: rows_per_sample must be > 0");
  const int nch = nch_for(D);
  IVH_DISPATCH_NCH(nch, rmsnorm_add_fwd_kernel, dim3(row_grid(M, 8192)), dim3(256), 0, (hipStream_t)stream,
                   res_in, branch, gamma, rowscale, rows_per_sample, w, eps, M, D, res_out, y, rstd);
  return ivh_host::check_launch("rmsnorm_add_fwd");
}

extern "C" int ivh_rmsnorm_add_fwd_bf16res(const uint16_t* res_in, const uint16_t* branch, const float* gamma, const float* rowscale,
                                           int rows_per_sample, const float* w, float eps, int M, int D,
                                           uint16_t* res_out, uint16_t* y, float* rstd, void* stream) {
  IVH_REQUIRE(M > 0 && D > 0 && D % 8 == 0, "rmsnorm_add_fwd_bf16res: bad shape M=%d D=%d", M, D);
  IVH_REQUIRE(res_in || branch, "rmsnorm_add_fwd_bf16res: need res_in or branch");
  IVH_REQUIRE(!y || w, "rmsnorm_add_fwd_bf16res: y requested without weight");
  IVH_REQUIRE(!rowscale || rows_per_sample > 0, "rmsnorm_add_fwd_bf16res: rows_per_sample must be > 0");
  const int nch = nch_for(D);
  IVH_DISPATCH_NCH_R(nch, rmsnorm_add_fwd_kernel, bf16_t, dim3(row_grid(M, 8192)), dim3(256), 0, (hipStream_t)stream,
                     res_in, branch, gamma, rowscale, rows_per_sample, w, eps, M, D, res_out, y, rstd);
  return ivh_host::check_launch("rmsnorm_add_fwd_bf16res");
}

// ---- DropPath sample skipping (ABI 2) ------------------------------------------------------------------------------------------------
extern "C" int ivh_droppath_plan(const float* rowscale, int n_sets, int B, int rows_per_sample, int32_t* slot, int32_t* count, void* stream) {
  IVH_REQUIRE(rowscale && slot && count && n_sets > 0 && B > 0 && rows_per_sample > 0, "droppath_plan: bad args");
  hipLaunchKernelGGL(droppath_plan_kernel, dim3(n_sets), dim3(64), 0, (hipStream_t)stream, rowscale, B, rows_per_sample, slot, count);
  return ivh_host::check_launch("droppath_plan");
}

#define IVH_DISPATCH_NCH_RS(nch, KERNEL, TR, grid, block, shmem, s, ...)                            \
  switch (nch) {                                                                                    \
    case 1: hipLaunchKernelGGL((KERNEL<1, 1, TR, true>), grid, block, shmem, s, __VA_ARGS__); break;      \
    case 2: hipLaunchKernelGGL((KERNEL<2, 1, TR, true>), grid, block, shmem, s, __VA_ARGS__); break;      \
    case 3: hipLaunchKernelGGL((KERNEL<3, 1, TR, true>), grid, block, shmem, s, __VA_ARGS__); break;      \
    case 4: hipLaunchKernelGGL((KERNEL<4, 1, TR, true>), grid, block, shmem, s, __VA_ARGS__); break;      \
    case 5: case 6: case 7: case 8: hipLaunchKernelGGL((KERNEL<2, 4, TR, true>), grid, block, shmem, s, __VA_ARGS__); break; \
    default: ivh_host::set_error("row width %d not supported (max 4096)", (nch) * 512); return -1;  \
  }

extern "C" int ivh_rmsnorm_add_fwd_skip(const void* res_in, int res_bf16, const uint16_t* branch, const float* gamma, const float* rowscale,
                                        int rows_per_sample, const float* w, float eps, int M, int D,
                                        void* res_out, uint16_t* y, float* rstd, const int32_t* branch_slot, const int32_t* y_slot, void* stream) {
  IVH_REQUIRE(M > 0 && D > 0 && D % 8 == 0, "rmsnorm_add_fwd_skip: bad shape M=%d D=%d", M, D);
  IVH_REQUIRE(res_in || branch, "rmsnorm_add_fwd_skip: need res_in or branch");
  IVH_REQUIRE(!y || w, "rmsnorm_add_fwd_skip: y requested without weight");
  IVH_REQUIRE(rows_per_sample > 0 && M % rows_per_sample == 0, "rmsnorm_add_fwd_skip: M=%d must be whole samples of %d rows", M, rows_per_sample);
  IVH_REQUIRE(res_in || !branch_slot, "rmsnorm_add_fwd_skip: a compacted branch needs the stream it is added to");
  const int nch = nch_for(D);
  if (res_bf16) {
    IVH_DISPATCH_NCH_RS(nch, rmsnorm_add_fwd_kernel, bf16_t, dim3(row_grid(M, 8192)), dim3(256), 0, (hipStream_t)stream,
                        (const bf16_t*)res_in, branch, gamma, rowscale, rows_per_sample, w, eps, M, D, (bf16_t*)res_out, y, rstd, branch_slot, y_slot);
  } else {
    IVH_DISPATCH_NCH_RS(nch, rmsnorm_add_fwd_kernel, float, dim3(row_grid(M, 8192)), dim3(256), 0, (hipStream_t)stream,
                        (const float*)res_in, branch, gamma, rowscale, rows_per_sample, w, eps, M, D, (float*)res_out, y, rstd, branch_slot, y_slot);
  }
  return ivh_host::check_launch("rmsnorm_add_fwd_skip");
}

extern "C" int ivh_norm_bwd_parts(int M) { return row_grid(M, BWD_PARTS_CAP); }

extern "C" int ivh_rmsnorm_add_bwd(const uint16_t* dy, const float* dres_out, const float* res_out, const float* rstd,
                                   const float* w, const uint16_t* branch, const float* gamma, const float* rowscale,
                                   int rows_per_sample, int M, int D, float* dres_in, uint16_t* dbranch,
                                   float* dw_part, float* dgamma_part, float* dbias_part, void* stream) {
  IVH_REQUIRE(M > 0 && D > 0 && D % 8 == 0, "rmsnorm_add_bwd: bad shape M=%d D=%d", M, D);
  IVH_REQUIRE(!dbias_part || dbranch, "rmsnorm_add_bwd: dbias_part is the column sum of dbranch");
  IVH_REQUIRE(dy || dres_out, "rmsnorm_add_bwd: need dy or dres_out");
  IVH_REQUIRE(!dy || (res_out && rstd && w && dw_part), "rmsnorm_add_bwd: dy needs res_out, rstd, w, dw_part");
  const int nch = nch_for(D);
  const int grid = row_grid(M, BWD_PARTS_CAP);
  const size_t sh = (size_t)4 * D * sizeof(float);
  IVH_DISPATCH_NCH(nch, rmsnorm_add_bwd_kernel, dim3(grid), dim3(256), sh, (hipStream_t)stream,
                   dy, dres_out, res_out, rstd, w, branch, gamma, rowscale, rows_per_sample, M, D,
                   dres_in, dbranch, dy ? dw_part : nullptr, dgamma_part, dbias_part);
  return ivh_host::check_launch("rmsnorm_add_bwd");
}

extern "C" int ivh_rmsnorm_add_bwd_bf16res(const uint16_t* dy, const uint16_t* dres_out, const uint16_t* res_out, const float* rstd,
                                           const float* w, const uint16_t* branch, const float* gamma, const float* rowscale,
                                           int rows_per_sample, int M, int D, uint16_t* dres_in, uint16_t* dbranch,
                                           float* dw_part, float* dgamma_part, float* dbias_part, const uint16_t* dres_extra, void* stream) {
  IVH_REQUIRE(M > 0 && D > 0 && D % 8 == 0, "rmsnorm_add_bwd_bf16res: bad shape M=%d D=%d", M, D);
  IVH_REQUIRE(!dres_extra || dres_out, "rmsnorm_add_bwd_bf16res: dres_extra joins dres_out (pass it as dres_out when it is the only gradient)");
  IVH_REQUIRE(!dbias_part || dbranch, "rmsnorm_add_bwd_bf16res: dbias_part is the column sum of dbranch");
  IVH_REQUIRE(dy || dres_out, "rmsnorm_add_bwd_bf16res: need dy or dres_out");
  IVH_REQUIRE(!dy || (res_out && rstd && w && dw_part), "rmsnorm_add_bwd_bf16res: dy needs res_out, rstd, w, dw_part");
  const int nch = nch_for(D);
  const int grid = row_grid(M, BWD_PARTS_CAP);
  const size_t sh = (size_t)4 * D * sizeof(float);
  float* dwp = dy ? dw_part : nullptr;
  const int rows = bwd_rows();
#define IVH_B16_BWD(N, R) hipLaunchKernelGGL((rmsnorm_add_bwd_b16_kernel<N, R>), dim3(grid), dim3(256), 0, (hipStream_t)stream, dy, dres_out, \
    res_out, rstd, w, branch, gamma, rowscale, rows_per_sample, M, D, dres_in, dbranch, dwp, dgamma_part, dbias_part)
  const bool interior = dy && dres_out && res_out && branch && gamma && dres_in && dbranch && dw_part && dgamma_part && dbias_part;
  if (interior && nch <= 8 && rows > 0 && (long)M * D * 2 < (1L << 31)) {     // a block's interior: the bytes-in-flight kernel
    const int n4 = (D / 8 + 255) / 256;                    // 16-byte chunks per lane when four waves share a row: 1 up to D = 2048, else 2
    if (dres_extra) {                                      // a tapped block (10 of the 1B step's 79 launches): one row per trip
      if (n4 == 1) hipLaunchKernelGGL((rmsnorm_add_bwd_b16_kernel<1, 1, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, dy, dres_out, res_out, rstd, w,
                                      branch, gamma, rowscale, rows_per_sample, M, D, dres_in, dbranch, dwp, dgamma_part, dbias_part, dres_extra);
      else hipLaunchKernelGGL((rmsnorm_add_bwd_b16_kernel<2, 1, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, dy, dres_out, res_out, rstd, w,
                              branch, gamma, rowscale, rows_per_sample, M, D, dres_in, dbranch, dwp, dgamma_part, dbias_part, dres_extra);
      return ivh_host::check_launch("rmsnorm_add_bwd_bf16res");
    }
    if (n4 == 1) { if (rows >= 4) IVH_B16_BWD(1, 4); else if (rows >= 2) IVH_B16_BWD(1, 2); else IVH_B16_BWD(1, 1); }
    else { if (rows >= 2) IVH_B16_BWD(2, 2); else IVH_B16_BWD(2, 1); }
    return ivh_host::check_launch("rmsnorm_add_bwd_bf16res");
  }
#undef IVH_B16_BWD
  IVH_DISPATCH_NCH_R(nch, rmsnorm_add_bwd_kernel, bf16_t, dim3(grid), dim3(256), sh, (hipStream_t)stream,
                     dy, dres_out, res_out, rstd, w, branch, gamma, rowscale, rows_per_sample, M, D,
                     dres_in, dbranch, dwp, dgamma_part, dbias_part, dres_extra);
  return ivh_host::check_launch("rmsnorm_add_bwd_bf16res");
}

extern "C" int ivh_rmsnorm_add_bwd_skip(const uint16_t* dy, const void* dres_out, int res_bf16, const void* res_out, const float* rstd,
                                        const float* w, const uint16_t* branch, const float* gamma, const float* rowscale,
                                        int rows_per_sample, int M, int D, void* dres_in, uint16_t* dbranch,
                                        float* dw_part, float* dgamma_part, float* dbias_part, const void* dres_extra,
                                        const int32_t* y_slot, const int32_t* branch_slot, void* stream) {
  IVH_REQUIRE(M > 0 && D > 0 && D % 8 == 0, "rmsnorm_add_bwd_skip: bad shape M=%d D=%d", M, D);
  IVH_REQUIRE(rows_per_sample > 0 && M % rows_per_sample == 0, "rmsnorm_add_bwd_skip: M=%d must be whole samples of %d rows", M, rows_per_sample);
  IVH_REQUIRE(!dres_extra || (dres_out && res_bf16), "rmsnorm_add_bwd_skip: dres_extra joins a bf16 dres_out");
  IVH_REQUIRE(!dbias_part || dbranch, "rmsnorm_add_bwd_skip: dbias_part is the column sum of dbranch");
  IVH_REQUIRE(dy || dres_out, "rmsnorm_add_bwd_skip: need dy or dres_out");
  IVH_REQUIRE(!dy || (res_out && rstd && w && dw_part), "rmsnorm_add_bwd_skip: dy needs res_out, rstd, w, dw_part");
  const int nch = nch_for(D);
  const int grid = row_grid(M, BWD_PARTS_CAP);
  const size_t sh = (size_t)4 * D * sizeof(float);
  float* dwp = dy ? dw_part : nullptr;
  hipStream_t s = (hipStream_t)stream;
  if (!res_bf16) {
    IVH_DISPATCH_NCH_RS(nch, rmsnorm_add_bwd_kernel, float, dim3(grid), dim3(256), sh, s, dy, (const float*)dres_out, (const float*)res_out, rstd, w, branch,
                        gamma, rowscale, rows_per_sample, M, D, (float*)dres_in, dbranch, dwp, dgamma_part, dbias_part, (const float*)nullptr, y_slot, branch_slot);
    return ivh_host::check_launch("rmsnorm_add_bwd_skip");
  }
  const bf16_t* dro = (const bf16_t*)dres_out; const bf16_t* ro = (const bf16_t*)res_out; const bf16_t* ex = (const bf16_t*)dres_extra;
  bf16_t* dri = (bf16_t*)dres_in;
  const bool interior = dy && dres_out && res_out && branch && gamma && dres_in && dbranch && dw_part && dgamma_part && dbias_part;
  if (interior && nch <= 8 && bwd_rows() > 0 && (long)M * D * 2 < (1L << 31)) {     // a block's interior: the bytes-in-flight kernel, one row per trip
    const int n4 = (D / 8 + 255) / 256;
#define IVH_B16_SKIP(N, EX) hipLaunchKernelGGL((rmsnorm_add_bwd_b16_kernel<N, 1, EX, true>), dim3(grid), dim3(256), 0, s, dy, dro, ro, rstd, w, branch, gamma, \
    rowscale, rows_per_sample, M, D, dri, dbranch, dwp, dgamma_part, dbias_part, ex, y_slot, branch_slot)
    if (n4 == 1) { if (ex) IVH_B16_SKIP(1, true); else IVH_B16_SKIP(1, false); }
    else { if (ex) IVH_B16_SKIP(2, true); else IVH_B16_SKIP(2, false); }
#undef IVH_B16_SKIP
    return ivh_host::check_launch("rmsnorm_add_bwd_skip");
  }
  IVH_DISPATCH_NCH_RS(nch, rmsnorm_add_bwd_kernel, bf16_t, dim3(grid), dim3(256), sh, s, dy, dro, ro, rstd, w, branch, gamma, rowscale, rows_per_sample, M, D,
                      dri, dbranch, dwp, dgamma_part, dbias_part, ex, y_slot, branch_slot);
  return ivh_host::check_launch("rmsnorm_add_bwd_skip");
}

// up to 4 column reductions of the same shape in one launch (blockIdx.y picks the array): the dw / dgamma (/ db) partials that one
// norm-backward kernel leaves behind
struct ColsumMulti { const float* part[4]; float* out[4]; };
__global__ __launch_bounds__(256) void colsum_finish_multi_kernel(ColsumMulti a, int n_part, int D, int accumulate) {
  __shared__ float red[16][16];
  colsum_finish_body(a.part[blockIdx.y], n_part, D, a.out[blockIdx.y], accumulate, blockIdx.x, red);
}

extern "C" int ivh_colsum_finish_multi(const float* const* parts, float* const* outs, int n, int n_part, int D, int accumulate, void* stream) {
  IVH_REQUIRE(parts && outs && n >= 1 && n <= 4 && n_part > 0 && D > 0, "colsum_finish_multi: bad args");
  ColsumMulti a{};
  for (int i = 0; i < n; ++i) {
    IVH_REQUIRE(parts[i] && outs[i], "colsum_finish_multi: null array %d", i);
    a.part[i] = parts[i]; a.out[i] = outs[i];
  }
  hipLaunchKernelGGL(colsum_finish_multi_kernel, dim3((D + 15) / 16, n), dim3(256), 0, (hipStream_t)stream, a, n_part, D, accumulate);
  return ivh_host::check_launch("colsum_finish_multi");
}

extern "C" int ivh_colsum_finish(const float* part, int n_part, int D, float* out, int accumulate, void* stream) {
  IVH_REQUIRE(part && out && n_part > 0 && D > 0, "colsum_finish: bad args");
  hipLaunchKernelGGL(colsum_finish_kernel, dim3((D + 15) / 16), dim3(256), 0, (hipStream_t)stream, part, n_part, D, out, accumulate);
  return ivh_host::check_launch("colsum_finish");
}

extern "C" int ivh_colsum_finish_dyn(const float* part, int n_part, int D, float* out, int accumulate, const int32_t* m_dev, int rows_per_unit,
                                     int parts_per_unit, void* stream) {
  IVH_REQUIRE(part && out && m_dev && n_part > 0 && D > 0 && rows_per_unit > 0 && parts_per_unit > 0, "colsum_finish_dyn: bad args");
  hipLaunchKernelGGL(colsum_finish_dyn_kernel, dim3((D + 15) / 16), dim3(256), 0, (hipStream_t)stream, part, n_part, D, out, accumulate, m_dev, rows_per_unit,
                     parts_per_unit);
  return ivh_host::check_launch("colsum_finish_dyn");
}

static inline int colsum_rb(int M) { int rb = (M + 63) / 64; return rb > 128 ? 128 : (rb < 1 ? 1 : rb); }
extern "C" int ivh_colsum_scratch_floats(int M, int N) { return colsum_rb(M) * N; }
extern "C" int ivh_colsum_bf16(const uint16_t* x, int64_t ld, int M, int N, float* out, float* scratch, void* stream) {
  IVH_REQUIRE(x && out && scratch && M > 0 && N > 0 && N % 8 == 0 && ld % 8 == 0, "colsum_bf16: bad args M=%d N=%d", M, N);
  const int rb = colsum_rb(M);
  hipLaunchKernelGGL(colsum_bf16_kernel, dim3((N + 511) / 512, rb), dim3(256), 0, (hipStream_t)stream, x, (long)ld, M, N, scratch);
  hipLaunchKernelGGL(colsum_finish_kernel, dim3((N + 15) / 16), dim3(256), 0, (hipStream_t)stream, scratch, rb, N, out, 0);
  return ivh_host::check_launch("colsum_bf16");
}

extern "C" int ivh_qk_rmsnorm_fwd_dyn(uint16_t* qkv, const float* wq, const float* wk, float eps, int M, int D,
                                      float* rstd_q, float* rstd_k, const int32_t* m_dev, void* stream) {
  IVH_REQUIRE(qkv && wq && wk && rstd_q && rstd_k && M > 0 && D % 8 == 0, "qk_rmsnorm_fwd: bad args");
  const int nch = nch_for(D);
  IVH_DISPATCH_NCH(nch, qk_rmsnorm_fwd_kernel, dim3(row_grid(M, 8192)), dim3(256), 0, (hipStream_t)stream,
                   qkv, wq, wk, eps, M, D, rstd_q, rstd_k, m_dev);
  return ivh_host::check_launch("qk_rmsnorm_fwd");
}
extern "C" int ivh_qk_rmsnorm_fwd(uint16_t* qkv, const float* wq, const float* wk, float eps, int M, int D,
                                  float* rstd_q, float* rstd_k, void* stream) {
  return ivh_qk_rmsnorm_fwd_dyn(qkv, wq, wk, eps, M, D, rstd_q, rstd_k, nullptr, stream);
}

// The q/k-norm backward's own workgroup count (= rows of its partial-sum arrays).  The bytes-in-flight kernel (129 VGPRs at D <= 2048: three
// waves per SIMD) wants exactly three resident workgroups per CU: 768 (profiles/r5_qk_rmsnorm_bwd_b16_grid_sweep_v1.txt: 256 / 384 / 512 / 768 /
// 1024 / 2048 workgroups -> 304 / 242 / 209 / 188 / 216 / 200 us per call at M = 53376); the generic kernel keeps the shared cap.
static int qk_bwd_b16() { static const int v = [] { const char* e = getenv("IVH_QKBWD_B16"); return e ? atoi(e) : 1; }(); return v; }   // 0: generic kernel (A/B)
static int qk_bwd_uses_b16(int M, int D) { return qk_bwd_b16() > 0 && (D / 8 + 255) / 256 <= 2 && (long)M * D * 6 < (1L << 31); }
extern "C" int ivh_qk_norm_bwd_parts(int M, int D) {
  static const int cap = [] { const char* e = getenv("IVH_QKBWD_PARTS"); const int n = e ? atoi(e) : 0; return n >= 64 && n <= 8192 ? n : 768; }();
  if (qk_bwd_uses_b16(M, D) && (D / 8 + 255) / 256 == 1) return M < cap ? (M < 1 ? 1 : M) : cap;
  return row_grid(M, BWD_PARTS_CAP);
}

extern "C" int ivh_qk_rmsnorm_bwd_dyn(const uint16_t* qkv, uint16_t* dqkv, const float* wq, const float* wk,
                                      const float* rstd_q, const float* rstd_k, int M, int D,
                                      float* dwq_part, float* dwk_part, const int32_t* m_dev, void* stream) {
  IVH_REQUIRE(qkv && dqkv && wq && wk && rstd_q && rstd_k && dwq_part && dwk_part && M > 0 && D % 8 == 0, "qk_rmsnorm_bwd: bad args");
  const int nch = nch_for(D);
  const int grid = ivh_qk_norm_bwd_parts(M, D);
  const int n4 = (D / 8 + 255) / 256;                        // 16-byte chunks per lane and segment when four waves share a token
  if (qk_bwd_uses_b16(M, D)) {
    if (n4 == 1) hipLaunchKernelGGL((qk_rmsnorm_bwd_b16_kernel<1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, qkv, dqkv, wq, wk, rstd_q, rstd_k, M, D, dwq_part, dwk_part, m_dev);
    else hipLaunchKernelGGL((qk_rmsnorm_bwd_b16_kernel<2>), dim3(grid), dim3(256), 0, (hipStream_t)stream, qkv, dqkv, wq, wk, rstd_q, rstd_k, M, D, dwq_part, dwk_part, m_dev);
    return ivh_host::check_launch("qk_rmsnorm_bwd");
  }
  IVH_DISPATCH_NCH(nch, qk_rmsnorm_bwd_kernel, dim3(grid), dim3(256), (size_t)4 * D * sizeof(float), (hipStream_t)stream,
                   qkv, dqkv, wq, wk, rstd_q, rstd_k, M, D, dwq_part, dwk_part, m_dev);
  return ivh_host::check_launch("qk_rmsnorm_bwd");
}
extern "C" int ivh_qk_rmsnorm_bwd(const uint16_t* qkv, uint16_t* dqkv, const float* wq, const float* wk,
                                  const float* rstd_q, const float* rstd_k, int M, int D,
                                  float* dwq_part, float* dwk_part, void* stream) {
  return ivh_qk_rmsnorm_bwd_dyn(qkv, dqkv, wq, wk, rstd_q, rstd_k, M, D, dwq_part, dwk_part, nullptr, stream);
}

extern "C" int ivh_ln_l2_fwd(const uint16_t* y, const float* w, const float* b, float eps, int M, int C,
                             uint16_t* out, float* stats, const void* target, int target_bf16, float* loss_rows, void* stream) {
  IVH_REQUIRE(y && w && b && M > 0 && C % 8 == 0, "ln_l2_fwd: bad args");
  const int nch = nch_for(C);
  // Rows wider than 2048 elements (the clip decoders' 3200): ONE WAVE PER ROW here too (round 5).  The shared-row form (four waves per row,
  // NCH = 2) that the backward kernels need for their per-column accumulators pays four LDS exchanges + barriers per row in a kernel that keeps
  // nothing per column: 192-203 us per 53376 x 3200 launch = 3.4 TB/s.  A wave holding the whole row (5-8 chunks per lane, 56 + 28 registers at
  // 3200) reduces with shuffles only.  IVH_LNL2_FWD_WIDE=4 restores the shared-row form (A/B).
  static const int wide = [] { const char* e = getenv("IVH_LNL2_FWD_WIDE"); return e ? atoi(e) : 1; }();
  if (nch >= 5 && nch <= 8 && wide == 1) {
    const dim3 grid(row_grid(M, 8192)), block(256);
#define IVH_LNL2_FWD1(N) hipLaunchKernelGGL((ln_l2_fwd_kernel<N, 1>), grid, block, 0, (hipStream_t)stream, y, w, b, eps, M, C, out, stats, target, target_bf16, loss_rows)
    if (nch == 5) IVH_LNL2_FWD1(5); else if (nch == 6) IVH_LNL2_FWD1(6); else if (nch == 7) IVH_LNL2_FWD1(7); else IVH_LNL2_FWD1(8);
#undef IVH_LNL2_FWD1
    return ivh_host::check_launch("ln_l2_fwd");
  }
  IVH_DISPATCH_NCH(nch, ln_l2_fwd_kernel, dim3(row_grid(M, 8192)), dim3(256), 0, (hipStream_t)stream,
                   y, w, b, eps, M, C, out, stats, target, target_bf16, loss_rows);
  return ivh_host::check_launch("ln_l2_fwd");
}

extern "C" int ivh_ln_l2_bwd(const uint16_t* y, const float* w, const float* b, const float* stats, const void* dout, int dout_bf16,
                             const void* target, int target_bf16, float dscale, const float* dscale_dev, int M, int C,
                             uint16_t* dy, float* dw_part, float* db_part, void* stream) {
  IVH_REQUIRE(y && w && b && stats && dy && dw_part && db_part && (dout || target) && M > 0 && C % 8 == 0, "ln_l2_bwd: bad args");
  const int nch = nch_for(C);
  const int grid = row_grid(M, BWD_PARTS_CAP);
  static const int pf = [] { const char* e = getenv("IVH_LNL2_PF"); return e ? atoi(e) : 1; }();      // IVH_LNL2_PF=0: the generic kernel (A/B)
  const int n4 = (C / 8 + 255) / 256;
  if (pf > 0 && !dout && target && target_bf16 && n4 <= 2 && (long)M * C * 2 < (1L << 31)) {
#define IVH_LNL2_PF_LAUNCH(N) hipLaunchKernelGGL((ln_l2_bwd_pf_kernel<N>), dim3(grid), dim3(256), 0, (hipStream_t)stream, y, w, b, stats, \
    (const bf16_t*)target, dscale, dscale_dev, M, C, dy, dw_part, db_part)
    if (n4 == 1) IVH_LNL2_PF_LAUNCH(1); else IVH_LNL2_PF_LAUNCH(2);
#undef IVH_LNL2_PF_LAUNCH
    return ivh_host::check_launch("ln_l2_bwd");
  }
  IVH_DISPATCH_NCH(nch, ln_l2_bwd_kernel, dim3(grid), dim3(256), (size_t)4 * C * sizeof(float), (hipStream_t)stream,
                   y, w, b, stats, dout, dout_bf16, target, target_bf16, dscale, dscale_dev, M, C, dy, dw_part, db_part);
  return ivh_host::check_launch("ln_l2_bwd");
}

extern "C" int ivh_sum_rows(const float* x, int n, float scale, float* out, void* stream) {
  IVH_REQUIRE(x && out && n > 0, "sum_rows: bad args");
  hipLaunchKernelGGL(sum_rows_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, n, scale, out);
  return ivh_host::check_launch("sum_rows");
}

#define IVH_DISPATCH_NCH_T(nch, KERNEL, TX, grid, block, shmem, s, ...)                                  \
  switch (nch) {                                                                                         \
    case 1: hipLaunchKernelGGL((KERNEL<1, TX>), grid, block, shmem, s, __VA_ARGS__); break;              \
    case 2: hipLaunchKernelGGL((KERNEL<2, TX>), grid, block, shmem, s, __VA_ARGS__); break;              \
    case 3: hipLaunchKernelGGL((KERNEL<3, TX>), grid, block, shmem, s, __VA_ARGS__); break;              \
    case 4: hipLaunchKernelGGL((KERNEL<4, TX>), grid, block, shmem, s, __VA_ARGS__); break;              \
    case 5: case 6: case 7: case 8: hipLaunchKernelGGL((KERNEL<2, TX, 4>), grid, block, shmem, s, __VA_ARGS__); break; \
    default: ivh_host::set_error("row width %d not supported (max 4096)", (nch) * 512); return -1;       \
  }

extern "C" int ivh_layernorm_fwd(const void* x, int x_fp32, const float* w, const float* b, const float* w2, const float* b2,
                                 float eps, int M, int C, uint16_t* y, uint16_t* y2, float* stats, void* stream) {
  IVH_REQUIRE(x && w && b && y && stats && M > 0 && C % 8 == 0, "layernorm_fwd: bad args");
  IVH_REQUIRE((y2 == nullptr) == (w2 == nullptr) && (w2 == nullptr) == (b2 == nullptr), "layernorm_fwd: second head needs w2, b2, y2");
  const int nch = nch_for(C);
  if (x_fp32) {
    IVH_DISPATCH_NCH_T(nch, layernorm_fwd_kernel, float, dim3(row_grid(M, 8192)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)x, w, b, w2, b2, eps, M, C, y, y2, stats);
  } else {
    IVH_DISPATCH_NCH_T(nch, layernorm_fwd_kernel, bf16_t, dim3(row_grid(M, 8192)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, w, b, w2, b2, eps, M, C, y, y2, stats);
  }
  return ivh_host::check_launch("layernorm_fwd");
}

extern "C" int ivh_layernorm_bwd(const void* x, int x_fp32, const float* w, const float* w2, const float* stats,
                                 const uint16_t* dy, const uint16_t* dy2, int M, int C, float* dx, int accumulate,
                                 float* dw_part, float* db_part, float* dw2_part, float* db2_part, void* stream) {
  IVH_REQUIRE(x && w && stats && dy && dx && dw_part && db_part && M > 0 && C % 8 == 0, "layernorm_bwd: bad args");
  IVH_REQUIRE((dy2 == nullptr) == (w2 == nullptr), "layernorm_bwd: second head needs w2 and dy2");
  const int nch = nch_for(C);
  const int grid = row_grid(M, BWD_PARTS_CAP);
  const size_t sh = (size_t)4 * C * sizeof(float);
  if (x_fp32) {
    IVH_DISPATCH_NCH_T(nch, layernorm_bwd_kernel, float, dim3(grid), dim3(256), sh, (hipStream_t)stream,
                       (const float*)x, w, w2, stats, dy, dy2, M, C, dx, accumulate, dw_part, db_part, dw2_part, db2_part);
  } else {
    IVH_DISPATCH_NCH_T(nch, layernorm_bwd_kernel, bf16_t, dim3(grid), dim3(256), sh, (hipStream_t)stream,
                       (const bf16_t*)x, w, w2, stats, dy, dy2, M, C, dx, accumulate, dw_part, db_part, dw2_part, db2_part);
  }
  return ivh_host::check_launch("layernorm_bwd");
}

extern "C" int ivh_token_mean_fwd(const float* x, int B, int L, int D, float* out, void* stream) {
  IVH_REQUIRE(x && out && B > 0 && L > 0 && D % 8 == 0, "token_mean_fwd: bad args");
  const long n = (long)B * (D / 8);
  hipLaunchKernelGGL(token_mean_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, B, L, D, out);
  return ivh_host::check_launch("token_mean_fwd");
}
extern "C" int ivh_token_mean_bwd(const float* dmean, int B, int L, int D, float* dx, void* stream) {
  IVH_REQUIRE(dmean && dx && B > 0 && L > 0 && D % 8 == 0, "token_mean_bwd: bad args");
  const long n = (long)B * L * (D / 8);
  hipLaunchKernelGGL(token_mean_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dmean, B, L, D, dx);
  return ivh_host::check_launch("token_mean_bwd");
}
