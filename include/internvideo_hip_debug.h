/* internvideo_hip_debug.h -- measurement hooks and hardware probes of libinternvideo_hip.so.
 *
 * NOT part of the drop-in contract: include/internvideo_hip.h is what a reference-side binding binds; nothing here is
 * needed to run the path.  These entry points exist for tools/ (timelines, K-loop ablations, A/B switches, counter calibration) and for
 * the tests that pin the hardware semantics the kernels rely on (tests/test_kernels_gpu.py).  Process-wide switches, not thread-safe.
 */
#ifndef INTERNVIDEO_HIP_DEBUG_H
#define INTERNVIDEO_HIP_DEBUG_H
#include "internvideo_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* measurement aids for the 256x256 kernel (tools/bench_gemm.py): stagger = start-up skew unit (-1 = automatic, 0 = off),
 * skip_stores != 0 drops the C / preact stores (results are then NOT written). */
int ivh_gemm256_debug(int stagger, int skip_stores);
/* device buffer of 128 uint64 (or NULL): workgroup 0 records s_memtime stamps (4 per tile: K loop start, K loop end, DMA wait done,
 * epilogue end) for wave 0 ([0..63]) and wave 4 ([64..127]) */
int ivh_gemm256_debug_stamps(void* buf_128_u64);
int ivh_gemm256_debug_max_wg(int n);            /* cap the persistent grid (0 = one workgroup per CU) */
int ivh_gemm256_debug_sched(int sched);         /* K-loop schedule: 0 = two-group ping-pong, 1 = rolling (gemm256.hip) */
/* measurement aid: buf = device array of rows x 4 uint64 that receives wave 0's shader-clock stamps (entry, loop start, loop end, exit) of every
 * forward workgroup of the 32x32x16 attention kernel launched while it is set (tools/attn_timeline.py); NULL switches it off */
int ivh_attn32_debug_stamps(void* buf, int64_t rows);
int ivh_gemm256_debug_split(int on);           /* 0 = never split the tail round along K (A/B, tests); default 1 */
/* half-width tiles of the 256x256 kernel (ivh_gemm256_half_plan in internvideo_hip.h): 0 = off (A/B, tests), 1 = on, a workgroup's half tile
 * runs between its whole tiles (default), 2 = on, half tiles last */
int ivh_gemm256_debug_half(int on);
int ivh_gemm256_debug_ablate(int mode);         /* K-loop ablation of the plain NT kernel: 0 off, 1 no MFMA, 2 no LDS-DMA, 3 no fragment reads (garbage results) */

/* probes used by tests/test_hw_probe.py to pin the hardware semantics the kernels rely on */
int ivh_probe_tr16(const uint16_t* in_4x16x4, uint16_t* out_64x4, void* stream);
int ivh_probe_mfma16(const uint16_t* a16x32, const uint16_t* b16x32, float* c16x16, void* stream);
int ivh_probe_mfma32(const uint16_t* a32x16, const uint16_t* b32x16, float* c32x32, void* stream);   /* c[i][j] = sum_k a[i][k] b[j][k], 32x32x16 layout */
/* known-rate MFMA stream (counter calibration, tools/pmc_mfma.py): `workgroups` x 4 waves x iters x 8 MFMAs 32x32x16 bf16 = 32768 FLOP each */
int ivh_probe_mfma_rate(int iters, int workgroups, float* sink, void* stream);
/* the same stream on either bf16 MFMA shape (0: 32x32x16, 1: 16x16x32; 262144 FLOP per wave and iteration both ways), 1 or 2 waves per SIMD */
int ivh_probe_mfma_rate2(int shape, int waves_per_simd, int iters, int workgroups, float* sink, void* stream);

/* a bounded number of CUs kept busy with memory traffic: `workgroups` x 256 threads copy `bytes` (multiple of 16) src -> dst.  bench.py runs it on a
 * side stream beside the training step with the wire bytes of an 8-GPU gradient all-reduce to price what collective kernels cost the step
 * (`comm_contention`): the part of the scaling prediction that one GPU can measure. */
int ivh_probe_cu_hog(const void* src, void* dst, int64_t bytes, int workgroups, void* stream);

/* round-5 prototype of the q/k-norm fusion (flash_attn32.hip, QKN): the 32x32 forward kernel on UN-normalised q, k with their per-token rstd
 * (rq, rk: fp32 [B * L]) and wqk = q_norm.weight * k_norm.weight (fp32 [H * hd]): Q scaled at its load, a per-key factor on the scores.
 * 64 < hd <= 96, Lq == Lk <= 512.  tools/probes/attn_qkn_fusion_probe.py prices it against qk_rmsnorm_fwd + the shipped kernel. */
int ivh_probe_attn32_fwd_qkn(const uint16_t* q, int64_t qsb, int64_t qsl, int64_t qsh, const uint16_t* k, const uint16_t* v, int64_t sb, int64_t sl, int64_t sh,
                             uint16_t* out, int64_t ob, int64_t ol, int64_t oh, float* lse, int B, int H, int Lq, int Lk, int hd, float scale,
                             const float* rq, const float* rk, const float* wqk, void* stream);

/* round-6 measurement switch (flash_attn32.hip, attn32pp_fwd_kernel): the forward as 8-wave workgroups whose two wave groups run one segment apart
 * (MFMA segment of one beside the softmax segment of the other).  0 = off (default; also the environment variable IVH_ATTN_PP read once),
 * 1 = on, 2 = on with the MFMA segments at raised wave priority, 3 = groups of even / odd waves, 4 / 5 = 1 / 2 without packed fp32 instructions,
 * 6 = the one-group kernel without packed fp32 instructions, 7 = the same at two waves per SIMD.  tools/bench_attn.py --pingpong prices it against the shipped kernel. */
int ivh_probe_attn32_pingpong(int mode);

/* round-6 switch: the three 32x32 attention kernels compiled with (0) or without (1, the default; environment IVH_ATTN_NOPK=0 selects 0) the packed fp32
 * instructions.  v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 of one wave do not run beside another wave's MFMAs on the same SIMD
 * (tools/probes/mfma_valu_mix.hip); results are bit-identical either way. */
int ivh_probe_attn32_unpacked(int on);

#ifdef __cplusplus
}
#endif
#endif
