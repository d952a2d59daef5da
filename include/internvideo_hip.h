/* C ABI of libinternvideo_hip.so: the gfx950 (MI355X) kernels of the InternVideo2 video-ViT training hot path.
 *
 * The reference (OpenGVLab/InternVideo) has no FFI of its own: its "operator API" for this path is the
 * Python seam B3 of SURVEY.md 8(b) (three flash_attn CUDA ops + cuDNN Conv3d + cuBLAS Linear).  Every entry
 * point below names the reference call site it replaces (paths relative to
 * InternVideo2/single_modality/, "P:" = models/internvideo2_pretrain.py).  Conventions:
 *   - plain pointers into device memory (HBM) + sizes; no torch / C++ types;
 *   - `stream` is a hipStream_t passed as void*; every function only ENQUEUES work on that stream;
 *   - bf16 tensors are `uint16_t` storage, row-major, innermost dimension contiguous unless a leading
 *     dimension (ld*, in elements) is given; base pointers 16-byte aligned, ld* multiples of 8;
 *   - return value 0 = enqueued, <0 = rejected (ivh_last_error() holds the reason, nothing was launched).
 */
#ifndef INTERNVIDEO_HIP_H
#define INTERNVIDEO_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* ivh_last_error(void);
int ivh_version(void);
/* Version of the BINARY interface; a caller built against this header checks ivh_abi_version() == IVH_ABI_VERSION before anything else.
 *   1 = rounds 1-5 (no such symbol: its absence means 1).  Within it, round 5 changed two contracts without a version:
 *       ivh_rmsnorm_add_bwd_bf16res gained `dres_extra` in front of `stream`, and ivh_qk_rmsnorm_bwd sizes its partial-sum arrays with
 *       ivh_qk_norm_bwd_parts(M, D) (up to 768 rows) instead of ivh_norm_bwd_parts(M) (up to 512).
 *   2 = round 6: ivh_gemm_desc grew by two trailing pointers (m_dev, k_dev: a descriptor allocated with the old size must not be passed);
 *       new entry points ivh_droppath_plan, ivh_rmsnorm_add_{fwd,bwd}_skip, ivh_colsum_finish_dyn, ivh_qk_rmsnorm_{fwd,bwd}_dyn,
 *       ivh_flash_attn_{fwd,bwd}_dyn.  Every signature of ABI 1 (as of round 5) is unchanged. */
#define IVH_ABI_VERSION 2
int ivh_abi_version(void);
/* Device-side dropout epoch.  The dropout masks of ivh_bert_embed_*, ivh_add_layernorm_* and ivh_flash_attn_*_dropout are counter based:
 * keep(element) = hash(seed, element index) >= p * 2^32, `seed` a launch argument (multi_modality/models/backbones/bert/xbert.py:288,331,
 * 506-510 nn.Dropout sites; config_bert_large.json hidden_dropout_prob / attention_probs_dropout_prob 0.1).  A launch argument is frozen
 * into a captured HIP graph; with a registered epoch (one uint32 in HBM) the effective seed is seed + *epoch * 0x9E3779B1, and the training
 * engine advances *epoch inside the captured step: fresh masks on every replay, the same masks in forward and backward of one step.
 * NULL (default) unregisters. */
int ivh_set_dropout_epoch(const void* dev_u32);
/* number of CUs, bytes of LDS per CU, gcnArchName check ("gfx950"); <0 if no usable device */
int ivh_device_info(int* n_cu, int* lds_bytes, char* arch, int arch_len);

/* ------------------------------------------------------------------------------------------------
 * GEMM  C[m,n] = epilogue( alpha * sum_k A(m,k) * B(n,k) )      bf16 in, fp32 MFMA accumulate.
 * Replaces cuBLAS nn.Linear (P:158,160,232,235,341,375-379), fused_dense_lib FusedMLP (P:268-269) and
 * the autograd dgrad / wgrad GEMMs behind them.
 *   a_kc / b_kc = 1: operand stored [rows][K] (K contiguous, ld = row stride)       -> "NT" forward
 *               = 0: operand stored [K][rows] (rows contiguous, ld = stride of k)  -> dgrad (B) / wgrad (A and B)
 * Epilogue, in order: +bias[n] -> store preact (bf16) -> activation -> * gelu'(dact_in[m,n]) -> store C.
 * batch > 1 runs `batch` independent problems (blockIdx.z) with the given element strides. */
typedef struct ivh_gemm_desc {
  const uint16_t* A; const uint16_t* B;
  int64_t lda, ldb;
  int32_t M, N, K;
  int32_t a_kc, b_kc;
  void* C; int64_t ldc; int32_t c_fp32;     /* 0: bf16 out, 1: fp32 out */
  const float* bias;                        /* [N] or NULL */
  int32_t act;                              /* 0 none, 1 GELU(erf), 2 GELU(tanh), 3 GELU(erf) with the DERIVATIVE exchanged:
                                               forward: `preact` receives gelu'(pre-activation) instead of the pre-activation
                                               (it falls out of the erf evaluation); backward: C *= dact_in as it is.  Saves
                                               the 128 erf + exp per lane and tile of the fc2 dgrad epilogue. */
  uint16_t* preact; int64_t ldp;            /* optional bf16 [M][N] copy of the pre-activation */
  const uint16_t* dact_in; int64_t ldd;     /* optional bf16 [M][N]: C *= act'(dact_in) (act selects the flavour) */
  float alpha;
  int32_t batch;
  float* colsum_part;                       /* optional fp32 [2 * ceil(M / 256)][N] (256^2 kernel, dact_in launches with act = 3 only): row block
                                               sums of C over m, i.e. the bias gradient of the layer whose dgrad this is, as a
                                               by-product of the epilogue; reduce with ivh_colsum_finish.  NULL = not wanted. */
  int64_t strideA, strideB, strideC, stride_bias, stride_preact, stride_dact;
  void* split_ws; int64_t split_ws_bytes;   /* optional device scratch (16-byte aligned, contents irrelevant) of at least
                                               ivh_gemm_split_workspace(d) bytes: lets the 256x256 kernel cut the tiles of a mostly
                                               empty last round into K slices, one per idle workgroup (same result up to the fp32
                                               summation order, which is fixed: run-to-run deterministic).  NULL = never split. */
  /* ABI 2 -- DEVICE-SIDE ROW COUNTS (DropPath skipping, see ivh_droppath_plan).  The token-row count of the problem may live in device
   * memory, so that a launch captured into a HIP graph follows the per-step DropPath draw: the launch is sized for M (K), the kernel reads
   * the real count and never starts the tiles (K steps) beyond it; rows at or past the count are neither read nor written.
   *   m_dev: int32 in HBM, 0 <= *m_dev <= M: rows of A / C / preact / dact_in (a_kc = 1: forward and dgrad launches);
   *   k_dev: int32 in HBM, 0 <= *k_dev <= K: contraction length of a weight gradient (a_kc = b_kc = 0, plain epilogue); per problem in
   *          ivh_gemm_grouped_bf16.  *k_dev = 0 writes C = 0.
   * 256x256 kernel only (bf16 output, batch 1, act 0 / 1-forward / 3); anything else is rejected.  colsum_part then holds
   * 2 * ceil(*m_dev / 256) valid rows: reduce with ivh_colsum_finish_dyn.  NULL (zero-initialised descriptors) = M / K as given. */
  const int32_t* m_dev; const int32_t* k_dev;
} ivh_gemm_desc;
int ivh_gemm_bf16(const ivh_gemm_desc* d, void* stream);
/* bytes of split_ws worth passing for *d (0 = the tail split does not apply / would not pay); the fields split_ws / split_ws_bytes of *d
 * are ignored by the query */
int64_t ivh_gemm_split_workspace(const ivh_gemm_desc* d);
/* n independent problems in one call.  Problems that share the operand layouts (ABI 2: K may differ from problem to problem) and have a plain bf16 epilogue (the four
 * weight-gradient GEMMs of up to three transformer blocks) run as ONE persistent 256x256 launch per <= 32 problems over their
 * concatenated tile lists, which fills the 256 CUs where each alone would leave 112-220 idle; anything else is launched one by one. */
int ivh_gemm_grouped_bf16(const ivh_gemm_desc* d, int n, void* stream);
/* ------------------------------------------------------------------------------------------------
 * fp8 (OCP e4m3fn) path -- BASELINE configs[4] "InternVideo2-6B encoder, 16x224^2 fp8 MFMA" (model: P:758-766, recipe
 * scripts/pretraining/6B_pt.sh:9-12,47-50; the reference itself trains in bf16: fp8 is this framework's option for the 6B GEMMs).
 * ivh_fp8_quantize: x bf16 [M][K] (ld) -> q e4m3 [M][K] (ldq, bytes) and, if qt != NULL, the transposed copy qt [K][ldt] (ldt >= M
 * rounded up to 16; the pad columns are written as zeros) with ONE scale for the tensor: scale_out[0] = max|x| / 448 (the
 * dequantisation multiplier, left in device memory), q = rne(x / scale).  amax_scratch: 4 bytes of device scratch.
 * ivh_gemm_fp8: C = epilogue(alpha * scale_a[0] * scale_b[0] * sum_k A(m,k) B(n,k)) with A, B = e4m3 bytes (d->A / d->B carry the byte
 * pointers, lda / ldb in bytes), both K-contiguous (a_kc = b_kc = 1: dgrad / wgrad use the transposed copies), fp32 accumulate on
 * v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales, epilogues as ivh_gemm_bf16 (bias, act 0..3, preact, dact_in; no batch,
 * no colsum_part).  K, lda, ldb multiples of 16. */
int ivh_fp8_quantize(const uint16_t* x, int64_t ld, int M, int K, uint8_t* q, int64_t ldq, uint8_t* qt, int64_t ldt,
                     float* scale_out, uint32_t* amax_scratch, void* stream);
/* Delayed scaling: quantise with the amax this call site saw on the PREVIOUS step (amax_prev, one fp32 in device memory; values beyond it
 * saturate at +-448) and collect this step's max|x| into amax_next (uint32 bit pattern of a non-negative float, atomicMax: the caller zeroes
 * it once per step).  One pass over x instead of two.  scale_out[0] = max(amax_prev, 1e-12) / 448. */
int ivh_fp8_quantize_delayed(const uint16_t* x, int64_t ld, int M, int K, uint8_t* q, int64_t ldq, uint8_t* qt, int64_t ldt,
                             const float* amax_prev, float* scale_out, uint32_t* amax_next, void* stream);
int ivh_gemm_fp8(const ivh_gemm_desc* d, const float* scale_a, const float* scale_b, void* stream);
/* Per-channel weight scales.  ivh_fp8_quantize_weight: W bf16 [N][K] (a Linear's weight, P:158-160 / 268-271) -> q e4m3 [N][K] with one
 * scale per ROW n (scale_rows[N]: the forward GEMM's B operand, output column n) and qt e4m3 [K][ldt] (ldt >= N rounded up to 16, pad
 * columns zero) with one scale per COLUMN k of W (scale_cols[K]: the dgrad GEMM's B operand W^T, output column k); a scale can leave a
 * contraction only along the output dimension, hence two independently rounded images.  amax_scratch: (N + K) * 4 bytes of device scratch.
 * ivh_gemm_fp8_cs: as ivh_gemm_fp8 with C = epilogue(alpha * scale_a[0] * scale_b_cols[n] * sum_k A(m,k) B(n,k)); scale_b_cols = d->N floats. */
int ivh_fp8_quantize_weight(const uint16_t* w, int64_t ld, int N, int K, uint8_t* q, int64_t ldq, uint8_t* qt, int64_t ldt,
                            float* scale_rows, float* scale_cols, uint32_t* amax_scratch, void* stream);
int ivh_gemm_fp8_cs(const ivh_gemm_desc* d, const float* scale_a, const float* scale_b_cols, void* stream);
int64_t ivh_gemm_fp8_split_workspace(const ivh_gemm_desc* d);   /* as ivh_gemm_split_workspace, for ivh_gemm_fp8 */
/* 0 = choose per problem (the persistent 256 x 256 e4m3 kernel for large problems), 1 = always the 128 x 128 e4m3 kernel (A/B, tests) */
int ivh_set_gemm_fp8_kernel(int choice);
/* Kernel selection for ivh_gemm_bf16: 0 = per-shape heuristic (default), 1 = 128x128 tile / 4-wave kernel,
 * 2 = 256x256 tile / 8-wave LDS-DMA ping-pong kernel.  Process-wide; meant for tests and benchmarks. */
int ivh_set_gemm_kernel(int choice);
/* which kernel ivh_gemm_bf16 would launch for *d under the current choice: 1 or 2 (launch-time cost model, gemm.hip) */
int ivh_gemm_select(const ivh_gemm_desc* d);
/* Half-width tiles of the 256x256 kernel (gemm256.hip, HALF): an output whose last column tile is at most 128 wide (1408 = 5.5 x 256,
 * 4224 = 16.5 x 256: the proj / fc2 / qkv shapes of single_modality/models/internvideo2_pretrain.py:158-160,268-271) gets that column
 * as tiles that skip their zero half, and the leftover whole tiles of the last round are cut into two column halves, all scheduled last.
 * ivh_gemm256_half_plan: the plan ivh_gemm_bf16 would use for `d` on `cap` workgroups (0 = the device's CUs),
 * out4 = {whole column tiles, first half-tile id, ids that are halves of whole tiles, total ids}; returns 1 when half tiles are used.
 * ivh_gemm256_half_rounds = modelled launch length in rounds or -1.  (Switching the feature off for A/B runs: internvideo_hip_debug.h, or env IVH_NO_HALF=1.) */
int ivh_gemm256_half_plan(const ivh_gemm_desc* d, int cap, int* out4);
double ivh_gemm256_half_rounds(const ivh_gemm_desc* d);

/* ------------------------------------------------------------------------------------------------
 * Residual-stream RMSNorm with fused LayerScale / DropPath / residual add.
 * Replaces flash_attn DropoutAddRMSNorm(prenorm=True) (P:466-467, called P:283-286), LayerScale
 * (P:131-146), DropPath (P:264,274) and the residual adds of Block.forward (P:279-292):
 *     res_out[m,:] = res_in[m,:] + rowscale[m / rows_per_sample] * gamma[:] * branch[m,:]        (fp32)
 *     y[m,:]       = bf16( res_out[m,:] * rsqrt(mean(res_out[m,:]^2) + eps) * w[:] )
 * res_in may be NULL (first block: res_out = branch-less input x0 given as `branch` with gamma NULL),
 * branch may be NULL (plain norm), gamma / rowscale may be NULL (= 1).  y / w may be NULL (add only:
 * the final `x + residual`, P:685-688).  rstd[m] (fp32) is saved for the backward. */
int ivh_rmsnorm_add_fwd(const float* res_in, const uint16_t* branch, const float* gamma, const float* rowscale,
                        int rows_per_sample, const float* w, float eps, int M, int D,
                        float* res_out, uint16_t* y, float* rstd, void* stream);
/* backward of the above.  Inputs: dy (bf16, grad wrt y, may be NULL), dres_out (fp32, grad wrt res_out from
 * later consumers, may be NULL), saved res_out, rstd.  Outputs: dres_in (fp32; may alias dres_out),
 * dbranch (bf16, = rowscale*gamma*dres), partial column sums for dw, dgamma and -- optionally -- of dbranch itself, which is the
 * bias gradient of the Linear that produced `branch` (attn.proj / mlp.fc2: saves a separate pass over dbranch):
 * dw_part / dgamma_part / dbias_part are [n_part][D] fp32 (dbias_part may be NULL), reduced by ivh_colsum_finish(_multi);
 * n_part = ivh_norm_bwd_parts(M). */
int ivh_norm_bwd_parts(int M);
int ivh_rmsnorm_add_bwd(const uint16_t* dy, const float* dres_out, const float* res_out, const float* rstd,
                        const float* w, const uint16_t* branch, const float* gamma, const float* rowscale,
                        int rows_per_sample, int M, int D,
                        float* dres_in, uint16_t* dbranch, float* dw_part, float* dgamma_part, float* dbias_part, void* stream);
/* The same pair with a bf16 residual stream: res_in / res_out / dres_out / dres_in are bf16 rows.  This is what the reference's own
 * bf16 recipe carries (DropoutAddRMSNorm(prenorm=True) with residual_in_fp32 left False, P:283-286, 467; the unfused path under
 * model.bfloat16() likewise): 8 instead of 12 bytes per element forward, 10 instead of 16 backward.  The sum is formed in fp32, the norm
 * is taken from it BEFORE it is rounded to bf16 for the stream (as flash_attn's fused kernel does); the backward normalises the stored
 * (rounded) rows. */
int ivh_rmsnorm_add_fwd_bf16res(const uint16_t* res_in, const uint16_t* branch, const float* gamma, const float* rowscale,
                                int rows_per_sample, const float* w, float eps, int M, int D,
                                uint16_t* res_out, uint16_t* y, float* rstd, void* stream);
int ivh_rmsnorm_add_bwd_bf16res(const uint16_t* dy, const uint16_t* dres_out, const uint16_t* res_out, const float* rstd,
                                const float* w, const uint16_t* branch, const float* gamma, const float* rowscale,
                                int rows_per_sample, int M, int D,
                                uint16_t* dres_in, uint16_t* dbranch, float* dw_part, float* dgamma_part, float* dbias_part,
                                const uint16_t* dres_extra, void* stream);
/* dres_extra (bf16 [M][D], may be NULL; needs dres_out): a second gradient of the same rows, added to dres_out in fp32 as it is loaded --
 * the gradient of a feature tap (P:669-688: the decoders read the stream after chosen blocks), which otherwise costs a read-modify-write
 * pass over dres_out before this call. */
/* ------------------------------------------------------------------------------------------------
 * DropPath SAMPLE SKIPPING (ABI 2).  timm's DropPath (P:264,274; rate linspace(0, drop_path_rate, depth): 0.25 in scripts/pretraining/
 * 1B_pt.sh:46) multiplies a branch's output by 0 for a dropped sample after computing it in full -- on average 12.5 % of the block work of
 * the 1B recipe.  Here the draw (`rowscale`, fp32 [n_sets][B], 0 = dropped, 1 / keep = kept; n_sets = 2 * depth branches) is turned into keep
 * maps on the device, the norm in front of a branch writes its output COMPACTED over the kept samples, the branch's GEMMs / attention /
 * q-k-norm run on the kept rows only (device-side counts: ivh_gemm_desc.m_dev / k_dev, the *_dyn entry points) and the next residual add
 * scatters the branch back by the same map.  Exact: a dropped sample's branch contributes 0 to the stream and to every gradient either way.
 *   slot  int32 [n_sets][B]: position of sample b among the kept samples of its set (ascending b), -1 = dropped;
 *   count int32 [n_sets][2]: {kept samples, kept samples * rows_per_sample}. */
int ivh_droppath_plan(const float* rowscale, int n_sets, int B, int rows_per_sample, int32_t* slot, int32_t* count, void* stream);
/* ivh_rmsnorm_add_fwd / _bwd with keep maps (res_bf16 selects the stream's type: res_in / res_out / dres_* are float or bf16 rows):
 *   branch_slot [M / rows_per_sample] or NULL: `branch` (forward) / `branch`, `dbranch` (backward) hold the kept samples only, sample s at
 *       rows branch_slot[s] * rows_per_sample ...; a dropped sample's stream passes through, nothing of it is read or written there;
 *   y_slot or NULL: the same for y (forward: a dropped sample's y is not computed) / dy (backward: it has none).
 * The stream, rstd and the partial sums keep the full row numbering; M must be whole samples. */
int ivh_rmsnorm_add_fwd_skip(const void* res_in, int res_bf16, const uint16_t* branch, const float* gamma, const float* rowscale,
                             int rows_per_sample, const float* w, float eps, int M, int D,
                             void* res_out, uint16_t* y, float* rstd, const int32_t* branch_slot, const int32_t* y_slot, void* stream);
int ivh_rmsnorm_add_bwd_skip(const uint16_t* dy, const void* dres_out, int res_bf16, const void* res_out, const float* rstd,
                             const float* w, const uint16_t* branch, const float* gamma, const float* rowscale,
                             int rows_per_sample, int M, int D, void* dres_in, uint16_t* dbranch,
                             float* dw_part, float* dgamma_part, float* dbias_part, const void* dres_extra,
                             const int32_t* y_slot, const int32_t* branch_slot, void* stream);
/* ivh_colsum_finish over partial rows that were produced per block of rows of a matrix whose row count lives in device memory: only the
 * first parts_per_unit * ceil(*m_dev / rows_per_unit) of the n_part rows are summed (ivh_gemm_desc.colsum_part under m_dev: 2 per 256). */
int ivh_colsum_finish_dyn(const float* part, int n_part, int D, float* out, int accumulate, const int32_t* m_dev, int rows_per_unit,
                          int parts_per_unit, void* stream);
/* out[d] (+)= sum_p part[p][d]  (deterministic second stage of every column reduction) */
int ivh_colsum_finish(const float* part, int n_part, int D, float* out, int accumulate, void* stream);
/* the same for n <= 4 (part, out) pairs of one shape in a single launch (the dw / dgamma / db partials of one norm backward) */
int ivh_colsum_finish_multi(const float* const* parts, float* const* outs, int n, int n_part, int D, int accumulate, void* stream);
/* column sums of a bf16 matrix: bias gradients (nn.Linear bias, P:160,232,235).  out fp32 [N]. */
int ivh_colsum_bf16(const uint16_t* x, int64_t ld, int M, int N, float* out, float* scratch, void* stream);
int ivh_colsum_scratch_floats(int M, int N);

/* ------------------------------------------------------------------------------------------------
 * q/k RMSNorm over the concatenated-heads axis (P:178-181 / P:198-206), in place on the packed
 * qkv (M, 3, D) bf16 buffer.  rstd_q/rstd_k [M] fp32 saved for backward. */
int ivh_qk_rmsnorm_fwd(uint16_t* qkv, const float* wq, const float* wk, float eps, int M, int D,
                       float* rstd_q, float* rstd_k, void* stream);
/* qkv holds the NORMALISED q,k (as left by the forward); dqkv holds d(q_hat), d(k_hat), dv and is
 * rewritten in place to dq, dk, dv (pre-norm).  dwq_part/dwk_part: [n_part][D], n_part = ivh_qk_norm_bwd_parts(M, D) (this kernel's own
 * workgroup count: three resident workgroups per CU for the bytes-in-flight form). */
int ivh_qk_norm_bwd_parts(int M, int D);
int ivh_qk_rmsnorm_bwd(const uint16_t* qkv, uint16_t* dqkv, const float* wq, const float* wk,
                       const float* rstd_q, const float* rstd_k, int M, int D,
                       float* dwq_part, float* dwk_part, void* stream);
/* ABI 2: the same with the row count in device memory (int32, 0 <= *m_dev <= M; NULL = M): rows at or past it are neither read nor written;
 * buffers and partial-sum arrays are sized for M. */
int ivh_qk_rmsnorm_fwd_dyn(uint16_t* qkv, const float* wq, const float* wk, float eps, int M, int D,
                           float* rstd_q, float* rstd_k, const int32_t* m_dev, void* stream);
int ivh_qk_rmsnorm_bwd_dyn(const uint16_t* qkv, uint16_t* dqkv, const float* wq, const float* wk,
                           const float* rstd_q, const float* rstd_k, int M, int D,
                           float* dwq_part, float* dwk_part, const int32_t* m_dev, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Non-causal softmax attention, equal-length sequences, no dropout.
 * Replaces flash_attn_varlen_qkvpacked_func with cu_seqlens = arange(0,(B+1)L,L)
 * (models/flash_attention_class.py:41-50) and Attention._naive_attn (P:173-191).
 * q: bf16 with element strides (qsb,qsl,qsh); k,v: bf16 sharing strides (sb,sl,sh) = (batch, token, head); head dim contiguous; hd in {64, 88, 96, 128} (any
 * multiple of 8 up to 128).  out: (B, L, H, hd) bf16 with strides; lse: (B, H, L) fp32 (natural log).
 * kv_len (int32 [B] on the device, or NULL): right-padded batches -- clip b attends to its first kv_len[b] keys only (clamped to
 * [1, Lk]); the key_padding_mask branch of FlashAttention.forward (models/flash_attention_class.py:51-62, unpad / pad) for masks
 * that are a prefix of ones, as text batches are.  Rows of padded queries are computed like any other: the caller zeroes them. */
int ivh_flash_attn_fwd(const uint16_t* q, int64_t qsb, int64_t qsl, int64_t qsh,
                       const uint16_t* k, const uint16_t* v, int64_t sb, int64_t sl, int64_t sh,
                       uint16_t* out, int64_t ob, int64_t ol, int64_t oh,
                       float* lse, int B, int H, int Lq, int Lk, int hd, float scale, const int32_t* kv_len, void* stream);
/* backward: dq written with strides (dqb,dql,dqh), dk/dv with (dsb,dsl,dsh).  delta (B,H,Lq) fp32: written by the dQ kernel
 * (<dO, O> per query), read by the dK/dV kernel.  Lq != Lk is allowed (the attention-pooling projector, P:50-80, is the Lq = 1
 * case).  With kv_len, dK / dV rows of padded keys are written as zeros. */
int ivh_flash_attn_bwd(const uint16_t* q, int64_t qsb, int64_t qsl, int64_t qsh,
                       const uint16_t* k, const uint16_t* v, int64_t sb, int64_t sl, int64_t sh,
                       const uint16_t* out, const uint16_t* dout, int64_t ob, int64_t ol, int64_t oh,
                       const float* lse, float* delta,
                       uint16_t* dq, int64_t dqb, int64_t dql, int64_t dqh,
                       uint16_t* dk, uint16_t* dv, int64_t dsb, int64_t dsl, int64_t dsh,
                       int B, int H, int Lq, int Lk, int hd, float scale, const int32_t* kv_len, void* stream);
/* ABI 2: the same pair with the number of clips in device memory (int32, 0 <= *nb_dev <= B; NULL = B): the launch is sized for B, the
 * workgroups of clips at or past *nb_dev leave at once and nothing of those clips is read or written.  32x32-MFMA kernels only (rejected
 * where they do not support the layout). */
int ivh_flash_attn_fwd_dyn(const uint16_t* q, int64_t qsb, int64_t qsl, int64_t qsh,
                           const uint16_t* k, const uint16_t* v, int64_t sb, int64_t sl, int64_t sh,
                           uint16_t* out, int64_t ob, int64_t ol, int64_t oh,
                           float* lse, int B, int H, int Lq, int Lk, int hd, float scale, const int32_t* kv_len, const int32_t* nb_dev, void* stream);
int ivh_flash_attn_bwd_dyn(const uint16_t* q, int64_t qsb, int64_t qsl, int64_t qsh,
                           const uint16_t* k, const uint16_t* v, int64_t sb, int64_t sl, int64_t sh,
                           const uint16_t* out, const uint16_t* dout, int64_t ob, int64_t ol, int64_t oh,
                           const float* lse, float* delta,
                           uint16_t* dq, int64_t dqb, int64_t dql, int64_t dqh,
                           uint16_t* dk, uint16_t* dv, int64_t dsb, int64_t dsl, int64_t dsh,
                           int B, int H, int Lq, int Lk, int hd, float scale, const int32_t* kv_len, const int32_t* nb_dev, void* stream);
/* The same with dropout on the attention probabilities (xbert.py:361,469 `attention_probs_dropout_prob`; head dims <= 64: the text tower):
 * O = (softmax(S) o M) V, M = keep-mask / (1 - p) from hash(seed, ((b H + h) Lq + query) Lk + key); lse stays that of the undropped row.
 * The backward call must be given the same (p_drop, seed). */
int ivh_flash_attn_fwd_dropout(const uint16_t* q, int64_t qsb, int64_t qsl, int64_t qsh,
                               const uint16_t* k, const uint16_t* v, int64_t sb, int64_t sl, int64_t sh,
                               uint16_t* out, int64_t ob, int64_t ol, int64_t oh, float* lse,
                               int B, int H, int Lq, int Lk, int hd, float scale, const int32_t* kv_len,
                               float p_drop, uint32_t seed, void* stream);
int ivh_flash_attn_bwd_dropout(const uint16_t* q, int64_t qsb, int64_t qsl, int64_t qsh,
                               const uint16_t* k, const uint16_t* v, int64_t sb, int64_t sl, int64_t sh,
                               const uint16_t* out, const uint16_t* dout, int64_t ob, int64_t ol, int64_t oh,
                               const float* lse, float* delta, uint16_t* dq, int64_t dqb, int64_t dql, int64_t dqh,
                               uint16_t* dk, uint16_t* dv, int64_t dsb, int64_t dsl, int64_t dsh,
                               int B, int H, int Lq, int Lk, int hd, float scale, const int32_t* kv_len,
                               float p_drop, uint32_t seed, void* stream);
/* Kernel family behind ivh_flash_attn_fwd / _bwd: 0 = automatic (default: the 32x32x16-MFMA kernels of csrc/flash_attn32.hip --
 * 32 queries per wave, LDS-DMA double-buffered key / value tiles -- whenever every stride is a multiple of 8 elements and the
 * outputs are 16-byte aligned, else the 16x16x32 kernels of csrc/flash_attn.hip), 1 = 16x16x32 kernels only, 2 = 32x32x16 kernels
 * or an error.  Process-wide; meant for tests and benchmarks (same results either way, within bf16 rounding). */
int ivh_set_attn_kernel(int choice);

/* ------------------------------------------------------------------------------------------------
 * Tubelet patch embedding on the VISIBLE tokens only.
 * Replaces cuDNN Conv3d k = s = (t,p,p) (P:320-331), the cls-token cat + pos_embed add (P:634-656) and
 * the boolean-mask gather x[~mask] (P:659).
 * vis_idx (B, L) int32: ascending token ids incl. the cls token 0 (built by ivh_mask_to_indices).
 * im2col: cols[(b, j), :] = patch (c, dt, dy, dx) of token vis_idx[b, j+1]-1, zero padded to Kp. */
int ivh_mask_to_indices(const uint8_t* mask, int B, int N1, int L, int32_t* vis_idx, int32_t* inv_idx, int32_t* status, void* stream);
int ivh_patch_im2col(const void* video, int video_fp32, const int32_t* vis_idx, int B, int C, int T, int H, int W,
                     int tubelet, int patch, int L, int Kp, uint16_t* cols, void* stream);
/* x0[b, 0, :] = cls + pos[0];  x0[b, j, :] = tok[b, j-1, :] + pos[vis_idx[b, j]]  (fp32 residual stream) */
int ivh_assemble_tokens(const uint16_t* tok, const float* cls, const float* pos, const int32_t* vis_idx,
                        int B, int L, int D, float* x0, void* stream);
/* backward pieces of assemble / add_pos_gather (scatter-free, deterministic):
 *   rows_to_bf16: dst[b, j, :] = bf16(src[b, j + skip, :])                         (dtok for the patch-embed wgrad)
 *   accum_rows  : dst[b, j + skip, :] (+)= src[b, j, :]   fp32 <- bf16|fp32         (decoder-input grads into the stream)
 *   pos_grad    : dpos[n, :] (+)= sum_k sum_b src[k, b, inv_idx[b, n + skip] - skip, :] over the clips that kept token n
 *                 (pos_embed / clip_pos_embed: skip 0; mae_pos_embed: skip 1; cls_token grad = dpos row 0 of x0's call) */
int ivh_rows_to_bf16(const float* src, int B, int L, int D, int skip, uint16_t* dst, void* stream);
int ivh_accum_rows(float* dst, const void* src, int src_bf16, int B, int L, int D, int skip, int accumulate, void* stream);
int ivh_pos_grad(const void* src, int src_bf16, int K, int B, int Lsrc, int D, const int32_t* inv_idx, int N1, int skip,
                 float* dpos, int accumulate, void* stream);
/* y[b, j, :] = bf16( x[b, j + skip, :] + pos[idx[b, j + skip] - skip, :] ): decoder inputs (P:713-714, P:736-737) */
int ivh_add_pos_gather(const float* x, const float* pos, const int32_t* vis_idx, int B, int L, int D, int skip,
                       uint16_t* y, void* stream);
/* the same on a bf16 tap (model.residual_dtype = "bf16": taps leave the block stack in the stream's own type), and the gradient of such a
 * tap for a consumer that dropped its first `skip` rows: dst[b, j + skip] = src[b, j], rows j < skip zero (bf16 rows, 16 bytes per lane) */
int ivh_add_pos_gather_bf16(const uint16_t* x, const float* pos, const int32_t* vis_idx, int B, int L, int D, int skip,
                            uint16_t* y, void* stream);
int ivh_rows_shift_bf16(uint16_t* dst, const uint16_t* src, int B, int L, int D, int skip, void* stream);
/* dst[k, b, j, :] = src[k, b, idx[b, j + skip] - skip, :] on raw rows of row_bytes (multiple of 16) bytes: the teacher-target
 * gather norm_clip_middle[~mask].reshape(K, B, -1, C) / norm_mae[~mask[:, 1:]] of engines/engine_for_pretraining.py:118-125
 * (and engines/engine_for_distill.py:100-103, multi_modality/models/internvideo2_stage2_visual.py:225-235).  Bit-exact copy. */
int ivh_gather_rows(const void* src, int row_bytes, int K, int B, int Nsrc, const int32_t* idx, int L, int skip,
                    void* dst, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Teacher tails (forward only).  The CLIP teacher (single_modality/models/internvl_clip_vision.py:336-465) runs the student's
 * block kernels on B*T per-frame sequences of L = 1 + H*W tokens; these two kernels finish it:
 *  frames_merge_l2: x [B][T][L][C] -> out [B][1 + T*(L-1)][C]: the T cls rows averaged into row 0, patch rows concatenated
 *      frame-major, every row divided by its l2 norm when l2 != 0 (:445-453).  With L = 1: mean over frames + l2 of the pooled
 *      feature (:455-456).
 *  pool_attn_map: out[s][l - skip] = mean_h softmax_l(scale <q[s,h,:], k[s,l,h,:]>)  -- `attn.mean(1)` of the attention-pool
 *      CrossAttention (:82-83) restricted to the patch keys (`attn[:, 0, 1:]`, :463); it feeds torch.multinomial in
 *      engines/engine_for_pretraining.py:105-116.  q [S][H*hd] bf16; k rows at k + s*ks_s + l*ks_l (elements). */
int ivh_frames_merge_l2(const void* x, int x_fp32, int B, int T, int L, int C, int l2, void* out, int out_fp32, void* stream);
int ivh_pool_attn_map(const uint16_t* q, const uint16_t* k, int64_t ks_s, int64_t ks_l, int S, int L, int H, int hd,
                      float scale, int skip, float* out, void* stream);
/* 1-query multi-head attention for head dims above the flash kernel's 128: the attention-pool projector of the 6B models (16 heads
 * over D = 3200 -> hd = 200; internvideo2_pretrain.py:18-114, internvl_clip_vision.py:23-86).  q [S][H*hd]; k, v rows at
 * base + s*ks_s + l*ks_l (elements, same strides); o [S][H*hd]; lse [S][H] (natural log);  backward: dq [S][H*hd],
 * dk / dv [S][L][H*hd] contiguous.  L <= 8192, hd <= 256 (multiple of 8). */
int ivh_pool_attn_fwd(const uint16_t* q, const uint16_t* k, const uint16_t* v, int64_t ks_s, int64_t ks_l, int S, int L, int H,
                      int hd, float scale, uint16_t* o, float* lse, void* stream);
int ivh_pool_attn_bwd(const uint16_t* q, const uint16_t* k, const uint16_t* v, int64_t ks_s, int64_t ks_l, const uint16_t* dout,
                      const float* lse, int S, int L, int H, int hd, float scale, uint16_t* dq, uint16_t* dk, uint16_t* dv, void* stream);

/* ------------------------------------------------------------------------------------------------
 * VideoMAE pixel-reconstruction path (InternVideo1/Pretrain/VideoMAE/modeling_pretrain.py "MP", engine_for_pretraining.py "ME").
 * The model has no cls token; index lists keep the student's convention (token ids + 1 with a leading pseudo-cls 0 in vis_idx)
 * so that ivh_mask_to_indices / ivh_patch_im2col are shared:
 *   vis_idx (B, 1 + Nvis) from mask' = [0 | mask];   msk_idx (B, Nmask) from mask'' = [1 | ~mask]   (values 1..N)
 *  assemble_tokens_nocls: x0[b, j] = tok[b, j] + pos[vis_idx[b, j+1] - 1]                      (MP:127-133, fp32 stream rows)
 *  mae_decoder_input    : cat([x_vis + pos[~mask], mask_token + pos[mask]], 1)                 (MP:381-389)
 *  rows_window(_bwd)    : bf16 copy of rows [start, start+count) of every clip's stream, and its zero-filling backward
 *                         (decoder tail x[:, -N_mask:], MP:264; gradient of the visible rows into encoder_to_decoder)
 *  pixel_target         : ME:66-98: un-normalise, (t 2)(h p)(w p) cubes of the masked tokens, per-cube per-channel mean /
 *                         unbiased-std (+1e-6) normalisation, (p0 p1 p2 c) order -> fp32 (B, Nmask, tubelet*p*p*3)
 *  mse_rows             : rows[m] = sum_c (pred - target)^2, dpred = bf16(2 dscale (pred - target))  (nn.MSELoss, ME:101-106) */
int ivh_assemble_tokens_nocls(const uint16_t* tok, const float* pos, const int32_t* vis_idx, int B, int L, int D, float* x0, void* stream);
int ivh_mae_decoder_input(const uint16_t* xvis, const float* mask_token, const float* pos, const int32_t* vis_idx,
                          const int32_t* msk_idx, int B, int Nvis, int Nmask, int D, float* out, void* stream);
int ivh_rows_window(const float* src, int B, int L, int D, int start, int count, uint16_t* dst, void* stream);
int ivh_rows_window_bwd(const void* src, int src_bf16, int B, int L, int D, int start, int count, float* dst, void* stream);
int ivh_pixel_target(const void* video, int video_fp32, const int32_t* msk_idx, int B, int C, int T, int H, int W,
                     int tubelet, int patch, int Nmask, int normalize, const float* mean3, const float* std3, float* out, void* stream);
int ivh_mse_rows(const void* pred, int pred_fp32, const float* target, int M, int C, float dscale, float* rows, uint16_t* dpred, void* stream);
/* rows[m] = 2 - 2 <s[m,:], t[m,:]>, ds = bf16(-2 dscale t): the alignment loss (2 - 2 (s * t).sum(-1)) of materialised, l2-normalised
 * student / teacher features (multi_modality/models/criterions.py:480-485 new_UTA_Loss.uta_loss; the fused decoder tail
 * ivh_ln_l2_fwd computes the same rows without materialising s).  s, t: bf16 or fp32 [M][C]. */
int ivh_cosine_rows(const void* s, int s_fp32, const void* t, int t_fp32, int M, int C, float dscale, float* rows, uint16_t* ds, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Decoder tail: LayerNorm(eps) -> x / ||x||_2 (P:355-365, P:393-403) and the distillation loss
 * (2 - 2 <s, t>).mean() of engines/engine_for_pretraining.py:131-148.
 * fwd: y (bf16 [M][C]) -> out (bf16 or NULL), stats (fp32 [M][3]: mean, rstd, 1/||ln||), and when target != NULL
 *      loss_part[row-block] partial sums of (2 - 2 <out_fp32, target>) for a deterministic mean. */
int ivh_ln_l2_fwd(const uint16_t* y, const float* w, const float* b, float eps, int M, int C,
                  uint16_t* out, float* stats, const void* target, int target_bf16, float* loss_rows, void* stream);
/* bwd: upstream gradient = dout ([M][C] fp32|bf16) or, when dout == NULL, dscale * target (the fused cosine loss:
 * dscale = -2 * ratio / n_rows, further multiplied by the device scalar dscale_dev[0] when given: the upstream
 * gradient of the loss).  Writes dy (bf16) and per-block partials of dw / db ([ivh_norm_bwd_parts(M)][C]). */
int ivh_ln_l2_bwd(const uint16_t* y, const float* w, const float* b, const float* stats, const void* dout, int dout_bf16,
                  const void* target, int target_bf16, float dscale, const float* dscale_dev, int M, int C,
                  uint16_t* dy, float* dw_part, float* db_part, void* stream);
/* ------------------------------------------------------------------------------------------------
 * Attention-pooling projector pieces (AttentionPoolingBlock / CrossAttention, P:18-114): mean query over all
 * tokens (P:110), LayerNorm with one or two affine heads sharing statistics (norm1_k / norm1_v read the same
 * tokens, P:99-101).  The q/k/v/proj Linears are ivh_gemm_bf16 and the 1-query attention is ivh_flash_attn_* (Lq=1). */
int ivh_token_mean_fwd(const float* x, int B, int L, int D, float* out, void* stream);
int ivh_token_mean_bwd(const float* dmean, int B, int L, int D, float* dx /* += */, void* stream);
/* x fp32|bf16 [M][C]; y (y2) bf16; stats fp32 [M][2] = (mean, rstd) */
int ivh_layernorm_fwd(const void* x, int x_fp32, const float* w, const float* b, const float* w2, const float* b2,
                      float eps, int M, int C, uint16_t* y, uint16_t* y2, float* stats, void* stream);
/* dx fp32 (= or += when accumulate); d{w,b}[2]_part: [ivh_norm_bwd_parts(M)][C] partial column sums */
int ivh_layernorm_bwd(const void* x, int x_fp32, const float* w, const float* w2, const float* stats,
                      const uint16_t* dy, const uint16_t* dy2, int M, int C, float* dx, int accumulate,
                      float* dw_part, float* db_part, float* dw2_part, float* db2_part, void* stream);
int ivh_sum_rows(const float* x, int n, float scale, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused AdamW on flat buffers (DeepSpeed FusedAdam adam_w_mode, utils.py:821-871):
 * master fp32, exp_avg, exp_avg_sq fp32, grad bf16 or fp32; writes the bf16 compute copy.
 * One call per parameter-group region (decay / no-decay).  The gradient is multiplied by grad_scale and, when
 * clip_coef != NULL, by the device scalar clip_coef[0] (global-norm clip, utils.py:860-861). */
int ivh_adamw_step(float* master, float* exp_avg, float* exp_avg_sq, const void* grad, int grad_bf16,
                   uint16_t* shadow_bf16, int64_t n, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int step, float grad_scale, const float* clip_coef, void* stream);
/* The same update with a per-segment learning-rate scale: layer-wise lr decay of the fine-tuning recipe
 * (single_modality/optim_factory.py:24-98 LayerDecayValueAssigner / get_parameter_groups "lr_scale", consumed at
 * engines/engine_for_finetuning.py:56 `param_group["lr"] = lr_schedule_values[it] * param_group["lr_scale"]`).  The flat REGION is a
 * run of nseg segments (whole parameters; boundaries multiples of 4 elements); seg_end[i] (device int64) = exclusive end offset of
 * segment i in the region, seg_scale[i] (device fp32) its lr_scale; this call updates the slice [seg_base, seg_base + n) of the region
 * (seg_base = 0 for the whole region, a shard offset under ZeRO-1).  Step of an element = lr * seg_scale[segment] for the Adam term
 * and the decoupled weight decay alike, as torch.optim.AdamW with per-group lr.  1 <= nseg <= 1024. */
int ivh_adamw_step_scaled(float* master, float* exp_avg, float* exp_avg_sq, const void* grad, int grad_bf16,
                          uint16_t* shadow_bf16, int64_t n, float lr, float beta1, float beta2, float eps,
                          float weight_decay, int step, float grad_scale, const float* clip_coef,
                          const int64_t* seg_end, const float* seg_scale, int nseg, int64_t seg_base, void* stream);
/* out[0] (+)= sum(g^2) over a flat buffer (fp32, deterministic two-stage); partial: ivh_sqnorm_scratch_floats() floats */
int ivh_sqnorm_scratch_floats(void);
int ivh_sqnorm(const void* g, int g_bf16, int64_t n, float* partial, float* out, int accumulate, void* stream);
/* coef[0] = min(1, max_norm / (sqrt(sumsq[0]) + 1e-6)); norm_out[0] = sqrt(sumsq[0])  (clip_grad_norm_ formula) */
int ivh_clip_coef(const float* sumsq, float max_norm, float* coef, float* norm_out, void* stream);
/* out[i] = sum_{r < W} float(in[r * chunk + i]) (fp32): the local half of a reduce-scatter whose wire format is bf16 -- the W chunks a
 * rank received from an all-to-all of bf16 gradient shards are accumulated in fp32, in rank order (deterministic).  W = 1 widens
 * bf16 -> fp32.  Replaces the fp32 gradient accumulation of DeepSpeed's ZeRO-1 reduce-scatter (single_modality/utils.py:863-871,
 * scripts/pretraining/1B_pt.sh:65).  chunk: multiple of 8 elements; both buffers 16-byte aligned. */
int ivh_shard_sum_bf16(const uint16_t* in, int W, int64_t chunk, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Stage-2 video-text contrastive logits + symmetric soft-target cross entropy
 * (multi_modality/models/criterions.py:15-103): v (n,C), t (n,C) fp32 (already all-gathered), idx int64 or NULL. */
/* ABI 2: out[i][j] = alpha * sum_k A[i][k] B[j][k] (fp32, row-major [ni][K] x [nj][K] -> [ni][nj]): the similarity logits of FRAME-LEVEL
 * (3-D) features, criterions.py:31-50 `einsum("mld,nd->mln")`, and the two products of their backward. */
int ivh_vtc_abt(const float* A, const float* B, int ni, int nj, int K, float alpha, float* out, void* stream);
int64_t ivh_vtc_workspace_floats(int n, int C);
/* sim (n,n), loss (1), dtemp (1 or NULL) fp32 outputs; dv, dt (n,C) or both NULL (forward only); ws: workspace */
int ivh_vtc_loss_fwd_bwd(const float* v, const float* t, const int64_t* idx, int n, int C, float temp,
                         float* sim, float* loss, float* dv, float* dt, float* dtemp, float* ws, void* stream);
/* the same with the temperature read from HBM (the learnable, clamped `temp` parameter: no host read, capturable into a HIP graph);
 * the caller keeps it positive (internvideo2_stage2_visual.py:291-294 clamps it to [0.001, 0.5] every step) */
int ivh_vtc_loss_fwd_bwd_dev(const float* v, const float* t, const int64_t* idx, int n, int C, const float* temp_dev,
                             float* sim, float* loss, float* dv, float* dt, float* dtemp, float* ws, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Stage-2 text / fusion tower (post-LN BERT, multi_modality/models/backbones/bert/xbert.py) row kernels.
 * ivh_bert_embed_fwd: y = LayerNorm((word[ids] + type[0]) + pos[row % L]) (BertEmbeddings.forward, xbert.py:298-334; token_type_ids are all
 *   zero on this path, xbert.py:1214-1215; dropout p = 0).  ids int32 [M] (M = B*L), tables fp32 [.][C], y bf16 [M][C], stats fp32 [M][2].
 * ivh_bert_embed_bwd: LayerNorm backward at the recomputed rows; the row gradients are added (fp32 atomics; the three outputs must be
 *   zeroed or hold a running sum) to dword[ids] (rows with ids == pad_id skipped: nn.Embedding(padding_idx), xbert.py:277-279),
 *   dpos[row % L] and dtype[0]; d{w,b}_part [ivh_norm_bwd_parts(M)][C] partial column sums (ivh_colsum_finish).
 * ivh_add_layernorm_fwd: y = LayerNorm(a + r) of BertSelfOutput / BertOutput (xbert.py:508-512, 592-596) and the MLM head transform
 *   (r = NULL, act = 1: y = LayerNorm(gelu_erf(a)), xbert.py:839-843); a, r, y bf16 [M][C], the sum and the statistics in fp32.
 * ivh_add_layernorm_bwd: dx = bf16(LayerNorm'(dy + dy2)) at the recomputed a + r (dy2 may be NULL): the gradient of both addends.
 * Hidden dropout (xbert.py:288,331 on the embedding output; :506,510 / :590,594 on the dense output before the residual add) is part of
 * these kernels: drop_p in [0, 1), mask = hash(seed, row * C + column) (counter based, common.h: the backward regenerates it from the same
 * (drop_p, seed)); embed: y = dropout(LayerNorm(..)); add_layernorm: y = LayerNorm(dropout(a) + r), and the backward writes dx (gradient
 * of r) and dx_a (gradient of a = dx with the mask; non-NULL exactly when drop_p > 0). */
int ivh_bert_embed_fwd(const int* ids, int M, int L, const float* word, const float* pos, const float* type, const float* w,
                       const float* b, float eps, int C, uint16_t* y, float* stats, float drop_p, uint32_t seed, void* stream);
int ivh_bert_embed_bwd(const int* ids, int M, int L, const float* word, const float* pos, const float* type, const float* w,
                       const float* stats, const uint16_t* dy, int C, int pad_id, float* dword, float* dpos, float* dtype,
                       float* dw_part, float* db_part, float drop_p, uint32_t seed, void* stream);
int ivh_add_layernorm_fwd(const uint16_t* a, const uint16_t* r, int act, const float* w, const float* b, float eps, int M, int C,
                          uint16_t* y, float* stats, float drop_p, uint32_t seed, void* stream);
int ivh_add_layernorm_bwd(const uint16_t* a, const uint16_t* r, int act, const float* w, const float* stats, const uint16_t* dy,
                          const uint16_t* dy2, int M, int C, uint16_t* dx, uint16_t* dx_a, float* dw_part, float* db_part,
                          float drop_p, uint32_t seed, void* stream);
/* Row-wise cross entropy with ignore_index, mean over the kept rows: nn.CrossEntropyLoss of the MLM head (xbert.py:1677-1682, V = 30522,
 * labels -100 off the masked tokens) and F.cross_entropy of the VTM head (criterions.py:177-181, V = 2).  logits bf16|fp32 [M][ld]
 * (columns V..ld-1 are padding and ignored), labels int32 [M].  inv_count[0] = 1 / #kept rows (device scalar, written here);
 * rows[m] = inv_count * (logsumexp - x[label]) or 0 (sum them: ivh_sum_rows); dlogits bf16 [M][ldd] = dscale * dscale_dev[0] * inv_count *
 * (softmax - onehot) (zero rows for ignored labels, zero padding columns) or NULL.  dscale_dev: device scalar (the upstream gradient of
 * the loss) or NULL = 1.  dlogits may alias bf16 logits (ldd == ld): each element is read before it is overwritten. */
int ivh_ce_rows(const void* logits, int logits_fp32, int ld, int M, int V, const int* labels, int ignore_index, float dscale,
                const float* dscale_dev, float* inv_count, float* rows, uint16_t* dlogits, int ldd, void* stream);

#ifdef __cplusplus
}
#endif
#endif
