// Stage-2 video<->text contrastive logits + symmetric soft-target cross entropy, forward and backward
// (reference InternVideo2/multi_modality/models/criterions.py:15-103, 200-216).
// n = B * world <= a few hundred rows of 512 features: latency-bound, fp32 throughout, everything stays in HBM/L2.
//   vn = v / max(|v|, 1e-12), tn likewise (F.normalize);  sim = vn tn^T / temp;  T = eq(idx, idx^T) / rowsum
//   loss = 1/(2n) sum_i sum_j T_ij [ (lse_row_i - sim_ij) + (lse_col_i - sim_ji) ]
#include "common.h"
#include "../../include/internvideo_hip.h"

namespace ivh {

__global__ __launch_bounds__(256) void vtc_normalize_kernel(const float* __restrict__ x, int n, int C, float* __restrict__ xn,
                                                            float* __restrict__ inv) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= n) return;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) { const float a = x[(long)row * C + c]; s += a * a; }
  s = wave_sum(s);
  const float iv = 1.0f / fmaxf(sqrtf(s), 1e-12f);
  if (lane == 0) inv[row] = iv;
  for (int c = lane; c < C; c += 64) xn[(long)row * C + c] = x[(long)row * C + c] * iv;
}

// out[i][j] = alpha * sum_k A[i][k] * B[j][k]   (A: [ni][K], B: [nj][K])
// `temp_dev` (device scalar, may be NULL) replaces the host value of the temperature: alpha = 1 / temp_dev[0] (graph-captured steps read the
// learnable temperature from HBM instead of baking its value into the launch arguments)
__global__ __launch_bounds__(256) void vtc_abt_kernel(const float* __restrict__ A, const float* __restrict__ B, int ni, int nj, int K,
                                                      float alpha, const float* __restrict__ temp_dev, float* __restrict__ out) {
  if (temp_dev) alpha = 1.0f / temp_dev[0];
  __shared__ float sa[16][17], sb[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int i = blockIdx.y * 16 + ty, j = blockIdx.x * 16 + tx;
  float acc = 0.f;
  for (int k0 = 0; k0 < K; k0 += 16) {
    const int ai = blockIdx.y * 16 + ty, bj = blockIdx.x * 16 + ty;
    sa[ty][tx] = (ai < ni && k0 + tx < K) ? A[(long)ai * K + k0 + tx] : 0.f;
    sb[ty][tx] = (bj < nj && k0 + tx < K) ? B[(long)bj * K + k0 + tx] : 0.f;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) acc += sa[ty][k] * sb[tx][k];
    __syncthreads();
  }
  if (i < ni && j < nj) out[(long)i * nj + j] = acc * alpha;
}

// out[i][c] = alpha * sum_j S(i,j) * X[j][c], S(i,j) = trans ? S[j][i] : S[i][j]     (S: n x n, X: n x C)
__global__ __launch_bounds__(256) void vtc_sx_kernel(const float* __restrict__ S, const float* __restrict__ X, int n, int C, int trans,
                                                     float alpha, const float* __restrict__ temp_dev, float* __restrict__ out) {
  if (temp_dev) alpha = 1.0f / temp_dev[0];
  const long id = (long)blockIdx.x * 256 + threadIdx.x;
  if (id >= (long)n * C) return;
  const int c = id % C, i = id / C;
  float acc = 0.f;
  for (int j = 0; j < n; ++j) acc += (trans ? S[(long)j * n + i] : S[(long)i * n + j]) * X[(long)j * C + c];
  out[id] = acc * alpha;
}

// row (which = 0) / column (which = 1) log-sum-exp of sim
__global__ __launch_bounds__(256) void vtc_lse_kernel(const float* __restrict__ sim, int n, float* __restrict__ lse_row, float* __restrict__ lse_col) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int id = blockIdx.x * 4 + wave;
  if (id >= 2 * n) return;
  const int which = id / n, r = id % n;
  float mx = -INFINITY;
  for (int j = lane; j < n; j += 64) mx = fmaxf(mx, which ? sim[(long)j * n + r] : sim[(long)r * n + j]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int j = lane; j < n; j += 64) s += __expf((which ? sim[(long)j * n + r] : sim[(long)r * n + j]) - mx);
  s = wave_sum(s);
  if (lane == 0) (which ? lse_col : lse_row)[r] = mx + __logf(s);
}

// per-row loss term and dsim row; one wave per row i.  T_ij = eq(idx_i, idx_j) / cnt_i (identity if idx == NULL)
__global__ __launch_bounds__(256) void vtc_loss_dsim_kernel(const float* __restrict__ sim, const long long* __restrict__ idx, int n,
                                                            const float* __restrict__ lse_row, const float* __restrict__ lse_col,
                                                            float* __restrict__ loss_rows, float* __restrict__ dsim, float* __restrict__ dtemp_rows,
                                                            float temp, const float* __restrict__ temp_dev) {
  if (temp_dev) temp = temp_dev[0];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 4 + wave;
  if (i >= n) return;
  float cnt = 0.f;
  if (idx) {
    const long long me = idx[i];
    for (int j = lane; j < n; j += 64) cnt += idx[j] == me ? 1.f : 0.f;
    cnt = wave_sum(cnt);
  } else cnt = 1.f;
  const float ri = lse_row[i], ci = lse_col[i];
  float li = 0.f, dt = 0.f;
  const float h = 0.5f / (float)n;
  for (int j = lane; j < n; j += 64) {
    const bool same = idx ? (idx[j] == idx[i]) : (j == i);
    const float T = same ? 1.f / cnt : 0.f;
    const float sij = sim[(long)i * n + j], sji = sim[(long)j * n + i];
    li += T * ((ri - sij) + (ci - sji));
    // d/dsim_ij of L1 (row softmax of row i) and of L2 (column softmax of column j; T_ji = T_ij by symmetry)
    const float g = h * ((__expf(sij - ri) - T) + (__expf(sij - lse_col[j]) - T));
    dsim[(long)i * n + j] = g;
    dt += g * sij;
  }
  li = wave_sum(li);
  dt = wave_sum(dt);
  if (lane == 0) { loss_rows[i] = li * h; dtemp_rows[i] = -dt / temp; }
}

// dx = (dxn - xn <xn, dxn>) * inv
__global__ __launch_bounds__(256) void vtc_normalize_bwd_kernel(const float* __restrict__ xn, const float* __restrict__ dxn,
                                                                const float* __restrict__ inv, int n, int C, float* __restrict__ dx) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= n) return;
  float d = 0.f;
  for (int c = lane; c < C; c += 64) d += xn[(long)row * C + c] * dxn[(long)row * C + c];
  d = wave_sum(d);
  const float iv = inv[row];
  for (int c = lane; c < C; c += 64) dx[(long)row * C + c] = (dxn[(long)row * C + c] - xn[(long)row * C + c] * d) * iv;
}

__global__ __launch_bounds__(256) void vtc_reduce2_kernel(const float* __restrict__ a, const float* __restrict__ b, int n,
                                                          float* __restrict__ oa, float* __restrict__ ob) {
  __shared__ float ra[256], rb[256];
  float sa = 0.f, sb = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) { sa += a[i]; sb += b[i]; }
  ra[threadIdx.x] = sa; rb[threadIdx.x] = sb;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { ra[threadIdx.x] += ra[threadIdx.x + o]; rb[threadIdx.x] += rb[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { oa[0] = ra[0]; if (ob) ob[0] = rb[0]; }
}

}  // namespace ivh

using namespace ivh;

// out[i][j] = alpha * sum_k A[i][k] B[j][k], fp32, any ni / nj / K: the frame-level (3-D) similarity logits of criterions.py:31-50 and the
// two products of their backward (the caller passes transposed copies; these are (B L) x B x 512 problems, latency-bound)
extern "C" int ivh_vtc_abt(const float* A, const float* B, int ni, int nj, int K, float alpha, float* out, void* stream) {
  IVH_REQUIRE(A && B && out && ni > 0 && nj > 0 && K > 0, "vtc_abt: bad args");
  hipLaunchKernelGGL(ivh::vtc_abt_kernel, dim3((nj + 15) / 16, (ni + 15) / 16), dim3(256), 0, (hipStream_t)stream, A, B, ni, nj, K, alpha,
                     (const float*)nullptr, out);
  return ivh_host::check_launch("vtc_abt");
}

extern "C" int64_t ivh_vtc_workspace_floats(int n, int C) { return (int64_t)4 * n * C + (int64_t)n * n + 6 * (int64_t)n; }

static int vtc_launch(const float* v, const float* t, const int64_t* idx, int n, int C, float temp, const float* temp_dev,
                      float* sim, float* loss, float* dv, float* dt, float* dtemp, float* ws, void* stream) {
  IVH_REQUIRE(v && t && sim && loss && ws && n > 0 && C > 0, "vtc_loss: bad args");
  IVH_REQUIRE((dv == nullptr) == (dt == nullptr), "vtc_loss: dv and dt must be given together");
  hipStream_t s = (hipStream_t)stream;
  float* vn = ws; float* tn = vn + (long)n * C; float* dvn = tn + (long)n * C; float* dtn = dvn + (long)n * C;
  float* dsim = dtn + (long)n * C;
  float* invv = dsim + (long)n * n; float* invt = invv + n; float* lr = invt + n; float* lc = lr + n;
  float* lrows = lc + n; float* trows = lrows + n;
  const dim3 rows((n + 3) / 4), blk(256);
  hipLaunchKernelGGL(vtc_normalize_kernel, rows, blk, 0, s, v, n, C, vn, invv);
  hipLaunchKernelGGL(vtc_normalize_kernel, rows, blk, 0, s, t, n, C, tn, invt);
  hipLaunchKernelGGL(vtc_abt_kernel, dim3((n + 15) / 16, (n + 15) / 16), blk, 0, s, vn, tn, n, n, C, 1.0f / temp, temp_dev, sim);
  hipLaunchKernelGGL(vtc_lse_kernel, dim3((2 * n + 3) / 4), blk, 0, s, sim, n, lr, lc);
  hipLaunchKernelGGL(vtc_loss_dsim_kernel, rows, blk, 0, s, sim, (const long long*)idx, n, lr, lc, lrows, dsim, trows, temp, temp_dev);
  hipLaunchKernelGGL(vtc_reduce2_kernel, dim3(1), blk, 0, s, lrows, trows, n, loss, dtemp);
  if (dv) {
    const long tot = (long)n * C;
    hipLaunchKernelGGL(vtc_sx_kernel, dim3((unsigned)((tot + 255) / 256)), blk, 0, s, dsim, tn, n, C, 0, 1.0f / temp, temp_dev, dvn);
    hipLaunchKernelGGL(vtc_sx_kernel, dim3((unsigned)((tot + 255) / 256)), blk, 0, s, dsim, vn, n, C, 1, 1.0f / temp, temp_dev, dtn);
    hipLaunchKernelGGL(vtc_normalize_bwd_kernel, rows, blk, 0, s, vn, dvn, invv, n, C, dv);
    hipLaunchKernelGGL(vtc_normalize_bwd_kernel, rows, blk, 0, s, tn, dtn, invt, n, C, dt);
  }
  return ivh_host::check_launch("vtc_loss_fwd_bwd");
}

extern "C" int ivh_vtc_loss_fwd_bwd(const float* v, const float* t, const int64_t* idx, int n, int C, float temp,
                                    float* sim, float* loss, float* dv, float* dt, float* dtemp, float* ws, void* stream) {
  IVH_REQUIRE(temp > 0.f, "vtc_loss: temperature must be positive (clamp it to [0.001, 0.5] first)");
  return vtc_launch(v, t, idx, n, C, temp, nullptr, sim, loss, dv, dt, dtemp, ws, stream);
}

extern "C" int ivh_vtc_loss_fwd_bwd_dev(const float* v, const float* t, const int64_t* idx, int n, int C, const float* temp_dev,
                                        float* sim, float* loss, float* dv, float* dt, float* dtemp, float* ws, void* stream) {
  IVH_REQUIRE(temp_dev, "vtc_loss: null temperature");
  return vtc_launch(v, t, idx, n, C, 1.0f, temp_dev, sim, loss, dv, dt, dtemp, ws, stream);
}
