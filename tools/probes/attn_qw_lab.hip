// 16x16x32 attention forward with QW 16-query blocks per wave and NW waves per workgroup (measurement aid, NOT part of the library): the forward
// kernel of flash_attn.hip (register-staged K / V tiles, two barriers per tile) re-parametrised so that every K fragment / transposed V fragment
// read from LDS feeds QW MFMAs.  Standalone: includes flash_attn.hip, compares every variant with its shipped kernel and times it.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -w -o /tmp/attn_qw_lab tools/probes/attn_qw_lab.hip && /tmp/attn_qw_lab
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

template <int HDP, int T>
__device__ __forceinline__ void lab_tile_load(const bf16_t* __restrict__ base, long sl, int row0, int nrows, int hd, u32x4* regs, int tid) {
  using C = AttnCfg<HDP>;
  constexpr int CPT = (64 * C::CPR + T - 1) / T;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int id = tid + T * i;
    const int r = id / C::CPR, cc = id % C::CPR;
    const int row = row0 + r;
    if (id < 64 * C::CPR && row < nrows && cc * 8 < hd) regs[i] = *reinterpret_cast<const u32x4*>(base + (long)row * sl + cc * 8);
    else regs[i] = u32x4{0u, 0u, 0u, 0u};
  }
}
template <int HDP, int T>
__device__ __forceinline__ void lab_tile_store(char* lds_tile, const u32x4* regs, int tid) {
  using C = AttnCfg<HDP>;
  constexpr int CPT = (64 * C::CPR + T - 1) / T;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int id = tid + T * i;
    const int r = id / C::CPR, cc = id % C::CPR;
    if (id < 64 * C::CPR) *reinterpret_cast<u32x4*>(lds_tile + r * C::RS + cc * 16) = regs[i];
  }
}

template <int HDP, int QW, int NW, int WPE>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(WPE))) void attn_fwd_qw_kernel(
    const bf16_t* __restrict__ q, long qsb, long qsl, long qsh, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, long sb, long sl, long sh,
    bf16_t* __restrict__ out, long ob, long ol, long oh, float* __restrict__ lse, int H, int Lq, int Lk, int hd, float scale) {
  using C = AttnCfg<HDP>;
  constexpr int T = NW * 64, CPT = (64 * C::CPR + T - 1) / T, QPW = NW * 16 * QW;
  __shared__ __attribute__((aligned(16))) char lds[2 * C::TILE];
  char* Kt = lds;
  char* Vt = lds + C::TILE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  const int ntq = (Lq + QPW - 1) / QPW;
  const int wid = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = wid / ntq;
  const int b = bh / H, h = bh - b * H, q0 = (wid - bh * ntq) * QPW;
  const bf16_t* qb = q + (long)b * qsb + (long)h * qsh;
  const bf16_t* kb = k + (long)b * sb + (long)h * sh;
  const bf16_t* vb = v + (long)b * sb + (long)h * sh;
  s16x8 qf[QW][C::KS];
  f32x4 o[QW][C::DT];
  float m[QW], l[QW];
  int qrow[QW];
#pragma unroll
  for (int w = 0; w < QW; ++w) {
    qrow[w] = q0 + (wave * QW + w) * 16 + (lane & 15);
    row_frags<HDP>(qb, qsl, qrow[w], Lq, hd, qf[w], lane);
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) o[w][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    m[w] = -INFINITY; l[w] = 0.f;
  }
  const float c2 = scale * LOG2E;
  u32x4 kr[CPT], vr[CPT];
  lab_tile_load<HDP, T>(kb, sl, 0, Lk, hd, kr, tid);
  lab_tile_load<HDP, T>(vb, sl, 0, Lk, hd, vr, tid);
  const int nt = (Lk + 63) / 64;
  auto tile = [&](const int t, auto ragged_tag) __attribute__((always_inline)) {
    constexpr bool RAGGED = decltype(ragged_tag)::value;
    __syncthreads();
    lab_tile_store<HDP, T>(Kt, kr, tid);
    lab_tile_store<HDP, T>(Vt, vr, tid);
    __syncthreads();
    if (!RAGGED) {
      lab_tile_load<HDP, T>(kb, sl, (t + 1) * 64, Lk, hd, kr, tid);
      lab_tile_load<HDP, T>(vb, sl, (t + 1) * 64, Lk, hd, vr, tid);
    }
    f32x4 s[QW][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int w = 0; w < QW; ++w) s[w][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        const s16x8 kfrag = frag_rows<HDP>(Kt, 16 * j, ks, lane);
#pragma unroll
        for (int w = 0; w < QW; ++w) s[w][j] = mfma16(kfrag, qf[w][ks], s[w][j]);
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
      mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      const float mn = fmaxf(m[w], mt * c2);
      const float alpha = fast_exp2(m[w] - mn);
      m[w] = mn;
      float ps = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s[w][j][r] = fast_exp2(fmaf(s[w][j][r], c2, -mn)); ps += s[w][j][r]; }
      l[w] = l[w] * alpha + ps;
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[w][dt][r] *= alpha;
      pf[w][0] = pack_frag(s[w][0], s[w][1]);
      pf[w][1] = pack_frag(s[w][2], s[w][3]);
    }
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt) {
        const s16x8 vfrag = frag_cols_tr<HDP>(Vt, dt, c, lane);
#pragma unroll
        for (int w = 0; w < QW; ++w) o[w][dt] = mfma16(vfrag, pf[w][c], o[w][dt]);
      }
  };
  for (int t = 0; t + 1 < nt; ++t) tile(t, std::false_type{});
  tile(nt - 1, std::true_type{});
#pragma unroll
  for (int w = 0; w < QW; ++w) {
    float lw = l[w];
    lw += __shfl_xor(lw, 16, 64);
    lw += __shfl_xor(lw, 32, 64);
    const float inv = 1.0f / lw;
    if (qrow[w] < Lq) {
      if (g == 0 && lse) lse[((long)b * H + h) * Lq + qrow[w]] = m[w] * LN2 + logf(lw);
      bf16_t* op = out + (long)b * ob + (long)qrow[w] * ol + (long)h * oh;
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt) {
        const int d = 16 * dt + 4 * g;
        if (d < hd) *reinterpret_cast<u32x2*>(op + d) = pack4(o[w][dt][0] * inv, o[w][dt][1] * inv, o[w][dt][2] * inv, o[w][dt][3] * inv);
      }
    }
  }
}

}  // namespace ivh

static float bf2f(uint16_t x) { unsigned u = (unsigned)x << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }

template <int QW, int NW, int WPE>
static void run_case(const char* name, const uint16_t* dq, uint16_t* o, float* lse, const std::vector<uint16_t>& ref, double den, int B, int H, int L, int hd) {
  const long D = (long)H * hd, qsl = 3 * D, qsb = (long)L * qsl, qsh = hd;
  const size_t n_out = (size_t)B * L * D;
  const int qpw = NW * 16 * QW;
  dim3 grid(B * H * ((L + qpw - 1) / qpw)), block(NW * 64);
  const float scale = 1.0f / sqrtf((float)hd);
  printf("{\"running\": \"%s\"}\n", name);
  auto fn = [&]() {
    ivh::attn_fwd_qw_kernel<96, QW, NW, WPE><<<grid, block, 0, 0>>>(dq, qsb, qsl, qsh, dq + D, dq + 2 * D, qsb, qsl, qsh, o, (long)L * D, D, (long)hd, lse, H, L, L, hd, scale);
  };
  hipMemset(o, 0, n_out * 2);
  fn();
  const hipError_t e1 = hipDeviceSynchronize();
  std::vector<uint16_t> got(n_out);
  hipMemcpy(got.data(), o, n_out * 2, hipMemcpyDeviceToHost);
  double nn = 0, mm = 0;
  for (size_t i = 0; i < n_out; ++i) { const double x = bf2f(ref[i]), y = bf2f(got[i]); nn += (x - y) * (x - y); mm = fmax(mm, fabs(x - y)); }
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
  const double us = ms / 40 * 1e3;
  printf("{\"variant\": \"%s\", \"rel_l2_vs_shipped\": %.3e, \"max_abs\": %.3e, \"us\": %.1f, \"tflops\": %.1f, \"sync_error\": %d}\n", name, sqrt(nn / fmax(den, 1e-30)), mm, us,
         4.0 * B * H * (double)L * L * hd / us / 1e6, (int)e1);
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
  float* lse;
  hipMalloc(&dq, n_qkv * 2); hipMalloc(&o_ref, n_out * 2); hipMalloc(&o_lab, n_out * 2); hipMalloc(&lse, (size_t)B * H * L * 4);
  hipMemcpy(dq, hq.data(), n_qkv * 2, hipMemcpyHostToDevice);
  const float scale = 1.0f / sqrtf((float)hd);
  ivh::attn_fwd_kernel<96, false><<<dim3(B * H * ((L + 63) / 64)), dim3(256), 0, 0>>>(dq, qsb, qsl, qsh, dq + D, dq + 2 * D, qsb, qsl, qsh, o_ref, (long)L * D, D, (long)hd, lse, H, L,
                                                                                     L, hd, scale, (const int32_t*)nullptr, ivh::DropCfg{0u, 1.0f, 0u});
  printf("{\"shipped_16x16_sync\": %d}\n", (int)hipDeviceSynchronize());
  std::vector<uint16_t> ref(n_out);
  hipMemcpy(ref.data(), o_ref, n_out * 2, hipMemcpyDeviceToHost);
  double den = 0;
  for (size_t i = 0; i < n_out; ++i) { const double x = bf2f(ref[i]); den += x * x; }
  run_case<1, 4, 4>("QW=1 NW=4", dq, o_lab, lse, ref, den, B, H, L, hd);
  run_case<2, 4, 2>("QW=2 NW=4", dq, o_lab, lse, ref, den, B, H, L, hd);
  run_case<3, 3, 2>("QW=3 NW=3", dq, o_lab, lse, ref, den, B, H, L, hd);
  run_case<3, 2, 2>("QW=3 NW=2", dq, o_lab, lse, ref, den, B, H, L, hd);
  run_case<4, 2, 1>("QW=4 NW=2 (1 wave / SIMD)", dq, o_lab, lse, ref, den, B, H, L, hd);
  return 0;
}
