// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the InternVideo2 hot path.
// gfx950 only: 64-lane wavefronts, MFMA 16x16x32 bf16, LDS-DMA (global_load_lds 16 B), ds_read_b64_tr_b16.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ivh {

typedef unsigned short bf16_t;  // storage type for bf16 in HBM / LDS
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define IVH_LDS __attribute__((address_space(3)))
#define IVH_GLOBAL __attribute__((address_space(1)))

constexpr int WAVE = 64;

// 16 bytes of zeros in device memory: the source for predicated-off LDS-DMA lanes (K tails, rows past
// the end, head-dim padding).  Every lane may read the same 16 bytes.
static __device__ __attribute__((aligned(16))) unsigned int g_zero_page[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {   // round-to-nearest-even (v_cvt_pk_bf16_f32)
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ u32x2 pack4(float a, float b, float c, float d) {
  f32x4 v = {a, b, c, d};
  bf16x4 r = __builtin_convertvector(v, bf16x4);
  return __builtin_bit_cast(u32x2, r);
}
__device__ __forceinline__ unsigned pack2(float a, float b) {
  f32x2 v = {a, b};
  bf16x2 r = __builtin_convertvector(v, bf16x2);
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ void unpack8(u32x4 v, float* o) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[2 * i] = __uint_as_float(v[i] << 16);
    o[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ u32x4 pack8(const float* f) {
  u32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = pack2(f[2 * i], f[2 * i + 1]);
  return r;
}

// LDS-DMA: 16 bytes per lane, LDS destination = wave-uniform base + lane*16 (hardware rule).
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const IVH_GLOBAL void*)gsrc, (IVH_LDS void*)lds_wave_base, 16, 0, 0);
}
// LDS transpose read: within each 16-lane group lane i points at 4 consecutive 16-bit elements
// (row i>>2, cols 4*(i&3)..+3 of a 4x16 block); it receives column i of that block: rows 0..3.
__device__ __forceinline__ s16x4 lds_tr16(const void* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((IVH_LDS s16x4*)p);
}
__device__ __forceinline__ f32x4 mfma16(s16x8 a, s16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_erf(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
__device__ __forceinline__ float gelu_tanh(float x) {
  const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  return 0.5f * x * (1.0f + tanhf(u));
}
__device__ __forceinline__ float dgelu_tanh(float x) {
  const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  const float t = tanhf(u);
  const float du = 0.7978845608028654f * (1.0f + 3.0f * 0.044715f * x * x);
  return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * du;
}

// Counter-based dropout mask: keep(element) = hash(seed, 64-bit element index) >= p * 2^32.  Stateless, so forward and backward kernels
// (and a host-side check, tests/test_bert_gpu.py::dropout_keep_mask) regenerate the same mask from (seed, index) alone.
// hash = two rounds of the lowbias32 integer finaliser over the low / high index words.
__device__ __forceinline__ unsigned ivh_hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
// `epoch` (device pointer or NULL, ivh_set_dropout_epoch): a step counter that lives in HBM.  A launch argument is frozen into a captured
// HIP graph, so a replayed step would draw the SAME masks every time; the effective seed is seed + *epoch * 0x9E3779B1 and the training
// engine advances *epoch with a kernel inside the captured step (forward and backward of one step read the same value).
struct DropCfg { unsigned thresh; float inv_keep; unsigned seed; const unsigned* epoch; };   // thresh = p * 2^32 (0 = no dropout), inv_keep = 1 / (1 - p)
__device__ __forceinline__ float drop_scale(const DropCfg& d, unsigned long long idx) {
  const unsigned seed = d.epoch ? d.seed + d.epoch[0] * 0x9E3779B1U : d.seed;
  const unsigned h = ivh_hash32(ivh_hash32((unsigned)idx ^ seed) ^ (unsigned)(idx >> 32) ^ 0x9e3779b9U);
  return h >= d.thresh ? d.inv_keep : 0.0f;
}

// XCD-aware, bijective remap of a linear workgroup id (block b runs on XCD b % 8): every XCD gets a
// contiguous range of the remapped ids so that neighbouring tiles share an L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, local = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + local;
}

}  // namespace ivh

// ---- host side error plumbing (C-ABI functions return 0 on success, <0 on error) ------------------
namespace ivh_host {
void set_error(const char* fmt, ...);
int check_launch(const char* what);
const unsigned* dropout_epoch();          // device pointer registered with ivh_set_dropout_epoch, or NULL
}  // namespace ivh_host

#define IVH_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      ivh_host::set_error(__VA_ARGS__);   \
      return -1;                          \
    }                                     \
  } while (0)
