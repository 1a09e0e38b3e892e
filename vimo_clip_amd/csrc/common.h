// Shared device helpers for libvmc (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vmc.h"
#include "tile_index.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define VMC_LDS __attribute__((address_space(3)))
#define VMC_GLOBAL __attribute__((address_space(1)))

#define VMC_CHECK_LAUNCH()                          \
  do {                                              \
    hipError_t e_ = hipGetLastError();              \
    if (e_ != hipSuccess) return (int)e_;           \
  } while (0)

// ---- 16-bit element traits ---------------------------------------------------------------------
struct BF16 {
  static constexpr int id = VMC_BF16;
  __device__ static inline float to_f32(uint16_t u) { return __uint_as_float(((uint32_t)u) << 16); }
  __device__ static inline uint16_t from_f32(float f) {
    __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    return __builtin_bit_cast(uint16_t, b);
  }
  __device__ static inline uint32_t pack(float lo, float hi) {  // one v_cvt_pk_bf16_f32
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){lo, hi}, bf16x2));
  }
  static constexpr uint32_t ONE_PAIR = 0x3F803F80u;  // (1.0, 1.0)
  __device__ static inline f32x4 mfma16(const uint4& a, const uint4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
  __device__ static inline f32x16 mfma32(const uint4& a, const uint4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
struct F16 {
  static constexpr int id = VMC_F16;
  __device__ static inline float to_f32(uint16_t u) { return (float)__builtin_bit_cast(_Float16, u); }
  __device__ static inline uint16_t from_f32(float f) {
    _Float16 h = (_Float16)f;
    return __builtin_bit_cast(uint16_t, h);
  }
  __device__ static inline uint32_t pack(float lo, float hi) {  // one v_cvt_pk_f16_f32
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){lo, hi}, f16x2));
  }
  static constexpr uint32_t ONE_PAIR = 0x3C003C00u;  // (1.0, 1.0)
  __device__ static inline f32x4 mfma16(const uint4& a, const uint4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
  __device__ static inline f32x16 mfma32(const uint4& a, const uint4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};

template <typename T>
__device__ inline uint32_t pack2(float lo, float hi) {
  return T::pack(lo, hi);
}
template <typename T>
__device__ inline void unpack2(uint32_t w, float& lo, float& hi) {
  lo = T::to_f32((uint16_t)(w & 0xFFFFu));
  hi = T::to_f32((uint16_t)(w >> 16));
}

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

template <int ACT>
__device__ inline float apply_act(float x) {
  // x * sigmoid(1.702 x) with one v_exp_f32 and one v_rcp_f32 (1 ulp each; the result is rounded to 16 bits anyway)
  if (ACT == VMC_ACT_QUICKGELU) return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930156f * x));
  if (ACT == VMC_ACT_GELU_ERF) return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
  if (ACT == VMC_ACT_RELU) return fmaxf(x, 0.0f);
  return x;
}
__device__ inline float apply_act_rt(float x, int act) {
  switch (act) {
    case VMC_ACT_QUICKGELU: return apply_act<VMC_ACT_QUICKGELU>(x);
    case VMC_ACT_GELU_ERF: return apply_act<VMC_ACT_GELU_ERF>(x);
    case VMC_ACT_RELU: return apply_act<VMC_ACT_RELU>(x);
    default: return x;
  }
}
__device__ inline float act_grad_rt(float x, int act) {
  switch (act) {
    case VMC_ACT_QUICKGELU: {
      float s = 1.0f / (1.0f + __expf(-1.702f * x));
      return s * (1.0f + 1.702f * x * (1.0f - s));
    }
    case VMC_ACT_GELU_ERF: {
      float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
      float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
      return cdf + x * pdf;
    }
    case VMC_ACT_RELU: return x > 0.0f ? 1.0f : 0.0f;
    default: return 1.0f;
  }
}

// Counter-based keep mask for dropout: element i of stream `seed` is kept iff hash32 >= p * 2^32.
__device__ inline uint32_t hash32(uint64_t seed, uint64_t i) {
  uint64_t x = (i + 0x9E3779B97F4A7C15ull) ^ seed;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  x ^= x >> 31;
  return (uint32_t)(x >> 32);
}
// Seed arguments of the ABI: a plain 63-bit value, or -- bit 63 set -- the address (low 63 bits) of a uint64 in device memory
// that holds the seed.  A hipGraph-captured training step passes addresses: the captured launches re-read the seeds that
// vmc_train_tick rewrites before every replay (host scalars would be frozen into the graph).
#define VMC_SEED_IS_PTR (1ull << 63)
__device__ inline uint64_t resolve_seed(uint64_t s) {
  return (s & VMC_SEED_IS_PTR) ? *(const uint64_t*)(uintptr_t)(s & ~VMC_SEED_IS_PTR) : s;
}
__device__ inline float dropout_factor(float p, uint64_t seed, uint64_t i) {
  if (p <= 0.f) return 1.0f;
  return hash32(seed, i) >= (uint32_t)((double)p * 4294967296.0) ? 1.0f / (1.0f - p) : 0.0f;
}

// One Adam / AdamW element update (torch.optim.Adam / AdamW, no amsgrad): every optimiser kernel goes through this expression, so
// the flat kernel and the tile kernel that also refreshes the 16-bit copies agree bit for bit.
__device__ __forceinline__ void adam_element(float& p, float g, float& m, float& v, float lr, float b1, float b2, float eps, float wd,
                                             int decoupled, float step_size, float inv_sqrt_bc2, float gscale) {
#pragma clang fp contract(off)      // which products fuse into FMAs must not depend on the kernel this is inlined into
  float gr = g * gscale;
  if (decoupled) p *= (1.0f - lr * wd); else gr += wd * p;
  m = b1 * m + (1.0f - b1) * gr;
  v = b2 * v + (1.0f - b2) * gr * gr;
  p -= step_size * m / (sqrtf(v) * inv_sqrt_bc2 + eps);
}

static inline int grid_for(size_t n, int block, int max_blocks = 256 * 16) {
  size_t g = (n + block - 1) / block;
  if (g > (size_t)max_blocks) g = max_blocks;
  if (g < 1) g = 1;
  return (int)g;
}
