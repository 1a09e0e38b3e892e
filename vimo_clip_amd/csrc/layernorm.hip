// LayerNorm forward / backward, one wave per row, fp32 statistics (gfx950).  HBM-bound: 16-byte loads.
#include "common.h"

#define LN_MAX_CHUNKS 16  // D <= 16 * 256 = 4096

template <typename T>
__device__ inline float4 load4(const void* base, size_t idx, bool is_f32) {
  if (is_f32) return *(const float4*)((const float*)base + idx);
  const uint2 w = *(const uint2*)((const uint16_t*)base + idx);
  float4 o;
  unpack2<T>(w.x, o.x, o.y);
  unpack2<T>(w.y, o.z, o.w);
  return o;
}
template <typename T>
__device__ inline void store4(void* base, size_t idx, bool is_f32, float4 v) {
  if (is_f32)
    *(float4*)((float*)base + idx) = v;
  else
    *(uint2*)((uint16_t*)base + idx) = make_uint2(pack2<T>(v.x, v.y), pack2<T>(v.z, v.w));
}

// NCH = float4 chunks per lane, a template constant (D <= 256 NCH): register arrays of the size the row needs, gamma / beta
// held in registers across rows, and the next row of the wave fetched while this one is reduced and written.
template <typename T, int NCH>
__global__ void __launch_bounds__(256) ln_fwd_kernel(const void* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, uint16_t* __restrict__ y16,
                                                     float* __restrict__ y32, float* __restrict__ mean_out,
                                                     float* __restrict__ rstd_out, int rows, int D, size_t ldx, float eps,
                                                     int x_f32) {
  const int lane = threadIdx.x & 63;
  const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * 4;
  const float invD = 1.0f / (float)D;
  float4 g[NCH], b[NCH], nv[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = c * 256 + lane * 4;
    g[c] = col < D ? *(const float4*)(gamma + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    b[c] = col < D ? *(const float4*)(beta + col) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  auto fetch = [&](int row) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 256 + lane * 4;
      if (col < D) nv[c] = load4<T>(x, (size_t)row * ldx + col, x_f32);
    }
  };
  if (wave_global < rows) fetch(wave_global);
  for (int row = wave_global; row < rows; row += nwaves) {
    float4 v[NCH];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 256 + lane * 4;
      if (col < D) {
        v[c] = nv[c];
        s += (v[c].x + v[c].y) + (v[c].z + v[c].w);
      }
    }
    if (row + nwaves < rows) fetch(row + nwaves);
    const float mean = wave_sum(s) * invD;
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 256 + lane * 4;
      if (col < D) {
        const float a = v[c].x - mean, bb = v[c].y - mean, cc = v[c].z - mean, d = v[c].w - mean;
        ss += (a * a + bb * bb) + (cc * cc + d * d);
      }
    }
    const float rstd = rsqrtf(wave_sum(ss) * invD + eps);
    if (lane == 0) {
      if (mean_out) mean_out[row] = mean;
      if (rstd_out) rstd_out[row] = rstd;
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 256 + lane * 4;
      if (col < D) {
        float4 o;
        o.x = (v[c].x - mean) * rstd * g[c].x + b[c].x;
        o.y = (v[c].y - mean) * rstd * g[c].y + b[c].y;
        o.z = (v[c].z - mean) * rstd * g[c].z + b[c].z;
        o.w = (v[c].w - mean) * rstd * g[c].w + b[c].w;
        if (y16) store4<T>(y16, (size_t)row * D + col, false, o);
        if (y32) store4<T>(y32, (size_t)row * D + col, true, o);
      }
    }
  }
}

extern "C" int vmc_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y16, float* y32, float* mean,
                                 float* rstd, int rows, int D, int ldx, float eps, int x_dtype, int dtype16, void* stream) {
  if (!x || !gamma || !beta || (!y16 && !y32) || rows <= 0 || D <= 0) return VMC_E_ARG;
  if (D % 4 || D > LN_MAX_CHUNKS * 256) return VMC_E_SHAPE;
  if (ldx % 4 || ldx < D) return VMC_E_ALIGN;
  if (x_dtype != VMC_F32 && x_dtype != dtype16) return VMC_E_DTYPE;
  const int grid = grid_for((size_t)rows, 4, 256 * 8);
  if (dtype16 != VMC_BF16 && dtype16 != VMC_F16) return VMC_E_DTYPE;
#define VMC_LN_FWD(NCH)                                                                                                     \
  do {                                                                                                                      \
    if (dtype16 == VMC_BF16)                                                                                                \
      hipLaunchKernelGGL((ln_fwd_kernel<BF16, NCH>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, gamma, beta,          \
                         (uint16_t*)y16, y32, mean, rstd, rows, D, (size_t)ldx, eps, x_dtype == VMC_F32);                    \
    else                                                                                                                    \
      hipLaunchKernelGGL((ln_fwd_kernel<F16, NCH>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, gamma, beta,           \
                         (uint16_t*)y16, y32, mean, rstd, rows, D, (size_t)ldx, eps, x_dtype == VMC_F32);                    \
  } while (0)
  if (D <= 512) VMC_LN_FWD(2);
  else if (D <= 768) VMC_LN_FWD(3);
  else if (D <= 1024) VMC_LN_FWD(4);
  else if (D <= 2048) VMC_LN_FWD(8);
  else VMC_LN_FWD(LN_MAX_CHUNKS);
#undef VMC_LN_FWD
  VMC_CHECK_LAUNCH();
  return 0;
}

// ---- fused residual add + LayerNorm ---------------------------------------------------------------
// x <- x + branch (fp32 residual stream updated in place), y16 = LN(x).  Keeps the residual update out of
// the GEMM epilogues (where 2 x 256 KB per tile of fp32 traffic is serialised behind a 1-block-per-CU
// main loop) and puts it in this HBM-streaming kernel instead.  CH = D / 256 chunks per lane, compile time.
template <typename T, int CH>
__global__ void __launch_bounds__(256) add_ln_kernel(float* __restrict__ x, const uint16_t* __restrict__ branch0,
                                                     const uint16_t* __restrict__ branch, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, uint16_t* __restrict__ y16, int rows, size_t ldx,
                                                     size_t ldb0, size_t ldb, float eps, int write_x) {
  constexpr int D = CH * 256;
  const int lane = threadIdx.x & 63;
  const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * 4;
  float4 g[CH], b[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    g[c] = *(const float4*)(gamma + c * 256 + lane * 4);
    b[c] = *(const float4*)(beta + c * 256 + lane * 4);
  }
  for (int row = wave_global; row < rows; row += nwaves) {
    float4 v[CH];
    uint2 br[CH], br0[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      {   // streaming (non-temporal): x is next touched by the following add+LN, ~1 GB of GEMM traffic later; keeping it out
          // of the caches leaves them to the 16-bit branch just written by the GEMM and to h, which the next GEMM re-reads
        const f32x4 t = __builtin_nontemporal_load((const f32x4*)(x + (size_t)row * ldx + c * 256 + lane * 4));
        v[c] = make_float4(t[0], t[1], t[2], t[3]);
      }
      br[c] = *(const uint2*)(branch + (size_t)row * ldb + c * 256 + lane * 4);
      if (branch0) br0[c] = *(const uint2*)(branch0 + (size_t)row * ldb0 + c * 256 + lane * 4);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      float a0, a1, a2, a3;
      if (branch0) {      // the add a previous pass skipped writing back: x = (x + branch0) + branch, in that order
        unpack2<T>(br0[c].x, a0, a1);
        unpack2<T>(br0[c].y, a2, a3);
        v[c].x += a0; v[c].y += a1; v[c].z += a2; v[c].w += a3;
      }
      unpack2<T>(br[c].x, a0, a1);
      unpack2<T>(br[c].y, a2, a3);
      v[c].x += a0; v[c].y += a1; v[c].z += a2; v[c].w += a3;
      s += (v[c].x + v[c].y) + (v[c].z + v[c].w);
      if (write_x) {
        const f32x4 t = {v[c].x, v[c].y, v[c].z, v[c].w};
        __builtin_nontemporal_store(t, (f32x4*)(x + (size_t)row * ldx + c * 256 + lane * 4));
      }
    }
    const float mean = wave_sum(s) * (1.0f / D);
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const float p = v[c].x - mean, q = v[c].y - mean, r = v[c].z - mean, t = v[c].w - mean;
      ss += (p * p + q * q) + (r * r + t * t);
    }
    const float rstd = rsqrtf(wave_sum(ss) * (1.0f / D) + eps);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const float o0 = (v[c].x - mean) * rstd * g[c].x + b[c].x, o1 = (v[c].y - mean) * rstd * g[c].y + b[c].y;
      const float o2 = (v[c].z - mean) * rstd * g[c].z + b[c].z, o3 = (v[c].w - mean) * rstd * g[c].w + b[c].w;
      *(uint2*)(y16 + (size_t)row * D + c * 256 + lane * 4) = make_uint2(pack2<T>(o0, o1), pack2<T>(o2, o3));
    }
  }
}

template <typename T>
static int launch_add_ln(float* x, const void* branch0, const void* branch, const float* gamma, const float* beta, void* y16, int rows,
                         int D, size_t ldx, size_t ldb0, size_t ldb, float eps, int write_x, hipStream_t s) {
  const int grid = grid_for((size_t)rows, 4, 256 * 8);
#define VMC_ADDLN(CHN)                                                                                                        \
  hipLaunchKernelGGL((add_ln_kernel<T, CHN>), dim3(grid), dim3(256), 0, s, x, (const uint16_t*)branch0, (const uint16_t*)branch, gamma, \
                     beta, (uint16_t*)y16, rows, ldx, ldb0, ldb, eps, write_x)
  switch (D / 256) {
    case 1: VMC_ADDLN(1); break;
    case 2: VMC_ADDLN(2); break;
    case 3: VMC_ADDLN(3); break;
    case 4: VMC_ADDLN(4); break;
    case 5: VMC_ADDLN(5); break;
    case 6: VMC_ADDLN(6); break;
    case 8: VMC_ADDLN(8); break;
    default: return VMC_E_SHAPE;
  }
#undef VMC_ADDLN
  VMC_CHECK_LAUNCH();
  return 0;
}

extern "C" int vmc_add_layernorm_fwd(float* x, const void* branch, const float* gamma, const float* beta, void* y16, int rows, int D,
                                     int ldx, int ldb, float eps, int write_x, int dtype16, void* stream) {
  if (!x || !branch || !gamma || !beta || !y16 || rows <= 0 || D <= 0) return VMC_E_ARG;
  if (D % 256 || D > 2048) return VMC_E_SHAPE;
  if (ldx % 4 || ldb % 4 || ldx < D || ldb < D) return VMC_E_ALIGN;
  if (dtype16 == VMC_BF16) return launch_add_ln<BF16>(x, nullptr, branch, gamma, beta, y16, rows, D, ldx, 0, ldb, eps, write_x, (hipStream_t)stream);
  if (dtype16 == VMC_F16) return launch_add_ln<F16>(x, nullptr, branch, gamma, beta, y16, rows, D, ldx, 0, ldb, eps, write_x, (hipStream_t)stream);
  return VMC_E_DTYPE;
}

extern "C" int vmc_add2_layernorm_fwd(float* x, const void* branch0, const void* branch, const float* gamma, const float* beta, void* y16,
                                      int rows, int D, int ldx, int ldb0, int ldb, float eps, int write_x, int dtype16, void* stream) {
  if (!x || !branch0 || !branch || !gamma || !beta || !y16 || rows <= 0 || D <= 0) return VMC_E_ARG;
  if (D % 256 || D > 2048) return VMC_E_SHAPE;
  if (ldx % 4 || ldb0 % 4 || ldb % 4 || ldx < D || ldb0 < D || ldb < D) return VMC_E_ALIGN;
  if (dtype16 == VMC_BF16)
    return launch_add_ln<BF16>(x, branch0, branch, gamma, beta, y16, rows, D, ldx, ldb0, ldb, eps, write_x, (hipStream_t)stream);
  if (dtype16 == VMC_F16)
    return launch_add_ln<F16>(x, branch0, branch, gamma, beta, y16, rows, D, ldx, ldb0, ldb, eps, write_x, (hipStream_t)stream);
  return VMC_E_DTYPE;
}

// ---- post-norm block tail: s = x + branch; y = LN(s) written as fp32 (next residual operand) and 16-bit (next GEMM
// operand) in one pass (TFAM AttentionLayer: norm_self / norm_cross / norm_ffn, AMO_CLIP.py:40,45,50) ----------------
template <typename T, int CH>
__global__ void __launch_bounds__(256) postnorm_kernel(const float* __restrict__ x, const uint16_t* __restrict__ branch,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ sum_out, float* __restrict__ y32, uint16_t* __restrict__ y16,
                                                       float* __restrict__ mean_out, float* __restrict__ rstd_out, int rows, float eps,
                                                       float p1, uint64_t seed1_arg, float p2, uint64_t seed2_arg) {
  const uint64_t seed1 = p1 > 0.f ? resolve_seed(seed1_arg) : 0, seed2 = p2 > 0.f ? resolve_seed(seed2_arg) : 0;
  // p1 / p2 > 0: the branch goes through one or two dropouts first (nn.Dropout after the FFN's second Linear and the
  // block's own dropout, AMO_CLIP.py:28,50) -- same counter-based masks as vmc_dropout on the flat element index
  constexpr int D = CH * 256;
  const int lane = threadIdx.x & 63;
  const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * 4;
  float4 g[CH], b[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    g[c] = *(const float4*)(gamma + c * 256 + lane * 4);
    b[c] = *(const float4*)(beta + c * 256 + lane * 4);
  }
  for (int row = wave_global; row < rows; row += nwaves) {
    float4 v[CH];
    uint2 br[CH];
    const size_t base = (size_t)row * D + lane * 4;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      v[c] = *(const float4*)(x + base + c * 256);
      br[c] = *(const uint2*)(branch + base + c * 256);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      float a0, a1, a2, a3;
      unpack2<T>(br[c].x, a0, a1);
      unpack2<T>(br[c].y, a2, a3);
      if (p1 > 0.f) {
        const uint64_t e = base + c * 256;
        a0 *= dropout_factor(p1, seed1, e); a1 *= dropout_factor(p1, seed1, e + 1);
        a2 *= dropout_factor(p1, seed1, e + 2); a3 *= dropout_factor(p1, seed1, e + 3);
        if (p2 > 0.f) {
          a0 *= dropout_factor(p2, seed2, e); a1 *= dropout_factor(p2, seed2, e + 1);
          a2 *= dropout_factor(p2, seed2, e + 2); a3 *= dropout_factor(p2, seed2, e + 3);
        }
      }
      v[c].x += a0; v[c].y += a1; v[c].z += a2; v[c].w += a3;
      s += (v[c].x + v[c].y) + (v[c].z + v[c].w);
      if (sum_out) *(float4*)(sum_out + base + c * 256) = v[c];
    }
    const float mean = wave_sum(s) * (1.0f / D);
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const float p = v[c].x - mean, q = v[c].y - mean, r = v[c].z - mean, t = v[c].w - mean;
      ss += (p * p + q * q) + (r * r + t * t);
    }
    const float rstd = rsqrtf(wave_sum(ss) * (1.0f / D) + eps);
    if (lane == 0) {
      if (mean_out) mean_out[row] = mean;
      if (rstd_out) rstd_out[row] = rstd;
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      float4 o;
      o.x = (v[c].x - mean) * rstd * g[c].x + b[c].x; o.y = (v[c].y - mean) * rstd * g[c].y + b[c].y;
      o.z = (v[c].z - mean) * rstd * g[c].z + b[c].z; o.w = (v[c].w - mean) * rstd * g[c].w + b[c].w;
      if (y32) *(float4*)(y32 + base + c * 256) = o;
      if (y16) *(uint2*)(y16 + base + c * 256) = make_uint2(pack2<T>(o.x, o.y), pack2<T>(o.z, o.w));
    }
  }
}

template <typename T>
static int launch_postnorm(const float* x, const void* branch, const float* gamma, const float* beta, float* sum_out, float* y32,
                           void* y16, float* mean, float* rstd, int rows, int D, float eps, float p1, uint64_t seed1, float p2,
                           uint64_t seed2, hipStream_t s) {
  const int grid = grid_for((size_t)rows, 4, 256 * 8);
#define VMC_PN(CHN)                                                                                                          \
  hipLaunchKernelGGL((postnorm_kernel<T, CHN>), dim3(grid), dim3(256), 0, s, x, (const uint16_t*)branch, gamma, beta, sum_out, y32, \
                     (uint16_t*)y16, mean, rstd, rows, eps, p1, seed1, p2, seed2)
  switch (D / 256) {
    case 1: VMC_PN(1); break;
    case 2: VMC_PN(2); break;
    case 3: VMC_PN(3); break;
    case 4: VMC_PN(4); break;
    case 6: VMC_PN(6); break;
    case 8: VMC_PN(8); break;
    default: return VMC_E_SHAPE;
  }
#undef VMC_PN
  VMC_CHECK_LAUNCH();
  return 0;
}

extern "C" int vmc_postnorm_dropout_fwd(const float* x, const void* branch, const float* gamma, const float* beta, float* sum_out,
                                        float* y32, void* y16, float* mean, float* rstd, int rows, int D, float eps, float drop_p1,
                                        uint64_t drop_seed1, float drop_p2, uint64_t drop_seed2, int dtype16, void* stream) {
  if (!x || !branch || !gamma || !beta || (!y32 && !y16) || rows <= 0 || D <= 0) return VMC_E_ARG;
  if (D % 256 || D > 2048) return VMC_E_SHAPE;
  if (drop_p1 < 0.f || drop_p1 >= 1.f || drop_p2 < 0.f || drop_p2 >= 1.f || (drop_p2 > 0.f && drop_p1 <= 0.f)) return VMC_E_ARG;
  if (dtype16 == VMC_BF16)
    return launch_postnorm<BF16>(x, branch, gamma, beta, sum_out, y32, y16, mean, rstd, rows, D, eps, drop_p1, drop_seed1, drop_p2,
                                 drop_seed2, (hipStream_t)stream);
  if (dtype16 == VMC_F16)
    return launch_postnorm<F16>(x, branch, gamma, beta, sum_out, y32, y16, mean, rstd, rows, D, eps, drop_p1, drop_seed1, drop_p2,
                                drop_seed2, (hipStream_t)stream);
  return VMC_E_DTYPE;
}

extern "C" int vmc_postnorm_fwd(const float* x, const void* branch, const float* gamma, const float* beta, float* sum_out, float* y32,
                                void* y16, float* mean, float* rstd, int rows, int D, float eps, int dtype16, void* stream) {
  return vmc_postnorm_dropout_fwd(x, branch, gamma, beta, sum_out, y32, y16, mean, rstd, rows, D, eps, 0.f, 0, 0.f, 0, dtype16, stream);
}

// ---- backward ---------------------------------------------------------------------------------
// dx = rstd * (g*w - mean(g*w) - xhat * mean(g*w*xhat));  dgamma = sum_rows g*xhat;  dbeta = sum_rows g.
// Each wave accumulates its rows' dgamma/dbeta partials in registers, then one partial row per block
// goes to the workspace [grid, 2, D]; a second kernel reduces the partials (deterministic, no atomics).
#define LN_BWD_BLOCKS 512
#define LN_BWD_MAX_CHUNKS 8  // D <= 2048 in the backward (register budget: 4 float4 arrays)

// NCH = float4 chunks per lane (D <= 256 NCH): a template constant so that the per-lane register arrays have the size the
// row needs (D = 768: 3 chunks instead of 8), which leaves room to fetch the NEXT row of this wave while the current one
// is reduced and written (one row per wave in flight was the limit: 3.2 TB/s).
template <typename T, int NCH>
__global__ void __launch_bounds__(256) ln_bwd_kernel(const void* __restrict__ dy, const void* __restrict__ dy2, const void* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, void* __restrict__ dx,
                                                     const void* __restrict__ addp, float* __restrict__ partial, int rows,
                                                     int D, size_t ldx, int dy_f32, int x_f32, int dx_f32,
                                                     uint16_t* __restrict__ dx16, float p1, uint64_t seed1_arg, float p2, uint64_t seed2_arg) {
  // dx16 != NULL (vmc_postnorm_bwd): also the 16-bit gradient of the branch that was added in front of the LayerNorm, through the one
  // or two dropouts the forward applied to it -- same masks (counter-based hash of the flat element index), same order of operations
  // as vmc_cast_dropout2 on the stored dx, so the two paths agree bit for bit
  const uint64_t seed1 = (dx16 && p1 > 0.f) ? resolve_seed(seed1_arg) : 0, seed2 = (dx16 && p2 > 0.f) ? resolve_seed(seed2_arg) : 0;
  const uint32_t thr1 = (uint32_t)((double)p1 * 4294967296.0), thr2 = (uint32_t)((double)p2 * 4294967296.0);
  const float sc1 = 1.0f / (1.0f - p1), sc2 = 1.0f / (1.0f - p2);
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [4 waves][2][D] floats
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wave_global = blockIdx.x * 4 + wave;
  const int nwaves = gridDim.x * 4;
  const float invD = 1.0f / (float)D;
  float4 dg[NCH], db[NCH], w[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    dg[c] = db[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int col = c * 256 + lane * 4;
    w[c] = col < D ? *(const float4*)(gamma + col) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float4 ngy[NCH], nxv[NCH], nad[NCH];
  float nmu = 0.f, nrs = 0.f;
  auto fetch = [&](int row) {
    nmu = mean[row];
    nrs = rstd[row];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 256 + lane * 4;
      if (col < D) {
        ngy[c] = load4<T>(dy, (size_t)row * D + col, dy_f32);
        if (dy2) {            // a second incoming gradient (16-bit): the LayerNorm output was consumed as fp32 residual AND as 16-bit operand
          const float4 t = load4<T>(dy2, (size_t)row * D + col, 0);
          ngy[c].x += t.x; ngy[c].y += t.y; ngy[c].z += t.z; ngy[c].w += t.w;
        }
        nxv[c] = load4<T>(x, (size_t)row * ldx + col, x_f32);
        if (addp) nad[c] = load4<T>(addp, (size_t)row * D + col, dx_f32);
      }
    }
  };
  if (wave_global < rows) fetch(wave_global);
  for (int row = wave_global; row < rows; row += nwaves) {
    const float mu = nmu, rs = nrs;
    float4 g[NCH], xh[NCH], ad[NCH];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 256 + lane * 4;
      if (col < D) {
        const float4 gy = ngy[c], xv = nxv[c];
        ad[c] = nad[c];
        xh[c] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
        dg[c].x += gy.x * xh[c].x; dg[c].y += gy.y * xh[c].y; dg[c].z += gy.z * xh[c].z; dg[c].w += gy.w * xh[c].w;
        db[c].x += gy.x; db[c].y += gy.y; db[c].z += gy.z; db[c].w += gy.w;
        g[c] = make_float4(gy.x * w[c].x, gy.y * w[c].y, gy.z * w[c].z, gy.w * w[c].w);
        s1 += (g[c].x + g[c].y) + (g[c].z + g[c].w);
        s2 += (g[c].x * xh[c].x + g[c].y * xh[c].y) + (g[c].z * xh[c].z + g[c].w * xh[c].w);
      }
    }
    if (row + nwaves < rows) fetch(row + nwaves);      // next row's loads fly under the reductions and stores of this one
    s1 = wave_sum(s1) * invD;
    s2 = wave_sum(s2) * invD;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 256 + lane * 4;
      if (col < D) {
        float4 o;
        o.x = rs * (g[c].x - s1 - xh[c].x * s2);
        o.y = rs * (g[c].y - s1 - xh[c].y * s2);
        o.z = rs * (g[c].z - s1 - xh[c].z * s2);
        o.w = rs * (g[c].w - s1 - xh[c].w * s2);
        if (addp) {  // fused residual-branch gradient: dx = LN'(dy) + add   (add has dx's dtype and layout)
          o.x += ad[c].x; o.y += ad[c].y; o.z += ad[c].z; o.w += ad[c].w;
        }
        store4<T>(dx, (size_t)row * D + col, dx_f32, o);
        if (dx16) {
          const size_t e = (size_t)row * D + col;
          float v[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (p1 > 0.f) v[j] = hash32(seed1, e + j) >= thr1 ? v[j] * sc1 : 0.0f;
            if (p2 > 0.f) v[j] = hash32(seed2, e + j) >= thr2 ? v[j] * sc2 : 0.0f;
          }
          store4<T>(dx16, e, false, make_float4(v[0], v[1], v[2], v[3]));
        }
      }
    }
  }
  // block-level reduction of dgamma/dbeta partials
  float* sm = (float*)smem;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = c * 256 + lane * 4;
    if (col < D) {
      *(float4*)(sm + (size_t)(wave * 2 + 0) * D + col) = dg[c];
      *(float4*)(sm + (size_t)(wave * 2 + 1) * D + col) = db[c];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * D; i += 256) {
    const int which = i / D, col = i % D;
    float a = 0.f;
#pragma unroll
    for (int w4 = 0; w4 < 4; ++w4) a += sm[(size_t)(w4 * 2 + which) * D + col];
    partial[((size_t)blockIdx.x * 2 + which) * D + col] = a;
  }
}

// column sums of the [P][2][D] partials: 16 columns x 16 row groups per block (2 D / 16 blocks instead of 2 D / 64 -- the
// 4-row-group version left the chip to 24 blocks and took longer than the LayerNorm backward it finishes)
__global__ void __launch_bounds__(256) ln_reduce_strided_kernel(const float* __restrict__ partial, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta, int P, int D) {
  __shared__ float sm[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4, which = blockIdx.y;
  const int col = blockIdx.x * 16 + tx;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (col < D) {
    const float* base = partial + (size_t)which * D + col;
    const size_t rs = (size_t)2 * D;
    int p = ty;
    for (; p + 48 < P; p += 64) {
      a0 += base[(size_t)p * rs];
      a1 += base[(size_t)(p + 16) * rs];
      a2 += base[(size_t)(p + 32) * rs];
      a3 += base[(size_t)(p + 48) * rs];
    }
    for (; p < P; p += 16) a0 += base[(size_t)p * rs];
  }
  sm[ty][tx] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (ty == 0 && col < D) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += sm[i][tx];
    (which ? dbeta : dgamma)[col] = t;
  }
}

static int ln_bwd_grid(int rows) { return grid_for((size_t)rows, 4, LN_BWD_BLOCKS); }

extern "C" size_t vmc_layernorm_bwd_workspace_bytes(int rows, int D) {
  return (size_t)ln_bwd_grid(rows) * 2 * D * sizeof(float);
}

extern "C" int vmc_layernorm_bwd2(const void* dy, const void* dy2, const void* x, const float* gamma, const float* mean, const float* rstd,
                                  const void* add, void* dx, float* dgamma, float* dbeta, int rows, int D, int ldx, int dy_dtype,
                                  int x_dtype, int dx_dtype, int dtype16, void* workspace, size_t workspace_bytes, void* stream);
extern "C" int vmc_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                 const void* add, void* dx, float* dgamma, float* dbeta, int rows, int D, int ldx, int dy_dtype,
                                 int x_dtype, int dx_dtype, int dtype16, void* workspace, size_t workspace_bytes, void* stream) {
  return vmc_layernorm_bwd2(dy, nullptr, x, gamma, mean, rstd, add, dx, dgamma, dbeta, rows, D, ldx, dy_dtype, x_dtype, dx_dtype, dtype16,
                            workspace, workspace_bytes, stream);
}
static int ln_bwd_impl(const void* dy, const void* dy2, const void* x, const float* gamma, const float* mean, const float* rstd,
                       const void* add, void* dx, float* dgamma, float* dbeta, int rows, int D, int ldx, int dy_dtype,
                       int x_dtype, int dx_dtype, int dtype16, void* workspace, size_t workspace_bytes, void* stream,
                       void* dx16, float p1, uint64_t seed1, float p2, uint64_t seed2);
extern "C" int vmc_layernorm_bwd2(const void* dy, const void* dy2, const void* x, const float* gamma, const float* mean, const float* rstd,
                                  const void* add, void* dx, float* dgamma, float* dbeta, int rows, int D, int ldx, int dy_dtype,
                                  int x_dtype, int dx_dtype, int dtype16, void* workspace, size_t workspace_bytes, void* stream) {
  return ln_bwd_impl(dy, dy2, x, gamma, mean, rstd, add, dx, dgamma, dbeta, rows, D, ldx, dy_dtype, x_dtype, dx_dtype, dtype16, workspace,
                     workspace_bytes, stream, nullptr, 0.f, 0, 0.f, 0);
}
// Backward of vmc_postnorm_dropout_fwd in one launch (+ the partial reduce): dsum (f32) = LN'(dy32 + dy16) on the saved pre-norm sum,
// dbranch16 = dsum through the forward's dropout masks, cast -- what vmc_layernorm_bwd2 + vmc_cast_dropout2 (or a plain cast) did in
// two passes over the gradient.
extern "C" int vmc_postnorm_bwd(const void* dy, const void* dy2, const float* sum, const float* gamma, const float* mean, const float* rstd,
                                float* dsum, void* dbranch16, float* dgamma, float* dbeta, int rows, int D, int dy_dtype, float p1,
                                uint64_t seed1, float p2, uint64_t seed2, int dtype16, void* workspace, size_t workspace_bytes, void* stream) {
  if (!dbranch16 || p1 < 0.f || p1 >= 1.f || p2 < 0.f || p2 >= 1.f || (p2 > 0.f && p1 <= 0.f)) return VMC_E_ARG;   // as the forward
  if (((uintptr_t)dbranch16 | (uintptr_t)dsum) & 15) return VMC_E_ALIGN;
  return ln_bwd_impl(dy, dy2, sum, gamma, mean, rstd, nullptr, dsum, dgamma, dbeta, rows, D, D, dy_dtype, VMC_F32, VMC_F32, dtype16, workspace,
                     workspace_bytes, stream, dbranch16, p1, seed1, p2, seed2);
}
static int ln_bwd_impl(const void* dy, const void* dy2, const void* x, const float* gamma, const float* mean, const float* rstd,
                       const void* add, void* dx, float* dgamma, float* dbeta, int rows, int D, int ldx, int dy_dtype,
                       int x_dtype, int dx_dtype, int dtype16, void* workspace, size_t workspace_bytes, void* stream,
                       void* dx16, float p1, uint64_t seed1, float p2, uint64_t seed2) {
  if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || !workspace || rows <= 0 || D <= 0) return VMC_E_ARG;
  if (D % 4 || D > LN_BWD_MAX_CHUNKS * 256) return VMC_E_SHAPE;
  if (ldx % 4 || ldx < D) return VMC_E_ALIGN;
  if (workspace_bytes < vmc_layernorm_bwd_workspace_bytes(rows, D)) return VMC_E_ARG;
  const int grid = ln_bwd_grid(rows);
  const size_t lds = (size_t)8 * D * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  if (dtype16 != VMC_BF16 && dtype16 != VMC_F16) return VMC_E_DTYPE;
#define VMC_LN_BWD(NCH)                                                                                                          \
  do {                                                                                                                           \
    if (dtype16 == VMC_BF16)                                                                                                     \
      hipLaunchKernelGGL((ln_bwd_kernel<BF16, NCH>), dim3(grid), dim3(256), lds, s, dy, dy2, x, gamma, mean, rstd, dx, add,      \
                         (float*)workspace, rows, D, (size_t)ldx, dy_dtype == VMC_F32, x_dtype == VMC_F32, dx_dtype == VMC_F32,  \
                         (uint16_t*)dx16, p1, seed1, p2, seed2);                                                                 \
    else                                                                                                                         \
      hipLaunchKernelGGL((ln_bwd_kernel<F16, NCH>), dim3(grid), dim3(256), lds, s, dy, dy2, x, gamma, mean, rstd, dx, add,       \
                         (float*)workspace, rows, D, (size_t)ldx, dy_dtype == VMC_F32, x_dtype == VMC_F32, dx_dtype == VMC_F32,  \
                         (uint16_t*)dx16, p1, seed1, p2, seed2);                                                                 \
  } while (0)
  if (D <= 512) VMC_LN_BWD(2);
  else if (D <= 768) VMC_LN_BWD(3);
  else if (D <= 1024) VMC_LN_BWD(4);
  else VMC_LN_BWD(LN_BWD_MAX_CHUNKS);
#undef VMC_LN_BWD
  VMC_CHECK_LAUNCH();
  hipLaunchKernelGGL(ln_reduce_strided_kernel, dim3((D + 15) / 16, 2), dim3(256), 0, s, (const float*)workspace, dgamma, dbeta, grid, D);
  VMC_CHECK_LAUNCH();
  return 0;
}
