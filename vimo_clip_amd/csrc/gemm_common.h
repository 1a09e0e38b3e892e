// Shared GEMM argument block and fused epilogue (bias -> activation -> alpha -> residual -> store).
#pragma once
#include "common.h"

struct GemmArgs {
  const char* A;
  const char* W;
  const float* bias;
  const char* res;
  char* C;
  int M, N, K;
  int lda, ldw, ldc, ldres;
  float alpha;
  int out_f32, res_f32;
  int out_row_group, res_row_mod;
  int tiles_m, tiles_n;
  char* zout;     // optional 16-bit side output [M, ldz]: the value before the activation (bias included) -- what the backward of
  int ldz;        // act(x W^T + b) needs, written by the same epilogue instead of a second pass (vmc_linear_preact)
  int variant;    // VMC_GEMM_* of include/vmc.h (per call; the library keeps no state)
  int walk_gc;    // gemm8: column panels per tile-walk group (0 = 4, the default; VMC_GEMM_GC is the builder's A/B switch)
  int k_slices;   // > 1: split-K (gemm_kernel only): blockIdx.y owns a K range and atomically adds into the f32 output
};

// One lane's run of NV (8 or 16) consecutive output columns of one row.
template <typename T, int ACT, int NV>
__device__ __forceinline__ void gemm_epilogue_row(const GemmArgs& g, int row, int col0, float (&v)[NV], bool vec8) {
  if (row >= g.M) return;
  const int orow = g.out_row_group ? row + row / g.out_row_group + 1 : row;
  const int rrow = g.res_row_mod ? row % g.res_row_mod : orow;
#pragma unroll
  for (int c4 = 0; c4 < NV / 4; ++c4) {
    const int col = col0 + 4 * c4;
    if (col >= g.N) continue;
    float* x = &v[4 * c4];
    if (g.bias) {
      const float4 b = *(const float4*)(g.bias + col);
      x[0] += b.x; x[1] += b.y; x[2] += b.z; x[3] += b.w;
    }
    if (g.zout) *(uint2*)((uint16_t*)g.zout + (size_t)orow * g.ldz + col) = make_uint2(pack2<T>(x[0], x[1]), pack2<T>(x[2], x[3]));
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = g.alpha * apply_act<ACT>(x[j]);
    if (g.res) {
      if (g.res_f32) {
        const float4 rr = *(const float4*)((const float*)g.res + (size_t)rrow * g.ldres + col);
        x[0] += rr.x; x[1] += rr.y; x[2] += rr.z; x[3] += rr.w;
      } else {
        const uint2 rr = *(const uint2*)((const uint16_t*)g.res + (size_t)rrow * g.ldres + col);
        float a, b;
        unpack2<T>(rr.x, a, b); x[0] += a; x[1] += b;
        unpack2<T>(rr.y, a, b); x[2] += a; x[3] += b;
      }
    }
  }
  if (g.out_f32) {
    float* dst = (float*)g.C + (size_t)orow * g.ldc + col0;
#pragma unroll
    for (int c4 = 0; c4 < NV / 4; ++c4)
      if (col0 + 4 * c4 < g.N) *(float4*)(dst + 4 * c4) = make_float4(v[4 * c4], v[4 * c4 + 1], v[4 * c4 + 2], v[4 * c4 + 3]);
  } else {
    uint16_t* dst = (uint16_t*)g.C + (size_t)orow * g.ldc + col0;
    if (vec8) {
#pragma unroll
      for (int c8 = 0; c8 < NV / 8; ++c8)
        if (col0 + 8 * c8 < g.N)
          *(uint4*)(dst + 8 * c8) = make_uint4(pack2<T>(v[8 * c8], v[8 * c8 + 1]), pack2<T>(v[8 * c8 + 2], v[8 * c8 + 3]),
                                               pack2<T>(v[8 * c8 + 4], v[8 * c8 + 5]), pack2<T>(v[8 * c8 + 6], v[8 * c8 + 7]));
    } else {
#pragma unroll
      for (int c4 = 0; c4 < NV / 4; ++c4)
        if (col0 + 4 * c4 < g.N)
          *(uint2*)(dst + 4 * c4) = make_uint2(pack2<T>(v[4 * c4], v[4 * c4 + 1]), pack2<T>(v[4 * c4 + 2], v[4 * c4 + 3]));
    }
  }
}

__device__ __forceinline__ bool gemm_vec8_ok(const GemmArgs& g) {
  return ((g.N & 7) == 0) && ((g.ldc & 7) == 0) && (g.res == nullptr || (g.ldres & 7) == 0);
}


// 8-phase 256x256 kernel family (gemm8.hip); returns VMC_E_SHAPE when the shape does not qualify.
int vmc_gemm8_launch(GemmArgs& g, int act, int dtype16, hipStream_t stream);

// 256 x 256 TN weight-gradient kernel (gemm_tn256.hip), routed by gemm_tn.hip
bool vmc_tn256_eligible(int M, int N, int K, int lddy, int ldx);
void vmc_tn256_slices(int M, int N, int K, int* slices, int* pairs_per_slice);
int vmc_tn256_launch(const void* dY, const void* X, float* dst, float* bdst, int M, int N, int K, int lddy, int ldx, int slices,
                     int pairs_per_slice, int dtype16, hipStream_t s);
