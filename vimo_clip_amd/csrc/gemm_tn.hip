// TN weight-gradient GEMM:  C[N,K] (f32) = dY[M,N]^T @ X[M,K], both operands token-major exactly as the forward
// pass left them -- no transposed copies.  The contraction runs over the M tokens.
//
// Block tile 256 (n) x 128 (k), 8 waves as 4 (n) x 2 (k), each wave 64 x 64 = 4 x 4 MFMA tiles.  64 tokens per
// pipeline stage.  A stage holds three [64 tokens][128 columns] sub-tiles (256-B rows of sixteen 16-B chunks): dY
// columns 0..127, dY columns 128..255, X columns 0..127.  They are filled by global_load_lds_dwordx4 (lane-linear
// LDS image; the XOR swizzle of the chunk slot by the token row is applied on the SOURCE address) into a 3-stage ring
// (144 KiB), two stages in flight ahead of the MFMAs, counted vmcnt; the two waves of a SIMD run one barrier apart (ping-pong, as
// gemm8.hip: one issues its MFMA cluster while the other issues its transposed reads and the DMA; student step 22.2 -> 21.9 ms).  Both MFMA operands are fetched with
// ds_read_b64_tr_b16: lane (c = lane & 15, g = lane >> 4) receives tokens 8g..8g+7 of column c, the k-slots of
// v_mfma_f32_16x16x32.  With the swizzle slot = chunk ^ (((row & 3) << 2) | ((row >> 2) & 3)) the 32 lanes of one
// transposed read touch 32 distinct 8-byte units of 64 banks (conflict free).
// The token range is split in slices (blockIdx.y): slabs in a workspace + deterministic reduce, as the NT split-K.
#include "gemm_tn_body.h"

// Block order: workgroups that share a dY panel (same n tile, same token slice) or an X panel (same k tile) get neighbouring logical
// ids, and xcd_remap hands every XCD a contiguous range of logical ids -- so they run on one XCD and a panel is fetched into that
// XCD's L2 once instead of once per workgroup.  With the plain (x = tile, y = slice) grid the six k tiles of a 768-wide dY panel sat on
// six different XCDs and every stage came from the Infinity Cache (44 GB/s per CU, its rate): 768 x 768 at M = 25 600 64 -> 50 us.
template <typename T>
__global__ void __launch_bounds__(512) gemm_tn_kernel(const uint16_t* __restrict__ dY, const uint16_t* __restrict__ X, float* __restrict__ C,
                                                      float* __restrict__ dbias, int M, int N, int K, int lddy, int ldx, int tiles_k,
                                                      int tiles_n, int slices) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int per_slice = tiles_n * tiles_k, total = per_slice * slices;
  // 16 and more k tiles (K >= 2048): the dispatch-order map (tile = id mod tiles, XCD = id mod 8) already gives every XCD a
  // 4 (n) x tiles_k / 8 (k) block per slice and measured 2 % faster than the chunked order; everything else is re-ordered
  const bool plain = tiles_k >= 16;
  const int L = plain ? (int)blockIdx.x : xcd_remap(blockIdx.x, total);
  const int slice = L / per_slice, rest = L - slice * per_slice;
  // inside a slice: k tiles in chunks of 8, n tiles inside a chunk, k tiles fastest -- the 32 consecutive ids of an XCD are then a
  // 4 (n) x 8 (k) block of tiles: 4 dY panels + 8 X panels per stage (256 KB) instead of 1 + 32 (544 KB at K = 4096)
  constexpr int KW = 8;
  const int nfull = tiles_k / KW, full_sz = tiles_n * KW;
  int tn, tk;
  if (plain) {
    tn = rest / tiles_k;
    tk = rest - tn * tiles_k;
  } else if (rest < nfull * full_sz) {
    const int c = rest / full_sz, w = rest - c * full_sz;
    tn = w / KW;
    tk = c * KW + (w - tn * KW);
  } else {
    const int rem = rest - nfull * full_sz, wl = tiles_k - nfull * KW;
    tn = rem / wl;
    tk = nfull * KW + (rem - tn * wl);
  }
  const int steps_all = (M + 63) / 64;
  const int per = (steps_all + slices - 1) / slices;
  const int s0 = slice * per, s1 = min(steps_all, s0 + per);
  tn_tile_body<T>(dY, X, C, dbias, M, N, K, lddy, ldx, tn, tk, s0, s1, slice, smem);
}

// out[i] = sum over slices of slab[s][i] (float4 per thread, fixed order: deterministic).  The bias slabs sit behind the weight
// slabs in the workspace and are reduced by the same launch (n4b float4s into outb).
__global__ void __launch_bounds__(256) tn_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out, int slices, size_t n4,
                                                        const float* __restrict__ bslabs, float* __restrict__ outb, size_t n4b) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4 + n4b) return;
  if (i >= n4) {
    i -= n4;
    slabs = bslabs; out = outb; n4 = n4b;
  }
  float4 s = ((const float4*)slabs)[i];
  for (int k = 1; k < slices; ++k) {
    const float4 v = ((const float4*)slabs)[(size_t)k * n4 + i];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  ((float4*)out)[i] = s;
}

static int tn_slices(int M, int N, int K) {
  const int tiles = ((N + 255) / 256) * ((K + 127) / 128);
  // one workgroup per CU (144 KiB of LDS): aim at whole rounds of 256 workgroups -- one round when the tiles allow it
  // (fewer, longer slices: half the slab traffic of the reduce), two when a single round would leave > 1/4 of the chip idle
  // (measured against ceil(512 / tiles) slices: 62 vs 83 us at 768x768, 164 vs 206 us at 3072x768, M = 25 600)
  int slices = 256 / tiles;
  if (slices >= 1 && tiles * slices < 192) slices = 512 / tiles;
  const int steps = (M + 63) / 64;
  if (slices > steps / 4) slices = steps / 4;
  if (slices < 1) slices = 1;
  const int per = (steps + slices - 1) / slices;
  return (steps + per - 1) / per;                      // no empty trailing slice
}
// Large problems made of whole 128-token pairs go to the 256 x 256 tile (gemm_tn256.hip); VMC_TN256=0 is the builder's A/B switch.
static bool tn_use256(int M, int N, int K, int lddy, int ldx) {
  static const bool on = !(getenv("VMC_TN256") && atoi(getenv("VMC_TN256")) == 0);
  return on && vmc_tn256_eligible(M, N, K, lddy, ldx);
}
extern "C" size_t vmc_linear_wgrad_tn_workspace_bytes(int M, int N, int K) {
  int s = tn_slices(M, N, K);
  if (vmc_tn256_eligible(M, N, K, N, K)) {     // the leading dimensions are not known here: room for whichever kernel the call takes
    int s2, per;
    vmc_tn256_slices(M, N, K, &s2, &per);
    if (s2 > s) s = s2;
  }
  return s > 1 ? (size_t)s * ((size_t)N * K + N) * sizeof(float) : 0;      // weight slabs, then bias slabs
}
extern "C" int vmc_linear_wgrad_bias_tn(const void* dY, const void* X, float* C, float* dbias, int M, int N, int K, int lddy, int ldx,
                                        void* workspace, size_t workspace_bytes, int dtype16, void* stream) {
  if (!dY || !X || !C || M <= 0 || N <= 0 || K <= 0) return VMC_E_ARG;
  if ((N % 8) || (K % 8)) return VMC_E_SHAPE;
  if ((lddy % 8) || (ldx % 8)) return VMC_E_ALIGN;
  if (((uintptr_t)dY | (uintptr_t)X | (uintptr_t)C | (uintptr_t)workspace | (uintptr_t)dbias) & 15) return VMC_E_ALIGN;
  const bool big = tn_use256(M, N, K, lddy, ldx);
  int slices = tn_slices(M, N, K), pairs_per_slice = 0;
  if (big) vmc_tn256_slices(M, N, K, &slices, &pairs_per_slice);
  if (slices > 1 && (!workspace || workspace_bytes < (size_t)slices * ((size_t)N * K + N) * sizeof(float))) return VMC_E_ARG;
  float* dst = slices > 1 ? (float*)workspace : C;
  float* bdst = !dbias ? nullptr : (slices > 1 ? (float*)workspace + (size_t)slices * N * K : dbias);
  hipStream_t s = (hipStream_t)stream;
  if (big) {
    const int rc = vmc_tn256_launch(dY, X, dst, bdst, M, N, K, lddy, ldx, slices, pairs_per_slice, dtype16, s);
    if (rc) return rc;
    if (slices > 1) {
      const size_t n4 = (size_t)N * K / 4, n4b = dbias ? (size_t)N / 4 : 0;
      hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)((n4 + n4b + 255) / 256)), dim3(256), 0, s, (const float*)workspace, C, slices, n4,
                         (const float*)bdst, dbias, n4b);
      VMC_CHECK_LAUNCH();
    }
    return 0;
  }
  const int tiles_k = (K + 127) / 128, tiles_n = (N + 255) / 256;
  dim3 grid(tiles_n * tiles_k * slices);
  const size_t lds = (size_t)TN_STAGES * TN_STAGE;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_kernel<BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_tn_kernel<F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  if (dtype16 == VMC_BF16)
    hipLaunchKernelGGL(gemm_tn_kernel<BF16>, grid, dim3(512), lds, s, (const uint16_t*)dY, (const uint16_t*)X, dst, bdst, M, N, K, lddy, ldx, tiles_k, tiles_n, slices);
  else if (dtype16 == VMC_F16)
    hipLaunchKernelGGL(gemm_tn_kernel<F16>, grid, dim3(512), lds, s, (const uint16_t*)dY, (const uint16_t*)X, dst, bdst, M, N, K, lddy, ldx, tiles_k, tiles_n, slices);
  else
    return VMC_E_DTYPE;
  VMC_CHECK_LAUNCH();
  if (slices > 1) {
    const size_t n4 = (size_t)N * K / 4, n4b = dbias ? (size_t)N / 4 : 0;
    hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)((n4 + n4b + 255) / 256)), dim3(256), 0, s, (const float*)workspace, C, slices, n4,
                       (const float*)bdst, dbias, n4b);
    VMC_CHECK_LAUNCH();
  }
  return 0;
}
extern "C" int vmc_linear_wgrad_tn(const void* dY, const void* X, float* C, int M, int N, int K, int lddy, int ldx, void* workspace,
                                   size_t workspace_bytes, int dtype16, void* stream) {
  return vmc_linear_wgrad_bias_tn(dY, X, C, nullptr, M, N, K, lddy, ldx, workspace, workspace_bytes, dtype16, stream);
}
