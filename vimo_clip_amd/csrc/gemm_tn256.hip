// TN weight-gradient GEMM on a 256 (n) x 256 (k) output tile:  C[N,K] (f32) = dY[M,N]^T @ X[M,K], operands token-major as the
// forward pass left them (gemm_tn.hip has the 256 x 128 form and the design notes of the transposed LDS reads).
//
// Why a second tile: the 256 x 128 kernel stages 48 KiB per 64 tokens for 2 * 256 * 128 * 64 FLOP = 85 FLOP per staged byte and is
// bound by what one CU pulls through the fabric (37 GB/s per CU measured on the 1024 x 4096, M = 65 792 problem: 9.6 TB/s over the
// chip, MFMA issue 37 %); a 256 x 256 tile stages 64 KiB for twice the FLOPs (128 FLOP/B, the ratio of the NT kernel of gemm8.hip).
// The loop IS gemm8.hip's 8-phase schedule -- 8 waves as 2 x 4, waves w and w + 4 of a SIMD one barrier apart, LDS = 8 half-tile
// slots of 16 KiB (A0, A1, B0, B1 for the even and the odd 64-token stage), one half-tile staged per phase, four in flight across
// the barriers (vmcnt(8)) -- with the operands renamed: "A" = dY column halves, "B" = X column halves, the "K tile" = 64 tokens.
//   half-tile image: [64 tokens][128 columns] = 256-B rows of sixteen 16-B chunks, chunk slot XOR tn_swz(token row) applied on the
//   DMA source (buffer_load ... lds: voffset = the lane's fixed offset, soffset = the panel in an SGPR);
//   fragments: ds_read_b64_tr_b16 pairs (lane (c, g) gets tokens 8g .. 8g+7 of column c); the column tile (mt / nt) of a fragment
//   only flips bits 1-2 of the chunk index, so its address is (address of tile 0) ^ (tile << 5);
//   MFMA operands (X, dY): a lane ends with four consecutive k of one output row -> 16-byte f32 stores.
// Bias gradient (column sums of dY) from the same launch: the two waves that own the tile's first 32 X columns of the first k
// tile feed their dY fragments into 8 more MFMAs per fresh A half against 0/1 pattern operands -- pattern mt has ones in the
// dummy columns 4 mt .. 4 mt + 3, so the four 16-row tiles of the wave land in the four column groups of ONE accumulator and
// every lane of the wave ends with the sum of one dY column.
// Host side (gemm_tn.hip) takes this kernel when M is a multiple of 128 (whole stage pairs, no token tail) and the problem has
// at least 16 tiles of 256 x 256; token slices, slabs and the fixed-order reduce are those of the 256 x 128 kernel.
#include "gemm_common.h"

#define T2_SLOT 16384

__device__ __forceinline__ uint32_t t2_lds_addr(const char* p) { return (uint32_t)(uintptr_t)(const VMC_LDS char*)p; }

// inline asm for the reason given in gemm8.hip (hipcc would drain vmcnt(0) in front of C++ LDS reads while LDS-DMA is in flight)
template <int IMM>
__device__ __forceinline__ void t2_read_tr(uint2& v, uint32_t addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM) : "memory");
}

struct T2Frags {
  uint2 a[4][2][2];    // [mt][kk][e] current dY half (64 output rows of this wave)
  uint2 b0[2][2][2];   // [nt][kk][e] X half 0 (this wave's 32 output columns), kept from phase 0 to phase 3
  uint2 b1[2][2][2];
};

__device__ __forceinline__ void t2_stage(const __amdgpu_buffer_rsrc_t rs, char* slot, uint32_t panel, const uint32_t (&off)[2], int wave_lds) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (VMC_LDS void*)(slot + i * 8192 + wave_lds), 16, off[i], panel, 0, 0);
}

// The offsets are laundered through an empty asm: otherwise hipcc hoists slot base + (offset ^ tile) for all 8 slots out of the
// loop (48 address registers) and spills, and a spill's scratch accesses would join the hand-counted vmcnt queue.  Slots are
// 16 KiB-aligned and the offsets < 16 KiB, so (base + offset) ^ (tile << 5) == base + (offset ^ (tile << 5)).
__device__ __forceinline__ void t2_read_a(const char* slot, const uint32_t (&aoff)[2], uint2 (&a)[4][2][2]) {
  uint32_t o0 = aoff[0], o1 = aoff[1];
  asm volatile("" : "+v"(o0), "+v"(o1));
  const uint32_t base = t2_lds_addr(slot);
  const uint32_t ad[2] = {base + o0, base + o1};
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const uint32_t x = ad[e] ^ (uint32_t)(mt << 5);
      t2_read_tr<0>(a[mt][0][e], x);
      t2_read_tr<8192>(a[mt][1][e], x);
    }
}
__device__ __forceinline__ void t2_read_b(const char* slot, const uint32_t (&boff)[2], uint2 (&b)[2][2][2]) {
  uint32_t o0 = boff[0], o1 = boff[1];
  asm volatile("" : "+v"(o0), "+v"(o1));
  const uint32_t base = t2_lds_addr(slot);
  const uint32_t ad[2] = {base + o0, base + o1};
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const uint32_t x = ad[e] ^ (uint32_t)(nt << 5);
      t2_read_tr<0>(b[nt][0][e], x);
      t2_read_tr<8192>(b[nt][1][e], x);
    }
}

__device__ __forceinline__ uint4 t2_op(const uint2 (&f)[2]) { return make_uint4(f[0].x, f[0].y, f[1].x, f[1].y); }

// one phase: barrier | 16 MFMAs (+ 8 for the bias gradient when this wave owns it and the A half is fresh) | barrier.
// psel = (lane & 15) >> 2: the pattern operand sits on the X side, whose lane (c, .) carries dummy column c -> ones iff c >> 2 == mt;
// the product's lane (r, g) register j is [dY column r of tile mt][dummy column 4 g + j], non-zero for mt == g only
template <typename T, bool BIAS_PHASE>
__device__ __forceinline__ void t2_mma(f32x4 (&acc)[4][2], const uint2 (&a)[4][2][2], const uint2 (&b)[2][2][2], bool do_bias, int psel,
                                       f32x4& bacc) {
  __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = T::mfma16(t2_op(b[nt][kk]), t2_op(a[mt][kk]), acc[mt][nt]);
  if constexpr (BIAS_PHASE) {
    if (do_bias) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const uint32_t one = psel == mt ? T::ONE_PAIR : 0u;
        const uint4 pat = make_uint4(one, one, one, one);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) bacc = T::mfma16(pat, t2_op(a[mt][kk]), bacc);
      }
    }
  }
  __builtin_amdgcn_s_setprio(0);
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
}

#define T2_WAIT8() asm volatile("s_waitcnt vmcnt(8)" ::: "memory")

// (tn, tk, slice) from the block id: the XCD-aware order of gemm_tn.hip (k tiles in chunks of 8, n tiles inside a chunk)
__device__ __forceinline__ void t2_coords(int bid, int tiles_n, int tiles_k, int slices, int& tn, int& tk, int& slice) {
  const int per_slice = tiles_n * tiles_k, total = per_slice * slices;
  const int L = xcd_remap(bid, total);
  slice = L / per_slice;
  const int rest = L - slice * per_slice;
  constexpr int KW = 8;
  const int nfull = tiles_k / KW, full_sz = tiles_n * KW;
  if (rest < nfull * full_sz) {
    const int c = rest / full_sz, w = rest - c * full_sz;
    tn = w / KW;
    tk = c * KW + (w - tn * KW);
  } else {
    const int rem = rest - nfull * full_sz, wl = tiles_k - nfull * KW;
    tn = rem / wl;
    tk = nfull * KW + (rem - tn * wl);
  }
}

// One 256 x 256 output tile (tn, tk) over the 64-token stages [s0, s0 + nkt) (nkt even, >= 2); partial result to slab `slice`.
template <typename T>
__device__ __forceinline__ void tn256_tile_body(const uint16_t* __restrict__ dY, const uint16_t* __restrict__ X, float* __restrict__ C,
                                                float* __restrict__ dbias, int M, int N, int K, int lddy, int ldx, int tn, int tk, int s0,
                                                int nkt, int slice, char* smem) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int r = lane & 15, g = lane >> 4, q = r >> 2;
  const int n0 = tn * 256, k0 = tk * 256;

  // LDS-DMA: image chunk c = i * 512 + tid of a half-tile -> token row c >> 4, slot c & 15, source chunk = slot ^ swz(row)
  uint32_t oa[2], ob[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int row, ch;
    tn_stage_src(i * 512 + tid, row, ch);
    oa[i] = (uint32_t)(row * lddy + ch * 8) * 2u;
    ob[i] = (uint32_t)(row * ldx + ch * 8) * 2u;
  }
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)dY, 0, (int)((size_t)M * lddy * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)((size_t)M * ldx * 2), 0x00020000);
  const uint32_t SA = 64u * (uint32_t)lddy * 2u, SB = 64u * (uint32_t)ldx * 2u;     // one 64-token stage
  const uint32_t pa0 = (uint32_t)s0 * SA + (uint32_t)n0 * 2u, pb0 = (uint32_t)s0 * SB + (uint32_t)k0 * 2u;
  const int wave_lds = wave * 1024;

  // transposed-read offsets: token row 8g + q + 4e, 8-byte piece of columns 4p .. 4p+3 inside the 16-column tile
  uint32_t aoff[2], boff[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    aoff[e] = (uint32_t)tn256_a_off(wm, lane, e);
    boff[e] = (uint32_t)tn256_b_off(wn, lane, e);
  }

  f32x4 acc[2][2][4][2];  // [mh][nh][mt][nt]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int d = 0; d < 2; ++d) acc[a][b][c][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 bacc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
  const bool do_bias = __builtin_amdgcn_readfirstlane((dbias != nullptr && tk == 0 && wn == 0) ? 1 : 0) != 0;

  char* const A0e = smem + 0 * T2_SLOT; char* const A1e = smem + 1 * T2_SLOT;
  char* const B0e = smem + 2 * T2_SLOT; char* const B1e = smem + 3 * T2_SLOT;
  char* const A0o = smem + 4 * T2_SLOT; char* const A1o = smem + 5 * T2_SLOT;
  char* const B0o = smem + 6 * T2_SLOT; char* const B1o = smem + 7 * T2_SLOT;

  const int last = nkt - 1;
  // prologue: same issue order as the steady state so the vmcnt(8) accounting holds from the first phase
  t2_stage(ra, A0e, pa0, oa, wave_lds); t2_stage(rb, B0e, pb0, ob, wave_lds); t2_stage(rb, B1e, pb0 + 256u, ob, wave_lds);
  t2_stage(ra, A1e, pa0 + 256u, oa, wave_lds); t2_stage(ra, A0o, pa0 + SA, oa, wave_lds); t2_stage(rb, B0o, pb0 + SB, ob, wave_lds);
  T2_WAIT8();
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();  // stagger the second wave of every SIMD by one barrier

  T2Frags f;
  for (int t = 0; t < nkt; t += 2) {
    const uint32_t t1 = (uint32_t)min(t + 1, last), t2 = (uint32_t)min(t + 2, last), t3 = (uint32_t)min(t + 3, last);
    const uint32_t a1 = pa0 + t1 * SA, a2 = pa0 + t2 * SA, a3 = pa0 + t3 * SA;
    const uint32_t b1 = pb0 + t1 * SB, b2 = pb0 + t2 * SB, b3 = pb0 + t3 * SB;
    // ---- even stage ----
    t2_read_b(B0e, boff, f.b0); t2_read_a(A0e, aoff, f.a);
    t2_stage(rb, B1o, b1 + 256u, ob, wave_lds); T2_WAIT8();
    t2_mma<T, true>(acc[0][0], f.a, f.b0, do_bias, q, bacc[0]);
    t2_read_b(B1e, boff, f.b1);
    t2_stage(ra, A1o, a1 + 256u, oa, wave_lds); T2_WAIT8();
    t2_mma<T, false>(acc[0][1], f.a, f.b1, false, q, bacc[0]);
    t2_read_a(A1e, aoff, f.a);
    t2_stage(ra, A0e, a2, oa, wave_lds);
    t2_mma<T, true>(acc[1][1], f.a, f.b1, do_bias, q, bacc[1]);
    t2_stage(rb, B0e, b2, ob, wave_lds); T2_WAIT8();
    t2_mma<T, false>(acc[1][0], f.a, f.b0, false, q, bacc[1]);
    // ---- odd stage ----
    t2_read_b(B0o, boff, f.b0); t2_read_a(A0o, aoff, f.a);
    t2_stage(rb, B1e, b2 + 256u, ob, wave_lds); T2_WAIT8();
    t2_mma<T, true>(acc[0][0], f.a, f.b0, do_bias, q, bacc[0]);
    t2_read_b(B1o, boff, f.b1);
    t2_stage(ra, A1e, a2 + 256u, oa, wave_lds); T2_WAIT8();
    t2_mma<T, false>(acc[0][1], f.a, f.b1, false, q, bacc[0]);
    t2_read_a(A1o, aoff, f.a);
    t2_stage(ra, A0o, a3, oa, wave_lds);
    t2_mma<T, true>(acc[1][1], f.a, f.b1, do_bias, q, bacc[1]);
    t2_stage(rb, B0o, b3, ob, wave_lds); T2_WAIT8();
    t2_mma<T, false>(acc[1][0], f.a, f.b0, false, q, bacc[1]);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tail DMAs (clamped re-loads of the last stage) must land before exit
  if (wm == 0) __builtin_amdgcn_s_barrier();        // balance the stagger barrier

  // lane (r, g): C[n = n0 + 128 mh + 64 wm + 16 mt + r][k = k0 + 128 nh + 32 wn + 16 nt + 4 g + j], j = 0..3 (K % 8 == 0: a started
  // group of four never crosses the edge)
  float* out = C + (size_t)slice * N * K;
#pragma unroll
  for (int mh = 0; mh < 2; ++mh)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int n = n0 + 128 * mh + 64 * wm + 16 * mt + r;
      if (n >= N) continue;
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const int k = k0 + 128 * nh + 32 * wn + 16 * nt + 4 * g;
          if (k < K)
            *(float4*)(out + (size_t)n * K + k) = make_float4(acc[mh][nh][mt][nt][0], acc[mh][nh][mt][nt][1], acc[mh][nh][mt][nt][2], acc[mh][nh][mt][nt][3]);
        }
    }
  if (do_bias) {        // lane (r, g) of the pattern product holds the sum of dY column 64 wm + 16 g + r of the A half (all four j)
    float* bo = dbias + (size_t)slice * N;
#pragma unroll
    for (int mh = 0; mh < 2; ++mh) {
      const int n = n0 + 128 * mh + 64 * wm + 16 * g + r;
      if (n < N) bo[n] = bacc[mh][0];
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(512, 2) gemm_tn256_kernel(const uint16_t* __restrict__ dY, const uint16_t* __restrict__ X, float* __restrict__ C,
                                                            float* __restrict__ dbias, int M, int N, int K, int lddy, int ldx, int tiles_k,
                                                            int tiles_n, int slices, int pairs_per_slice) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int tn, tk, slice;
  t2_coords(blockIdx.x, tiles_n, tiles_k, slices, tn, tk, slice);
  const int pairs_all = M >> 7;
  const int p0 = slice * pairs_per_slice, p1 = min(pairs_all, p0 + pairs_per_slice);
  tn256_tile_body<T>(dY, X, C, dbias, M, N, K, lddy, ldx, tn, tk, 2 * p0, 2 * (p1 - p0), slice, smem);     // no empty slice (host)
}

// ---- many weight gradients in ONE launch ------------------------------------------------------------------------------------------
// A training step's weight gradients are many independent problems that are each too small to fill the chip (TFAM at B = 512:
// 768 x 768 .. 2304 x 768 over 8192 tokens = 9 .. 27 tiles): launched one by one they are cut into up to 14 token slices whose
// slabs a second launch reduces (28 + 7 us for 9.7 GFLOP).  Grouped, every tile runs over ALL tokens of its problem (no slabs, no
// reduce, one launch), and the tiles of up to 32 problems fill the rounds.  Same tile body; block -> (problem, tile) through the
// prefix table in the kernel argument; the XCD-aware order is applied to the whole grid.
struct Tn256Group {
  vmc_wgrad_tn_problem p[VMC_WGRAD_GROUP_MAX];
  int tile0[VMC_WGRAD_GROUP_MAX + 1];     // first block of problem i; tile0[n] = grid size
  int n;
};
template <typename T>
__global__ void __launch_bounds__(512, 2) gemm_tn256_group_kernel(const Tn256Group g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int L = xcd_remap(blockIdx.x, g.tile0[g.n]);
  int i = 0;
  while (i + 1 < g.n && g.tile0[i + 1] <= L) ++i;
  const vmc_wgrad_tn_problem& pr = g.p[i];
  const int tiles_k = (pr.K + 255) / 256, tiles_n = (pr.N + 255) / 256;
  const int rest = L - g.tile0[i];
  constexpr int KW = 8;                         // k tiles in chunks of 8, n tiles inside a chunk (t2_coords)
  const int nfull = tiles_k / KW, full_sz = tiles_n * KW;
  int tn, tk;
  if (rest < nfull * full_sz) {
    const int c = rest / full_sz, w = rest - c * full_sz;
    tn = w / KW;
    tk = c * KW + (w - tn * KW);
  } else {
    const int rem = rest - nfull * full_sz, wl = tiles_k - nfull * KW;
    tn = rem / wl;
    tk = nfull * KW + (rem - tn * wl);
  }
  tn256_tile_body<T>((const uint16_t*)pr.dY, (const uint16_t*)pr.X, pr.C, pr.dbias, pr.M, pr.N, pr.K, pr.lddy, pr.ldx, tn, tk, 0, pr.M >> 6, 0,
                     smem);
}

// ---- host side ----------------------------------------------------------------------------------------------------------------
// Eligibility and slicing are shared with gemm_tn.hip through these two functions (declared in gemm_common.h).
bool vmc_tn256_eligible(int M, int N, int K, int lddy, int ldx) {
  if (M < 256 || (M & 127)) return false;                                    // whole stage pairs, no token tail
  const long tiles = (long)((N + 255) / 256) * ((K + 255) / 256);
  if (tiles < 16) return false;                                              // fewer: too many slabs for the reduce (768 x 768: 28)
  if ((size_t)M * lddy * 2 >= (1ull << 31) || (size_t)M * ldx * 2 >= (1ull << 31)) return false;     // buffer descriptors
  // short slices are all prologue and slab traffic (M = 4096, 512 x 2048: 16 slices of two pairs, 32 vs 27 us)
  static const int min_pairs = getenv("VMC_TN256_MINPAIRS") ? atoi(getenv("VMC_TN256_MINPAIRS")) : 8;
  int slices, per;
  vmc_tn256_slices(M, N, K, &slices, &per);
  return per >= min_pairs;
}
void vmc_tn256_slices(int M, int N, int K, int* slices, int* pairs_per_slice) {
  const int tiles = ((N + 255) / 256) * ((K + 255) / 256), pairs = M / 128;
  // one workgroup per CU (128 KiB of LDS): one round of at most 256 workgroups
  int s = 256 / tiles;
  if (s > pairs / 2) s = pairs / 2;
  if (s < 1) s = 1;
  const int per = (pairs + s - 1) / s;
  *pairs_per_slice = per;
  *slices = (pairs + per - 1) / per;                  // no empty trailing slice
}

int vmc_tn256_launch(const void* dY, const void* X, float* dst, float* bdst, int M, int N, int K, int lddy, int ldx, int slices,
                     int pairs_per_slice, int dtype16, hipStream_t s) {
  const int tiles_k = (K + 255) / 256, tiles_n = (N + 255) / 256;
  const size_t lds = 8 * T2_SLOT;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_tn256_kernel<BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_tn256_kernel<F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  dim3 grid(tiles_n * tiles_k * slices);
  if (dtype16 == VMC_BF16)
    hipLaunchKernelGGL(gemm_tn256_kernel<BF16>, grid, dim3(512), lds, s, (const uint16_t*)dY, (const uint16_t*)X, dst, bdst, M, N, K, lddy, ldx,
                       tiles_k, tiles_n, slices, pairs_per_slice);
  else if (dtype16 == VMC_F16)
    hipLaunchKernelGGL(gemm_tn256_kernel<F16>, grid, dim3(512), lds, s, (const uint16_t*)dY, (const uint16_t*)X, dst, bdst, M, N, K, lddy, ldx,
                       tiles_k, tiles_n, slices, pairs_per_slice);
  else
    return VMC_E_DTYPE;
  VMC_CHECK_LAUNCH();
  return 0;
}

extern "C" int vmc_linear_wgrad_tn_group(const vmc_wgrad_tn_problem* probs, int n, int dtype16, void* stream) {
  if (!probs || n <= 0 || n > VMC_WGRAD_GROUP_MAX) return VMC_E_ARG;
  if (dtype16 != VMC_BF16 && dtype16 != VMC_F16) return VMC_E_DTYPE;
  Tn256Group g;
  g.n = n;
  int tiles = 0;
  for (int i = 0; i < n; ++i) {
    const vmc_wgrad_tn_problem& p = probs[i];
    if (!p.dY || !p.X || !p.C || p.M < 256 || p.N <= 0 || p.K <= 0) return VMC_E_ARG;
    if ((p.M & 127) || (p.N % 8) || (p.K % 8)) return VMC_E_SHAPE;
    if ((p.lddy % 8) || (p.ldx % 8) || p.lddy < p.N || p.ldx < p.K) return VMC_E_ALIGN;
    if (((uintptr_t)p.dY | (uintptr_t)p.X | (uintptr_t)p.C | (uintptr_t)p.dbias) & 15) return VMC_E_ALIGN;
    if ((size_t)p.M * p.lddy * 2 >= (1ull << 31) || (size_t)p.M * p.ldx * 2 >= (1ull << 31)) return VMC_E_SHAPE;      // buffer descriptors
    g.p[i] = p;
    g.tile0[i] = tiles;
    tiles += ((p.N + 255) / 256) * ((p.K + 255) / 256);
  }
  g.tile0[n] = tiles;
  const size_t lds = 8 * T2_SLOT;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_tn256_group_kernel<BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_tn256_group_kernel<F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  if (dtype16 == VMC_BF16)
    hipLaunchKernelGGL(gemm_tn256_group_kernel<BF16>, dim3(tiles), dim3(512), lds, (hipStream_t)stream, g);
  else
    hipLaunchKernelGGL(gemm_tn256_group_kernel<F16>, dim3(tiles), dim3(512), lds, (hipStream_t)stream, g);
  VMC_CHECK_LAUNCH();
  return 0;
}
