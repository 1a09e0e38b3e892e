// vmc_linear: C = epilogue(A @ W^T), 16-bit operands, fp32 MFMA accumulation (gfx950).
//
// Structure (cdna_hip_programming.md §5): block tile (16*MT*WM) x (64*WN), BK = 64, two LDS stages
// filled by global_load_lds_dwordx4 (LDS image lane-linear, XOR swizzle applied on the per-lane SOURCE
// address and on the ds_read_b128 address), one vmcnt(0)+barrier per K tile with the next tile's DMA
// in flight under the MFMAs.  Operands are passed to v_mfma_f32_16x16x32 as (W, X) so each lane owns
// 16 consecutive output columns of one row -> bias/residual/stores are 16-byte vectors.
#include "gemm_common.h"

template <int MT, int WM, int WN>
struct GemmCfg {
  static constexpr int BM = 16 * MT * WM;
  static constexpr int BN = 64 * WN;
  static constexpr int NT = 64 * WM * WN;
  static constexpr int A_BYTES = BM * 128;
  static constexpr int B_BYTES = BN * 128;
  static constexpr int STAGE = A_BYTES + B_BYTES;
  static constexpr int A_ITERS = BM * 8 / NT;
  static constexpr int B_ITERS = BN * 8 / NT;
  static constexpr int LDS = 2 * STAGE;
  static_assert(BM * 8 % NT == 0 && BN * 8 % NT == 0, "staging must divide evenly");
};

// One K tile: issue the LDS-DMA of the NEXT tile into `wr`, then MFMA over the tile in `rd`.
// `rd` / `wr` are __restrict__ so the compiler's waitcnt pass knows the ds_reads cannot alias the
// in-flight global_load_lds and does not drain vmcnt in front of them (checked in the .s).
template <typename T, int MT, int WM, int WN>
__device__ __forceinline__ void gemm_step(const char* __restrict__ rd, char* __restrict__ wr, bool do_stage, size_t koff,
                                          const char* const (&a_src)[GemmCfg<MT, WM, WN>::A_ITERS],
                                          const char* const (&b_src)[GemmCfg<MT, WM, WN>::B_ITERS], int wave_lds,
                                          const int (&xoff)[2], const int (&woff)[2][4], f32x4 (&acc)[MT][4]) {
  using Cfg = GemmCfg<MT, WM, WN>;
  if (do_stage) {
#pragma unroll
    for (int i = 0; i < Cfg::A_ITERS; ++i)
      __builtin_amdgcn_global_load_lds((const VMC_GLOBAL void*)(a_src[i] + koff),
                                       (VMC_LDS void*)(wr + i * Cfg::NT * 16 + wave_lds), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < Cfg::B_ITERS; ++i)
      __builtin_amdgcn_global_load_lds((const VMC_GLOBAL void*)(b_src[i] + koff),
                                       (VMC_LDS void*)(wr + Cfg::A_BYTES + i * Cfg::NT * 16 + wave_lds), 16, 0, 0);
  }
  // keep the DMA issue ahead of this tile's ds_reads/MFMAs (hipcc otherwise sinks it to the end of the
  // step, right in front of the vmcnt(0) that waits for it)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    uint4 wf[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) wf[nt] = *(const uint4*)(rd + woff[kk][nt]);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      uint4 xf = *(const uint4*)(rd + xoff[kk] + mt * 2048);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = T::mfma16(wf[nt], xf, acc[mt][nt]);
    }
  }
}

// inline-asm fragment read for the ring variant (same reason as gemm8.hip: no compiler-inserted vmcnt(0) while DMAs fly)
template <int IMM>
__device__ __forceinline__ void ring_read128(uint4& v, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM) : "memory");
}
template <int N>
__device__ __forceinline__ void ring_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// NS == 2: the two-stage loop above.  NS > 2: an NS-stage LDS ring with NS-1 K tiles in flight and counted vmcnt --
// for the latency-bound small problems (few workgroups per CU, e.g. the 256-row tail of a large GEMM), where one
// exposed L2/HBM round trip per K tile is the whole run time.
// KS > 1 (ring variant only): KS groups of WM x WN waves share one output tile and split its K tiles between them -- group s takes
// K tiles s, s + KS, ... through its own NS-stage ring -- and their accumulators are summed through LDS in the fixed order
// 0, 1, .., KS-1 before the one epilogue (deterministic).  For the problems whose run time is ONE tile's dependent K chain (the
// 256-row tails of the encoder's GEMMs: 16 .. 64 K tiles on 128 .. 256 workgroups): the chain is KS times shorter and KS times
// more K tiles are in flight.
// U > 1 (ring variant, KS == 1): U consecutive K tiles per barrier -- their fragments are all read before ONE lgkmcnt wait and the
// MFMAs then run in the unchanged K order (same bits as U = 1): the loop of a small tile is a barrier, a vmcnt wait and two LDS round
// trips per 64-deep K tile, which is what a latency-bound launch consists of.
template <typename T, int ACT, int MT, int WM, int WN, int NS = 2, int KS = 1, int U = 1>
__global__ void __launch_bounds__(64 * WM * WN * KS) gemm_kernel(const GemmArgs g) {
  using Cfg = GemmCfg<MT, WM, WN>;
  static_assert(KS == 1 || NS > 2, "K-slice groups run the ring loop");
  static_assert(U == 1 || (KS == 1 && NS > U + 1), "multi-tile steps: ring of more than U + 1 slots, no K-slice groups");
  extern __shared__ __attribute__((aligned(16))) char smem_all[];

  const int ks = KS > 1 ? (int)threadIdx.x / Cfg::NT : 0;          // this thread's K-slice group
  const int tid = KS > 1 ? (int)threadIdx.x - ks * Cfg::NT : (int)threadIdx.x;
  char* const smem = smem_all + (KS > 1 ? ks * NS * Cfg::STAGE : 0);
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 15, q = lane >> 4;

  const int tile = xcd_remap(blockIdx.x, g.tiles_m * g.tiles_n);
  const int tm = tile / g.tiles_n, tn = tile % g.tiles_n;
  const int m0 = tm * Cfg::BM, n0 = tn * Cfg::BN;

  // ---- per-thread staging sources (row pointers are loop-invariant, only k advances) ----
  const char* a_src[Cfg::A_ITERS];
  const char* b_src[Cfg::B_ITERS];
#pragma unroll
  for (int i = 0; i < Cfg::A_ITERS; ++i) {
    int row, ch;
    stage_src_x(i * Cfg::NT + tid, row, ch);
    int grow = min(m0 + row, g.M - 1);
    a_src[i] = g.A + ((size_t)grow * g.lda + ch * 8) * 2;
  }
#pragma unroll
  for (int i = 0; i < Cfg::B_ITERS; ++i) {
    int row, ch;
    stage_src_w(i * Cfg::NT + tid, row, ch);
    int grow = min(n0 + row, g.N - 1);
    b_src[i] = g.W + ((size_t)grow * g.ldw + ch * 8) * 2;
  }
  const int wave_lds = wave * 1024;  // this wave's 64 x 16 B slice inside each NT*16-byte staging pass

  f32x4 acc[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment read offsets (bytes) inside a stage
  int xoff[2], woff[2][4];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    xoff[kk] = lds_off_x(wm * 16 * MT + r, 4 * kk + q);  // + mt * 2048
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) woff[kk][nt] = Cfg::A_BYTES + lds_off_w(wn * 64 + gemm_w_row(r, nt), 4 * kk + q);
  }

  // split-K: blockIdx.y owns K tiles [kt0, kt0 + nkt) (k_slices == 1: the whole K range)
  const int nkt_all = g.K / 64;
  const int per_slice = (nkt_all + g.k_slices - 1) / g.k_slices;
  const int kt0 = blockIdx.y * per_slice;
  const int nkt_slice = min(nkt_all, kt0 + per_slice) - kt0;
  if (nkt_slice <= 0) return;
  // K-slice groups: group ks owns K tiles ks, ks + KS, ... of the range; every group runs the loop (and its barriers) as often as
  // group 0, a group without a tile in the last round skips its DMA and MFMAs
  const int nkt = KS > 1 ? (nkt_slice - ks + KS - 1) / KS : nkt_slice;
  const int nkt_loop = KS > 1 ? (nkt_slice + KS - 1) / KS : nkt_slice;
#pragma unroll
  for (int i = 0; i < Cfg::A_ITERS; ++i) a_src[i] += (size_t)(kt0 + ks) * 128;
#pragma unroll
  for (int i = 0; i < Cfg::B_ITERS; ++i) b_src[i] += (size_t)(kt0 + ks) * 128;
  if constexpr (NS > 2) {
    constexpr int LOADS = Cfg::A_ITERS + Cfg::B_ITERS;      // LDS-DMA instructions per thread per stage
    static_assert(LOADS * (NS - 2) <= 63, "vmcnt field");
    auto stage_in = [&](int t) {
      char* dst = smem + (t % NS) * Cfg::STAGE;
      const size_t koff = (size_t)t * 128 * KS;
#pragma unroll
      for (int i = 0; i < Cfg::A_ITERS; ++i)
        __builtin_amdgcn_global_load_lds((const VMC_GLOBAL void*)(a_src[i] + koff), (VMC_LDS void*)(dst + i * Cfg::NT * 16 + wave_lds), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < Cfg::B_ITERS; ++i)
        __builtin_amdgcn_global_load_lds((const VMC_GLOBAL void*)(b_src[i] + koff),
                                         (VMC_LDS void*)(dst + Cfg::A_BYTES + i * Cfg::NT * 16 + wave_lds), 16, 0, 0);
    };
    if constexpr (U > 1) {
      constexpr int F = NS - U;                 // K tiles in flight at the top of a step (nkt % U == 0: launcher)
      static_assert(LOADS * (F - U > 0 ? F - U : 0) <= 63, "vmcnt field");
#pragma unroll
      for (int t = 0; t < F; ++t)
        if (t < nkt) stage_in(t);
      for (int kt = 0; kt < nkt; kt += U) {
        // tiles kt .. kt+U-1 must have landed; the younger ones (a multiple of U, at most F - U) may stay in flight
        const int younger = min(F - U, nkt - kt - U);
        if (younger >= F - U) ring_wait_vm<LOADS * (F - U > 0 ? F - U : 0)>();
        else if (F - 2 * U > 0 && younger >= F - 2 * U) ring_wait_vm<LOADS * (F - 2 * U > 0 ? F - 2 * U : 0)>();
        else ring_wait_vm<0>();
        __builtin_amdgcn_s_barrier();            // every wave sees these tiles and is done reading the U before them ...
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (kt + F + u < nkt) stage_in(kt + F + u);      // ... whose ring slots the new DMAs overwrite
        uint4 wf[U][2][4], xf[U][2][MT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t base = (uint32_t)(uintptr_t)(const VMC_LDS char*)(smem + ((kt + u) % NS) * Cfg::STAGE);
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) ring_read128<0>(wf[u][kk][nt], base + (uint32_t)woff[kk][nt]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) ring_read128<0>(xf[u][kk][mt], base + (uint32_t)(xoff[kk] + mt * 2048));
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = T::mfma16(wf[u][kk][nt], xf[u][kk][mt], acc[mt][nt]);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
      if (t < nkt) stage_in(t);
    for (int kt = 0; kt < nkt_loop; ++kt) {
      // stage kt must have landed; up to NS-2 younger stages may stay in flight
      const int younger = max(0, min(NS - 2, nkt - 1 - kt));
      if (younger >= NS - 2) ring_wait_vm<LOADS * (NS - 2)>();
      else if (NS > 3 && younger == NS - 3) ring_wait_vm<LOADS * (NS - 3 > 0 ? NS - 3 : 0)>();
      else if (NS > 4 && younger == NS - 4) ring_wait_vm<LOADS * (NS - 4 > 0 ? NS - 4 : 0)>();
      else ring_wait_vm<0>();
      __builtin_amdgcn_s_barrier();            // every wave sees stage kt and is done reading stage kt-1 ...
      __builtin_amdgcn_sched_barrier(0);
      if (kt + NS - 1 < nkt) stage_in(kt + NS - 1);   // ... whose ring slot the new DMA overwrites
      if (KS > 1 && kt >= nkt) continue;       // this group has no K tile in the last round (the barrier above is still taken)
      const uint32_t base = (uint32_t)(uintptr_t)(const VMC_LDS char*)(smem + (kt % NS) * Cfg::STAGE);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        uint4 wf[4], xf[MT];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) ring_read128<0>(wf[nt], base + (uint32_t)woff[kk][nt]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) ring_read128<0>(xf[mt], base + (uint32_t)(xoff[kk] + mt * 2048));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = T::mfma16(wf[nt], xf[mt], acc[mt][nt]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    }  // U == 1
  } else {
  char* const buf0 = smem;
  char* const buf1 = smem + Cfg::STAGE;
  // prologue: tile 0 -> buf0
#pragma unroll
  for (int i = 0; i < Cfg::A_ITERS; ++i)
    __builtin_amdgcn_global_load_lds((const VMC_GLOBAL void*)(a_src[i]), (VMC_LDS void*)(buf0 + i * Cfg::NT * 16 + wave_lds), 16, 0, 0);
#pragma unroll
  for (int i = 0; i < Cfg::B_ITERS; ++i)
    __builtin_amdgcn_global_load_lds((const VMC_GLOBAL void*)(b_src[i]),
                                     (VMC_LDS void*)(buf0 + Cfg::A_BYTES + i * Cfg::NT * 16 + wave_lds), 16, 0, 0);
  int kt = 0;
  for (; kt + 1 < nkt; kt += 2) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    gemm_step<T, MT, WM, WN>(buf0, buf1, true, (size_t)(kt + 1) * 128, a_src, b_src, wave_lds, xoff, woff, acc);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    gemm_step<T, MT, WM, WN>(buf1, buf0, kt + 2 < nkt, (size_t)(kt + 2) * 128, a_src, b_src, wave_lds, xoff, woff, acc);
  }
  if (kt < nkt) {  // odd tile count: last tile sits in buf0
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    gemm_step<T, MT, WM, WN>(buf0, buf1, false, 0, a_src, b_src, wave_lds, xoff, woff, acc);
  }
  }  // NS == 2

  if constexpr (KS > 1) {      // sum the groups' accumulators in the order 0, 1, .., KS-1 (group 0 keeps the result and runs the epilogue)
    __syncthreads();           // every group is done with its ring: the exchange buffer aliases it
    float4* const xch = (float4*)smem_all;                              // [KS-1][Cfg::NT threads][MT*4] float4
    if (ks > 0) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          xch[((size_t)(ks - 1) * (MT * 4) + mt * 4 + nt) * Cfg::NT + tid] = make_float4(acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3]);
    }
    __syncthreads();
    if (ks > 0) return;
#pragma unroll
    for (int s = 0; s < KS - 1; ++s)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const float4 v = xch[((size_t)s * (MT * 4) + mt * 4 + nt) * Cfg::NT + tid];
          acc[mt][nt][0] += v.x; acc[mt][nt][1] += v.y; acc[mt][nt][2] += v.z; acc[mt][nt][3] += v.w;
        }
  }
  // ---- epilogue: lane owns C[row][col0 .. col0+15] for each mt ----
  const int col0 = n0 + wn * 64 + 16 * q;
  if (g.k_slices > 1) {   // split-K: slice s writes its partial tile into slab s of the workspace (plain 16-byte stores)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = m0 + wm * 16 * MT + 16 * mt + r;
      if (row >= g.M) continue;
      float* dst = (float*)g.C + ((size_t)blockIdx.y * g.M + row) * g.ldc + col0;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        if (col0 + 4 * nt < g.N) *(float4*)(dst + 4 * nt) = make_float4(acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3]);
    }
    return;
  }
  const bool vec8 = ((g.N & 7) == 0) && ((g.ldc & 7) == 0) && (g.res == nullptr || (g.ldres & 7) == 0);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = m0 + wm * 16 * MT + 16 * mt + r;
    if (row >= g.M) continue;
    const int orow = g.out_row_group ? row + row / g.out_row_group + 1 : row;
    const int rrow = g.res_row_mod ? row % g.res_row_mod : orow;
    float v[16];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) v[4 * nt + j] = acc[mt][nt][j];
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
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
      for (int c4 = 0; c4 < 4; ++c4)
        if (col0 + 4 * c4 < g.N) *(float4*)(dst + 4 * c4) = make_float4(v[4 * c4], v[4 * c4 + 1], v[4 * c4 + 2], v[4 * c4 + 3]);
    } else {
      uint16_t* dst = (uint16_t*)g.C + (size_t)orow * g.ldc + col0;
      if (vec8) {
#pragma unroll
        for (int c8 = 0; c8 < 2; ++c8)
          if (col0 + 8 * c8 < g.N)
            *(uint4*)(dst + 8 * c8) = make_uint4(pack2<T>(v[8 * c8], v[8 * c8 + 1]), pack2<T>(v[8 * c8 + 2], v[8 * c8 + 3]),
                                                 pack2<T>(v[8 * c8 + 4], v[8 * c8 + 5]), pack2<T>(v[8 * c8 + 6], v[8 * c8 + 7]));
      } else {
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4)
          if (col0 + 4 * c4 < g.N)
            *(uint2*)(dst + 4 * c4) = make_uint2(pack2<T>(v[4 * c4], v[4 * c4 + 1]), pack2<T>(v[4 * c4 + 2], v[4 * c4 + 3]));
      }
    }
  }
}

template <typename T, int ACT, int MT, int WM, int WN, int NS = 2, int KS = 1, int U = 1>
static int launch_cfg(GemmArgs& g, hipStream_t stream) {
  using Cfg = GemmCfg<MT, WM, WN>;
  auto kern = gemm_kernel<T, ACT, MT, WM, WN, NS, KS, U>;
  if (U > 1 && (g.k_slices > 1 || (g.K / 64) % U != 0)) return VMC_E_ARG;
  constexpr int XCH = (KS - 1) * Cfg::NT * MT * 4 * 16;                 // accumulator exchange of the K-slice groups (aliases the rings)
  constexpr int LDS = KS * NS * Cfg::STAGE > XCH ? KS * NS * Cfg::STAGE : XCH;
  static_assert(LDS <= 160 * 1024, "LDS");
  if (KS > 1 && g.k_slices > 1) return VMC_E_ARG;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  g.tiles_m = (g.M + Cfg::BM - 1) / Cfg::BM;
  g.tiles_n = (g.N + Cfg::BN - 1) / Cfg::BN;
  hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n, g.k_slices), dim3(Cfg::NT * KS), LDS, stream, g);
  VMC_CHECK_LAUNCH();
  return 0;
}

template <typename T, int ACT>
static int launch_shape(GemmArgs& g, hipStream_t stream) {
  // Tile choice: the 256x256 tile needs >= ~1 tile per CU to pay; smaller problems take smaller tiles
  // so the grid still covers the 256 CUs (TFAM/head GEMMs have M = B*16 or B rows).
  const long t256 = (long)((g.M + 255) / 256) * ((g.N + 255) / 256);
  const long t128 = (long)((g.M + 127) / 128) * ((g.N + 127) / 128);
  if constexpr (ACT == VMC_ACT_NONE) {      // builder A/B switch for the mid-size regime (profiles/README.md round 3)
    static const int force = getenv("VMC_GEMM_CFG") ? atoi(getenv("VMC_GEMM_CFG")) : 0;
    switch (force) {
      case 1: return launch_cfg<T, ACT, 4, 2, 2, 3>(g, stream);      // 128 x 128, 3-stage ring
      case 2: return launch_cfg<T, ACT, 4, 2, 2, 4>(g, stream);      // 128 x 128, 4-stage ring
      case 3: return launch_cfg<T, ACT, 4, 2, 1, 4>(g, stream);      // 128 x 64, 4-stage ring
      case 4: return launch_cfg<T, ACT, 2, 2, 2, 4>(g, stream);      // 64 x 128, 4-stage ring
      case 5: return launch_cfg<T, ACT, 2, 2, 1, 4>(g, stream);      // 64 x 64, 4-stage ring
      case 6: return launch_cfg<T, ACT, 4, 2, 2>(g, stream);         // 128 x 128, two stages (today's mid-size kernel)
      case 7: return launch_cfg<T, ACT, 4, 2, 1, 6>(g, stream);      // 128 x 64, 6-stage ring
      case 8: return launch_cfg<T, ACT, 2, 2, 2, 6>(g, stream);      // 64 x 128, 6-stage ring
      case 9: return launch_cfg<T, ACT, 8, 2, 2>(g, stream);         // 256 x 128, two stages
      case 10: return launch_cfg<T, ACT, 8, 2, 2, 3>(g, stream);     // 256 x 128, 3-stage ring
      case 11: return launch_cfg<T, ACT, 4, 2, 4>(g, stream);        // 128 x 256, two stages
      default: break;
    }
  }
  if (t256 >= 192) return launch_cfg<T, ACT, 8, 2, 4>(g, stream);
  if (t128 >= 128) return launch_cfg<T, ACT, 4, 2, 2>(g, stream);
  // 64x64 tiles: latency-bound -> 4-stage ring.  A workgroup streams its (BM + 64) x K operand bytes at the ~70 GB/s one CU pulls
  // from L2, so a long K chain on few workgroups is bound by that: when 64-row tiles give at most 128 workgroups (the 256-row tail
  // of c_proj: 4 x 16), 32-row tiles put twice as many CUs on it (same per-row arithmetic and K order: same bits; 27.3 -> 21.8 us at
  // K = 4096, 9.6 -> 7.9 us at K = 1024; 16-row single-wave tiles were slower again: 25.5 / 8.8 us).
  // K >= 1024 on at most one workgroup per CU is ONE tile's dependent K chain: K-slice groups inside the workgroup (gemm_kernel's KS)
  // shorten it.  Measured on the encoder's 256-row tails (us, KS = 1 -> routed): out_proj 8.5 -> 6.7 and c_proj (K = 4096) 22.8 -> 15.4
  // with four groups on 3-stage rings over 32-row tiles; qkv 10.6 -> 8.4 and c_fc 11.2 -> 9.6 with two groups on 4-stage rings over 64^2
  // tiles; two groups / four stages on the 32-row tiles and two groups / three stages on 64^2 were slower (7.9 / 20.4, 8.7 / 9.7);
  // 128 x 768 x 2048 (TFAM ffn.3 at B = 8, per-op path) 12.5 -> 8.2; encoder step 40.42 -> 40.07 ms (+0.9 %).  NOT routed by default
  // (VMC_GEMM_KS=1 turns it on): the groups sum the K tiles in another order than the persistent kernel, so a row's bits would depend
  // on whether it falls into a GEMM's 256-row tail -- the encoder's outputs are bit-identical under any batch split today
  // (tests/test_gpu_encoder.py), and 0.9 % does not buy that back.
  static const bool ks_on = getenv("VMC_GEMM_KS") && atoi(getenv("VMC_GEMM_KS")) != 0;
  const long t64 = (long)((g.M + 63) / 64) * ((g.N + 63) / 64);
  // K tiles per barrier (gemm_kernel's U; same K order, same bits).  Measured on the encoder's 256-row tails and the TFAM small-batch
  // shapes (us, U = 1 -> 4 on an 8-slot ring): 10.5 -> 9.9 (qkv), 7.8 -> 7.0, 8.5 -> 7.7 (out_proj), 22.7 -> 20.1 (c_proj, K = 4096),
  // 11.1 = 11.1 (c_fc), 128 x 768 x 2048 12.4 -> 10.8; U = 2 on six slots: no gain.  An 8-slot ring of 64^2 tiles is 128 KiB (one
  // workgroup per CU), so only problems of at most one workgroup per CU take it.  VMC_GEMM_U=1 turns it off (builder A/B switch).
  static const int mt_sw = getenv("VMC_GEMM_U") ? atoi(getenv("VMC_GEMM_U")) : 4;
  const bool u_ok = g.k_slices == 1 && g.K >= 512 && t64 <= 256;
  if (t64 <= 128 && g.K >= 1024) {
    if (ks_on && g.k_slices == 1) return launch_cfg<T, ACT, 1, 2, 1, 3, 4>(g, stream);
    if (mt_sw == 2 && u_ok && (g.K / 64) % 2 == 0) return launch_cfg<T, ACT, 1, 2, 1, 6, 1, 2>(g, stream);
    if (mt_sw == 4 && u_ok && (g.K / 64) % 4 == 0) return launch_cfg<T, ACT, 1, 2, 1, 8, 1, 4>(g, stream);
    return launch_cfg<T, ACT, 1, 2, 1, 4>(g, stream);
  }
  if (ks_on && g.K >= 1024 && g.k_slices == 1 && t64 <= 256) return launch_cfg<T, ACT, 2, 2, 1, 4, 2>(g, stream);
  if (mt_sw == 2 && u_ok && (g.K / 64) % 2 == 0) return launch_cfg<T, ACT, 2, 2, 1, 6, 1, 2>(g, stream);
  if (mt_sw == 4 && u_ok && (g.K / 64) % 4 == 0) return launch_cfg<T, ACT, 2, 2, 1, 8, 1, 4>(g, stream);
  return launch_cfg<T, ACT, 2, 2, 1, 4>(g, stream);
}

template <typename T>
static int launch_act(GemmArgs& g, int act, hipStream_t stream) {
  switch (act) {
    case VMC_ACT_NONE: return launch_shape<T, VMC_ACT_NONE>(g, stream);
    case VMC_ACT_QUICKGELU: return launch_shape<T, VMC_ACT_QUICKGELU>(g, stream);
    case VMC_ACT_GELU_ERF: return launch_shape<T, VMC_ACT_GELU_ERF>(g, stream);
    case VMC_ACT_RELU: return launch_shape<T, VMC_ACT_RELU>(g, stream);
  }
  return VMC_E_ARG;
}

static int linear_impl(const void* A, const void* W, const float* bias, const void* res, void* C, void* Z,
                       int M, int N, int K, int lda, int ldw, int ldc, int ldres, int ldz,
                       int act, float alpha, int out_dtype, int res_dtype, int out_row_group, int res_row_mod,
                       int dtype16, int variant, void* stream);

extern "C" int vmc_linear(const void* A, const void* W, const float* bias, const void* res, void* C,
                          int M, int N, int K, int lda, int ldw, int ldc, int ldres,
                          int act, float alpha, int out_dtype, int res_dtype, int out_row_group, int res_row_mod,
                          int dtype16, void* stream) {
  return linear_impl(A, W, bias, res, C, nullptr, M, N, K, lda, ldw, ldc, ldres, 0, act, alpha, out_dtype, res_dtype, out_row_group,
                     res_row_mod, dtype16, VMC_GEMM_DEFAULT, stream);
}

extern "C" int vmc_linear_variant(const void* A, const void* W, const float* bias, const void* res, void* C,
                                  int M, int N, int K, int lda, int ldw, int ldc, int ldres,
                                  int act, float alpha, int out_dtype, int res_dtype, int out_row_group, int res_row_mod,
                                  int dtype16, int variant, void* stream) {
  if (variant < 0 || variant >= VMC_GEMM_VARIANTS) return VMC_E_ARG;
  return linear_impl(A, W, bias, res, C, nullptr, M, N, K, lda, ldw, ldc, ldres, 0, act, alpha, out_dtype, res_dtype, out_row_group,
                     res_row_mod, dtype16, variant, stream);
}

extern "C" int vmc_linear_preact(const void* A, const void* W, const float* bias, const void* res, void* C, void* Z,
                                 int M, int N, int K, int lda, int ldw, int ldc, int ldres, int ldz,
                                 int act, float alpha, int out_dtype, int res_dtype, int out_row_group, int res_row_mod,
                                 int dtype16, void* stream) {
  return linear_impl(A, W, bias, res, C, Z, M, N, K, lda, ldw, ldc, ldres, ldz, act, alpha, out_dtype, res_dtype, out_row_group,
                     res_row_mod, dtype16, VMC_GEMM_DEFAULT, stream);
}

static int linear_impl(const void* A, const void* W, const float* bias, const void* res, void* C, void* Z,
                       int M, int N, int K, int lda, int ldw, int ldc, int ldres, int ldz,
                       int act, float alpha, int out_dtype, int res_dtype, int out_row_group, int res_row_mod,
                       int dtype16, int variant, void* stream) {
  if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0) return VMC_E_ARG;
  if (Z && (ldz < N || (ldz % 4) || ((uintptr_t)Z & 15) || out_row_group)) return VMC_E_ARG;
  if (dtype16 != VMC_BF16 && dtype16 != VMC_F16) return VMC_E_DTYPE;
  if (K % 64 != 0 || N % 4 != 0) return VMC_E_SHAPE;
  if (lda % 8 || ldw % 8 || ldc % 4 || (res && (ldres % 4))) return VMC_E_ALIGN;
  if (((uintptr_t)A | (uintptr_t)W | (uintptr_t)C | (uintptr_t)res | (uintptr_t)bias) & 15) return VMC_E_ALIGN;
  if (lda < K || ldw < K || ldc < N) return VMC_E_ARG;
  if (out_dtype != VMC_F32 && out_dtype != dtype16) return VMC_E_DTYPE;
  if (res && res_dtype != VMC_F32 && res_dtype != dtype16) return VMC_E_DTYPE;
  GemmArgs g;
  g.A = (const char*)A; g.W = (const char*)W; g.bias = bias; g.res = (const char*)res; g.C = (char*)C;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = ldw; g.ldc = ldc; g.ldres = ldres;
  g.alpha = alpha; g.out_f32 = (out_dtype == VMC_F32); g.res_f32 = (res_dtype == VMC_F32);
  g.out_row_group = out_row_group; g.res_row_mod = res_row_mod;
  g.tiles_m = g.tiles_n = 0;
  g.k_slices = 1;
  g.zout = (char*)Z; g.ldz = ldz;
  // large problems with an even K-tile count take the 8-phase 256x256 kernel (gemm8.hip);
  // variant VMC_GEMM_TWOSTAGE forces the two-stage kernels (A/B measurements through vmc_linear_variant).
  g.variant = variant;
  const long t256 = (long)((M + 255) / 256) * ((N + 255) / 256);
  // Between 129 and 191 tiles of 256x256 the 128x128 kernel would need a second round of its 512 resident workgroups (t128 > 512)
  // while the 8-phase kernel still finishes in one: M = 4096, N = 2304, K = 768 (TFAM qkv at 256 clips) 31.6 -> 21.2 us; at <= 128
  // tiles the two are equal (20.8 / 20.0 us at 96 tiles) and below 96 the small tiles win (profiles/README.md, round 2).
  const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
  const bool big = t256 >= 192 || (t256 > 128 && t128 > 512);
  if (variant != VMC_GEMM_TWOSTAGE && big && (K % 128) == 0) {
    // Round quantisation: T tiles on 256 CUs cost ceil(T/256) tile-times.  When the last, partial round holds only a few
    // tiles (ViT-L/14: 257 x tn tiles -> tn tiles in a round of their own: +25 % at tn = 4; student ViT-B/32: 100 x 3 tiles ->
    // 44 tiles in a second round), the tile rows that do not fit the full rounds go to the small-tile kernels, which spread
    // them over the whole chip, in a second launch.
    const int tm = (M + 255) / 256, tn = (N + 255) / 256;
    // whole tile rows that fit in the full rounds (a last main round may leave up to tn - 1 CUs idle); the rest is the tail
    const int rounds = (int)(t256 / 256);
    const int main_rows = rounds > 0 ? (rounds * 256) / tn : 0;
    const long tail = (long)(tm - main_rows) * tn;              // tiles handed to the small-tile kernel
    if (variant != VMC_GEMM_NO_TAIL_SPLIT && !out_row_group && !res_row_mod && main_rows > 0 && main_rows < tm && t256 % 256 != 0 &&
        tail <= (rounds >= 2 ? 64 : 96)) {
      const int m_main = main_rows * 256;
      GemmArgs t = g;
      g.M = m_main;
      int rc = vmc_gemm8_launch(g, act, dtype16, (hipStream_t)stream);
      if (rc) return rc;
      t.M = M - m_main;
      t.A += (size_t)m_main * lda * 2;
      t.C += (size_t)m_main * ldc * (t.out_f32 ? 4 : 2);
      if (t.res) t.res += (size_t)m_main * ldres * (t.res_f32 ? 4 : 2);
      if (t.zout) t.zout += (size_t)m_main * ldz * 2;
      if (dtype16 == VMC_BF16) return launch_act<BF16>(t, act, (hipStream_t)stream);
      return launch_act<F16>(t, act, (hipStream_t)stream);
    }
    return vmc_gemm8_launch(g, act, dtype16, (hipStream_t)stream);
  }
  if (dtype16 == VMC_BF16) return launch_act<BF16>(g, act, (hipStream_t)stream);
  return launch_act<F16>(g, act, (hipStream_t)stream);
}


// out[w] = sum_p partial[p*W + w], float4 per thread, fixed order (deterministic)
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out, int P, size_t W4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < W4; i += (size_t)gridDim.x * blockDim.x) {
    float4 a = ((const float4*)partial)[i];
    for (int p = 1; p < P; ++p) {
      const float4 v = ((const float4*)partial)[(size_t)p * W4 + i];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    ((float4*)out)[i] = a;
  }
}

static int splitk_slices(int M, int N, int K) {
  const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
  int slices = (768 + tiles - 1) / tiles;              // ~3 workgroups per CU
  const int nkt = K / 64;
  if (slices > nkt / 4) slices = nkt / 4;              // at least 4 K tiles per slice
  if (slices < 1) slices = 1;
  const int per = (nkt + slices - 1) / slices;         // the kernel's per-slice tile count ...
  return (nkt + per - 1) / per;                        // ... and no empty trailing slice (its slab would stay unwritten)
}

// Weight-gradient GEMM (K8): C[M,N] (f32, contiguous) = A[M,K] @ W[N,K]^T with a long contraction (K = tokens) and a
// small output: 128x128 tiles x K slices so that the grid covers the chip; every slice writes a partial slab into the
// workspace and a second kernel sums the slabs in a fixed order (deterministic, no atomics).
extern "C" size_t vmc_linear_splitk_workspace_bytes(int M, int N, int K) {
  const int s = splitk_slices(M, N, K);
  return s > 1 ? (size_t)s * M * N * sizeof(float) : 0;
}

extern "C" int vmc_linear_splitk_f32(const void* A, const void* W, float* C, int M, int N, int K, int lda, int ldw, void* workspace,
                                     size_t workspace_bytes, int dtype16, void* stream) {
  if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0) return VMC_E_ARG;
  if (K % 64 != 0 || N % 4 != 0) return VMC_E_SHAPE;
  if (lda % 8 || ldw % 8) return VMC_E_ALIGN;
  if (((uintptr_t)A | (uintptr_t)W | (uintptr_t)C | (uintptr_t)workspace) & 15) return VMC_E_ALIGN;
  const int slices = splitk_slices(M, N, K);
  if (slices > 1 && (!workspace || workspace_bytes < vmc_linear_splitk_workspace_bytes(M, N, K))) return VMC_E_ARG;
  GemmArgs g;
  g.A = (const char*)A; g.W = (const char*)W; g.bias = nullptr; g.res = nullptr;
  g.C = slices > 1 ? (char*)workspace : (char*)C;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = ldw; g.ldc = N; g.ldres = 0;
  g.alpha = 1.0f; g.out_f32 = 1; g.res_f32 = 0; g.out_row_group = 0; g.res_row_mod = 0;
  g.k_slices = slices;
  g.variant = VMC_GEMM_DEFAULT;
  g.zout = nullptr; g.ldz = 0;
  hipStream_t s = (hipStream_t)stream;
  int rc = dtype16 == VMC_BF16 ? launch_cfg<BF16, VMC_ACT_NONE, 4, 2, 2>(g, s)
           : dtype16 == VMC_F16 ? launch_cfg<F16, VMC_ACT_NONE, 4, 2, 2>(g, s) : VMC_E_DTYPE;
  if (rc || slices == 1) return rc;
  const size_t W4 = (size_t)M * N / 4;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid_for(W4, 256)), dim3(256), 0, s, (const float*)workspace, C, slices, W4);
  VMC_CHECK_LAUNCH();
  return 0;
}


