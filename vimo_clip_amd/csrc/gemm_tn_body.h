// Device body of the TN weight-gradient GEMM (see gemm_tn.hip for the design notes): shared by gemm_tn.hip (one problem per
// launch, token slices) and tfam_train.hip (all weight gradients of a TFAM layer in one grouped launch).
#pragma once
#include "gemm_common.h"

__device__ __forceinline__ uint32_t tn_lds_addr(const char* p) { return (uint32_t)(uintptr_t)(const VMC_LDS char*)p; }

// inline asm for the reason given in gemm8.hip: hipcc would drain vmcnt(0) in front of C++ LDS reads while LDS-DMA
// is in flight.  Completion is ordered by the explicit lgkmcnt wait in front of the MFMAs.
template <int IMM>
__device__ __forceinline__ void tn_read_tr(uint2& v, uint32_t addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM) : "memory");
}

constexpr int TN_SUB = 64 * 256;        // one sub-tile
constexpr int TN_STAGE = 3 * TN_SUB;    // 48 KiB
constexpr int TN_STAGES = 3;

// One 256 (n) x 128 (k) output tile (tn, tk) over the token steps [s0, s1) of 64 tokens; `slab` = which [N, K] (and [N]) slab of
// C / dbias the partial result goes to (0 when the token range is not sliced).  512 threads, TN_STAGES * TN_STAGE bytes of LDS.
template <typename T>
__device__ __forceinline__ void tn_tile_body(const uint16_t* __restrict__ dY, const uint16_t* __restrict__ X, float* __restrict__ C,
                                             float* __restrict__ dbias, int M, int N, int K, int lddy, int ldx, int tn, int tk, int s0,
                                             int s1, int slab, char* smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int r = lane & 15, g = lane >> 4, q = r >> 2, p = r & 3;
  const int n0 = tn * 256, k0 = tk * 128;

  // LDS-DMA sources: image chunk c = i*512 + tid of a sub-tile -> token row c>>4, slot c&15, source chunk slot^swz(row).
  // Columns past the matrix edge are clamped to the tile's first column (in bounds; those outputs are never stored);
  // token rows past M are clamped to M-1 and zeroed in LDS before use (tail stage only).
  const uint16_t* src[6];
  int srow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = i * 512 + tid, row = c >> 4, ch = (c & 15) ^ tn_swz(row);
    srow[i] = row;
    int col = n0 + ch * 8;
    src[i] = dY + (col < N ? col : n0);
    col = n0 + 128 + ch * 8;
    src[2 + i] = dY + (col < N ? col : n0);
    col = k0 + ch * 8;
    src[4 + i] = X + (col < K ? col : k0);
  }
  const int wave_lds = wave * 1024;   // 64 lanes x 16 B
  auto stage_in = [&](int step) {
    char* dst = smem + (step % TN_STAGES) * TN_STAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = min(step * 64 + srow[i], M - 1);
#pragma unroll
      for (int sub = 0; sub < 3; ++sub) {
        const size_t ld = sub < 2 ? (size_t)lddy : (size_t)ldx;
        __builtin_amdgcn_global_load_lds((const VMC_GLOBAL void*)(src[2 * sub + i] + (size_t)m * ld),
                                         (VMC_LDS void*)(dst + sub * TN_SUB + i * 8192 + wave_lds), 16, 0, 0);
      }
    }
  };

  // transposed-read offsets inside a stage: token row 8g + q (+4 for the second half), columns 16 t + 4 p of the wave's
  // 64-column window; ks (32 tokens) adds 8192 bytes as an immediate.
  uint32_t aoff[4][2], boff[4][2];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int row = 8 * g + q + 4 * e;
      const int cha = ((wn & 1) * 64 + 16 * t + 4 * p) >> 3, chb = (wk * 64 + 16 * t + 4 * p) >> 3;
      aoff[t][e] = (wn >> 1) * TN_SUB + row * 256 + ((cha ^ tn_swz(row)) << 4) + (p & 1) * 8;
      boff[t][e] = 2 * TN_SUB + row * 256 + ((chb ^ tn_swz(row)) << 4) + (p & 1) * 8;
    }

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // bias gradient = column sums of dY = dY^T . 1: the waves that own the first 64 X columns of the first K tile feed the
  // dY fragments they hold anyway into four more MFMAs against a ones operand (every output column then carries the sum).
  const bool do_bias = dbias != nullptr && tk == 0 && wk == 0;
  const uint4 ones = make_uint4(T::ONE_PAIR, T::ONE_PAIR, T::ONE_PAIR, T::ONE_PAIR);
  f32x4 bacc[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) bacc[a] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (s0 < s1) {
    // Ping-pong form of the stage loop (the schedule of gemm8.hip): waves w and w + 4 share a SIMD; the upper four run one
    // s_barrier behind the lower four, so while one wave of a SIMD issues its 16 (+4) MFMAs its partner issues the transposed
    // reads of its next 32 tokens and the stage DMA.  Two phases per 64-token stage:
    //   phase (st, ks): 16 ds_read_b64_tr of tokens 32 ks.. | ks = 0: DMA of stage st + 2, ks = 1: wait for stage st + 1 |
    //                   lgkmcnt(0) | barrier | MFMAs | barrier
    // lgkmcnt(0) sits in FRONT of the first barrier: every wave's reads of a ring slot have completed when any wave passes it,
    // so the slot of stage st - 1 (last read in phase (st-1, 1)) may be restaged one phase later, in phase (st, 0).  A stage is
    // waited for (counted vmcnt, each wave for its own DMA) in the phase before its first read; the barriers publish it.
    const int grp = wave >> 2;
    stage_in(s0);
    if (s0 + 1 < s1) stage_in(s0 + 1);
    if (s0 + 1 < s1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();      // stagger
    for (int st = s0; st < s1; ++st) {
      char* cur = smem + (st % TN_STAGES) * TN_STAGE;
      if (st * 64 + 64 > M) {                // tail stage: token rows past M become zeros in all three sub-tiles
        const int valid = M - st * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i)
          if (srow[i] >= valid) {
#pragma unroll
            for (int sub = 0; sub < 3; ++sub) *(uint4*)(cur + sub * TN_SUB + (i * 512 + tid) * 16) = make_uint4(0, 0, 0, 0);
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();        // two, so that both groups see the zeros before either reads (they run one apart)
      }
      __builtin_amdgcn_sched_barrier(0);
      const uint32_t base = tn_lds_addr(cur);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        uint2 af[4][2], bf[4][2];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            if (ks == 0) { tn_read_tr<0>(af[t][e], base + aoff[t][e]); tn_read_tr<0>(bf[t][e], base + boff[t][e]); }
            else { tn_read_tr<8192>(af[t][e], base + aoff[t][e]); tn_read_tr<8192>(bf[t][e], base + boff[t][e]); }
          }
        if (ks == 0) {
          if (st + 2 < s1) stage_in(st + 2);                 // into the ring slot of stage st - 1
        } else if (st + 1 < s1) {                            // stage st + 1 (issued in phase (st-1, 0)) must have landed
          if (st + 2 < s1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b)
            acc[a][b] = T::mfma16(make_uint4(bf[b][0].x, bf[b][0].y, bf[b][1].x, bf[b][1].y),
                                  make_uint4(af[a][0].x, af[a][0].y, af[a][1].x, af[a][1].y), acc[a][b]);      // (X, dY): 4 consecutive k per lane
        if (do_bias) {
#pragma unroll
          for (int a = 0; a < 4; ++a) bacc[a] = T::mfma16(ones, make_uint4(af[a][0].x, af[a][0].y, af[a][1].x, af[a][1].y), bacc[a]);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
      }
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();      // balance the stagger
  }
  // operands passed as (X, dY): the lane holds C[n = n0 + 64 wn + 16 a + r][k = k0 + 64 wk + 16 b + 4 g + j], four consecutive k of
  // one row -> one 16-byte store per accumulator (K % 8 == 0: a started group of four never crosses the edge); with (dY, X) the
  // same tile left as 64 four-byte stores per lane and the store issue, not the 128 MB of a TFAM step's gradients, set the time
  float* out = C + (size_t)slab * N * K;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int n = n0 + 64 * wn + 16 * a + r;
    if (n >= N) continue;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int k = k0 + 64 * wk + 16 * b + 4 * g;
      if (k < K) *(float4*)(out + (size_t)n * K + k) = make_float4(acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]);
    }
  }
  if (do_bias && g == 0) {        // every row of the ones product carries the column sums: row 0 (g = 0, j = 0), column r
    float* bo = dbias + (size_t)slab * N;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int n = n0 + 64 * wn + 16 * a + r;
      if (n < N) bo[n] = bacc[a][0];
    }
  }
}
