// 8-phase 256x256x64 GEMM for the large ViT/TFAM linears (gfx950), after the structure described in
// cdna_hip_programming.md §5 "The 256^2 8-phase template":
//   * 8 waves (2 x 4); waves w and w+4 share a SIMD.  The wm=1 half runs ONE s_barrier behind the wm=0
//     half, so on every SIMD one wave is in its MFMA cluster while its partner issues ds_reads and the
//     LDS-DMA for later tiles (ping-pong; each phase = {reads, stage, barrier, 16 MFMA, barrier}).
//   * LDS = 8 half-tile slots of 16 KiB (A0,A1,B0,B1 for the even and the odd K tile).  One half-tile is
//     staged per phase with 2 global_load_lds_dwordx4 per thread; four half-tiles stay in flight across
//     the barriers (counted vmcnt(8), never 0 in the loop).
//   * every wave owns 64 rows of each A half and 32 columns of each B half (tile_index.h), so all waves
//     share the same deadlines: A0/B0 in phase 0, B1 in phase 1, A1 in phase 2 of a K tile.
// Schedule per K tile t (slot parity t&1), s(X) = stage half-tile X:
//   phase 0: read B0,A0(t)  s(B1(t+1))  wait   MFMA (mh0,nh0)
//   phase 1: read B1(t)     s(A1(t+1))  wait   MFMA (mh0,nh1)
//   phase 2: read A1(t)     s(A0(t+2))         MFMA (mh1,nh1)
//   phase 3:                s(B0(t+2))  wait   MFMA (mh1,nh0)
// RAW: a half-tile is read one phase after the vmcnt(8) that retires it (each wave waits for its own DMA,
// the barrier publishes it).  WAR: a slot is restaged >= 2 phases after its last ds_read.
#include "gemm_common.h"

#define G8_SLOT 16384

template <typename T>
struct G8Frags {
  uint4 a[4][2];   // [mt][kk] current A half (64 rows of this wave)
  uint4 b0[2][2];  // [nt][kk] B half 0 (kept from phase 0 to phase 3)
  uint4 b1[2][2];  // [nt][kk] B half 1
};

__device__ __forceinline__ void g8_stage(char* slot, const char* const (&src)[2], size_t koff, int wave_lds) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
    __builtin_amdgcn_global_load_lds((const VMC_GLOBAL void*)(src[i] + koff), (VMC_LDS void*)(slot + i * 8192 + wave_lds), 16, 0, 0);
}

// Fragment reads are inline asm on purpose: hipcc's waitcnt pass cannot prove that a ds_read does not alias an
// in-flight LDS-DMA and would drain vmcnt(0) in front of every read group (seen in the .s with plain C++
// loads, even with __restrict__ slots).  The asm reads are invisible to it; their completion is ordered by
// the explicit `s_waitcnt lgkmcnt(0)` + sched_barrier(0) in g8_mma (cdna_hip_programming.md §5.4 rule 18).
template <int IMM>
__device__ __forceinline__ void lds_read128(uint4& v, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM) : "memory");
}
__device__ __forceinline__ uint32_t lds_addr(const char* p) { return (uint32_t)(uintptr_t)(const VMC_LDS char*)p; }

__device__ __forceinline__ void g8_read_a(const char* slot, const int (&xoff)[2], uint4 (&a)[4][2]) {
  const uint32_t base = lds_addr(slot);
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const uint32_t ad = base + (uint32_t)xoff[kk];
    lds_read128<0>(a[0][kk], ad);
    lds_read128<2048>(a[1][kk], ad);
    lds_read128<4096>(a[2][kk], ad);
    lds_read128<6144>(a[3][kk], ad);
  }
}
__device__ __forceinline__ void g8_read_b(const char* slot, const int (&woff)[2][2], uint4 (&b)[2][2]) {
  const uint32_t base = lds_addr(slot);
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) lds_read128<0>(b[nt][kk], base + (uint32_t)woff[kk][nt]);
}

template <typename T>
__device__ __forceinline__ void g8_mma(f32x4 (&acc)[4][2], const uint4 (&a)[4][2], const uint4 (&b)[2][2]) {
  __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = T::mfma16(b[nt][kk], a[mt][kk], acc[mt][nt]);
  __builtin_amdgcn_s_setprio(0);
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
}

#define G8_WAIT8() asm volatile("s_waitcnt vmcnt(8)" ::: "memory")
// First K tile of a persistent output tile: the previous tile's epilogue stores (ns per wave, counted exactly)
// sit in the vmcnt queue BEHIND the already-issued LDS-DMAs of this tile and in front of the new ones, so they
// may stay outstanding: allow 8 + ns.  Any other count falls back to the conservative vmcnt(8).
__device__ __forceinline__ void g8_wait8_plus(int ns) {
  if (ns == 16) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else if (ns == 32) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}

// Two K tiles (even slots *e, odd slots *o).  k1/k2/k3 = byte offsets of tiles t+1, t+2, t+3 (clamped).
template <typename T>
__device__ __forceinline__ void g8_iter(char* __restrict__ A0e, char* __restrict__ A1e, char* __restrict__ B0e,
                                        char* __restrict__ B1e, char* __restrict__ A0o, char* __restrict__ A1o,
                                        char* __restrict__ B0o, char* __restrict__ B1o, const char* const (&sA0)[2],
                                        const char* const (&sA1)[2], const char* const (&sB0)[2], const char* const (&sB1)[2],
                                        size_t k1, size_t k2, size_t k3, int wave_lds, const int (&xoff)[2],
                                        const int (&woff)[2][2], f32x4 (&acc)[2][2][4][2], G8Frags<T>& f) {
  // ---- even tile ----
  g8_read_b(B0e, woff, f.b0); g8_read_a(A0e, xoff, f.a);
  g8_stage(B1o, sB1, k1, wave_lds); G8_WAIT8();
  g8_mma<T>(acc[0][0], f.a, f.b0);
  g8_read_b(B1e, woff, f.b1);
  g8_stage(A1o, sA1, k1, wave_lds); G8_WAIT8();
  g8_mma<T>(acc[0][1], f.a, f.b1);
  g8_read_a(A1e, xoff, f.a);
  g8_stage(A0e, sA0, k2, wave_lds);
  g8_mma<T>(acc[1][1], f.a, f.b1);
  g8_stage(B0e, sB0, k2, wave_lds); G8_WAIT8();
  g8_mma<T>(acc[1][0], f.a, f.b0);
  // ---- odd tile ----
  g8_read_b(B0o, woff, f.b0); g8_read_a(A0o, xoff, f.a);
  g8_stage(B1e, sB1, k2, wave_lds); G8_WAIT8();
  g8_mma<T>(acc[0][0], f.a, f.b0);
  g8_read_b(B1o, woff, f.b1);
  g8_stage(A1e, sA1, k2, wave_lds); G8_WAIT8();
  g8_mma<T>(acc[0][1], f.a, f.b1);
  g8_read_a(A1o, xoff, f.a);
  g8_stage(A0o, sA0, k3, wave_lds);
  g8_mma<T>(acc[1][1], f.a, f.b1);
  g8_stage(B0o, sB0, k3, wave_lds); G8_WAIT8();
  g8_mma<T>(acc[1][0], f.a, f.b0);
}

typedef unsigned int g8_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void g8_nt_store(const uint4& o, uint16_t* dst) {
  __builtin_nontemporal_store((g8_u32x4){o.x, o.y, o.z, o.w}, (g8_u32x4*)dst);
}

// ---- epilogue of the persistent kernel: one accumulator QUADRANT (128 rows x 128 columns of the tile; 32 registers, 4 stores per
// wave) at a time, bias from LDS, then the quadrant is zeroed for the next tile.  Addresses are a uniform tile base plus a 32-bit
// per-lane offset; the offset is laundered through an empty asm so that hipcc forms the four row offsets here instead of keeping
// sixteen of them live across the K loop.  Output stores are non-temporal (the tensor, 400 .. 540 MB per ViT-L/14 launch, is
// larger than the Infinity Cache and is read next by another kernel).
// Column halves of the persistent 16x16x32 kernel are INTERLEAVED: B half nh holds the tile's columns 64 j + 32 nh + [0, 32), j = 0..3,
// so a wave's two halves are the two 64-byte halves of the same 128-byte output line of every row (written one phase apart by the
// same wave) instead of lines 256 bytes apart whose other halves come from the neighbouring wave.
#define G8P_NH_BYTES 64u        /* byte offset of half 1 inside the wave's 128-byte row segment */
#define G8P_NH_BIAS 128         /* the same in the fp32 bias image */
template <typename T, int ACT, bool BIAS, bool ZOUT, int MH, int NH>
__device__ __forceinline__ void g8p_fin_quadrant(const GemmArgs& g, f32x4 (&aq)[4][2], char* ctile, char* ztile, uint32_t clane, uint32_t zlane,
                                                 uint32_t bias_ad) {
  float bv[8];
  if constexpr (BIAS) {
    uint4 braw[2];
    lds_read128<G8P_NH_BIAS * NH>(braw[0], bias_ad);
    lds_read128<G8P_NH_BIAS * NH + 16>(braw[1], bias_ad);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    bv[0] = __uint_as_float(braw[0].x); bv[1] = __uint_as_float(braw[0].y); bv[2] = __uint_as_float(braw[0].z); bv[3] = __uint_as_float(braw[0].w);
    bv[4] = __uint_as_float(braw[1].x); bv[5] = __uint_as_float(braw[1].y); bv[6] = __uint_as_float(braw[1].z); bv[7] = __uint_as_float(braw[1].w);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) bv[j] = 0.f;
  }
  uint32_t cl = clane;
  asm volatile("" : "+v"(cl));
  const uint32_t zl = cl;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    float v[8];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) v[4 * nt + j] = aq[mt][nt][j] + bv[4 * nt + j];
    if constexpr (ZOUT)      // pre-activation side output for the backward (training)
      *(uint4*)(ztile + (zl + (uint32_t)(128 * MH + 16 * mt) * (uint32_t)g.ldc * 2u + G8P_NH_BYTES * NH)) =
          make_uint4(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]), pack2<T>(v[4], v[5]), pack2<T>(v[6], v[7]));
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = g.alpha * apply_act<ACT>(v[j]);
    g8_nt_store(make_uint4(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]), pack2<T>(v[4], v[5]), pack2<T>(v[6], v[7])),
                (uint16_t*)(ctile + (cl + (uint32_t)(128 * MH + 16 * mt) * (uint32_t)g.ldc * 2u + G8P_NH_BYTES * NH)));
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) aq[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
}

// Which fast epilogue the whole problem qualifies for (tile interior checked by the caller).
#define G8_EPI_GENERIC 0
#define G8_EPI_STORE16 1
#define G8_EPI_RES32 2
VMC_HD int g8_epi_kind(const GemmArgs& g) {
  const bool vec8 = ((g.N & 7) == 0) && ((g.ldc & 7) == 0) && (g.res == nullptr || (g.ldres & 7) == 0);
  if (!vec8 || g.out_row_group) return G8_EPI_GENERIC;
  if (!g.out_f32 && !g.res && (!g.zout || (g.ldz & 7) == 0)) return G8_EPI_STORE16;
  if (g.out_f32 && g.res && g.res_f32 && !g.zout && !g.res_row_mod) return G8_EPI_RES32;
  return G8_EPI_GENERIC;
}

// Tile order: ids run over column GROUPS of 4 tiles, rows inside a group.  The 32 tiles an XCD runs at once are then 8 row
// panels x 4 column panels: the 4 W panels (2 MB at K = 1024) stay resident in that XCD's 4 MB L2 for the whole sweep and
// only A panels stream.  Widths 2 / 8 / 16 measured within 0.5 % (2-3 % slower on qkv): profiles/README.md round 2.
__device__ __forceinline__ void g8_tile_coords(const GemmArgs& g, int tile, int& tm, int& tn) {
  const int GC = g.walk_gc > 0 ? g.walk_gc : 4;
  const int gsz = g.tiles_m * GC, nfull = g.tiles_n / GC;
  const int cg = tile / gsz;
  if (cg < nfull) {
    const int rem = tile - cg * gsz;
    tm = rem / GC;
    tn = cg * GC + rem % GC;
  } else {
    const int w = g.tiles_n - nfull * GC, rem = tile - nfull * gsz;
    tm = rem / w;
    tn = nfull * GC + rem % w;
  }
}

template <typename T, int ACT>
__global__ void __launch_bounds__(512, 2) gemm8_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int r = lane & 15, q = lane >> 4;

  int tm, tn;   // each XCD walks a contiguous id range (xcd_remap)
  g8_tile_coords(g, xcd_remap(blockIdx.x, g.tiles_m * g.tiles_n), tm, tn);
  const int m0 = tm * 256, n0 = tn * 256;

  const char *sA0[2], *sA1[2], *sB0[2], *sB1[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int row, ch;
    stage_src_x(i * 512 + tid, row, ch);
    sA0[i] = g.A + ((size_t)min(m0 + row, g.M - 1) * g.lda + ch * 8) * 2;
    sA1[i] = g.A + ((size_t)min(m0 + 128 + row, g.M - 1) * g.lda + ch * 8) * 2;
    stage_src_w8(i * 512 + tid, row, ch);
    sB0[i] = g.W + ((size_t)min(n0 + row, g.N - 1) * g.ldw + ch * 8) * 2;
    sB1[i] = g.W + ((size_t)min(n0 + 128 + row, g.N - 1) * g.ldw + ch * 8) * 2;
  }
  const int wave_lds = wave * 1024;
  int xoff[2], woff[2][2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    xoff[kk] = lds_off_x(64 * wm + r, 4 * kk + q);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) woff[kk][nt] = lds_off_w8(32 * wn + g8_w_row(r, nt), 4 * kk + q);
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

  char* const A0e = smem + 0 * G8_SLOT; char* const A1e = smem + 1 * G8_SLOT;
  char* const B0e = smem + 2 * G8_SLOT; char* const B1e = smem + 3 * G8_SLOT;
  char* const A0o = smem + 4 * G8_SLOT; char* const A1o = smem + 5 * G8_SLOT;
  char* const B0o = smem + 6 * G8_SLOT; char* const B1o = smem + 7 * G8_SLOT;

  const int nkt = g.K >> 6;  // even, >= 2 (checked by the launcher)
  const int last = nkt - 1;
  // prologue: same issue order as the steady state so the vmcnt(8) accounting holds from the first phase
  g8_stage(A0e, sA0, 0, wave_lds); g8_stage(B0e, sB0, 0, wave_lds); g8_stage(B1e, sB1, 0, wave_lds);
  g8_stage(A1e, sA1, 0, wave_lds); g8_stage(A0o, sA0, 128, wave_lds); g8_stage(B0o, sB0, 128, wave_lds);
  G8_WAIT8();
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();  // stagger the second wave of every SIMD by one barrier

  G8Frags<T> f;
  for (int t = 0; t < nkt; t += 2) {
    const size_t k1 = (size_t)min(t + 1, last) * 128, k2 = (size_t)min(t + 2, last) * 128, k3 = (size_t)min(t + 3, last) * 128;
    g8_iter<T>(A0e, A1e, B0e, B1e, A0o, A1o, B0o, B1o, sA0, sA1, sB0, sB1, k1, k2, k3, wave_lds, xoff, woff, acc, f);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tail DMAs (clamped re-loads of the last tile) must land before exit
  if (wm == 0) __builtin_amdgcn_s_barrier();        // balance the stagger barrier

  const bool vec8 = gemm_vec8_ok(g);
  if (vec8 && !g.out_f32 && !g.res && !g.out_row_group && (!g.zout || (g.ldz & 7) == 0) && m0 + 256 <= g.M && n0 + 256 <= g.N) {
    float bv[2][8];
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
      const int col = n0 + 128 * nh + 32 * wn + 8 * q;
      if (g.bias) {
        const float4 b0 = *(const float4*)(g.bias + col), b1 = *(const float4*)(g.bias + col + 4);
        bv[nh][0] = b0.x; bv[nh][1] = b0.y; bv[nh][2] = b0.z; bv[nh][3] = b0.w;
        bv[nh][4] = b1.x; bv[nh][5] = b1.y; bv[nh][6] = b1.z; bv[nh][7] = b1.w;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) bv[nh][j] = 0.f;
      }
    }
    uint16_t* cbase = (uint16_t*)g.C + (size_t)(m0 + 64 * wm + r) * g.ldc + n0 + 32 * wn + 8 * q;
    uint16_t* zbase = g.zout ? (uint16_t*)g.zout + (size_t)(m0 + 64 * wm + r) * g.ldz + n0 + 32 * wn + 8 * q : nullptr;
    // Output stores are non-temporal: the tensor (400 .. 540 MB per ViT-L/14 launch) is larger than the Infinity Cache and is read
    // next by another kernel; +1.2 .. 1.5 % on the qkv / c_fc shapes.  Staging the tile through LDS for whole-row stores was
    // measured too and does not pay (profiles/README.md, round 2).
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
          float v[8];
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[4 * nt + j] = acc[mh][nh][mt][nt][j] + bv[nh][4 * nt + j];
          if (zbase)     // pre-activation side output for the backward (training)
            *(uint4*)(zbase + (size_t)(128 * mh + 16 * mt) * g.ldz + 128 * nh) =
                make_uint4(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]), pack2<T>(v[4], v[5]), pack2<T>(v[6], v[7]));
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = g.alpha * apply_act<ACT>(v[j]);
          g8_nt_store(make_uint4(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]), pack2<T>(v[4], v[5]), pack2<T>(v[6], v[7])),
                      cbase + (size_t)(128 * mh + 16 * mt) * g.ldc + 128 * nh);
        }
    return;
  }
  // Interior tile accumulating into an fp32 residual stream (x += A W^T + b; out_proj / c_proj): the residual rows of
  // four tile rows are fetched together (8 x 32 B per lane in flight), then added and stored; no per-row branches.
  if (vec8 && g.out_f32 && g.res && g.res_f32 && !g.zout && !g.out_row_group && !g.res_row_mod && m0 + 256 <= g.M && n0 + 256 <= g.N) {
    float bv[2][8];
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
      const int col = n0 + 128 * nh + 32 * wn + 8 * q;
#pragma unroll
      for (int j = 0; j < 8; ++j) bv[nh][j] = 0.f;
      if (g.bias) {
        const float4 b0 = *(const float4*)(g.bias + col), b1 = *(const float4*)(g.bias + col + 4);
        bv[nh][0] = b0.x; bv[nh][1] = b0.y; bv[nh][2] = b0.z; bv[nh][3] = b0.w;
        bv[nh][4] = b1.x; bv[nh][5] = b1.y; bv[nh][6] = b1.z; bv[nh][7] = b1.w;
      }
    }
    const size_t roff = (size_t)(m0 + 64 * wm + r);
    const int coff = n0 + 32 * wn + 8 * q;
    const float* rbase = (const float*)g.res + roff * g.ldres + coff;
    float* cbase = (float*)g.C + roff * g.ldc + coff;
#pragma unroll
    for (int mh = 0; mh < 2; ++mh) {
      float4 rr[4][2][2];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
          const float* src = rbase + (size_t)(128 * mh + 16 * mt) * g.ldres + 128 * nh;
          rr[mt][nh][0] = *(const float4*)src;
          rr[mt][nh][1] = *(const float4*)(src + 4);
        }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
          float* dst = cbase + (size_t)(128 * mh + 16 * mt) * g.ldc + 128 * nh;
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            const float4 x = rr[mt][nh][nt];
            float4 o;
            o.x = x.x + g.alpha * apply_act<ACT>(acc[mh][nh][mt][nt][0] + bv[nh][4 * nt + 0]);
            o.y = x.y + g.alpha * apply_act<ACT>(acc[mh][nh][mt][nt][1] + bv[nh][4 * nt + 1]);
            o.z = x.z + g.alpha * apply_act<ACT>(acc[mh][nh][mt][nt][2] + bv[nh][4 * nt + 2]);
            o.w = x.w + g.alpha * apply_act<ACT>(acc[mh][nh][mt][nt][3] + bv[nh][4 * nt + 3]);
            *(float4*)(dst + 4 * nt) = o;
          }
        }
    }
    return;
  }
#pragma unroll
  for (int mh = 0; mh < 2; ++mh)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int row = m0 + 128 * mh + 64 * wm + 16 * mt + r;
#pragma unroll
      for (int nh = 0; nh < 2; ++nh) {
        float v[8];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int j = 0; j < 4; ++j) v[4 * nt + j] = acc[mh][nh][mt][nt][j];
        gemm_epilogue_row<T, ACT, 8>(g, row, n0 + 128 * nh + 32 * wn + 8 * q, v, vec8);
      }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Persistent form of the same 8-phase tile loop: one workgroup per CU walks logical block ids b, b + gridDim.x, ... (the order
// a one-tile-per-workgroup grid is dispatched in).  Two things are taken off the ~8 us a tile spends outside its K loop:
//  * the last K iteration of a tile stages the first six half-tiles of the NEXT tile instead of the clamped re-loads the
//    one-tile kernel issues there, so the next tile starts with its operands in LDS (no dispatch, no exposed first-load latency);
//  * the epilogue runs per accumulator quadrant in the phases where the quadrant is final and not yet rewritten, in the slot of
//    a phase where this wave would otherwise sit at the barrier while its SIMD partner runs the MFMA cluster: quadrant (0,0) is
//    final after phase 0 of the last K tile and is rewritten by phase 0 of the next tile, (0,1) one phase later, ... :
//        last K tile, phase 1: (0,0)   phase 2: (0,1)   phase 3: (1,1)      next tile, phase 0: (1,0)
//    Conversions and stores overlap the partner's MFMAs; no accumulator is live twice and none is zeroed on the critical path.
// Operand addresses are buffer offsets: voffset = the lane's fixed offset, soffset = the panel (SGPR), so walking tiles costs no
// VGPRs and no 64-bit per-lane pointer exists (with global_load_lds hipcc sometimes materialises four of them and spills).
// Whole interior tiles, 16-bit output with bias, K >= 256 only (the launcher checks).
struct G8Panel { uint32_t a, b; };   // byte offsets of the A / W panels of one K tile inside their operands (uniform)

__device__ __forceinline__ void g8p_stage(const __amdgpu_buffer_rsrc_t rs, char* slot, uint32_t panel, const uint32_t (&off)[2],
                                          int wave_lds) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (VMC_LDS void*)(slot + i * 8192 + wave_lds), 16, off[i], panel, 0, 0);
}

// The vmcnt queue is in order and also carries the epilogue stores and the bias DMA, so "all but the four youngest half-tiles
// have landed" is a different count per iteration kind.  With QS = stores per quadrant and wave (4; 8 with the pre-activation
// side output) and NB = 1 if a bias DMA sits in front of phase 0's stage of a tile's first iteration:
//   G8P_STEADY  every wait 8
//   G8P_ROLL    last iteration of a tile: QS stores follow the stage of odd phases 1, 2, 3    -> odd waits 8, 8 + QS, 8 + 3 QS
//   G8P_HEAD    first iteration of the workgroup's first tile                                  -> even waits 8 + NB (x3)
//   G8P_HEAD    (pending) first iteration of a later tile: 3 QS stores + bias DMA + QS stores (quadrant (1,0), after phase 0's
//               stage) are younger than the awaited half-tile -> even waits 8 + NB + 4 QS (x2), 8 + NB + 2 QS; odd phase 0: 8 + QS
// (derivation: count the operations younger than the half-tile the wait must retire)
// The iteration kind is a template parameter (three copies of the loop body: head, steady, roll): selecting the count at run
// time costs 3-5 scalar branches per wait, ~200 cycles per K tile (measured: -6.7 % at K = 4096).  Only the head copy, run once
// per tile, keeps a run-time choice (first tile of the workgroup or not).
enum { G8P_STEADY = 0, G8P_ROLL = 1, G8P_HEAD = 2 };
template <int KIND, int S, int R, int F, int N>
__device__ __forceinline__ void g8p_wait(bool pending) {
  if constexpr (KIND == G8P_STEADY) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S) : "memory");
  else if constexpr (KIND == G8P_ROLL) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(R) : "memory");
  else if (pending) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
  else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(F) : "memory");
}

// Two K tiles.  p1 / p2 / p3 = panels of K tiles t+1, t+2, t+3 (t+2 and t+3 may already belong to the next output tile).
// fin_c / fin_b = uniform output base and LDS bias address of the tile whose quadrants are written here: the current tile in a
// G8P_ROLL iteration, the previous one in a pending G8P_HEAD (clane = this lane's offset inside an output tile).
template <typename T, int ACT, bool BIAS, bool ZOUT, int KIND>
__device__ __forceinline__ void g8p_iter(const GemmArgs& g, char* __restrict__ A0e, char* __restrict__ A1e, char* __restrict__ B0e,
                                         char* __restrict__ B1e, char* __restrict__ A0o, char* __restrict__ A1o,
                                         char* __restrict__ B0o, char* __restrict__ B1o, const __amdgpu_buffer_rsrc_t ra,
                                         const __amdgpu_buffer_rsrc_t rb, const G8Panel& p1, const G8Panel& p2,
                                         const G8Panel& p3, uint32_t HA, uint32_t HB, const uint32_t (&oa)[2], const uint32_t (&ob)[2],
                                         int wave_lds, const int (&xoff)[2], const int (&woff)[2][2], f32x4 (&acc)[2][2][4][2],
                                         G8Frags<T>& f, bool pending, char* fin_c, char* fin_z, uint32_t clane, uint32_t zlane, uint32_t fin_b) {
  constexpr int QS = ZOUT ? 8 : 4, NB = BIAS ? 1 : 0;
  // ---- even tile ----
  g8_read_b(B0e, woff, f.b0); g8_read_a(A0e, xoff, f.a);
  g8p_stage(rb, B1o, p1.b + HB, ob, wave_lds);
  if (KIND == G8P_HEAD && pending) g8p_fin_quadrant<T, ACT, BIAS, ZOUT, 1, 0>(g, acc[1][0], fin_c, fin_z, clane, zlane, fin_b);
  g8p_wait<KIND, 8, 8, 8 + NB, 8 + NB + 4 * QS>(pending);
  g8_mma<T>(acc[0][0], f.a, f.b0);
  g8_read_b(B1e, woff, f.b1);
  g8p_stage(ra, A1o, p1.a + HA, oa, wave_lds); g8p_wait<KIND, 8, 8, 8 + NB, 8 + NB + 4 * QS>(pending);
  g8_mma<T>(acc[0][1], f.a, f.b1);
  g8_read_a(A1e, xoff, f.a);
  g8p_stage(ra, A0e, p2.a, oa, wave_lds);
  g8_mma<T>(acc[1][1], f.a, f.b1);
  g8p_stage(rb, B0e, p2.b, ob, wave_lds); g8p_wait<KIND, 8, 8, 8 + NB, 8 + NB + 2 * QS>(pending);
  g8_mma<T>(acc[1][0], f.a, f.b0);
  // ---- odd tile ----
  g8_read_b(B0o, woff, f.b0); g8_read_a(A0o, xoff, f.a);
  g8p_stage(rb, B1e, p2.b + HB, ob, wave_lds); g8p_wait<KIND, 8, 8, 8, 8 + QS>(pending);
  g8_mma<T>(acc[0][0], f.a, f.b0);
  g8_read_b(B1o, woff, f.b1);
  g8p_stage(ra, A1e, p2.a + HA, oa, wave_lds);
  if constexpr (KIND == G8P_ROLL) g8p_fin_quadrant<T, ACT, BIAS, ZOUT, 0, 0>(g, acc[0][0], fin_c, fin_z, clane, zlane, fin_b);
  g8p_wait<KIND, 8, 8 + QS, 8, 8>(pending);
  g8_mma<T>(acc[0][1], f.a, f.b1);
  g8_read_a(A1o, xoff, f.a);
  g8p_stage(ra, A0o, p3.a, oa, wave_lds);
  if constexpr (KIND == G8P_ROLL) g8p_fin_quadrant<T, ACT, BIAS, ZOUT, 0, 1>(g, acc[0][1], fin_c, fin_z, clane, zlane, fin_b);
  g8_mma<T>(acc[1][1], f.a, f.b1);
  g8p_stage(rb, B0o, p3.b, ob, wave_lds);
  if constexpr (KIND == G8P_ROLL) g8p_fin_quadrant<T, ACT, BIAS, ZOUT, 1, 1>(g, acc[1][1], fin_c, fin_z, clane, zlane, fin_b);
  g8p_wait<KIND, 8, 8 + 3 * QS, 8, 8>(pending);
  g8_mma<T>(acc[1][0], f.a, f.b0);
}

template <typename T, int ACT, bool BIAS, bool ZOUT>
__global__ void __launch_bounds__(512, 2) gemm8p_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int r = lane & 15, q = lane >> 4;
  const int ntiles = g.tiles_m * g.tiles_n;

  uint32_t oa[2], ob[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int row, ch;
    stage_src_x(i * 512 + tid, row, ch);
    oa[i] = (uint32_t)(row * g.lda + ch * 8) * 2u;
    stage_src_w8(i * 512 + tid, row, ch);
    ob[i] = (uint32_t)((64 * (row >> 5) + (row & 31)) * g.ldw + ch * 8) * 2u;        // half nh: W rows 64 j + 32 nh + [0, 32)
  }
  const uint32_t HA = 128u * (uint32_t)g.lda * 2u, HB = 32u * (uint32_t)g.ldw * 2u;    // second half-tile (A: rows + 128, W: rows + 32)
  const uint32_t TA = 2 * HA, TB = 256u * (uint32_t)g.ldw * 2u;                         // one tile row / column panel
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, (int)((size_t)g.M * g.lda * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)g.W, 0, (int)((size_t)g.N * g.ldw * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rbias = __builtin_amdgcn_make_buffer_rsrc((void*)(BIAS ? g.bias : (const float*)g.W), 0, g.N * 4, 0x00020000);
  const int wave_lds = wave * 1024;
  int xoff[2], woff[2][2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    xoff[kk] = lds_off_x(64 * wm + r, 4 * kk + q);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) woff[kk][nt] = lds_off_w8(32 * wn + g8_w_row(r, nt), 4 * kk + q);
  }
  char* const A0e = smem + 0 * G8_SLOT; char* const A1e = smem + 1 * G8_SLOT;
  char* const B0e = smem + 2 * G8_SLOT; char* const B1e = smem + 3 * G8_SLOT;
  char* const A0o = smem + 4 * G8_SLOT; char* const A1o = smem + 5 * G8_SLOT;
  char* const B0o = smem + 6 * G8_SLOT; char* const B1o = smem + 7 * G8_SLOT;
  // The bias of a tile's 256 columns reaches its epilogue through LDS (1 KiB behind the operand slots, two buffers by tile
  // parity): a global load in the epilogue would make the compiler wait for it with a vmcnt that also covers the freshly
  // issued half-tiles (in-order counter) and expose their latency.  Each wave DMAs the 64 floats of its column quarter.
  char* const bias_lds = smem + 8 * G8_SLOT;
  const uint32_t bias_lane = (uint32_t)lane * 4u;
  const uint32_t bias_ad0 = lds_addr(bias_lds) + (uint32_t)(64 * wn + 8 * q) * 4u;
  const uint32_t clane = ((uint32_t)(64 * wm + r) * (uint32_t)g.ldc + 64 * wn + 8 * q) * 2u;
  const uint32_t zlane = clane;     // the side output has the row stride of the output (launcher: ldz == ldc); one VGPR fewer

  const int nkt = g.K >> 6;   // even, >= 4 (checked by the launcher)
  int bid = blockIdx.x, tm, tn;
  if (bid >= ntiles) return;  // fewer tiles than workgroups (192 .. 255 tiles: one tile each, the epilogue overlap is still worth it)
  g8_tile_coords(g, xcd_remap(bid, ntiles), tm, tn);
  G8Panel cur = {(uint32_t)tm * TA, (uint32_t)tn * TB};
  // prologue of the first tile: same issue order as the steady state so the vmcnt accounting holds from the first phase
  g8p_stage(ra, A0e, cur.a, oa, wave_lds); g8p_stage(rb, B0e, cur.b, ob, wave_lds); g8p_stage(rb, B1e, cur.b + HB, ob, wave_lds);
  g8p_stage(ra, A1e, cur.a + HA, oa, wave_lds); g8p_stage(ra, A0o, cur.a + 128, oa, wave_lds); g8p_stage(rb, B0o, cur.b + 128, ob, wave_lds);
  G8_WAIT8();
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();  // stagger the second wave of every SIMD by one barrier

  f32x4 acc[2][2][4][2];  // [mh][nh][mt][nt]; zero here, afterwards every quadrant epilogue leaves its quadrant zeroed
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int d = 0; d < 2; ++d) acc[a][b][c][d] = (f32x4){0.f, 0.f, 0.f, 0.f};

  G8Frags<T> f;
  int parity = 0;
  char* prev_c = nullptr;                     // the tile whose quadrant (1,0) is still to be written: output base ...
  char* prev_z = nullptr;                     // ... side-output base ...
  uint32_t prev_b = bias_ad0;                 // ... and LDS address of its bias
  for (;;) {
    const int nbid = bid + (int)gridDim.x;
    const bool has_next = nbid < ntiles;
    int ntm = tm, ntn = tn;
    if (has_next) g8_tile_coords(g, xcd_remap(nbid, ntiles), ntm, ntn);
    // after the last tile the roll-over stages re-load this tile's last K tiles (in bounds, never read)
    const G8Panel nxt = has_next ? G8Panel{(uint32_t)ntm * TA, (uint32_t)ntn * TB}
                                 : G8Panel{cur.a + (uint32_t)(nkt - 2) * 128u, cur.b + (uint32_t)(nkt - 2) * 128u};
    char* const here_c = g.C + ((size_t)tm * 256 * g.ldc + (size_t)tn * 256) * 2;
    char* const here_z = ZOUT ? g.zout + ((size_t)tm * 256 * g.ldc + (size_t)tn * 256) * 2 : nullptr;
    const uint32_t here_b = bias_ad0 + (uint32_t)parity * 1024u;

#define G8P_ITER(KIND, P1, P2, P3, PEND, FC, FZ, FB) \
  g8p_iter<T, ACT, BIAS, ZOUT, KIND>(g, A0e, A1e, B0e, B1e, A0o, A1o, B0o, B1o, ra, rb, P1, P2, P3, HA, HB, oa, ob, wave_lds, xoff, woff, acc, f, PEND, FC, FZ, \
                                     clane, zlane, FB)
    {  // head: K tiles 0, 1 (already staged); stages 1, 2, 3; the previous tile's quadrant (1,0) goes out in phase 0
      if constexpr (BIAS)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rbias, (VMC_LDS void*)(bias_lds + parity * 1024 + wn * 256), 4, bias_lane,
                                                 (uint32_t)(tn * 256 + 64 * wn) * 4u, 0, 0);
      const G8Panel p1 = {cur.a + 128u, cur.b + 128u}, p2 = {cur.a + 256u, cur.b + 256u}, p3 = {cur.a + 384u, cur.b + 384u};
      G8P_ITER(G8P_HEAD, p1, p2, p3, prev_c != nullptr, prev_c, prev_z, prev_b);
    }
    for (int t = 2; t + 2 < nkt; t += 2) {
      const G8Panel p1 = {cur.a + (uint32_t)(t + 1) * 128u, cur.b + (uint32_t)(t + 1) * 128u};
      const G8Panel p2 = {cur.a + (uint32_t)(t + 2) * 128u, cur.b + (uint32_t)(t + 2) * 128u};
      const G8Panel p3 = {cur.a + (uint32_t)(t + 3) * 128u, cur.b + (uint32_t)(t + 3) * 128u};
      G8P_ITER(G8P_STEADY, p1, p2, p3, false, here_c, here_z, here_b);
    }
    {  // roll: K tiles nkt-2, nkt-1; stages the next tile's K tiles 0 and 1; quadrants (0,0), (0,1), (1,1) of this tile go out
      const G8Panel p1 = {cur.a + (uint32_t)(nkt - 1) * 128u, cur.b + (uint32_t)(nkt - 1) * 128u};
      const G8Panel p3 = {nxt.a + 128u, nxt.b + 128u};
      G8P_ITER(G8P_ROLL, p1, nxt, p3, false, here_c, here_z, here_b);
    }
#undef G8P_ITER
    prev_c = here_c; prev_z = here_z; prev_b = here_b;
    if (!has_next) break;
    parity ^= 1;
    bid = nbid; tm = ntm; tn = ntn; cur = nxt;
  }
  g8p_fin_quadrant<T, ACT, BIAS, ZOUT, 1, 0>(g, acc[1][0], prev_c, prev_z, clane, zlane, prev_b);   // last tile's last quadrant
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tail DMAs must land before exit
  if (wm == 0) __builtin_amdgcn_s_barrier();        // balance the stagger barrier
}

// ---------------------------------------------------------------------------------------------------------------------------
// The persistent walk on v_mfma_f32_32x32x16 fragments (VERDICT r2 item 6 i): same tile, LDS images, staging, barriers and vmcnt
// accounting (4 / 8 stores per quadrant and wave, as the 16x16x32 form); 8 MFMAs of 32 cycles per phase instead of 16 of 16, i.e.
// half the operand-register reads per FLOP.  Lane (x = lane & 31, h = lane >> 5); fragment maps in tile_index.h (g8_w_row32).
template <typename T>
struct G8Frags32 {
  uint4 a[2][4];   // [mt][kk] current A half (64 rows of this wave as two 32-row tiles, four k-steps of 16)
  uint4 b0[4];     // [kk] B half 0 (this wave's 32 columns)
  uint4 b1[4];
};
// k-step kk of a fragment = logical chunk 2 kk + h = (2 kk) ^ h, so its swizzled offset is (offset of k-step 0) ^ (kk << 5)
__device__ __forceinline__ void g8_read_a32(const char* slot, int xoff0, uint4 (&a)[2][4]) {
  const uint32_t base = lds_addr(slot);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const uint32_t ad = base + (uint32_t)(xoff0 ^ (kk << 5));
    lds_read128<0>(a[0][kk], ad);
    lds_read128<4096>(a[1][kk], ad);
  }
}
__device__ __forceinline__ void g8_read_b32(const char* slot, int woff0, uint4 (&b)[4]) {
  const uint32_t base = lds_addr(slot);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) lds_read128<0>(b[kk], base + (uint32_t)(woff0 ^ (kk << 5)));
}
template <typename T>
__device__ __forceinline__ void g8_mma32(f32x16 (&acc)[2], const uint4 (&a)[2][4], const uint4 (&b)[4]) {
  __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) acc[mt] = T::mfma32(b[kk], a[mt][kk], acc[mt]);
  __builtin_amdgcn_s_setprio(0);
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
}

// quadrant epilogue: lane (x, h) holds rows 32 mt + x, columns 8 h + [0, 8) (registers 0..7) and 16 + 8 h + [0, 8) (8..15) of the
// wave's 64 x 32 window: 2 x 2 sixteen-byte stores, 32 contiguous bytes per row and instruction
template <typename T, int ACT, bool BIAS, bool ZOUT, int MH, int NH>
__device__ __forceinline__ void g8p_fin_quadrant32(const GemmArgs& g, f32x16 (&aq)[2], char* ctile, char* ztile, uint32_t clane,
                                                   uint32_t bias_ad) {
  float bv[16];
  if constexpr (BIAS) {
    uint4 braw[4];
    lds_read128<512 * NH>(braw[0], bias_ad);
    lds_read128<512 * NH + 16>(braw[1], bias_ad);
    lds_read128<512 * NH + 64>(braw[2], bias_ad);
    lds_read128<512 * NH + 80>(braw[3], bias_ad);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bv[4 * i + 0] = __uint_as_float(braw[i].x); bv[4 * i + 1] = __uint_as_float(braw[i].y);
      bv[4 * i + 2] = __uint_as_float(braw[i].z); bv[4 * i + 3] = __uint_as_float(braw[i].w);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 16; ++j) bv[j] = 0.f;
  }
  uint32_t cl = clane;
  asm volatile("" : "+v"(cl));
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = aq[mt][8 * o + j] + bv[8 * o + j];
      const uint32_t off = cl + (uint32_t)(128 * MH + 32 * mt) * (uint32_t)g.ldc * 2u + 256u * NH + 32u * o;
      if constexpr (ZOUT)
        *(uint4*)(ztile + off) = make_uint4(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]), pack2<T>(v[4], v[5]), pack2<T>(v[6], v[7]));
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = g.alpha * apply_act<ACT>(v[j]);
      g8_nt_store(make_uint4(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]), pack2<T>(v[4], v[5]), pack2<T>(v[6], v[7])),
                  (uint16_t*)(ctile + off));
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) aq[mt][j] = 0.f;
  }
}

template <typename T, int ACT, bool BIAS, bool ZOUT, int KIND>
__device__ __forceinline__ void g8p_iter32(const GemmArgs& g, char* __restrict__ A0e, char* __restrict__ A1e, char* __restrict__ B0e,
                                           char* __restrict__ B1e, char* __restrict__ A0o, char* __restrict__ A1o,
                                           char* __restrict__ B0o, char* __restrict__ B1o, const __amdgpu_buffer_rsrc_t ra,
                                           const __amdgpu_buffer_rsrc_t rb, const G8Panel& p1, const G8Panel& p2,
                                           const G8Panel& p3, uint32_t HA, uint32_t HB, const uint32_t (&oa)[2], const uint32_t (&ob)[2],
                                           int wave_lds, int xoff, int woff, f32x16 (&acc)[2][2][2], G8Frags32<T>& f, bool pending,
                                           char* fin_c, char* fin_z, uint32_t clane, uint32_t fin_b) {
  constexpr int QS = ZOUT ? 8 : 4, NB = BIAS ? 1 : 0;
  // ---- even tile ----
  g8_read_b32(B0e, woff, f.b0); g8_read_a32(A0e, xoff, f.a);
  g8p_stage(rb, B1o, p1.b + HB, ob, wave_lds);
  if (KIND == G8P_HEAD && pending) g8p_fin_quadrant32<T, ACT, BIAS, ZOUT, 1, 0>(g, acc[1][0], fin_c, fin_z, clane, fin_b);
  g8p_wait<KIND, 8, 8, 8 + NB, 8 + NB + 4 * QS>(pending);
  g8_mma32<T>(acc[0][0], f.a, f.b0);
  g8_read_b32(B1e, woff, f.b1);
  g8p_stage(ra, A1o, p1.a + HA, oa, wave_lds); g8p_wait<KIND, 8, 8, 8 + NB, 8 + NB + 4 * QS>(pending);
  g8_mma32<T>(acc[0][1], f.a, f.b1);
  g8_read_a32(A1e, xoff, f.a);
  g8p_stage(ra, A0e, p2.a, oa, wave_lds);
  g8_mma32<T>(acc[1][1], f.a, f.b1);
  g8p_stage(rb, B0e, p2.b, ob, wave_lds); g8p_wait<KIND, 8, 8, 8 + NB, 8 + NB + 2 * QS>(pending);
  g8_mma32<T>(acc[1][0], f.a, f.b0);
  // ---- odd tile ----
  g8_read_b32(B0o, woff, f.b0); g8_read_a32(A0o, xoff, f.a);
  g8p_stage(rb, B1e, p2.b + HB, ob, wave_lds); g8p_wait<KIND, 8, 8, 8, 8 + QS>(pending);
  g8_mma32<T>(acc[0][0], f.a, f.b0);
  g8_read_b32(B1o, woff, f.b1);
  g8p_stage(ra, A1e, p2.a + HA, oa, wave_lds);
  if constexpr (KIND == G8P_ROLL) g8p_fin_quadrant32<T, ACT, BIAS, ZOUT, 0, 0>(g, acc[0][0], fin_c, fin_z, clane, fin_b);
  g8p_wait<KIND, 8, 8 + QS, 8, 8>(pending);
  g8_mma32<T>(acc[0][1], f.a, f.b1);
  g8_read_a32(A1o, xoff, f.a);
  g8p_stage(ra, A0o, p3.a, oa, wave_lds);
  if constexpr (KIND == G8P_ROLL) g8p_fin_quadrant32<T, ACT, BIAS, ZOUT, 0, 1>(g, acc[0][1], fin_c, fin_z, clane, fin_b);
  g8_mma32<T>(acc[1][1], f.a, f.b1);
  g8p_stage(rb, B0o, p3.b, ob, wave_lds);
  if constexpr (KIND == G8P_ROLL) g8p_fin_quadrant32<T, ACT, BIAS, ZOUT, 1, 1>(g, acc[1][1], fin_c, fin_z, clane, fin_b);
  g8p_wait<KIND, 8, 8 + 3 * QS, 8, 8>(pending);
  g8_mma32<T>(acc[1][0], f.a, f.b0);
}

template <typename T, int ACT, bool BIAS, bool ZOUT>
__global__ void __launch_bounds__(512, 2) gemm8p32_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int x = lane & 31, h = lane >> 5;
  const int ntiles = g.tiles_m * g.tiles_n;

  uint32_t oa[2], ob[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int row, ch;
    stage_src_x(i * 512 + tid, row, ch);
    oa[i] = (uint32_t)(row * g.lda + ch * 8) * 2u;
    stage_src_w8(i * 512 + tid, row, ch);
    ob[i] = (uint32_t)(row * g.ldw + ch * 8) * 2u;
  }
  const uint32_t HA = 128u * (uint32_t)g.lda * 2u, HB = 128u * (uint32_t)g.ldw * 2u;
  const uint32_t TA = 2 * HA, TB = 2 * HB;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, (int)((size_t)g.M * g.lda * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)g.W, 0, (int)((size_t)g.N * g.ldw * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rbias = __builtin_amdgcn_make_buffer_rsrc((void*)(BIAS ? g.bias : (const float*)g.W), 0, g.N * 4, 0x00020000);
  const int wave_lds = wave * 1024;
  const int xoff = lds_off_x(64 * wm + x, h), woff = lds_off_w8(32 * wn + g8_w_row32(x), h);
  char* const A0e = smem + 0 * G8_SLOT; char* const A1e = smem + 1 * G8_SLOT;
  char* const B0e = smem + 2 * G8_SLOT; char* const B1e = smem + 3 * G8_SLOT;
  char* const A0o = smem + 4 * G8_SLOT; char* const A1o = smem + 5 * G8_SLOT;
  char* const B0o = smem + 6 * G8_SLOT; char* const B1o = smem + 7 * G8_SLOT;
  char* const bias_lds = smem + 8 * G8_SLOT;
  const uint32_t bias_lane = (uint32_t)lane * 4u;
  const uint32_t bias_ad0 = lds_addr(bias_lds) + (uint32_t)(32 * wn + 8 * h) * 4u;
  const uint32_t clane = ((uint32_t)(64 * wm + x) * (uint32_t)g.ldc + 32 * wn + 8 * h) * 2u;   // also the side output's (ldz == ldc)

  const int nkt = g.K >> 6;   // even, >= 4 (checked by the launcher)
  int bid = blockIdx.x, tm, tn;
  if (bid >= ntiles) return;
  g8_tile_coords(g, xcd_remap(bid, ntiles), tm, tn);
  G8Panel cur = {(uint32_t)tm * TA, (uint32_t)tn * TB};
  g8p_stage(ra, A0e, cur.a, oa, wave_lds); g8p_stage(rb, B0e, cur.b, ob, wave_lds); g8p_stage(rb, B1e, cur.b + HB, ob, wave_lds);
  g8p_stage(ra, A1e, cur.a + HA, oa, wave_lds); g8p_stage(ra, A0o, cur.a + 128, oa, wave_lds); g8p_stage(rb, B0o, cur.b + 128, ob, wave_lds);
  G8_WAIT8();
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();  // stagger the second wave of every SIMD by one barrier

  f32x16 acc[2][2][2];  // [mh][nh][mt]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[a][b][c][j] = 0.f;

  G8Frags32<T> f;
  int parity = 0;
  char* prev_c = nullptr;
  char* prev_z = nullptr;
  uint32_t prev_b = bias_ad0;
  for (;;) {
    const int nbid = bid + (int)gridDim.x;
    const bool has_next = nbid < ntiles;
    int ntm = tm, ntn = tn;
    if (has_next) g8_tile_coords(g, xcd_remap(nbid, ntiles), ntm, ntn);
    const G8Panel nxt = has_next ? G8Panel{(uint32_t)ntm * TA, (uint32_t)ntn * TB}
                                 : G8Panel{cur.a + (uint32_t)(nkt - 2) * 128u, cur.b + (uint32_t)(nkt - 2) * 128u};
    char* const here_c = g.C + ((size_t)tm * 256 * g.ldc + (size_t)tn * 256) * 2;
    char* const here_z = ZOUT ? g.zout + ((size_t)tm * 256 * g.ldc + (size_t)tn * 256) * 2 : nullptr;
    const uint32_t here_b = bias_ad0 + (uint32_t)parity * 1024u;

#define G8P_ITER32(KIND, P1, P2, P3, PEND, FC, FZ, FB) \
  g8p_iter32<T, ACT, BIAS, ZOUT, KIND>(g, A0e, A1e, B0e, B1e, A0o, A1o, B0o, B1o, ra, rb, P1, P2, P3, HA, HB, oa, ob, wave_lds, xoff, woff, acc, f, PEND, \
                                       FC, FZ, clane, FB)
    {
      if constexpr (BIAS)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rbias, (VMC_LDS void*)(bias_lds + parity * 1024 + wn * 256), 4, bias_lane,
                                                 (uint32_t)(tn * 256 + 64 * wn) * 4u, 0, 0);
      const G8Panel p1 = {cur.a + 128u, cur.b + 128u}, p2 = {cur.a + 256u, cur.b + 256u}, p3 = {cur.a + 384u, cur.b + 384u};
      G8P_ITER32(G8P_HEAD, p1, p2, p3, prev_c != nullptr, prev_c, prev_z, prev_b);
    }
    for (int t = 2; t + 2 < nkt; t += 2) {
      const G8Panel p1 = {cur.a + (uint32_t)(t + 1) * 128u, cur.b + (uint32_t)(t + 1) * 128u};
      const G8Panel p2 = {cur.a + (uint32_t)(t + 2) * 128u, cur.b + (uint32_t)(t + 2) * 128u};
      const G8Panel p3 = {cur.a + (uint32_t)(t + 3) * 128u, cur.b + (uint32_t)(t + 3) * 128u};
      G8P_ITER32(G8P_STEADY, p1, p2, p3, false, here_c, here_z, here_b);
    }
    {
      const G8Panel p1 = {cur.a + (uint32_t)(nkt - 1) * 128u, cur.b + (uint32_t)(nkt - 1) * 128u};
      const G8Panel p3 = {nxt.a + 128u, nxt.b + 128u};
      G8P_ITER32(G8P_ROLL, p1, nxt, p3, false, here_c, here_z, here_b);
    }
#undef G8P_ITER32
    prev_c = here_c; prev_z = here_z; prev_b = here_b;
    if (!has_next) break;
    parity ^= 1;
    bid = nbid; tm = ntm; tn = ntn; cur = nxt;
  }
  g8p_fin_quadrant32<T, ACT, BIAS, ZOUT, 1, 0>(g, acc[1][0], prev_c, prev_z, clane, prev_b);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (wm == 0) __builtin_amdgcn_s_barrier();
}

template <typename T, int ACT>
static int g8_launch(GemmArgs& g, hipStream_t stream) {
  auto kern = gemm8_kernel<T, ACT>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * G8_SLOT);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  g.tiles_m = (g.M + 255) / 256;
  g.tiles_n = (g.N + 255) / 256;
  hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n), dim3(512), 8 * G8_SLOT, stream, g);
  VMC_CHECK_LAUNCH();
  return 0;
}

template <typename T, int ACT, bool BIAS, bool ZOUT>
static int g8p32_launch(GemmArgs& g, hipStream_t stream) {
  auto kern = gemm8p32_kernel<T, ACT, BIAS, ZOUT>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * G8_SLOT + 2048);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  g.tiles_m = g.M / 256;
  g.tiles_n = g.N / 256;
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), 8 * G8_SLOT + 2048, stream, g);
  VMC_CHECK_LAUNCH();
  return 0;
}

template <typename T, int ACT, bool BIAS, bool ZOUT>
static int g8p_launch(GemmArgs& g, hipStream_t stream) {
  auto kern = gemm8p_kernel<T, ACT, BIAS, ZOUT>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * G8_SLOT + 2048);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  g.tiles_m = g.M / 256;
  g.tiles_n = g.N / 256;
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), 8 * G8_SLOT + 2048, stream, g);
  VMC_CHECK_LAUNCH();
  return 0;
}

// The persistent kernel takes problems made of whole 256x256 tiles (at least 192 of them, at least two K iterations) with a
// plain 16-bit output: the encoders' qkv / out_proj / c_fc / c_proj (bias), the same with the pre-activation side output of
// training (QuickGELU c_fc), and the bias-free input-gradient GEMMs.  Everything else -- fp32 residual epilogue, row remaps,
// other activation / side-output combinations -- runs one tile per workgroup.
static bool g8p_eligible(const GemmArgs& g) {
  if ((g.M & 255) || (g.N & 255) || (long)(g.M / 256) * (g.N / 256) < 192 || g.K < 256) return false;
  if ((size_t)g.M * g.lda * 2 >= (1ull << 31) || (size_t)g.N * g.ldw * 2 >= (1ull << 31)) return false;   // buffer descriptors / 32-bit offsets
  if ((size_t)255 * g.ldc * 2 + 512 >= (1ull << 31) || (g.zout && g.ldz != g.ldc)) return false;
  return g8_epi_kind(g) == G8_EPI_STORE16;
}

template <typename T, int ACT>
static int g8_pick(GemmArgs& g, hipStream_t s) {
  // instantiated combinations only (each is a ~2000-instruction kernel); erf-GELU spills in the persistent kernel, and scratch
  // accesses count in vmcnt -> one-tile kernel
  if (g.variant != VMC_GEMM_ONE_TILE && g8p_eligible(g)) {
    const bool b = g.bias != nullptr, z = g.zout != nullptr;
    static const bool env32 = getenv("VMC_GEMM_MFMA32") && atoi(getenv("VMC_GEMM_MFMA32")) != 0;      // builder A/B switch
    if ((env32 || g.variant == VMC_GEMM_MFMA32) && !z) {
      if constexpr (ACT == VMC_ACT_NONE) {
        if (b) return g8p32_launch<T, ACT, true, false>(g, s);
        return g8p32_launch<T, ACT, false, false>(g, s);
      } else if constexpr (ACT == VMC_ACT_QUICKGELU) {
        if (b) return g8p32_launch<T, ACT, true, false>(g, s);
      }
    }
    if constexpr (ACT == VMC_ACT_NONE) {
      if (b && !z) return g8p_launch<T, ACT, true, false>(g, s);
      if (!b && !z) return g8p_launch<T, ACT, false, false>(g, s);
    } else if constexpr (ACT == VMC_ACT_QUICKGELU) {
      if (b && !z) return g8p_launch<T, ACT, true, false>(g, s);
      if (b && z) return g8p_launch<T, ACT, true, true>(g, s);
    } else if constexpr (ACT == VMC_ACT_RELU) {
      if (b && !z) return g8p_launch<T, ACT, true, false>(g, s);
    }
  }
  return g8_launch<T, ACT>(g, s);
}

template <typename T>
static int g8_act(GemmArgs& g, int act, hipStream_t s) {
  switch (act) {
    case VMC_ACT_NONE: return g8_pick<T, VMC_ACT_NONE>(g, s);
    case VMC_ACT_QUICKGELU: return g8_pick<T, VMC_ACT_QUICKGELU>(g, s);
    case VMC_ACT_GELU_ERF: return g8_pick<T, VMC_ACT_GELU_ERF>(g, s);
    case VMC_ACT_RELU: return g8_pick<T, VMC_ACT_RELU>(g, s);
  }
  return VMC_E_ARG;
}

int vmc_gemm8_launch(GemmArgs& g, int act, int dtype16, hipStream_t stream) {
  if ((g.K & 127) != 0) return VMC_E_SHAPE;  // K tiles are consumed in pairs
  static const int walk_gc = getenv("VMC_GEMM_GC") ? atoi(getenv("VMC_GEMM_GC")) : 0;      // builder A/B switch (profiles/README.md)
  g.walk_gc = walk_gc;
  if (dtype16 == VMC_BF16) return g8_act<BF16>(g, act, stream);
  if (dtype16 == VMC_F16) return g8_act<F16>(g, act, stream);
  return VMC_E_DTYPE;
}
