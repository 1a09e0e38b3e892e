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

template <typename T, int ACT>
__global__ void __launch_bounds__(512, 2) gemm8_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int r = lane & 15, q = lane >> 4;

  // Tile order: each XCD walks a contiguous id range (xcd_remap); ids run over column GROUPS of 4 tiles, rows
  // inside a group.  The 32 tiles an XCD runs at once are then 8 row panels x 4 column panels: the 4 W panels
  // (2 MB at K = 1024) stay resident in that XCD's 4 MB L2 for the whole sweep and only A panels stream.
  const int tile = xcd_remap(blockIdx.x, g.tiles_m * g.tiles_n);
  int tm, tn;
  {
    constexpr int GC = 4;      // widths 2 / 8 / 16 measured within 0.5 % (2-3 % slower on qkv): profiles/README.md round 2
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
  // Interior tile writing 16-bit rows without residual / row remap (qkv, c_fc, dgrad): branch-free epilogue, the bias of
  // this lane's 2 x 8 columns loaded once instead of once per row (the generic path below cannot hoist it past the stores).
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

template <typename T>
static int g8_act(GemmArgs& g, int act, hipStream_t s) {
  switch (act) {
    case VMC_ACT_NONE: return g8_launch<T, VMC_ACT_NONE>(g, s);
    case VMC_ACT_QUICKGELU: return g8_launch<T, VMC_ACT_QUICKGELU>(g, s);
    case VMC_ACT_GELU_ERF: return g8_launch<T, VMC_ACT_GELU_ERF>(g, s);
    case VMC_ACT_RELU: return g8_launch<T, VMC_ACT_RELU>(g, s);
  }
  return VMC_E_ARG;
}

int vmc_gemm8_launch(GemmArgs& g, int act, int dtype16, hipStream_t stream) {
  if ((g.K & 127) != 0) return VMC_E_SHAPE;  // K tiles are consumed in pairs
  if (dtype16 == VMC_BF16) return g8_act<BF16>(g, act, stream);
  if (dtype16 == VMC_F16) return g8_act<F16>(g, act, stream);
  return VMC_E_DTYPE;
}
