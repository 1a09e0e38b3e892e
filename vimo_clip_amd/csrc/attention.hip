// Attention kernels (gfx950).
//   vmc_attention_vit_fwd : CLIP ViT self-attention, head_dim 64, no mask — MFMA, whole K/V head in LDS.
//   vmc_attention_fwd/bwd : generic masked attention in fp32 (TFAM self/cross attention; also the
//                           training backward of the ViT blocks).
#include "common.h"

// ==================================================================================================
// ViT forward.  One workgroup = one (frame, head); 4 waves; each wave owns 16-query tiles.
//   S^T = K Q^T  (mfma(K_frag, Q_frag)): lane (r = lane&15, q = lane>>4) holds, for query r, the keys
//   16 nt + 4 q + j  -> the whole softmax row of a query sits in 4 lanes (2 shuffles per reduction) and
//   the exponentiated accumulators are directly the B operand of O^T = V^T P^T with the k-slots
//   permuted (tile_index.h attn_pv_key); V^T fragments come from ds_read_b64_tr_b16.
// ==================================================================================================
// One chunk of NKS 32-key steps starting at step KS0 for a 16-query tile: S^T = K Q^T, running max update,
// rescale of the accumulators, P = exp2(c2 s - c2 m) packed to 16 bits, O^T += V^T P^T, rowsum += 1^T P^T.
// attention.hip is compiled with -fno-honor-nans (Makefile): fmaxf then lowers to bare v_max_f32 / v_max3_f32 instead of
// quieting each operand first (v_max_f32 x, x); scores are finite or -inf, never NaN.
__device__ __forceinline__ float vmax3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// NC > 0: the key count is the compile-time constant NC (the three CLIP geometries: 50 / 197 / 257 tokens), so the
// padded-key masks fold to constants -- only the one tile that straddles NC executes compares, tiles past it skip their
// MFMAs.  NC == 0: runtime N (every tile carries a wave-uniform test, which hipcc lowers to two v_cndmask per score).
template <typename T, int KS0, int NKS, int NC>
__device__ __forceinline__ void attn_chunk(const char* k_lds, const char* v_lds, int koff0, int koff1, const int (&voff)[4],
                                           const uint4 (&qf)[2], int Nrt, int q, float c2, float& m, f32x4 (&o)[4], f32x4& osum) {
  constexpr int NTC = 2 * NKS, NT0 = 2 * KS0;
  const int N = NC > 0 ? NC : Nrt;
  f32x4 s[NTC];
#pragma unroll
  for (int t = 0; t < NTC; ++t) {
    if (NC > 0 && 16 * (NT0 + t) >= NC) {      // tile of padded keys only
      s[t] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      continue;
    }
    s[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    s[t] = T::mfma16(*(const uint4*)(k_lds + koff0 + (NT0 + t) * 2048), qf[0], s[t]);
    s[t] = T::mfma16(*(const uint4*)(k_lds + koff1 + (NT0 + t) * 2048), qf[1], s[t]);
  }
#pragma unroll
  for (int t = 0; t < NTC; ++t)
    if (16 * (NT0 + t) + 16 > N && !(NC > 0 && 16 * (NT0 + t) >= NC)) {  // only the tile that straddles N is masked
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (16 * (NT0 + t) + 4 * q + j >= N) s[t][j] = -INFINITY;
    }
  float mx = vmax3(s[0][0], s[0][1], vmax3(s[0][2], s[0][3], -INFINITY));
#pragma unroll
  for (int t = 1; t < NTC; ++t) {
    if (NC > 0 && 16 * (NT0 + t) >= NC) continue;
    mx = vmax3(vmax3(mx, s[t][0], s[t][1]), s[t][2], s[t][3]);
  }
  mx = vmax3(mx, __shfl_xor(mx, 16, 64), -INFINITY);
  mx = vmax3(mx, __shfl_xor(mx, 32, 64), -INFINITY);
  const float mnew = vmax3(m, mx, -INFINITY);   // key 0 is always valid, so mnew is finite from the first chunk on
  if (KS0 > 0) {                            // rescale what the earlier chunk accumulated (exp2(-inf) never occurs here)
    const float alpha = __builtin_amdgcn_exp2f((m - mnew) * c2);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] *= alpha;
    osum *= alpha;
  }
  m = mnew;
  const float mc = mnew * c2;
  uint4 pf[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const f32x4 a = s[2 * ks], b = s[2 * ks + 1];
    pf[ks].x = pack2<T>(__builtin_amdgcn_exp2f(__builtin_fmaf(a[0], c2, -mc)), __builtin_amdgcn_exp2f(__builtin_fmaf(a[1], c2, -mc)));
    pf[ks].y = pack2<T>(__builtin_amdgcn_exp2f(__builtin_fmaf(a[2], c2, -mc)), __builtin_amdgcn_exp2f(__builtin_fmaf(a[3], c2, -mc)));
    pf[ks].z = pack2<T>(__builtin_amdgcn_exp2f(__builtin_fmaf(b[0], c2, -mc)), __builtin_amdgcn_exp2f(__builtin_fmaf(b[1], c2, -mc)));
    pf[ks].w = pack2<T>(__builtin_amdgcn_exp2f(__builtin_fmaf(b[2], c2, -mc)), __builtin_amdgcn_exp2f(__builtin_fmaf(b[3], c2, -mc)));
  }
  const uint4 ones = make_uint4(T::ONE_PAIR, T::ONE_PAIR, T::ONE_PAIR, T::ONE_PAIR);
  // V^T fragments: 4-key x 16-column transposed blocks; this lane supplies row (r>>2), columns 4*(r&3).. of each block:
  // keys 32 ks + 4 q + (r>>2) (+16 for the second block); the swizzle does not depend on ks (tile_index.h)
  // explicit one-step-ahead double buffer: without it hipcc issues each pair of transposed reads right in front of
  // the MFMA that consumes it (lgkmcnt(0) per MFMA)
  auto load_v = [&](int ks, s16x4 (&v0)[4], s16x4 (&v1)[4]) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      v0[dt] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((VMC_LDS s16x4*)(v_lds + voff[dt] + (KS0 + ks) * 4096));
      v1[dt] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((VMC_LDS s16x4*)(v_lds + voff[dt] + (KS0 + ks) * 4096 + 2048));
    }
  };
  auto mma_v = [&](int ks, const s16x4 (&v0)[4], const s16x4 (&v1)[4]) {
    osum = T::mfma16(ones, pf[ks], osum);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      uint4 vf;
      const uint2 a = __builtin_bit_cast(uint2, v0[dt]), b = __builtin_bit_cast(uint2, v1[dt]);
      vf.x = a.x; vf.y = a.y; vf.z = b.x; vf.w = b.y;
      o[dt] = T::mfma16(vf, pf[ks], o[dt]);
    }
  };
  s16x4 va0[4], va1[4], vb0[4], vb1[4];
  load_v(0, va0, va1);
#pragma unroll
  for (int ks = 0; ks < NKS; ks += 2) {
    if (ks + 1 < NKS) load_v(ks + 1, vb0, vb1);
    mma_v(ks, va0, va1);
    if (ks + 1 < NKS) {
      if (ks + 2 < NKS) load_v(ks + 2, va0, va1);
      mma_v(ks + 1, vb0, vb1);
    }
  }
}

// q / k / v are given as separate bases with their own row strides (the packed in_proj output is q = qkv, k = qkv + D, v = qkv + 2D,
// all with stride 3D); NQ <= N query rows per frame are processed (NQ = 1: only the class token's query, the last block of the
// encoder, whose other rows nothing reads).
// NW waves per workgroup.  NW = 4: one wave per SIMD per workgroup; hipcc hoists the loop-invariant K fragments (34 ds_read_b128 =
// 136 VGPRs) out of the query-tile loop, which pins the kernel at 252 VGPRs = 2 waves per SIMD.  NW = 8 (REREAD): a compiler
// barrier at the top of every query tile makes the K fragments be re-read from LDS per tile (they are there anyway), the body
// fits 128 VGPRs and two co-resident workgroups put FOUR waves on every SIMD: the LDS / MFMA latencies one wave exposes are
// covered by the other three (the loop was latency bound: ~30 s_waitcnt per tile at 2 waves per SIMD, 78 % issue utilisation).
// PERSIST: the grid is 2 workgroups per CU and each walks (frame, head) pairs bh = blockIdx.x, + gridDim.x, ...: no workgroup
// turnover between heads (PMC: wave lifetime x 8 rounds = 125 us of a 172 us launch).  stagger > 0: workgroups of the second
// half of the grid (the second resident workgroup of each CU under round-robin dispatch -- speed only, never correctness) sleep
// stagger x 8k cycles before their first head, so that one workgroup's K / V staging (HBM bound) runs under its neighbour's MFMAs.
template <typename T, int NT, int NC, int NW = 4, bool REREAD = false, bool PERSIST = false>
__global__ void __launch_bounds__(64 * NW, NW / 2) attn_vit_kernel(const uint16_t* __restrict__ qp, const uint16_t* __restrict__ kp,
                                                       const uint16_t* __restrict__ vp, uint16_t* __restrict__ out,
                                                       float* __restrict__ lse, int N, int NQ, int H, size_t ldq, size_t ldkv, float scale,
                                                       int n_bh = 0, int stagger = 0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NKEYS = 16 * NT;
  char* const k_lds = smem;
  char* const v_lds = smem + NKEYS * 128;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  if constexpr (PERSIST) {
    if (stagger > 0 && blockIdx.x >= (gridDim.x >> 1))
      for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(127);
  }
  for (int bh = blockIdx.x; bh < (PERSIST ? n_bh : (int)blockIdx.x + 1); bh += PERSIST ? (int)gridDim.x : 1) {
  if (PERSIST && bh != (int)blockIdx.x) __syncthreads();      // every wave is done with the previous head's K / V images
  const int f = bh / H, h = bh % H;
  const int D = H * 64;
  const uint16_t* qbase = qp + (size_t)f * NQ * ldq + h * 64;
  const uint16_t* kbase = kp + (size_t)f * N * ldkv + h * 64;
  const uint16_t* vbase = vp + (size_t)f * N * ldkv + h * 64;

  // ---- stage K and V (zero rows for padded keys) ----
  // all loads first (NKEYS*8/256 <= 9 chunks of K and of V per thread, registers are free before the compute
  // phase), then all LDS writes: one exposed HBM/L2 latency per block instead of one per chunk
  {
    constexpr int NTH = 64 * NW;
    constexpr int ITERS = (NKEYS * 8 + NTH - 1) / NTH;
    uint4 kreg[ITERS], vreg[ITERS];
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      const int idx = i * NTH + tid;
      const int row = idx >> 3, c = idx & 7;
      kreg[i] = vreg[i] = make_uint4(0, 0, 0, 0);
      if (idx < NKEYS * 8 && row < N) {
        kreg[i] = *(const uint4*)(kbase + (size_t)row * ldkv + c * 8);
        vreg[i] = *(const uint4*)(vbase + (size_t)row * ldkv + c * 8);
      }
    }
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      const int idx = i * NTH + tid;
      const int row = idx >> 3, c = idx & 7;
      if (idx < NKEYS * 8) {
        *(uint4*)(k_lds + lds_off_x(row, c)) = kreg[i];
        *(uint4*)(v_lds + lds_off_v(row, c)) = vreg[i];
      }
    }
  }
  __syncthreads();

  const float c2 = scale * 1.4426950408889634f;
  const int nqt = (NQ + 15) >> 4;
  // per-lane LDS offsets; tile index / k-step only add compile-time constants (keeps address VGPRs low)
  const int koff0 = lds_off_x(r, q), koff1 = lds_off_x(r, 4 + q);
  int voff[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) voff[dt] = lds_off_v(4 * q + (r >> 2), 2 * dt + ((r & 3) >> 1)) + (r & 1) * 8;
  uint4 qnext[2];
  // nqt = 17 for N = 257: one wave owns five query tiles, the others four.  Waves w of co-resident workgroups share a
  // SIMD, so the long wave rotates with the block index instead of always landing on SIMD 0.
  const int w0 = (wave + bh) & (NW - 1);
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) qnext[kk] = *(const uint4*)(qbase + (size_t)min(w0 * 16 + r, NQ - 1) * ldq + (4 * kk + q) * 8);
  for (int qt = w0; qt < nqt; qt += NW) {
    if constexpr (REREAD) asm volatile("" ::: "memory");      // K fragments are re-read per tile, not kept in 136 registers
    const int qrow = qt * 16 + r;
    uint4 qf[2] = {qnext[0], qnext[1]};
    if (qt + NW < nqt) {  // prefetch the next query tile's fragments under this tile's MFMAs
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) qnext[kk] = *(const uint4*)(qbase + (size_t)min(qrow + 16 * NW, NQ - 1) * ldq + (4 * kk + q) * 8);
    }

    // Keys are processed in one or two chunks (online softmax across chunks): two chunks keep the live score
    // registers at <= 40 (N = 257: 10 + 8 tiles) so nothing spills at 2 waves/SIMD and the LDS reads pipeline.
    f32x4 o[4], osum = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY;
    constexpr int KS = NT / 2;                         // 32-key steps
    constexpr int KA = NT >= 14 ? (KS + 1) / 2 : KS;   // steps in the first chunk
    attn_chunk<T, 0, KA, NC>(k_lds, v_lds, koff0, koff1, voff, qf, N, q, c2, m, o, osum);
    if constexpr (KS > KA) attn_chunk<T, KA, KS - KA, NC>(k_lds, v_lds, koff0, koff1, voff, qf, N, q, c2, m, o, osum);
    const float sum = osum[0];   // every accumulator row holds the full row sum of query r (16-bit rounded P, as P V uses)
    const float inv = 1.0f / sum;
    if (lse != nullptr && q == 0 && qrow < NQ) lse[((size_t)f * H + h) * NQ + qrow] = m * scale + __logf(sum);
    if (qrow < NQ) {
      uint16_t* orow = out + ((size_t)f * NQ + qrow) * D + h * 64 + 4 * q;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        *(uint2*)(orow + 16 * dt) = make_uint2(pack2<T>(o[dt][0] * inv, o[dt][1] * inv), pack2<T>(o[dt][2] * inv, o[dt][3] * inv));
    }
  }
  }      // (frame, head) walk
}

struct VitOperands { const uint16_t *q, *k, *v; size_t ldq, ldkv; int NQ; };

template <typename T, int NT, int NC = 0, int NW = 4, bool REREAD = false, bool PERSIST = false>
static int launch_vit(const VitOperands& a, void* out, float* lse, int F, int N, int H, hipStream_t stream, int stagger = 0) {
  auto kern = attn_vit_kernel<T, NT, NC, NW, REREAD, PERSIST>;
  constexpr int LDS = 16 * NT * 128 * 2;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int grid = PERSIST ? (F * H < 512 ? F * H : 512) : F * H;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), LDS, stream, a.q, a.k, a.v, (uint16_t*)out, lse, N, a.NQ, H, a.ldq, a.ldkv, 0.125f, F * H,
                     stagger);
  VMC_CHECK_LAUNCH();
  return 0;
}

template <typename T>
static int dispatch_vit(const VitOperands& qkv, void* out, float* lse, int F, int N, int H, hipStream_t s) {
  if (N == 257 && qkv.NQ == N) {                                             // ViT-L/14 @ 224
    // default: 8 waves, K fragments re-read per tile (112 VGPRs, 4 waves per SIMD): 167-169 us against 171-174 us per ViT-L/14 layer.
    // VMC_ATTN_VARIANT (builder A/B switch): 9 = the 4-wave kernel of round 2 (252 VGPRs), 2 = 4 waves + re-read, 10+s / 20+s =
    // persistent walks with a stagger of s sleeps (measured slower or equal: profiles/README.md)
    static const int variant = getenv("VMC_ATTN_VARIANT") ? atoi(getenv("VMC_ATTN_VARIANT")) : 1;
    if (variant == 1) return launch_vit<T, 18, 257, 8, true>(qkv, out, lse, F, N, H, s);
    if (variant == 2) return launch_vit<T, 18, 257, 4, true>(qkv, out, lse, F, N, H, s);
    if (variant >= 10 && variant < 20) return launch_vit<T, 18, 257, 4, false, true>(qkv, out, lse, F, N, H, s, variant - 10);
    if (variant >= 20 && variant < 30) return launch_vit<T, 18, 257, 8, true, true>(qkv, out, lse, F, N, H, s, variant - 20);
  }
  if (N == 257) return launch_vit<T, 18, 257>(qkv, out, lse, F, N, H, s);   // ViT-L/14 @ 224
  if (N == 197) return launch_vit<T, 14, 197>(qkv, out, lse, F, N, H, s);   // ViT-B/16
  if (N == 50) return launch_vit<T, 4, 50>(qkv, out, lse, F, N, H, s);      // ViT-B/32
  if (N <= 32) return launch_vit<T, 2>(qkv, out, lse, F, N, H, s);
  if (N <= 64) return launch_vit<T, 4>(qkv, out, lse, F, N, H, s);
  if (N <= 128) return launch_vit<T, 8>(qkv, out, lse, F, N, H, s);
  if (N <= 224) return launch_vit<T, 14>(qkv, out, lse, F, N, H, s);
  if (N <= 288) return launch_vit<T, 18>(qkv, out, lse, F, N, H, s);
  return VMC_E_SHAPE;
}

extern "C" int vmc_attention_vit_fwd(const void* qkv, void* out, float* lse, int F, int N, int H, int dtype16, void* stream) {
  if (!qkv || !out || F <= 0 || N <= 0 || H <= 0) return VMC_E_ARG;
  if (((uintptr_t)qkv | (uintptr_t)out) & 15) return VMC_E_ALIGN;
  const size_t D = (size_t)H * 64;
  const VitOperands a = {(const uint16_t*)qkv, (const uint16_t*)qkv + D, (const uint16_t*)qkv + 2 * D, 3 * D, 3 * D, N};
  if (dtype16 == VMC_BF16) return dispatch_vit<BF16>(a, out, lse, F, N, H, (hipStream_t)stream);
  if (dtype16 == VMC_F16) return dispatch_vit<F16>(a, out, lse, F, N, H, (hipStream_t)stream);
  return VMC_E_DTYPE;
}

extern "C" int vmc_attention_vit_cls_fwd(const void* q_cls, const void* kv, void* out, int F, int N, int H, int dtype16, void* stream) {
  if (!q_cls || !kv || !out || F <= 0 || N <= 0 || H <= 0) return VMC_E_ARG;
  if (((uintptr_t)q_cls | (uintptr_t)kv | (uintptr_t)out) & 15) return VMC_E_ALIGN;
  const size_t D = (size_t)H * 64;
  const VitOperands a = {(const uint16_t*)q_cls, (const uint16_t*)kv, (const uint16_t*)kv + D, D, 2 * D, 1};
  if (dtype16 == VMC_BF16) return dispatch_vit<BF16>(a, out, nullptr, F, N, H, (hipStream_t)stream);
  if (dtype16 == VMC_F16) return dispatch_vit<F16>(a, out, nullptr, F, N, H, (hipStream_t)stream);
  return VMC_E_DTYPE;
}

// ==================================================================================================
// Short-sequence masked attention on MFMA (TFAM self / cross attention: T = 16..64 tokens, head_dim 64 or 96).
// One wave per (batch, head): K and V of the head (<= 64 keys) in LDS, same S^T = K Q^T / O^T = V^T P^T
// formulation as the ViT kernel; key-padding mask -> -inf before the softmax.  Rows are padded by 16 B instead
// of XOR-swizzled (tiles of a few KB: bank conflicts are irrelevant, launch count is what matters here).
// ==================================================================================================
template <typename T, int DH, int NT>
__global__ void __launch_bounds__(64) attn_small_kernel(const uint16_t* __restrict__ qp, const uint16_t* __restrict__ kp,
                                                        const uint16_t* __restrict__ vp, const uint8_t* __restrict__ mask,
                                                        uint16_t* __restrict__ op, float* __restrict__ lse, int H, int Tq, int Tk,
                                                        int ldq, int ldk, int ldv, int ldo, float scale, float drop_p,
                                                        uint64_t seed_arg) {
  // drop_p > 0: nn.MultiheadAttention's dropout on the attention probabilities (AMO_CLIP.py:37-45, training): the softmax is
  // normalised by the sum of the UNdropped probabilities, P V uses p * keep / (1 - drop_p) with vmc_dropout's counter-based mask on
  // the flat index of (batch, head, query, key) -- the same masks as the generic fp32 kernels, regenerated by the backward.
  const uint64_t seed = drop_p > 0.f ? resolve_seed(seed_arg) : 0;
  constexpr int NKEYS = 16 * NT, CH = DH / 8, RS = DH * 2 + 16, KK = DH / 32, DT = DH / 16;
  __shared__ __attribute__((aligned(16))) char k_lds[NKEYS * RS];
  __shared__ __attribute__((aligned(16))) char v_lds[NKEYS * RS];
  const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const uint16_t* kb = kp + (size_t)b * Tk * ldk + h * DH;
  const uint16_t* vb = vp + (size_t)b * Tk * ldv + h * DH;
  // the first query tile's fragments travel with the K / V staging loads (one exposed round trip instead of two); later tiles are
  // fetched one tile ahead
  uint4 qnext[KK];
  {
    const uint16_t* qr0 = qp + ((size_t)b * Tq + min(r, Tq - 1)) * ldq + h * DH;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) qnext[kk] = *(const uint4*)(qr0 + (4 * kk + q) * 8);
  }
  {
    constexpr int ITERS = (NKEYS * CH + 63) / 64;
    uint4 kreg[ITERS], vreg[ITERS];
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      const int idx = i * 64 + lane, row = idx / CH, c = idx % CH;
      kreg[i] = vreg[i] = make_uint4(0, 0, 0, 0);
      if (idx < NKEYS * CH && row < Tk) {
        kreg[i] = *(const uint4*)(kb + (size_t)row * ldk + c * 8);
        vreg[i] = *(const uint4*)(vb + (size_t)row * ldv + c * 8);
      }
    }
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      const int idx = i * 64 + lane, row = idx / CH, c = idx % CH;
      if (idx < NKEYS * CH) {
        *(uint4*)(k_lds + row * RS + c * 16) = kreg[i];
        *(uint4*)(v_lds + row * RS + c * 16) = vreg[i];
      }
    }
  }
  // which of this lane's keys (16 nt + 4 q + j) may be attended
  bool live[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int key = 16 * nt + 4 * q + j;
      live[nt][j] = key < Tk && (mask == nullptr || mask[(size_t)b * Tk + key] != 0);
    }
  __syncthreads();

  const float c2 = scale * 1.4426950408889634f;
  const uint4 ones = make_uint4(T::ONE_PAIR, T::ONE_PAIR, T::ONE_PAIR, T::ONE_PAIR);
  for (int qt = 0; qt * 16 < Tq; ++qt) {
    const int qrow = qt * 16 + r;
    uint4 qf[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) qf[kk] = qnext[kk];
    if ((qt + 1) * 16 < Tq) {
      const uint16_t* qr = qp + ((size_t)b * Tq + min(qrow + 16, Tq - 1)) * ldq + h * DH;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) qnext[kk] = *(const uint4*)(qr + (4 * kk + q) * 8);
    }
    f32x4 s[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      s[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < KK; ++kk)
        s[nt] = T::mfma16(*(const uint4*)(k_lds + (16 * nt + r) * RS + (4 * kk + q) * 16), qf[kk], s[nt]);
    }
    float m = -INFINITY;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (!live[nt][j]) s[nt][j] = -INFINITY;
        m = fmaxf(m, s[nt][j]);
      }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float mc = m * c2;   // a fully masked row gives exp2(NaN): NaN output, as torch
    f32x4 o[DT], osum = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NT / 2; ++ks) {
      const f32x4 a = s[2 * ks], bb = s[2 * ks + 1];
      float pe[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        pe[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(a[j], c2, -mc));
        pe[4 + j] = __builtin_amdgcn_exp2f(__builtin_fmaf(bb[j], c2, -mc));
      }
      uint4 pf = make_uint4(pack2<T>(pe[0], pe[1]), pack2<T>(pe[2], pe[3]), pack2<T>(pe[4], pe[5]), pack2<T>(pe[6], pe[7]));
      osum = T::mfma16(ones, pf, osum);
      if (drop_p > 0.f) {                      // wave-uniform
        const size_t base = (((size_t)b * H + h) * Tq + min(qrow, Tq - 1)) * Tk + 32 * ks + 4 * q;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          pe[j] *= dropout_factor(drop_p, seed, base + j);
          pe[4 + j] *= dropout_factor(drop_p, seed, base + 16 + j);
        }
        pf = make_uint4(pack2<T>(pe[0], pe[1]), pack2<T>(pe[2], pe[3]), pack2<T>(pe[4], pe[5]), pack2<T>(pe[6], pe[7]));
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        // transposed 4-key x 16-column blocks: this lane supplies key row 32 ks + 4 q + (r>>2) (+16), columns 16 dt + 4 (r&3)..
        const char* p0 = v_lds + (32 * ks + 4 * q + (r >> 2)) * RS + (16 * dt + 4 * (r & 3)) * 2;
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((VMC_LDS s16x4*)p0);
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((VMC_LDS s16x4*)(p0 + 16 * RS));
        uint4 vf;
        const uint2 x0 = __builtin_bit_cast(uint2, v0), x1 = __builtin_bit_cast(uint2, v1);
        vf.x = x0.x; vf.y = x0.y; vf.z = x1.x; vf.w = x1.y;
        o[dt] = T::mfma16(vf, pf, o[dt]);
      }
    }
    const float sum = osum[0];
    const float inv = 1.0f / sum;
    if (qrow < Tq) {
      uint16_t* orow = op + ((size_t)b * Tq + qrow) * ldo + h * DH + 4 * q;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
        *(uint2*)(orow + 16 * dt) = make_uint2(pack2<T>(o[dt][0] * inv, o[dt][1] * inv), pack2<T>(o[dt][2] * inv, o[dt][3] * inv));
      if (lse != nullptr && q == 0) lse[((size_t)b * H + h) * Tq + qrow] = m * scale + __logf(sum);
    }
  }
}

template <typename T>
static int launch_small(const void* q, const void* k, const void* v, const uint8_t* mask, void* out, float* lse, int B, int H, int Tq,
                        int Tk, int dh, int ldq, int ldk, int ldv, int ldo, float scale, float drop_p, uint64_t seed, hipStream_t s) {
#define VMC_SMALL(DHV, NTV)                                                                                                      \
  hipLaunchKernelGGL((attn_small_kernel<T, DHV, NTV>), dim3(B * H), dim3(64), 0, s, (const uint16_t*)q, (const uint16_t*)k,      \
                     (const uint16_t*)v, mask, (uint16_t*)out, lse, H, Tq, Tk, ldq, ldk, ldv, ldo, scale, drop_p, seed)
  if (dh == 64 && Tk <= 32) VMC_SMALL(64, 2);
  else if (dh == 64) VMC_SMALL(64, 4);
  else if (dh == 96 && Tk <= 32) VMC_SMALL(96, 2);
  else VMC_SMALL(96, 4);
#undef VMC_SMALL
  VMC_CHECK_LAUNCH();
  return 0;
}

// ==================================================================================================
// Generic masked attention, fp32 math.  One wave per (batch, head, query).  Scores live in LDS.
// ==================================================================================================
#define ATT_MAX_TK 2048
#define ATT_MAX_DH 128

template <typename T>
__device__ inline float dot16(const uint16_t* __restrict__ a16, const float* __restrict__ bf, int dh) {
  float acc = 0.f;
  for (int d = 0; d < dh; d += 8) {
    const uint4 w = *(const uint4*)(a16 + d);
    float x0, x1;
    unpack2<T>(w.x, x0, x1); acc += x0 * bf[d] + x1 * bf[d + 1];
    unpack2<T>(w.y, x0, x1); acc += x0 * bf[d + 2] + x1 * bf[d + 3];
    unpack2<T>(w.z, x0, x1); acc += x0 * bf[d + 4] + x1 * bf[d + 5];
    unpack2<T>(w.w, x0, x1); acc += x0 * bf[d + 6] + x1 * bf[d + 7];
  }
  return acc;
}

template <typename T>
__global__ void __launch_bounds__(64) attn_generic_fwd(const uint16_t* __restrict__ qp, const uint16_t* __restrict__ kp,
                                                       const uint16_t* __restrict__ vp, const uint8_t* __restrict__ mask,
                                                       uint16_t* __restrict__ op, float* __restrict__ lse, int H, int Tq,
                                                       int Tk, int dh, int ldq, int ldk, int ldv, int ldo, float scale,
                                                       float drop_p, unsigned long long seed_arg) {
  const unsigned long long seed = drop_p > 0.f ? resolve_seed(seed_arg) : 0;
  __shared__ float qs[ATT_MAX_DH];
  __shared__ float ps[ATT_MAX_TK];
  const int lane = threadIdx.x;
  const int t = blockIdx.x % Tq, h = (blockIdx.x / Tq) % H, b = blockIdx.x / (Tq * H);
  const uint16_t* qrow = qp + ((size_t)b * Tq + t) * ldq + h * dh;
  for (int d = lane; d < dh; d += 64) qs[d] = T::to_f32(qrow[d]) * scale;  // q * dh^-1/2 first, as torch MHA does
  __syncthreads();
  float m = -INFINITY;
  for (int key = lane; key < Tk; key += 64) {
    float s = -INFINITY;
    if (mask == nullptr || mask[(size_t)b * Tk + key]) s = dot16<T>(kp + ((size_t)b * Tk + key) * ldk + h * dh, qs, dh);
    ps[key] = s;
    m = fmaxf(m, s);
  }
  m = wave_max(m);
  float sum = 0.f;
  for (int key = lane; key < Tk; key += 64) {
    const float p = __expf(ps[key] - m);  // all-masked row: (-inf) - (-inf) = NaN, as torch
    // dropout acts on the normalised probabilities (nn.MultiheadAttention(dropout=p)); the row sum does not see it
    ps[key] = p * dropout_factor(drop_p, seed, (((size_t)b * H + h) * Tq + t) * Tk + key);
    sum += p;
  }
  sum = wave_sum(sum);
  __syncthreads();
  const float inv = 1.0f / sum;
  for (int d = lane; d < dh; d += 64) {
    float acc = 0.f;
    const uint16_t* vcol = vp + (size_t)b * Tk * ldv + h * dh + d;
    for (int key = 0; key < Tk; ++key) acc += ps[key] * T::to_f32(vcol[(size_t)key * ldv]);
    op[((size_t)b * Tq + t) * ldo + h * dh + d] = T::from_f32(acc * inv);
  }
  if (lse != nullptr && lane == 0) lse[((size_t)b * H + h) * Tq + t] = m + __logf(sum);
}

// dq pass: one wave per (b, h, query).  Also writes delta[b,h,t] = sum_d dO*O.
template <typename T>
__global__ void __launch_bounds__(64) attn_generic_bwd_q(const uint16_t* __restrict__ qp, const uint16_t* __restrict__ kp,
                                                         const uint16_t* __restrict__ vp, const uint8_t* __restrict__ mask,
                                                         const uint16_t* __restrict__ op, const uint16_t* __restrict__ dop,
                                                         const float* __restrict__ lse, uint16_t* __restrict__ dqp,
                                                         float* __restrict__ delta, int H, int Tq, int Tk, int dh, int ldq,
                                                         int ldk, int ldv, int ldo, int lddq, float scale, float drop_p,
                                                         unsigned long long seed_arg) {
  const unsigned long long seed = drop_p > 0.f ? resolve_seed(seed_arg) : 0;
  __shared__ float qs[ATT_MAX_DH];
  __shared__ float dos[ATT_MAX_DH];
  __shared__ float ds[ATT_MAX_TK];
  const int lane = threadIdx.x;
  const int t = blockIdx.x % Tq, h = (blockIdx.x / Tq) % H, b = blockIdx.x / (Tq * H);
  const size_t qi = ((size_t)b * Tq + t);
  float dl = 0.f;
  for (int d = lane; d < dh; d += 64) {
    qs[d] = T::to_f32(qp[qi * ldq + h * dh + d]) * scale;
    const float g = T::to_f32(dop[qi * ldo + h * dh + d]);
    dos[d] = g;
    dl += g * T::to_f32(op[qi * ldo + h * dh + d]);
  }
  dl = wave_sum(dl);
  __syncthreads();
  const float l = lse[((size_t)b * H + h) * Tq + t];
  if (lane == 0) delta[((size_t)b * H + h) * Tq + t] = dl;
  for (int key = lane; key < Tk; key += 64) {
    float dsv = 0.f;
    if (mask == nullptr || mask[(size_t)b * Tk + key]) {
      const size_t ki = (size_t)b * Tk + key;
      const float p = __expf(dot16<T>(kp + ki * ldk + h * dh, qs, dh) - l);
      const float dp = dot16<T>(vp + ki * ldv + h * dh, dos, dh);
      dsv = p * (dp * dropout_factor(drop_p, seed, (((size_t)b * H + h) * Tq + t) * Tk + key) - dl);
    }
    ds[key] = dsv;
  }
  __syncthreads();
  for (int d = lane; d < dh; d += 64) {
    float acc = 0.f;
    const uint16_t* kcol = kp + (size_t)b * Tk * ldk + h * dh + d;
    for (int key = 0; key < Tk; ++key) acc += ds[key] * T::to_f32(kcol[(size_t)key * ldk]);
    dqp[qi * lddq + h * dh + d] = T::from_f32(acc * scale);
  }
}

// dk/dv pass: one wave per (b, h, key); loops over the queries.
template <typename T>
__global__ void __launch_bounds__(64) attn_generic_bwd_kv(const uint16_t* __restrict__ qp, const uint16_t* __restrict__ kp,
                                                          const uint16_t* __restrict__ vp, const uint8_t* __restrict__ mask,
                                                          const uint16_t* __restrict__ dop, const float* __restrict__ lse,
                                                          const float* __restrict__ delta, uint16_t* __restrict__ dkp,
                                                          uint16_t* __restrict__ dvp, int H, int Tq, int Tk, int dh, int ldq,
                                                          int ldk, int ldv, int ldo, int lddk, int lddv, float scale,
                                                          float drop_p, unsigned long long seed_arg) {
  const unsigned long long seed = drop_p > 0.f ? resolve_seed(seed_arg) : 0;
  __shared__ float ks[ATT_MAX_DH];
  __shared__ float vs[ATT_MAX_DH];
  __shared__ float pbuf[ATT_MAX_TK];   // p[t]  for this key
  __shared__ float dsbuf[ATT_MAX_TK];  // dS[t] for this key
  const int lane = threadIdx.x;
  const int key = blockIdx.x % Tk, h = (blockIdx.x / Tk) % H, b = blockIdx.x / (Tk * H);
  const size_t ki = (size_t)b * Tk + key;
  const bool live = (mask == nullptr) || mask[ki];
  for (int d = lane; d < dh; d += 64) {
    ks[d] = T::to_f32(kp[ki * ldk + h * dh + d]);
    vs[d] = T::to_f32(vp[ki * ldv + h * dh + d]);
  }
  __syncthreads();
  for (int t = lane; t < Tq; t += 64) {
    float p = 0.f, dsv = 0.f;
    if (live) {
      const size_t qi = (size_t)b * Tq + t;
      const size_t si = ((size_t)b * H + h) * Tq + t;
      p = __expf(dot16<T>(qp + qi * ldq + h * dh, ks, dh) * scale - lse[si]);
      const float dp = dot16<T>(dop + qi * ldo + h * dh, vs, dh);
      const float dfac = dropout_factor(drop_p, seed, (((size_t)b * H + h) * Tq + t) * Tk + key);
      dsv = p * (dp * dfac - delta[si]);
      p *= dfac;  // dV sees the dropped probabilities
    }
    pbuf[t] = p;
    dsbuf[t] = dsv;
  }
  __syncthreads();
  for (int d = lane; d < dh; d += 64) {
    float ak = 0.f, av = 0.f;
    for (int t = 0; t < Tq; ++t) {
      const size_t qi = (size_t)b * Tq + t;
      ak += dsbuf[t] * T::to_f32(qp[qi * ldq + h * dh + d]);
      av += pbuf[t] * T::to_f32(dop[qi * ldo + h * dh + d]);
    }
    dkp[ki * lddk + h * dh + d] = T::from_f32(ak * scale);
    dvp[ki * lddv + h * dh + d] = T::from_f32(av);
  }
}

// ==================================================================================================
// MFMA attention backward (no dropout): one workgroup of 4 waves per (batch, head); Q, K, V, dO of the head in LDS.
//   phase A (a wave owns key tiles): P, dS in the S orientation (S = Q K^T: lane = key, registers = queries) are
//            directly the B operands of dV^T += dO^T P and dK^T += Q^T dS (contraction over queries; dO^T / Q^T
//            fragments by ds_read_b64_tr_b16);
//   phase B (a wave owns query tiles): dS^T in the S^T orientation feeds dQ^T += K^T dS^T (contraction over keys).
// Scores are recomputed from Q, K and the forward's log-sum-exp (two cheap MFMAs per tile pair); delta = rowsum(dO*O)
// is computed in the prologue.  Rows of the LDS images are padded by 16 B when they fit, else unpadded.
// ==================================================================================================
template <typename T, int DH>
__global__ void __launch_bounds__(256, 3) attn_bwd_mfma_kernel(const uint16_t* __restrict__ qp, const uint16_t* __restrict__ kp,
                                                            const uint16_t* __restrict__ vp, const uint8_t* __restrict__ mask,
                                                            const uint16_t* __restrict__ op, const uint16_t* __restrict__ dop,
                                                            const float* __restrict__ lse, uint16_t* __restrict__ dqp,
                                                            uint16_t* __restrict__ dkp, uint16_t* __restrict__ dvp, int H, int Tq,
                                                            int Tk, int ldq, int ldk, int ldv, int ldo, int lddq, int lddk, int lddv,
                                                            float scale, int RS, float drop_p, uint64_t seed_arg) {
  // drop_p > 0 (dropout on the probabilities, forward above): with f = keep / (1 - drop_p) regenerated from the seed,
  // dV = (P f)^T dO,  dS = P (dP f - delta),  delta = rowsum(dO O) as without dropout.
  const uint64_t seed = drop_p > 0.f ? resolve_seed(seed_arg) : 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int CH = DH / 8, KK = DH / 32, DT = DH / 16;
  const int TQP = (Tq + 31) & ~31, TKP = (Tk + 31) & ~31;       // padded to whole 32-row steps (zero rows)
  char* const q_lds = smem;
  char* const do_lds = q_lds + TQP * RS;
  char* const k_lds = do_lds + TQP * RS;
  char* const v_lds = k_lds + TKP * RS;
  float* const lse_s = (float*)(v_lds + TKP * RS);                // [TQP] log2-domain lse: lse * log2(e)
  float* const del_s = lse_s + TQP;                               // [TQP] delta
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nthr = blockDim.x, nwaves = nthr >> 6;      // 1 .. 4 waves: as many as there are tile tasks (launch_bwd_mfma)
  const int r = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const size_t qrow0 = (size_t)b * Tq, krow0 = (size_t)b * Tk;

  // ---- stage Q, dO, K, V (zero padding rows); delta and lse ----
  // ONE exposed memory round trip: every global load of the prologue (the first PF chunks per thread of each image -- all of them for
  // the shapes this kernel sees: Tq, Tk <= 64 -- the O chunks and the lse of the delta rows) is issued before the first LDS write;
  // as four run-time loops (stage Q | dO, stage K | V, delta from O and dO re-read from global) the prologue was five dependent
  // round trips, most of a 13 us workgroup at ViT-B/32 (N = 50) and of the 7.7 us launch of a 16-token TFAM clip.
  constexpr int PF = 4;
  uint4 rq[PF], rdo[PF], rk[PF], rv[PF];
#pragma unroll
  for (int i = 0; i < PF; ++i) {
    const int idx = tid + i * nthr, row = idx / CH, c = idx % CH;
    rq[i] = rdo[i] = rk[i] = rv[i] = make_uint4(0, 0, 0, 0);
    if (idx < TQP * CH && row < Tq) {
      rq[i] = *(const uint4*)(qp + (qrow0 + row) * ldq + h * DH + c * 8);
      rdo[i] = *(const uint4*)(dop + (qrow0 + row) * ldo + h * DH + c * 8);
    }
    if (idx < TKP * CH && row < Tk) {
      rk[i] = *(const uint4*)(kp + (krow0 + row) * ldk + h * DH + c * 8);
      rv[i] = *(const uint4*)(vp + (krow0 + row) * ldv + h * DH + c * 8);
    }
  }
  // delta = rowsum(dO * O): four lanes per row (column chunks c = 8 part, 8 part + 32, ...), summed by two shuffles; O from global
  // (prefetched here), dO from its LDS image after the first barrier
  constexpr int DPF = 2;                                  // delta items per thread held in registers (TQP * 4 <= DPF * nthr)
  uint4 ro[DPF][KK];
  float rl[DPF];
#pragma unroll
  for (int u = 0; u < DPF; ++u) {
    const int item = tid + u * nthr, row = item >> 2, part = item & 3;
    rl[u] = 0.f;
#pragma unroll
    for (int m = 0; m < KK; ++m) ro[u][m] = make_uint4(0, 0, 0, 0);
    if (item < TQP * 4 && row < Tq) {
#pragma unroll
      for (int m = 0; m < KK; ++m)
        if (8 * part + 32 * m < DH) ro[u][m] = *(const uint4*)(op + (qrow0 + row) * ldo + h * DH + 8 * part + 32 * m);
      if (part == 0) rl[u] = lse[((size_t)b * H + h) * Tq + row] * 1.4426950408889634f;
    }
  }
#pragma unroll
  for (int i = 0; i < PF; ++i) {
    const int idx = tid + i * nthr, row = idx / CH, c = idx % CH;
    if (idx < TQP * CH) {
      *(uint4*)(q_lds + row * RS + c * 16) = rq[i];
      *(uint4*)(do_lds + row * RS + c * 16) = rdo[i];
    }
    if (idx < TKP * CH) {
      *(uint4*)(k_lds + row * RS + c * 16) = rk[i];
      *(uint4*)(v_lds + row * RS + c * 16) = rv[i];
    }
  }
  for (int idx = tid + PF * nthr; idx < TQP * CH; idx += nthr) {        // longer sequences: the rest chunk by chunk
    const int row = idx / CH, c = idx % CH;
    uint4 a = make_uint4(0, 0, 0, 0), d = a;
    if (row < Tq) {
      a = *(const uint4*)(qp + (qrow0 + row) * ldq + h * DH + c * 8);
      d = *(const uint4*)(dop + (qrow0 + row) * ldo + h * DH + c * 8);
    }
    *(uint4*)(q_lds + row * RS + c * 16) = a;
    *(uint4*)(do_lds + row * RS + c * 16) = d;
  }
  for (int idx = tid + PF * nthr; idx < TKP * CH; idx += nthr) {
    const int row = idx / CH, c = idx % CH;
    uint4 a = make_uint4(0, 0, 0, 0), d = a;
    if (row < Tk) {
      a = *(const uint4*)(kp + (krow0 + row) * ldk + h * DH + c * 8);
      d = *(const uint4*)(vp + (krow0 + row) * ldv + h * DH + c * 8);
    }
    *(uint4*)(k_lds + row * RS + c * 16) = a;
    *(uint4*)(v_lds + row * RS + c * 16) = d;
  }
  __syncthreads();
  auto delta_item = [&](int item, const uint4 (&ov)[KK], float l2) {
    const int row = item >> 2, part = item & 3;
    float dl = 0.f;
    if (row < Tq) {
#pragma unroll
      for (int m = 0; m < KK; ++m) {
        const int c = 8 * part + 32 * m;
        if (c < DH) {
          const uint4 ow = ov[m], dw = *(const uint4*)(do_lds + row * RS + c * 2);
          const uint32_t oa[4] = {ow.x, ow.y, ow.z, ow.w}, da[4] = {dw.x, dw.y, dw.z, dw.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float o0, o1, d0, d1;
            unpack2<T>(oa[j], o0, o1);
            unpack2<T>(da[j], d0, d1);
            dl += o0 * d0 + o1 * d1;
          }
        }
      }
    }
    dl += __shfl_xor(dl, 1, 64);
    dl += __shfl_xor(dl, 2, 64);
    if (part == 0) {
      del_s[row] = dl;
      lse_s[row] = l2;
    }
  };
#pragma unroll
  for (int u = 0; u < DPF; ++u)
    if (tid + u * nthr < TQP * 4) delta_item(tid + u * nthr, ro[u], rl[u]);
  for (int item = tid + DPF * nthr; item < TQP * 4; item += nthr) {      // longer sequences
    const int row = item >> 2, part = item & 3;
    uint4 ov[KK];
#pragma unroll
    for (int m = 0; m < KK; ++m)
      ov[m] = (row < Tq && 8 * part + 32 * m < DH) ? *(const uint4*)(op + (qrow0 + row) * ldo + h * DH + 8 * part + 32 * m) : make_uint4(0, 0, 0, 0);
    delta_item(item, ov, (row < Tq && part == 0) ? lse[((size_t)b * H + h) * Tq + row] * 1.4426950408889634f : 0.f);
  }
  __syncthreads();

  const float c2 = scale * 1.4426950408889634f;
  const int nkt = TKP >> 4, nqt = TQP >> 4;
  // Tiles with at least one real row.  Phase A (a key tile -> dK, dV) and phase B (a query tile -> dQ) only read the LDS images, so
  // they are independent TASKS dealt round-robin to the workgroup's waves (launched: min(4, tasks)): at T = 16 (one live tile each way) waves 0 and 1 run the two
  // phases side by side instead of one after the other, and all-padding tiles (rows 16..31 of a 16-token clip) are not computed.
  const int nktL = (Tk + 15) >> 4, nqtL = (Tq + 15) >> 4;

  for (int task = wave; task < nktL + nqtL; task += nwaves) {
  if (task < nktL) {
  // ================= phase A: dK, dV =================
    const int nt = task;
    const int key = 16 * nt + r;                               // this lane's key column
    const bool klive = key < Tk && (mask == nullptr || mask[(size_t)b * Tk + key] != 0);
    uint4 kf[KK], vf[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      kf[kk] = *(const uint4*)(k_lds + key * RS + (4 * kk + g) * 16);
      vf[kk] = *(const uint4*)(v_lds + key * RS + (4 * kk + g) * 16);
    }
    f32x4 dvt[DT], dkt[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) dvt[dt] = dkt[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int s2 = 0; s2 < nqt / 2; ++s2) {
      uint32_t pp[4], dd[4];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int qt = 2 * s2 + half;
        f32x4 sc = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = sc;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
          const uint4 qf = *(const uint4*)(q_lds + (16 * qt + r) * RS + (4 * kk + g) * 16);
          const uint4 df = *(const uint4*)(do_lds + (16 * qt + r) * RS + (4 * kk + g) * 16);
          sc = T::mfma16(qf, kf[kk], sc);                      // S[query 4g+j][key r]
          dp = T::mfma16(df, vf[kk], dp);                      // dP[query][key]
        }
        float pv[4], dv4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int qi = 16 * qt + 4 * g + j;
          const float p = (klive && qi < Tq) ? __builtin_amdgcn_exp2f(__builtin_fmaf(sc[j], c2, -lse_s[qi])) : 0.f;
          const float f = drop_p > 0.f ? dropout_factor(drop_p, seed, (((size_t)b * H + h) * Tq + min(qi, Tq - 1)) * Tk + min(key, Tk - 1)) : 1.0f;
          pv[j] = p * f;
          dv4[j] = p * (dp[j] * f - del_s[qi]);
        }
        pp[2 * half] = pack2<T>(pv[0], pv[1]); pp[2 * half + 1] = pack2<T>(pv[2], pv[3]);
        dd[2 * half] = pack2<T>(dv4[0], dv4[1]); dd[2 * half + 1] = pack2<T>(dv4[2], dv4[3]);
      }
      const uint4 pfrag = make_uint4(pp[0], pp[1], pp[2], pp[3]), dsfrag = make_uint4(dd[0], dd[1], dd[2], dd[3]);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        // transposed 4-query x 16-column blocks of dO and Q: rows 32 s2 + 4 g + (r>>2) (+16), columns 16 dt + 4 (r&3)..
        const int off = (32 * s2 + 4 * g + (r >> 2)) * RS + (16 * dt + 4 * (r & 3)) * 2;
        const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((VMC_LDS s16x4*)(do_lds + off));
        const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((VMC_LDS s16x4*)(do_lds + off + 16 * RS));
        const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((VMC_LDS s16x4*)(q_lds + off));
        const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((VMC_LDS s16x4*)(q_lds + off + 16 * RS));
        const uint2 x0 = __builtin_bit_cast(uint2, a0), x1 = __builtin_bit_cast(uint2, a1);
        const uint2 y0 = __builtin_bit_cast(uint2, b0), y1 = __builtin_bit_cast(uint2, b1);
        dvt[dt] = T::mfma16(make_uint4(x0.x, x0.y, x1.x, x1.y), pfrag, dvt[dt]);      // dV^T[d][key]
        dkt[dt] = T::mfma16(make_uint4(y0.x, y0.y, y1.x, y1.y), dsfrag, dkt[dt]);     // dK^T[d][key]
      }
    }
    if (key < Tk) {
      uint16_t* dvr = dvp + (krow0 + key) * lddv + h * DH + 4 * g;
      uint16_t* dkr = dkp + (krow0 + key) * lddk + h * DH + 4 * g;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        *(uint2*)(dvr + 16 * dt) = make_uint2(pack2<T>(dvt[dt][0], dvt[dt][1]), pack2<T>(dvt[dt][2], dvt[dt][3]));
        *(uint2*)(dkr + 16 * dt) = make_uint2(pack2<T>(dkt[dt][0] * scale, dkt[dt][1] * scale), pack2<T>(dkt[dt][2] * scale, dkt[dt][3] * scale));
      }
    }
  } else {
  // ================= phase B: dQ =================
    const int qt = task - nktL;
    const int qi = 16 * qt + r;                                // this lane's query column
    const bool qlive = qi < Tq;
    const float l2 = lse_s[qi], dl = del_s[qi];
    uint4 qf[KK], df[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      qf[kk] = *(const uint4*)(q_lds + qi * RS + (4 * kk + g) * 16);
      df[kk] = *(const uint4*)(do_lds + qi * RS + (4 * kk + g) * 16);
    }
    f32x4 dqt[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) dqt[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int s2 = 0; s2 < nkt / 2; ++s2) {
      uint32_t dd[4];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int nt = 2 * s2 + half;
        f32x4 sc = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = sc;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
          const uint4 kf = *(const uint4*)(k_lds + (16 * nt + r) * RS + (4 * kk + g) * 16);
          const uint4 vf = *(const uint4*)(v_lds + (16 * nt + r) * RS + (4 * kk + g) * 16);
          sc = T::mfma16(kf, qf[kk], sc);                      // S^T[key 4g+j][query r]
          dp = T::mfma16(vf, df[kk], dp);
        }
        float dv4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int key = 16 * nt + 4 * g + j;
          const bool live = qlive && key < Tk && (mask == nullptr || mask[(size_t)b * Tk + key] != 0);
          const float p = live ? __builtin_amdgcn_exp2f(__builtin_fmaf(sc[j], c2, -l2)) : 0.f;
          const float f = drop_p > 0.f ? dropout_factor(drop_p, seed, (((size_t)b * H + h) * Tq + min(qi, Tq - 1)) * Tk + min(key, Tk - 1)) : 1.0f;
          dv4[j] = p * (dp[j] * f - dl);
        }
        dd[2 * half] = pack2<T>(dv4[0], dv4[1]); dd[2 * half + 1] = pack2<T>(dv4[2], dv4[3]);
      }
      const uint4 dsfrag = make_uint4(dd[0], dd[1], dd[2], dd[3]);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int off = (32 * s2 + 4 * g + (r >> 2)) * RS + (16 * dt + 4 * (r & 3)) * 2;
        const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((VMC_LDS s16x4*)(k_lds + off));
        const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((VMC_LDS s16x4*)(k_lds + off + 16 * RS));
        const uint2 x0 = __builtin_bit_cast(uint2, a0), x1 = __builtin_bit_cast(uint2, a1);
        dqt[dt] = T::mfma16(make_uint4(x0.x, x0.y, x1.x, x1.y), dsfrag, dqt[dt]);    // dQ^T[d][query]
      }
    }
    if (qlive) {
      uint16_t* dqr = dqp + (qrow0 + qi) * lddq + h * DH + 4 * g;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
        *(uint2*)(dqr + 16 * dt) = make_uint2(pack2<T>(dqt[dt][0] * scale, dqt[dt][1] * scale), pack2<T>(dqt[dt][2] * scale, dqt[dt][3] * scale));
    }
  }
  }  // tasks
}

template <typename T, int DH>
static int launch_bwd_mfma(const void* q, const void* k, const void* v, const uint8_t* mask, const void* out, const void* dout,
                           const float* lse, void* dq, void* dk, void* dv, int B, int H, int Tq, int Tk, int ldq, int ldk, int ldv,
                           int ldo, int lddq, int lddk, int lddv, float scale, float drop_p, uint64_t seed, hipStream_t s) {
  const int TQP = (Tq + 31) & ~31, TKP = (Tk + 31) & ~31;
  int RS = DH * 2 + 16;
  size_t lds = (size_t)2 * (TQP + TKP) * RS + (size_t)2 * TQP * sizeof(float);
  if (lds > 160 * 1024) {
    RS = DH * 2;
    lds = (size_t)2 * (TQP + TKP) * RS + (size_t)2 * TQP * sizeof(float);
  }
  if (lds > 160 * 1024) return VMC_E_SHAPE;
  auto kern = attn_bwd_mfma_kernel<T, DH>;
  static size_t attr = 0;
  if (lds > attr) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr = 160 * 1024;
  }
  // one wave per tile task, at most four: a 16-token clip (one key tile + one query tile) runs as a 2-wave workgroup, which lets
  // five of them share a CU instead of three 4-wave ones with two idle waves each
  const int tasks = (Tk + 15) / 16 + (Tq + 15) / 16;
  const int threads = 64 * (tasks < 4 ? tasks : 4);
  hipLaunchKernelGGL(kern, dim3(B * H), dim3(threads), lds, s, (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v, mask,
                     (const uint16_t*)out, (const uint16_t*)dout, lse, (uint16_t*)dq, (uint16_t*)dk, (uint16_t*)dv, H, Tq, Tk, ldq, ldk,
                     ldv, ldo, lddq, lddk, lddv, scale, RS, drop_p, seed);
  VMC_CHECK_LAUNCH();
  return 0;
}

static int check_generic(int B, int H, int Tq, int Tk, int dh, int ldq, int ldk, int ldv, int ldo) {
  if (B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0) return VMC_E_ARG;
  if (dh <= 0 || dh > ATT_MAX_DH || (dh % 8) || Tk > ATT_MAX_TK || Tq > ATT_MAX_TK) return VMC_E_SHAPE;
  if ((ldq % 8) || (ldk % 8) || (ldv % 8) || (ldo % 8)) return VMC_E_ALIGN;
  return 0;
}

extern "C" int vmc_attention_fwd(const void* q, const void* k, const void* v, const uint8_t* key_mask, void* out,
                                 float* lse, int B, int H, int Tq, int Tk, int dh, int ldq, int ldk, int ldv, int ldo,
                                 float dropout_p, uint64_t dropout_seed, int dtype16, void* stream) {
  if (!q || !k || !v || !out || dropout_p < 0.f || dropout_p >= 1.f) return VMC_E_ARG;
  int rc = check_generic(B, H, Tq, Tk, dh, ldq, ldk, ldv, ldo);
  if (rc) return rc;
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) return VMC_E_ALIGN;
  const float scale = 1.0f / sqrtf((float)dh);
  if (Tk <= 64 && (dh == 64 || dh == 96) && (ldo % 4) == 0) {   // short sequences: MFMA kernel, one wave per (b, h)
    if (dtype16 == VMC_BF16) return launch_small<BF16>(q, k, v, key_mask, out, lse, B, H, Tq, Tk, dh, ldq, ldk, ldv, ldo, scale, dropout_p, dropout_seed, (hipStream_t)stream);
    if (dtype16 == VMC_F16) return launch_small<F16>(q, k, v, key_mask, out, lse, B, H, Tq, Tk, dh, ldq, ldk, ldv, ldo, scale, dropout_p, dropout_seed, (hipStream_t)stream);
    return VMC_E_DTYPE;
  }
  dim3 grid(B * H * Tq);
  if (dtype16 == VMC_BF16)
    hipLaunchKernelGGL(attn_generic_fwd<BF16>, grid, dim3(64), 0, (hipStream_t)stream, (const uint16_t*)q, (const uint16_t*)k,
                       (const uint16_t*)v, key_mask, (uint16_t*)out, lse, H, Tq, Tk, dh, ldq, ldk, ldv, ldo, scale, dropout_p,
                       (unsigned long long)dropout_seed);
  else if (dtype16 == VMC_F16)
    hipLaunchKernelGGL(attn_generic_fwd<F16>, grid, dim3(64), 0, (hipStream_t)stream, (const uint16_t*)q, (const uint16_t*)k,
                       (const uint16_t*)v, key_mask, (uint16_t*)out, lse, H, Tq, Tk, dh, ldq, ldk, ldv, ldo, scale, dropout_p,
                       (unsigned long long)dropout_seed);
  else
    return VMC_E_DTYPE;
  VMC_CHECK_LAUNCH();
  return 0;
}

extern "C" size_t vmc_attention_bwd_workspace_bytes(int B, int H, int Tq) { return (size_t)B * H * Tq * sizeof(float); }

extern "C" int vmc_attention_bwd(const void* q, const void* k, const void* v, const uint8_t* key_mask, const void* out,
                                 const void* dout, const float* lse, void* dq, void* dk, void* dv, int B, int H, int Tq,
                                 int Tk, int dh, int ldq, int ldk, int ldv, int ldo, int lddq, int lddk, int lddv,
                                 float dropout_p, uint64_t dropout_seed, void* workspace, size_t workspace_bytes, int dtype16,
                                 void* stream) {
  if (!q || !k || !v || !out || !dout || !lse || !dq || !dk || !dv || !workspace) return VMC_E_ARG;
  int rc = check_generic(B, H, Tq, Tk, dh, ldq, ldk, ldv, ldo);
  if (rc) return rc;
  if (workspace_bytes < vmc_attention_bwd_workspace_bytes(B, H, Tq)) return VMC_E_ARG;
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)dout) & 15) return VMC_E_ALIGN;
  const float scale = 1.0f / sqrtf((float)dh);
  if ((dh == 64 || dh == 96) && ((lddq | lddk | lddv | ldo) % 4) == 0) {   // MFMA path when the head fits in LDS
    const int TQP = (Tq + 31) & ~31, TKP = (Tk + 31) & ~31;
    if ((size_t)2 * (TQP + TKP) * dh * 2 + (size_t)2 * TQP * sizeof(float) <= 160 * 1024) {
      hipStream_t st = (hipStream_t)stream;
      if (dtype16 == VMC_BF16)
        return dh == 64 ? launch_bwd_mfma<BF16, 64>(q, k, v, key_mask, out, dout, lse, dq, dk, dv, B, H, Tq, Tk, ldq, ldk, ldv, ldo, lddq, lddk, lddv, scale, dropout_p, dropout_seed, st)
                        : launch_bwd_mfma<BF16, 96>(q, k, v, key_mask, out, dout, lse, dq, dk, dv, B, H, Tq, Tk, ldq, ldk, ldv, ldo, lddq, lddk, lddv, scale, dropout_p, dropout_seed, st);
      if (dtype16 == VMC_F16)
        return dh == 64 ? launch_bwd_mfma<F16, 64>(q, k, v, key_mask, out, dout, lse, dq, dk, dv, B, H, Tq, Tk, ldq, ldk, ldv, ldo, lddq, lddk, lddv, scale, dropout_p, dropout_seed, st)
                        : launch_bwd_mfma<F16, 96>(q, k, v, key_mask, out, dout, lse, dq, dk, dv, B, H, Tq, Tk, ldq, ldk, ldv, ldo, lddq, lddk, lddv, scale, dropout_p, dropout_seed, st);
      return VMC_E_DTYPE;
    }
  }
  float* delta = (float*)workspace;
  hipStream_t s = (hipStream_t)stream;
#define VMC_LAUNCH_BWD(TT)                                                                                                   \
  hipLaunchKernelGGL(attn_generic_bwd_q<TT>, dim3(B * H * Tq), dim3(64), 0, s, (const uint16_t*)q, (const uint16_t*)k,       \
                     (const uint16_t*)v, key_mask, (const uint16_t*)out, (const uint16_t*)dout, lse, (uint16_t*)dq, delta, H, \
                     Tq, Tk, dh, ldq, ldk, ldv, ldo, lddq, scale, dropout_p, (unsigned long long)dropout_seed);                                                            \
  hipLaunchKernelGGL(attn_generic_bwd_kv<TT>, dim3(B * H * Tk), dim3(64), 0, s, (const uint16_t*)q, (const uint16_t*)k,      \
                     (const uint16_t*)v, key_mask, (const uint16_t*)dout, lse, delta, (uint16_t*)dk, (uint16_t*)dv, H, Tq, Tk, \
                     dh, ldq, ldk, ldv, ldo, lddk, lddv, scale, dropout_p, (unsigned long long)dropout_seed);
  if (dtype16 == VMC_BF16) {
    VMC_LAUNCH_BWD(BF16)
  } else if (dtype16 == VMC_F16) {
    VMC_LAUNCH_BWD(F16)
  } else {
    return VMC_E_DTYPE;
  }
#undef VMC_LAUNCH_BWD
  VMC_CHECK_LAUNCH();
  return 0;
}
