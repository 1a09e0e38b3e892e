// Device code and launch helpers of the fused TFAM chains (gfx950): tfam_fused.hip (eval forward, folded LayerNorm weights)
// and tfam_train.hip (training forward + backward) instantiate what they need from here.
#pragma once
#include "common.h"

namespace {

constexpr int TF_BM = 32;
constexpr int TF_NTH = 256;
constexpr int TF_MAX_T = 64;      // tokens per clip (queries in parts of 32 per row block, up to four key tiles)

enum { PRO_F32 = 0, PRO_LN = 1, PRO_16 = 2, PRO_ATTN = 3, PRO_LNBWD = 4 };
enum { EPI_ACT16 = 0, EPI_RESID32 = 1, EPI_BIAS32 = 2 };

struct TfArgs {
  const void* A;            // PRO_F32 / PRO_LN: float [M, lda]; PRO_16: 16-bit [M, lda]
  int lda;
  const float* ln_g;        // PRO_LN
  const float* ln_b;
  float eps;
  float* xout;              // PRO_LN, optional: LN(A) as fp32 [M, K] (residual operand of a later launch)
  const uint16_t* q;        // PRO_ATTN: q / k in FRAGMENT-MAJOR layout (tf_frag_off), v row-major [B*Tk, ldv] (head h at columns h*DH..)
  const uint16_t* k;
  const uint16_t* v;
  int ldv;
  uint16_t* frag[2];        // EPI_ACT16, optional: output columns [i*frag_D, (i+1)*frag_D) go to frag[i] in fragment-major layout
  int frag_D, frag_T, frag_H, frag_DH;   // instead of `out` (the q / k operands of the fused attention prologue)
  const uint8_t* kmask;     // [B, Tk], 1 = attend; may be null
  int T, Tk, H, B, cpb;     // cpb: clips per row block
  float scale;
  const uint16_t* W;        // [N, K] 16-bit, row stride ldw
  int ldw;
  const float* bias;
  const float* resid;       // EPI_RESID32: fp32 [M, ldres]
  int ldres;
  void* out;
  int ldo;
  int M, N, K;
  int rpb;                  // token rows per row block (<= 32)
  int n_tiles, n_rb;
  int parts;                // PRO_ATTN launches of clips longer than 32 tokens: row block rb = (clip rb / parts, query rows 32 (rb % parts)..);
                            // 0 / 1: uniform blocks of rpb rows (cpb whole clips)
  int ntt_q, ntt_k;         // PRO_ATTN: token tiles per (clip, head) record of the q / k fragment buffers (tf_ntt)
  int frag_NTT;             // EPI_ACT16 fragment outputs: token tiles per record (0 = 2)
  int act;
  // ---- training chain (TR instantiations only; tfam_train.hip) ----------------------------------------------------------
  int ln_affine;            // PRO_LN: gamma / beta applied in the prologue (unfolded weights)
  uint16_t* x16out;         // PRO_LN / PRO_F32: the 16-bit A operand rows [M, K] (this block's BN columns): the weight gradient's X
  float p_attn;             // PRO_ATTN: dropout on the attention probabilities (index ((clip H + h) T + t) Tk + key, as vmc_attention_*)
  uint64_t seed_attn;
  float* lse;               // PRO_ATTN: [B, H, T] log-sum-exp of the scaled scores (vmc_attention_bwd's input)
  uint16_t* oout;           // PRO_ATTN: normalised attention output rows [M, K] 16-bit
  float p_drop1, p_drop2;   // epilogue dropouts on (acc + bias): index grow * N + col, as vmc_dropout on the [M, N] tensor
  uint64_t seed1, seed2;    //   EPI_RESID32: out = resid + drop2(drop1(acc + bias));  EPI_ACT16: out = drop1(act(acc + bias))
  uint8_t* keep_out;        // EPI_RESID32: [M, N] 1 = kept by every mask (the backward's branch-gradient mask)
  uint16_t* zout;           // EPI_ACT16: pre-activation [M, ldo]
  const uint16_t* gate;     // EPI_ACT16 (backward): out = acc * (gate > 0 ? gate_scale : 0): ReLU' and the FFN dropout from the saved h
  float gate_scale;
  const float* dxin;        // PRO_LNBWD: upstream gradient wrt the LayerNorm output, fp32 [M, K]; A = the LayerNorm INPUT y
  const uint8_t* keep_in;   // PRO_LNBWD: [M, K] keep mask of the branch dropout in front of the LayerNorm (null: none)
  float keep_scale;         //   product of the 1 / (1 - p) of those dropouts
  float* dyout;             // PRO_LNBWD: d(pre-norm sum) fp32 [M, K] (this block's BN columns): the residual path's gradient
  uint16_t* d16out;         // PRO_LNBWD: masked branch gradient, 16-bit [M, K] (the weight gradient's dY)
  float* stats;             // PRO_LNBWD: [M, 2] mean, rstd of the rows of y (n tile 0 writes them; LayerNorm parameter gradients)
};

__device__ __forceinline__ int swz16(int chunk, int row) { return chunk ^ (row & 15); }

// First clip and first query token of row block rb, and the global row of its local row `lrow` (valid = a real row of the block).
__device__ __forceinline__ void tf_block_origin(const TfArgs& a, int rb, int& clip0, int& tq0) {
  if (a.parts > 1) {
    clip0 = rb / a.parts;
    tq0 = 32 * (rb - clip0 * a.parts);
  } else {
    clip0 = rb * a.cpb;
    tq0 = 0;
  }
}
__device__ __forceinline__ int tf_grow(const TfArgs& a, int rb, int lrow, bool& valid) {
  if (a.parts > 1) {
    const int clip = rb / a.parts, t = 32 * (rb - clip * a.parts) + lrow;
    valid = lrow < 32 && t < a.T;
    return clip * a.T + min(t, a.T - 1);
  }
  const int g = rb * a.rpb + lrow;
  valid = lrow < a.rpb && g < a.M;
  return min(g, a.M - 1);
}

// Fragment-major layout of a q / k matrix: the 16 B an MFMA lane needs (token r of a 16-token tile, head-dim chunk 4 kk + qq)
// sit at lane (16 qq + r) x 16 B of a 1-KiB record per (clip, head, token tile, kk) -- one fully coalesced wave load per
// fragment, where row-major [token][feature] rows give 16 rows x 64 B per instruction (measured: the fragment-shaped loads
// of 2 (clip, head) pairs cost 3.7 us of a 9.9 us launch).  Element offset of (clip, head, token t, head-dim d):
// NTT = token tiles of 16 per (clip, head): tf_ntt(T) = max(2, ceil(T / 16)) (clips of up to 32 tokens keep the two-tile records).
__host__ __device__ __forceinline__ int tf_ntt(int T) { return T <= 32 ? 2 : (T + 15) >> 4; }
__host__ __device__ __forceinline__ size_t tf_frag_off(int clip, int head, int t, int d, int H, int DH, int NTT = 2) {
  const int KK = DH >> 5;
  return ((((size_t)(clip * H + head) * NTT + (t >> 4)) * KK + (d >> 5)) * 64 + ((d >> 3) & 3) * 16 + (t & 15)) * 8 + (d & 7);
}
__host__ __device__ __forceinline__ size_t tf_frag_elems(int clips, int H, int DH, int NTT = 2) {
  return (size_t)clips * H * NTT * (DH >> 5) * 512;
}

// (n tile, row block) of a block id: blocks that share a W tile agree mod 8 -> same XCD (round-robin dispatch).
__device__ __forceinline__ void tf_block_map(int bid, int n_tiles, int n_rb, int& nt, int& rb) {
  const int n8 = n_tiles & ~7;
  if (bid < n8 * n_rb) {
    const int g = bid >> 3;
    rb = g % n_rb;
    nt = (g / n_rb) * 8 + (bid & 7);
  } else {
    const int rem = bid - n8 * n_rb, tail = n_tiles - n8;
    nt = n8 + rem % tail;
    rb = rem / tail;
  }
}

// W tile [BN rows x K] -> LDS image by LDS-DMA; image chunk p = (row, phys) holds logical chunk phys ^ (row & 15).
__device__ __forceinline__ void tf_stage_w(const TfArgs& a, char* w_img, int n0, int BN, int wave, int lane, int nw) {
  const int cpr = a.K >> 3;                    // 16-B chunks per row
  const int total = (BN * cpr) >> 6;           // wave instructions (64 chunks each)
  for (int ii = wave; ii < total; ii += nw) {
    const int p = ii * 64 + lane;
    const int row = p / cpr, phys = p - row * cpr;
    const int nrow = min(n0 + row, a.N - 1);   // rows past N (odd head widths) re-read the last row; never stored
    const uint16_t* src = a.W + (size_t)nrow * a.ldw + (swz16(phys, row) << 3);
    __builtin_amdgcn_global_load_lds((const VMC_GLOBAL void*)src, (VMC_LDS void*)(w_img + ii * 1024), 16, 0, 0);
  }
}

// ---- A operand producers --------------------------------------------------------------------------------------------
// thread t: row t / TPR, TPR = 2 NW threads per row, float4 at columns 4*(t8 + TPR i).
// The GEMM consumes z = (y - mean) * rstd: gamma is folded into the packed weight columns and beta W^T into the packed bias
// (LN(y) W^T + b = z (W . gamma)^T + (b + W beta)), so the 32 row groups do not each re-read gamma and beta.  The fp32 side
// output x = z * gamma + beta (the residual operand of a later launch) needs them only for this block's BN columns.
template <typename T, int KD, int BN, int NW, bool TR = false>
__device__ __forceinline__ void tf_pro_ln(const TfArgs& a, char* a_img, int rb, int nt, int tid) {
  constexpr int TPR = 2 * NW;                    // threads per row (32 rows x TPR = 64 NW threads)
  constexpr int NI = KD / (4 * TPR);
  const int row = tid / TPR, t8 = tid % TPR;
  const int grow = min(rb * a.rpb + min(row, a.rpb - 1), a.M - 1);
  const float* src = (const float*)a.A + (size_t)grow * a.lda;
  float4 x[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) x[i] = *(const float4*)(src + 4 * (t8 + TPR * i));
  float4 gam[TR ? NI : 1], bet[TR ? NI : 1];      // training: issued with the row loads, so the prologue stays ONE memory round trip
  if constexpr (TR) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      gam[i] = *(const float4*)(a.ln_g + 4 * (t8 + TPR * i));
      bet[i] = *(const float4*)(a.ln_b + 4 * (t8 + TPR * i));
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) s += (x[i].x + x[i].y) + (x[i].z + x[i].w);
#pragma unroll
  for (int o = 1; o < TPR; o <<= 1) s += __shfl_xor(s, o, 64);
  const float mean = s * (1.0f / KD);
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    x[i].x -= mean; x[i].y -= mean; x[i].z -= mean; x[i].w -= mean;
    v += (x[i].x * x[i].x + x[i].y * x[i].y) + (x[i].z * x[i].z + x[i].w * x[i].w);
  }
#pragma unroll
  for (int o = 1; o < TPR; o <<= 1) v += __shfl_xor(v, o, 64);
  const float rstd = rsqrtf(v * (1.0f / KD) + a.eps);
  const bool live = row < a.rpb && rb * a.rpb + row < a.M;
  const bool emit = a.xout != nullptr && live;
  float* xo = a.xout + (size_t)grow * KD;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c4 = t8 + TPR * i;
    float4 z;
    z.x = x[i].x * rstd; z.y = x[i].y * rstd; z.z = x[i].z * rstd; z.w = x[i].w * rstd;
    if constexpr (TR) {
      // training: unfolded weights, so the GEMM consumes LN(y) itself; its 16-bit rows are also the weight gradient's X operand
      const float4 g = gam[i], b = bet[i];
      z = make_float4(z.x * g.x + b.x, z.y * g.y + b.y, z.z * g.z + b.z, z.w * g.w + b.w);
      const uint2 z16 = make_uint2(pack2<T>(z.x, z.y), pack2<T>(z.z, z.w));
      if (live && ((4 * c4) / BN) % a.n_tiles == nt) {
        if (a.xout != nullptr) *(float4*)(xo + 4 * c4) = z;
        if (a.x16out != nullptr) *(uint2*)(a.x16out + (size_t)grow * KD + 4 * c4) = z16;
      }
      *(uint2*)(a_img + row * (KD * 2) + (swz16(c4 >> 1, row) << 4) + ((c4 & 1) << 3)) = z16;
    } else {
      if (emit && ((4 * c4) / BN) % a.n_tiles == nt) {            // this block's BN columns of the fp32 side output
        const float4 g = *(const float4*)(a.ln_g + 4 * c4), b = *(const float4*)(a.ln_b + 4 * c4);
        *(float4*)(xo + 4 * c4) = make_float4(z.x * g.x + b.x, z.y * g.y + b.y, z.z * g.z + b.z, z.w * g.w + b.w);
      }
      *(uint2*)(a_img + row * (KD * 2) + (swz16(c4 >> 1, row) << 4) + ((c4 & 1) << 3)) =
          make_uint2(pack2<T>(z.x, z.y), pack2<T>(z.z, z.w));
    }
  }
}

// Backward of a post-norm block tail, as the prologue of the GEMM that consumes the branch gradient:
//   y = x + drop(branch), x_out = LN(y)   =>   dy = rstd (g - mean(g) - z mean(g z)),  g = dx_out . gamma,  z = (y - mean) rstd
// dy (fp32) is the gradient of the residual path, keep . dy . keep_scale (16-bit) is the gradient of the branch: the GEMM's A
// operand (dgrad of the branch's last linear) and, saved, the dY operand of that linear's weight gradient.  The row statistics are
// recomputed from the saved y (the rows pass through the registers anyway).
template <typename T, int KD, int BN, int NW>
__device__ __forceinline__ void tf_pro_lnbwd(const TfArgs& a, char* a_img, int rb, int nt, int tid) {
  constexpr int TPR = 2 * NW;
  constexpr int NI = KD / (4 * TPR);
  const int row = tid / TPR, t8 = tid % TPR;
  const int grow = min(rb * a.rpb + min(row, a.rpb - 1), a.M - 1);
  const float* src = (const float*)a.A + (size_t)grow * a.lda;
  const float* dsrc = a.dxin + (size_t)grow * KD;
  float4 z[NI], g[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) z[i] = *(const float4*)(src + 4 * (t8 + TPR * i));
#pragma unroll
  for (int i = 0; i < NI; ++i) g[i] = *(const float4*)(dsrc + 4 * (t8 + TPR * i));
  uint32_t keep[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
    keep[i] = a.keep_in != nullptr ? *(const uint32_t*)(a.keep_in + (size_t)grow * KD + 4 * (t8 + TPR * i)) : 0x01010101u;
  float4 gam[NI];                                // every global load of the prologue is issued before the first reduction
#pragma unroll
  for (int i = 0; i < NI; ++i) gam[i] = *(const float4*)(a.ln_g + 4 * (t8 + TPR * i));
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) s += (z[i].x + z[i].y) + (z[i].z + z[i].w);
#pragma unroll
  for (int o = 1; o < TPR; o <<= 1) s += __shfl_xor(s, o, 64);
  const float mean = s * (1.0f / KD);
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    z[i].x -= mean; z[i].y -= mean; z[i].z -= mean; z[i].w -= mean;
    v += (z[i].x * z[i].x + z[i].y * z[i].y) + (z[i].z * z[i].z + z[i].w * z[i].w);
  }
#pragma unroll
  for (int o = 1; o < TPR; o <<= 1) v += __shfl_xor(v, o, 64);
  const float rstd = rsqrtf(v * (1.0f / KD) + a.eps);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const float4 gm = gam[i];
    z[i].x *= rstd; z[i].y *= rstd; z[i].z *= rstd; z[i].w *= rstd;
    g[i].x *= gm.x; g[i].y *= gm.y; g[i].z *= gm.z; g[i].w *= gm.w;
    s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
    s2 += (g[i].x * z[i].x + g[i].y * z[i].y) + (g[i].z * z[i].z + g[i].w * z[i].w);
  }
#pragma unroll
  for (int o = 1; o < TPR; o <<= 1) {
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  s1 *= 1.0f / KD;
  s2 *= 1.0f / KD;
  const bool live = row < a.rpb && rb * a.rpb + row < a.M;
  if (live && nt == 0 && t8 == 0 && a.stats != nullptr) *(float2*)(a.stats + 2 * (size_t)grow) = make_float2(mean, rstd);
  const float ks = a.keep_scale;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c4 = t8 + TPR * i;
    float4 d;
    d.x = rstd * (g[i].x - s1 - z[i].x * s2); d.y = rstd * (g[i].y - s1 - z[i].y * s2);
    d.z = rstd * (g[i].z - s1 - z[i].z * s2); d.w = rstd * (g[i].w - s1 - z[i].w * s2);
    const uint32_t k4 = keep[i];
    const uint2 d16 = make_uint2(pack2<T>((k4 & 0xFFu) ? d.x * ks : 0.f, (k4 & 0xFF00u) ? d.y * ks : 0.f),
                                 pack2<T>((k4 & 0xFF0000u) ? d.z * ks : 0.f, (k4 & 0xFF000000u) ? d.w * ks : 0.f));
    if (live && ((4 * c4) / BN) % a.n_tiles == nt) {
      if (a.dyout != nullptr) *(float4*)(a.dyout + (size_t)grow * KD + 4 * c4) = d;
      if (a.d16out != nullptr) *(uint2*)(a.d16out + (size_t)grow * KD + 4 * c4) = d16;
    }
    *(uint2*)(a_img + row * (KD * 2) + (swz16(c4 >> 1, row) << 4) + ((c4 & 1) << 3)) = d16;
  }
}

template <typename T, int NW, bool TR = false>
__device__ __forceinline__ void tf_pro_f32(const TfArgs& a, char* a_img, int rb, int tid, int nt = 0, int BN = 16) {
  constexpr int TPR = 2 * NW;
  const int row = tid / TPR, t8 = tid % TPR;
  const int grow = min(rb * a.rpb + min(row, a.rpb - 1), a.M - 1);
  const float* src = (const float*)a.A + (size_t)grow * a.lda;
  const int ni = a.K / (4 * TPR);
  const bool emit16 = TR && a.x16out != nullptr && row < a.rpb && rb * a.rpb + row < a.M;
  for (int i0 = 0; i0 < ni; i0 += 8) {
    float4 x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i0 + i < ni) x[i] = *(const float4*)(src + 4 * (t8 + TPR * (i0 + i)));
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i0 + i < ni) {
        const int c4 = t8 + TPR * (i0 + i);
        const uint2 x16 = make_uint2(pack2<T>(x[i].x, x[i].y), pack2<T>(x[i].z, x[i].w));
        *(uint2*)(a_img + row * (a.K * 2) + (swz16(c4 >> 1, row) << 4) + ((c4 & 1) << 3)) = x16;
        if constexpr (TR) {                       // the cast rows are the weight gradient's X operand
          if (emit16 && ((4 * c4) / BN) % a.n_tiles == nt) *(uint2*)(a.x16out + (size_t)grow * a.K + 4 * c4) = x16;
        }
      }
  }
}

// 16-bit A rows by LDS-DMA (same image as W)
__device__ __forceinline__ void tf_pro_16(const TfArgs& a, char* a_img, int rb, int wave, int lane, int nw) {
  const int cpr = a.K >> 3;
  const int total = (TF_BM * cpr) >> 6;
  for (int ii = wave; ii < total; ii += nw) {
    const int p = ii * 64 + lane;
    const int row = p / cpr, phys = p - row * cpr;
    const int grow = min(rb * a.rpb + min(row, a.rpb - 1), a.M - 1);
    const uint16_t* src = (const uint16_t*)a.A + (size_t)grow * a.lda + (swz16(phys, row) << 3);
    __builtin_amdgcn_global_load_lds((const VMC_GLOBAL void*)src, (VMC_LDS void*)(a_img + ii * 1024), 16, 0, 0);
  }
}

// V rows of the block's clips -> LDS image [vrows][D] by LDS-DMA: vrows = (cpb-1)*Tk + 16*NKT, NKT = key tiles (Tk <= 16: keys
// 16..31 of the 32-key P V step are fed as zeros, no LDS rows); rows past the data repeat the last key (finite values under
// a zero probability).
__device__ __forceinline__ void tf_stage_v(const TfArgs& a, char* v_img, int c0, int vrows, int wave, int lane, int nw) {
  const int cpr = a.K >> 3;
  const int total = (vrows * cpr + 63) >> 6;
  for (int ii = wave; ii < total; ii += nw) {
    const int p = min(ii * 64 + lane, vrows * cpr - 1);
    const int row = p / cpr, phys = p - row * cpr;
    const int ci = min(row / a.Tk, a.cpb - 1);
    const int key = min(row - ci * a.Tk, a.Tk - 1);
    const int clip = min(c0 + ci, a.B - 1);
    const uint16_t* src = a.v + ((size_t)clip * a.Tk + key) * a.ldv + (swz16(phys, row) << 3);
    __builtin_amdgcn_global_load_lds((const VMC_GLOBAL void*)src, (VMC_LDS void*)(v_img + ii * 1024), 16, 0, 0);
  }
}

// Masked attention of the block's (clip, head) pairs; one wave per pair, PB pairs in flight per wave.  S^T = K Q^T with the
// K and Q fragments loaded straight from global (16 rows x 64 B per instruction), P stays in registers as the B operand
// of O^T = V^T P^T, V^T fragments by ds_read_b64_tr_b16 from the LDS image.  O (normalised) is written as the 16-bit A
// operand of the out_proj GEMM.
template <int DH, int QT, int NKT, int NW>
struct TfAttnFrags {
  static constexpr int KK = DH / 32, PB = (16 / NW) / QT;     // (clip, head) pairs in flight per wave
  uint4 kf[PB][NKT][KK], qf[PB][QT][KK];
  uint32_t livebits[2];
};

// K / Q fragments of this wave's first PB (clip, head) pairs (fragment-major buffers: one coalesced 1-KiB load each) + the
// liveness of this lane's keys: plain global loads, issued together with the LDS-DMA of the V and W images so that one round
// trip covers all of them.  Token rows >= T (or Tk) of a tile were never written: they only reach masked scores / unused rows.
template <int DH, int QT, int NKT, int NW>
__device__ __forceinline__ void tf_attn_load(const TfArgs& a, TfAttnFrags<DH, QT, NKT, NW>& f, int c0, int tq0, int p0, int lane) {
  constexpr int KK = DH / 32, PB = (16 / NW) / QT;
  const int npairs = a.cpb * a.H;
  const int nq = a.ntt_q > 0 ? a.ntt_q : 2, nk = a.ntt_k > 0 ? a.ntt_k : 2, qt0 = tq0 >> 4;
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int p = min(p0 + i, npairs - 1);
    const int ci = p / a.H, h = p - ci * a.H;
    const int clip = min(c0 + ci, a.B - 1);
    const size_t krec = (size_t)(clip * a.H + h) * nk * KK;     // 1-KiB records of this (clip, head): [token tile][kk]
    const size_t qrec = (size_t)(clip * a.H + h) * nq * KK;
#pragma unroll
    for (int nt = 0; nt < NKT; ++nt)
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) f.kf[i][nt][kk] = *(const uint4*)(a.k + (krec + min(nt, nk - 1) * KK + kk) * 512 + lane * 8);
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) f.qf[i][qt][kk] = *(const uint4*)(a.q + (qrec + min(qt0 + qt, nq - 1) * KK + kk) * 512 + lane * 8);
  }
}

// livebits[ci] bit 4 nt + j: key 16 nt + 4 q + j of clip ci exists and is not masked (nt < 4)
__device__ __forceinline__ void tf_attn_live(const TfArgs& a, uint32_t (&livebits)[2], int c0, int lane) {
  const int q = lane >> 4;
  livebits[0] = livebits[1] = 0u;
#pragma unroll
  for (int ci = 0; ci < 2; ++ci) {
    const int clip = min(c0 + min(ci, a.cpb - 1), a.B - 1);
    const uint8_t* mk = a.kmask != nullptr ? a.kmask + (size_t)clip * a.Tk : nullptr;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int key = 16 * nt + 4 * q + j;
        if (key < a.Tk && (mk == nullptr || mk[key] != 0)) livebits[ci] |= 1u << (4 * nt + j);
      }
  }
}

// Masked attention of the block's (clip, head) pairs; one wave per pair, PB pairs in flight per wave.  S^T = K Q^T with the
// K and Q fragments loaded straight from global (16 rows x 64 B per instruction), P stays in registers as the B operand
// of O^T = V^T P^T, V^T fragments by ds_read_b64_tr_b16 from the LDS image.  O (normalised) is written as the 16-bit A
// operand of the out_proj GEMM.
template <typename T, int DH, int QT, int NKT, int NW, bool TR = false>
__device__ __forceinline__ void tf_pro_attn(const TfArgs& a, TfAttnFrags<DH, QT, NKT, NW>& f, char* a_img, const char* v_img, int c0, int tq0,
                                            int wave, int lane, int nt_blk = 0) {
  constexpr int KK = DH / 32, DT = DH / 16, PB = (16 / NW) / QT, NKS = (NKT + 1) / 2;      // NKS: 32-key steps of P V
  const int r = lane & 15, q = lane >> 4;
  const int npairs = a.cpb * a.H;
  const int rowb = a.K * 2;
  const float c2 = a.scale * 1.4426950408889634f;
  const uint4 ones = make_uint4(T::ONE_PAIR, T::ONE_PAIR, T::ONE_PAIR, T::ONE_PAIR);
  uint64_t seed = 0;
  if constexpr (TR) seed = a.p_attn > 0.f ? resolve_seed(a.seed_attn) : 0;
  for (int p0 = wave * PB; p0 < npairs; p0 += NW * PB) {
    if (p0 != wave * PB) tf_attn_load<DH, QT, NKT, NW>(a, f, c0, tq0, p0, lane);     // later batches (nhead > 8): a round trip of their own
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const bool valid = p0 + i < npairs;           // no early exit: the PB chains are independent and get interleaved
      const int p = min(p0 + i, npairs - 1);
      const int ci = p / a.H, h = p - ci * a.H;
      const uint32_t lb = ci ? f.livebits[1] : f.livebits[0];
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        const int tq = tq0 + 16 * qt + r;           // this lane's query token inside its clip
        f32x4 s[NKT];
#pragma unroll
        for (int nt = 0; nt < NKT; ++nt) {
          s[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kk = 0; kk < KK; ++kk) s[nt] = T::mfma16(f.kf[i][nt][kk], f.qf[i][qt][kk], s[nt]);
        }
        float m = -INFINITY;
#pragma unroll
        for (int nt = 0; nt < NKT; ++nt)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (!((lb >> (4 * nt + j)) & 1u)) s[nt][j] = -INFINITY;
            m = fmaxf(m, s[nt][j]);
          }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        const float mc = m * c2;                  // a fully masked row gives exp2(NaN): NaN output, as torch
        float pe[NKS][8];
        uint4 pf[NKS];
        f32x4 osum = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            pe[ks][j] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[2 * ks][j], c2, -mc));
            pe[ks][4 + j] = 2 * ks + 1 < NKT ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[2 * ks + 1 < NKT ? 2 * ks + 1 : 0][j], c2, -mc)) : 0.f;
          }
          pf[ks] = make_uint4(pack2<T>(pe[ks][0], pe[ks][1]), pack2<T>(pe[ks][2], pe[ks][3]), pack2<T>(pe[ks][4], pe[ks][5]),
                              pack2<T>(pe[ks][6], pe[ks][7]));
          osum = T::mfma16(ones, pf[ks], osum);
        }
        if constexpr (TR) {
          // nn.MultiheadAttention(dropout=p): softmax normalised by the UNdropped sum, P V on p * keep / (1 - p) with the
          // counter-based mask of vmc_attention_fwd / _bwd (same flat (clip, head, query, key) index)
          if (a.p_attn > 0.f) {                   // wave-uniform
            const int clip = min(c0 + ci, a.B - 1);
            const size_t base = (((size_t)clip * a.H + h) * a.T + min(tq, a.T - 1)) * a.Tk + 4 * q;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                pe[ks][j] *= dropout_factor(a.p_attn, seed, base + 32 * ks + j);
                if (2 * ks + 1 < NKT) pe[ks][4 + j] *= dropout_factor(a.p_attn, seed, base + 32 * ks + 16 + j);
              }
              pf[ks] = make_uint4(pack2<T>(pe[ks][0], pe[ks][1]), pack2<T>(pe[ks][2], pe[ks][3]), pack2<T>(pe[ks][4], pe[ks][5]),
                                  pack2<T>(pe[ks][6], pe[ks][7]));
            }
          }
          if (a.lse != nullptr && nt_blk == 0 && valid && q == 0 && tq < a.T) {
            const int clip = c0 + ci;
            if (clip < a.B) a.lse[((size_t)clip * a.H + h) * a.T + tq] = m * a.scale + __logf(osum[0]);
          }
        }
        f32x4 o[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // transposed 4-key x 16-column blocks: this lane supplies key row 32 ks + 4 q + (r>>2) (+16), columns 16 dt + 4 (r&3)..
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          const int vrow0 = ci * a.Tk + 32 * ks + 4 * q + (r >> 2);
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            const int col = h * DH + 16 * dt + 4 * (r & 3);
            const char* p0a = v_img + vrow0 * rowb + (swz16(col >> 3, vrow0) << 4) + (((col >> 2) & 1) << 3);
            const char* p1a = v_img + (vrow0 + 16) * rowb + (swz16(col >> 3, vrow0 + 16) << 4) + (((col >> 2) & 1) << 3);
            const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((VMC_LDS s16x4*)p0a);
            uint2 x1 = make_uint2(0u, 0u);
            if (2 * ks + 1 < NKT) x1 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((VMC_LDS s16x4*)p1a));
            const uint2 x0 = __builtin_bit_cast(uint2, v0);
            o[dt] = T::mfma16(make_uint4(x0.x, x0.y, x1.x, x1.y), pf[ks], o[dt]);
          }
        }
        const float inv = 1.0f / osum[0];
        const int arow = ci * a.T + 16 * qt + r;   // row of the A image (cpb == 1 whenever tq0 > 0 or T > 16)
        if (valid && tq < a.T) {
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            const int col = h * DH + 16 * dt + 4 * q;
            *(uint2*)(a_img + arow * rowb + (swz16(col >> 3, arow) << 4) + (((col >> 2) & 1) << 3)) =
                make_uint2(pack2<T>(o[dt][0] * inv, o[dt][1] * inv), pack2<T>(o[dt][2] * inv, o[dt][3] * inv));
          }
        }
      }
    }
  }
}

__device__ __forceinline__ uint32_t tf_lds_addr(const char* p) { return (uint32_t)(uintptr_t)(const VMC_LDS char*)p; }

// ---- epilogue: one lane's 4 consecutive columns of one row -------------------------------------------------------------
// pre_added: bias (and residual) are already inside v (tf_gemm_body seeds the accumulators with them)
// frag_dst: (EPI_ACT16) where this lane's 4 columns go in a fragment-major side buffer, or nullptr for the row-major output
template <typename T, int EPI>
__device__ __forceinline__ void tf_epilogue(const TfArgs& a, int grow, int col, f32x4 v, bool pre_added = false,
                                            uint16_t* frag_dst = nullptr) {
  if (col >= a.N) return;
  if (col + 3 < a.N) {
    if (!pre_added) {
      const float4 b = *(const float4*)(a.bias + col);
      v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
      if (EPI == EPI_RESID32) {
        const float4 rr = *(const float4*)(a.resid + (size_t)grow * a.ldres + col);
        v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
      }
    }
    if (EPI == EPI_ACT16) {
      const uint2 o = make_uint2(pack2<T>(apply_act_rt(v[0], a.act), apply_act_rt(v[1], a.act)),
                                 pack2<T>(apply_act_rt(v[2], a.act), apply_act_rt(v[3], a.act)));
      if (frag_dst != nullptr) {                    // q / k columns: fragment-major side buffer
        *(uint2*)frag_dst = o;
      } else {
        *(uint2*)((uint16_t*)a.out + (size_t)grow * a.ldo + col) = o;
      }
    } else if ((a.ldo & 3) == 0) {
      *(float4*)((float*)a.out + (size_t)grow * a.ldo + col) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) ((float*)a.out)[(size_t)grow * a.ldo + col + j] = v[j];
    }
  } else {                                       // ragged last columns (class count not a multiple of 4)
    for (int j = 0; j < 4 && col + j < a.N; ++j) {
      float x = v[j] + a.bias[col + j];
      if (EPI == EPI_RESID32) x += a.resid[(size_t)grow * a.ldres + col + j];
      if (EPI == EPI_ACT16) ((uint16_t*)a.out)[(size_t)grow * a.ldo + col + j] = T::from_f32(apply_act_rt(x, a.act));
      else ((float*)a.out)[(size_t)grow * a.ldo + col + j] = x;
    }
  }
}

// Training epilogue (N % 4 == 0).  v holds acc (+ bias when pre_added); rr = the residual operand, prefetched by the caller.
//   EPI_ACT16   z = v (+ bias) -> zout;  o = act(z) [* gate'] [* drop1];  16-bit row-major to `out` (if non-null) AND to frag_dst
//   EPI_RESID32 out = rr + drop2(drop1(v (+ bias)));  keep_out = kept by both masks
//   EPI_BIAS32  out = v (+ bias)
template <typename T, int EPI>
__device__ __forceinline__ void tf_epilogue_train(const TfArgs& a, int grow, int col, f32x4 v, bool pre_added, uint16_t* frag_dst,
                                                  const float4& rr) {
  if (col >= a.N) return;
  if (!pre_added && a.bias != nullptr) {
    const float4 b = *(const float4*)(a.bias + col);
    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
  }
  const size_t e0 = (size_t)grow * a.N + col;      // flat index of this lane's first element in the [M, N] tensor
  if (EPI == EPI_ACT16) {
    if (a.zout != nullptr) *(uint2*)(a.zout + (size_t)grow * a.ldo + col) = make_uint2(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]));
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = apply_act_rt(v[j], a.act);
    if (a.gate != nullptr) {
      const uint2 g = *(const uint2*)(a.gate + (size_t)grow * a.ldo + col);
      float g0, g1, g2, g3;
      unpack2<T>(g.x, g0, g1);
      unpack2<T>(g.y, g2, g3);
      v[0] = g0 > 0.f ? v[0] * a.gate_scale : 0.f; v[1] = g1 > 0.f ? v[1] * a.gate_scale : 0.f;
      v[2] = g2 > 0.f ? v[2] * a.gate_scale : 0.f; v[3] = g3 > 0.f ? v[3] * a.gate_scale : 0.f;
    }
    if (a.p_drop1 > 0.f) {
      const uint64_t s1 = resolve_seed(a.seed1);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] *= dropout_factor(a.p_drop1, s1, e0 + j);
    }
    const uint2 o = make_uint2(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]));
    if (frag_dst != nullptr) *(uint2*)frag_dst = o;
    if (a.out != nullptr) *(uint2*)((uint16_t*)a.out + (size_t)grow * a.ldo + col) = o;
  } else if (EPI == EPI_RESID32) {
    uint32_t keep = 0x01010101u;
    if (a.p_drop1 > 0.f) {
      const uint64_t s1 = resolve_seed(a.seed1);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float f = dropout_factor(a.p_drop1, s1, e0 + j);
        v[j] *= f;
        if (f == 0.f) keep &= ~(0xFFu << (8 * j));
      }
    }
    if (a.p_drop2 > 0.f) {
      const uint64_t s2 = resolve_seed(a.seed2);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float f = dropout_factor(a.p_drop2, s2, e0 + j);
        v[j] *= f;
        if (f == 0.f) keep &= ~(0xFFu << (8 * j));
      }
    }
    if (a.keep_out != nullptr) *(uint32_t*)(a.keep_out + e0) = keep;
    *(float4*)((float*)a.out + (size_t)grow * a.ldo + col) = make_float4(v[0] + rr.x, v[1] + rr.y, v[2] + rr.z, v[3] + rr.w);
  } else {
    *(float4*)((float*)a.out + (size_t)grow * a.ldo + col) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// ---- single-shot K kernel ----------------------------------------------------------------------------------------------
// LDS: [A image 32 x K][W image BN x K][V image (PRO_ATTN)]; the K-slice exchange of the accumulators re-uses the A image
// Every global access of the prologue (A rows / LayerNorm parameters / K, Q fragments / mask bytes / the LDS-DMA of the W
// and V images) is issued before the first wait, so a workgroup pays ONE memory round trip before its MFMAs.
template <typename T, int BN, int PRO, int EPI, int KD, int DH, int QT, int NKT, int NW, bool TR = false>
__device__ __forceinline__ void tf_gemm_body(const TfArgs& a, int bid, char* tf_smem) {
  constexpr int NTN = BN / 16;
  constexpr int ROWB = KD * 2;
  constexpr int KSP = NW / 2;                    // K slices (waves = 2 row tiles x KSP K slices)
  constexpr int KSTEPS = KD / 32 / KSP;          // 32-wide k-steps per slice
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int nt, rb;
  tf_block_map(bid, a.n_tiles, a.n_rb, nt, rb);
  char* a_img = tf_smem;
  char* w_img = a_img + TF_BM * ROWB;
  // more than 32 keys: the V image (up to 64 rows x K) does not fit beside the W tile; W is staged into the SAME region once the
  // attention prologue is done with V (one more exposed LDS-DMA latency, only for these shapes)
  constexpr bool LATEW = PRO == PRO_ATTN && NKT > 2;
  char* v_img = LATEW ? w_img : w_img + BN * ROWB;
  char* red = a_img;                             // K-slice exchange re-uses the A image once every wave is past its MFMAs
  const int n0 = nt * BN;

  if constexpr (PRO == PRO_ATTN) {
    int c0, tq0;
    tf_block_origin(a, rb, c0, tq0);
    TfAttnFrags<DH, QT, NKT, NW> fr;
    tf_attn_load<DH, QT, NKT, NW>(a, fr, c0, tq0, wave * ((16 / NW) / QT), lane);
    tf_attn_live(a, fr.livebits, c0, lane);
    tf_stage_v(a, v_img, c0, (a.cpb - 1) * a.Tk + 16 * NKT, wave, lane, NW);
    if constexpr (!LATEW) tf_stage_w(a, w_img, n0, BN, wave, lane, NW);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                             // V (and W) images complete
    tf_pro_attn<T, DH, QT, NKT, NW, TR>(a, fr, a_img, v_img, c0, tq0, wave, lane, nt);
    if constexpr (LATEW) {
      __syncthreads();                           // every wave is done with the V image
      tf_stage_w(a, w_img, n0, BN, wave, lane, NW);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  } else {
    tf_stage_w(a, w_img, n0, BN, wave, lane, NW);
    if constexpr (PRO == PRO_16) tf_pro_16(a, a_img, rb, wave, lane, NW);
    if constexpr (PRO == PRO_LN) tf_pro_ln<T, KD, BN, NW, TR>(a, a_img, rb, nt, tid);
    if constexpr (PRO == PRO_LNBWD) tf_pro_lnbwd<T, KD, BN, NW>(a, a_img, rb, nt, tid);
    if constexpr (PRO == PRO_F32) tf_pro_f32<T, NW, TR>(a, a_img, rb, tid, nt, BN);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if constexpr (TR && PRO == PRO_ATTN) {
    // this block's BN columns of the attention output (saved for vmc_attention_bwd's delta and the out_proj weight gradient)
    if (a.oout != nullptr && tid < TF_BM * (BN / 8)) {
      const int row = tid / (BN / 8), c8 = (n0 >> 3) + tid % (BN / 8);
      bool rvalid;
      const int grow = tf_grow(a, rb, row, rvalid);
      if (rvalid && c8 * 8 < a.K)
        *(uint4*)(a.oout + (size_t)grow * a.K + c8 * 8) = *(const uint4*)(a_img + row * ROWB + (swz16(c8, row) << 4));
    }
  }

  const int r = lane & 15, q = lane >> 4;
  const int rt = wave & 1, ks = wave >> 1;
  f32x4 acc[NTN];
#pragma unroll
  for (int n = 0; n < NTN; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int arow = 16 * rt + r;
  bool valid_e;
  const int grow_e = tf_grow(a, rb, arow, valid_e);
  const bool store_e = ks == 0 && valid_e;
  float4 rres[NTN];                              // TR: residual operand, prefetched under the MFMA loop
#pragma unroll
  for (int n = 0; n < NTN; ++n) rres[n] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (store_e && a.N % 4 == 0) {
    // the epilogue's bias (+ residual) operands ride under the MFMA loop: start the K-slice-0 accumulators from them
#pragma unroll
    for (int n = 0; n < NTN; ++n) {
      const int col = n0 + 16 * n + 4 * q;
      if (col < a.N) {
        if constexpr (TR) {
          if (a.bias != nullptr) {
            const float4 b = *(const float4*)(a.bias + col);
            acc[n] = (f32x4){b.x, b.y, b.z, b.w};
          }
          if constexpr (EPI == EPI_RESID32) rres[n] = *(const float4*)(a.resid + (size_t)grow_e * a.ldres + col);
        } else {
          const float4 b = *(const float4*)(a.bias + col);
          acc[n] = (f32x4){b.x, b.y, b.z, b.w};
          if constexpr (EPI == EPI_RESID32) {
            const float4 rr = *(const float4*)(a.resid + (size_t)grow_e * a.ldres + col);
            acc[n] += (f32x4){rr.x, rr.y, rr.z, rr.w};
          }
        }
      }
    }
  }
  const bool pre_added = a.N % 4 == 0;
  const char* ap = a_img + arow * ROWB;
  constexpr int KB = KSTEPS % 4 == 0 ? 4 : (KSTEPS % 3 == 0 ? 3 : (KSTEPS % 2 == 0 ? 2 : 1));   // k-steps whose reads are batched
#pragma unroll
  for (int k0 = 0; k0 < KSTEPS; k0 += KB) {
    uint4 af[KB], wf[KB][NTN];
#pragma unroll
    for (int kk = 0; kk < KB; ++kk) {
      const int chunk = (ks * KSTEPS + k0 + kk) * 4 + q;
      af[kk] = *(const uint4*)(ap + (swz16(chunk, arow) << 4));
#pragma unroll
      for (int n = 0; n < NTN; ++n) {
        const int wrow = 16 * n + r;
        wf[kk][n] = *(const uint4*)(w_img + wrow * ROWB + (swz16(chunk, wrow) << 4));
      }
    }
#pragma unroll
    for (int kk = 0; kk < KB; ++kk)
#pragma unroll
      for (int n = 0; n < NTN; ++n) acc[n] = T::mfma16(wf[kk][n], af[kk], acc[n]);
  }
  __syncthreads();
  if (ks > 0) {
#pragma unroll
    for (int n = 0; n < NTN; ++n) *(f32x4*)(red + ((((ks - 1) * 2 + rt) * NTN + n) * 64 + lane) * 16) = acc[n];
  }
  __syncthreads();
  if (ks == 0) {
    const int grow = grow_e;
    if (valid_e) {
      int fclip = 0, ft = 0;
      if (EPI == EPI_ACT16 && a.frag_D > 0) {          // token coordinates of this lane's row: once, not per column tile
        fclip = grow / a.frag_T;
        ft = grow - fclip * a.frag_T;
      }
#pragma unroll
      for (int n = 0; n < NTN; ++n) {
        f32x4 o = acc[n];
#pragma unroll
        for (int s2 = 0; s2 < KSP - 1; ++s2) o += *(const f32x4*)(red + (((s2 * 2 + rt) * NTN + n) * 64 + lane) * 16);
        uint16_t* fdst = nullptr;
        if (EPI == EPI_ACT16 && a.frag_D > 0) {
          const int c0 = __builtin_amdgcn_readfirstlane(n0 + 16 * n);      // wave-uniform: scalar divisions
          const int fi = c0 / a.frag_D;
          if (fi < 2 && a.frag[fi] != nullptr) {      // a 16-column MFMA tile never straddles a head (head_dim % 16 == 0)
            const int c = c0 - fi * a.frag_D, head = c / a.frag_DH, d = c - head * a.frag_DH + 4 * q;
            fdst = a.frag[fi] + tf_frag_off(fclip, head, ft, d, a.frag_H, a.frag_DH, a.frag_NTT > 0 ? a.frag_NTT : 2);
          }
        }
        if constexpr (TR) tf_epilogue_train<T, EPI>(a, grow, n0 + 16 * n + 4 * q, o, pre_added, fdst, rres[n]);
        else tf_epilogue<T, EPI>(a, grow, n0 + 16 * n + 4 * q, o, pre_added, fdst);
      }
    }
  }
}

template <typename T, int BN, int PRO, int EPI, int KD, int DH, int QT, int NKT, int NW, bool TR = false>
__global__ void __launch_bounds__(64 * NW) tf_gemm_kernel(const TfArgs a) {
  extern __shared__ __attribute__((aligned(16))) char tf_smem[];
  tf_gemm_body<T, BN, PRO, EPI, KD, DH, QT, NKT, NW, TR>(a, blockIdx.x, tf_smem);
}

// Two independent problems in one launch: a layer's qkv projection (raw tokens or LayerNorm prologue) + the K|V projection of
// the raw motion tokens for that layer's cross attention: neither depends on the other, so the K|V GEMM costs no launch of its own.
template <typename T, int BN, int PRO, int EPI, int KD, int NW, bool TR = false>
__global__ void __launch_bounds__(64 * NW) tf_gemm_pair_kernel(const TfArgs a, const TfArgs b, int blocks_a) {
  extern __shared__ __attribute__((aligned(16))) char tf_smem[];
  if ((int)blockIdx.x < blocks_a) tf_gemm_body<T, BN, PRO, EPI, KD, 64, 1, 1, NW, TR>(a, blockIdx.x, tf_smem);
  else tf_gemm_body<T, BN, PRO_F32, EPI, KD, 64, 1, 1, NW, TR>(b, blockIdx.x - blocks_a, tf_smem);      // b: always raw fp32 rows
}

// ---- chunked-K kernel (FFN second linear): A 16-bit and W both by LDS-DMA through an NST-deep ring -----------------------
template <int N>
__device__ __forceinline__ void tf_wait_vm() {
  if (N <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <typename T, int BN, int KC, int NST, bool TR = false>
__global__ void __launch_bounds__(TF_NTH) tf_gemm_ring_kernel(const TfArgs a) {
  extern __shared__ __attribute__((aligned(16))) char tf_smem[];
  constexpr int NTN = BN / 16;
  constexpr int CPR = KC / 8;                          // 16-B chunks per image row
  constexpr int A_INSTR = TF_BM * CPR / 64 / 4;        // LDS-DMA instructions per wave per chunk
  constexpr int W_INSTR = (BN * CPR / 64 + 3) / 4;
  constexpr int PER = A_INSTR + W_INSTR;
  constexpr int STAGE = (TF_BM + BN) * KC * 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int nt, rb;
  tf_block_map(blockIdx.x, a.n_tiles, a.n_rb, nt, rb);
  const int n0 = nt * BN;
  const int nchunks = a.K / KC;
  char* red = tf_smem + NST * STAGE;

  auto stage = [&](int c) {
    char* base = tf_smem + (c % NST) * STAGE;
    const size_t koff = (size_t)c * KC;
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
      const int ii = wave + 4 * i;
      const int p = ii * 64 + lane, row = p / CPR, phys = p - row * CPR;
      const int grow = min(rb * a.rpb + min(row, a.rpb - 1), a.M - 1);
      const uint16_t* src = (const uint16_t*)a.A + (size_t)grow * a.lda + koff + (swz16(phys, row) << 3);
      __builtin_amdgcn_global_load_lds((const VMC_GLOBAL void*)src, (VMC_LDS void*)(base + ii * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < W_INSTR; ++i) {
      const int ii = min(wave + 4 * i, BN * CPR / 64 - 1);     // surplus instructions of a wave re-stage the last piece
      const int p = ii * 64 + lane, row = p / CPR, phys = p - row * CPR;
      const uint16_t* src = a.W + (size_t)min(n0 + row, a.N - 1) * a.ldw + koff + (swz16(phys, row) << 3);
      __builtin_amdgcn_global_load_lds((const VMC_GLOBAL void*)src, (VMC_LDS void*)(base + TF_BM * KC * 2 + ii * 1024), 16, 0, 0);
    }
  };

  const int r = lane & 15, q = lane >> 4;
  const int rt = wave & 1, ks = wave >> 1;
  f32x4 acc[NTN];
#pragma unroll
  for (int n = 0; n < NTN; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < NST - 1; ++c)
    if (c < nchunks) stage(c);
  const int arow = 16 * rt + r;
  for (int c = 0; c < nchunks; ++c) {
    if (c + NST - 1 < nchunks) {
      stage(c + NST - 1);
      tf_wait_vm<(NST - 1) * PER>();             // everything but the NST-1 youngest chunks has landed: chunk c is in
    } else {
      tf_wait_vm<0>();
    }
    __builtin_amdgcn_s_barrier();
    const char* a_img = tf_smem + (c % NST) * STAGE;
    const char* w_img = a_img + TF_BM * KC * 2;
    uint4 af[KC / 64], wf[NTN][KC / 64];
#pragma unroll
    for (int kk = 0; kk < KC / 64; ++kk) {
      const int chunk = (ks * (KC / 64) + kk) * 4 + q;
      asm volatile("ds_read_b128 %0, %1" : "=v"(af[kk]) : "v"(tf_lds_addr(a_img + arow * (KC * 2) + (swz16(chunk, arow) << 4))) : "memory");
#pragma unroll
      for (int n = 0; n < NTN; ++n) {
        const int wrow = 16 * n + r;
        asm volatile("ds_read_b128 %0, %1" : "=v"(wf[n][kk]) : "v"(tf_lds_addr(w_img + wrow * (KC * 2) + (swz16(chunk, wrow) << 4))) : "memory");
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < KC / 64; ++kk)
#pragma unroll
      for (int n = 0; n < NTN; ++n) acc[n] = T::mfma16(wf[n][kk], af[kk], acc[n]);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                // ring slot c % NST may be restaged by the next iteration
  }
  if (ks == 1) {
#pragma unroll
    for (int n = 0; n < NTN; ++n) *(f32x4*)(red + ((rt * NTN + n) * 64 + lane) * 16) = acc[n];
  }
  __syncthreads();
  if (ks == 0) {
    const int lrow = 16 * rt + r, grow = rb * a.rpb + lrow;
    if (lrow < a.rpb && grow < a.M) {
#pragma unroll
      for (int n = 0; n < NTN; ++n) {
        const f32x4 o = *(const f32x4*)(red + ((rt * NTN + n) * 64 + lane) * 16);
        if constexpr (TR) {
          const int col = n0 + 16 * n + 4 * q;
          const float4 rr = col < a.N ? *(const float4*)(a.resid + (size_t)grow * a.ldres + col) : make_float4(0.f, 0.f, 0.f, 0.f);
          tf_epilogue_train<T, EPI_RESID32>(a, grow, col, acc[n] + o, false, nullptr, rr);
        } else {
          tf_epilogue<T, EPI_RESID32>(a, grow, n0 + 16 * n + 4 * q, acc[n] + o);
        }
      }
    }
  }
}

// ---- mean-pool + the two LayerNorms of the tail: pooled16[b] = LN_cls(mean_t LN_ffn(y[b, t])) ------------------------------
template <typename T, int D>
__global__ void __launch_bounds__(256) tf_pool_kernel(const float* __restrict__ y, const float* __restrict__ g1, const float* __restrict__ b1,
                                                      const float* __restrict__ g2, const float* __restrict__ b2,
                                                      uint16_t* __restrict__ out, int Tn, float eps, float* __restrict__ pooled32 = nullptr) {
  constexpr int NI = D / 256;                    // float4 per lane per row
  __shared__ float part[4][D];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
  float4 accp[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) accp[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int t = wave; t < Tn; t += 4) {
    const float* row = y + ((size_t)b * Tn + t) * D;
    float4 x[NI];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      x[i] = *(const float4*)(row + 4 * (lane + 64 * i));
      s += (x[i].x + x[i].y) + (x[i].z + x[i].w);
    }
    const float mean = wave_sum(s) * (1.0f / D);
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      x[i].x -= mean; x[i].y -= mean; x[i].z -= mean; x[i].w -= mean;
      v += (x[i].x * x[i].x + x[i].y * x[i].y) + (x[i].z * x[i].z + x[i].w * x[i].w);
    }
    const float rstd = rsqrtf(wave_sum(v) * (1.0f / D) + eps);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const float4 g = *(const float4*)(g1 + 4 * (lane + 64 * i)), bb = *(const float4*)(b1 + 4 * (lane + 64 * i));
      accp[i].x += x[i].x * rstd * g.x + bb.x; accp[i].y += x[i].y * rstd * g.y + bb.y;
      accp[i].z += x[i].z * rstd * g.z + bb.z; accp[i].w += x[i].w * rstd * g.w + bb.w;
    }
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) *(float4*)(&part[wave][4 * (lane + 64 * i)]) = accp[i];
  __syncthreads();
  if (wave == 0) {
    float4 x[NI];
    float s = 0.f;
    const float invT = 1.0f / Tn;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = 4 * (lane + 64 * i);
      const float4 p0 = *(const float4*)(&part[0][c]), p1 = *(const float4*)(&part[1][c]), p2 = *(const float4*)(&part[2][c]),
                   p3 = *(const float4*)(&part[3][c]);
      x[i].x = ((p0.x + p1.x) + (p2.x + p3.x)) * invT; x[i].y = ((p0.y + p1.y) + (p2.y + p3.y)) * invT;
      x[i].z = ((p0.z + p1.z) + (p2.z + p3.z)) * invT; x[i].w = ((p0.w + p1.w) + (p2.w + p3.w)) * invT;
      if (pooled32 != nullptr) *(float4*)(pooled32 + (size_t)b * D + c) = x[i];      // training: the classifier LayerNorm's input
      s += (x[i].x + x[i].y) + (x[i].z + x[i].w);
    }
    const float mean = wave_sum(s) * (1.0f / D);
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      x[i].x -= mean; x[i].y -= mean; x[i].z -= mean; x[i].w -= mean;
      v += (x[i].x * x[i].x + x[i].y * x[i].y) + (x[i].z * x[i].z + x[i].w * x[i].w);
    }
    const float rstd = rsqrtf(wave_sum(v) * (1.0f / D) + eps);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = 4 * (lane + 64 * i);
      const float4 g = *(const float4*)(g2 + c), bb = *(const float4*)(b2 + c);
      *(uint2*)(out + (size_t)b * D + c) = make_uint2(pack2<T>(x[i].x * rstd * g.x + bb.x, x[i].y * rstd * g.y + bb.y),
                                                      pack2<T>(x[i].z * rstd * g.z + bb.z, x[i].w * rstd * g.w + bb.w));
    }
  }
}

// ---- pack time: fold a LayerNorm's affine part into the linear that consumes it -------------------------------------------
// w16[r, k] = W[r, k] * gamma[k] (16-bit), bias_out[r] = bias[r] + sum_k W[r, k] * beta[k] (fp32); one wave per row.
template <typename T>
__global__ void __launch_bounds__(256) tf_fold_ln_kernel(const float* __restrict__ W, const float* __restrict__ bias,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         uint16_t* __restrict__ w16, float* __restrict__ bias_out, int rows, int cols) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* wr = W + (size_t)row * cols;
  float acc = 0.f;
  for (int c = 4 * lane; c < cols; c += 256) {
    const float4 w = *(const float4*)(wr + c), g = *(const float4*)(gamma + c), b = *(const float4*)(beta + c);
    acc += (w.x * b.x + w.y * b.y) + (w.z * b.z + w.w * b.w);
    *(uint2*)(w16 + (size_t)row * cols + c) = make_uint2(pack2<T>(w.x * g.x, w.y * g.y), pack2<T>(w.z * g.z, w.w * g.w));
  }
  acc = wave_sum(acc);
  if (lane == 0) bias_out[row] = bias[row] + acc;
}

// ---- host side ---------------------------------------------------------------------------------------------------------
constexpr size_t TF_LDS_MAX = 160 * 1024;

constexpr int TF_NW = 8;       // waves per workgroup of the single-shot kernels (2 row tiles x 4 K slices)

template <int BN, int PRO, int KD, int NKT>
constexpr size_t tf_lds_bytes(int cpb, int Tk) {
  if (PRO == PRO_ATTN && NKT > 2) {              // V and W share a region (LATEW)
    const size_t v = (size_t)((cpb - 1) * Tk + 16 * NKT) * KD * 2 + 1024, w = (size_t)BN * KD * 2;
    return (size_t)TF_BM * KD * 2 + (v > w ? v : w);
  }
  return (size_t)(TF_BM + BN) * KD * 2 +
         (PRO == PRO_ATTN ? (size_t)((cpb - 1) * Tk + 16 * NKT) * KD * 2 + 1024 : 0);    // + one LDS-DMA piece of slack
}

template <typename K>
int tf_set_lds(K kern, bool& done) {
  if (!done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TF_LDS_MAX);
    if (e != hipSuccess) return (int)e;
    done = true;
  }
  return 0;
}

template <typename T, int BN, int PRO, int EPI, int KD, int DH, int QT, int NKT = 1, bool TR = false>
int tf_launch(TfArgs& a, hipStream_t s) {
  const size_t lds = tf_lds_bytes<BN, PRO, KD, NKT>(a.cpb, a.Tk);
  if (lds > TF_LDS_MAX || a.K != KD) return VMC_E_SHAPE;
  if ((TF_NW / 2 - 1) * 2 * (BN / 16) * 1024 > TF_BM * KD * 2) return VMC_E_SHAPE;      // K-slice exchange must fit the A image
  a.n_tiles = (a.N + BN - 1) / BN;
  a.n_rb = (PRO == PRO_ATTN && a.parts > 1) ? a.B * a.parts : (a.M + a.rpb - 1) / a.rpb;
  auto kern = tf_gemm_kernel<T, BN, PRO, EPI, KD, DH, QT, NKT, TF_NW, TR>;
  static bool attr_done = false;                // per instantiation
  if (int rc = tf_set_lds(kern, attr_done)) return rc;
  hipLaunchKernelGGL(kern, dim3(a.n_tiles * a.n_rb), dim3(64 * TF_NW), lds, s, a);
  VMC_CHECK_LAUNCH();
  return 0;
}

template <typename T, int BN, int PRO, int EPI, int KD, bool TR = false>
int tf_launch_pair(TfArgs& a, TfArgs& b, hipStream_t s) {
  const size_t lds = tf_lds_bytes<BN, PRO, KD, 1>(1, 0);
  if (lds > TF_LDS_MAX || a.K != KD || b.K != KD) return VMC_E_SHAPE;
  a.n_tiles = (a.N + BN - 1) / BN; a.n_rb = (a.M + a.rpb - 1) / a.rpb;
  b.n_tiles = (b.N + BN - 1) / BN; b.n_rb = (b.M + b.rpb - 1) / b.rpb;
  auto kern = tf_gemm_pair_kernel<T, BN, PRO, EPI, KD, TF_NW, TR>;
  static bool attr_done = false;
  if (int rc = tf_set_lds(kern, attr_done)) return rc;
  const int na = a.n_tiles * a.n_rb;
  hipLaunchKernelGGL(kern, dim3(na + b.n_tiles * b.n_rb), dim3(64 * TF_NW), lds, s, a, b, na);
  VMC_CHECK_LAUNCH();
  return 0;
}

template <typename T, int BN, int KC, int NST, bool TR = false>
int tf_launch_ring(TfArgs& a, hipStream_t s) {
  const size_t lds = (size_t)NST * (TF_BM + BN) * KC * 2 + 2 * (BN / 16) * 1024;
  if (lds > TF_LDS_MAX || a.K % KC) return VMC_E_SHAPE;
  a.n_tiles = (a.N + BN - 1) / BN;
  a.n_rb = (a.M + a.rpb - 1) / a.rpb;
  auto kern = tf_gemm_ring_kernel<T, BN, KC, NST, TR>;
  static bool attr_done = false;
  if (int rc = tf_set_lds(kern, attr_done)) return rc;
  hipLaunchKernelGGL(kern, dim3(a.n_tiles * a.n_rb), dim3(TF_NTH), lds, s, a);
  VMC_CHECK_LAUNCH();
  return 0;
}

// Output-column tile.  A workgroup's cost is one memory round trip plus (32 A rows + BN W rows) x K bytes at the ~70 GB/s one
// CU pulls from L2, whatever BN is; what BN decides is how many workgroups there are.  Take the narrowest tile (most CUs
// streaming W) whose grid still fits one resident round (2 workgroups per CU while the LDS footprint allows, else 1).
inline int tf_pick_bn(int M, int N, int rpb, int K, bool attn, int vrows = 48) {
  const int n_rb = (M + rpb - 1) / rpb;
  const int cands[4] = {16, 32, 48, 64};
  int best = 16;
  for (int i = 0; i < 4; ++i) {
    const int bn = cands[i];
    if (N % bn && !(N < bn)) continue;
    if (attn && bn > 32) break;
    if (attn && vrows > 48 && bn > 16) break;    // more than 32 keys: the 16-column kernels (V and W share a region)
    const size_t lds = (attn && vrows > 48) ? (size_t)TF_BM * K * 2 + (size_t)vrows * K * 2 + 1024
                                             : (size_t)(TF_BM + bn) * K * 2 + (attn ? (size_t)vrows * K * 2 + 1024 : 0);
    if (lds > TF_LDS_MAX) break;
    best = bn;
    const int per_cu = lds <= 80 * 1024 ? 2 : 1;
    if ((long)((N + bn - 1) / bn) * n_rb <= 256L * per_cu) break;
  }
  return best;
}

#define TF_BN_SWITCH(bn, CALL)            \
  switch (bn) {                           \
    case 64: return CALL(64);             \
    case 48: return CALL(48);             \
    case 32: return CALL(32);             \
    default: return CALL(16);             \
  }

template <typename T, int PRO, int EPI, int KD>
int tf_dispatch_bn(TfArgs& a, int bn, hipStream_t s) {
#define TF_CALL(BNV) tf_launch<T, BNV, PRO, EPI, KD, 64, 1>(a, s)
  TF_BN_SWITCH(bn, TF_CALL)
#undef TF_CALL
}

template <typename T, int KD, int BN, int DH>
int tf_dispatch_attn3(TfArgs& a, hipStream_t s) {
  const int qt = a.T > 16 ? 2 : 1, nkt = a.Tk > 32 ? 4 : (a.Tk > 16 ? 2 : 1);
  if (nkt == 4) {
    if constexpr (BN == 16) return qt == 1 ? tf_launch<T, 16, PRO_ATTN, EPI_RESID32, KD, DH, 1, 4>(a, s) : tf_launch<T, 16, PRO_ATTN, EPI_RESID32, KD, DH, 2, 4>(a, s);
    else return VMC_E_SHAPE;
  }
  if (qt == 1 && nkt == 1) return tf_launch<T, BN, PRO_ATTN, EPI_RESID32, KD, DH, 1, 1>(a, s);
  if (qt == 1) return tf_launch<T, BN, PRO_ATTN, EPI_RESID32, KD, DH, 1, 2>(a, s);
  if (nkt == 1) return tf_launch<T, BN, PRO_ATTN, EPI_RESID32, KD, DH, 2, 1>(a, s);
  return tf_launch<T, BN, PRO_ATTN, EPI_RESID32, KD, DH, 2, 2>(a, s);
}

template <typename T, int KD>
int tf_dispatch_attn(TfArgs& a, int bn, int dh, hipStream_t s) {
  if (bn == 32) return dh == 64 ? tf_dispatch_attn3<T, KD, 32, 64>(a, s) : tf_dispatch_attn3<T, KD, 32, 96>(a, s);
  return dh == 64 ? tf_dispatch_attn3<T, KD, 16, 64>(a, s) : tf_dispatch_attn3<T, KD, 16, 96>(a, s);
}

struct TfDims {
  int B, T, Tk, D, H, ff, L, C, has_cross;
};

inline int tf_check(const TfDims& d) {
  if (d.B <= 0 || d.T <= 0 || d.T > TF_MAX_T || d.L <= 0 || d.C <= 0) return VMC_E_SHAPE;
  if (d.D != 512 && d.D != 768) return VMC_E_SHAPE;
  if (d.H <= 0 || d.D % d.H) return VMC_E_SHAPE;
  const int dh = d.D / d.H;
  if (dh != 64 && dh != 96) return VMC_E_SHAPE;
  if (d.ff % 512 || d.ff <= 0) return VMC_E_SHAPE;
  if (d.has_cross && (d.Tk <= 0 || d.Tk > TF_MAX_T)) return VMC_E_SHAPE;
  const int cpb = d.T <= 16 ? 2 : 1;
  if ((cpb * d.H) % 4) return VMC_E_SHAPE;
  return 0;
}

}  // namespace
