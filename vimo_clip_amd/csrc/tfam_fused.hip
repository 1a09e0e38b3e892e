// Fused TFAM forward chain for short clips (gfx950): the small-batch, weight-streaming regime of
// TFAM/models/AMO_CLIP.py:37-51 (AttentionLayer.forward), :84,:170 (mean-pool + classifier).
//
// At the reference batch (8 clips x 16 tokens = 128 token rows) a layer is ~15 MB of 16-bit weights against
// 0.25 GFLOP: every linear is a skinny GEMM whose only cost is getting its weight rows onto the chip once and
// its 128..256 activation rows into every workgroup.  The chain below is what is left after removing everything
// that is not a true all-to-all dependency (cdna_hip_programming.md 5.6: "cut at every all-to-all seam"):
//
//   kv     K|V of ALL layers' cross attention = motion tokens x [L*2D, D]^T      (hoisted: AMO_CLIP.py:43-45
//          projects the same raw motion tokens in every layer)
//   per layer (6 launches; the three post-norm LayerNorms, the two attentions, the bias / ReLU / residual adds
//   and the fp32 -> 16-bit casts all live in prologues / epilogues):
//     1  qkv  = LN_prev(y) Wqkv^T + b            prologue LayerNorm (layer 0: the raw fp32 tokens), side output x = LN_prev(y)
//     2  y    = x + softmax(q k^T) v Wo^T + b    prologue: masked self attention of the block's clips, written straight
//                                                into the LDS image of the GEMM's A operand
//     3  q    = LN_self(y) Wq^T + b              side output x1 = LN_self(y)
//     4  y    = x1 + attn(q, K_l, V_l) Wo^T + b  cross attention against the hoisted K|V
//     5  h    = relu(LN_cross(y) W1^T + b)       side output x2 = LN_cross(y)
//     6  y    = x2 + h W2^T + b                  K = dim_feedforward: LDS-DMA ring over K chunks
//   pool   16-bit LN_cls(mean_t LN_ffn(y))       mean over ALL T rows, padded ones included (AMO_CLIP.py:170)
//   head   GELU_erf(. W^T + b), then the class logits (fp32)
//
// Geometry of one GEMM workgroup: 256 threads, 32 token rows x BN (16/32/64) output columns, the full K (<= 768 ... 1024)
// of both operands resident in LDS: W rows arrive by LDS-DMA (global_load_lds_dwordx4, full 128-B lines, XOR-16
// swizzle applied on the per-lane SOURCE address), A rows go through registers because they are transformed on the
// way (LayerNorm / cast / attention).  Waves split the tile 2 (row tiles of 16) x 2 (K halves); the two K halves are
// summed through LDS.  MFMA operands are passed as (W, X) so a lane owns 4 consecutive output columns of one row.
// Workgroups that share a W tile are given block ids that agree mod 8 (same XCD, same L2).
//
// Supported: T, Tk <= 32, d_model in {512, 768}, head_dim in {64, 96}, (clips per block) * nhead a multiple of 4.
// Everything else returns VMC_E_SHAPE and the caller uses the general per-op path.
#include "tfam_kernels.h"

// ---- pack layout ---------------------------------------------------------------------------------------------------------
// 16-bit weight pack: per layer [self_in 3D x D | self_out D x D | cross_q D x D | cross_out D x D | ffn0 ff x D | ffn3 D x ff],
// then [kv_all (L*2D) x D | cls1 (D/2) x D | cls4 C x (D/2)].  kv_all rows l*2D..: rows D:3D of layer l's cross in_proj.
// fp32 parameter pack: per layer [b_self_in 3D | b_self_out D | b_cross_q D | b_cross_out D | b_ffn0 ff | b_ffn3 D |
// norm_self g,b | norm_cross g,b | norm_ffn g,b], then [b_kv_all L*2D | cls_ln g,b | b_cls1 D/2 | b_cls4 C].
static size_t tf_w_layer(int D, int ff) { return (size_t)6 * D * D + (size_t)2 * ff * D; }
static size_t tf_p_layer(int D, int ff) { return (size_t)3 * D + 4 * (size_t)D + ff + 6 * (size_t)D; }   // 3D + D + D + D + ff + D + 3 x 2D

extern "C" long long vmc_tfam_pack_offset(int slot, int layer, int D, int ff, int L, int C) {
  (void)C;
  const size_t wl = tf_w_layer(D, ff), pl = tf_p_layer(D, ff);
  const size_t DD = (size_t)D * D;
  switch (slot) {
    case VMC_TFAM_W_SELF_IN: return (long long)(layer * wl);
    case VMC_TFAM_W_SELF_OUT: return (long long)(layer * wl + 3 * DD);
    case VMC_TFAM_W_CROSS_Q: return (long long)(layer * wl + 4 * DD);
    case VMC_TFAM_W_CROSS_OUT: return (long long)(layer * wl + 5 * DD);
    case VMC_TFAM_W_FFN0: return (long long)(layer * wl + 6 * DD);
    case VMC_TFAM_W_FFN3: return (long long)(layer * wl + 6 * DD + (size_t)ff * D);
    case VMC_TFAM_W_KV_ALL: return (long long)(L * wl + (size_t)layer * 2 * DD);
    case VMC_TFAM_W_CLS1: return (long long)(L * wl + (size_t)L * 2 * DD);
    case VMC_TFAM_W_CLS4: return (long long)(L * wl + (size_t)L * 2 * DD + (size_t)(D / 2) * D);
    case VMC_TFAM_W_END: return (long long)(L * wl + (size_t)L * 2 * DD + (size_t)(D / 2) * D + (size_t)C * (D / 2));
    case VMC_TFAM_P_SELF_IN_B: return (long long)(layer * pl);
    case VMC_TFAM_P_SELF_OUT_B: return (long long)(layer * pl + 3 * D);
    case VMC_TFAM_P_CROSS_Q_B: return (long long)(layer * pl + 4 * D);
    case VMC_TFAM_P_CROSS_OUT_B: return (long long)(layer * pl + 5 * D);
    case VMC_TFAM_P_FFN0_B: return (long long)(layer * pl + 6 * D);
    case VMC_TFAM_P_FFN3_B: return (long long)(layer * pl + 6 * D + ff);
    case VMC_TFAM_P_NORM_SELF: return (long long)(layer * pl + 7 * D + ff);        // gamma, then beta at +D
    case VMC_TFAM_P_NORM_CROSS: return (long long)(layer * pl + 9 * D + ff);
    case VMC_TFAM_P_NORM_FFN: return (long long)(layer * pl + 11 * D + ff);
    case VMC_TFAM_P_KV_ALL_B: return (long long)(L * pl + (size_t)layer * 2 * D);
    case VMC_TFAM_P_CLS_LN: return (long long)(L * pl + (size_t)L * 2 * D);
    case VMC_TFAM_P_CLS1_B: return (long long)(L * pl + (size_t)L * 2 * D + 2 * D);
    case VMC_TFAM_P_CLS4_B: return (long long)(L * pl + (size_t)L * 2 * D + 2 * D + D / 2);
    case VMC_TFAM_P_END: return (long long)(L * pl + (size_t)L * 2 * D + 2 * D + D / 2 + ((C + 3) & ~3));
    default: return -1;
  }
}

// workspace: [y f32 M*D][xa f32 M*D][xb f32 M*D][qkv16 M*3D][q16 M*D][h16 M*ff][kv16 Mk*L*2D][pool16 B*D][g16 B*D/2]
//            [q frag][self-k frag][cross-k frag x L]   (fragment-major, tf_frag_off; the row-major q / k columns stay unused)
namespace {
struct TfWs {
  float *y, *xa, *xb;
  uint16_t *qkv, *q, *h, *kv, *pool, *g;
  uint16_t *qf, *kf, *kxf;      // fragment-major q, self k, and (per layer) cross k
  size_t kxf_stride;            // elements between two layers' cross-k buffers
  size_t bytes;
};
inline TfWs tf_ws(void* base, const TfDims& d) {
  const size_t M = (size_t)d.B * d.T, Mk = (size_t)d.B * (d.has_cross ? d.Tk : 0);
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  char* p = (char*)base;
  TfWs w;
  size_t o = 0;
  w.y = (float*)(p + o); o += al(M * d.D * 4);
  w.xa = (float*)(p + o); o += al(M * d.D * 4);
  w.xb = (float*)(p + o); o += al(M * d.D * 4);
  w.qkv = (uint16_t*)(p + o); o += al(M * 3 * d.D * 2);
  w.q = (uint16_t*)(p + o); o += al(M * d.D * 2);
  w.h = (uint16_t*)(p + o); o += al(M * d.ff * 2);
  w.kv = (uint16_t*)(p + o); o += al(Mk * d.L * 2 * d.D * 2);
  w.pool = (uint16_t*)(p + o); o += al((size_t)d.B * d.D * 2);
  w.g = (uint16_t*)(p + o); o += al((size_t)d.B * (d.D / 2) * 2);
  const int H = d.H > 0 ? d.H : 8, dh = d.D / H;
  const size_t fe = tf_frag_elems(d.B, H, dh, tf_ntt(d.T > d.Tk ? d.T : d.Tk));
  w.qf = (uint16_t*)(p + o); o += al(fe * 2);
  w.kf = (uint16_t*)(p + o); o += al(fe * 2);
  w.kxf = (uint16_t*)(p + o); w.kxf_stride = al(fe * 2) / 2; o += (d.has_cross ? d.L : 0) * al(fe * 2);
  w.bytes = o;
  return w;
}

template <typename T, int PRO, int EPI>
int tf_gemm_k(TfArgs& a, int bn, hipStream_t s) {
  switch (a.K) {
    case 768: return tf_dispatch_bn<T, PRO, EPI, 768>(a, bn, s);
    case 512: return tf_dispatch_bn<T, PRO, EPI, 512>(a, bn, s);
    case 384: if constexpr (PRO == PRO_16) return tf_dispatch_bn<T, PRO, EPI, 384>(a, bn, s); else return VMC_E_SHAPE;
    case 256: if constexpr (PRO == PRO_16) return tf_dispatch_bn<T, PRO, EPI, 256>(a, bn, s); else return VMC_E_SHAPE;
    default: return VMC_E_SHAPE;
  }
}

inline TfArgs tf_kv_layer_args(const float* motion, const uint16_t* wp, const float* pp, int layer, const TfDims& d, const TfWs& w);

// stand-alone K|V projection of every layer (vmc_tfam_kv_fwd; vmc_tfam_forward pairs each layer's with its qkv launch instead)
template <typename T>
int tf_kv_impl(const float* motion, const uint16_t* wp, const float* pp, const TfDims& d, const TfWs& w, hipStream_t s) {
  for (int l = 0; l < d.L; ++l) {
    TfArgs a = tf_kv_layer_args(motion, wp, pp, l, d, w);
    if (int rc = tf_gemm_k<T, PRO_F32, EPI_ACT16>(a, tf_pick_bn(a.M, a.N, a.rpb, a.K, false), s)) return rc;
  }
  return 0;
}

// a layer's qkv projection and its cross-attention K|V projection in ONE launch
template <typename T, int PRO>
int tf_qkv_kv_pair(TfArgs& a, TfArgs& b, int bn, hipStream_t s) {
#define TF_PAIR(BNV) (a.K == 768 ? tf_launch_pair<T, BNV, PRO, EPI_ACT16, 768>(a, b, s) : tf_launch_pair<T, BNV, PRO, EPI_ACT16, 512>(a, b, s))
  TF_BN_SWITCH(bn, TF_PAIR)
#undef TF_PAIR
}

// K|V of ONE layer (rows layer*2D.. of kv_all), written into that layer's columns of ws.kv
inline TfArgs tf_kv_layer_args(const float* motion, const uint16_t* wp, const float* pp, int layer, const TfDims& d, const TfWs& w) {
  TfArgs a = {};
  a.A = motion; a.lda = d.D;
  a.M = d.B * d.Tk; a.N = 2 * d.D; a.K = d.D; a.rpb = 32;
  a.W = wp + vmc_tfam_pack_offset(VMC_TFAM_W_KV_ALL, layer, d.D, d.ff, d.L, d.C); a.ldw = d.D;
  a.bias = pp + vmc_tfam_pack_offset(VMC_TFAM_P_KV_ALL_B, layer, d.D, d.ff, d.L, d.C);
  a.out = w.kv + (size_t)layer * 2 * d.D; a.ldo = d.L * 2 * d.D; a.act = VMC_ACT_NONE;
  a.frag[0] = w.kxf + (size_t)layer * w.kxf_stride; a.frag[1] = nullptr;      // K columns fragment-major, V columns row-major
  a.frag_D = d.D; a.frag_T = d.Tk; a.frag_H = d.H; a.frag_DH = d.D / d.H; a.frag_NTT = tf_ntt(d.Tk);
  return a;
}

// column tile of a paired launch: the widest that keeps both problems inside one resident round (1 workgroup per CU)
inline int tf_pick_bn_pair(int Ma, int Na, int rpba, int Mb, int Nb, int K) {
  const int cands[4] = {16, 32, 48, 64};
  int best = 64;
  for (int i = 0; i < 4; ++i) {
    const int bn = cands[i];
    if ((Na % bn) || (Nb % bn) || (size_t)(TF_BM + bn) * K * 2 > TF_LDS_MAX) continue;
    best = bn;
    const long blocks = (long)(Na / bn) * ((Ma + rpba - 1) / rpba) + (long)(Nb / bn) * ((Mb + 31) / 32);
    if (blocks <= 256) break;
  }
  return best;
}

// one AttentionLayer.  x_in: fp32 tokens of layer 0 (null for later layers: the input is then LN_ffn[layer-1](w.y)).
template <typename T>
int tf_layer_impl(const float* x_in, const uint8_t* mask, const uint8_t* mask_kv, const uint16_t* wp, const float* pp, int layer,
                  const TfDims& d, const TfWs& w, hipStream_t s, const float* merge_kv_motion = nullptr) {
  const int M = d.B * d.T, D = d.D, dh = D / d.H;
  // row blocks: two whole clips (T <= 16), one clip (T <= 32), or -- longer clips -- uniform 32-row blocks for the row-wise GEMMs
  // and (clip, 32-query part) blocks for the two attention launches
  const int cpb = d.T <= 16 ? 2 : 1, parts = d.T > 32 ? (d.T + 31) / 32 : 1, rpb = d.T > 32 ? 32 : cpb * d.T;
  const float scale = 1.0f / sqrtf((float)dh);
  auto W = [&](int slot) { return wp + vmc_tfam_pack_offset(slot, layer, D, d.ff, d.L, d.C); };
  auto P = [&](int slot, int l) { return pp + vmc_tfam_pack_offset(slot, l, D, d.ff, d.L, d.C); };
  int rc;
  const float* resid;
  {  // 1: qkv
    TfArgs a = {};
    a.M = M; a.N = 3 * D; a.K = D; a.rpb = rpb;
    a.W = W(VMC_TFAM_W_SELF_IN); a.ldw = D; a.bias = P(VMC_TFAM_P_SELF_IN_B, layer);
    a.out = w.qkv; a.ldo = 3 * D; a.act = VMC_ACT_NONE;
    a.frag[0] = w.qf; a.frag[1] = w.kf; a.frag_D = D; a.frag_T = d.T; a.frag_H = d.H; a.frag_DH = dh;    // V columns stay row-major
    a.frag_NTT = tf_ntt(d.T);
    const bool pair = merge_kv_motion != nullptr && d.has_cross;
    TfArgs b = {};
    int bn = tf_pick_bn(M, a.N, rpb, D, false);
    if (pair) {
      b = tf_kv_layer_args(merge_kv_motion, wp, pp, layer, d, w);
      bn = tf_pick_bn_pair(M, a.N, rpb, b.M, b.N, D);
    }
    if (x_in != nullptr) {
      a.A = x_in; a.lda = D;
      rc = pair ? tf_qkv_kv_pair<T, PRO_F32>(a, b, bn, s) : tf_gemm_k<T, PRO_F32, EPI_ACT16>(a, bn, s);
      resid = x_in;
    } else {
      a.A = w.y; a.lda = D; a.eps = 1e-5f;
      a.ln_g = P(VMC_TFAM_P_NORM_FFN, layer - 1); a.ln_b = a.ln_g + D; a.xout = w.xa;
      rc = pair ? tf_qkv_kv_pair<T, PRO_LN>(a, b, bn, s) : tf_gemm_k<T, PRO_LN, EPI_ACT16>(a, bn, s);
      resid = w.xa;
    }
    if (rc) return rc;
  }
  auto keyrows = [](int tk) { return tk > 32 ? 64 : (tk > 16 ? 32 : 16); };
  const int bn_attn = tf_pick_bn(M, D, rpb, D, true, (cpb - 1) * d.T + keyrows(d.T));
  const int bn_cross = tf_pick_bn(M, D, rpb, D, true, (cpb - 1) * d.Tk + keyrows(d.Tk));
  {  // 2: y = resid + selfattn(qkv) Wo^T + b
    TfArgs a = {};
    a.M = M; a.N = D; a.K = D; a.rpb = rpb; a.cpb = cpb;
    a.q = w.qf; a.k = w.kf; a.v = w.qkv + 2 * D; a.ldv = 3 * D;
    a.parts = parts; a.ntt_q = a.ntt_k = tf_ntt(d.T);
    a.kmask = mask; a.T = d.T; a.Tk = d.T; a.H = d.H; a.B = d.B; a.scale = scale;
    a.W = W(VMC_TFAM_W_SELF_OUT); a.ldw = D; a.bias = P(VMC_TFAM_P_SELF_OUT_B, layer);
    a.resid = resid; a.ldres = D; a.out = w.y; a.ldo = D;
    if ((rc = (D == 768 ? tf_dispatch_attn<T, 768>(a, bn_attn, dh, s) : tf_dispatch_attn<T, 512>(a, bn_attn, dh, s)))) return rc;
  }
  const float* ln_g = P(VMC_TFAM_P_NORM_SELF, layer);
  const float* x2 = nullptr;
  if (d.has_cross) {
    {  // 3: q = LN_self(y) Wq^T + b ; xb = LN_self(y)
      TfArgs a = {};
      a.M = M; a.N = D; a.K = D; a.rpb = rpb;
      a.A = w.y; a.lda = D; a.eps = 1e-5f; a.ln_g = ln_g; a.ln_b = ln_g + D; a.xout = w.xb;
      a.W = W(VMC_TFAM_W_CROSS_Q); a.ldw = D; a.bias = P(VMC_TFAM_P_CROSS_Q_B, layer);
      a.out = w.q; a.ldo = D; a.act = VMC_ACT_NONE;
      a.frag[0] = w.qf; a.frag[1] = nullptr; a.frag_D = D; a.frag_T = d.T; a.frag_H = d.H; a.frag_DH = dh; a.frag_NTT = tf_ntt(d.T);
      if ((rc = tf_gemm_k<T, PRO_LN, EPI_ACT16>(a, tf_pick_bn(M, a.N, rpb, D, false), s))) return rc;
    }
    {  // 4: y = xb + crossattn(q, K_l, V_l) Wo^T + b
      TfArgs a = {};
      a.M = M; a.N = D; a.K = D; a.rpb = rpb; a.cpb = cpb;
      a.q = w.qf;
      a.k = w.kxf + (size_t)layer * w.kxf_stride; a.v = w.kv + (size_t)layer * 2 * D + D; a.ldv = d.L * 2 * D;
      a.parts = parts; a.ntt_q = tf_ntt(d.T); a.ntt_k = tf_ntt(d.Tk);
      a.kmask = mask_kv; a.T = d.T; a.Tk = d.Tk; a.H = d.H; a.B = d.B; a.scale = scale;
      a.W = W(VMC_TFAM_W_CROSS_OUT); a.ldw = D; a.bias = P(VMC_TFAM_P_CROSS_OUT_B, layer);
      a.resid = w.xb; a.ldres = D; a.out = w.y; a.ldo = D;
      if ((rc = (D == 768 ? tf_dispatch_attn<T, 768>(a, bn_cross, dh, s) : tf_dispatch_attn<T, 512>(a, bn_cross, dh, s)))) return rc;
    }
    ln_g = P(VMC_TFAM_P_NORM_CROSS, layer);
  }
  {  // 5: h = relu(LN(y) W1^T + b) ; xa = LN(y)
    TfArgs a = {};
    a.M = M; a.N = d.ff; a.K = D; a.rpb = rpb;
    a.A = w.y; a.lda = D; a.eps = 1e-5f; a.ln_g = ln_g; a.ln_b = ln_g + D; a.xout = w.xa;
    a.W = W(VMC_TFAM_W_FFN0); a.ldw = D; a.bias = P(VMC_TFAM_P_FFN0_B, layer);
    a.out = w.h; a.ldo = d.ff; a.act = VMC_ACT_RELU;
    if ((rc = tf_gemm_k<T, PRO_LN, EPI_ACT16>(a, tf_pick_bn(M, a.N, rpb, D, false), s))) return rc;
    x2 = w.xa;
  }
  {  // 6: y = xa + h W2^T + b
    TfArgs a = {};
    a.M = M; a.N = D; a.K = d.ff; a.rpb = rpb;
    a.A = w.h; a.lda = d.ff;
    a.W = W(VMC_TFAM_W_FFN3); a.ldw = d.ff; a.bias = P(VMC_TFAM_P_FFN3_B, layer);
    a.resid = x2; a.ldres = D; a.out = w.y; a.ldo = D;
    if ((rc = tf_launch_ring<T, 16, 512, 3>(a, s))) return rc;
  }
  return 0;
}

template <typename T>
int tf_head_impl(const uint16_t* wp, const float* pp, float* logits, const TfDims& d, const TfWs& w, hipStream_t s) {
  const int D = d.D;
  const float* lnf = pp + vmc_tfam_pack_offset(VMC_TFAM_P_NORM_FFN, d.L - 1, D, d.ff, d.L, d.C);
  const float* lnc = pp + vmc_tfam_pack_offset(VMC_TFAM_P_CLS_LN, 0, D, d.ff, d.L, d.C);
  if (D == 768) hipLaunchKernelGGL((tf_pool_kernel<T, 768>), dim3(d.B), dim3(256), 0, s, w.y, lnf, lnf + D, lnc, lnc + D, w.pool, d.T, 1e-5f);
  else hipLaunchKernelGGL((tf_pool_kernel<T, 512>), dim3(d.B), dim3(256), 0, s, w.y, lnf, lnf + D, lnc, lnc + D, w.pool, d.T, 1e-5f);
  VMC_CHECK_LAUNCH();
  int rc;
  {
    TfArgs a = {};
    a.M = d.B; a.N = D / 2; a.K = D; a.rpb = 32;
    a.A = w.pool; a.lda = D;
    a.W = wp + vmc_tfam_pack_offset(VMC_TFAM_W_CLS1, 0, D, d.ff, d.L, d.C); a.ldw = D;
    a.bias = pp + vmc_tfam_pack_offset(VMC_TFAM_P_CLS1_B, 0, D, d.ff, d.L, d.C);
    a.out = w.g; a.ldo = D / 2; a.act = VMC_ACT_GELU_ERF;
    if ((rc = tf_gemm_k<T, PRO_16, EPI_ACT16>(a, 16, s))) return rc;
  }
  {
    TfArgs a = {};
    a.M = d.B; a.N = d.C; a.K = D / 2; a.rpb = 32;
    a.A = w.g; a.lda = D / 2;
    a.W = wp + vmc_tfam_pack_offset(VMC_TFAM_W_CLS4, 0, D, d.ff, d.L, d.C); a.ldw = D / 2;
    a.bias = pp + vmc_tfam_pack_offset(VMC_TFAM_P_CLS4_B, 0, D, d.ff, d.L, d.C);
    a.out = logits; a.ldo = d.C;
    if ((rc = tf_gemm_k<T, PRO_16, EPI_BIAS32>(a, 16, s))) return rc;
  }
  return 0;
}
}  // namespace

extern "C" int vmc_tfam_fold_layernorm(const float* W, const float* bias, const float* gamma, const float* beta, void* w16_out,
                                       float* bias_out, int rows, int cols, int dtype16, void* stream) {
  if (!W || !bias || !gamma || !beta || !w16_out || !bias_out || rows <= 0 || cols <= 0) return VMC_E_ARG;
  if (cols % 4) return VMC_E_SHAPE;
  if (((uintptr_t)W | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)w16_out) & 7) return VMC_E_ALIGN;
  const dim3 grid((rows + 3) / 4), block(256);
  if (dtype16 == VMC_BF16)
    hipLaunchKernelGGL((tf_fold_ln_kernel<BF16>), grid, block, 0, (hipStream_t)stream, W, bias, gamma, beta, (uint16_t*)w16_out, bias_out, rows, cols);
  else if (dtype16 == VMC_F16)
    hipLaunchKernelGGL((tf_fold_ln_kernel<F16>), grid, block, 0, (hipStream_t)stream, W, bias, gamma, beta, (uint16_t*)w16_out, bias_out, rows, cols);
  else return VMC_E_DTYPE;
  VMC_CHECK_LAUNCH();
  return 0;
}

extern "C" size_t vmc_tfam_workspace_bytes(int B, int T, int Tk, int D, int ff, int L, int C, int has_cross) {
  TfDims d = {B, T, Tk, D, 8, ff, L, C, has_cross};      // the head count does not change any size (frag buffers: B * D * 32 elements)
  return tf_ws(nullptr, d).bytes;
}

#define TF_DT(call)                                   \
  if (dtype16 == VMC_BF16) return call<BF16>;         \
  if (dtype16 == VMC_F16) return call<F16>;           \
  return VMC_E_DTYPE

extern "C" int vmc_tfam_kv_fwd(const float* motion, const void* wpack, const float* ppack, void* ws, size_t ws_bytes, int B, int T, int Tk,
                               int D, int H, int ff, int L, int C, int dtype16, void* stream) {
  TfDims d = {B, T, Tk, D, H, ff, L, C, 1};
  if (int rc = tf_check(d)) return rc;
  const TfWs w = tf_ws(ws, d);
  if (ws == nullptr || ws_bytes < w.bytes) return VMC_E_ARG;
  if (dtype16 == VMC_BF16) return tf_kv_impl<BF16>(motion, (const uint16_t*)wpack, ppack, d, w, (hipStream_t)stream);
  if (dtype16 == VMC_F16) return tf_kv_impl<F16>(motion, (const uint16_t*)wpack, ppack, d, w, (hipStream_t)stream);
  return VMC_E_DTYPE;
}

extern "C" int vmc_tfam_layer_fwd(const float* x_in, const uint8_t* mask, const uint8_t* mask_kv, const void* wpack, const float* ppack,
                                  int layer, void* ws, size_t ws_bytes, int B, int T, int Tk, int D, int H, int ff, int L, int C,
                                  int has_cross, int dtype16, void* stream) {
  TfDims d = {B, T, Tk, D, H, ff, L, C, has_cross};
  if (int rc = tf_check(d)) return rc;
  if (layer < 0 || layer >= L || (layer == 0) != (x_in != nullptr)) return VMC_E_ARG;
  const TfWs w = tf_ws(ws, d);
  if (ws == nullptr || ws_bytes < w.bytes) return VMC_E_ARG;
  if (dtype16 == VMC_BF16)
    return tf_layer_impl<BF16>(x_in, mask, mask_kv, (const uint16_t*)wpack, ppack, layer, d, w, (hipStream_t)stream);
  if (dtype16 == VMC_F16)
    return tf_layer_impl<F16>(x_in, mask, mask_kv, (const uint16_t*)wpack, ppack, layer, d, w, (hipStream_t)stream);
  return VMC_E_DTYPE;
}

extern "C" int vmc_tfam_head_fwd(const void* wpack, const float* ppack, float* logits, void* ws, size_t ws_bytes, int B, int T, int Tk, int D,
                                 int H, int ff, int L, int C, int has_cross, int dtype16, void* stream) {
  TfDims d = {B, T, Tk, D, H, ff, L, C, has_cross};
  if (int rc = tf_check(d)) return rc;
  const TfWs w = tf_ws(ws, d);
  if (ws == nullptr || ws_bytes < w.bytes) return VMC_E_ARG;
  if (dtype16 == VMC_BF16) return tf_head_impl<BF16>((const uint16_t*)wpack, ppack, logits, d, w, (hipStream_t)stream);
  if (dtype16 == VMC_F16) return tf_head_impl<F16>((const uint16_t*)wpack, ppack, logits, d, w, (hipStream_t)stream);
  return VMC_E_DTYPE;
}

extern "C" int vmc_tfam_forward(const float* x, const float* motion, const uint8_t* mask, const uint8_t* mask_kv, const void* wpack,
                                const float* ppack, float* logits, void* ws, size_t ws_bytes, int B, int T, int Tk, int D, int H, int ff,
                                int L, int C, int has_cross, int dtype16, void* stream) {
  TfDims d = {B, T, Tk, D, H, ff, L, C, has_cross};
  if (int rc = tf_check(d)) return rc;
  if (!x || !wpack || !ppack || !logits || (has_cross && !motion)) return VMC_E_ARG;
  const TfWs w = tf_ws(ws, d);
  if (ws == nullptr || ws_bytes < w.bytes) return VMC_E_ARG;
  if (dtype16 != VMC_BF16 && dtype16 != VMC_F16) return VMC_E_DTYPE;
  hipStream_t s = (hipStream_t)stream;
  for (int l = 0; l < L; ++l) {
    const float* xin = l == 0 ? x : nullptr;
    const float* mkv = has_cross ? motion : nullptr;     // each layer's K|V projection of the motion tokens rides its qkv launch
    const int rc = dtype16 == VMC_BF16 ? tf_layer_impl<BF16>(xin, mask, mask_kv, (const uint16_t*)wpack, ppack, l, d, w, s, mkv)
                                       : tf_layer_impl<F16>(xin, mask, mask_kv, (const uint16_t*)wpack, ppack, l, d, w, s, mkv);
    if (rc) return rc;
  }
  return vmc_tfam_head_fwd(wpack, ppack, logits, ws, ws_bytes, B, T, Tk, D, H, ff, L, C, has_cross, dtype16, stream);
}
