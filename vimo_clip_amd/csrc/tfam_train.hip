// Fused TFAM TRAINING chains for short clips (gfx950): forward + backward of AttentionLayer / AMO_CLIP in train mode
// (TFAM/models/AMO_CLIP.py:37-51,99-171 under TFAM/train_and_eval.py:66-101).  include/vmc.h ("K11-K14 fused, TRAINING")
// states the contract; tfam_kernels.h holds the GEMM body shared with the eval chain (tfam_fused.hip).
//
// Forward = the eval chain's six launches per layer with the training arithmetic in place (TR instantiations):
//   F1 qkv (+ the layer's K|V projection of the motion tokens in the same launch)   LN affine in the prologue, x16 side output
//   F2 y1 = x + drop(attn(qkv) Wo^T + b)      P-dropout inside the attention prologue, O and lse saved, keep mask saved
//   F3 q  = LN_self(y1) Wq^T + b
//   F4 y2 = x1 + drop(attn(q, K, V) Wo^T + b)
//   F5 h  = drop(relu(LN(y) W1^T + b))
//   F6 y3 = x2 + drop(drop(h W2^T + b))
// Backward of a layer, dgrad chain (every launch depends on the previous one):
//   L1 dh   = [LNbwd_ffn(dx3) . keep3] W2          . relu'/dropout gate from the saved h      (prologue PRO_LNBWD)
//   L2 dx2  = dy3 + dh W1                                                                     (K = ff ring)
//   L3 dOc  = [LNbwd_cross(dx2) . keep2] Wo_c
//   L4 dq, dk|dv = attention backward (vmc_attention_bwd)
//   L5 dx1  = dy2 + dq Wq
//   L6 dOs  = [LNbwd_self(dx1) . keep1] Wo_s
//   L7 dqkv = attention backward
//   L8 dx0  = dy1 + dqkv Wqkv        (= dx3 of the layer below; skipped for layer 0)           (K = 3D ring)
// and ONE grouped launch (tr_wgrad_group_kernel) for the seven weight gradients dW = dY^T X (contraction over the <= 256
// token rows: the TN body of gemm_tn.hip, one 256 x 128 tile per workgroup), their bias gradients (ones-MFMA on the dY
// fragments) and the three LayerNorms' gamma / beta gradients.
#include "tfam_kernels.h"
#include "gemm_tn_body.h"

namespace {

struct TrDims {
  int B, T, Tk, D, H, ff, L, C, has_cross;
};

inline int tr_check(const TrDims& d) {
  TfDims e = {d.B, d.T, d.Tk, d.D, d.H, d.ff, d.L, d.C, d.has_cross};
  if (int rc = tf_check(e)) return rc;
  if (d.B * d.T > 256 || (d.has_cross && d.B * d.Tk > 256)) return VMC_E_SHAPE;      // the grouped weight gradient holds <= 4 token stages
  if (d.C % 4 || d.B > 32) return VMC_E_SHAPE;
  return 0;
}

// ---- workspace ----------------------------------------------------------------------------------------------------------
struct TrLayerWs {
  // saved by the forward
  float* xin32;             // layer > 0: LN_ffn[l-1](y3[l-1]) (the residual operand; layer 0 uses the caller's tokens)
  uint16_t* x0_16;          // 16-bit rows fed to the qkv GEMM
  uint16_t* qkv16;          // [M, 3D]
  uint16_t* o_self;         // [M, D]
  float* lse_self;          // [B, H, T]
  float* y1;                // pre-norm sums (fp32)
  uint8_t* keep1;
  float* x1_32;
  uint16_t* x1_16;
  uint16_t* q16;            // [M, D]
  uint16_t* kv16;           // [Mk, 2D]
  uint16_t* o_cross;
  float* lse_cross;
  float* y2;
  uint8_t* keep2;
  float* x2_32;
  uint16_t* x2_16;
  uint16_t* h16;            // [M, ff] after ReLU and dropout
  float* y3;
  uint8_t* keep3;
  // backward
  float *dx3, *dy3, *dx2, *dy2, *dx1, *dy1;
  uint16_t *d3_16, *dh16, *d2_16, *doc16, *dq16, *dkv16, *d1_16, *dos16, *dqkv16;
  float *st3, *st2, *st1;   // [M, 2] mean, rstd
  float* attn_ws;
};
struct TrWs {
  uint16_t* motion16;       // [Mk, D]
  uint16_t *qf, *kf, *kxf;  // fragment-major q / k operands of the fused attention prologues (forward-only)
  float* pooled32;          // [B, D]
  uint16_t* pool16;         // [B, D]   LN_cls(pooled) 16-bit
  uint16_t* a16;            // [B, D/2] classifier.1 pre-activation
  uint16_t* g16;            // [B, D/2] drop(gelu(a))
  float* da;                // [B, D/2]
  float* dpl;               // [B, D]   gradient wrt LN_cls(pooled)
  char* layers;
  size_t layer_bytes;
  size_t bytes;
};

inline size_t tr_al(size_t x) { return (x + 255) & ~(size_t)255; }

inline TrLayerWs tr_layer_ws(char* base, const TrDims& d, size_t* bytes_out = nullptr) {
  const size_t M = (size_t)d.B * d.T, Mk = (size_t)d.B * (d.has_cross ? d.Tk : 0), D = d.D, ff = d.ff;
  TrLayerWs w;
  size_t o = 0;
  auto take = [&](size_t n) { char* p = base + o; o += tr_al(n); return p; };
  w.xin32 = (float*)take(M * D * 4);
  w.x0_16 = (uint16_t*)take(M * D * 2);
  w.qkv16 = (uint16_t*)take(M * 3 * D * 2);
  w.o_self = (uint16_t*)take(M * D * 2);
  w.lse_self = (float*)take((size_t)d.B * d.H * d.T * 4);
  w.y1 = (float*)take(M * D * 4);
  w.keep1 = (uint8_t*)take(M * D);
  w.x1_32 = (float*)take(M * D * 4);
  w.x1_16 = (uint16_t*)take(M * D * 2);
  w.q16 = (uint16_t*)take(M * D * 2);
  w.kv16 = (uint16_t*)take(Mk * 2 * D * 2);
  w.o_cross = (uint16_t*)take(M * D * 2);
  w.lse_cross = (float*)take((size_t)d.B * d.H * d.T * 4);
  w.y2 = (float*)take(M * D * 4);
  w.keep2 = (uint8_t*)take(M * D);
  w.x2_32 = (float*)take(M * D * 4);
  w.x2_16 = (uint16_t*)take(M * D * 2);
  w.h16 = (uint16_t*)take(M * ff * 2);
  w.y3 = (float*)take(M * D * 4);
  w.keep3 = (uint8_t*)take(M * D);
  w.dx3 = (float*)take(M * D * 4);
  w.dy3 = (float*)take(M * D * 4);
  w.dx2 = (float*)take(M * D * 4);
  w.dy2 = (float*)take(M * D * 4);
  w.dx1 = (float*)take(M * D * 4);
  w.dy1 = (float*)take(M * D * 4);
  w.d3_16 = (uint16_t*)take(M * D * 2);
  w.dh16 = (uint16_t*)take(M * ff * 2);
  w.d2_16 = (uint16_t*)take(M * D * 2);
  w.doc16 = (uint16_t*)take(M * D * 2);
  w.dq16 = (uint16_t*)take(M * D * 2);
  w.dkv16 = (uint16_t*)take(Mk * 2 * D * 2);
  w.d1_16 = (uint16_t*)take(M * D * 2);
  w.dos16 = (uint16_t*)take(M * D * 2);
  w.dqkv16 = (uint16_t*)take(M * 3 * D * 2);
  w.st3 = (float*)take(M * 2 * 4);
  w.st2 = (float*)take(M * 2 * 4);
  w.st1 = (float*)take(M * 2 * 4);
  w.attn_ws = (float*)take((size_t)d.B * d.H * d.T * 4);
  if (bytes_out) *bytes_out = o;
  return w;
}

inline TrWs tr_ws(void* base, const TrDims& d) {
  const size_t Mk = (size_t)d.B * (d.has_cross ? d.Tk : 0), D = d.D;
  char* p = (char*)base;
  TrWs w;
  size_t o = 0;
  auto take = [&](size_t n) { char* q = p + o; o += tr_al(n); return q; };
  w.motion16 = (uint16_t*)take(Mk * D * 2);
  const size_t fe = tf_frag_elems(d.B, d.H, d.D / d.H, tf_ntt(d.T > d.Tk ? d.T : d.Tk));
  w.qf = (uint16_t*)take(fe * 2);
  w.kf = (uint16_t*)take(fe * 2);
  w.kxf = (uint16_t*)take(fe * 2);
  w.pooled32 = (float*)take((size_t)d.B * D * 4);
  w.pool16 = (uint16_t*)take((size_t)d.B * D * 2);
  w.a16 = (uint16_t*)take((size_t)d.B * (D / 2) * 2);
  w.g16 = (uint16_t*)take((size_t)d.B * (D / 2) * 2);
  w.da = (float*)take((size_t)d.B * (D / 2) * 4);
  w.dpl = (float*)take((size_t)d.B * D * 4);
  tr_layer_ws(nullptr, d, &w.layer_bytes);
  w.layers = p + o;
  o += (size_t)d.L * w.layer_bytes;
  w.bytes = o;
  return w;
}
inline TrLayerWs tr_lw(const TrWs& w, const TrDims& d, int layer) { return tr_layer_ws(w.layers + (size_t)layer * w.layer_bytes, d); }

// ---- launch helpers (training instantiations; column tiles 16 / 32 / 64) ---------------------------------------------------
// heavy: a LayerNorm (backward) prologue re-reads 32 fp32 rows (and more) per workgroup -- one workgroup per CU, wider tiles
inline int tr_pick_bn(int M, int N, int rpb, int K, bool attn, bool heavy = false) {
  if (attn) return 16;
  static const int force_ff = getenv("VMC_TR_BN_FF") ? atoi(getenv("VMC_TR_BN_FF")) : 0;      // builder A/B switch
  static const int force_d = getenv("VMC_TR_BN_D") ? atoi(getenv("VMC_TR_BN_D")) : 0;
  if (heavy && force_ff && N >= 1024 && N % force_ff == 0) return force_ff;
  if (heavy && force_d && N < 1024 && N % force_d == 0) return force_d;
  const int n_rb = (M + rpb - 1) / rpb;
  const int cands[3] = {16, 32, 64};
  int best = 16;
  for (int i = 0; i < 3; ++i) {
    const int bn = cands[i];
    if (N % bn) continue;
    const size_t lds = (size_t)(TF_BM + bn) * K * 2;
    if (lds > TF_LDS_MAX) break;
    best = bn;
    const int per_cu = (lds <= 80 * 1024 && !heavy) ? 2 : 1;
    if ((long)(N / bn) * n_rb <= 256L * per_cu) break;
  }
  return best;
}

template <typename T, int PRO, int EPI, int KD>
int tr_dispatch_bn(TfArgs& a, int bn, hipStream_t s) {
  switch (bn) {
    case 64: return tf_launch<T, 64, PRO, EPI, KD, 64, 1, 1, true>(a, s);
    case 32: return tf_launch<T, 32, PRO, EPI, KD, 64, 1, 1, true>(a, s);
    default: return tf_launch<T, 16, PRO, EPI, KD, 64, 1, 1, true>(a, s);
  }
}
template <typename T, int PRO, int EPI>
int tr_gemm(TfArgs& a, hipStream_t s) {
  const int bn = tr_pick_bn(a.M, a.N, a.rpb, a.K, false, PRO == PRO_LN || PRO == PRO_LNBWD);
  if (bn != 16 && a.N % bn) return VMC_E_SHAPE;      // a ragged last tile (140 classes) only at the 16-column tile
  switch (a.K) {
    case 768: return tr_dispatch_bn<T, PRO, EPI, 768>(a, bn, s);
    case 512: return tr_dispatch_bn<T, PRO, EPI, 512>(a, bn, s);
    case 384: if constexpr (PRO == PRO_16) return tr_dispatch_bn<T, PRO, EPI, 384>(a, bn, s); else return VMC_E_SHAPE;
    case 256: if constexpr (PRO == PRO_16) return tr_dispatch_bn<T, PRO, EPI, 256>(a, bn, s); else return VMC_E_SHAPE;
    default: return VMC_E_SHAPE;
  }
}
template <typename T, int KD, int DH>
int tr_attn3(TfArgs& a, hipStream_t s) {
  const int qt = a.T > 16 ? 2 : 1, nkt = a.Tk > 32 ? 4 : (a.Tk > 16 ? 2 : 1);
  if (nkt == 4) return qt == 1 ? tf_launch<T, 16, PRO_ATTN, EPI_RESID32, KD, DH, 1, 4, true>(a, s) : tf_launch<T, 16, PRO_ATTN, EPI_RESID32, KD, DH, 2, 4, true>(a, s);
  if (qt == 1 && nkt == 1) return tf_launch<T, 16, PRO_ATTN, EPI_RESID32, KD, DH, 1, 1, true>(a, s);
  if (qt == 1) return tf_launch<T, 16, PRO_ATTN, EPI_RESID32, KD, DH, 1, 2, true>(a, s);
  if (nkt == 1) return tf_launch<T, 16, PRO_ATTN, EPI_RESID32, KD, DH, 2, 1, true>(a, s);
  return tf_launch<T, 16, PRO_ATTN, EPI_RESID32, KD, DH, 2, 2, true>(a, s);
}
template <typename T>
int tr_attn(TfArgs& a, int dh, hipStream_t s) {
  if (a.K == 768) return dh == 64 ? tr_attn3<T, 768, 64>(a, s) : tr_attn3<T, 768, 96>(a, s);
  return dh == 64 ? tr_attn3<T, 512, 64>(a, s) : tr_attn3<T, 512, 96>(a, s);
}
template <typename T, int PRO>
int tr_pair(TfArgs& a, TfArgs& b, hipStream_t s) {
  // widest tile that keeps both problems inside one resident round; 64 when none does
  int bn = 64;
  for (int c = 32; c <= 64; c *= 2) {
    if ((a.N % c) || (b.N % c)) continue;
    const long blocks = (long)(a.N / c) * ((a.M + a.rpb - 1) / a.rpb) + (long)(b.N / c) * ((b.M + 31) / 32);
    if (blocks <= 256) { bn = c; break; }
  }
  if ((a.N % bn) || (b.N % bn)) return VMC_E_SHAPE;
  if (a.K == 768) return bn == 64 ? tf_launch_pair<T, 64, PRO, EPI_ACT16, 768, true>(a, b, s) : tf_launch_pair<T, 32, PRO, EPI_ACT16, 768, true>(a, b, s);
  return bn == 64 ? tf_launch_pair<T, 64, PRO, EPI_ACT16, 512, true>(a, b, s) : tf_launch_pair<T, 32, PRO, EPI_ACT16, 512, true>(a, b, s);
}
template <typename T>
int tr_ring(TfArgs& a, hipStream_t s) {
  if (a.K % 512 == 0) return tf_launch_ring<T, 16, 512, 3, true>(a, s);
  if (a.K % 384 == 0) return tf_launch_ring<T, 16, 384, 3, true>(a, s);
  return VMC_E_SHAPE;
}

// ---- forward of one layer ------------------------------------------------------------------------------------------------
template <typename T>
int tr_layer_fwd(const float* x_in, const float* motion, const uint8_t* mask, const uint8_t* mask_kv, const vmc_tfam_layer_params* layers,
                 int layer, const TrDims& d, const TrWs& ws, float p, const uint64_t* seeds, hipStream_t s) {
  const int M = d.B * d.T, D = d.D, dh = D / d.H;
  const int cpb = d.T <= 16 ? 2 : 1, parts = d.T > 32 ? (d.T + 31) / 32 : 1, rpb = d.T > 32 ? 32 : cpb * d.T;
  const float scale = 1.0f / sqrtf((float)dh);
  const vmc_tfam_layer_params& P = layers[layer];
  const TrLayerWs w = tr_lw(ws, d, layer);
  const bool drop = p > 0.f;
  int rc;
  const float* resid;
  {  // F1: qkv (+ K|V of the motion tokens)
    TfArgs a = {};
    a.M = M; a.N = 3 * D; a.K = D; a.rpb = rpb;
    a.W = (const uint16_t*)P.w_self_in; a.ldw = D; a.bias = P.b_self_in;
    a.out = w.qkv16; a.ldo = 3 * D; a.act = VMC_ACT_NONE;
    a.frag[0] = ws.qf; a.frag[1] = ws.kf; a.frag_D = D; a.frag_T = d.T; a.frag_H = d.H; a.frag_DH = dh; a.frag_NTT = tf_ntt(d.T);
    a.x16out = w.x0_16;
    TfArgs b = {};
    if (d.has_cross) {
      b.A = motion; b.lda = D; b.M = d.B * d.Tk; b.N = 2 * D; b.K = D; b.rpb = 32;
      b.W = (const uint16_t*)P.w_cross_in + (size_t)D * D; b.ldw = D; b.bias = P.b_cross_in + D;
      b.out = w.kv16; b.ldo = 2 * D; b.act = VMC_ACT_NONE;
      b.frag[0] = ws.kxf; b.frag[1] = nullptr; b.frag_D = D; b.frag_T = d.Tk; b.frag_H = d.H; b.frag_DH = dh; b.frag_NTT = tf_ntt(d.Tk);
      b.x16out = layer == 0 ? ws.motion16 : nullptr;
    }
    if (layer == 0) {
      a.A = x_in; a.lda = D;
      rc = d.has_cross ? tr_pair<T, PRO_F32>(a, b, s) : tr_gemm<T, PRO_F32, EPI_ACT16>(a, s);
      resid = x_in;
    } else {
      const TrLayerWs wp = tr_lw(ws, d, layer - 1);
      a.A = wp.y3; a.lda = D; a.eps = 1e-5f; a.ln_g = layers[layer - 1].ln_ffn_g; a.ln_b = layers[layer - 1].ln_ffn_b;
      a.ln_affine = 1; a.xout = w.xin32;
      rc = d.has_cross ? tr_pair<T, PRO_LN>(a, b, s) : tr_gemm<T, PRO_LN, EPI_ACT16>(a, s);
      resid = w.xin32;
    }
    if (rc) return rc;
  }
  {  // F2: y1 = resid + drop(selfattn(qkv) Wo^T + b)
    TfArgs a = {};
    a.M = M; a.N = D; a.K = D; a.rpb = rpb; a.cpb = cpb;
    a.q = ws.qf; a.k = ws.kf; a.v = w.qkv16 + 2 * D; a.ldv = 3 * D;
    a.parts = parts; a.ntt_q = a.ntt_k = tf_ntt(d.T);
    a.kmask = mask; a.T = d.T; a.Tk = d.T; a.H = d.H; a.B = d.B; a.scale = scale;
    a.W = (const uint16_t*)P.w_self_out; a.ldw = D; a.bias = P.b_self_out;
    a.resid = resid; a.ldres = D; a.out = w.y1; a.ldo = D;
    a.lse = w.lse_self; a.oout = w.o_self;
    if (drop) { a.p_attn = p; a.seed_attn = seeds[0]; a.p_drop1 = p; a.seed1 = seeds[1]; a.keep_out = w.keep1; }
    if ((rc = tr_attn<T>(a, dh, s))) return rc;
  }
  const float* yin = w.y1;
  const float *lng = P.ln_self_g, *lnb = P.ln_self_b;
  if (d.has_cross) {
    {  // F3: q = LN_self(y1) Wq^T + b ; x1 = LN_self(y1)
      TfArgs a = {};
      a.M = M; a.N = D; a.K = D; a.rpb = rpb;
      a.A = w.y1; a.lda = D; a.eps = 1e-5f; a.ln_g = lng; a.ln_b = lnb; a.ln_affine = 1; a.xout = w.x1_32; a.x16out = w.x1_16;
      a.W = (const uint16_t*)P.w_cross_in; a.ldw = D; a.bias = P.b_cross_in;
      a.out = w.q16; a.ldo = D; a.act = VMC_ACT_NONE;
      a.frag[0] = ws.qf; a.frag[1] = nullptr; a.frag_D = D; a.frag_T = d.T; a.frag_H = d.H; a.frag_DH = dh; a.frag_NTT = tf_ntt(d.T);
      if ((rc = tr_gemm<T, PRO_LN, EPI_ACT16>(a, s))) return rc;
    }
    {  // F4: y2 = x1 + drop(crossattn(q, K, V) Wo^T + b)
      TfArgs a = {};
      a.M = M; a.N = D; a.K = D; a.rpb = rpb; a.cpb = cpb;
      a.q = ws.qf; a.k = ws.kxf; a.v = w.kv16 + D; a.ldv = 2 * D;
      a.parts = parts; a.ntt_q = tf_ntt(d.T); a.ntt_k = tf_ntt(d.Tk);
      a.kmask = mask_kv; a.T = d.T; a.Tk = d.Tk; a.H = d.H; a.B = d.B; a.scale = scale;
      a.W = (const uint16_t*)P.w_cross_out; a.ldw = D; a.bias = P.b_cross_out;
      a.resid = w.x1_32; a.ldres = D; a.out = w.y2; a.ldo = D;
      a.lse = w.lse_cross; a.oout = w.o_cross;
      if (drop) { a.p_attn = p; a.seed_attn = seeds[2]; a.p_drop1 = p; a.seed1 = seeds[3]; a.keep_out = w.keep2; }
      if ((rc = tr_attn<T>(a, dh, s))) return rc;
    }
    yin = w.y2; lng = P.ln_cross_g; lnb = P.ln_cross_b;
  }
  {  // F5: h = drop(relu(LN(y) W1^T + b)) ; x2 = LN(y)
    TfArgs a = {};
    a.M = M; a.N = d.ff; a.K = D; a.rpb = rpb;
    a.A = yin; a.lda = D; a.eps = 1e-5f; a.ln_g = lng; a.ln_b = lnb; a.ln_affine = 1; a.xout = w.x2_32; a.x16out = w.x2_16;
    a.W = (const uint16_t*)P.w_ffn0; a.ldw = D; a.bias = P.b_ffn0;
    a.out = w.h16; a.ldo = d.ff; a.act = VMC_ACT_RELU;
    if (drop) { a.p_drop1 = p; a.seed1 = seeds[4]; }
    if ((rc = tr_gemm<T, PRO_LN, EPI_ACT16>(a, s))) return rc;
  }
  {  // F6: y3 = x2 + drop(drop(h W2^T + b))
    TfArgs a = {};
    a.M = M; a.N = D; a.K = d.ff; a.rpb = rpb;
    a.A = w.h16; a.lda = d.ff;
    a.W = (const uint16_t*)P.w_ffn3; a.ldw = d.ff; a.bias = P.b_ffn3;
    a.resid = w.x2_32; a.ldres = D; a.out = w.y3; a.ldo = D;
    if (drop) { a.p_drop1 = p; a.seed1 = seeds[5]; a.p_drop2 = p; a.seed2 = seeds[6]; a.keep_out = w.keep3; }
    if ((rc = tr_ring<T>(a, s))) return rc;
  }
  return 0;
}

// ---- head ----------------------------------------------------------------------------------------------------------------
template <typename T>
int tr_head_fwd(const vmc_tfam_layer_params* layers, const vmc_tfam_head_params& Hd, float* logits, const TrDims& d, const TrWs& ws,
                float p_mlp, uint64_t seed, hipStream_t s) {
  const int D = d.D;
  const TrLayerWs wl = tr_lw(ws, d, d.L - 1);
  const vmc_tfam_layer_params& PL = layers[d.L - 1];
  if (D == 768)
    hipLaunchKernelGGL((tf_pool_kernel<T, 768>), dim3(d.B), dim3(256), 0, s, wl.y3, PL.ln_ffn_g, PL.ln_ffn_b, Hd.cls_ln_g, Hd.cls_ln_b, ws.pool16,
                       d.T, 1e-5f, ws.pooled32);
  else
    hipLaunchKernelGGL((tf_pool_kernel<T, 512>), dim3(d.B), dim3(256), 0, s, wl.y3, PL.ln_ffn_g, PL.ln_ffn_b, Hd.cls_ln_g, Hd.cls_ln_b, ws.pool16,
                       d.T, 1e-5f, ws.pooled32);
  VMC_CHECK_LAUNCH();
  int rc;
  {
    TfArgs a = {};
    a.M = d.B; a.N = D / 2; a.K = D; a.rpb = 32;
    a.A = ws.pool16; a.lda = D;
    a.W = (const uint16_t*)Hd.w_cls1; a.ldw = D; a.bias = Hd.b_cls1;
    a.out = ws.g16; a.ldo = D / 2; a.act = VMC_ACT_GELU_ERF; a.zout = ws.a16;
    if (p_mlp > 0.f) { a.p_drop1 = p_mlp; a.seed1 = seed; }
    if ((rc = tr_gemm<T, PRO_16, EPI_ACT16>(a, s))) return rc;
  }
  {
    TfArgs a = {};
    a.M = d.B; a.N = d.C; a.K = D / 2; a.rpb = 32;
    a.A = ws.g16; a.lda = D / 2;
    a.W = (const uint16_t*)Hd.w_cls4; a.ldw = D / 2; a.bias = Hd.b_cls4;
    a.out = logits; a.ldo = d.C;
    if ((rc = tr_gemm<T, PRO_16, EPI_BIAS32>(a, s))) return rc;
  }
  return 0;
}

// Head backward, 3 launches of fp32 FMAs (B <= 32 rows: 0.4 + 2.4 MFLOP of dgrad, the same of wgrad).  Work items of a launch:
// weight-gradient elements one per thread; the dgrad rows by workgroups of 64 output columns x 8 waves, the waves striding the
// contraction index (coalesced weight reads, B running sums per thread, one LDS reduction).
// H1: gW4[c, j] = sum_b dl[b, c] g[b, j];  gb4[c] = sum_b dl[b, c];  da[b, j] = (sum_c dl[b, c] W4[c, j]) drop'(b, j) gelu'(a[b, j])
template <typename T, int MAXB>
__global__ void __launch_bounds__(512) tr_head_bwd1_kernel(const float* __restrict__ dl, const float* __restrict__ W4, const uint16_t* __restrict__ g16,
                                                           const uint16_t* __restrict__ a16, float* __restrict__ gW4, float* __restrict__ gb4,
                                                           float* __restrict__ da, int B, int C, int Dh, float p, uint64_t seed_arg) {
  extern __shared__ __attribute__((aligned(16))) float hb_smem[];      // [8][64] reduction + [MAXB][C] dlogits
  float* red = hb_smem;
  float* dls = hb_smem + 512;
  const int nda = Dh / 64, tid = threadIdx.x;
  if ((int)blockIdx.x < nda) {
    for (int i = tid; i < MAXB * C; i += 512) dls[i] = i < B * C ? dl[i] : 0.f;
    __syncthreads();
    const int j = blockIdx.x * 64 + (tid & 63), wave = tid >> 6;
    float acc[MAXB];
#pragma unroll
    for (int b = 0; b < MAXB; ++b) acc[b] = 0.f;
    for (int c = wave; c < C; c += 8) {
      const float w = W4[(size_t)c * Dh + j];
#pragma unroll
      for (int b = 0; b < MAXB; ++b) acc[b] += dls[b * C + c] * w;
    }
    const uint64_t seed = p > 0.f ? resolve_seed(seed_arg) : 0;
    for (int b = 0; b < B; ++b) {
      __syncthreads();
      red[wave * 64 + (tid & 63)] = acc[b];
      __syncthreads();
      if (wave == 0) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) v += red[k * 64 + tid];
        const int e = b * Dh + j;
        if (p > 0.f) v *= dropout_factor(p, seed, (uint64_t)e);
        da[e] = v * act_grad_rt(T::to_f32(a16[e]), VMC_ACT_GELU_ERF);
      }
    }
    return;
  }
  const int i = ((int)blockIdx.x - nda) * 512 + tid;
  const int nW = C * Dh;
  if (i < nW) {
    if (gW4 == nullptr) return;
    const int c = i / Dh, j = i - c * Dh;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) acc += dl[b * C + c] * T::to_f32(g16[b * Dh + j]);
    gW4[i] = acc;
  } else if (i < nW + C) {
    if (gb4 == nullptr) return;
    const int c = i - nW;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) acc += dl[b * C + c];
    gb4[c] = acc;
  }
}
// H2: gW1[j, k] = sum_b da[b, j] pool16[b, k];  gb1[j] = sum_b da[b, j];  dpl[b, k] = sum_j da[b, j] W1[j, k]
template <typename T, int MAXB>
__global__ void __launch_bounds__(512) tr_head_bwd2_kernel(const float* __restrict__ da, const float* __restrict__ W1, const uint16_t* __restrict__ pool16,
                                                           float* __restrict__ gW1, float* __restrict__ gb1, float* __restrict__ dpl, int B, int Dh,
                                                           int D) {
  extern __shared__ __attribute__((aligned(16))) float hb_smem[];      // [8][64] reduction + [MAXB][Dh] da
  float* red = hb_smem;
  float* das = hb_smem + 512;
  const int ndp = D / 64, tid = threadIdx.x;
  if ((int)blockIdx.x < ndp) {
    for (int i = tid; i < MAXB * Dh; i += 512) das[i] = i < B * Dh ? da[i] : 0.f;
    __syncthreads();
    const int k = blockIdx.x * 64 + (tid & 63), wave = tid >> 6;
    float acc[MAXB];
#pragma unroll
    for (int b = 0; b < MAXB; ++b) acc[b] = 0.f;
#pragma unroll 4
    for (int j = wave; j < Dh; j += 8) {
      const float w = W1[(size_t)j * D + k];
#pragma unroll
      for (int b = 0; b < MAXB; ++b) acc[b] += das[b * Dh + j] * w;
    }
    for (int b = 0; b < B; ++b) {
      __syncthreads();
      red[wave * 64 + (tid & 63)] = acc[b];
      __syncthreads();
      if (wave == 0) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) v += red[q * 64 + tid];
        dpl[(size_t)b * D + k] = v;
      }
    }
    return;
  }
  const int i = ((int)blockIdx.x - ndp) * 512 + tid;
  const int nW = Dh * D;
  if (i < nW) {
    if (gW1 == nullptr) return;
    const int j = i / D, k = i - j * D;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) acc += da[b * Dh + j] * T::to_f32(pool16[b * D + k]);
    gW1[i] = acc;
  } else if (i < nW + Dh) {
    if (gb1 == nullptr) return;
    const int j = i - nW;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) acc += da[b * Dh + j];
    gb1[j] = acc;
  }
}
__device__ __forceinline__ float tr_block_sum(float v, float* red) {      // 256 threads
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
// H3: blocks 0..B-1: LayerNorm_cls backward of clip b, then the mean-pool's backward dx3[b, t, :] = dpooled[b, :] / T for ALL T rows
// (AMO_CLIP.py:170 pools padded rows too); block B: the classifier LayerNorm's gamma / beta gradients (sums over the B clips).
template <int D>
__global__ void __launch_bounds__(256) tr_head_bwd3_kernel(const float* __restrict__ pooled, const float* __restrict__ dpl, const float* __restrict__ gamma,
                                                           float* __restrict__ dx3, float* __restrict__ g_gamma, float* __restrict__ g_beta, int B,
                                                           int Tn, float eps) {
  constexpr int NI = D / 256;
  __shared__ float red[4];
  const int tid = threadIdx.x;
  float ag[NI], ab[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) ag[i] = ab[i] = 0.f;
  const bool params = (int)blockIdx.x == B;
  const int b0 = params ? 0 : blockIdx.x, b1 = params ? B : blockIdx.x + 1;
  for (int b = b0; b < b1; ++b) {
    float x[NI], g[NI];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      x[i] = pooled[(size_t)b * D + tid + 256 * i];
      s += x[i];
    }
    const float mean = tr_block_sum(s, red) * (1.0f / D);
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      x[i] -= mean;
      v += x[i] * x[i];
    }
    const float rstd = rsqrtf(tr_block_sum(v, red) * (1.0f / D) + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const float dy = dpl[(size_t)b * D + tid + 256 * i];
      x[i] *= rstd;
      ag[i] += dy * x[i];
      ab[i] += dy;
      g[i] = dy * gamma[tid + 256 * i];
      s1 += g[i];
      s2 += g[i] * x[i];
    }
    if (params) continue;
    s1 = tr_block_sum(s1, red) * (1.0f / D);
    s2 = tr_block_sum(s2, red) * (1.0f / D);
    const float invT = 1.0f / Tn;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const float dp = rstd * (g[i] - s1 - x[i] * s2) * invT;
      for (int t = 0; t < Tn; ++t) dx3[((size_t)b * Tn + t) * D + tid + 256 * i] = dp;
    }
  }
  if (params) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      if (g_gamma != nullptr) g_gamma[tid + 256 * i] = ag[i];
      if (g_beta != nullptr) g_beta[tid + 256 * i] = ab[i];
    }
  }
}

template <typename T>
int tr_head_bwd(const float* dlogits, const vmc_tfam_head_params& Hd, const TrDims& d, const TrWs& ws, float p_mlp, uint64_t seed,
                hipStream_t s) {
  const int D = d.D, Dh = D / 2, B = d.B, C = d.C;
  const TrLayerWs wl = tr_lw(ws, d, d.L - 1);
  const int n1 = Dh / 64 + (C * Dh + C + 511) / 512, n2 = D / 64 + (Dh * D + Dh + 511) / 512;
#define TR_HB(MB)                                                                                                                             \
  do {                                                                                                                                        \
    hipLaunchKernelGGL((tr_head_bwd1_kernel<T, MB>), dim3(n1), dim3(512), (512 + MB * C) * sizeof(float), s, dlogits, Hd.w32_cls4, ws.g16,     \
                       ws.a16, Hd.gw_cls4, Hd.gb_cls4, ws.da, B, C, Dh, p_mlp, seed);                                                         \
    hipLaunchKernelGGL((tr_head_bwd2_kernel<T, MB>), dim3(n2), dim3(512), (512 + MB * Dh) * sizeof(float), s, ws.da, Hd.w32_cls1, ws.pool16,   \
                       Hd.gw_cls1, Hd.gb_cls1, ws.dpl, B, Dh, D);                                                                             \
  } while (0)
  if ((size_t)32 * C * sizeof(float) > 60 * 1024) return VMC_E_SHAPE;      // the dlogits rows of H1 sit in LDS
  if (B <= 8) TR_HB(8);
  else if (B <= 16) TR_HB(16);
  else TR_HB(32);
#undef TR_HB
  VMC_CHECK_LAUNCH();
  if (D == 768)
    hipLaunchKernelGGL((tr_head_bwd3_kernel<768>), dim3(B + 1), dim3(256), 0, s, ws.pooled32, ws.dpl, Hd.cls_ln_g, wl.dx3, Hd.g_cls_ln_g, Hd.g_cls_ln_b, B, d.T,
                       1e-5f);
  else
    hipLaunchKernelGGL((tr_head_bwd3_kernel<512>), dim3(B + 1), dim3(256), 0, s, ws.pooled32, ws.dpl, Hd.cls_ln_g, wl.dx3, Hd.g_cls_ln_g, Hd.g_cls_ln_b, B, d.T,
                       1e-5f);
  VMC_CHECK_LAUNCH();
  return 0;
}

// ---- grouped weight gradients of a layer -----------------------------------------------------------------------------------
struct TrTnProb {
  const uint16_t* dY;       // [M, N] 16-bit, row stride lddy
  const uint16_t* X;        // [M, K] 16-bit, row stride ldx
  float* C;                 // [N, K] fp32 (row stride K)
  float* dbias;             // [N] or null
  int M, N, K, lddy, ldx, tiles_k, tile0;
};
struct TrLnProb {
  const float* dx;          // gradient wrt the LayerNorm output [M, D]
  const float* y;           // LayerNorm input [M, D]
  const float* stats;       // [M, 2] mean, rstd
  float* g_gamma;
  float* g_beta;
};
constexpr int TR_MAX_PROB = 32, TR_MAX_LN = 12;      // up to four layers' problems in one launch (deferred weight gradients)
struct TrWgradGroup {
  TrTnProb p[TR_MAX_PROB];
  TrLnProb ln[TR_MAX_LN];
  int nprob, total_tiles, nln, M, D;
};

template <typename T>
__global__ void __launch_bounds__(512) tr_wgrad_group_kernel(const TrWgradGroup g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // the LayerNorm parameter-gradient blocks come FIRST: they are short, and the grid (240 + 36 workgroups at one per CU) does not
  // fit one resident round -- queued last they would start a second round behind the 256 x 128 tiles
  const int n_ln = g.nln * (g.D / 64);
  const int bid = (int)blockIdx.x - n_ln;
  if (bid >= 0) {
    int i = 0;
    for (int k = 1; k < g.nprob; ++k)
      if (bid >= g.p[k].tile0) i = k;
    const TrTnProb pr = g.p[i];
    const int local = bid - pr.tile0;
    tn_tile_body<T>(pr.dY, pr.X, pr.C, pr.dbias, pr.M, pr.N, pr.K, pr.lddy, pr.ldx, local / pr.tiles_k, local % pr.tiles_k, 0, (pr.M + 63) / 64, 0,
                    smem);
    return;
  }
  // LayerNorm parameter gradients: 64 columns per workgroup, the 8 waves stride the token rows
  const int idx = blockIdx.x, cbs = g.D / 64;
  const TrLnProb lp = g.ln[idx / cbs];
  const int col = (idx % cbs) * 64 + (threadIdx.x & 63), wave = threadIdx.x >> 6;
  float ag = 0.f, ab = 0.f;
  for (int r = wave; r < g.M; r += 8) {
    const float2 st = *(const float2*)(lp.stats + 2 * (size_t)r);
    const float dxv = lp.dx[(size_t)r * g.D + col];
    ag += dxv * (lp.y[(size_t)r * g.D + col] - st.x) * st.y;
    ab += dxv;
  }
  float* red = (float*)smem;
  red[threadIdx.x] = ag;
  red[512 + threadIdx.x] = ab;
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int k = 1; k < 8; ++k) {
      ag += red[64 * k + threadIdx.x];
      ab += red[512 + 64 * k + threadIdx.x];
    }
    if (lp.g_gamma != nullptr) lp.g_gamma[col] = ag;
    if (lp.g_beta != nullptr) lp.g_beta[col] = ab;
  }
}

inline void tr_add_prob(TrWgradGroup& g, const uint16_t* dY, const uint16_t* X, float* C, float* dbias, int M, int N, int K, int lddy, int ldx) {
  if (C == nullptr && dbias == nullptr) return;
  if (g.nprob >= TR_MAX_PROB) { g.nprob = TR_MAX_PROB + 1; return; }      // overflow: reported by tr_wgrad_launch
  TrTnProb& p = g.p[g.nprob++];
  p.dY = dY; p.X = X; p.C = C; p.dbias = dbias; p.M = M; p.N = N; p.K = K; p.lddy = lddy; p.ldx = ldx;
  p.tiles_k = (K + 127) / 128;
  p.tile0 = g.total_tiles;
  g.total_tiles += ((N + 255) / 256) * p.tiles_k;
}
inline void tr_add_ln(TrWgradGroup& g, const float* dx, const float* y, const float* stats, float* gg, float* gb) {
  if (gg == nullptr && gb == nullptr) return;
  if (g.nln >= TR_MAX_LN) { g.nln = TR_MAX_LN + 1; return; }
  TrLnProb& l = g.ln[g.nln++];
  l.dx = dx; l.y = y; l.stats = stats; l.g_gamma = gg; l.g_beta = gb;
}
template <typename T>
int tr_wgrad_launch(const TrWgradGroup& g, hipStream_t s) {
  if (g.nprob > TR_MAX_PROB || g.nln > TR_MAX_LN) return VMC_E_SHAPE;
  const int blocks = g.total_tiles + g.nln * (g.D / 64);
  if (blocks == 0) return 0;
  for (int i = 0; i < g.nprob; ++i)
    if (g.p[i].C == nullptr) return VMC_E_ARG;      // a bias gradient without its weight gradient is not a case of this chain
  const size_t lds = (size_t)TN_STAGES * TN_STAGE;
  auto kern = tr_wgrad_group_kernel<T>;
  static bool attr_done = false;
  if (int rc = tf_set_lds(kern, attr_done)) return rc;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, s, g);
  VMC_CHECK_LAUNCH();
  return 0;
}

// ---- backward of one layer -------------------------------------------------------------------------------------------------
// defer != null: the layer's weight-gradient problems are appended to *defer instead of being launched (vmc_tfam_train_bwd launches
// all layers' problems once, after the last dgrad chain: one HBM-write-bound launch instead of four with a tail each)
template <typename T>
int tr_layer_bwd(const uint8_t* mask, const uint8_t* mask_kv, const vmc_tfam_layer_params* layers, int layer, const TrDims& d, const TrWs& ws,
                 float p, const uint64_t* seeds, int dtype16, hipStream_t s, TrWgradGroup* defer = nullptr) {
  const int M = d.B * d.T, Mk = d.B * d.Tk, D = d.D, dh = D / d.H;
  const int rpb = d.T > 32 ? 32 : (d.T <= 16 ? 2 : 1) * d.T;      // the backward's GEMMs are all row-wise: uniform blocks
  const vmc_tfam_layer_params& P = layers[layer];
  const TrLayerWs w = tr_lw(ws, d, layer);
  const bool drop = p > 0.f;
  const float inv = drop ? 1.0f / (1.0f - p) : 1.0f;
  const size_t aws = (size_t)d.B * d.H * d.T * sizeof(float);
  int rc;
  {  // L1: dh = [LNbwd_ffn(dx3) . keep3] W2 . gate(h)
    TfArgs a = {};
    a.M = M; a.N = d.ff; a.K = D; a.rpb = rpb;
    a.A = w.y3; a.lda = D; a.eps = 1e-5f; a.ln_g = P.ln_ffn_g; a.dxin = w.dx3;
    a.keep_in = drop ? w.keep3 : nullptr; a.keep_scale = inv * inv;
    a.dyout = w.dy3; a.d16out = w.d3_16; a.stats = w.st3;
    a.W = (const uint16_t*)P.wt_ffn3; a.ldw = D;
    a.out = w.dh16; a.ldo = d.ff; a.act = VMC_ACT_NONE; a.gate = w.h16; a.gate_scale = inv;
    if ((rc = tr_gemm<T, PRO_LNBWD, EPI_ACT16>(a, s))) return rc;
  }
  {  // L2: dx2 = dy3 + dh W1
    TfArgs a = {};
    a.M = M; a.N = D; a.K = d.ff; a.rpb = rpb;
    a.A = w.dh16; a.lda = d.ff;
    a.W = (const uint16_t*)P.wt_ffn0; a.ldw = d.ff;
    a.resid = w.dy3; a.ldres = D; a.out = w.dx2; a.ldo = D;
    if ((rc = tr_ring<T>(a, s))) return rc;
  }
  const float* dx_self = w.dx2;
  if (d.has_cross) {
    {  // L3: dOc = [LNbwd_cross(dx2) . keep2] Wo_c
      TfArgs a = {};
      a.M = M; a.N = D; a.K = D; a.rpb = rpb;
      a.A = w.y2; a.lda = D; a.eps = 1e-5f; a.ln_g = P.ln_cross_g; a.dxin = w.dx2;
      a.keep_in = drop ? w.keep2 : nullptr; a.keep_scale = inv;
      a.dyout = w.dy2; a.d16out = w.d2_16; a.stats = w.st2;
      a.W = (const uint16_t*)P.wt_cross_out; a.ldw = D;
      a.out = w.doc16; a.ldo = D; a.act = VMC_ACT_NONE;
      if ((rc = tr_gemm<T, PRO_LNBWD, EPI_ACT16>(a, s))) return rc;
    }
    // L4: cross-attention backward
    if ((rc = vmc_attention_bwd(w.q16, w.kv16, w.kv16 + D, mask_kv, w.o_cross, w.doc16, w.lse_cross, w.dq16, w.dkv16, w.dkv16 + D, d.B, d.H, d.T,
                                d.Tk, dh, D, 2 * D, 2 * D, D, D, 2 * D, 2 * D, drop ? p : 0.f, drop ? seeds[2] : 0, w.attn_ws, aws, dtype16, s)))
      return rc;
    {  // L5: dx1 = dy2 + dq Wq
      TfArgs a = {};
      a.M = M; a.N = D; a.K = D; a.rpb = rpb;
      a.A = w.dq16; a.lda = D;
      a.W = (const uint16_t*)P.wt_cross_in; a.ldw = 3 * D;      // columns 0:D of the transposed packed in_proj = Wq^T
      a.resid = w.dy2; a.ldres = D; a.out = w.dx1; a.ldo = D;
      if ((rc = tr_gemm<T, PRO_16, EPI_RESID32>(a, s))) return rc;
    }
    dx_self = w.dx1;
  }
  {  // L6: dOs = [LNbwd_self(dx) . keep1] Wo_s
    TfArgs a = {};
    a.M = M; a.N = D; a.K = D; a.rpb = rpb;
    a.A = w.y1; a.lda = D; a.eps = 1e-5f; a.ln_g = P.ln_self_g; a.dxin = dx_self;
    a.keep_in = drop ? w.keep1 : nullptr; a.keep_scale = inv;
    a.dyout = w.dy1; a.d16out = w.d1_16; a.stats = w.st1;
    a.W = (const uint16_t*)P.wt_self_out; a.ldw = D;
    a.out = w.dos16; a.ldo = D; a.act = VMC_ACT_NONE;
    if ((rc = tr_gemm<T, PRO_LNBWD, EPI_ACT16>(a, s))) return rc;
  }
  // L7: self-attention backward
  if ((rc = vmc_attention_bwd(w.qkv16, w.qkv16 + D, w.qkv16 + 2 * D, mask, w.o_self, w.dos16, w.lse_self, w.dqkv16, w.dqkv16 + D, w.dqkv16 + 2 * D,
                              d.B, d.H, d.T, d.T, dh, 3 * D, 3 * D, 3 * D, D, 3 * D, 3 * D, 3 * D, drop ? p : 0.f, drop ? seeds[0] : 0, w.attn_ws, aws,
                              dtype16, s)))
    return rc;
  if (layer > 0) {  // L8: dx3 of the layer below = dy1 + dqkv Wqkv
    const TrLayerWs wp = tr_lw(ws, d, layer - 1);
    TfArgs a = {};
    a.M = M; a.N = D; a.K = 3 * D; a.rpb = rpb;
    a.A = w.dqkv16; a.lda = 3 * D;
    a.W = (const uint16_t*)P.wt_self_in; a.ldw = 3 * D;
    a.resid = w.dy1; a.ldres = D; a.out = wp.dx3; a.ldo = D;
    if ((rc = tr_ring<T>(a, s))) return rc;
  }
  // weight, bias and LayerNorm-parameter gradients: one grouped launch
  TrWgradGroup local = {};
  TrWgradGroup& g = defer ? *defer : local;
  g.M = M; g.D = D;
  tr_add_prob(g, w.dqkv16, w.x0_16, P.gw_self_in, P.gb_self_in, M, 3 * D, D, 3 * D, D);
  tr_add_prob(g, w.d1_16, w.o_self, P.gw_self_out, P.gb_self_out, M, D, D, D, D);
  if (d.has_cross) {
    tr_add_prob(g, w.dq16, w.x1_16, P.gw_cross_in, P.gb_cross_in, M, D, D, D, D);
    tr_add_prob(g, w.dkv16, ws.motion16, P.gw_cross_in ? P.gw_cross_in + (size_t)D * D : nullptr, P.gb_cross_in ? P.gb_cross_in + D : nullptr, Mk,
                2 * D, D, 2 * D, D);
    tr_add_prob(g, w.d2_16, w.o_cross, P.gw_cross_out, P.gb_cross_out, M, D, D, D, D);
  }
  tr_add_prob(g, w.dh16, w.x2_16, P.gw_ffn0, P.gb_ffn0, M, d.ff, D, d.ff, D);
  tr_add_prob(g, w.d3_16, w.h16, P.gw_ffn3, P.gb_ffn3, M, D, d.ff, D, d.ff);
  tr_add_ln(g, w.dx3, w.y3, w.st3, P.g_ln_ffn_g, P.g_ln_ffn_b);
  if (d.has_cross) tr_add_ln(g, w.dx2, w.y2, w.st2, P.g_ln_cross_g, P.g_ln_cross_b);
  tr_add_ln(g, dx_self, w.y1, w.st1, P.g_ln_self_g, P.g_ln_self_b);
  return defer ? 0 : tr_wgrad_launch<T>(g, s);
}

inline bool tr_ws_ok(const void* ws, size_t bytes, const TrWs& w) { return ws != nullptr && bytes >= w.bytes && (((uintptr_t)ws) & 255) == 0; }

}  // namespace

extern "C" size_t vmc_tfam_train_workspace_bytes(int B, int T, int Tk, int D, int H, int ff, int L, int C, int has_cross) {
  TrDims d = {B, T, Tk, D, H, ff, L, C, has_cross};
  if (tr_check(d)) return 0;
  return tr_ws(nullptr, d).bytes;
}

#define TR_PROLOG()                                                         \
  TrDims d = {B, T, Tk, D, H, ff, L, C, has_cross};                         \
  if (int rc_ = tr_check(d)) return rc_;                                    \
  if (dtype16 != VMC_BF16 && dtype16 != VMC_F16) return VMC_E_DTYPE;        \
  const TrWs ws = tr_ws(workspace, d);                                      \
  if (!tr_ws_ok(workspace, workspace_bytes, ws)) return VMC_E_ARG;          \
  hipStream_t s = (hipStream_t)stream

extern "C" int vmc_tfam_layer_train_fwd(const float* x_in, const float* motion, const uint8_t* mask, const uint8_t* mask_kv,
                                        const vmc_tfam_layer_params* layers, int layer, void* workspace, size_t workspace_bytes, int B, int T, int Tk,
                                        int D, int H, int ff, int L, int C, int has_cross, float p_drop, const uint64_t* seeds, int dtype16,
                                        void* stream) {
  TR_PROLOG();
  if (!layers || layer < 0 || layer >= L || (layer == 0) != (x_in != nullptr) || (has_cross && !motion) || (p_drop > 0.f && !seeds)) return VMC_E_ARG;
  if (p_drop < 0.f || p_drop >= 1.f) return VMC_E_ARG;
  return dtype16 == VMC_BF16 ? tr_layer_fwd<BF16>(x_in, motion, mask, mask_kv, layers, layer, d, ws, p_drop, seeds, s)
                             : tr_layer_fwd<F16>(x_in, motion, mask, mask_kv, layers, layer, d, ws, p_drop, seeds, s);
}

extern "C" int vmc_tfam_head_train_fwd(const vmc_tfam_layer_params* layers, const vmc_tfam_head_params* head, float* logits, void* workspace,
                                       size_t workspace_bytes, int B, int T, int Tk, int D, int H, int ff, int L, int C, int has_cross, float p_mlp,
                                       uint64_t seed, int dtype16, void* stream) {
  TR_PROLOG();
  if (!layers || !head || !logits || p_mlp < 0.f || p_mlp >= 1.f) return VMC_E_ARG;
  return dtype16 == VMC_BF16 ? tr_head_fwd<BF16>(layers, *head, logits, d, ws, p_mlp, seed, s) : tr_head_fwd<F16>(layers, *head, logits, d, ws, p_mlp, seed, s);
}

extern "C" int vmc_tfam_head_bwd(const float* dlogits, const vmc_tfam_layer_params* layers, const vmc_tfam_head_params* head, void* workspace,
                                 size_t workspace_bytes, int B, int T, int Tk, int D, int H, int ff, int L, int C, int has_cross, float p_mlp,
                                 uint64_t seed, int dtype16, void* stream) {
  TR_PROLOG();
  if (!dlogits || !layers || !head) return VMC_E_ARG;
  return dtype16 == VMC_BF16 ? tr_head_bwd<BF16>(dlogits, *head, d, ws, p_mlp, seed, s) : tr_head_bwd<F16>(dlogits, *head, d, ws, p_mlp, seed, s);
}

extern "C" int vmc_tfam_layer_bwd(const uint8_t* mask, const uint8_t* mask_kv, const vmc_tfam_layer_params* layers, int layer, void* workspace,
                                  size_t workspace_bytes, int B, int T, int Tk, int D, int H, int ff, int L, int C, int has_cross, float p_drop,
                                  const uint64_t* seeds, int dtype16, void* stream) {
  TR_PROLOG();
  if (!layers || layer < 0 || layer >= L || (p_drop > 0.f && !seeds)) return VMC_E_ARG;
  return dtype16 == VMC_BF16 ? tr_layer_bwd<BF16>(mask, mask_kv, layers, layer, d, ws, p_drop, seeds, dtype16, s)
                             : tr_layer_bwd<F16>(mask, mask_kv, layers, layer, d, ws, p_drop, seeds, dtype16, s);
}

extern "C" int vmc_tfam_train_fwd(const float* x, const float* motion, const uint8_t* mask, const uint8_t* mask_kv, const vmc_tfam_layer_params* layers,
                                  const vmc_tfam_head_params* head, float* logits, void* workspace, size_t workspace_bytes, int B, int T, int Tk, int D,
                                  int H, int ff, int L, int C, int has_cross, float p_drop, float p_mlp, const uint64_t* seeds, int dtype16,
                                  void* stream) {
  if (!x || !layers || !head || !logits) return VMC_E_ARG;
  for (int l = 0; l < L; ++l) {
    const int rc = vmc_tfam_layer_train_fwd(l == 0 ? x : nullptr, motion, mask, mask_kv, layers, l, workspace, workspace_bytes, B, T, Tk, D, H, ff, L, C,
                                            has_cross, p_drop, seeds ? seeds + 7 * l : nullptr, dtype16, stream);
    if (rc) return rc;
  }
  if (p_mlp > 0.f && !seeds) return VMC_E_ARG;
  return vmc_tfam_head_train_fwd(layers, head, logits, workspace, workspace_bytes, B, T, Tk, D, H, ff, L, C, has_cross, p_mlp, seeds ? seeds[7 * L] : 0,
                                 dtype16, stream);
}

extern "C" int vmc_tfam_train_bwd(const float* dlogits, const uint8_t* mask, const uint8_t* mask_kv, const vmc_tfam_layer_params* layers,
                                  const vmc_tfam_head_params* head, void* workspace, size_t workspace_bytes, int B, int T, int Tk, int D, int H, int ff,
                                  int L, int C, int has_cross, float p_drop, float p_mlp, const uint64_t* seeds, int dtype16, void* stream) {
  if (p_mlp > 0.f && !seeds) return VMC_E_ARG;
  int rc = vmc_tfam_head_bwd(dlogits, layers, head, workspace, workspace_bytes, B, T, Tk, D, H, ff, L, C, has_cross, p_mlp, seeds ? seeds[7 * L] : 0, dtype16,
                             stream);
  if (rc) return rc;
  TR_PROLOG();
  if (!layers || (p_drop > 0.f && !seeds)) return VMC_E_ARG;
  // the dgrad chains of all layers first, then every weight gradient in ONE launch (the per-layer entry point launches its own).
  // Four layers' problems fit the table; deeper models flush it every four layers.
  TrWgradGroup g = {};
  int pending = 0;
  for (int l = L - 1; l >= 0; --l) {
    rc = dtype16 == VMC_BF16 ? tr_layer_bwd<BF16>(mask, mask_kv, layers, l, d, ws, p_drop, seeds ? seeds + 7 * l : nullptr, dtype16, s, &g)
                             : tr_layer_bwd<F16>(mask, mask_kv, layers, l, d, ws, p_drop, seeds ? seeds + 7 * l : nullptr, dtype16, s, &g);
    if (rc) return rc;
    if (++pending == 4 || l == 0) {
      rc = dtype16 == VMC_BF16 ? tr_wgrad_launch<BF16>(g, s) : tr_wgrad_launch<F16>(g, s);
      if (rc) return rc;
      g = TrWgradGroup{};
      pending = 0;
    }
  }
  return 0;
}
