/*
 * vmc.h — C ABI of libvmc.so, the MI355X (gfx950) kernels under the ViMoCLIP hot path.
 *
 * The reference (MarcosRodrigoT/VIMO-CLIP) has no FFI: its hot path sits behind Python nn.Module /
 * function signatures and runs on stock PyTorch ATen ops.  Each entry point below replaces the ATen
 * op site(s) named in its comment (paths relative to the reference checkout; SURVEY.md §2b ids K0..K16).
 * The Python mirror of the reference interface (vimo_clip_amd/) calls these through ctypes.
 *
 * Conventions
 *   - plain pointers + sizes only; every buffer (inputs, outputs, workspace) is owned by the caller
 *     (PyTorch's allocator); kernels never allocate, free or retain pointers;
 *   - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*); no hidden streams,
 *     no device synchronisation, safe under hipGraph capture;
 *   - return 0 on success, a positive hipError_t on a HIP failure, a negative VMC_E_* on a bad argument;
 *     never throws, never aborts;
 *   - stateless and re-entrant; one process per GPU.
 *   - "16-bit" tensors are bf16 or f16, selected per call by `dtype16` (VMC_BF16 / VMC_F16); all
 *     reductions, softmax, LayerNorm statistics and GEMM accumulation are fp32.
 */
#ifndef VMC_H
#define VMC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { VMC_F32 = 0, VMC_BF16 = 1, VMC_F16 = 2 };
enum { VMC_ACT_NONE = 0, VMC_ACT_QUICKGELU = 1, VMC_ACT_GELU_ERF = 2, VMC_ACT_RELU = 3 };
enum {
  VMC_E_ARG = -1,       /* null pointer / non-positive size */
  VMC_E_ALIGN = -2,     /* a stride or pointer violates the documented alignment */
  VMC_E_SHAPE = -3,     /* unsupported shape (e.g. K % 64 != 0, head_dim not supported) */
  VMC_E_DTYPE = -4
};

/* Library identity: returns the ABI version (1). */
int vmc_abi_version(void);
/* Human-readable message for a return code (static storage). */
const char* vmc_error_string(int code);
/* Diagnostics: `workgroups` workgroups of 256 threads run iters x 64 bf16 MFMAs and write {shader cycles (s_memtime), 100 MHz
 * ticks (s_memrealtime)} as two u64 each into out[2 * workgroups]: cycles / ticks x 100 MHz = the clock the chip sustains under an
 * MFMA load on THIS box (bench.py prints it beside the rooflines so that box-to-box swings of clock-sensitive kernels are visible). */
int vmc_clock_probe(void* out, int workgroups, int iters, int mfma_shape /* 0: 16x16x32, 1: 32x32x16 (same FLOPs per round) */, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * K0 — frame preprocess + patch extraction.
 * Replaces the per-frame CPU loop  to_pil_image -> Resize/CenterCrop/ToTensor/Normalize
 * (models/student_model.py:77-78) and PIL + CLIPImageProcessor (extract_embeddings.py:89-93) for frames
 * that are already R x R (resize and crop are identities), and the im2col of visual.conv1
 * (models/student_model.py:84).
 *   frames  u8 [F,3,R,R] (NCHW, as FlowStudentModel.forward receives them)
 *   patches 16-bit [F*g*g, kpad], g = R/p, column k = c*p*p + dy*p + dx, columns >= 3*p*p zero
 *   value   ((wrap ? (256 - v) & 255 : v) / 255 - mean[c]) / std[c]   (wrap: SURVEY.md §7 quirk 1)
 */
int vmc_preprocess_patches_u8(const uint8_t* frames, void* patches, int F, int R, int p, int kpad,
                              int wrap_quirk, int dtype16, void* stream);

/* Same patch extraction for frames that are already normalised floats: `pixel_values` f32 [F,3,R,R], the
 * argument of CLIPModel.get_image_features (extract_embeddings.py:91-94). */
int vmc_patches_f32(const float* pixel_values, void* patches, int F, int R, int p, int kpad, int dtype16,
                    void* stream);

/* Split-precision patch operands for the INFERENCE patch-embedding GEMM (0.2 % of the encoder's FLOPs, but its 16-bit operand
 * rounding was the largest single term of the embedding error).  Same op sites as the two entries above.
 *   vmc_patches_u8_exact:  patches 16-bit [F*g*g, 2*kpad] = [v | v], v = the raw pixel 0..255 (exact in 16 bits; wrap_quirk as
 *     above).  Pair with W = [W'_hi | W'_lo], W'[n,k] = conv1[n,k] / std_c, bias'[n] = -255 sum_k conv1[n,k] mean_c / std_c and
 *     alpha = 1/255 in vmc_linear:  alpha (A W^T + bias') = sum_k ((v/255 - mean_c) / std_c) conv1[n,k]  to fp32 accuracy.
 *   vmc_patches_f32_split: patches [F*g*g, 3*kpad] = [x_hi | x_lo | x_hi], x_lo = x - x_hi; pair with W = [W_hi | W_hi | W_lo].
 * Columns >= 3 p^2 of every kpad-wide part are zero. */
int vmc_patches_u8_exact(const uint8_t* frames, void* patches, int F, int R, int p, int kpad, int wrap_quirk, int dtype16,
                         void* stream);
int vmc_patches_f32_split(const float* pixel_values, void* patches, int F, int R, int p, int kpad, int dtype16, void* stream);

/* Pillow-exact antialiased resample of planar u8 images [planes, in_h, in_w] along one axis — the BICUBIC
 * ``Resize`` of clip._transform (models/student_model.py:77-78) and of CLIPImageProcessor (extract_embeddings.py:91),
 * both of which call PIL.Image.resize.  out pixel o (of the resampled axis) = clip8((2^21 + sum_t in[lo_o + t] *
 * coeffs[o*ksize + t]) >> 22) with (lo_o, n_o) = bounds[2o], bounds[2o+1]; tables are Pillow's precompute_coeffs in
 * 22-bit fixed point, computed by the host in float64.  Only outputs [out_first, out_first + out_count) are written
 * (the centre crop is fused): horizontal -> out [planes, in_h, out_count], vertical -> out [planes, out_count, in_w].
 * wrap_quirk applies v -> (256 - v) & 255 to the INPUT pixels (the student's float -> to_pil_image wrap). */
int vmc_resample_u8(const uint8_t* in, uint8_t* out, const int* bounds, const int* coeffs, int planes, int in_h, int in_w,
                    int out_first, int out_count, int ksize, int horizontal, int wrap_quirk, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * K1/K3/K5/K6/K7/K9/K11-K14 — Linear layers:  C = epilogue(A @ W^T)   (MFMA, fp32 accumulate).
 * Replaces nn.Linear / F.linear / conv1-as-GEMM / `x @ proj` op sites: visual.conv1, attn.in_proj,
 * attn.out_proj, mlp.c_fc, mlp.c_proj, proj (OpenAI clip VisionTransformer, called at
 * models/student_model.py:84), ResidualMLP.fc1/fc2 (models/student_model.py:33), classification_head
 * (:55-59), nn.MultiheadAttention in/out projections, ffn.0/ffn.3, classifier.1/.4, projection_layer
 * (TFAM/models/AMO_CLIP.py:19-29,84,86).
 *   A [M,K] 16-bit row-major (row stride lda elements), W [N,K] 16-bit row-major (ldw)
 *   v = act(acc + bias[n]);  C[orow, n] = alpha * v + res[rrow, n]
 *   orow = out_row_group ? m + m / out_row_group + 1 : m      (patch rows -> token rows, class row skipped)
 *   rrow = res_row_mod  ? m % res_row_mod : orow              (positional-embedding broadcast)
 *   bias (f32 [N]) and res may be NULL; C and res are f32 or 16-bit (out_dtype / res_dtype).
 * Requirements: K % 64 == 0, N % 4 == 0; lda, ldw % 8 == 0; ldc, ldres % 4 == 0; 16-byte aligned bases.
 */
int vmc_linear(const void* A, const void* W, const float* bias, const void* res, void* C,
               int M, int N, int K, int lda, int ldw, int ldc, int ldres,
               int act, float alpha, int out_dtype, int res_dtype, int out_row_group, int res_row_mod,
               int dtype16, void* stream);
/* vmc_linear with a side output for training: Z[m, n] (16-bit, row stride ldz >= N, ldz % 4 == 0, 16-byte aligned) receives
 * A W^T + bias BEFORE the activation, from the same epilogue that writes C = alpha * act(.) (+ res) -- the tensor the backward
 * of an activation needs (autograd of the QuickGELU / GELU linears, models/student_model.py:27-34, clip model.py MLP), without
 * a second pass over the 4D-wide matrix.  Z == NULL: identical to vmc_linear.  Not combined with out_row_group. */
int vmc_linear_preact(const void* A, const void* W, const float* bias, const void* res, void* C, void* Z,
                      int M, int N, int K, int lda, int ldw, int ldc, int ldres, int ldz,
                      int act, float alpha, int out_dtype, int res_dtype, int out_row_group, int res_row_mod,
                      int dtype16, void* stream);

/* Weight-gradient form of vmc_linear (autograd of F.linear w.r.t. the weight, train.py:104): C [M,N] f32 contiguous =
 * A [M,K] @ W [N,K]^T with a long contraction K (the token count) and a small output.  128x128 tiles x K slices; every
 * slice writes a partial slab into the workspace, a second kernel sums the slabs in a fixed order (deterministic).
 * K % 64 == 0, N % 4 == 0, lda/ldw % 8 == 0; workspace >= vmc_linear_splitk_workspace_bytes (may be 0 -> NULL). */
size_t vmc_linear_splitk_workspace_bytes(int M, int N, int K);
int vmc_linear_splitk_f32(const void* A, const void* W, float* C, int M, int N, int K, int lda, int ldw, void* workspace,
                          size_t workspace_bytes, int dtype16, void* stream);

/* Weight gradient straight from the token-major operands the forward left behind (no transposed copies):
 * C [N,K] f32 contiguous = dY[M,N]^T @ X[M,K], contraction over the M tokens (any M; rows past M read as zeros).
 * Both MFMA operands come from [64 tokens][128 columns] LDS tiles through ds_read_b64_tr_b16.  Token range split in
 * slices -> slabs in the workspace + deterministic reduce.  N % 8 == 0, K % 8 == 0, lddy/ldx % 8 == 0. */
size_t vmc_linear_wgrad_tn_workspace_bytes(int M, int N, int K);
int vmc_linear_wgrad_tn(const void* dY, const void* X, float* C, int M, int N, int K, int lddy, int ldx, void* workspace,
                        size_t workspace_bytes, int dtype16, void* stream);
/* Same launch, plus the bias gradient dbias[N] = column sums of dY (f32, 16-byte aligned; NULL = vmc_linear_wgrad_tn):
 * the dY fragments already in registers go through four more MFMAs against a ones operand, replacing a separate
 * column-sum pass over dY (autograd of F.linear's bias, train.py:104). */
int vmc_linear_wgrad_bias_tn(const void* dY, const void* X, float* C, float* dbias, int M, int N, int K, int lddy, int ldx,
                             void* workspace, size_t workspace_bytes, int dtype16, void* stream);

/* Many TN weight gradients in ONE launch (the weight gradients of a training step: TFAM/train_and_eval.py:94-95, train.py:103-104 --
 * `loss.backward()` leaves one small dW = dY^T X per linear).  Every 256 x 256 tile runs over all M tokens of its problem: no token
 * slices, no workspace, no reduce launch.  Per problem: M % 128 == 0 and M >= 256, N % 8 == 0, K % 8 == 0, lddy / ldx % 8 == 0,
 * operands < 2 GiB, pointers 16-byte aligned; C [N, K] f32 contiguous, dbias [N] f32 or NULL (column sums of dY).  n <= VMC_WGRAD_GROUP_MAX. */
#define VMC_WGRAD_GROUP_MAX 32
typedef struct vmc_wgrad_tn_problem {
  const void* dY;      /* [M, lddy] 16-bit */
  const void* X;       /* [M, ldx] 16-bit */
  float* C;            /* [N, K] */
  float* dbias;        /* [N] or NULL */
  int M, N, K, lddy, ldx, reserved;
} vmc_wgrad_tn_problem;
int vmc_linear_wgrad_tn_group(const vmc_wgrad_tn_problem* probs, int n, int dtype16, void* stream);

/* vmc_linear with the large-problem kernel chosen PER CALL (A/B measurements in one process; the library keeps no
 * state): VMC_GEMM_TWOSTAGE = two-stage tiles only, VMC_GEMM_DEFAULT = what vmc_linear does (8-phase 256x256 kernel; the
 * tile rows of a small last partial round handed to the small-tile kernel in a second launch; whole-tile problems with a
 * bias and a 16-bit output walk their tiles in a persistent workgroup per CU that prefetches the next tile's operands),
 * VMC_GEMM_NO_TAIL_SPLIT = the same without that split, VMC_GEMM_PERSISTENT = the persistent walk for every eligible
 * epilogue, VMC_GEMM_ONE_TILE = never persistent (one tile per workgroup).  Results are identical bit for bit across them. */
enum { VMC_GEMM_TWOSTAGE = 0, VMC_GEMM_DEFAULT = 1, VMC_GEMM_NO_TAIL_SPLIT = 2, VMC_GEMM_PERSISTENT = 3, VMC_GEMM_ONE_TILE = 4,
       VMC_GEMM_MFMA32 = 5, /* the persistent walk on v_mfma_f32_32x32x16 fragments (bias / QuickGELU / bias-free epilogues); the
                               k-steps are summed 16 at a time, so the last bit may differ from the other variants */
       VMC_GEMM_VARIANTS = 6 };
int vmc_linear_variant(const void* A, const void* W, const float* bias, const void* res, void* C,
                       int M, int N, int K, int lda, int ldw, int ldc, int ldres,
                       int act, float alpha, int out_dtype, int res_dtype, int out_row_group, int res_row_mod,
                       int dtype16, int variant, void* stream);

/* 16-bit 2-D transpose  out[c, r] = in[r, c]  (rows x cols, element strides ld_in / ld_out); used for
 * the dgrad/wgrad operand layouts of K8 (autograd of F.linear, train.py:104). */
int vmc_transpose16(const void* in, void* out, int rows, int cols, int ld_in, int ld_out, void* stream);

/* f32 -> 16-bit cast of a weight matrix [rows, cols] (+ optional transposed copy [cols, rows]); the
 * 16-bit compute copies of the fp32 master parameters (models/student_model.py:45 keeps fp32). */
int vmc_cast_weight(const float* w, void* w16, void* w16_t, int rows, int cols, int ld_out, int ld_out_t,
                    int dtype16, void* stream);

/* The same for many weights in ONE launch: after an optimiser step (`optimizer.step()`, train.py:107, TFAM/train_and_eval.py:96)
 * every 16-bit compute copy of the trained parameters is refreshed in place.  `desc` is a DEVICE array of n_desc records of
 * 48 bytes {const float* w; void* w16; void* w16_t; int rows, cols, ld_out, ld_out_t, tile0, tiles_x;} (NULL copies are skipped);
 * record i owns the 64x64 tiles [tile0, tile0 of record i+1) of a 1-D grid of total_tiles, tiles_x = ceil(cols / 64). */
int vmc_cast_weights_multi(const void* desc, int n_desc, int total_tiles, int dtype16, void* stream);

/* Column sums  out[n] = sum_m in[m, n]  (bias gradients, K8).  N % 4 == 0, ld_in % 4 == 0.  workspace: >= vmc_colsum_workspace_bytes. */
size_t vmc_colsum_workspace_bytes(int M, int N);
int vmc_colsum(const void* in, float* out, int M, int N, int ld_in, int in_dtype, void* workspace,
               size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * K2 — LayerNorm (eps inside sqrt, fp32 statistics).  Replaces ln_pre/ln_1/ln_2/ln_post of the CLIP
 * ViT and norm_self/norm_cross/norm_ffn/classifier.0 (TFAM/models/AMO_CLIP.py:32-34,84).
 *   x [rows, D] (row stride ldx; f32 or 16-bit), gamma/beta f32 [D]
 *   y16 (16-bit, ld = D) and/or y32 (f32, ld = D): either may be NULL
 *   mean/rstd f32 [rows]: optional saves for the backward
 * Requirements: D % 4 == 0, D <= 4096.
 */
int vmc_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y16, float* y32,
                      float* mean, float* rstd, int rows, int D, int ldx, float eps, int x_dtype,
                      int dtype16, void* stream);
/* Fused residual update + LayerNorm of the pre-LN ViT blocks: x (fp32 residual stream, row stride ldx) <- x +
 * branch (16-bit attention / MLP branch output, row stride ldb), written back when write_x != 0, and
 * y16 [rows, D] = LN(x).  Replaces `x = x + attn(...)` / `x = x + mlp(...)` followed by ln_2 / the next ln_1 /
 * ln_post (OpenAI clip ResidualAttentionBlock.forward).  D % 256 == 0, D <= 2048. */
int vmc_add_layernorm_fwd(float* x, const void* branch, const float* gamma, const float* beta, void* y16, int rows,
                          int D, int ldx, int ldb, float eps, int write_x, int dtype16, void* stream);

/* The same with TWO 16-bit branches, x <- (x + branch0) + branch (in that order) and y16 = LN(x).  Lets the first add+LayerNorm of
 * a pre-norm block (after out_proj; x = x + attention(...), modeling_clip.py / OpenAI clip ResidualAttentionBlock.forward) skip
 * writing the fp32 stream back (write_x = 0) and the second one (after c_proj) redo that add from the kept attention branch:
 * 22 instead of 24 bytes per element and layer, bit-identical stream. */
int vmc_add2_layernorm_fwd(float* x, const void* branch0, const void* branch, const float* gamma, const float* beta, void* y16,
                           int rows, int D, int ldx, int ldb0, int ldb, float eps, int write_x, int dtype16, void* stream);
/* Post-norm block tail of the TFAM AttentionLayer (TFAM/models/AMO_CLIP.py:40,45,50: norm(x + dropout(branch))):
 * s = x (f32 [rows,D]) + branch (16-bit [rows,D]); y = LN(s) written as f32 (y32, next residual operand) and/or 16-bit
 * (y16, next GEMM operand); optional saves for the backward: sum_out = s, mean, rstd.  D % 256 == 0, D <= 2048. */
int vmc_postnorm_fwd(const float* x, const void* branch, const float* gamma, const float* beta, float* sum_out, float* y32,
                     void* y16, float* mean, float* rstd, int rows, int D, float eps, int dtype16, void* stream);
/* The same tail with the branch passed through one or two dropouts first: LN(x + drop2(drop1(branch))) -- the FFN's trailing
 * nn.Dropout and the block's own dropout (AMO_CLIP.py:28,50; :40,45 for the attention tails).  Masks are vmc_dropout's
 * counter-based masks on the flat element index (same seed -> same mask), so the backward can regenerate them.
 * drop_p2 > 0 requires drop_p1 > 0. */
int vmc_postnorm_dropout_fwd(const float* x, const void* branch, const float* gamma, const float* beta, float* sum_out, float* y32,
                             void* y16, float* mean, float* rstd, int rows, int D, float eps, float drop_p1,
                             uint64_t drop_seed1, float drop_p2, uint64_t drop_seed2, int dtype16, void* stream);
/* Backward (autograd of the LayerNorms, train.py:104): dx [rows,D] (f32 or 16-bit per dx_dtype) =
 * LN'(dy) + add, where `add` (optional, dx's dtype/layout) is the gradient arriving over the residual
 * branch that forks at x (fused so the fork needs no separate add pass).  dgamma/dbeta f32 [D],
 * overwritten.  dy f32 or 16-bit.  D <= 2048. */
size_t vmc_layernorm_bwd_workspace_bytes(int rows, int D);
int vmc_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                      const void* add, void* dx, float* dgamma, float* dbeta, int rows, int D, int ldx,
                      int dy_dtype, int x_dtype, int dx_dtype, int dtype16, void* workspace,
                      size_t workspace_bytes, void* stream);
/* The same with TWO incoming gradients dy + dy2 (dy2 16-bit [rows, D], may be NULL): a post-norm LayerNorm output feeds both the fp32
 * residual path and the 16-bit GEMM operand of the next sub-layer (AMO_CLIP.py:40-50), so its backward receives one gradient of each
 * type; summed in fp32 inside the kernel (was: a separate add pass). */
int vmc_layernorm_bwd2(const void* dy, const void* dy2, const void* x, const float* gamma, const float* mean, const float* rstd,
                       const void* add, void* dx, float* dgamma, float* dbeta, int rows, int D, int ldx,
                       int dy_dtype, int x_dtype, int dx_dtype, int dtype16, void* workspace,
                       size_t workspace_bytes, void* stream);
/* Backward of vmc_postnorm_dropout_fwd in one launch (+ the partial reduce of dgamma / dbeta): with the saved pre-norm sum, mean and rstd,
 *   dsum (f32 [rows, D])       = LayerNorm'(dy + dy2)            (dy f32 or 16-bit, dy2 16-bit or NULL, as vmc_layernorm_bwd2)
 *   dbranch16 (16-bit [rows, D]) = cast(dsum * mask1/(1-p1) * mask2/(1-p2))   (the forward's masks, regenerated from the seeds)
 * i.e. what vmc_layernorm_bwd2 followed by vmc_cast_dropout2 (p = 0: a plain cast) computes, bit for bit, without the second pass over
 * the gradient (autograd of `norm(x + dropout(branch))`, TFAM/models/AMO_CLIP.py:40-50 under TFAM/train_and_eval.py:83). */
int vmc_postnorm_bwd(const void* dy, const void* dy2, const float* sum, const float* gamma, const float* mean, const float* rstd,
                     float* dsum, void* dbranch16, float* dgamma, float* dbeta, int rows, int D, int dy_dtype, float p1,
                     uint64_t seed1, float p2, uint64_t seed2, int dtype16, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * K4 — ViT self-attention  softmax(Q K^T / sqrt(dh)) V, no mask, head_dim 64 (MFMA, K/V tile in LDS).
 * Replaces the SDPA inside nn.MultiheadAttention of every CLIP residual block.
 *   qkv 16-bit [F*N, 3*D] packed as the in_proj output: columns [0,D)=Q, [D,2D)=K, [2D,3D)=V, head h at
 *   columns h*64..h*64+63;  out 16-bit [F*N, D].   D = H*64, N <= 288.
 *   lse f32 [F, H, N] optional (log-sum-exp of the scaled scores, saved for the backward).
 */
int vmc_attention_vit_fwd(const void* qkv, void* out, float* lse, int F, int N, int H, int dtype16,
                          void* stream);

/* The same attention for the CLASS-TOKEN query only: what the last residual block of the encoder needs (only x[:, 0] reaches
 * ln_post: `x = self.ln_post(x[:, 0, :])`, OpenAI clip model.py; modeling_clip.py:650 pooled_output = last_hidden_state[:, 0]).
 *   q_cls 16-bit [F, D] (the class rows' queries), kv 16-bit [F*N, 2*D] (columns [0,D)=K, [D,2D)=V of every token),
 *   out 16-bit [F, D].  Same kernel, one 16-row query tile per (frame, head) instead of ceil(N/16). */
int vmc_attention_vit_cls_fwd(const void* q_cls, const void* kv, void* out, int F, int N, int H, int dtype16, void* stream);

/* K4/K11/K12 generic masked attention (fp32 math), any head_dim <= 128 with head_dim % 8 == 0.
 * Replaces F.multi_head_attention_forward's core (q*dh^-1/2, key_padding_mask -> -inf, softmax, @V)
 * for the TFAM self/cross attention (TFAM/models/AMO_CLIP.py:39-45) and serves as the backward's
 * forward-recompute reference.
 *   q  16-bit, element (b, t, h, d) at q[(b*Tq + t)*ldq + h*dh + d];  k, v likewise with Tk, ldk, ldv
 *   key_mask u8 [B, Tk], 1 = attend, 0 = padding (the reference's mask_rgb/mask_flow), may be NULL
 *   out 16-bit [(b*Tq+t)*ldo + h*dh + d];  lse f32 [B,H,Tq] optional.
 *   A query row whose keys are all masked yields NaN, as torch does.
 *   dropout_p > 0 drops attention probabilities after the softmax (nn.MultiheadAttention(dropout=p), train
 *   mode) with the counter-based keep mask hash(dropout_seed, ((b*H+h)*Tq+t)*Tk+key); the backward must be
 *   given the same (p, seed).
 */
int vmc_attention_fwd(const void* q, const void* k, const void* v, const uint8_t* key_mask, void* out,
                      float* lse, int B, int H, int Tq, int Tk, int dh, int ldq, int ldk, int ldv, int ldo,
                      float dropout_p, uint64_t dropout_seed, int dtype16, void* stream);
/* Backward of the above (autograd of the MHA core, train.py:104 / TFAM/train_and_eval.py:82): dq, dk, dv
 * 16-bit with their own row strides, fully overwritten.  workspace >= vmc_attention_bwd_workspace_bytes. */
size_t vmc_attention_bwd_workspace_bytes(int B, int H, int Tq);
int vmc_attention_bwd(const void* q, const void* k, const void* v, const uint8_t* key_mask, const void* out,
                      const void* dout, const float* lse, void* dq, void* dk, void* dv,
                      int B, int H, int Tq, int Tk, int dh, int ldq, int ldk, int ldv, int ldo,
                      int lddq, int lddk, int lddv, float dropout_p, uint64_t dropout_seed, void* workspace,
                      size_t workspace_bytes, int dtype16, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Small element-wise / pooling pieces.
 */
/* rows[f*row_stride + 0..D) = a[0..D) + b[0..D)  for f < F  (class token + positional_embedding[0]). */
int vmc_set_class_rows(void* x, const float* a, const float* b, int F, int D, size_t row_stride, int x_dtype,
                       int dtype16, void* stream);
/* y = act(x) and dx = dy * act'(x) on [n] elements (training path keeps pre-activations). */
int vmc_act_fwd(const void* x, void* y, size_t n, int act, int dtype16, void* stream);
int vmc_act_bwd(const void* x, const void* dy, void* dx, size_t n, int act, int dtype16, void* stream);
/* out[b, :] = mean_t x[b, t, :]  over ALL T rows (TFAM/models/AMO_CLIP.py:170 pools padded rows too;
 * models/student_model.py:93).  x f32 or 16-bit, out16 and/or out32. */
int vmc_mean_pool(const void* x, void* out16, float* out32, int B, int T, int D, int x_dtype, int dtype16,
                  void* stream);
/* x[b,t,:] += pe[t,:]  sinusoidal positional encoding of TFAM/models/AMO_CLIP.py:88-97 (f32 in place). */
int vmc_add_sinusoidal_pe(float* x, int B, int T, int D, void* stream);
/* y = a + alpha * b elementwise, f32. */
int vmc_axpby_f32(const float* a, const float* b, float* y, size_t n, float alpha, float beta, void* stream);
/* y = a + b on flat arrays (sum of the two gradients meeting where a tensor is used twice; autograd of
 * the residual forks, train.py:104); each operand f32 or 16-bit. */
int vmc_add(const void* a, const void* b, void* y, size_t n, int a_dtype, int b_dtype, int y_dtype, int dtype16,
            void* stream);
/* Backward of vmc_mean_pool: dx[b,t,:] = dout[b,:] / T. */
int vmc_mean_pool_bwd(const void* dout, void* dx, int B, int T, int D, int dout_dtype, int dx_dtype, int dtype16,
                      void* stream);
/* Training-path token assembly of K1 (class token concat + positional embedding, OpenAI clip
 * VisionTransformer.forward): x[f,0,:] = cls + pos[0]; x[f,1+p,:] = xp[f*(N-1)+p,:] + pos[1+p].
 * xp 16-bit [F*(N-1), D]; cls f32 [D]; pos f32 [N,D]; x [F*N, D] f32 or 16-bit. */
int vmc_assemble_tokens(const void* xp, const float* cls, const float* pos, void* x, int F, int N, int D,
                        int x_dtype, int dtype16, void* stream);
/* Inverted dropout, y = x * keep/(1-p), keep(i) = hash(seed, i) >= p (nn.Dropout of
 * TFAM/models/AMO_CLIP.py:27-35,84).  Stateless: the backward applies the same call to dy. */
int vmc_dropout(const void* x, void* y, size_t n, float p, uint64_t seed, int x_dtype, int dtype16, void* stream);
/* y16 = cast(x * mask1/(1-p1) * mask2/(1-p2)) in one pass (p = 0 skips a mask): the backward of vmc_postnorm_dropout_fwd's branch,
 * i.e. of `x + self.dropout(branch)` / the FFN's two trailing dropouts (AMO_CLIP.py:28,40,45,50), with the cast to the branch's type. */
int vmc_cast_dropout2(const float* x, void* y16, size_t n, float p1, uint64_t seed1, float p2, uint64_t seed2, int dtype16,
                      void* stream);
/* y = x * scale[0], the scale read from device memory (the scalar gradient arriving at a loss node). */
int vmc_scale_by_device_scalar(const float* x, float* y, size_t n, const float* scale, void* stream);
/* f32 <-> 16-bit casts on flat arrays. */
int vmc_cast_f32_to_16(const float* x, void* y, size_t n, int dtype16, void* stream);
int vmc_cast_16_to_f32(const void* x, float* y, size_t n, int dtype16, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * K10 — losses (forward value + gradient in one pass, fp32).
 * Cosine / MSE distillation: losses.py:5-44.   student, teacher f32 [rows, E] (teacher row stride ldt
 * covers the `rgb_emb[:, :-1]` slice of train.py:98 via teacher_rows_per_clip / teacher_clip_stride).
 *   loss f32[1] (overwritten), dstudent f32 [rows,E] = d loss / d student (may be NULL).
 */
size_t vmc_loss_workspace_bytes(int rows);
int vmc_distill_loss(const float* student, const float* teacher, float* loss, float* dstudent,
                     int rows, int E, int rows_per_clip, size_t teacher_clip_stride, int mode_cosine,
                     void* workspace, size_t workspace_bytes, void* stream);
/* BCE-with-logits, pos_weight = pw*y + 1 (losses.py:59-67; pw < 0 means "None" -> weight 1; also
 * nn.BCEWithLogitsLoss of TFAM/train_and_eval.py:58).  logits/targets f32 [n]; mean over n. */
int vmc_bce_loss(const float* logits, const float* targets, float* loss, float* dlogits, int n, float pos_weight,
                 void* workspace, size_t workspace_bytes, void* stream);
/* Softmax cross entropy, mean over rows — nn.CrossEntropyLoss() of the MammalNet (single-label) variants:
 * train_frame_diff_mn.py:82,102 passes class indices (``labels.argmax(dim=1)``), TFAM/train_and_eval_frame_diff_MN.py:59,83
 * passes the float one-hot rows themselves (probability targets).  Exactly one of target_index (int64 [rows]) and
 * target_prob (f32 [rows,C]) is non-NULL.  logits f32 [rows,C]; loss f32[1]; dlogits f32 [rows,C] or NULL;
 * workspace >= vmc_loss_workspace_bytes(rows). */
int vmc_cross_entropy_loss(const float* logits, const long long* target_index, const float* target_prob, float* loss,
                           float* dlogits, int rows, int C, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * K11-K14 fused — the TFAM forward as a chain of weight-streaming launches for short clips (T, Tk <= 32,
 * d_model 512 / 768, head_dim 64 / 96): AttentionLayer.forward (TFAM/models/AMO_CLIP.py:37-51) in 6 launches
 * with the LayerNorms, both attentions, bias / ReLU / residual adds and casts in prologues and epilogues;
 * the K|V projections of ALL layers' cross attention hoisted into one GEMM over the raw motion tokens
 * (:43-45 projects the same tokens in every layer); mean-pool over all T rows + classifier (:84,:170).
 * Eval-mode arithmetic (dropout = identity).  Shapes outside the supported set return VMC_E_SHAPE and the
 * caller uses the per-op entry points above.
 *
 * Weights are read from two packs the caller fills once per weight version (vmc_cast_weight / plain copies into
 * the offsets vmc_tfam_pack_offset returns):
 *   wpack  16-bit: per layer self_attn.in_proj_weight [3D,D] | self_attn.out_proj.weight | cross_attn.in_proj_weight[0:D]
 *          | cross_attn.out_proj.weight | ffn.0.weight [ff,D] | ffn.3.weight [D,ff]; then, for l = 0..L-1,
 *          cross_attn.in_proj_weight[D:3D] of layer l ([L*2D, D] in all) | classifier.1.weight | classifier.4.weight
 *   ppack  fp32: per layer the six biases in the same order, norm_self / norm_cross / norm_ffn (weight then bias);
 *          then in_proj_bias[D:3D] per layer | classifier.0 weight, bias | classifier.1.bias | classifier.4.bias
 * vmc_tfam_pack_offset(slot, layer, ...) = element offset of a slot (16-bit elements for W slots, floats for P slots;
 * `layer` is ignored by the global slots except KV_ALL, where it selects that layer's 2D rows); *_END = pack size.
 */
enum {
  VMC_TFAM_W_SELF_IN = 0, VMC_TFAM_W_SELF_OUT, VMC_TFAM_W_CROSS_Q, VMC_TFAM_W_CROSS_OUT, VMC_TFAM_W_FFN0, VMC_TFAM_W_FFN3,
  VMC_TFAM_W_KV_ALL, VMC_TFAM_W_CLS1, VMC_TFAM_W_CLS4, VMC_TFAM_W_END,
  VMC_TFAM_P_SELF_IN_B = 16, VMC_TFAM_P_SELF_OUT_B, VMC_TFAM_P_CROSS_Q_B, VMC_TFAM_P_CROSS_OUT_B, VMC_TFAM_P_FFN0_B,
  VMC_TFAM_P_FFN3_B, VMC_TFAM_P_NORM_SELF, VMC_TFAM_P_NORM_CROSS, VMC_TFAM_P_NORM_FFN, VMC_TFAM_P_KV_ALL_B,
  VMC_TFAM_P_CLS_LN, VMC_TFAM_P_CLS1_B, VMC_TFAM_P_CLS4_B, VMC_TFAM_P_END
};
long long vmc_tfam_pack_offset(int slot, int layer, int D, int ff, int L, int C);
/* Pack-time helper for the three linears that consume a LayerNorm output (self_attn.in_proj of layers >= 1 <- norm_ffn of the
 * previous layer; cross_attn q rows <- norm_self; ffn.0 <- norm_cross, or norm_self when has_cross = 0): the chain feeds them
 * the STANDARDISED rows z = (y - mean) * rstd and expects the affine part folded in,
 *   w16_out[r, k] = W[r, k] * gamma[k],   bias_out[r] = bias[r] + sum_k W[r, k] * beta[k]
 * (LN(y) W^T + b = z (W . gamma)^T + (b + W beta)).  W fp32 [rows, cols] contiguous; every other slot is a plain cast / copy. */
int vmc_tfam_fold_layernorm(const float* W, const float* bias, const float* gamma, const float* beta, void* w16_out,
                            float* bias_out, int rows, int cols, int dtype16, void* stream);
/* Scratch for one forward of B clips (activations of one layer at a time + the hoisted K|V); caller-owned. */
size_t vmc_tfam_workspace_bytes(int B, int T, int Tk, int D, int ff, int L, int C, int has_cross);
/* Hoisted cross-attention K|V: ws.kv[B*Tk, L*2D] = motion[B*Tk, D] (fp32) x kv_all^T + bias  (AMO_CLIP.py:43-45, all layers). */
int vmc_tfam_kv_fwd(const float* motion, const void* wpack, const float* ppack, void* workspace, size_t workspace_bytes,
                    int B, int T, int Tk, int D, int H, int ff, int L, int C, int dtype16, void* stream);
/* One AttentionLayer (AMO_CLIP.py:37-51); wpack / ppack as described above, with the LayerNorm-consuming slots folded.  Layer 0 reads the fp32 tokens x_in [B*T, D]; later layers (x_in = NULL)
 * continue from the pre-LayerNorm sum the previous layer left in the workspace.  mask [B,T] / mask_kv [B,Tk]: 1 = real
 * token (the reference's mask_rgb / mask_flow before inversion, :125-126), NULL = all real.  has_cross = 0: self
 * attention + FFN only (rgb-only / flow-only / concatenated-token modes, :136-147,:159-167). */
int vmc_tfam_layer_fwd(const float* x_in, const uint8_t* mask, const uint8_t* mask_kv, const void* wpack, const float* ppack,
                       int layer, void* workspace, size_t workspace_bytes, int B, int T, int Tk, int D, int H, int ff, int L,
                       int C, int has_cross, int dtype16, void* stream);
/* logits[B, C] (fp32) = classifier(mean over all T rows of LN_ffn(last layer))  (AMO_CLIP.py:84,:170). */
int vmc_tfam_head_fwd(const void* wpack, const float* ppack, float* logits, void* workspace, size_t workspace_bytes,
                      int B, int T, int Tk, int D, int H, int ff, int L, int C, int has_cross, int dtype16, void* stream);
/* kv + L layers + head in one call (AMO_CLIP.forward, :99-171, eval mode). */
int vmc_tfam_forward(const float* x, const float* motion, const uint8_t* mask, const uint8_t* mask_kv, const void* wpack,
                     const float* ppack, float* logits, void* workspace, size_t workspace_bytes, int B, int T, int Tk,
                     int D, int H, int ff, int L, int C, int has_cross, int dtype16, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * K11-K14 fused, TRAINING — forward + backward of the TFAM block for short clips as two launch chains
 * (TFAM/models/AMO_CLIP.py:37-51,99-171 in train mode under TFAM/train_and_eval.py:66-101: `output = model(...)`,
 * `loss.backward()`).  Same shape set as the eval chain above.
 *
 * Forward: the six launches per layer of the eval chain with the training arithmetic added in place -- LayerNorm with its
 * affine part in the GEMM prologue (unfolded weights), dropout on the attention probabilities inside the attention
 * prologue, the branch dropouts (AMO_CLIP.py:40,45,50 and ffn's own nn.Dropout :27-28) inside the GEMM epilogues -- and the
 * tensors the backward needs written as side outputs (16-bit GEMM operands, pre-norm sums, keep masks, lse).
 * Backward, per layer (8 launches + 1): LayerNorm backward in the PROLOGUE of the dgrad GEMM that consumes the branch
 * gradient (three times), the two dgrad GEMMs with the residual gradient added in the epilogue, vmc_attention_bwd twice,
 * the qkv dgrad, and ONE grouped launch for all seven weight gradients, their bias gradients and the three LayerNorms'
 * parameter gradients.  Gradients are WRITTEN (not accumulated) through the pointers below; NULL = not wanted.
 *
 * Parameters are passed as structs of device pointers (no packing, nothing copied): 16-bit compute copies [N, K] and their
 * transposes [K, N] (what vmc_cast_weights_multi keeps current after every optimiser step), fp32 masters for biases and
 * LayerNorm parameters, fp32 gradient destinations (e.g. views into a flat gradient arena).
 *   w_cross_in = cross_attn.in_proj_weight [3D, D] (q rows 0:D, k|v rows D:3D), wt_cross_in its transpose [D, 3D].
 *
 * Dropout: p_drop (AttentionLayer's dropout: attention probabilities, the three branch dropouts, the FFN's two), p_mlp
 * (classifier.3).  seeds[7*L + 1] host array, each a value or (bit 63) a device address (include "seed arguments"): per layer
 * {self-attn P, self branch, cross-attn P, cross branch, FFN inner, FFN trailing, FFN branch}, then the classifier's.
 * The element index of every mask equals the per-op path's (vmc_dropout on the [M, N] tensor; vmc_attention_fwd's
 * (b, h, q, k)), so both paths draw identical masks from identical seeds.  The backward must be given the same values.
 */
typedef struct vmc_tfam_layer_params {
  const void *w_self_in, *w_self_out, *w_cross_in, *w_cross_out, *w_ffn0, *w_ffn3;          /* 16-bit [N, K] */
  const void *wt_self_in, *wt_self_out, *wt_cross_in, *wt_cross_out, *wt_ffn0, *wt_ffn3;    /* 16-bit [K, N] */
  const float *b_self_in, *b_self_out, *b_cross_in, *b_cross_out, *b_ffn0, *b_ffn3;
  const float *ln_self_g, *ln_self_b, *ln_cross_g, *ln_cross_b, *ln_ffn_g, *ln_ffn_b;
  float *gw_self_in, *gw_self_out, *gw_cross_in, *gw_cross_out, *gw_ffn0, *gw_ffn3;
  float *gb_self_in, *gb_self_out, *gb_cross_in, *gb_cross_out, *gb_ffn0, *gb_ffn3;
  float *g_ln_self_g, *g_ln_self_b, *g_ln_cross_g, *g_ln_cross_b, *g_ln_ffn_g, *g_ln_ffn_b;
} vmc_tfam_layer_params;
typedef struct vmc_tfam_head_params {
  const void *w_cls1, *w_cls4;                   /* 16-bit classifier.1.weight [D/2, D], classifier.4.weight [C, D/2] */
  const float *w32_cls1, *w32_cls4;              /* their fp32 masters (the head's backward is 8..16 rows of work: fp32 FMAs) */
  const float *cls_ln_g, *cls_ln_b, *b_cls1, *b_cls4;
  float *g_cls_ln_g, *g_cls_ln_b, *gw_cls1, *gb_cls1, *gw_cls4, *gb_cls4;
} vmc_tfam_head_params;
/* Saved activations + backward scratch of one step of B clips; caller-owned, must stay untouched between fwd and bwd. */
size_t vmc_tfam_train_workspace_bytes(int B, int T, int Tk, int D, int H, int ff, int L, int C, int has_cross);
/* AttentionLayer.forward in train mode.  layers = all L structs (layer l > 0 reads layer l-1's norm_ffn);  x_in: the fp32 tokens
 * (layer 0; NULL otherwise); motion: raw fp32 motion tokens [B*Tk, D] (has_cross).  seeds: the 7 seeds of THIS layer. */
int vmc_tfam_layer_train_fwd(const float* x_in, const float* motion, const uint8_t* mask, const uint8_t* mask_kv,
                             const vmc_tfam_layer_params* layers, int layer, void* workspace, size_t workspace_bytes,
                             int B, int T, int Tk, int D, int H, int ff, int L, int C, int has_cross, float p_drop,
                             const uint64_t* seeds, int dtype16, void* stream);
/* mean-pool + classifier in train mode (AMO_CLIP.py:84,:170), logits fp32 [B, C]. */
int vmc_tfam_head_train_fwd(const vmc_tfam_layer_params* layers, const vmc_tfam_head_params* head, float* logits,
                            void* workspace, size_t workspace_bytes, int B, int T, int Tk, int D, int H, int ff, int L, int C,
                            int has_cross, float p_mlp, uint64_t seed, int dtype16, void* stream);
/* Backward of the head: from dlogits fp32 [B, C] to the gradient wrt the last layer's output (left in the workspace) and
 * the classifier's parameter gradients (autograd of AMO_CLIP.py:170-171). */
int vmc_tfam_head_bwd(const float* dlogits, const vmc_tfam_layer_params* layers, const vmc_tfam_head_params* head,
                      void* workspace, size_t workspace_bytes, int B, int T, int Tk, int D, int H, int ff, int L, int C,
                      int has_cross, float p_mlp, uint64_t seed, int dtype16, void* stream);
/* Backward of one AttentionLayer (autograd of AMO_CLIP.py:37-51): consumes the gradient wrt its output from the workspace,
 * leaves the gradient wrt its input there for layer-1 (not computed for layer 0: the tokens need no gradient), writes the
 * layer's parameter gradients.  x_in (layer 0 only) is not read: its 16-bit cast was saved by the forward. */
int vmc_tfam_layer_bwd(const uint8_t* mask, const uint8_t* mask_kv, const vmc_tfam_layer_params* layers, int layer,
                       void* workspace, size_t workspace_bytes, int B, int T, int Tk, int D, int H, int ff, int L, int C,
                       int has_cross, float p_drop, const uint64_t* seeds, int dtype16, void* stream);
/* Whole chains: L x layer_train_fwd + head_train_fwd; head_bwd + L x layer_bwd (last layer first). */
int vmc_tfam_train_fwd(const float* x, const float* motion, const uint8_t* mask, const uint8_t* mask_kv,
                       const vmc_tfam_layer_params* layers, const vmc_tfam_head_params* head, float* logits, void* workspace,
                       size_t workspace_bytes, int B, int T, int Tk, int D, int H, int ff, int L, int C, int has_cross,
                       float p_drop, float p_mlp, const uint64_t* seeds, int dtype16, void* stream);
int vmc_tfam_train_bwd(const float* dlogits, const uint8_t* mask, const uint8_t* mask_kv, const vmc_tfam_layer_params* layers,
                       const vmc_tfam_head_params* head, void* workspace, size_t workspace_bytes, int B, int T, int Tk, int D,
                       int H, int ff, int L, int C, int has_cross, float p_drop, float p_mlp,
                       const uint64_t* seeds, int dtype16, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * K15 — fused Adam / AdamW over one flat fp32 buffer (train.py:66; TFAM/train_and_eval.py:53).
 *   decoupled_wd = 1: p *= 1 - lr*wd first (AdamW); 0: g += wd*p (Adam L2).
 *   grad_scale multiplies g first (clip_grad_norm_ coefficient / DDP averaging).
 */
int vmc_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                  float eps, float weight_decay, int decoupled_wd, int step, float grad_scale, void* stream);
/* Device-resident step state for hipGraph-captured training steps (the step count, the bias corrections, the learning rate and
 * the dropout seeds must not be host scalars frozen into the graph -- ADVICE r1).
 *   state  u64 [2 + n_seeds] in device memory: [0] step count t (start at 0), [1] base seed, [2..] this step's dropout seeds
 *   hyper  f32 [4], 16-byte aligned: {lr, lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t), grad_scale}; the host writes [0] and [3]
 * vmc_train_tick: t += 1, recomputes hyper[1..2] and the n_seeds seeds (splitmix of base seed, t and the index); enqueue it once
 * per step before the forward.  vmc_adam_step_dev = vmc_adam_step reading its four scalars from `hyper`.
 * SEED ARGUMENTS (vmc_dropout, vmc_postnorm_dropout_fwd, vmc_attention_fwd / _bwd): a value < 2^63 is the seed itself; with bit
 * 63 set the low 63 bits are the ADDRESS of a u64 in device memory holding it (e.g. &state[2 + i]), read at kernel start. */
int vmc_train_tick(void* state, float* hyper, float beta1, float beta2, int n_seeds, void* stream);
int vmc_adam_step_dev(float* p, const float* g, float* m, float* v, size_t n, const float* hyper, float beta1, float beta2,
                      float eps, float weight_decay, int decoupled_wd, void* stream);
/* The same update launched with at most max_workgroups workgroups of 256 threads: the geometry for an update that runs on a side
 * stream BESIDE other kernels (FusedAdam.enable_backward_overlap: 2 per CU leaves the wave slots the backward's workgroups need). */
int vmc_adam_step_dev_bg(float* p, const float* g, float* m, float* v, size_t n, const float* hyper, float beta1, float beta2,
                         float eps, float weight_decay, int decoupled_wd, int max_workgroups, void* stream);
/* AdamW update AND refresh of the 16-bit compute copies in one pass over the masters (what `optimizer.step()` followed by
 * vmc_cast_weights_multi does in two: TFAM/train_and_eval.py:96 in a captured step).  desc / n_desc / total_tiles: the records of
 * vmc_cast_weights_multi, every `w` pointing into the flat parameter arena p_base; g_base / m_base / v_base are the parallel arenas
 * (element i of a tensor at the same offset in all four).  ranges: DEVICE array of n_ranges records of 16 bytes
 * {uint64 offset; int32 n; int32 block0} for the arena slices that have no compute copy (biases, LayerNorm parameters, ...), record i
 * owning blocks [block0, block0 of i+1) of 1024 elements.  Same arithmetic as vmc_adam_step_dev, bit for bit. */
int vmc_adam_cast_multi(const void* desc, int n_desc, int total_tiles, const void* ranges, int n_ranges, int total_range_blocks,
                        float* p_base, const float* g_base, float* m_base, float* v_base, const float* hyper, float beta1, float beta2,
                        float eps, float weight_decay, int decoupled_wd, int dtype16, void* stream);
/* sum of squares of a flat f32 buffer, accumulated (+=) into out[0] (global grad norm). */
int vmc_sumsq(const float* x, size_t n, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VMC_H */
