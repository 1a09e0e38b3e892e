// Losses (value + gradient in one pass) and the fused Adam/AdamW update (gfx950, fp32, HBM-bound).
#include "common.h"

// ---- distillation loss (losses.py:5-44) ---------------------------------------------------------
// One wave per row.  Cosine: ns = max(|s|, eps), nt = max(|t|, eps), c = clamp(s.t/(ns nt), +-(1-eps)),
// loss = mean(1 - c).  d loss / d s = -(1/rows) * [ t/(ns nt) - (s.t) s / (ns^3 nt) * 1{|s| > eps} ] * 1{c not clamped}.
// Row partial losses go to the workspace; a single-block kernel sums them in a fixed order.
__global__ void __launch_bounds__(256) distill_rows_kernel(const float* __restrict__ s, const float* __restrict__ t,
                                                           float* __restrict__ row_loss, float* __restrict__ ds, int rows, int E,
                                                           int rows_per_clip, size_t teacher_clip_stride, int cosine) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* sr = s + (size_t)row * E;
  const float* tr = t + (size_t)(row / rows_per_clip) * teacher_clip_stride + (size_t)(row % rows_per_clip) * E;
  const float eps = 1e-5f;
  if (cosine) {
    float dot = 0.f, ss = 0.f, tt = 0.f;
    for (int i = lane; i < E; i += 64) {
      const float a = sr[i], b = tr[i];
      dot += a * b; ss += a * a; tt += b * b;
    }
    dot = wave_sum(dot); ss = wave_sum(ss); tt = wave_sum(tt);
    const float sn_raw = sqrtf(ss), tn_raw = sqrtf(tt);
    const float sn = fmaxf(sn_raw, eps), tn = fmaxf(tn_raw, eps);
    const float c_raw = dot / (sn * tn);
    const float c = fminf(fmaxf(c_raw, -1.0f + eps), 1.0f - eps);
    if (lane == 0) row_loss[row] = 1.0f - c;
    if (ds) {
      // torch.clamp passes gradient on the closed interval [min, max]
      const bool pass = (c_raw >= -1.0f + eps) && (c_raw <= 1.0f - eps);
      const float k = pass ? -1.0f / (float)rows : 0.0f;
      const float inv = 1.0f / (sn * tn);
      // norm clamp(min=eps) passes gradient when |s| >= eps; d|s|/ds = s/|s| (0 at s = 0)
      const float kn = (sn_raw >= eps && sn_raw > 0.f) ? dot / (sn * sn * tn * sn_raw) : 0.0f;
      for (int i = lane; i < E; i += 64) ds[(size_t)row * E + i] = k * (tr[i] * inv - sr[i] * kn);
    }
  } else {
    float acc = 0.f;
    const float k = 2.0f / ((float)rows * (float)E);
    for (int i = lane; i < E; i += 64) {
      const float d = sr[i] - tr[i];
      acc += d * d;
      if (ds) ds[(size_t)row * E + i] = k * d;
    }
    acc = wave_sum(acc);
    if (lane == 0) row_loss[row] = acc / (float)E;
  }
}

__global__ void __launch_bounds__(256) sum_scale_kernel(const float* __restrict__ part, float* __restrict__ out, int n, float scale) {
  __shared__ float sm[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) a += part[i];
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = ((sm[0] + sm[1]) + (sm[2] + sm[3])) * scale;
}

extern "C" size_t vmc_loss_workspace_bytes(int rows) { return (size_t)(rows > 0 ? rows : 1) * sizeof(float); }

extern "C" int vmc_distill_loss(const float* student, const float* teacher, float* loss, float* dstudent, int rows, int E,
                                int rows_per_clip, size_t teacher_clip_stride, int mode_cosine, void* workspace,
                                size_t workspace_bytes, void* stream) {
  if (!student || !teacher || !loss || !workspace || rows <= 0 || E <= 0 || rows_per_clip <= 0) return VMC_E_ARG;
  if (workspace_bytes < vmc_loss_workspace_bytes(rows)) return VMC_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(distill_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, student, teacher, (float*)workspace, dstudent, rows, E,
                     rows_per_clip, teacher_clip_stride, mode_cosine);
  VMC_CHECK_LAUNCH();
  hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, s, (const float*)workspace, loss, rows, 1.0f / (float)rows);
  VMC_CHECK_LAUNCH();
  return 0;
}

// ---- BCE with logits, pos_weight = pw*y + 1 (losses.py:59-67) ------------------------------------
// l = (1-y) x + (1 + (w-1) y) (log1p(exp(-|x|)) + max(-x, 0));  dl/dx = (1-y) - (1 + (w-1) y) (1 - sigmoid(x))
#define BCE_BLOCKS 64
__global__ void __launch_bounds__(256) bce_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ part,
                                                  float* __restrict__ dx, int n, float pw, float* __restrict__ loss_out) {
  __shared__ float sm[4];
  float acc = 0.f;
  const float invn = 1.0f / (float)n;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float xi = x[i], yi = y[i];
    const float w = pw < 0.f ? 1.0f : pw * yi + 1.0f;
    const float lw = 1.0f + (w - 1.0f) * yi;
    acc += (1.0f - yi) * xi + lw * (log1pf(expf(-fabsf(xi))) + fmaxf(-xi, 0.0f));
    if (dx) {
      const float sig = 1.0f / (1.0f + expf(-xi));
      dx[i] = ((1.0f - yi) - lw * (1.0f - sig)) * invn;
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float tot = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    if (loss_out != nullptr) loss_out[0] = tot * invn;      // single-workgroup launch (small n): no second pass
    else part[blockIdx.x] = tot;
  }
}

extern "C" int vmc_bce_loss(const float* logits, const float* targets, float* loss, float* dlogits, int n, float pos_weight,
                            void* workspace, size_t workspace_bytes, void* stream) {
  if (!logits || !targets || !loss || !workspace || n <= 0) return VMC_E_ARG;
  if (workspace_bytes < BCE_BLOCKS * sizeof(float)) return VMC_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  int blocks = (n + 255) / 256;
  if (blocks > BCE_BLOCKS) blocks = BCE_BLOCKS;
  if (n <= 8192) {      // a TFAM batch of a few clips x 140 classes: one workgroup, one launch (the launch count is the cost there)
    hipLaunchKernelGGL(bce_kernel, dim3(1), dim3(256), 0, s, logits, targets, (float*)workspace, dlogits, n, pos_weight, loss);
    VMC_CHECK_LAUNCH();
    return 0;
  }
  hipLaunchKernelGGL(bce_kernel, dim3(blocks), dim3(256), 0, s, logits, targets, (float*)workspace, dlogits, n, pos_weight, (float*)nullptr);
  VMC_CHECK_LAUNCH();
  hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, s, (const float*)workspace, loss, blocks, 1.0f / (float)n);
  VMC_CHECK_LAUNCH();
  return 0;
}

// ---- softmax cross entropy (nn.CrossEntropyLoss, mean over rows) ------------------------------------
// One wave per row: m = max x, lse = m + log sum exp(x - m).  Index targets: l = lse - x[t].  Probability targets
// (float [rows, C], what the MammalNet TFAM loop passes): l = sum_c y_c (lse - x_c).  d l / d x_c = softmax_c * sum(y) - y_c.
__global__ void __launch_bounds__(256) ce_rows_kernel(const float* __restrict__ x, const long long* __restrict__ tidx,
                                                      const float* __restrict__ tprob, float* __restrict__ row_loss,
                                                      float* __restrict__ dx, int rows, int C) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (size_t)row * C;
  float m = -INFINITY;
  for (int c = lane; c < C; c += 64) m = fmaxf(m, xr[c]);
  m = wave_max(m);
  float se = 0.f, ysum = 0.f, yx = 0.f;
  for (int c = lane; c < C; c += 64) {
    se += expf(xr[c] - m);
    if (tprob) {
      const float y = tprob[(size_t)row * C + c];
      ysum += y;
      yx += y * xr[c];
    }
  }
  se = wave_sum(se);
  const float lse = m + logf(se);
  long long t = -1;
  if (tprob) {
    ysum = wave_sum(ysum);
    yx = wave_sum(yx);
  } else {
    t = tidx[row];
    ysum = 1.0f;
    yx = (t >= 0 && t < C) ? xr[t] : 0.0f;
  }
  if (lane == 0) row_loss[row] = ysum * lse - yx;
  if (dx) {
    const float inv = 1.0f / (float)rows, rse = 1.0f / se;
    for (int c = lane; c < C; c += 64) {
      const float y = tprob ? tprob[(size_t)row * C + c] : (c == t ? 1.0f : 0.0f);
      dx[(size_t)row * C + c] = (expf(xr[c] - m) * rse * ysum - y) * inv;
    }
  }
}

extern "C" int vmc_cross_entropy_loss(const float* logits, const long long* target_index, const float* target_prob, float* loss,
                                      float* dlogits, int rows, int C, void* workspace, size_t workspace_bytes, void* stream) {
  if (!logits || !loss || !workspace || rows <= 0 || C <= 0) return VMC_E_ARG;
  if ((target_index == nullptr) == (target_prob == nullptr)) return VMC_E_ARG;       // exactly one kind of target
  if (workspace_bytes < vmc_loss_workspace_bytes(rows)) return VMC_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(ce_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, logits, target_index, target_prob, (float*)workspace, dlogits, rows, C);
  VMC_CHECK_LAUNCH();
  hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, s, (const float*)workspace, loss, rows, 1.0f / (float)rows);
  VMC_CHECK_LAUNCH();
  return 0;
}

// ---- fused Adam / AdamW ---------------------------------------------------------------------------
// 16 B read (p, g, m, v) + 12 B write per parameter; float4 vectorised, grid-stride.
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps,
                                                   float wd, int decoupled, float step_size, float inv_sqrt_bc2, float gscale,
                                                   const float* __restrict__ hyper) {
  if (hyper != nullptr) {      // captured training step: {lr, lr / (1 - b1^t), 1 / sqrt(1 - b2^t), grad_scale} live in device memory
    lr = hyper[0]; step_size = hyper[1]; inv_sqrt_bc2 = hyper[2]; gscale = hyper[3];
  }
  const size_t n4 = n >> 2;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  // two float4 of each of the four streams in flight per thread: a grid of 2 workgroups per CU (what the backward-overlapped
  // step launches, so that the update never takes the wave slots the backward's workgroups need) still covers the HBM latency
  for (size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += 2 * stride) {
    const size_t i1 = i0 + stride;
    const bool two = i1 < n4;
    float4 pp[2], gg[2], mm[2], vv[2];
    pp[0] = ((float4*)p)[i0]; gg[0] = ((const float4*)g)[i0]; mm[0] = ((float4*)m)[i0]; vv[0] = ((float4*)v)[i0];
    if (two) { pp[1] = ((float4*)p)[i1]; gg[1] = ((const float4*)g)[i1]; mm[1] = ((float4*)m)[i1]; vv[1] = ((float4*)v)[i1]; }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (u == 1 && !two) break;
      float* P = &pp[u].x; const float* G = &gg[u].x; float* M = &mm[u].x; float* V = &vv[u].x;
#pragma unroll
      for (int j = 0; j < 4; ++j) adam_element(P[j], G[j], M[j], V[j], lr, b1, b2, eps, wd, decoupled, step_size, inv_sqrt_bc2, gscale);
      const size_t i = u ? i1 : i0;
      ((float4*)p)[i] = pp[u]; ((float4*)m)[i] = mm[u]; ((float4*)v)[i] = vv[u];
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t k = (n4 << 2) + threadIdx.x;
    float pk = p[k], mk = m[k], vk = v[k];
    adam_element(pk, g[k], mk, vk, lr, b1, b2, eps, wd, decoupled, step_size, inv_sqrt_bc2, gscale);
    p[k] = pk; m[k] = mk; v[k] = vk;
  }
}

extern "C" int vmc_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                             float weight_decay, int decoupled_wd, int step, float grad_scale, void* stream) {
  if (!p || !g || !m || !v || n == 0 || step <= 0) return VMC_E_ARG;
  if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return VMC_E_ALIGN;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for((n + 3) / 4, 256, 256 * 8)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1,
                     beta2, eps, weight_decay, decoupled_wd, step_size, inv_sqrt_bc2, grad_scale, (const float*)nullptr);
  VMC_CHECK_LAUNCH();
  return 0;
}

// ---- device-resident step state (hipGraph-captured training steps) ------------------------------------------------------
// state[0] = step count t, state[1] = base seed, state[2 .. 2 + n_seeds) = the dropout seeds of this step (one per call site);
// hyper = {lr (host-written), lr / (1 - b1^t), 1 / sqrt(1 - b2^t), grad_scale (host-written)}.
__global__ void train_tick_kernel(unsigned long long* __restrict__ state, float* __restrict__ hyper, float b1, float b2, int n_seeds) {
  __shared__ unsigned long long t_sh;
  if (threadIdx.x == 0) {
    const unsigned long long t = state[0] + 1;
    state[0] = t;
    t_sh = t;
    const double bc1 = 1.0 - pow((double)b1, (double)t), bc2 = 1.0 - pow((double)b2, (double)t);
    hyper[1] = (float)((double)hyper[0] / bc1);
    hyper[2] = (float)(1.0 / sqrt(bc2));
  }
  __syncthreads();
  const unsigned long long t = t_sh, base = state[1];
  for (int i = threadIdx.x; i < n_seeds; i += blockDim.x) {
    unsigned long long x = base ^ (t * 0x9E3779B97F4A7C15ull + (unsigned long long)i * 0xD1B54A32D192ED03ull);
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    x ^= x >> 31;
    state[2 + i] = x & ~VMC_SEED_IS_PTR;        // a seed VALUE never carries the pointer tag
  }
}

extern "C" int vmc_train_tick(void* state, float* hyper, float beta1, float beta2, int n_seeds, void* stream) {
  if (!state || !hyper || n_seeds < 0) return VMC_E_ARG;
  hipLaunchKernelGGL(train_tick_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (unsigned long long*)state, hyper, beta1, beta2, n_seeds);
  VMC_CHECK_LAUNCH();
  return 0;
}

extern "C" int vmc_adam_step_dev(float* p, const float* g, float* m, float* v, size_t n, const float* hyper, float beta1, float beta2,
                                 float eps, float weight_decay, int decoupled_wd, void* stream) {
  return vmc_adam_step_dev_bg(p, g, m, v, n, hyper, beta1, beta2, eps, weight_decay, decoupled_wd, 256 * 8, stream);
}
extern "C" int vmc_adam_step_dev_bg(float* p, const float* g, float* m, float* v, size_t n, const float* hyper, float beta1, float beta2,
                                    float eps, float weight_decay, int decoupled_wd, int max_workgroups, void* stream) {
  if (!p || !g || !m || !v || !hyper || n == 0 || max_workgroups <= 0) return VMC_E_ARG;
  if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)hyper) & 15) return VMC_E_ALIGN;
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for((n + 3) / 4, 256, max_workgroups)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, 0.f, beta1,
                     beta2, eps, weight_decay, decoupled_wd, 0.f, 0.f, 0.f, hyper);
  VMC_CHECK_LAUNCH();
  return 0;
}

// ---- sum of squares (global grad-norm for clip_grad_norm_, train.py:105-106) ----------------------
__global__ void __launch_bounds__(256) sumsq_kernel(const float* __restrict__ x, size_t n, float* __restrict__ out) {
  __shared__ float sm[4];
  float a = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a += x[i] * x[i];
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, (sm[0] + sm[1]) + (sm[2] + sm[3]));
}
extern "C" int vmc_sumsq(const float* x, size_t n, float* out, void* stream) {
  if (!x || !out || n == 0) return VMC_E_ARG;
  hipLaunchKernelGGL(sumsq_kernel, dim3(grid_for(n, 256, 1024)), dim3(256), 0, (hipStream_t)stream, x, n, out);
  VMC_CHECK_LAUNCH();
  return 0;
}
