// HBM-bound helper kernels: frame preprocess + patch extraction, casts, transposes, pooling (gfx950).
#include "common.h"

extern "C" int vmc_abi_version(void) { return 1; }

extern "C" const char* vmc_error_string(int code) {
  switch (code) {
    case 0: return "success";
    case VMC_E_ARG: return "vmc: bad argument (null pointer, non-positive size or too-small workspace)";
    case VMC_E_ALIGN: return "vmc: pointer or stride violates the documented alignment";
    case VMC_E_SHAPE: return "vmc: unsupported shape";
    case VMC_E_DTYPE: return "vmc: unsupported dtype combination";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "vmc: unknown error";
  }
}

// ---- K0: u8 NCHW frames -> normalised 16-bit patch matrix --------------------------------------
// One thread per 4 consecutive pixels of one (frame, channel, row): coalesced 4-byte reads; the four
// results land in (at most two) patch rows.
template <typename T>
__global__ void __launch_bounds__(256) preprocess_kernel(const uint8_t* __restrict__ frames, uint16_t* __restrict__ patches,
                                                         int F, int R, int p, int kpad, int wrap) {
  const int g = R / p;
  const int quads_per_row = R >> 2;
  const size_t total = (size_t)F * 3 * R * quads_per_row;
  const float mean[3] = {0.48145466f, 0.4578275f, 0.40821073f};
  const float istd[3] = {1.0f / 0.26862954f, 1.0f / 0.26130258f, 1.0f / 0.27577711f};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int xq = (int)(i % quads_per_row);
    size_t t = i / quads_per_row;
    const int y = (int)(t % R);
    t /= R;
    const int c = (int)(t % 3);
    const int f = (int)(t / 3);
    const uint32_t px = *(const uint32_t*)(frames + i * 4);
    const int py = y / p, dy = y % p;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int x = xq * 4 + j;
      uint32_t v = (px >> (8 * j)) & 0xFFu;
      if (wrap) v = (256u - v) & 0xFFu;
      // same operation order as ToTensor (/255) then Normalize ((x - mean) / std)
      const float val = ((float)v / 255.0f - mean[c]) * istd[c];
      const int pxi = x / p, dx = x % p;
      const size_t prow = ((size_t)f * g + py) * g + pxi;
      patches[prow * kpad + (c * p + dy) * p + dx] = T::from_f32(val);
    }
  }
}

__global__ void zero_pad_cols_kernel(uint16_t* __restrict__ patches, size_t rows, int k, int kpad) {
  const int padw = kpad - k;
  const size_t total = rows * padw;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
    patches[(i / padw) * kpad + k + (i % padw)] = 0;
}

extern "C" int vmc_preprocess_patches_u8(const uint8_t* frames, void* patches, int F, int R, int p, int kpad, int wrap_quirk,
                                         int dtype16, void* stream) {
  if (!frames || !patches || F <= 0 || R <= 0 || p <= 0) return VMC_E_ARG;
  if (R % p || R % 4 || kpad < 3 * p * p || kpad % 8) return VMC_E_SHAPE;
  if ((uintptr_t)frames & 3) return VMC_E_ALIGN;
  const size_t total = (size_t)F * 3 * R * (R / 4);
  hipStream_t s = (hipStream_t)stream;
  const int k = 3 * p * p;
  if (kpad > k) {
    const size_t rows = (size_t)F * (R / p) * (R / p);
    hipLaunchKernelGGL(zero_pad_cols_kernel, dim3(grid_for(rows * (kpad - k), 256)), dim3(256), 0, s, (uint16_t*)patches, rows, k, kpad);
    VMC_CHECK_LAUNCH();
  }
  if (dtype16 == VMC_BF16)
    hipLaunchKernelGGL(preprocess_kernel<BF16>, dim3(grid_for(total, 256)), dim3(256), 0, s, frames, (uint16_t*)patches, F, R, p, kpad, wrap_quirk);
  else if (dtype16 == VMC_F16)
    hipLaunchKernelGGL(preprocess_kernel<F16>, dim3(grid_for(total, 256)), dim3(256), 0, s, frames, (uint16_t*)patches, F, R, p, kpad, wrap_quirk);
  else
    return VMC_E_DTYPE;
  VMC_CHECK_LAUNCH();
  return 0;
}

// Same patch extraction for frames that are already normalised floats (HF `pixel_values`, the
// argument of CLIPModel.get_image_features at extract_embeddings.py:94).
template <typename T>
__global__ void __launch_bounds__(256) patches_f32_kernel(const float* __restrict__ pix, uint16_t* __restrict__ patches, int F, int R,
                                                          int p, int kpad) {
  const int g = R / p;
  const int quads_per_row = R >> 2;
  const size_t total = (size_t)F * 3 * R * quads_per_row;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int xq = (int)(i % quads_per_row);
    size_t t = i / quads_per_row;
    const int y = (int)(t % R);
    t /= R;
    const int c = (int)(t % 3);
    const int f = (int)(t / 3);
    const float4 px = *(const float4*)(pix + i * 4);
    const float v[4] = {px.x, px.y, px.z, px.w};
    const int py = y / p, dy = y % p;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int x = xq * 4 + j;
      const int pxi = x / p, dx = x % p;
      const size_t prow = ((size_t)f * g + py) * g + pxi;
      patches[prow * kpad + (c * p + dy) * p + dx] = T::from_f32(v[j]);
    }
  }
}

extern "C" int vmc_patches_f32(const float* pixel_values, void* patches, int F, int R, int p, int kpad, int dtype16, void* stream) {
  if (!pixel_values || !patches || F <= 0 || R <= 0 || p <= 0) return VMC_E_ARG;
  if (R % p || R % 4 || kpad < 3 * p * p || kpad % 8) return VMC_E_SHAPE;
  if ((uintptr_t)pixel_values & 15) return VMC_E_ALIGN;
  const size_t total = (size_t)F * 3 * R * (R / 4);
  hipStream_t s = (hipStream_t)stream;
  const int k = 3 * p * p;
  if (kpad > k) {
    const size_t rows = (size_t)F * (R / p) * (R / p);
    hipLaunchKernelGGL(zero_pad_cols_kernel, dim3(grid_for(rows * (kpad - k), 256)), dim3(256), 0, s, (uint16_t*)patches, rows, k, kpad);
    VMC_CHECK_LAUNCH();
  }
  if (dtype16 == VMC_BF16)
    hipLaunchKernelGGL(patches_f32_kernel<BF16>, dim3(grid_for(total, 256)), dim3(256), 0, s, pixel_values, (uint16_t*)patches, F, R, p, kpad);
  else if (dtype16 == VMC_F16)
    hipLaunchKernelGGL(patches_f32_kernel<F16>, dim3(grid_for(total, 256)), dim3(256), 0, s, pixel_values, (uint16_t*)patches, F, R, p, kpad);
  else
    return VMC_E_DTYPE;
  VMC_CHECK_LAUNCH();
  return 0;
}

// ---- split-precision patch operands (inference): the patch-embedding GEMM is 0.2 % of the encoder's FLOPs but its 16-bit
// operand rounding was the largest single contribution to the embedding error (tests/test_gpu_encoder.py per-stage trace: the
// stream error right after ln_pre was 60-75 % of the final one).  Two extra K slices make that GEMM fp32-accurate:
//   u8 frames:   A = [v | v]            (v = raw pixel 0..255, exact in 16 bits),  W = [W'_hi | W'_lo],  W' = W / std_c,
//                bias' = -255 sum_k W mean_c / std_c,  alpha = 1/255   ->  alpha (A W^T + bias') = sum_k ((v/255 - mean)/std) W
//   f32 pixels:  A = [x_hi | x_lo | x_hi],  W = [W_hi | W_hi | W_lo]   (x_lo = x - x_hi: the product's cross terms)
template <typename T>
__global__ void __launch_bounds__(256) patches_u8_exact_kernel(const uint8_t* __restrict__ frames, uint16_t* __restrict__ patches,
                                                               int F, int R, int p, int kpad, int wrap) {
  const int g = R / p;
  const int quads_per_row = R >> 2;
  const size_t total = (size_t)F * 3 * R * quads_per_row;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int xq = (int)(i % quads_per_row);
    size_t t = i / quads_per_row;
    const int y = (int)(t % R);
    t /= R;
    const int c = (int)(t % 3);
    const int f = (int)(t / 3);
    const uint32_t px = *(const uint32_t*)(frames + i * 4);
    const int py = y / p, dy = y % p;
    if ((p & 1) == 0) {        // even patch size (14 / 16 / 32): a pixel pair at an even x never straddles a patch -> 4-byte stores
#pragma unroll
      for (int j = 0; j < 4; j += 2) {
        const int x = xq * 4 + j;
        uint32_t v0 = (px >> (8 * j)) & 0xFFu, v1 = (px >> (8 * j + 8)) & 0xFFu;
        if (wrap) { v0 = (256u - v0) & 0xFFu; v1 = (256u - v1) & 0xFFu; }
        const int pxi = x / p, dx = x % p;
        const size_t prow = ((size_t)f * g + py) * g + pxi;
        const uint32_t h2 = (uint32_t)T::from_f32((float)v0) | ((uint32_t)T::from_f32((float)v1) << 16);
        uint16_t* dst = patches + prow * (2 * (size_t)kpad) + (c * p + dy) * p + dx;
        *(uint32_t*)dst = h2;
        *(uint32_t*)(dst + kpad) = h2;
      }
      continue;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int x = xq * 4 + j;
      uint32_t v = (px >> (8 * j)) & 0xFFu;
      if (wrap) v = (256u - v) & 0xFFu;
      const int pxi = x / p, dx = x % p;
      const size_t prow = ((size_t)f * g + py) * g + pxi;
      const uint16_t h = T::from_f32((float)v);              // integers <= 255 are exact in bf16 and f16
      uint16_t* dst = patches + prow * (2 * (size_t)kpad) + (c * p + dy) * p + dx;
      dst[0] = h;
      dst[kpad] = h;
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(256) patches_f32_split_kernel(const float* __restrict__ pix, uint16_t* __restrict__ patches, int F, int R,
                                                                int p, int kpad) {
  const int g = R / p;
  const int quads_per_row = R >> 2;
  const size_t total = (size_t)F * 3 * R * quads_per_row;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int xq = (int)(i % quads_per_row);
    size_t t = i / quads_per_row;
    const int y = (int)(t % R);
    t /= R;
    const int c = (int)(t % 3);
    const int f = (int)(t / 3);
    const float4 px = *(const float4*)(pix + i * 4);
    const float v[4] = {px.x, px.y, px.z, px.w};
    const int py = y / p, dy = y % p;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int x = xq * 4 + j;
      const int pxi = x / p, dx = x % p;
      const size_t prow = ((size_t)f * g + py) * g + pxi;
      const uint16_t hi = T::from_f32(v[j]);
      const uint16_t lo = T::from_f32(v[j] - T::to_f32(hi));
      uint16_t* dst = patches + prow * (3 * (size_t)kpad) + (c * p + dy) * p + dx;
      dst[0] = hi;
      dst[kpad] = lo;
      dst[2 * kpad] = hi;
    }
  }
}

__global__ void zero_pad_cols_multi_kernel(uint16_t* __restrict__ patches, size_t rows, int k, int kpad, int parts) {
  const int padw = kpad - k;
  const size_t total = rows * padw * parts;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t row = i / ((size_t)padw * parts);
    const int rem = (int)(i % ((size_t)padw * parts));
    patches[row * ((size_t)kpad * parts) + (size_t)(rem / padw) * kpad + k + (rem % padw)] = 0;
  }
}

extern "C" int vmc_patches_u8_exact(const uint8_t* frames, void* patches, int F, int R, int p, int kpad, int wrap_quirk, int dtype16,
                                    void* stream) {
  if (!frames || !patches || F <= 0 || R <= 0 || p <= 0) return VMC_E_ARG;
  if (R % p || R % 4 || kpad < 3 * p * p || kpad % 8) return VMC_E_SHAPE;
  if ((uintptr_t)frames & 3) return VMC_E_ALIGN;
  const size_t total = (size_t)F * 3 * R * (R / 4);
  hipStream_t s = (hipStream_t)stream;
  const int k = 3 * p * p;
  if (kpad > k) {
    const size_t rows = (size_t)F * (R / p) * (R / p);
    hipLaunchKernelGGL(zero_pad_cols_multi_kernel, dim3(grid_for(rows * (kpad - k) * 2, 256)), dim3(256), 0, s, (uint16_t*)patches, rows, k, kpad, 2);
    VMC_CHECK_LAUNCH();
  }
  if (dtype16 == VMC_BF16)
    hipLaunchKernelGGL(patches_u8_exact_kernel<BF16>, dim3(grid_for(total, 256)), dim3(256), 0, s, frames, (uint16_t*)patches, F, R, p, kpad, wrap_quirk);
  else if (dtype16 == VMC_F16)
    hipLaunchKernelGGL(patches_u8_exact_kernel<F16>, dim3(grid_for(total, 256)), dim3(256), 0, s, frames, (uint16_t*)patches, F, R, p, kpad, wrap_quirk);
  else
    return VMC_E_DTYPE;
  VMC_CHECK_LAUNCH();
  return 0;
}

extern "C" int vmc_patches_f32_split(const float* pixel_values, void* patches, int F, int R, int p, int kpad, int dtype16, void* stream) {
  if (!pixel_values || !patches || F <= 0 || R <= 0 || p <= 0) return VMC_E_ARG;
  if (R % p || R % 4 || kpad < 3 * p * p || kpad % 8) return VMC_E_SHAPE;
  if ((uintptr_t)pixel_values & 15) return VMC_E_ALIGN;
  const size_t total = (size_t)F * 3 * R * (R / 4);
  hipStream_t s = (hipStream_t)stream;
  const int k = 3 * p * p;
  if (kpad > k) {
    const size_t rows = (size_t)F * (R / p) * (R / p);
    hipLaunchKernelGGL(zero_pad_cols_multi_kernel, dim3(grid_for(rows * (kpad - k) * 3, 256)), dim3(256), 0, s, (uint16_t*)patches, rows, k, kpad, 3);
    VMC_CHECK_LAUNCH();
  }
  if (dtype16 == VMC_BF16)
    hipLaunchKernelGGL(patches_f32_split_kernel<BF16>, dim3(grid_for(total, 256)), dim3(256), 0, s, pixel_values, (uint16_t*)patches, F, R, p, kpad);
  else if (dtype16 == VMC_F16)
    hipLaunchKernelGGL(patches_f32_split_kernel<F16>, dim3(grid_for(total, 256)), dim3(256), 0, s, pixel_values, (uint16_t*)patches, F, R, p, kpad);
  else
    return VMC_E_DTYPE;
  VMC_CHECK_LAUNCH();
  return 0;
}

// ---- Pillow-exact antialiased resample of planar u8 images along one axis (K0) --------------
// out[p, i, j] = clip8((2^21 + sum_x in[...] * coeff[o, x]) >> 22), the fixed-point arithmetic of Pillow's
// ImagingResample{Horizontal,Vertical}_8bpc; bounds/coeffs come from the host (float64, Pillow's precompute_coeffs).
// Only the output range [out_first, out_first + out_count) is produced (centre crop fused).
__global__ void __launch_bounds__(256) resample_u8_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                          const int* __restrict__ bounds, const int* __restrict__ coeffs, int planes,
                                                          int in_h, int in_w, int out_first, int out_count, int ksize, int horizontal,
                                                          int wrap) {
  const int oh = horizontal ? in_h : out_count, ow = horizontal ? out_count : in_w;
  const size_t total = (size_t)planes * oh * ow;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % ow), y = (int)((i / ow) % oh);
    const size_t p = i / ((size_t)ow * oh);
    const int o = out_first + (horizontal ? x : y);
    const int lo = bounds[2 * o], n = bounds[2 * o + 1];
    const int* k = coeffs + (size_t)o * ksize;
    const uint8_t* src = in + p * (size_t)in_h * in_w + (horizontal ? (size_t)y * in_w + lo : (size_t)lo * in_w + x);
    const int step = horizontal ? 1 : in_w;
    int acc = 1 << 21;
    for (int t = 0; t < n; ++t) {
      int v = src[(size_t)t * step];
      if (wrap) v = (256 - v) & 255;
      acc += v * k[t];
    }
    acc >>= 22;
    out[i] = (uint8_t)(acc < 0 ? 0 : (acc > 255 ? 255 : acc));
  }
}

extern "C" int vmc_resample_u8(const uint8_t* in, uint8_t* out, const int* bounds, const int* coeffs, int planes, int in_h, int in_w,
                               int out_first, int out_count, int ksize, int horizontal, int wrap_quirk, void* stream) {
  if (!in || !out || !bounds || !coeffs || planes <= 0 || in_h <= 0 || in_w <= 0 || out_count <= 0 || ksize <= 0 || out_first < 0)
    return VMC_E_ARG;
  const size_t total = (size_t)planes * (horizontal ? (size_t)in_h * out_count : (size_t)out_count * in_w);
  hipLaunchKernelGGL(resample_u8_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, in, out, bounds, coeffs, planes, in_h,
                     in_w, out_first, out_count, ksize, horizontal, wrap_quirk);
  VMC_CHECK_LAUNCH();
  return 0;
}

// ---- 16-bit transpose through a padded LDS tile --------------------------------------------------
__global__ void __launch_bounds__(256) transpose16_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int rows,
                                                          int cols, size_t ld_in, size_t ld_out) {
  __shared__ uint16_t tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? in[(size_t)r * ld_in + c] : (uint16_t)0;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows) out[(size_t)c * ld_out + r] = tile[tx][i];
  }
}

extern "C" int vmc_transpose16(const void* in, void* out, int rows, int cols, int ld_in, int ld_out, void* stream) {
  if (!in || !out || rows <= 0 || cols <= 0 || ld_in < cols || ld_out < rows) return VMC_E_ARG;
  dim3 grid((cols + 63) / 64, (rows + 63) / 64);
  hipLaunchKernelGGL(transpose16_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)in, (uint16_t*)out, rows, cols,
                     (size_t)ld_in, (size_t)ld_out);
  VMC_CHECK_LAUNCH();
  return 0;
}

// ---- f32 weight -> 16-bit (+ transposed copy) ----------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) cast_weight_kernel(const float* __restrict__ w, uint16_t* __restrict__ w16,
                                                          uint16_t* __restrict__ w16t, int rows, int cols, size_t ld_out,
                                                          size_t ld_out_t) {
  __shared__ uint16_t tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    uint16_t v = 0;
    if (r < rows && c < cols) {
      v = T::from_f32(w[(size_t)r * cols + c]);
      if (w16) w16[(size_t)r * ld_out + c] = v;
    }
    tile[i][tx] = v;
  }
  if (!w16t) return;
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows) w16t[(size_t)c * ld_out_t + r] = tile[tx][i];
  }
}

extern "C" int vmc_cast_weight(const float* w, void* w16, void* w16_t, int rows, int cols, int ld_out, int ld_out_t, int dtype16,
                               void* stream) {
  if (!w || (!w16 && !w16_t) || rows <= 0 || cols <= 0) return VMC_E_ARG;
  if ((w16 && ld_out < cols) || (w16_t && ld_out_t < rows)) return VMC_E_ARG;
  dim3 grid((cols + 63) / 64, (rows + 63) / 64);
  if (dtype16 == VMC_BF16)
    hipLaunchKernelGGL(cast_weight_kernel<BF16>, grid, dim3(256), 0, (hipStream_t)stream, w, (uint16_t*)w16, (uint16_t*)w16_t, rows,
                       cols, (size_t)ld_out, (size_t)ld_out_t);
  else if (dtype16 == VMC_F16)
    hipLaunchKernelGGL(cast_weight_kernel<F16>, grid, dim3(256), 0, (hipStream_t)stream, w, (uint16_t*)w16, (uint16_t*)w16_t, rows,
                       cols, (size_t)ld_out, (size_t)ld_out_t);
  else
    return VMC_E_DTYPE;
  VMC_CHECK_LAUNCH();
  return 0;
}

// ---- the same for MANY weights in one launch (after an optimiser step: every cached compute copy of the trained parameters) ----
// desc[i] = {w, w16, w16t, rows, cols, ld, ldt, tile0, tiles_x}: tensor i owns tiles [tile0, tile0 of i+1) of the 1-D grid.
struct CastDesc {
  const float* w;
  uint16_t* w16;
  uint16_t* w16t;
  int rows, cols, ld, ldt, tile0, tiles_x;
};
static_assert(sizeof(CastDesc) == 48, "vmc_cast_weights_multi descriptor layout (include/vmc.h)");

template <typename T>
__global__ void __launch_bounds__(256) cast_weights_multi_kernel(const CastDesc* __restrict__ desc, int n) {
  __shared__ uint16_t tile[64][68];             // 136-B rows: 8-byte aligned row starts for the 4-element writes
  int lo = 0, hi = n - 1;                       // last descriptor whose tile0 <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (desc[mid].tile0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const CastDesc d = desc[lo];
  const int t = blockIdx.x - d.tile0;
  const int r0 = (t / d.tiles_x) * 64, c0 = (t % d.tiles_x) * 64;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;      // 16 x 16 threads, 4 elements each per pass, 4 passes
  // vector path: 16-byte loads of the master, 8-byte stores of both copies (all of a tensor's rows / strides must allow it)
  const bool vec = (((uintptr_t)d.w & 15) == 0) && (d.cols & 3) == 0 && (d.rows & 3) == 0 && (!d.w16 || ((d.ld & 3) == 0 && ((uintptr_t)d.w16 & 7) == 0)) &&
                   (!d.w16t || ((d.ldt & 3) == 0 && ((uintptr_t)d.w16t & 7) == 0));
  if (vec) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = r0 + ty + 16 * i, c = c0 + 4 * tx;
      uint2 pk = make_uint2(0u, 0u);
      if (r < d.rows && c < d.cols) {
        const float4 v = *(const float4*)(d.w + (size_t)r * d.cols + c);
        pk = make_uint2((uint32_t)T::from_f32(v.x) | ((uint32_t)T::from_f32(v.y) << 16), (uint32_t)T::from_f32(v.z) | ((uint32_t)T::from_f32(v.w) << 16));
        if (d.w16) *(uint2*)(d.w16 + (size_t)r * d.ld + c) = pk;
      }
      *(uint2*)&tile[ty + 16 * i][4 * tx] = pk;
    }
    if (!d.w16t) return;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = c0 + ty + 16 * i, r = r0 + 4 * tx;          // output row c of the transposed copy, 4 consecutive r
      if (c < d.cols && r < d.rows) {
        const int cl = ty + 16 * i;
        *(uint2*)(d.w16t + (size_t)c * d.ldt + r) = make_uint2((uint32_t)tile[4 * tx][cl] | ((uint32_t)tile[4 * tx + 1][cl] << 16),
                                                              (uint32_t)tile[4 * tx + 2][cl] | ((uint32_t)tile[4 * tx + 3][cl] << 16));
      }
    }
    return;
  }
  const int sx = threadIdx.x & 63, sy = threadIdx.x >> 6;
  for (int i = sy; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + sx;
    uint16_t v = 0;
    if (r < d.rows && c < d.cols) {
      v = T::from_f32(d.w[(size_t)r * d.cols + c]);
      if (d.w16) d.w16[(size_t)r * d.ld + c] = v;
    }
    tile[i][sx] = v;
  }
  if (!d.w16t) return;
  __syncthreads();
  for (int i = sy; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + sx;
    if (c < d.cols && r < d.rows) d.w16t[(size_t)c * d.ldt + r] = tile[sx][i];
  }
}

// ---- AdamW update + refresh of the 16-bit copies in ONE pass over the masters (captured training steps) -----------------------
// The optimiser step followed by vmc_cast_weights_multi reads every matrix twice (28 B per parameter for the update, 4 + 4 for
// the copies); here a 64 x 64 tile of a matrix is updated in registers (adam_element: the expression of adam_kernel, same
// bits) and leaves as p, m, v AND both 16-bit copies -- 28 + 4 B per parameter, one launch less.  The descriptors are those of
// vmc_cast_weights_multi (w points into the flat parameter arena p_base; g / m / v are the parallel arenas); parameters without
// compute copies (biases, LayerNorm, ...) are covered by `ranges`.
template <typename T>
__global__ void __launch_bounds__(256) adam_cast_multi_kernel(const CastDesc* __restrict__ desc, int n, const float* p_base,
                                                              const float* __restrict__ g_base, float* __restrict__ m_base,
                                                              float* __restrict__ v_base, const float* __restrict__ hyper, float b1, float b2,
                                                              float eps, float wd, int decoupled) {
  __shared__ uint16_t tile[64][68];
  const float lr = hyper[0], step_size = hyper[1], inv_sqrt_bc2 = hyper[2], gscale = hyper[3];
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (desc[mid].tile0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const CastDesc d = desc[lo];
  const int t = blockIdx.x - d.tile0;
  const int r0 = (t / d.tiles_x) * 64, c0 = (t % d.tiles_x) * 64;
  float* const w = const_cast<float*>(d.w);
  const size_t base = (size_t)(d.w - p_base);
  const float* const g = g_base + base;
  float* const m = m_base + base;
  float* const v = v_base + base;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const bool vec = (d.cols & 3) == 0 && (d.rows & 3) == 0 && (!d.w16 || ((d.ld & 3) == 0 && ((uintptr_t)d.w16 & 7) == 0)) &&
                   (!d.w16t || ((d.ldt & 3) == 0 && ((uintptr_t)d.w16t & 7) == 0));       // arena slots are 256-byte aligned
  if (vec) {
    float4 pp[4], gg[4], mm[4], vv[4];
    bool in[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {       // all loads of the tile first: 16 float4 in flight per thread
      const int r = r0 + ty + 16 * i, c = c0 + 4 * tx;
      in[i] = r < d.rows && c < d.cols;
      if (in[i]) {
        const size_t e = (size_t)r * d.cols + c;
        pp[i] = *(const float4*)(w + e); gg[i] = *(const float4*)(g + e); mm[i] = *(const float4*)(m + e); vv[i] = *(const float4*)(v + e);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = r0 + ty + 16 * i, c = c0 + 4 * tx;
      uint2 pk = make_uint2(0u, 0u);
      if (in[i]) {
        float* P = &pp[i].x; const float* G = &gg[i].x; float* M = &mm[i].x; float* V = &vv[i].x;
#pragma unroll
        for (int j = 0; j < 4; ++j) adam_element(P[j], G[j], M[j], V[j], lr, b1, b2, eps, wd, decoupled, step_size, inv_sqrt_bc2, gscale);
        const size_t e = (size_t)r * d.cols + c;
        *(float4*)(w + e) = pp[i]; *(float4*)(m + e) = mm[i]; *(float4*)(v + e) = vv[i];
        pk = make_uint2((uint32_t)T::from_f32(P[0]) | ((uint32_t)T::from_f32(P[1]) << 16), (uint32_t)T::from_f32(P[2]) | ((uint32_t)T::from_f32(P[3]) << 16));
        if (d.w16) *(uint2*)(d.w16 + (size_t)r * d.ld + c) = pk;
      }
      *(uint2*)&tile[ty + 16 * i][4 * tx] = pk;
    }
    if (!d.w16t) return;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = c0 + ty + 16 * i, r = r0 + 4 * tx;
      if (c < d.cols && r < d.rows) {
        const int cl = ty + 16 * i;
        *(uint2*)(d.w16t + (size_t)c * d.ldt + r) = make_uint2((uint32_t)tile[4 * tx][cl] | ((uint32_t)tile[4 * tx + 1][cl] << 16),
                                                              (uint32_t)tile[4 * tx + 2][cl] | ((uint32_t)tile[4 * tx + 3][cl] << 16));
      }
    }
    return;
  }
  const int sx = threadIdx.x & 63, sy = threadIdx.x >> 6;
  for (int i = sy; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + sx;
    uint16_t h = 0;
    if (r < d.rows && c < d.cols) {
      const size_t e = (size_t)r * d.cols + c;
      float pk = w[e], mk = m[e], vk = v[e];
      adam_element(pk, g[e], mk, vk, lr, b1, b2, eps, wd, decoupled, step_size, inv_sqrt_bc2, gscale);
      w[e] = pk; m[e] = mk; v[e] = vk;
      h = T::from_f32(pk);
      if (d.w16) d.w16[(size_t)r * d.ld + c] = h;
    }
    tile[i][sx] = h;
  }
  if (!d.w16t) return;
  __syncthreads();
  for (int i = sy; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + sx;
    if (c < d.cols && r < d.rows) d.w16t[(size_t)c * d.ldt + r] = tile[sx][i];
  }
}

// the parameters without compute copies: ranges[i] = {offset into the arenas, elements, first block}; 1024 elements per block
struct AdamRange {
  unsigned long long off;
  int n, block0;
};
static_assert(sizeof(AdamRange) == 16, "vmc_adam_cast_multi range layout (include/vmc.h)");
__global__ void __launch_bounds__(256) adam_ranges_kernel(const AdamRange* __restrict__ rr, int n, float* __restrict__ p_base,
                                                          const float* __restrict__ g_base, float* __restrict__ m_base,
                                                          float* __restrict__ v_base, const float* __restrict__ hyper, float b1, float b2, float eps,
                                                          float wd, int decoupled) {
  const float lr = hyper[0], step_size = hyper[1], inv_sqrt_bc2 = hyper[2], gscale = hyper[3];
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (rr[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const AdamRange r = rr[lo];
  const int i0 = ((int)blockIdx.x - r.block0) * 1024 + 4 * threadIdx.x;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = i0 + j;
    if (i < r.n) {
      const size_t e = (size_t)r.off + i;
      float pk = p_base[e], mk = m_base[e], vk = v_base[e];
      adam_element(pk, g_base[e], mk, vk, lr, b1, b2, eps, wd, decoupled, step_size, inv_sqrt_bc2, gscale);
      p_base[e] = pk; m_base[e] = mk; v_base[e] = vk;
    }
  }
}

extern "C" int vmc_adam_cast_multi(const void* desc, int n_desc, int total_tiles, const void* ranges, int n_ranges, int total_range_blocks,
                                   float* p_base, const float* g_base, float* m_base, float* v_base, const float* hyper, float beta1,
                                   float beta2, float eps, float weight_decay, int decoupled_wd, int dtype16, void* stream) {
  if (!p_base || !g_base || !m_base || !v_base || !hyper) return VMC_E_ARG;
  if ((n_desc > 0) != (desc != nullptr && total_tiles > 0) || (n_ranges > 0) != (ranges != nullptr && total_range_blocks > 0)) return VMC_E_ARG;
  if (((uintptr_t)p_base | (uintptr_t)g_base | (uintptr_t)m_base | (uintptr_t)v_base | (uintptr_t)hyper) & 15) return VMC_E_ALIGN;
  if (dtype16 != VMC_BF16 && dtype16 != VMC_F16) return VMC_E_DTYPE;
  if (n_desc > 0) {
    if (dtype16 == VMC_BF16)
      hipLaunchKernelGGL(adam_cast_multi_kernel<BF16>, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, (const CastDesc*)desc, n_desc, p_base,
                         g_base, m_base, v_base, hyper, beta1, beta2, eps, weight_decay, decoupled_wd);
    else
      hipLaunchKernelGGL(adam_cast_multi_kernel<F16>, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, (const CastDesc*)desc, n_desc, p_base,
                         g_base, m_base, v_base, hyper, beta1, beta2, eps, weight_decay, decoupled_wd);
    VMC_CHECK_LAUNCH();
  }
  if (n_ranges > 0) {
    hipLaunchKernelGGL(adam_ranges_kernel, dim3(total_range_blocks), dim3(256), 0, (hipStream_t)stream, (const AdamRange*)ranges, n_ranges, p_base,
                       g_base, m_base, v_base, hyper, beta1, beta2, eps, weight_decay, decoupled_wd);
    VMC_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int vmc_cast_weights_multi(const void* desc, int n_desc, int total_tiles, int dtype16, void* stream) {
  if (!desc || n_desc <= 0 || total_tiles <= 0) return VMC_E_ARG;
  if (dtype16 == VMC_BF16)
    hipLaunchKernelGGL(cast_weights_multi_kernel<BF16>, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, (const CastDesc*)desc, n_desc);
  else if (dtype16 == VMC_F16)
    hipLaunchKernelGGL(cast_weights_multi_kernel<F16>, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, (const CastDesc*)desc, n_desc);
  else
    return VMC_E_DTYPE;
  VMC_CHECK_LAUNCH();
  return 0;
}

// ---- column sums (bias gradients): partials per row-slab, then a reduce ---------------------------
#define COLSUM_SLABS 256
// partial[slab, n] = sum of rows [slab*rp, (slab+1)*rp): 64 lanes x 4 columns (8-/16-byte loads) across, 4 row
// phases per block combined through LDS
template <typename T>
__global__ void __launch_bounds__(256) colsum_partial_kernel(const void* __restrict__ in, float* __restrict__ partial, int M, int N,
                                                             size_t ld, int in_f32) {
  __shared__ float4 sm[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int col = (blockIdx.x * 64 + tx) * 4;
  const int rows_per = (M + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col < N) {
    if (in_f32) {
      for (int r = r0 + ty; r < r1; r += 4) {
        const float4 v = *(const float4*)((const float*)in + (size_t)r * ld + col);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      }
    } else {
      for (int r = r0 + ty; r < r1; r += 4) {
        const uint2 w = *(const uint2*)((const uint16_t*)in + (size_t)r * ld + col);
        float x0, x1, x2, x3;
        unpack2<T>(w.x, x0, x1);
        unpack2<T>(w.y, x2, x3);
        a.x += x0; a.y += x1; a.z += x2; a.w += x3;
      }
    }
  }
  sm[ty][tx] = a;
  __syncthreads();
  if (ty == 0 && col < N) {
    float4 o;
    o.x = (sm[0][tx].x + sm[1][tx].x) + (sm[2][tx].x + sm[3][tx].x);
    o.y = (sm[0][tx].y + sm[1][tx].y) + (sm[2][tx].y + sm[3][tx].y);
    o.z = (sm[0][tx].z + sm[1][tx].z) + (sm[2][tx].z + sm[3][tx].z);
    o.w = (sm[0][tx].w + sm[1][tx].w) + (sm[2][tx].w + sm[3][tx].w);
    *(float4*)(partial + (size_t)blockIdx.y * N + col) = o;
  }
}
__global__ void __launch_bounds__(256) reduce_partials_kernel(const float* __restrict__ partial, float* __restrict__ out, int P, int W) {
  __shared__ float sm[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + tx;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (col < W) {
    int p = ty;
    for (; p + 12 < P; p += 16) {   // four independent loads in flight
      a0 += partial[(size_t)p * W + col];
      a1 += partial[(size_t)(p + 4) * W + col];
      a2 += partial[(size_t)(p + 8) * W + col];
      a3 += partial[(size_t)(p + 12) * W + col];
    }
    for (; p < P; p += 4) a0 += partial[(size_t)p * W + col];
  }
  sm[ty][tx] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (ty == 0 && col < W) out[col] = (sm[0][tx] + sm[1][tx]) + (sm[2][tx] + sm[3][tx]);
}
// enough slabs to fill the chip (N/256 column blocks x slabs), few enough that the second pass stays short
static int colsum_slabs(int M) { int s = M / 128; return s < 1 ? 1 : (s > 64 ? 64 : s); }
extern "C" size_t vmc_colsum_workspace_bytes(int M, int N) { return (size_t)colsum_slabs(M) * N * sizeof(float); }
extern "C" int vmc_colsum(const void* in, float* out, int M, int N, int ld_in, int in_dtype, void* workspace, size_t workspace_bytes,
                          void* stream) {
  if (!in || !out || !workspace || M <= 0 || N <= 0 || ld_in < N) return VMC_E_ARG;
  if (workspace_bytes < vmc_colsum_workspace_bytes(M, N)) return VMC_E_ARG;
  if ((N % 4) || (ld_in % 4)) return VMC_E_ALIGN;
  const int slabs = colsum_slabs(M);
  dim3 grid((N / 4 + 63) / 64, slabs);
  hipStream_t s = (hipStream_t)stream;
  if (in_dtype == VMC_F16)
    hipLaunchKernelGGL(colsum_partial_kernel<F16>, grid, dim3(256), 0, s, in, (float*)workspace, M, N, (size_t)ld_in, 0);
  else
    hipLaunchKernelGGL(colsum_partial_kernel<BF16>, grid, dim3(256), 0, s, in, (float*)workspace, M, N, (size_t)ld_in, in_dtype == VMC_F32);
  VMC_CHECK_LAUNCH();
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((N + 63) / 64), dim3(256), 0, s, (const float*)workspace, out, slabs, N);
  VMC_CHECK_LAUNCH();
  return 0;
}

// ---- class-token rows ----------------------------------------------------------------------------
template <typename T>
__global__ void set_class_rows_kernel(void* __restrict__ x, const float* __restrict__ a, const float* __restrict__ b, int F, int D,
                                      size_t row_stride, int x_f32) {
  const size_t total = (size_t)F * D;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % D);
    const size_t f = i / D;
    const float v = a[d] + b[d];
    if (x_f32) ((float*)x)[f * row_stride + d] = v;
    else ((uint16_t*)x)[f * row_stride + d] = T::from_f32(v);
  }
}
extern "C" int vmc_set_class_rows(void* x, const float* a, const float* b, int F, int D, size_t row_stride, int x_dtype, int dtype16,
                                  void* stream) {
  if (!x || !a || !b || F <= 0 || D <= 0) return VMC_E_ARG;
  const int grid = grid_for((size_t)F * D, 256);
  if (dtype16 == VMC_F16)
    hipLaunchKernelGGL(set_class_rows_kernel<F16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, a, b, F, D, row_stride, x_dtype == VMC_F32);
  else
    hipLaunchKernelGGL(set_class_rows_kernel<BF16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, a, b, F, D, row_stride, x_dtype == VMC_F32);
  VMC_CHECK_LAUNCH();
  return 0;
}

// ---- activations on flat 16-bit arrays (training path) ---------------------------------------------
template <typename T>
__global__ void act_fwd_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, size_t n, int act) {
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += (size_t)gridDim.x * blockDim.x * 8) {
    if (i + 8 <= n) {
      const uint4 w = *(const uint4*)(x + i);
      const uint32_t in[4] = {w.x, w.y, w.z, w.w};
      uint32_t o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float a, b;
        unpack2<T>(in[j], a, b);
        o[j] = pack2<T>(apply_act_rt(a, act), apply_act_rt(b, act));
      }
      *(uint4*)(y + i) = make_uint4(o[0], o[1], o[2], o[3]);
    } else {
      for (size_t k = i; k < n; ++k) y[k] = T::from_f32(apply_act_rt(T::to_f32(x[k]), act));
    }
  }
}
template <typename T>
__global__ void act_bwd_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy, uint16_t* __restrict__ dx, size_t n,
                               int act) {
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += (size_t)gridDim.x * blockDim.x * 8) {
    if (i + 8 <= n) {
      const uint4 w = *(const uint4*)(x + i);
      const uint4 g = *(const uint4*)(dy + i);
      const uint32_t in[4] = {w.x, w.y, w.z, w.w};
      const uint32_t gi[4] = {g.x, g.y, g.z, g.w};
      uint32_t o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float a, b, ga, gb;
        unpack2<T>(in[j], a, b);
        unpack2<T>(gi[j], ga, gb);
        o[j] = pack2<T>(ga * act_grad_rt(a, act), gb * act_grad_rt(b, act));
      }
      *(uint4*)(dx + i) = make_uint4(o[0], o[1], o[2], o[3]);
    } else {
      for (size_t k = i; k < n; ++k) dx[k] = T::from_f32(T::to_f32(dy[k]) * act_grad_rt(T::to_f32(x[k]), act));
    }
  }
}
extern "C" int vmc_act_fwd(const void* x, void* y, size_t n, int act, int dtype16, void* stream) {
  if (!x || !y || n == 0) return VMC_E_ARG;
  if (((uintptr_t)x | (uintptr_t)y) & 15) return VMC_E_ALIGN;
  const int grid = grid_for((n + 7) / 8, 256);
  if (dtype16 == VMC_F16) hipLaunchKernelGGL(act_fwd_kernel<F16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (uint16_t*)y, n, act);
  else if (dtype16 == VMC_BF16) hipLaunchKernelGGL(act_fwd_kernel<BF16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (uint16_t*)y, n, act);
  else return VMC_E_DTYPE;
  VMC_CHECK_LAUNCH();
  return 0;
}
extern "C" int vmc_act_bwd(const void* x, const void* dy, void* dx, size_t n, int act, int dtype16, void* stream) {
  if (!x || !dy || !dx || n == 0) return VMC_E_ARG;
  if (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15) return VMC_E_ALIGN;
  const int grid = grid_for((n + 7) / 8, 256);
  if (dtype16 == VMC_F16) hipLaunchKernelGGL(act_bwd_kernel<F16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (const uint16_t*)dy, (uint16_t*)dx, n, act);
  else if (dtype16 == VMC_BF16) hipLaunchKernelGGL(act_bwd_kernel<BF16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (const uint16_t*)dy, (uint16_t*)dx, n, act);
  else return VMC_E_DTYPE;
  VMC_CHECK_LAUNCH();
  return 0;
}

// ---- mean over T ----------------------------------------------------------------------------------
template <typename T>
__global__ void mean_pool_kernel(const void* __restrict__ x, uint16_t* __restrict__ o16, float* __restrict__ o32, int B, int Tn, int D,
                                 int x_f32) {
  const size_t total = (size_t)B * D;
  const float inv = 1.0f / (float)Tn;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % D);
    const size_t b = i / D;
    float a = 0.f;
    for (int t = 0; t < Tn; ++t) {
      const size_t idx = (b * Tn + t) * D + d;
      a += x_f32 ? ((const float*)x)[idx] : T::to_f32(((const uint16_t*)x)[idx]);
    }
    a *= inv;
    if (o16) o16[i] = T::from_f32(a);
    if (o32) o32[i] = a;
  }
}
extern "C" int vmc_mean_pool(const void* x, void* out16, float* out32, int B, int Tn, int D, int x_dtype, int dtype16, void* stream) {
  if (!x || (!out16 && !out32) || B <= 0 || Tn <= 0 || D <= 0) return VMC_E_ARG;
  const int grid = grid_for((size_t)B * D, 256);
  if (dtype16 == VMC_F16) hipLaunchKernelGGL(mean_pool_kernel<F16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (uint16_t*)out16, out32, B, Tn, D, x_dtype == VMC_F32);
  else if (dtype16 == VMC_BF16) hipLaunchKernelGGL(mean_pool_kernel<BF16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (uint16_t*)out16, out32, B, Tn, D, x_dtype == VMC_F32);
  else return VMC_E_DTYPE;
  VMC_CHECK_LAUNCH();
  return 0;
}

// ---- sinusoidal positional encoding (TFAM/models/AMO_CLIP.py:88-97), added in place ---------------
__global__ void add_pe_kernel(float* __restrict__ x, int B, int Tn, int D) {
  const size_t total = (size_t)B * Tn * D;
  const float c = -9.210340371976184f / (float)D;  // -ln(10000)/D
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % D);
    const int t = (int)((i / D) % Tn);
    const float div = expf((float)(d & ~1) * c);
    const float ang = (float)t * div;
    x[i] += (d & 1) ? cosf(ang) : sinf(ang);
  }
}
extern "C" int vmc_add_sinusoidal_pe(float* x, int B, int Tn, int D, void* stream) {
  if (!x || B <= 0 || Tn <= 0 || D <= 0 || (D & 1)) return VMC_E_ARG;
  hipLaunchKernelGGL(add_pe_kernel, dim3(grid_for((size_t)B * Tn * D, 256)), dim3(256), 0, (hipStream_t)stream, x, B, Tn, D);
  VMC_CHECK_LAUNCH();
  return 0;
}

// ---- y = alpha*a + beta*b, f32 ------------------------------------------------------------------------
__global__ void axpby_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, size_t n, float alpha,
                             float beta) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    y[i] = alpha * a[i] + beta * b[i];
}
extern "C" int vmc_axpby_f32(const float* a, const float* b, float* y, size_t n, float alpha, float beta, void* stream) {
  if (!a || !b || !y || n == 0) return VMC_E_ARG;
  hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, y, n, alpha, beta);
  VMC_CHECK_LAUNCH();
  return 0;
}

// ---- flat casts ------------------------------------------------------------------------------------
template <typename T>
__global__ void cast_to16_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, size_t n) {
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * blockDim.x * 4) {
    if (i + 4 <= n) {
      const float4 v = *(const float4*)(x + i);
      *(uint2*)(y + i) = make_uint2(pack2<T>(v.x, v.y), pack2<T>(v.z, v.w));
    } else {
      for (size_t k = i; k < n; ++k) y[k] = T::from_f32(x[k]);
    }
  }
}
template <typename T>
__global__ void cast_to32_kernel(const uint16_t* __restrict__ x, float* __restrict__ y, size_t n) {
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * blockDim.x * 4) {
    if (i + 4 <= n) {
      const uint2 w = *(const uint2*)(x + i);
      float4 v;
      unpack2<T>(w.x, v.x, v.y);
      unpack2<T>(w.y, v.z, v.w);
      *(float4*)(y + i) = v;
    } else {
      for (size_t k = i; k < n; ++k) y[k] = T::to_f32(x[k]);
    }
  }
}
extern "C" int vmc_cast_f32_to_16(const float* x, void* y, size_t n, int dtype16, void* stream) {
  if (!x || !y || n == 0) return VMC_E_ARG;
  if (((uintptr_t)x & 15) | ((uintptr_t)y & 7)) return VMC_E_ALIGN;
  const int grid = grid_for((n + 3) / 4, 256);
  if (dtype16 == VMC_F16) hipLaunchKernelGGL(cast_to16_kernel<F16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (uint16_t*)y, n);
  else if (dtype16 == VMC_BF16) hipLaunchKernelGGL(cast_to16_kernel<BF16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (uint16_t*)y, n);
  else return VMC_E_DTYPE;
  VMC_CHECK_LAUNCH();
  return 0;
}
extern "C" int vmc_cast_16_to_f32(const void* x, float* y, size_t n, int dtype16, void* stream) {
  if (!x || !y || n == 0) return VMC_E_ARG;
  if (((uintptr_t)y & 15) | ((uintptr_t)x & 7)) return VMC_E_ALIGN;
  const int grid = grid_for((n + 3) / 4, 256);
  if (dtype16 == VMC_F16) hipLaunchKernelGGL(cast_to32_kernel<F16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, y, n);
  else if (dtype16 == VMC_BF16) hipLaunchKernelGGL(cast_to32_kernel<BF16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, y, n);
  else return VMC_E_DTYPE;
  VMC_CHECK_LAUNCH();
  return 0;
}
