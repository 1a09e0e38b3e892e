// Pure index arithmetic of the MFMA tiles (LDS images, swizzles, fragment -> element maps).
// Shared by the kernels and by tests/host/test_tile_index.cpp, which emulates the lanes of a wave on
// the CPU (LDS-DMA placement rule + the MFMA operand/accumulator lane maps) and checks every kernel's
// index math against a plain matrix product before anything runs on a GPU.
//
// Hardware facts used (cdna_hip_programming.md §3, §5; MI355X_MICROARCH.md §LDS):
//   v_mfma_f32_16x16x32_{bf16,f16}: lane l holds A[row l&15][k = 8(l>>4)+j], B[k = 8(l>>4)+j][col l&15],
//   j = 0..7; accumulator reg j of lane l is D[row 4(l>>4)+j][col l&15].
//   global_load_lds_dwordx4: lane i of a wave writes LDS bytes [base + 16 i, +16) (lane-linear image),
//   the global source address is per lane -> swizzles are applied to the SOURCE and to the READ.
//   ds_read_b128 is serviced in four 16-lane groups; bank = (addr/4) % 64.
#pragma once

#if defined(__HIPCC__)
#define VMC_HD __host__ __device__ inline
#else
#define VMC_HD inline
#endif

// LDS tile image: rows of 64 16-bit elements = 128 B = 8 chunks of 16 B.  Physical chunk = logical
// chunk XOR f(row); f chosen so that every ds_read_b128 lane group touches 16 distinct 16-B slots.
// X-operand rows are read as 16 consecutive rows per instruction:
VMC_HD int swz_x(int row) { return (row >> 1) & 7; }
// W-operand rows are read as rows {16*(r>>2) + 4*nt + (r&3)} (r = lane&15) so that each lane ends up
// with 16 consecutive output columns; distinct slots need row bits {1,4,5}:
VMC_HD int swz_w(int row) { return ((row >> 1) & 1) | (((row >> 4) & 3) << 1); }

VMC_HD int lds_off_x(int row, int chunk) { return row * 128 + ((chunk ^ swz_x(row)) << 4); }
VMC_HD int lds_off_w(int row, int chunk) { return row * 128 + ((chunk ^ swz_w(row)) << 4); }

// GEMM wave tile: (16*MT) rows x 64 cols.  Lane (r = lane&15, q = lane>>4):
//   X fragment (mt, kk): tile row 16*mt + r, logical chunk 4*kk + q
//   W fragment (nt, kk): tile row 16*(r>>2) + 4*nt + (r&3), logical chunk 4*kk + q
//   acc[mt][nt][j] = C[row 16*mt + r][col 16*q + 4*nt + j]      (operands passed as mfma(W, X))
VMC_HD int gemm_w_row(int r, int nt) { return 16 * (r >> 2) + 4 * nt + (r & 3); }
VMC_HD int gemm_c_col(int q, int nt, int j) { return 16 * q + 4 * nt + j; }

// ---- 8-phase GEMM (256x256x64 tile as 4 half-tile slots of 128 rows per K tile) ------------------
// Wave (wm, wn) of the 2x4 wave grid owns rows {128*mh + 64*wm + [0,64)} and cols {128*nh + 32*wn + [0,32)},
// mh, nh in {0,1}: 64 rows of EACH A half-tile and 32 columns of EACH B half-tile, so every wave needs
// half-tile A0/B0 in phase 0, B1 in phase 1, A1 in phase 2 (uniform deadlines for the LDS-DMA pipeline).
//   X fragment (mh, mt, kk): slot A_mh, row 64*wm + 16*mt + r            (mt = 0..3)
//   W fragment (nh, nt, kk): slot B_nh, row 32*wn + g8_w_row(r, nt)      (nt = 0..1)
//   acc[mh][mt][nh][nt][j] = C[128*mh + 64*wm + 16*mt + r][128*nh + 32*wn + 8*q + 4*nt + j]
VMC_HD int g8_w_row(int r, int nt) { return 8 * (r >> 2) + 4 * nt + (r & 3); }
VMC_HD int swz_w8(int row) { return ((row >> 1) & 1) | (((row >> 3) & 3) << 1); }
VMC_HD int lds_off_w8(int row, int chunk) { return row * 128 + ((chunk ^ swz_w8(row)) << 4); }
VMC_HD void stage_src_w8(int idx, int& row, int& chunk) { row = idx >> 3; chunk = (idx & 7) ^ swz_w8(row); }

// The same tile on v_mfma_f32_32x32x16 (lane l: A[row l&31][k = 8(l>>5)+j], B[k = 8(l>>5)+j][col l&31]; accumulator reg i of lane l is
// D[row 8(i>>2) + 4(l>>5) + (i&3)][col l&31]).  Lane (x = lane&31, h = lane>>5), operands passed as mfma(W, X):
//   X fragment (mh, mt, kk): slot A_mh, row 64*wm + 32*mt + x, logical chunk 2*kk + h      (mt = 0..1, kk = 0..3)
//   W fragment (nh, kk):     slot B_nh, row 32*wn + g8_w_row32(x), logical chunk 2*kk + h
//   acc[mh][nh][mt][i] = C[128*mh + 64*wm + 32*mt + x][128*nh + 32*wn + g8_c_col32(h, i)]
// g8_w_row32 is chosen so that registers 0..7 / 8..15 of a lane are 8 consecutive output columns each and the two lane halves
// interleave: one 16-byte store per register octet, 32 contiguous bytes per row and instruction.  With swz_x (A) and swz_w8 (B)
// every 16-lane group of the ds_read_b128 touches 16 distinct 16-B slots (tests/host/test_tile_index.cpp).
VMC_HD int g8_w_row32(int w) { return 16 * (w >> 4) + 8 * ((w >> 2) & 1) + 4 * ((w >> 3) & 1) + (w & 3); }
VMC_HD int g8_c_col32(int h, int i) { return 16 * (i >> 3) + 8 * h + (i & 7); }

// Staging: 16-B chunk `idx` (LDS order) of an operand tile -> (row, logical chunk) to fetch.
VMC_HD void stage_src_x(int idx, int& row, int& chunk) { row = idx >> 3; chunk = (idx & 7) ^ swz_x(row); }
VMC_HD void stage_src_w(int idx, int& row, int& chunk) { row = idx >> 3; chunk = (idx & 7) ^ swz_w(row); }

// ---- TN weight gradient, 256 x 256 tile (gemm_tn256.hip) -----------------------------------------
// Half-tile image: [64 tokens][128 columns] = 256-B rows of sixteen 16-B chunks; physical chunk slot = logical chunk ^ tn_swz(token row).
// ds_read_b64_tr_b16 (lane i of 16-lane group G gets element i&3 of the 8-byte pieces addressed by lanes 16G + 4e + (i>>2), e = 0..3):
// lane (r = lane&15, g = lane>>4), q = r>>2, p = r&3 addresses token row 8g + q + 4e (e = which half of the lane's 8 tokens), columns
// 16*tile + 4p .. +3 of its wave's window, and receives tokens 8g + 4e + 0..3 of column 16*tile + r.  tile only flips chunk bits 1-2:
// address(tile) = address(0) ^ (tile << 5); the second 32 tokens of a stage are +8192 bytes.
//   A (dY) fragment (mh, mt, kk): slot A_mh, window = columns 64*wm + [0,64);  B (X) fragment (nh, nt, kk): slot B_nh, columns 32*wn + [0,32)
//   acc[mh][nh][mt][nt][j] = C[128*mh + 64*wm + 16*mt + r][128*nh + 32*wn + 16*nt + 4*g + j]      (operands passed as mfma(X, dY))
VMC_HD int tn_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
VMC_HD void tn_stage_src(int idx, int& row, int& chunk) { row = idx >> 4; chunk = (idx & 15) ^ tn_swz(row); }
VMC_HD int tn256_a_off(int wm, int lane, int e) {
  const int r = lane & 15, g = lane >> 4, q = r >> 2, p = r & 3, row = 8 * g + q + 4 * e;
  return row * 256 + (((8 * wm + (p >> 1)) ^ tn_swz(row)) << 4) + (p & 1) * 8;
}
VMC_HD int tn256_b_off(int wn, int lane, int e) {
  const int r = lane & 15, g = lane >> 4, q = r >> 2, p = r & 3, row = 8 * g + q + 4 * e;
  return row * 256 + (((4 * wn + (p >> 1)) ^ tn_swz(row)) << 4) + (p & 1) * 8;
}

// Bijective XCD-aware remap of a 1-D block id (blocks b and b+8 share an XCD; give each XCD a
// contiguous range of tiles so neighbouring tiles share operand panels in that XCD's L2).
VMC_HD int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + loc;
}

// ---- attention (ViT, head_dim 64) -------------------------------------------------------------
// K tile image = X-style rows (key major).  V tile image: rows = keys, read by ds_read_b64_tr_b16 as
// 4-key x 16-column blocks; swizzle keeps the 32 lanes of a half on 32 distinct 8-B bank pairs.
VMC_HD int swz_v(int key) { return ((key >> 1) & 3) << 1; }
VMC_HD int lds_off_v(int key, int chunk) { return key * 128 + ((chunk ^ swz_v(key)) << 4); }
// S^T = K Q^T accumulators: tile nt, reg j of lane (r, q) is score[query r][key 16*nt + 4*q + j].
// P operand of k-step s (32 keys): element j<4 -> key 32 s + 4 q + j ; j>=4 -> key 32 s + 16 + 4 q + (j-4)
VMC_HD int attn_pv_key(int s, int q, int j) { return 32 * s + 16 * (j >> 2) + 4 * q + (j & 3); }
