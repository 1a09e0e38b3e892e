// Host emulation of one wave's lanes: checks the LDS images, swizzles and fragment->element maps of
// vimo_clip_amd/csrc/tile_index.h against plain matrix products, using the documented hardware rules
// (global_load_lds lane-linear placement; v_mfma_f32_16x16x32 operand/accumulator lane maps;
// ds_read_b64_tr_b16 gather; ds_read_b128 / tr_b16 bank groups).  Built with g++ by tests/test_host_index.py.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>

#include "../../vimo_clip_amd/csrc/tile_index.h"

static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { if (fails < 20) { printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } ++fails; } } while (0)

typedef int16_t e16;
struct Frag { e16 v[8]; };

// D[4(l>>4)+j][l&15] = sum_k A[row][k] B[k][col]; A from first-operand lanes, B from second-operand lanes
static void mfma16(const Frag a[64], const Frag b[64], long acc[64][4]) {
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 4; ++j) {
      const int row = 4 * (l >> 4) + j, col = l & 15;
      long s = 0;
      for (int k = 0; k < 32; ++k) s += (long)a[(k / 8) * 16 + row].v[k % 8] * (long)b[(k / 8) * 16 + col].v[k % 8];
      acc[l][j] += s;
    }
}

static const int B128_GROUPS[4][16] = {
    {0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
    {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
    {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
    {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};

// worst multiplicity of a 16-B slot (of the 256-B bank row) inside any ds_read_b128 lane group
static int b128_conflict(const int addr[64]) {
  int worst = 1;
  for (int g = 0; g < 4; ++g) {
    int cnt[16] = {0};
    std::set<int> seen;
    for (int i = 0; i < 16; ++i) {
      const int a = addr[B128_GROUPS[g][i]];
      if (seen.insert(a).second) { int s = (a / 16) % 16; if (++cnt[s] > worst) worst = cnt[s]; }
    }
  }
  return worst;
}

static void test_gemm(int MT, int WM, int WN) {
  const int BM = 16 * MT * WM, BN = 64 * WN, NT = 64 * WM * WN;
  std::vector<e16> A(BM * 64), W(BN * 64);
  for (auto& x : A) x = (e16)(rand() % 7 - 3);
  for (auto& x : W) x = (e16)(rand() % 7 - 3);
  std::vector<uint8_t> lds(BM * 128 + BN * 128, 0xEE);
  // staging: thread tid, pass i writes 16 B at LDS byte (i*NT + tid)*16 (wave base + lane*16)
  for (int tid = 0; tid < NT; ++tid) {
    for (int i = 0; i < BM * 8 / NT; ++i) {
      int row, ch; stage_src_x(i * NT + tid, row, ch);
      memcpy(&lds[(size_t)(i * NT + tid) * 16], &A[row * 64 + ch * 8], 16);
    }
    for (int i = 0; i < BN * 8 / NT; ++i) {
      int row, ch; stage_src_w(i * NT + tid, row, ch);
      memcpy(&lds[(size_t)BM * 128 + (size_t)(i * NT + tid) * 16], &W[row * 64 + ch * 8], 16);
    }
  }
  for (int wave = 0; wave < WM * WN; ++wave) {
    const int wm = wave / WN, wn = wave % WN;
    for (int mt = 0; mt < MT; ++mt)
      for (int nt = 0; nt < 4; ++nt) {
        long acc[64][4]; memset(acc, 0, sizeof acc);
        for (int kk = 0; kk < 2; ++kk) {
          Frag wf[64], xf[64]; int xaddr[64], waddr[64];
          for (int l = 0; l < 64; ++l) {
            const int r = l & 15, q = l >> 4;
            xaddr[l] = lds_off_x(wm * 16 * MT + 16 * mt + r, 4 * kk + q);
            waddr[l] = BM * 128 + lds_off_w(wn * 64 + gemm_w_row(r, nt), 4 * kk + q);
            memcpy(xf[l].v, &lds[xaddr[l]], 16);
            memcpy(wf[l].v, &lds[waddr[l]], 16);
          }
          CHECK(b128_conflict(xaddr) == 1, "gemm X read conflict %d", b128_conflict(xaddr));
          CHECK(b128_conflict(waddr) == 1, "gemm W read conflict %d", b128_conflict(waddr));
          mfma16(wf, xf, acc);  // operands (W, X)
        }
        for (int l = 0; l < 64; ++l)
          for (int j = 0; j < 4; ++j) {
            const int r = l & 15, q = l >> 4;
            const int row = wm * 16 * MT + 16 * mt + r, col = wn * 64 + gemm_c_col(q, nt, j);
            long ref = 0;
            for (int k = 0; k < 64; ++k) ref += (long)A[row * 64 + k] * W[col * 64 + k];
            CHECK(acc[l][j] == ref, "gemm MT%d WM%d WN%d wave %d mt %d nt %d lane %d j %d: %ld vs %ld", MT, WM, WN, wave, mt, nt, l, j, acc[l][j], ref);
          }
      }
  }
}

// ds_read_b64_tr_b16: lane i of 16-lane group G receives element e = the (i&3)-th 16-bit element of the
// 8-byte piece addressed by lane 16G + 4e + (i>>2)
static void tr_read(const std::vector<uint8_t>& lds, const int addr[64], e16 out[64][4]) {
  for (int l = 0; l < 64; ++l) {
    const int G = l >> 4, i = l & 15;
    for (int e = 0; e < 4; ++e) {
      const int src = 16 * G + 4 * e + (i >> 2);
      memcpy(&out[l][e], &lds[addr[src] + 2 * (i & 3)], 2);
    }
  }
}
static int tr_conflict(const int addr[64]) {  // 32-lane halves, 8-byte units of the 256-B bank row
  int worst = 1;
  for (int h = 0; h < 2; ++h) {
    int cnt[32] = {0};
    std::set<int> seen;
    for (int l = 32 * h; l < 32 * h + 32; ++l)
      if (seen.insert(addr[l]).second) { int u = (addr[l] / 8) % 32; if (++cnt[u] > worst) worst = cnt[u]; }
  }
  return worst;
}

static void test_attention(int NT, int N) {
  const int NK = 16 * NT;
  std::vector<e16> Q(16 * 64), K(NK * 64, 0), V(NK * 64, 0);
  for (auto& x : Q) x = (e16)(rand() % 5 - 2);
  for (int i = 0; i < N * 64; ++i) { K[i] = (e16)(rand() % 5 - 2); V[i] = (e16)(rand() % 5 - 2); }
  std::vector<uint8_t> kl(NK * 128), vl(NK * 128);
  for (int row = 0; row < NK; ++row)
    for (int c = 0; c < 8; ++c) {
      memcpy(&kl[lds_off_x(row, c)], &K[row * 64 + c * 8], 16);
      memcpy(&vl[lds_off_v(row, c)], &V[row * 64 + c * 8], 16);
    }
  // S^T tiles
  std::vector<std::vector<long>> s(NT, std::vector<long>(64 * 4, 0));
  for (int nt = 0; nt < NT; ++nt) {
    long acc[64][4]; memset(acc, 0, sizeof acc);
    for (int kk = 0; kk < 2; ++kk) {
      Frag kf[64], qf[64]; int kaddr[64];
      for (int l = 0; l < 64; ++l) {
        const int r = l & 15, q = l >> 4;
        kaddr[l] = lds_off_x(16 * nt + r, 4 * kk + q);
        memcpy(kf[l].v, &kl[kaddr[l]], 16);
        memcpy(qf[l].v, &Q[r * 64 + (4 * kk + q) * 8], 16);
      }
      CHECK(b128_conflict(kaddr) == 1, "attn K read conflict");
      mfma16(kf, qf, acc);
    }
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 4; ++j) {
        const int r = l & 15, q = l >> 4, key = 16 * nt + 4 * q + j;
        long ref = 0;
        for (int d = 0; d < 64; ++d) ref += (long)Q[r * 64 + d] * K[key * 64 + d];
        CHECK(acc[l][j] == ref, "attn S nt %d lane %d j %d", nt, l, j);
        s[nt][l * 4 + j] = (key < N) ? (acc[l][j] % 3) : 0;  // small "P" values, masked keys -> 0
      }
  }
  // O^T = V^T P^T
  for (int dt = 0; dt < 4; ++dt) {
    long o[64][4]; memset(o, 0, sizeof o);
    for (int ks = 0; ks < NT / 2; ++ks) {
      Frag pf[64], vf[64]; int a0[64], a1[64]; e16 t0[64][4], t1[64][4];
      for (int l = 0; l < 64; ++l) {
        const int r = l & 15, q = l >> 4;
        for (int j = 0; j < 8; ++j) pf[l].v[j] = (e16)s[2 * ks + (j >> 2)][l * 4 + (j & 3)];
        const int key0 = 32 * ks + 4 * q + (r >> 2), chunk = 2 * dt + ((r & 3) >> 1), half = (r & 1) * 8;
        a0[l] = lds_off_v(key0, chunk) + half;
        a1[l] = lds_off_v(key0 + 16, chunk) + half;
      }
      CHECK(tr_conflict(a0) == 1 && tr_conflict(a1) == 1, "attn V tr-read conflict %d %d", tr_conflict(a0), tr_conflict(a1));
      tr_read(vl, a0, t0); tr_read(vl, a1, t1);
      for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) { vf[l].v[e] = t0[l][e]; vf[l].v[4 + e] = t1[l][e]; }
      // the k-slot -> key map both operands must agree on
      for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) {
        const int r = l & 15, q = l >> 4;
        CHECK(vf[l].v[j] == V[attn_pv_key(ks, q, j) * 64 + 16 * dt + r], "V frag mismatch ks %d lane %d j %d", ks, l, j);
      }
      mfma16(vf, pf, o);
    }
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 4; ++j) {
        const int r = l & 15, q = l >> 4, d = 16 * dt + 4 * q + j;
        long ref = 0;
        for (int key = 0; key < NK; ++key) {
          long sv = 0;
          for (int dd = 0; dd < 64; ++dd) sv += (long)Q[r * 64 + dd] * K[key * 64 + dd];
          const long p = (key < N) ? (sv % 3) : 0;
          ref += p * V[key * 64 + d];
        }
        CHECK(o[l][j] == ref, "attn O dt %d lane %d j %d: %ld vs %ld", dt, l, j, o[l][j], ref);
      }
  }
}


// 8-phase kernel: 4 half-tile slots (A0, A1, B0, B1) of one K tile, 8 waves as 2 x 4
static void test_gemm8() {
  std::vector<e16> A(256 * 64), W(256 * 64);
  for (auto& x : A) x = (e16)(rand() % 7 - 3);
  for (auto& x : W) x = (e16)(rand() % 7 - 3);
  std::vector<uint8_t> slot[4];  // A0 A1 B0 B1
  for (auto& s : slot) s.assign(16384, 0xEE);
  for (int tid = 0; tid < 512; ++tid)
    for (int i = 0; i < 2; ++i) {
      int row, ch;
      stage_src_x(i * 512 + tid, row, ch);
      memcpy(&slot[0][(size_t)(i * 512 + tid) * 16], &A[row * 64 + ch * 8], 16);
      memcpy(&slot[1][(size_t)(i * 512 + tid) * 16], &A[(128 + row) * 64 + ch * 8], 16);
      stage_src_w8(i * 512 + tid, row, ch);
      memcpy(&slot[2][(size_t)(i * 512 + tid) * 16], &W[row * 64 + ch * 8], 16);
      memcpy(&slot[3][(size_t)(i * 512 + tid) * 16], &W[(128 + row) * 64 + ch * 8], 16);
    }
  for (int wave = 0; wave < 8; ++wave) {
    const int wm = wave >> 2, wn = wave & 3;
    for (int mh = 0; mh < 2; ++mh) for (int nh = 0; nh < 2; ++nh)
      for (int mt = 0; mt < 4; ++mt) for (int nt = 0; nt < 2; ++nt) {
        long acc[64][4]; memset(acc, 0, sizeof acc);
        for (int kk = 0; kk < 2; ++kk) {
          Frag wf[64], xf[64]; int xaddr[64], waddr[64];
          for (int l = 0; l < 64; ++l) {
            const int r = l & 15, q = l >> 4;
            xaddr[l] = lds_off_x(64 * wm + r, 4 * kk + q) + mt * 2048;
            waddr[l] = lds_off_w8(32 * wn + g8_w_row(r, nt), 4 * kk + q);
            memcpy(xf[l].v, &slot[mh][xaddr[l]], 16);
            memcpy(wf[l].v, &slot[2 + nh][waddr[l]], 16);
          }
          CHECK(b128_conflict(xaddr) == 1, "g8 X read conflict %d", b128_conflict(xaddr));
          CHECK(b128_conflict(waddr) == 1, "g8 W read conflict %d", b128_conflict(waddr));
          mfma16(wf, xf, acc);
        }
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
          const int r = l & 15, q = l >> 4;
          const int row = 128 * mh + 64 * wm + 16 * mt + r, col = 128 * nh + 32 * wn + 8 * q + 4 * nt + j;
          long ref = 0;
          for (int k = 0; k < 64; ++k) ref += (long)A[row * 64 + k] * W[col * 64 + k];
          CHECK(acc[l][j] == ref, "g8 wave %d mh %d nh %d mt %d nt %d lane %d j %d", wave, mh, nh, mt, nt, l, j);
        }
      }
  }
}

// v_mfma_f32_32x32x16: first operand lane l holds A[row l&31][k = 8(l>>5)+j], second B[k = 8(l>>5)+j][col l&31];
// accumulator reg i of lane l is D[row 8(i>>2) + 4(l>>5) + (i&3)][col l&31]
static void mfma32(const Frag a[64], const Frag b[64], long acc[64][16]) {
  for (int l = 0; l < 64; ++l)
    for (int i = 0; i < 16; ++i) {
      const int row = 8 * (i >> 2) + 4 * (l >> 5) + (i & 3), col = l & 31;
      long s = 0;
      for (int k = 0; k < 16; ++k) s += (long)a[(k / 8) * 32 + row].v[k % 8] * (long)b[(k / 8) * 32 + col].v[k % 8];
      acc[l][i] += s;
    }
}

// the 8-phase tile on 32x32x16 fragments (same LDS images and staging as test_gemm8)
static void test_gemm8_mfma32() {
  std::vector<e16> A(256 * 64), W(256 * 64);
  for (auto& x : A) x = (e16)(rand() % 7 - 3);
  for (auto& x : W) x = (e16)(rand() % 7 - 3);
  std::vector<uint8_t> slot[4];  // A0 A1 B0 B1
  for (auto& s : slot) s.assign(16384, 0xEE);
  for (int tid = 0; tid < 512; ++tid)
    for (int i = 0; i < 2; ++i) {
      int row, ch;
      stage_src_x(i * 512 + tid, row, ch);
      memcpy(&slot[0][(size_t)(i * 512 + tid) * 16], &A[row * 64 + ch * 8], 16);
      memcpy(&slot[1][(size_t)(i * 512 + tid) * 16], &A[(128 + row) * 64 + ch * 8], 16);
      stage_src_w8(i * 512 + tid, row, ch);
      memcpy(&slot[2][(size_t)(i * 512 + tid) * 16], &W[row * 64 + ch * 8], 16);
      memcpy(&slot[3][(size_t)(i * 512 + tid) * 16], &W[(128 + row) * 64 + ch * 8], 16);
    }
  for (int wave = 0; wave < 8; ++wave) {
    const int wm = wave >> 2, wn = wave & 3;
    for (int mh = 0; mh < 2; ++mh) for (int nh = 0; nh < 2; ++nh)
      for (int mt = 0; mt < 2; ++mt) {
        long acc[64][16]; memset(acc, 0, sizeof acc);
        for (int kk = 0; kk < 4; ++kk) {
          Frag wf[64], xf[64]; int xaddr[64], waddr[64];
          for (int l = 0; l < 64; ++l) {
            const int x = l & 31, h = l >> 5;
            // the kernel forms the k-step address as (offset of k-step 0) XOR (kk << 5), A row tiles by an immediate
            xaddr[l] = (lds_off_x(64 * wm + x, h) ^ (kk << 5)) + mt * 4096;
            waddr[l] = lds_off_w8(32 * wn + g8_w_row32(x), h) ^ (kk << 5);
            CHECK(xaddr[l] == lds_off_x(64 * wm + 32 * mt + x, 2 * kk + h), "g8/32 X address form");
            CHECK(waddr[l] == lds_off_w8(32 * wn + g8_w_row32(x), 2 * kk + h), "g8/32 W address form");
            memcpy(xf[l].v, &slot[mh][xaddr[l]], 16);
            memcpy(wf[l].v, &slot[2 + nh][waddr[l]], 16);
          }
          CHECK(b128_conflict(xaddr) == 1, "g8/32 X read conflict %d", b128_conflict(xaddr));
          CHECK(b128_conflict(waddr) == 1, "g8/32 W read conflict %d", b128_conflict(waddr));
          mfma32(wf, xf, acc);
        }
        for (int l = 0; l < 64; ++l) for (int i = 0; i < 16; ++i) {
          const int x = l & 31, h = l >> 5;
          const int row = 128 * mh + 64 * wm + 32 * mt + x, col = 128 * nh + 32 * wn + g8_c_col32(h, i);
          long ref = 0;
          for (int k = 0; k < 64; ++k) ref += (long)A[row * 64 + k] * W[col * 64 + k];
          CHECK(acc[l][i] == ref, "g8/32 wave %d mh %d nh %d mt %d lane %d i %d", wave, mh, nh, mt, l, i);
        }
      }
  }
}

// TN weight gradient on the 256 x 256 tile: one 64-token stage = 4 half-tile slots (dY cols 0..127 / 128..255, X cols 0..127 / 128..255)
static void test_tn256() {
  std::vector<e16> dY(64 * 256), X(64 * 256);      // [token][column]
  for (auto& x : dY) x = (e16)(rand() % 7 - 3);
  for (auto& x : X) x = (e16)(rand() % 7 - 3);
  std::vector<uint8_t> slot[4];  // A0 A1 B0 B1
  for (auto& s : slot) s.assign(16384, 0xEE);
  for (int tid = 0; tid < 512; ++tid)
    for (int i = 0; i < 2; ++i) {
      int row, ch;
      tn_stage_src(i * 512 + tid, row, ch);
      for (int h = 0; h < 2; ++h) {
        memcpy(&slot[h][(size_t)(i * 512 + tid) * 16], &dY[row * 256 + 128 * h + ch * 8], 16);
        memcpy(&slot[2 + h][(size_t)(i * 512 + tid) * 16], &X[row * 256 + 128 * h + ch * 8], 16);
      }
    }
  for (int wave = 0; wave < 8; ++wave) {
    const int wm = wave >> 2, wn = wave & 3;
    for (int mh = 0; mh < 2; ++mh) for (int nh = 0; nh < 2; ++nh)
      for (int mt = 0; mt < 4; ++mt) for (int nt = 0; nt < 2; ++nt) {
        long acc[64][4]; memset(acc, 0, sizeof acc);
        for (int kk = 0; kk < 2; ++kk) {
          Frag af[64], bf[64];
          for (int e = 0; e < 2; ++e) {
            int aaddr[64], baddr[64];
            for (int l = 0; l < 64; ++l) {
              aaddr[l] = (tn256_a_off(wm, l, e) ^ (mt << 5)) + 8192 * kk;
              baddr[l] = (tn256_b_off(wn, l, e) ^ (nt << 5)) + 8192 * kk;
            }
            CHECK(tr_conflict(aaddr) == 1, "tn256 A read conflict %d", tr_conflict(aaddr));
            CHECK(tr_conflict(baddr) == 1, "tn256 B read conflict %d", tr_conflict(baddr));
            e16 oa[64][4], ob[64][4];
            tr_read(slot[mh], aaddr, oa);
            tr_read(slot[2 + nh], baddr, ob);
            for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) { af[l].v[4 * e + j] = oa[l][j]; bf[l].v[4 * e + j] = ob[l][j]; }
          }
          mfma16(bf, af, acc);       // (X, dY)
        }
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
          const int r = l & 15, g = l >> 4;
          const int n = 128 * mh + 64 * wm + 16 * mt + r, k = 128 * nh + 32 * wn + 16 * nt + 4 * g + j;
          long ref = 0;
          for (int t = 0; t < 64; ++t) ref += (long)dY[t * 256 + n] * X[t * 256 + k];
          CHECK(acc[l][j] == ref, "tn256 wave %d mh %d nh %d mt %d nt %d lane %d j %d: %ld vs %ld", wave, mh, nh, mt, nt, l, j, acc[l][j], ref);
        }
      }
    // bias gradient: pattern operand on the X side (ones iff (lane & 15) >> 2 == mt); lane (r, g) ends with the sum of column 64 wm + 16 g + r
    if (wn == 0)
      for (int mh = 0; mh < 2; ++mh) {
        long acc[64][4]; memset(acc, 0, sizeof acc);
        for (int mt = 0; mt < 4; ++mt)
          for (int kk = 0; kk < 2; ++kk) {
            Frag af[64], pf[64];
            for (int e = 0; e < 2; ++e) {
              int aaddr[64];
              for (int l = 0; l < 64; ++l) aaddr[l] = (tn256_a_off(wm, l, e) ^ (mt << 5)) + 8192 * kk;
              e16 oa[64][4];
              tr_read(slot[mh], aaddr, oa);
              for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) af[l].v[4 * e + j] = oa[l][j];
            }
            for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) pf[l].v[j] = (e16)((((l & 15) >> 2) == mt) ? 1 : 0);
            mfma16(pf, af, acc);
          }
        for (int l = 0; l < 64; ++l) {
          const int r = l & 15, g = l >> 4, n = 128 * mh + 64 * wm + 16 * g + r;
          long ref = 0;
          for (int t = 0; t < 64; ++t) ref += dY[t * 256 + n];
          CHECK(acc[l][0] == ref, "tn256 bias wave %d mh %d lane %d: %ld vs %ld", wave, mh, l, acc[l][0], ref);
        }
      }
  }
}

static void test_xcd_remap() {
  for (int nwg : {1, 7, 8, 9, 63, 64, 100, 1028, 3084}) {
    std::set<int> seen;
    for (int b = 0; b < nwg; ++b) { int t = xcd_remap(b, nwg); CHECK(t >= 0 && t < nwg, "remap range"); seen.insert(t); }
    CHECK((int)seen.size() == nwg, "remap not bijective for %d", nwg);
  }
}

int main() {
  srand(1234);
  test_gemm(8, 2, 4);
  test_gemm(4, 2, 2);
  test_gemm(2, 2, 1);
  test_gemm8();
  test_gemm8_mfma32();
  test_tn256();
  test_attention(18, 257);
  test_attention(4, 50);
  test_attention(2, 17);
  test_xcd_remap();
  if (fails) { printf("%d failures\n", fails); return 1; }
  printf("tile index emulation OK\n");
  return 0;
}
