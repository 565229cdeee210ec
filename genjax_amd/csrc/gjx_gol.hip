// gjx_gol.hip -- Game-of-Life inversion by single-site Gibbs sweeps, batched chains
// (examples/gol/core.py:24-45 cell model, :108-163 conditional of one cell, :166-204 the
// nine-colour sweep, :207-230 summaries).
//
// Layout: B chains x n x n cells, one byte per cell, row-major; the target ("future") board is
// shared by all chains (stride 0) or per chain.  One launch = one colour pass (cells
// (oy + 3a, ox + 3b)): cells of one colour are >= 3 apart and a cell's conditional reads a
// 5 x 5 window, so a pass updates in place without races, and consecutive passes are ordered
// by the stream.
// (A temporally blocked variant -- one launch per sweep, 96 x 96 tiles with an 18-cell halo whose
// earlier-pass updates each block recomputes -- was bit-identical but slower, 0.94 vs 0.86 ms per
// sweep at 64 x 1024^2: the pass is bound by the ~350 instructions of a cell's conditional, not by
// the staging, and the halo costs 1.37 x the updates.)
// Kernel shape: a 256-thread block updates 16 x 16 cells of the colour = a 48 x 48 cell area.
// The 50 x 50 window of the current board and the 48 x 48 window of the target are staged in
// LDS (zero padded at the board edge, core.py:56), each thread then works from registers:
// 25 + 9 byte reads, integer neighbour counts, 18 table look-ups, one 2-way softmax, one
// Threefry block.  HBM-bound at scale: per pass ~ (50*50 + 48*48) / 256 = 18.8 B read and 1 B
// written per updated cell, i.e. ~ 19 B per cell per sweep.
#include "gjx_common.h"

constexpr int GOL_T = 16;               // updated cells per block edge
constexpr int GOL_A = 3 * GOL_T;        // cell area per block edge (48)
constexpr int GOL_CW = GOL_A + 2;       // current-board window edge (50): rows i-2 .. i+2
constexpr int GOL_LD = 64;              // LDS row stride (bytes): 14 staged dwords + shift slack

struct GolTable {
  float t[2][2];  // [bit][det]: log-density of flip(det ? 1 - f : f) at bit
};

static GolTable gol_table(float f) {
  GolTable g;
  const float omf = 1.0f - f;
  g.t[1][1] = logf(omf);
  g.t[1][0] = logf(f);
  g.t[0][1] = log1pf(-omf);
  g.t[0][0] = log1pf(-f);
  return g;
}

__device__ __forceinline__ int gol_rule(int centre, int nb) {
  return centre ? ((nb == 2) | (nb == 3)) : (nb == 3);
}

// Conditional of one cell given the 5 x 5 window w of the current board around it and the 3 x 3
// window fb of the target (core.py:108-163): scores of bit = 0 / 1 summed over the nine affected
// target cells in row-major order, p(bit = 1) = 1 - softmax(scores)[0].
// INTERIOR: the 3 x 3 neighbourhood lies inside the board (no masking).
template <bool INTERIOR>
__device__ __forceinline__ float gol_p_one(const GolTable& tab, const int (&w)[5][5], const int (&fb)[3][3],
                                           int i, int j, int n) {
  // 3-wide horizontal sums of the five rows, then 3-high sums of those: the nine 3 x 3 block sums
  int h[5][3];
#pragma unroll
  for (int r = 0; r < 5; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) h[r][c] = w[r][c] + w[r][c + 1] + w[r][c + 2];
  const int me = w[2][2];
  float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy) {
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const int sum9 = h[dy + 1][dx + 1] + h[dy + 2][dx + 1] + h[dy + 3][dx + 1];
      int r0, r1;   // rule outcome of the target cell's predecessor neighbourhood with me = 0 / 1
      if (dy == 0 && dx == 0) {
        const int nb = sum9 - me;           // alive next iff nb == 3, or alive now and nb == 2
        r0 = nb == 3;
        r1 = (nb | 1) == 3;
      } else {
        const int ctr = w[2 + dy][2 + dx];
        const int nb0 = sum9 - me - ctr;    // (nb | centre) == 3  <=>  nb == 3 or (centre and nb == 2)
        r0 = (nb0 | ctr) == 3;
        r1 = ((nb0 + 1) | ctr) == 3;
      }
      const int bit = fb[dy + 1][dx + 1];
      // the target bit is shared by both candidates: pick the table row once
      const float dead = bit ? tab.t[1][0] : tab.t[0][0], alive = bit ? tab.t[1][1] : tab.t[0][1];
      float t0 = r0 ? alive : dead, t1 = r1 ? alive : dead;
      if (!INTERIOR) {
        // mask of get_gibbs_probs_fast (core.py:141-149; square boards)
        const int fy = i + dy, fx = j + dx;
        const bool in_range = (fy >= 0) & (fy < n) & (fx >= 0) & (fx < n);
        t0 = in_range ? t0 : 0.0f;
        t1 = in_range ? t1 : 0.0f;
      }
      s0 += t0;
      s1 += t1;
    }
  }
  // softmax(scores)[0], then bit ~ flip(1 - p_zero) (core.py:151-163)
  const float m = fmaxf(s0, s1);
  const float e0 = expf(s0 - m), e1 = expf(s1 - m);
  const float p_zero = __fdiv_rn(e0, gjx_add_rn(e0, e1));
  return gjx_sub_rn(1.0f, p_zero);
}

template <bool VEC>
__global__ void __launch_bounds__(GJX_BLOCK)
    k_gol_gibbs_pass(GolTable tab, const uint32_t* __restrict__ pass_keys, int n, int oy, int ox,
                     int nyp, int nxp, const uint8_t* __restrict__ future, int64_t future_stride,
                     uint8_t* boards) {
  __shared__ uint8_t s_cur[GOL_CW][GOL_LD];
  __shared__ uint8_t s_fut[GOL_A][GOL_LD];
  const int chain = blockIdx.z;
  uint8_t* cur = boards + (int64_t)chain * n * n;
  const uint8_t* fut = future + (int64_t)chain * future_stride;
  const int i0 = oy + GOL_A * blockIdx.y, j0 = ox + GOL_A * blockIdx.x;  // first updated cell
  // Staging.  VEC (n % 4 == 0): aligned dword loads -- a dword lies entirely inside or outside
  // the board, so the zero padding is per dword; the tile then starts `sh` bytes into the row.
  int sh_c = 0, sh_f = 0;
  if (VEC) {
    constexpr int DW = 14;  // dwords per staged row: 3 bytes of alignment slack + 50 <= 56
    const int xc = (j0 - 2) & ~3, xf = (j0 - 1) & ~3;  // floor to a multiple of 4 (two's complement)
    sh_c = (j0 - 2) - xc;
    sh_f = (j0 - 1) - xf;
    for (int idx = threadIdx.x; idx < GOL_CW * DW; idx += GJX_BLOCK) {
      const int r = idx / DW, d = idx - r * DW;
      const int y = i0 - 2 + r, x = xc + 4 * d;
      uint32_t v = 0u;
      if (y >= 0 && y < n && x >= 0 && x < n) v = *reinterpret_cast<const uint32_t*>(cur + (int64_t)y * n + x);
      *reinterpret_cast<uint32_t*>(&s_cur[r][4 * d]) = v;
    }
    for (int idx = threadIdx.x; idx < GOL_A * DW; idx += GJX_BLOCK) {
      const int r = idx / DW, d = idx - r * DW;
      const int y = i0 - 1 + r, x = xf + 4 * d;
      uint32_t v = 0u;
      if (y >= 0 && y < n && x >= 0 && x < n) v = *reinterpret_cast<const uint32_t*>(fut + (int64_t)y * n + x);
      *reinterpret_cast<uint32_t*>(&s_fut[r][4 * d]) = v;
    }
  } else {
    for (int idx = threadIdx.x; idx < GOL_CW * GOL_CW; idx += GJX_BLOCK) {
      const int r = idx / GOL_CW, c = idx - r * GOL_CW;
      const int y = i0 - 2 + r, x = j0 - 2 + c;
      s_cur[r][c] = (y >= 0 && y < n && x >= 0 && x < n) ? cur[(int64_t)y * n + x] : (uint8_t)0;
    }
    for (int idx = threadIdx.x; idx < GOL_A * GOL_A; idx += GJX_BLOCK) {
      const int r = idx / GOL_A, c = idx - r * GOL_A;
      const int y = i0 - 1 + r, x = j0 - 1 + c;
      s_fut[r][c] = (y >= 0 && y < n && x >= 0 && x < n) ? fut[(int64_t)y * n + x] : (uint8_t)0;
    }
  }
  __syncthreads();
  const int ty = threadIdx.x / GOL_T, tx = threadIdx.x - ty * GOL_T;
  const int a = GOL_T * blockIdx.y + ty, b = GOL_T * blockIdx.x + tx;  // index within the colour
  const int i = oy + 3 * a, j = ox + 3 * b;
  if (a >= nyp || b >= nxp) return;
  // 5 x 5 window of the current board around (i, j), and its 3-wide horizontal sums
  int w[5][5];
#pragma unroll
  for (int r = 0; r < 5; ++r)
#pragma unroll
    for (int c = 0; c < 5; ++c) w[r][c] = s_cur[3 * ty + r][sh_c + 3 * tx + c];
  int fb[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) fb[r][c] = s_fut[3 * ty + r][sh_f + 3 * tx + c];
  const bool interior = (i >= 1) & (i + 1 < n) & (j >= 1) & (j + 1 < n);
  const float p_one = interior ? gol_p_one<true>(tab, w, fb, i, j, n) : gol_p_one<false>(tab, w, fb, i, j, n);
  const uint32_t k0 = pass_keys[2 * chain], k1 = pass_keys[2 * chain + 1];
  const float u = gjx_uniform_at(k0, k1, (uint64_t)a * (uint64_t)nxp + (uint64_t)b, 0.0f, 1.0f);
  cur[(int64_t)i * n + j] = (u < p_one) ? (uint8_t)1 : (uint8_t)0;
}

// ---------------------------------------------------------------------------
// Bit-packed colour pass.  Boards and target as one bit per cell (row pitch wp = ceil(n / 64) 64-bit words,
// bits beyond n zero).  Same conditional, same draws, same bits as k_gol_gibbs_pass; what changes is how the
// 18 rule outcomes and the two scores of a cell are formed:
//  * cells of one colour are >= 3 apart, so every target cell has at most ONE updated cell in its 3 x 3
//    neighbourhood: the rule outcome of all nine targets of every updated cell with that cell = 0 (= 1) is simply
//    the Life step of the board with ALL cells of the colour cleared (set).  Two bit-parallel Life steps per
//    48-row window (~40 64-bit operations per row of 48 cells) replace ~175 instructions per cell.
//  * a score is the left-to-right f32 sum of nine table values T[target bit][outcome].  The sum of the first six
//    (rows i - 1 and i: 12 bits) is looked up in a 4096-entry LDS table built per block with the same additions
//    in the same order; the last three are added one by one.  Bit-identical by construction.
//  * staging is 100 64-bit loads per 48 x 48 window instead of 5 KB of bytes; the 256 new bits leave as at most
//    32 atomic XORs of the changed bits (tiles are 48 wide, words 64: neighbours share words).
// A block walks over many tiles (the table build is amortised).
// ---------------------------------------------------------------------------
constexpr int GOLP_ROWS = GOL_CW;                      // 50 window rows, window bit c = column j0 - 2 + c
constexpr uint64_t GOLP_CM = 0x0000924924924924ull;    // bits 2, 5, ..., 47: the updated columns of a window

// 64 cells of row `row` starting at column x0 (may be negative); cells outside [0, n) read as 0
__device__ __forceinline__ uint64_t golp_load(const uint64_t* __restrict__ row, int wp, int x0) {
  const int w = x0 >> 6, sh = x0 & 63;   // arithmetic shift: floor
  const uint64_t lo = (w >= 0 && w < wp) ? row[w] : 0ull;
  const uint64_t hi = (w + 1 >= 0 && w + 1 < wp) ? row[w + 1] : 0ull;
  return sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
}

__device__ __forceinline__ uint64_t golp_maj(uint64_t a, uint64_t b, uint64_t c) { return (a & b) | (c & (a | b)); }

// Bit-parallel Life over 64 columns.  golp_hsum: the 3-wide horizontal sum of a row (bit 0 in s, bit 1 in c);
// golp_rule: the next state of the middle row from the sums of the rows above / at / below it.
struct GolpSum {
  uint64_t s, c;
};
__device__ __forceinline__ GolpSum golp_hsum(uint64_t x) {
  const uint64_t l = x << 1, r = x >> 1;
  return GolpSum{x ^ l ^ r, golp_maj(x, l, r)};
}
__device__ __forceinline__ uint64_t golp_rule(const GolpSum& u, const GolpSum& m, const GolpSum& d, uint64_t mid) {
  const uint64_t b0 = u.s ^ m.s ^ d.s, k0 = golp_maj(u.s, m.s, d.s);  // ones: bit 0, carry into the twos
  const uint64_t t1 = u.c ^ m.c ^ d.c, k1 = golp_maj(u.c, m.c, d.c);  // twos: bit 1 (before k0), carry into the fours
  const uint64_t b1 = t1 ^ k0, k2 = t1 & k0;
  const uint64_t b2 = k1 ^ k2, b3 = k1 & k2;                          // sum of the 3 x 3 block, centre included
  const uint64_t is3 = ~b3 & ~b2 & b1 & b0, is4 = ~b3 & b2 & ~b1 & ~b0;
  return is3 | (mid & is4);                                           // alive next: 3 in the block, or alive and 4
}

__device__ __forceinline__ uint64_t golp_spread3(uint64_t x) {   // bit k -> bit 3 k (k < 21)
  x &= 0x1fffffull;
  x = (x | x << 32) & 0x1f00000000ffffull;
  x = (x | x << 16) & 0x1f0000ff0000ffull;
  x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}

__global__ void __launch_bounds__(GJX_BLOCK)
    k_gol_pass_packed(GolTable tab, const uint32_t* __restrict__ pass_keys, int n, int wp, int oy, int ox, int nyp,
                      int nxp, int tiles_x, int tiles_y, int64_t n_tiles, const uint64_t* __restrict__ future,
                      int64_t future_stride, uint64_t* boards) {
  __shared__ float s_tail[64][3];     // [f | r << 3][k]: T[f bit k][r bit k]
  __shared__ float s_lut3[64];        // (t0 + t1) + t2 of one row code
  __shared__ float s_lut6[4096];      // six terms: rows i - 1 (low 6 bits of the index) and i
  __shared__ uint64_t s_cur[2][GOLP_ROWS];
  // target bits and the two Life steps as 32-bit views of every row: [.][0] = columns 0..31, [.][1] = columns
  // 18..49 of the window; a cell's three columns lie in one of them (64-bit shifts run at quarter rate)
  __shared__ uint32_t s_fut[2][GOLP_ROWS][2], s_n0[2][GOLP_ROWS][2], s_n1[2][GOLP_ROWS][2];
  for (int e = threadIdx.x; e < 192; e += GJX_BLOCK) {
    const int c6 = e / 3, k = e - 3 * c6;
    s_tail[c6][k] = tab.t[(c6 >> k) & 1][(c6 >> (3 + k)) & 1];
  }
  __syncthreads();
  if (threadIdx.x < 64) s_lut3[threadIdx.x] = (s_tail[threadIdx.x][0] + s_tail[threadIdx.x][1]) + s_tail[threadIdx.x][2];
  __syncthreads();
  for (int e = threadIdx.x; e < 4096; e += GJX_BLOCK) {
    const int lo = e & 63, hi = e >> 6;
    s_lut6[e] = ((s_lut3[lo] + s_tail[hi][0]) + s_tail[hi][1]) + s_tail[hi][2];
  }
  __syncthreads();
  const int64_t cells = (int64_t)n * wp;   // words per board
  const int ty = threadIdx.x / GOL_T, tx = threadIdx.x - ty * GOL_T;
  const uint32_t per_chain = (uint32_t)(tiles_x * tiles_y);
  // Every iteration works on TWO tiles (`tile` and `tile + gridDim.x`): their staging, Life steps and cells are
  // independent instruction streams between the same three barriers -- the step is a chain of dependent LDS and
  // ALU latencies, and one tile at a time leaves the SIMDs idle for most of it.
  // Threads 0..127 stage / run the Life steps of the first tile, 128..255 of the second; all 256 threads then
  // update one cell of each.
  const int half = threadIdx.x >> 7, st_t = threadIdx.x & 127, st_row = st_t & 63;
  const bool stager = st_row < GOLP_ROWS;   // of a half: threads 0..49 board rows, 64..113 target rows
  const bool small = n_tiles <= 0x7fffffffll;   // 32-bit tile arithmetic (a 64-bit division is ~150 scalar instructions)
  auto locate = [&](int64_t t, int& chain_, int& by_, int& bx_) {
    uint32_t rem_;
    if (small) {
      chain_ = (int)((uint32_t)t / per_chain);
      rem_ = (uint32_t)t - (uint32_t)chain_ * per_chain;
    } else {
      chain_ = (int)(t / per_chain);
      rem_ = (uint32_t)(t - (int64_t)chain_ * per_chain);
    }
    by_ = (int)(rem_ / (uint32_t)tiles_x);
    bx_ = (int)(rem_ - (uint32_t)by_ * (uint32_t)tiles_x);
  };
  // one staged row of tile t: 50 cells from column j0 - 2, zero outside the board
  auto fetch = [&](int64_t t) -> uint64_t {
    if (t >= n_tiles) return 0ull;
    int chain_, by_, bx_;
    locate(t, chain_, by_, bx_);
    const int y = oy + GOL_A * by_ - 2 + st_row;
    if (y < 0 || y >= n) return 0ull;
    const uint64_t* src = (st_t < 64) ? boards + (int64_t)chain_ * cells : future + (int64_t)chain_ * future_stride;
    return golp_load(src + (int64_t)y * wp, wp, ox + GOL_A * bx_ - 2) & ((1ull << GOLP_ROWS) - 1ull);
  };
  // The rows of the NEXT pair of tiles are loaded while this pair is worked on (a tile's window holds no cell
  // that another tile of the pass updates, so what is read early is what would be read late).
  const int64_t stride = 2 * (int64_t)gridDim.x;
  uint64_t staged = stager ? fetch((int64_t)blockIdx.x + (int64_t)half * gridDim.x) : 0ull;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += stride) {
    if (stager) {
      if (st_t < 64) {
        s_cur[half][st_row] = staged;
      } else {
        s_fut[half][st_row][0] = (uint32_t)staged;
        s_fut[half][st_row][1] = (uint32_t)(staged >> 18);
      }
      staged = fetch(tile + stride + (int64_t)half * gridDim.x);
    }
    int chain[2], by[2], bx[2], vx[2], vy[2];
    uint64_t cm[2];
    bool live[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int64_t t = tile + (int64_t)h * gridDim.x;
      live[h] = t < n_tiles;
      locate(live[h] ? t : tile, chain[h], by[h], bx[h]);
      vx[h] = min(GOL_T, nxp - GOL_T * bx[h]);   // updated cells of this tile
      vy[h] = live[h] ? min(GOL_T, nyp - GOL_T * by[h]) : 0;
      cm[h] = GOLP_CM & ((1ull << (2 + 3 * vx[h])) - 1ull);
    }
    __syncthreads();
    // --- the two Life steps of the target rows r = 1 .. 48: colour cells cleared -> s_n0, set -> s_n1
    if (st_t < GOL_A) {
      // one lane per target row, both variants: of the rows r - 1, r, r + 1 exactly one holds colour cells, the
      // horizontal sums of the other two are shared
      const int r = 1 + st_t;
      const uint64_t cmh = cm[half];
      const int vyh = vy[half];
      GolpSum h0[3], h1[3];
      uint64_t mid0 = 0ull, mid1 = 0ull;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const int rr = r - 1 + d, q = rr - 2;
        const uint64_t row = s_cur[half][rr];
        const bool colour = q >= 0 && q % 3 == 0 && q / 3 < vyh;
        const uint64_t x0 = colour ? (row & ~cmh) : row, x1 = colour ? (row | cmh) : row;
        h0[d] = golp_hsum(x0);
        h1[d] = colour ? golp_hsum(x1) : h0[d];
        if (d == 1) {
          mid0 = x0;
          mid1 = x1;
        }
      }
      const uint64_t nx0 = golp_rule(h0[0], h0[1], h0[2], mid0), nx1 = golp_rule(h1[0], h1[1], h1[2], mid1);
      s_n0[half][r][0] = (uint32_t)nx0;
      s_n0[half][r][1] = (uint32_t)(nx0 >> 18);
      s_n1[half][r][0] = (uint32_t)nx1;
      s_n1[half][r][1] = (uint32_t)(nx1 >> 18);
    }
    __syncthreads();
    // --- one updated cell of each tile per thread
    bool bit_new[2] = {false, false};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int a = GOL_T * by[h] + ty, b = GOL_T * bx[h] + tx;
      const int i = oy + 3 * a, j = ox + 3 * b;
      if (ty < vy[h] && tx < vx[h]) {
        const int r0 = 3 * ty + 1;
        const int view = tx > 9, o = 3 * tx + 1 - 18 * view;   // columns o .. o + 2 of the view (<= 30)
        int cf[3], c0[3], c1[3];   // row codes: target bits | outcomes << 3
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const int f = (int)(s_fut[h][r0 + d][view] >> o) & 7;
          cf[d] = f;
          c0[d] = f | ((int)((s_n0[h][r0 + d][view] >> o) & 7u) << 3);
          c1[d] = f | ((int)((s_n1[h][r0 + d][view] >> o) & 7u) << 3);
        }
        float s0, s1;
        const bool interior = (i >= 1) & (i + 1 < n) & (j >= 1) & (j + 1 < n);
        if (interior) {
          s0 = s_lut6[c0[0] | (c0[1] << 6)];
          s1 = s_lut6[c1[0] | (c1[1] << 6)];
          // the last three terms from the four table values in scalar registers
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const bool fb = (cf[2] >> k) & 1;
            const float dead = fb ? tab.t[1][0] : tab.t[0][0], alive = fb ? tab.t[1][1] : tab.t[0][1];
            s0 += ((c0[2] >> (3 + k)) & 1) ? alive : dead;
            s1 += ((c1[2] >> (3 + k)) & 1) ? alive : dead;
          }
        } else {
          s0 = 0.0f;
          s1 = 0.0f;
#pragma unroll
          for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              const int fy = i + d - 1, fx = j + k - 1;   // mask of get_gibbs_probs_fast (core.py:141-149)
              const bool in_range = (fy >= 0) & (fy < n) & (fx >= 0) & (fx < n);
              s0 += in_range ? s_tail[c0[d]][k] : 0.0f;
              s1 += in_range ? s_tail[c1[d]][k] : 0.0f;
            }
        }
        // softmax(scores)[0]; exp(0) = 1 exactly, so only the smaller score needs the exponential
        const float m = fmaxf(s0, s1);
        const float e = expf(fminf(s0, s1) - m);
        const float e0 = (s0 == m) ? 1.0f : e, e1 = (s1 == m) ? 1.0f : e;
        const float p_zero = __fdiv_rn(e0, gjx_add_rn(e0, e1));
        const float p_one = gjx_sub_rn(1.0f, p_zero);
        const uint32_t k0 = pass_keys[2 * chain[h]], k1 = pass_keys[2 * chain[h] + 1];
        // draw index a * nxp + b < 2^32 for every board the entry point admits (n <= 32768)
        const float u = gjx_uniform_at(k0, k1, (uint64_t)((uint32_t)a * (uint32_t)nxp + (uint32_t)b), 0.0f, 1.0f);
        bit_new[h] = u < p_one;
      }
    }
    // --- write back: the 16 new bits of a row of updated cells as one XOR of what changed
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const unsigned long long bal = __ballot(bit_new[h]);
      if (tx == 0 && ty < vy[h]) {
        const uint64_t new16 = (bal >> (16 * (ty & 3))) & 0xffffull;
        const uint64_t fresh = golp_spread3(new16) << 2;
        const uint64_t diff = (fresh ^ s_cur[h][3 * ty + 2]) & cm[h];
        if (diff) {
          const int i = oy + 3 * (GOL_T * by[h] + ty);
          const int x0 = ox + GOL_A * bx[h] - 2, w = x0 >> 6, sh = x0 & 63;
          unsigned long long* row = reinterpret_cast<unsigned long long*>(boards + (int64_t)chain[h] * cells + (int64_t)i * wp);
          const uint64_t lo = diff << sh, hi = sh ? (diff >> (64 - sh)) : 0ull;
          if (lo && w >= 0 && w < wp) atomicXor(row + w, (unsigned long long)lo);
          if (hi && w + 1 >= 0 && w + 1 < wp) atomicXor(row + w + 1, (unsigned long long)hi);
        }
      }
    }
    __syncthreads();   // the staged rows are reused by the next pair of tiles
  }
}

// bytes <-> bits (row pitch wp words; bits beyond n are zero)
__global__ void __launch_bounds__(GJX_BLOCK)
    k_gol_pack(const uint8_t* __restrict__ bytes, int n, int wp, int64_t n_words, uint64_t* __restrict__ bits) {
  for (int64_t wi = (int64_t)blockIdx.x * GJX_BLOCK + threadIdx.x; wi < n_words; wi += (int64_t)gridDim.x * GJX_BLOCK) {
    const int64_t row = wi / wp;   // chain * n + y
    const int w = (int)(wi - row * wp);
    const uint8_t* src = bytes + row * n + 64 * (int64_t)w;
    const int cnt = min(64, n - 64 * w);
    uint64_t v = 0ull;
    if (cnt == 64 && ((((uintptr_t)src) & 7u) == 0)) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const uint64_t x = reinterpret_cast<const uint64_t*>(src)[q];
#pragma unroll
        for (int k = 0; k < 8; ++k) v |= (uint64_t)(((x >> (8 * k)) & 0xffull) != 0ull) << (8 * q + k);
      }
    } else {
      for (int k = 0; k < cnt; ++k) v |= (uint64_t)(src[k] != 0) << k;
    }
    bits[wi] = v;
  }
}

__global__ void __launch_bounds__(GJX_BLOCK)
    k_gol_unpack(const uint64_t* __restrict__ bits, int n, int wp, int64_t n_words, uint8_t* __restrict__ bytes) {
  for (int64_t wi = (int64_t)blockIdx.x * GJX_BLOCK + threadIdx.x; wi < n_words; wi += (int64_t)gridDim.x * GJX_BLOCK) {
    const int64_t row = wi / wp;
    const int w = (int)(wi - row * wp);
    uint8_t* dst = bytes + row * n + 64 * (int64_t)w;
    const int cnt = min(64, n - 64 * w);
    const uint64_t v = bits[wi];
    if (cnt == 64 && ((((uintptr_t)dst) & 7u) == 0)) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        uint64_t x = 0ull;
#pragma unroll
        for (int k = 0; k < 8; ++k) x |= ((v >> (8 * q + k)) & 1ull) << (8 * k);
        reinterpret_cast<uint64_t*>(dst)[q] = x;
      }
    } else {
      for (int k = 0; k < cnt; ++k) dst[k] = (uint8_t)((v >> k) & 1ull);
    }
  }
}

__global__ void __launch_bounds__(GJX_BLOCK)
    k_gol_init(const uint32_t* __restrict__ keys, int64_t cells, uint8_t* __restrict__ boards) {
  const int chain = blockIdx.y;
  const uint32_t k0 = keys[2 * chain], k1 = keys[2 * chain + 1];
  for (int64_t c = (int64_t)blockIdx.x * GJX_BLOCK + threadIdx.x; c < cells;
       c += (int64_t)gridDim.x * GJX_BLOCK)
    boards[(int64_t)chain * cells + c] =
        (gjx_uniform_at(k0, k1, (uint64_t)c, 0.0f, 1.0f) < 0.5f) ? (uint8_t)1 : (uint8_t)0;
}

__device__ __forceinline__ int gol_cell(const uint8_t* b, int n, int y, int x) {
  return (y >= 0 && y < n && x >= 0 && x < n) ? b[(int64_t)y * n + x] : 0;
}

__global__ void __launch_bounds__(GJX_BLOCK)
    k_gol_step(const uint8_t* __restrict__ boards, int n, uint8_t* __restrict__ next) {
  const int chain = blockIdx.y;
  const uint8_t* b = boards + (int64_t)chain * n * n;
  const int64_t cells = (int64_t)n * n;
  for (int64_t c = (int64_t)blockIdx.x * GJX_BLOCK + threadIdx.x; c < cells;
       c += (int64_t)gridDim.x * GJX_BLOCK) {
    const int y = (int)(c / n), x = (int)(c - (int64_t)y * n);
    int nb = 0;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx)
        if (dy != 0 || dx != 0) nb += gol_cell(b, n, y + dy, x + dx);
    next[(int64_t)chain * cells + c] = (uint8_t)gol_rule(b[c], nb);
  }
}

// per chain: number of cells of step(board) that differ from the target, and the sum of the
// step bits' log-densities (predictive_posterior_score); block partials -> 64-bit / f64 atomics
__global__ void __launch_bounds__(GJX_BLOCK)
    k_gol_summary(GolTable tab, const uint8_t* __restrict__ boards, const uint8_t* __restrict__ future,
                  int64_t future_stride, int n, int64_t* __restrict__ n_incorrect,
                  double* __restrict__ pred_score) {
  const int chain = blockIdx.y;
  const uint8_t* b = boards + (int64_t)chain * n * n;
  const uint8_t* fut = future + (int64_t)chain * future_stride;
  const int64_t cells = (int64_t)n * n;
  int bad = 0;
  double sc = 0.0;
  for (int64_t c = (int64_t)blockIdx.x * GJX_BLOCK + threadIdx.x; c < cells;
       c += (int64_t)gridDim.x * GJX_BLOCK) {
    const int y = (int)(c / n), x = (int)(c - (int64_t)y * n);
    int nb = 0;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx)
        if (dy != 0 || dx != 0) nb += gol_cell(b, n, y + dy, x + dx);
    const int det = gol_rule(b[c], nb), bit = fut[c];
    bad += det != bit;
    sc += (double)tab.t[bit][det];
  }
  __shared__ int s_bad[GJX_BLOCK];
  __shared__ double s_sc[GJX_BLOCK];
  s_bad[threadIdx.x] = bad;
  s_sc[threadIdx.x] = sc;
  __syncthreads();
  for (int o = GJX_BLOCK / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      s_bad[threadIdx.x] += s_bad[threadIdx.x + o];
      s_sc[threadIdx.x] += s_sc[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    atomicAdd(reinterpret_cast<unsigned long long*>(n_incorrect) + chain, (unsigned long long)s_bad[0]);
    atomicAdd(pred_score + chain, s_sc[0]);
  }
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
static int32_t gol_check(const char* who, int64_t n_chains, int32_t n) {
  GJX_REQUIRE(n_chains >= 1 && n_chains <= 65535, GJX_ELIMIT, "%s: n_chains=%lld not in [1, 65535]", who,
              (long long)n_chains);
  GJX_REQUIRE(n >= 1 && n <= 32768, GJX_ELIMIT, "%s: board edge n=%d not in [1, 32768]", who, n);
  return GJX_OK;
}

GJX_API int32_t gjx_gol_init(const uint32_t* d_keys, int64_t n_chains, int32_t n, uint8_t* d_boards,
                             void* stream) {
  GJX_REQUIRE(d_keys && d_boards, GJX_EINVAL, "gjx_gol_init: null argument");
  int32_t rc = gol_check("gjx_gol_init", n_chains, n);
  if (rc != GJX_OK) return rc;
  const int64_t cells = (int64_t)n * n;
  dim3 grid((unsigned)gjx_grid_for(cells), (unsigned)n_chains);
  k_gol_init<<<grid, GJX_BLOCK, 0, (hipStream_t)stream>>>(d_keys, cells, d_boards);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

GJX_API int32_t gjx_gol_gibbs_pass(const uint32_t* d_pass_keys, int64_t n_chains, int32_t n,
                                   int32_t oy, int32_t ox, float p_flip, const uint8_t* d_future,
                                   int64_t future_stride, uint8_t* d_boards, void* stream) {
  GJX_REQUIRE(d_pass_keys && d_future && d_boards, GJX_EINVAL, "gjx_gol_gibbs_pass: null argument");
  int32_t rc = gol_check("gjx_gol_gibbs_pass", n_chains, n);
  if (rc != GJX_OK) return rc;
  GJX_REQUIRE(oy >= 0 && oy < 3 && ox >= 0 && ox < 3, GJX_EINVAL,
              "gjx_gol_gibbs_pass: colour offsets must be in [0, 3)");
  GJX_REQUIRE(p_flip > 0.0f && p_flip < 1.0f, GJX_EINVAL, "gjx_gol_gibbs_pass: p_flip must be in (0, 1)");
  GJX_REQUIRE(future_stride == 0 || future_stride == (int64_t)n * n, GJX_EINVAL,
              "gjx_gol_gibbs_pass: future_stride must be 0 (shared target) or n*n");
  const int nyp = (n - oy + 2) / 3, nxp = (n - ox + 2) / 3;  // len(arange(o, n, 3))
  if (nyp == 0 || nxp == 0) return GJX_OK;
  dim3 grid((unsigned)((nxp + GOL_T - 1) / GOL_T), (unsigned)((nyp + GOL_T - 1) / GOL_T),
            (unsigned)n_chains);
  const bool vec = (n % 4 == 0) && ((((uintptr_t)d_boards) | ((uintptr_t)d_future)) & 3u) == 0;
  if (vec)
    k_gol_gibbs_pass<true><<<grid, GJX_BLOCK, 0, (hipStream_t)stream>>>(
        gol_table(p_flip), d_pass_keys, n, oy, ox, nyp, nxp, d_future, future_stride, d_boards);
  else
    k_gol_gibbs_pass<false><<<grid, GJX_BLOCK, 0, (hipStream_t)stream>>>(
        gol_table(p_flip), d_pass_keys, n, oy, ox, nyp, nxp, d_future, future_stride, d_boards);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

GJX_API int32_t gjx_gol_sweep(const uint32_t* d_pass_keys9, int64_t n_chains, int32_t n, float p_flip,
                              const uint8_t* d_future, int64_t future_stride, uint8_t* d_boards,
                              void* stream) {
  GJX_REQUIRE(d_pass_keys9 != nullptr, GJX_EINVAL, "gjx_gol_sweep: null argument");
  for (int oy = 0; oy < 3; ++oy)
    for (int ox = 0; ox < 3; ++ox) {
      int32_t rc = gjx_gol_gibbs_pass(d_pass_keys9 + (int64_t)(3 * oy + ox) * n_chains * 2, n_chains, n,
                                      oy, ox, p_flip, d_future, future_stride, d_boards, stream);
      if (rc != GJX_OK) return rc;
    }
  return GJX_OK;
}

GJX_API int32_t gjx_gol_words_per_row(int32_t n) { return n >= 1 ? (n + 63) / 64 : 0; }

static int32_t gol_convert(const char* who, bool pack, const void* d_in, int64_t n_chains, int32_t n, void* d_out,
                           void* stream) {
  GJX_REQUIRE(d_in && d_out, GJX_EINVAL, "%s: null argument", who);
  int32_t rc = gol_check(who, n_chains, n);
  if (rc != GJX_OK) return rc;
  const int wp = (n + 63) / 64;
  const int64_t n_words = n_chains * (int64_t)n * wp;
  if (pack)
    k_gol_pack<<<gjx_grid_for(n_words), GJX_BLOCK, 0, (hipStream_t)stream>>>((const uint8_t*)d_in, n, wp, n_words,
                                                                             (uint64_t*)d_out);
  else
    k_gol_unpack<<<gjx_grid_for(n_words), GJX_BLOCK, 0, (hipStream_t)stream>>>((const uint64_t*)d_in, n, wp, n_words,
                                                                               (uint8_t*)d_out);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

GJX_API int32_t gjx_gol_pack(const uint8_t* d_boards, int64_t n_chains, int32_t n, uint64_t* d_bits, void* stream) {
  return gol_convert("gjx_gol_pack", true, d_boards, n_chains, n, d_bits, stream);
}

GJX_API int32_t gjx_gol_unpack(const uint64_t* d_bits, int64_t n_chains, int32_t n, uint8_t* d_boards, void* stream) {
  return gol_convert("gjx_gol_unpack", false, d_bits, n_chains, n, d_boards, stream);
}

GJX_API int32_t gjx_gol_gibbs_pass_packed(const uint32_t* d_pass_keys, int64_t n_chains, int32_t n, int32_t oy,
                                          int32_t ox, float p_flip, const uint64_t* d_future_bits,
                                          int64_t future_stride_words, uint64_t* d_board_bits, void* stream) {
  GJX_REQUIRE(d_pass_keys && d_future_bits && d_board_bits, GJX_EINVAL, "gjx_gol_gibbs_pass_packed: null argument");
  int32_t rc = gol_check("gjx_gol_gibbs_pass_packed", n_chains, n);
  if (rc != GJX_OK) return rc;
  GJX_REQUIRE(oy >= 0 && oy < 3 && ox >= 0 && ox < 3, GJX_EINVAL,
              "gjx_gol_gibbs_pass_packed: colour offsets must be in [0, 3)");
  GJX_REQUIRE(p_flip > 0.0f && p_flip < 1.0f, GJX_EINVAL, "gjx_gol_gibbs_pass_packed: p_flip must be in (0, 1)");
  const int wp = (n + 63) / 64;
  GJX_REQUIRE(future_stride_words == 0 || future_stride_words == (int64_t)n * wp, GJX_EINVAL,
              "gjx_gol_gibbs_pass_packed: future_stride_words must be 0 (shared target) or n * ceil(n / 64)");
  const int nyp = (n - oy + 2) / 3, nxp = (n - ox + 2) / 3;  // len(arange(o, n, 3))
  if (nyp == 0 || nxp == 0) return GJX_OK;
  const int tiles_x = (nxp + GOL_T - 1) / GOL_T, tiles_y = (nyp + GOL_T - 1) / GOL_T;
  const int64_t n_tiles = n_chains * (int64_t)tiles_x * tiles_y;
  const int64_t cap = 256 * 8;   // blocks walk over pairs of tiles: the per-block table build is amortised
  const int64_t want = (n_tiles + 1) / 2;
  const int grid = (int)(want < cap ? want : cap);
  k_gol_pass_packed<<<grid, GJX_BLOCK, 0, (hipStream_t)stream>>>(gol_table(p_flip), d_pass_keys, n, wp, oy, ox, nyp, nxp,
                                                                tiles_x, tiles_y, n_tiles, d_future_bits,
                                                                future_stride_words, d_board_bits);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

GJX_API int32_t gjx_gol_sweep_packed(const uint32_t* d_pass_keys9, int64_t n_chains, int32_t n, float p_flip,
                                     const uint64_t* d_future_bits, int64_t future_stride_words,
                                     uint64_t* d_board_bits, void* stream) {
  GJX_REQUIRE(d_pass_keys9 != nullptr, GJX_EINVAL, "gjx_gol_sweep_packed: null argument");
  for (int oy = 0; oy < 3; ++oy)
    for (int ox = 0; ox < 3; ++ox) {
      int32_t rc = gjx_gol_gibbs_pass_packed(d_pass_keys9 + (int64_t)(3 * oy + ox) * n_chains * 2, n_chains, n, oy, ox,
                                             p_flip, d_future_bits, future_stride_words, d_board_bits, stream);
      if (rc != GJX_OK) return rc;
    }
  return GJX_OK;
}

GJX_API int32_t gjx_gol_step(const uint8_t* d_boards, int64_t n_chains, int32_t n, uint8_t* d_next,
                             void* stream) {
  GJX_REQUIRE(d_boards && d_next && d_boards != d_next, GJX_EINVAL,
              "gjx_gol_step: null argument or in-place step");
  int32_t rc = gol_check("gjx_gol_step", n_chains, n);
  if (rc != GJX_OK) return rc;
  dim3 grid((unsigned)gjx_grid_for((int64_t)n * n), (unsigned)n_chains);
  k_gol_step<<<grid, GJX_BLOCK, 0, (hipStream_t)stream>>>(d_boards, n, d_next);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

GJX_API int32_t gjx_gol_summary(const uint8_t* d_boards, const uint8_t* d_future, int64_t future_stride,
                                int64_t n_chains, int32_t n, float p_flip, int64_t* d_n_incorrect,
                                double* d_pred_score, void* stream) {
  GJX_REQUIRE(d_boards && d_future && d_n_incorrect && d_pred_score, GJX_EINVAL,
              "gjx_gol_summary: null argument");
  int32_t rc = gol_check("gjx_gol_summary", n_chains, n);
  if (rc != GJX_OK) return rc;
  GJX_REQUIRE(p_flip > 0.0f && p_flip < 1.0f, GJX_EINVAL, "gjx_gol_summary: p_flip must be in (0, 1)");
  GJX_REQUIRE(future_stride == 0 || future_stride == (int64_t)n * n, GJX_EINVAL,
              "gjx_gol_summary: future_stride must be 0 (shared target) or n*n");
  hipStream_t s = (hipStream_t)stream;
  GJX_CHECK_HIP(hipMemsetAsync(d_n_incorrect, 0, sizeof(int64_t) * n_chains, s));
  GJX_CHECK_HIP(hipMemsetAsync(d_pred_score, 0, sizeof(double) * n_chains, s));
  int gx = gjx_grid_for((int64_t)n * n);
  if (gx > 256) gx = 256;
  dim3 grid((unsigned)gx, (unsigned)n_chains);
  k_gol_summary<<<grid, GJX_BLOCK, 0, s>>>(gol_table(p_flip), d_boards, d_future, future_stride, n,
                                           d_n_incorrect, d_pred_score);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

// Host: the Seed key schedule of seed(_run_sampler_and_get_summary)(key, sampler)
// (core.py:312-364; pjax.py:1333-1389) for one chain: init site key and the nine pass keys of
// every sweep.  pass_keys: n_frames x per_frame x 9 x 2.
GJX_API void gjx_gol_keys(const uint32_t key[2], int32_t n_frames, int32_t per_frame,
                          uint32_t init_key[2], uint32_t* pass_keys) {
  uint32_t st[2] = {key[0], key[1]}, nx[2], s_outer[2];
  gjx_tf2x32(st[0], st[1], 0u, 1u, init_key[0], init_key[1]);  // S -> init flips
  gjx_tf2x32(st[0], st[1], 0u, 0u, nx[0], nx[1]);
  gjx_tf2x32(nx[0], nx[1], 0u, 1u, s_outer[0], s_outer[1]);    // S -> outer scan
  for (int f = 0; f < n_frames; ++f) {
    uint32_t kf[2], s_inner[2];
    gjx_tf2x32(s_outer[0], s_outer[1], 0u, (uint32_t)f, kf[0], kf[1]);  // fold_in(., f)
    gjx_tf2x32(kf[0], kf[1], 0u, 1u, s_inner[0], s_inner[1]);           // S -> inner scan
    for (int g = 0; g < per_frame; ++g) {
      uint32_t ks[2];
      gjx_tf2x32(s_inner[0], s_inner[1], 0u, (uint32_t)g, ks[0], ks[1]);  // fold_in(., g)
      for (int c = 0; c < 9; ++c) {
        uint32_t* out = pass_keys + (((int64_t)f * per_frame + g) * 9 + c) * 2;
        gjx_tf2x32(ks[0], ks[1], 0u, 1u, out[0], out[1]);  // S -> colour pass c
        gjx_tf2x32(ks[0], ks[1], 0u, 0u, nx[0], nx[1]);
        ks[0] = nx[0];
        ks[1] = nx[1];
      }
    }
  }
}
