// gjx_loc.hip -- fused kernels for the localization model family
// (examples/localization/core.py:81-128, 252-395) behind init/extend (smc.py:272-371, 449-530),
// mh (mcmc.py:267-309) and hmc (mcmc.py:434-578).
//
// Design (gfx950):
//  * one thread = one particle, 256-thread blocks, grid-stride over <= 2048 blocks;
//  * the world (walls, ray table, observation) is wave-uniform and travels in the kernel
//    arguments, so it lives in SGPRs (s_load) and costs no VGPRs or HBM traffic;
//  * lidar runs walls outer / rays inner with the per-ray state in VGPRs.  When the ray count
//    is even, rays r and r + R/2 point in exactly opposite directions, so ONE ray-segment
//    solve (denominator, reciprocal, t, s) serves both: the opposite ray's parameters are
//    (-t, s).  That halves the dominant loop;
//  * ray directions come from ONE sincos(theta) per pose plus the angle-addition formulas with a
//    host-built table of (cos a_r, sin a_r); reciprocals are v_rcp_f32.  Both stay within a
//    few ulp of the reference's cos(theta + a_r) / IEEE division, far inside the 1e-5 relative
//    tolerance on log-weights (the reference's own f32 rounding of theta + a_r is 2.4e-7);
//  * PRNG draws are counter-based on the global particle index: no HBM traffic.
// Algorithmic HBM bytes per particle: extend 44 B, mh/hmc 36 B for ALL fused moves.  The
// kernels are VALU-bound (finding 5).
#include "gjx_common.h"

struct LocK {
  float width, height, xlo, xhi, ylo, yhi;
  int W, R, RH, paired;       // RH = rays solved explicitly (R/2 when paired, else R)
  // all walls, reordered: the NO oblique (or zero-length) walls first, then the axis-aligned ones
  float x1[GJX_MAX_WALLS], y1[GJX_MAX_WALLS], wdx[GJX_MAX_WALLS], wdy[GJX_MAX_WALLS];
  // Axis-aligned walls grouped into lines: nvl vertical lines x = lc[l] (l < nvl) followed by nhl
  // horizontal lines y = lc[l]; line l owns lns[l] consecutive segments [slo, slo + slen] along
  // the line (segments stored in line order).  thr_c / thr_s: |cos| / |sin| of a ray below which
  // the reference's denominator clamp could fire for some vertical / horizontal wall (=> exact
  // path).
  int NO, nvl, nhl;
  float thr_c, thr_s;
  float lc[GJX_MAX_WALLS], slo[GJX_MAX_WALLS], slen[GJX_MAX_WALLS];
  int lns[GJX_MAX_WALLS];
  float ca[GJX_MAX_RAYS], sa[GJX_MAX_RAYS];  // cos / sin of linspace(0, 2pi, R, endpoint=False)
  float obs_s[GJX_MAX_RAYS];                 // obs_r / sigma
  float inv_sigma, lp_const, max_range;
  float mean0[3];             // initial means (core.py:331-333)
  float sig[3];               // prior sigmas of THIS step (core.py:351-353)
  int is_initial;
};

// Waves per SIMD by ray slots P (template parameter of the kernel the attribute sits on): the tuned value for the 4-
// and 8-slot kernels (8 / 16 rays); the 16- and 32-slot kernels get 3 / 2 waves = 168 / 256 registers.  At 5 waves
// they spilled 0.6-1.8 KB per lane, and hipcc 7.2 placed some of that spill code at the top of a divergent join, in
// front of the s_or_b64 that restores exec (reached with exec = 0 when no lane took the branch): k_loc_extend<16>
// lost its particle index that way and faulted.  tests/test_kernel_resources.py scans every built kernel for the pattern.
#define GJX_WAVES_FOR(P, N) ((P) <= 8 ? (N) : ((P) <= 16 ? 3 : 2))

// Register budget of the MH kernel: 5 waves per SIMD (<= 96 VGPRs).  Measured at N = 2^24,
// K = 10 on MI355X: default allocation (101 VGPRs, 4 waves) 3.56 ms, 5 waves 3.31 ms, 6 waves
// (56 B/lane of scratch) 3.27 ms.
#ifndef GJX_MH_WAVES_N
#define GJX_MH_WAVES_N 5
#endif
#define GJX_MH_WAVES __attribute__((amdgpu_waves_per_eu(GJX_WAVES_FOR(P, GJX_MH_WAVES_N), GJX_WAVES_FOR(P, GJX_MH_WAVES_N))))

// Register budget of the extend kernel (0 = the compiler's choice, 99 VGPRs = 4 waves; sweep in profiles/r02_occupancy_sweeps.txt).
#ifndef GJX_EXT_WAVES_N
#define GJX_EXT_WAVES_N 5
#endif
#if GJX_EXT_WAVES_N > 0
#define GJX_EXT_WAVES __attribute__((amdgpu_waves_per_eu(GJX_WAVES_FOR(P, GJX_EXT_WAVES_N), GJX_WAVES_FOR(P, GJX_EXT_WAVES_N))))
#else
#define GJX_EXT_WAVES
#endif

// Register budget of the HMC kernels (sweep in profiles/r02_hmc_waves_sweep.txt).
#ifndef GJX_HMC_WAVES_N
#define GJX_HMC_WAVES_N 5
#endif
#if GJX_HMC_WAVES_N > 0
#define GJX_HMC_WAVES __attribute__((amdgpu_waves_per_eu(GJX_WAVES_FOR(P, GJX_HMC_WAVES_N), GJX_WAVES_FOR(P, GJX_HMC_WAVES_N))))
#else
#define GJX_HMC_WAVES
#endif

struct MoveKeys {
  uint32_t k[GJX_MAX_MOVES][4][2];
};

static int32_t build_lock(const gjx_loc_world* w, const float* obs, int is_initial, LocK* k,
                          const char* who) {
  GJX_REQUIRE(w != nullptr, GJX_EINVAL, "%s: world is null", who);
  GJX_REQUIRE(w->n_walls >= 1 && w->n_walls <= GJX_MAX_WALLS, GJX_ELIMIT,
              "%s: n_walls=%d not in [1, %d]", who, w->n_walls, GJX_MAX_WALLS);
  GJX_REQUIRE(w->n_rays >= 1 && w->n_rays <= GJX_MAX_RAYS, GJX_ELIMIT,
              "%s: n_rays=%d not in [1, %d]", who, w->n_rays, GJX_MAX_RAYS);
  GJX_REQUIRE(w->sensor_sigma > 0.0f, GJX_EINVAL, "%s: sensor_sigma must be positive", who);
  memset(k, 0, sizeof(*k));
  k->width = w->width;
  k->height = w->height;
  k->xlo = w->clip_margin;
  k->xhi = w->width - w->clip_margin;
  k->ylo = w->clip_margin;
  k->yhi = w->height - w->clip_margin;
  k->W = w->n_walls;
  k->R = w->n_rays;
  k->paired = (w->n_rays % 2 == 0) ? 1 : 0;
  k->RH = k->paired ? w->n_rays / 2 : w->n_rays;
  // wall classes: 0 oblique or zero-length (general solve), 1 vertical, 2 horizontal
  int cls[GJX_MAX_WALLS], order[GJX_MAX_WALLS], no = 0;
  for (int i = 0; i < w->n_walls; ++i) {
    const float dx = w->walls[i][2] - w->walls[i][0], dy = w->walls[i][3] - w->walls[i][1];
    cls[i] = (dx == 0.0f && dy != 0.0f) ? 1 : ((dy == 0.0f && dx != 0.0f) ? 2 : 0);
  }
  for (int c = 0; c < 3; ++c)
    for (int i = 0; i < w->n_walls; ++i)
      if (cls[i] == c) order[no++] = i;
  k->NO = 0;
  for (int j = 0; j < w->n_walls; ++j) {
    const int i = order[j];
    k->x1[j] = w->walls[i][0];
    k->y1[j] = w->walls[i][1];
    k->wdx[j] = w->walls[i][2] - w->walls[i][0];
    k->wdy[j] = w->walls[i][3] - w->walls[i][1];
    if (cls[i] == 0) k->NO++;
  }
  // lines: group the vertical walls by x, then the horizontal walls by y
  int nl = 0, ns = 0;
  float min_len[3] = {0.0f, 0.0f, 0.0f};
  for (int c = 1; c <= 2; ++c) {
    bool used[GJX_MAX_WALLS] = {false};
    for (int i = 0; i < w->n_walls; ++i) {
      if (cls[i] != c || used[i]) continue;
      const float coord = (c == 1) ? w->walls[i][0] : w->walls[i][1];
      k->lc[nl] = coord;
      k->lns[nl] = 0;
      for (int j = i; j < w->n_walls; ++j) {
        if (cls[j] != c || used[j]) continue;
        if (((c == 1) ? w->walls[j][0] : w->walls[j][1]) != coord) continue;
        used[j] = true;
        const float a = (c == 1) ? w->walls[j][1] : w->walls[j][0];
        const float b = (c == 1) ? w->walls[j][3] : w->walls[j][2];
        k->slo[ns] = fminf(a, b);
        k->slen[ns] = fabsf(b - a);
        if (min_len[c] == 0.0f || k->slen[ns] < min_len[c]) min_len[c] = k->slen[ns];
        ++ns;
        k->lns[nl]++;
      }
      ++nl;
    }
    if (c == 1) k->nvl = nl;
  }
  k->nhl = nl - k->nvl;
  // |c * wdy| < 1e-10 for some vertical wall  <=  |c| < 1e-10 / min |wdy|  (5 % margin)
  k->thr_c = min_len[1] > 0.0f ? 1.05e-10f / min_len[1] : 0.0f;
  k->thr_s = min_len[2] > 0.0f ? 1.05e-10f / min_len[2] : 0.0f;
  for (int r = 0; r < w->n_rays; ++r) {
    // angle a_r = 2 pi r / R (core.py:267, jnp.linspace endpoint=False), table in double
    const double a = 6.283185307179586476925286766559 * (double)r / (double)w->n_rays;
    k->ca[r] = (float)cos(a);
    k->sa[r] = (float)sin(a);
    k->obs_s[r] = obs ? obs[r] / w->sensor_sigma : 0.0f;
  }
  k->inv_sigma = 1.0f / w->sensor_sigma;
  k->lp_const = HALF_LOG_2PI_F + logf(w->sensor_sigma);
  k->max_range = w->max_range;
  for (int i = 0; i < 3; ++i) {
    k->mean0[i] = w->init_mean[i];
    k->sig[i] = is_initial ? w->init_sigma[i] : w->drift_sigma[i];
  }
  k->is_initial = is_initial;
  return GJX_OK;
}

__device__ __forceinline__ float clampf(float v, float lo, float hi) {
  return fminf(fmaxf(v, lo), hi);
}

__device__ __forceinline__ float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }

// sin/cos of a pose heading: Cody-Waite reduction by pi/2 (three fma terms) and the Cephes
// single-precision minimax polynomials; <= 1.5 ulp absolute for |x| <= 8192, library call beyond.
__device__ __forceinline__ void sincos_pose(float x, float& s, float& c) {
  const float kf = rintf(x * 0.63661977236758134308f);
  float r = fmaf(-kf, 0x1.921fb6p+0f, x);
  r = fmaf(-kf, -0x1.777a5cp-25f, r);
  r = fmaf(-kf, -0x1.ee59dap-50f, r);
  const float z = r * r;
  float ps = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
  ps = fmaf(ps, z, -1.6666654611e-1f);
  const float sr = fmaf(ps * z, r, r);
  float pc = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
  pc = fmaf(pc, z, 4.166664568298827e-2f);
  const float cr = fmaf(pc * z, z, fmaf(-0.5f, z, 1.0f));
  const int q = (int)kf;
  const float a = (q & 1) ? cr : sr, b = (q & 1) ? sr : cr;
  s = (q & 2) ? -a : a;
  c = ((q + 1) & 2) ? -b : b;
  // beyond the range of the reduction: the library call, behind a WAVE-UNIFORM branch (no wave of a sane run takes
  // it) with a select inside -- a divergent if / else here put a join in the middle of every kernel's particle loop
  const bool big = fabsf(x) > 8192.0f;
  if (__builtin_amdgcn_ballot_w64(big) != 0ull) {
    float s2, c2;
    sincosf(x, &s2, &c2);
    s = big ? s2 : s;
    c = big ? c2 : c;
  }
}

// Ray state of one pose.  P = rays solved explicitly; when PAIRED the opposite ray of slot r
// (ray r + RH) shares the solve.  bp = best distance of ray r; bm = best distance of the opposite
// ray (held NEGATED inside lidar_dev's wall loop).  kp/km: winner ids for the gradient:
// 0..W-1 wall, -1 right, -2 left, -3 top, -4 bottom, -5 range cap.
template <int P, bool PAIRED, bool GRAD>
struct Rays {
  float dx[P], dy[P], rc[P], rs[P], bp[P], bm[PAIRED ? P : 1];
  int kp[GRAD ? P : 1], km[(GRAD && PAIRED) ? P : 1];
};

// One ray solved exactly as the reference writes it (float compares, own denominator clamp).
__device__ __forceinline__ void ray_exact(const LocK& w, float px, float py, float c, float sn,
                                          float& best, int& kb) {
  const float ex = w.width - px, ey = w.height - py;
  const float inf = __builtin_inff();
  const float t_right = c > 0.0f ? ex * rcp_fast(c) : inf, t_left = c < 0.0f ? -px * rcp_fast(c) : inf;
  const float t_top = sn > 0.0f ? ey * rcp_fast(sn) : inf, t_bottom = sn < 0.0f ? -py * rcp_fast(sn) : inf;
  const float tx = fminf(t_right, t_left), ty = fminf(t_top, t_bottom);
  const float b = fminf(tx, ty);
  best = fminf(b, w.max_range);
  kb = !(b <= w.max_range) ? -5 : ((ty < tx) ? (sn > 0.0f ? -3 : -4) : (c > 0.0f ? -1 : -2));
  for (int k = 0; k < w.W; ++k) {
    const float wdx = w.wdx[k], wdy = w.wdy[k];
    const float twx = w.x1[k] - px, twy = w.y1[k] - py;
    float denom = c * wdy - sn * wdx;
    denom = (fabsf(denom) < 1e-10f) ? 1e-10f : denom;
    const float rinv = rcp_fast(denom);
    const float t = (twx * wdy - twy * wdx) * rinv;
    const float s = (twx * sn - twy * c) * rinv;
    const bool v = (s >= 0.0f) & (s <= 1.0f) & (t >= 0.0f) & (t < best);
    best = v ? t : best;
    kb = v ? k : kb;
  }
}

// distance_to_wall_lidar + compute_ray_wall_intersection (core.py:252-308, 81-128) for one
// clipped pose: fills bp (rays 0..RH-1) and bm (rays RH..2RH-1) with min(boundary, walls, range).
//
// Fast path.  The running minima are positive, so the reference's float tests become integer
// tests on the bit patterns (non-negative floats order like unsigned ints, negative floats like
// reversed signed ints):
//     0 <= s <= 1             <=>  bits(s) <=u bits(1.0f)
//     0 <= t <  best          <=>  bits(t) <u  bits(best)
//     0 <= -t < best_opposite <=>  (int)bits(t) < (int)bits(-best_opposite)
// which is 3 compares per ray PAIR instead of 6 (they differ from the float tests only for
// t or s exactly -0.0 / +0.0, i.e. a pose exactly on a wall line).  Lanes that do not meet
// the preconditions -- a clamped denominator (ray parallel to a wall within 1e-10) or a pose
// outside the world rectangle -- are redone with ray_exact().
// Running-minimum update of one ray slot with candidate parameter t (valid iff `on`).
// bits(t) is compared as an unsigned int against the ray's best (0 <= t < best) and as a signed
// int against the NEGATED best of the opposite ray (0 <= -t < best_opposite).  Without winner
// ids the update is a masked integer min: an off-segment candidate becomes 0x7FFFFFFF, which
// loses both the unsigned min (best <= +inf) and the signed min (INT_MAX).
template <int P, bool PAIRED, bool GRAD>
__device__ __forceinline__ void ray_update(Rays<P, PAIRED, GRAD>& q, int r, float t, bool on,
                                           int id) {
  const uint32_t tb = __float_as_uint(t);
  if (GRAD) {
    const bool vp = on & (tb < __float_as_uint(q.bp[r]));
    q.bp[r] = vp ? t : q.bp[r];
    q.kp[r] = vp ? id : q.kp[r];
    if (PAIRED) {
      const bool vm = on & ((int32_t)tb < __float_as_int(q.bm[r]));
      q.bm[r] = vm ? t : q.bm[r];
      q.km[r] = vm ? id : q.km[r];
    }
  } else {
    const uint32_t tm = on ? tb : 0x7FFFFFFFu;
    q.bp[r] = __uint_as_float(min(__float_as_uint(q.bp[r]), tm));
    if (PAIRED) q.bm[r] = __int_as_float(min(__float_as_int(q.bm[r]), (int32_t)tm));
  }
}

// distance_to_wall_lidar + compute_ray_wall_intersection (core.py:252-308, 81-128) for one
// clipped pose: fills bp (rays 0..RH-1) and bm (rays RH..2RH-1) with min(boundary, walls, range).
//
// Fast path.  The running minima are positive, so the reference's float tests become integer
// tests on the bit patterns (non-negative floats order like unsigned ints, negative floats like
// reversed signed ints):
//     0 <= s <= 1             <=>  bits(s) <=u bits(1.0f)
//     0 <= t <  best          <=>  bits(t) <u  bits(best)
//     0 <= -t < best_opposite <=>  (int)bits(t) < (int)bits(-best_opposite)
// (they differ from the float tests only for t or s exactly -0.0 / +0.0, i.e. a pose exactly on
// a wall line).
// Axis-aligned walls (every wall of the reference's basic world) are grouped into lines.  For
// the vertical line x = X the general solve t = A/denom, s = B/denom collapses to
//     t = (X - px) / c,     s * |wdy| = (py + t * sn) - y_lo,
// so a (line, ray pair) costs one multiply by the ray's reciprocal 1/c, and a (segment, ray
// pair) one fma and one integer compare against the segment length; no per-wall reciprocal.
// The winner of a vertical (horizontal) line has the same t-derivatives as the right/left
// (top/bottom) world edge, so the gradient path tags it with that edge's id.
// Lanes that do not meet the preconditions -- a ray so close to parallel to a wall that the
// reference's denominator clamp (1e-10) fires, or a pose outside the world rectangle -- are
// redone with ray_exact().
template <int P, bool PAIRED, bool GRAD, bool FULL>
__device__ __forceinline__ void lidar_dev(const LocK& w, float px, float py, float th,
                                          Rays<P, PAIRED, GRAD>& q) {
  float s0, c0;
  sincos_pose(th, s0, c0);
  const float ex = w.width - px, ey = w.height - py;
  bool exact = !(px > 0.0f) | !(py > 0.0f) | !(ex > 0.0f) | !(ey > 0.0f) | !(w.max_range > 0.0f);
  float hv[GRAD ? P : 1], hm[(GRAD && PAIRED) ? P : 1];   // GRAD: running minima of the horizontal class
#pragma unroll
  for (int r = 0; r < P; ++r) {
    if (FULL || r < w.RH) {
      const float c = c0 * w.ca[r] - s0 * w.sa[r];
      const float s = s0 * w.ca[r] + c0 * w.sa[r];
      q.dx[r] = c;
      q.dy[r] = s;
      const float rc = rcp_fast(c), rs = rcp_fast(s);
      q.rc[r] = rc;
      q.rs[r] = rs;
      exact |= (fabsf(c) < w.thr_c) | (fabsf(s) < w.thr_s);
      // inside the rectangle the hits of the two opposite edges have opposite signs: the
      // forward ray takes the positive one, the opposite ray minus the negative one
      const float tr = ex * rc, tl = -px * rc, tt = ey * rs, tb = -py * rs;
      if (GRAD) {
        // winner CLASSES instead of winner ids: a vertical line has the t-derivatives of the right / left world
        // edge, a horizontal line those of the top / bottom edge, so the gradient only asks which class won.  bp / bm
        // start as the VERTICAL class (edge), hv / hm as the HORIZONTAL class; the lines of a class run the id-free
        // masked integer minima into their class and one comparison per ray picks the class afterwards
        q.bp[r] = fmaxf(tr, tl);
        hv[r] = fmaxf(tt, tb);
        if (PAIRED) {
          q.bm[r] = fminf(tr, tl);
          hm[r] = fminf(tt, tb);
        }
      } else {
        q.bp[r] = fminf(fminf(fmaxf(tr, tl), fmaxf(tt, tb)), w.max_range);
        if (PAIRED) q.bm[r] = fmaxf(fmaxf(fminf(tr, tl), fminf(tt, tb)), -w.max_range);
      }
    }
  }
  // axis-aligned walls, grouped into lines
  // (a flat per-segment table without the inner loop cuts the scalar instructions by 60 % but
  // makes the compiler allocate > 128 VGPRs: measured 3.18 ms at 4 waves vs 3.22 ms here)
  int si = 0;
  // one line against every ray slot: T[r] = ray parameter of the line, ON[r] = the hit lies on
  // one of the line's wall segments
  // (the per-ray hit masks are accumulated as wave-uniform 64-bit lane masks: one s_or per
  // segment and ray instead of the exec-merge sequence a loop-carried bool compiles to)
#define GJX_LINE_HITS(L, T, ON, POS, OTHER, RCP, DIR)                                         \
  float T[P];                                                                                 \
  bool ON[P];                                                                                 \
  {                                                                                           \
    const float tw_ = w.lc[L] - (POS);                                                        \
    uint64_t m_[P];                                                                           \
    _Pragma("unroll") for (int r = 0; r < P; ++r) {                                           \
      T[r] = tw_ * q.RCP[r];                                                                  \
      m_[r] = 0ull;                                                                           \
    }                                                                                         \
    const int ns_ = w.lns[L];                                                                 \
    for (int j_ = 0; j_ < ns_; ++j_, ++si) {                                                  \
      const float d_ = (OTHER)-w.slo[si];                                                     \
      const uint32_t len_ = __float_as_uint(w.slen[si]);                                      \
      _Pragma("unroll") for (int r = 0; r < P; ++r) m_[r] |= __builtin_amdgcn_ballot_w64(     \
          __float_as_uint(fmaf(T[r], q.DIR[r], d_)) <= len_);                                 \
    }                                                                                         \
    _Pragma("unroll") for (int r = 0; r < P; ++r) ON[r] =                                     \
        __builtin_amdgcn_inverse_ballot_w64(m_[r]);                                           \
  }
  // without winner ids two lines share one 3-way integer min per running minimum
  // (v_min3_u32 / v_min3_i32); BP / BM name the arrays of running minima the update goes to
#define GJX_LINE_PAIR_UPDATE_TO(BP, BM, TA, OA, TB, OB)                                       \
  _Pragma("unroll") for (int r = 0; r < P; ++r) if (FULL || r < w.RH) {                       \
    const uint32_t ma_ = OA[r] ? __float_as_uint(TA[r]) : 0x7FFFFFFFu;                        \
    const uint32_t mb_ = OB[r] ? __float_as_uint(TB[r]) : 0x7FFFFFFFu;                        \
    BP[r] = __uint_as_float(min(min(__float_as_uint(BP[r]), ma_), mb_));                      \
    if (PAIRED)                                                                               \
      BM[r] = __int_as_float(min(min(__float_as_int(BM[r]), (int32_t)ma_), (int32_t)mb_));    \
  }
#define GJX_LINE_PAIR_UPDATE(TA, OA, TB, OB) GJX_LINE_PAIR_UPDATE_TO(q.bp, q.bm, TA, OA, TB, OB)
#define GJX_LINE_UPDATE_TO(BP, BM, T, ON)                                                     \
  _Pragma("unroll") for (int r = 0; r < P; ++r) if (FULL || r < w.RH) {                       \
    const uint32_t m_ = ON[r] ? __float_as_uint(T[r]) : 0x7FFFFFFFu;                          \
    BP[r] = __uint_as_float(min(__float_as_uint(BP[r]), m_));                                 \
    if (PAIRED) BM[r] = __int_as_float(min(__float_as_int(BM[r]), (int32_t)m_));              \
  }
  if constexpr (GRAD) {
    // vertical lines into (bp, bm), horizontal lines into (hv, hm); then per ray the nearer class wins (the vertical
    // one on a tie, as the reference's running minimum meets the right / left edge first) and the range cap is
    // applied: a class at exactly the cap keeps its derivative (`b <= max_range`, as the reference's clip passes the
    // gradient at the bound).  Same minima as a per-line select chain with ids, ~0.5 x its instructions; which of two
    // candidates at exactly the same distance gives the derivative is the only thing that can differ.
    const int nl = w.nvl + w.nhl;
    int l = 0;
    for (; l + 1 < w.nvl; l += 2) {
      GJX_LINE_HITS(l, ta, oa, px, py, rc, dy)
      GJX_LINE_HITS(l + 1, tb, ob, px, py, rc, dy)
      GJX_LINE_PAIR_UPDATE_TO(q.bp, q.bm, ta, oa, tb, ob)
    }
    for (; l < w.nvl; ++l) {
      GJX_LINE_HITS(l, t, on, px, py, rc, dy)
      GJX_LINE_UPDATE_TO(q.bp, q.bm, t, on)
    }
    for (; l + 1 < nl; l += 2) {
      GJX_LINE_HITS(l, ta, oa, py, px, rs, dx)
      GJX_LINE_HITS(l + 1, tb, ob, py, px, rs, dx)
      GJX_LINE_PAIR_UPDATE_TO(hv, hm, ta, oa, tb, ob)
    }
    for (; l < nl; ++l) {
      GJX_LINE_HITS(l, t, on, py, px, rs, dx)
      GJX_LINE_UPDATE_TO(hv, hm, t, on)
    }
#pragma unroll
    for (int r = 0; r < P; ++r)
      if (FULL || r < w.RH) {
        const bool horz = hv[r] < q.bp[r];
        const float b = horz ? hv[r] : q.bp[r];
        const bool capped = !(b <= w.max_range);
        q.bp[r] = capped ? w.max_range : b;
        q.kp[r] = capped ? -5 : (horz ? -3 : -1);
        if (PAIRED) {   // negated distances: the larger one is the nearer hit
          const bool horzm = hm[r] > q.bm[r];
          const float bn = horzm ? hm[r] : q.bm[r];
          const bool cappedm = !(bn >= -w.max_range);
          q.bm[r] = cappedm ? -w.max_range : bn;
          q.km[r] = cappedm ? -5 : (horzm ? -3 : -1);
        }
      }
  } else {
    int l = 0;
    for (; l + 1 < w.nvl; l += 2) {
      GJX_LINE_HITS(l, ta, oa, px, py, rc, dy)
      GJX_LINE_HITS(l + 1, tb, ob, px, py, rc, dy)
      GJX_LINE_PAIR_UPDATE(ta, oa, tb, ob)
    }
    for (; l < w.nvl; ++l) {
      GJX_LINE_HITS(l, t, on, px, py, rc, dy)
#pragma unroll
      for (int r = 0; r < P; ++r)
        if (FULL || r < w.RH) ray_update<P, PAIRED, GRAD>(q, r, t[r], on[r], -1);
    }
    const int nl = w.nvl + w.nhl;
    for (; l + 1 < nl; l += 2) {
      GJX_LINE_HITS(l, ta, oa, py, px, rs, dx)
      GJX_LINE_HITS(l + 1, tb, ob, py, px, rs, dx)
      GJX_LINE_PAIR_UPDATE(ta, oa, tb, ob)
    }
    for (; l < nl; ++l) {
      GJX_LINE_HITS(l, t, on, py, px, rs, dx)
#pragma unroll
      for (int r = 0; r < P; ++r)
        if (FULL || r < w.RH) ray_update<P, PAIRED, GRAD>(q, r, t[r], on[r], -3);
    }
  }
#undef GJX_LINE_HITS
#undef GJX_LINE_PAIR_UPDATE
#undef GJX_LINE_PAIR_UPDATE_TO
#undef GJX_LINE_UPDATE_TO
  // oblique walls: general ray-segment solve (the first NO walls)
  for (int k = 0; k < w.NO; ++k) {
    const float wdx = w.wdx[k], wdy = w.wdy[k];
    const float twx = w.x1[k] - px, twy = w.y1[k] - py;
    const float A = twx * wdy - twy * wdx;
#pragma unroll
    for (int r = 0; r < P; ++r) {
      if (FULL || r < w.RH) {
        const float denom = q.dx[r] * wdy - q.dy[r] * wdx;
        exact |= fabsf(denom) < 1e-10f;  // clamped in the reference: such lanes are redone below
        const float rinv = rcp_fast(denom);
        const float t = A * rinv;
        const float s = (twx * q.dy[r] - twy * q.dx[r]) * rinv;
        // opposite ray: denominator and s-numerator flip sign => (t, s) -> (-t, s)
        ray_update<P, PAIRED, GRAD>(q, r, t, __float_as_uint(s) <= 0x3F800000u, k);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < P; ++r)
    if (PAIRED && (FULL || r < w.RH)) q.bm[r] = -q.bm[r];
  if (exact) {
#pragma unroll
    for (int r = 0; r < P; ++r) {
      if (FULL || r < w.RH) {
        int kb;
        ray_exact(w, px, py, q.dx[r], q.dy[r], q.bp[r], kb);
        if (GRAD) q.kp[r] = kb;
        if (PAIRED) {
          ray_exact(w, px, py, -q.dx[r], -q.dy[r], q.bm[r], kb);
          if (GRAD) q.km[r] = kb;
        }
      }
    }
  }
}

// Per-ray sensor terms: e[r] = 0.5 z^2 with z = (obs - d)/sigma, so that
// log N(obs_r; d_r, sigma) = -e[r] - lp_const.  Returns sum_r log N.
template <int P, bool PAIRED, bool FULL>
__device__ __forceinline__ float meas_terms(const LocK& w, float x, float y, float th,
                                            float (&ep)[P], float (&em)[PAIRED ? P : 1]) {
  Rays<P, PAIRED, false> q;
  const float px = clampf(x, w.xlo, w.xhi), py = clampf(y, w.ylo, w.yhi);
  lidar_dev<P, PAIRED, false, FULL>(w, px, py, th, q);
  float sum = 0.0f;
#pragma unroll
  for (int r = 0; r < P; ++r) {
    if (FULL || r < w.RH) {
      const float z = w.obs_s[r] - q.bp[r] * w.inv_sigma;
      ep[r] = 0.5f * (z * z);
      sum += -ep[r] - w.lp_const;
      if (PAIRED) {
        const float zm = w.obs_s[r + w.RH] - q.bm[r] * w.inv_sigma;
        em[r] = 0.5f * (zm * zm);
        sum += -em[r] - w.lp_const;
      }
    }
  }
  return sum;
}

// One ray's share of d(sum_r log N(obs_r; t_r, sigma)) / d(px, py, phi): gd = dlogN/dt times the t-derivatives of
// the ray's winning candidate.  (c, s) = ray direction, rc = 1 / c and rs = 1 / s the reciprocals lidar_dev already
// formed (v_rcp is odd: the opposite ray passes -rc, -rs).
//   wall by id (k >= 0): t = A/D, dt/dpx = -wdy/D, dt/dpy = wdx/D, dt/dphi = t (s wdy + c wdx)/D
//   vertical class (k = -1, -2: right / left edge or a vertical wall line): dt/dpx = -1/c, dt/dphi = t s/c
//   horizontal class (k = -3, -4): dt/dpy = -1/s, dt/dphi = -t c/s;   range cap (k = -5): none
// Branch-free for the two classes (every wall of the reference's worlds): the class masks select gd or 0 and the
// reciprocal, no reciprocal of its own.  Walls that won by id (oblique walls, and every wall on the rare lanes
// ray_exact() redid) keep the general formula behind a branch that is skipped when no lane of the wave has one.
__device__ __forceinline__ void ray_grad_acc(const LocK& w, int k, float gd, float t, float c, float s, float rc,
                                             float rs, float& gpx, float& gpy, float& gth) {
  const bool vert = (k | 1) == -1, horz = (k | 1) == -3;
  const float gv = vert ? gd : 0.0f, gh = horz ? gd : 0.0f;
  gpx = fmaf(-gv, rc, gpx);                       // gd * (-1/c)
  gpy = fmaf(-gh, rs, gpy);                       // gd * (-1/s)
  const float dir = vert ? s * rc : (horz ? -c * rs : 0.0f);   // dt/dphi = t * dir (1 / s may be inf when neither won)
  gth = fmaf(gv + gh, t * dir, gth);
  // a wall by id: behind a WAVE-UNIFORM branch (no wave of the reference's worlds takes it away from the rare exact
  // lanes) with the lanes selected inside -- a divergent `if` per ray is 2 P joins in the middle of the gradient
  const bool by_id = k >= 0;
  if (__builtin_amdgcn_ballot_w64(by_id) != 0ull) {
    const int kk = by_id ? k : 0;
    const float wdx = w.wdx[kk], wdy = w.wdy[kk];
    float D = c * wdy - s * wdx;
    const bool tiny = fabsf(D) < 1e-10f;
    D = tiny ? 1e-10f : D;
    const float rD = rcp_fast(D);
    gpx += by_id ? gd * (-wdy * rD) : 0.0f;
    gpy += by_id ? gd * (wdx * rD) : 0.0f;
    gth += by_id ? gd * (tiny ? 0.0f : t * (s * wdy + c * wdx) * rD) : 0.0f;
  }
}

// assess() with all sites constrained + analytic gradient w.r.t. (x, y, theta): the function
// jax.grad differentiates in hmc (mcmc.py:59-88, 508, 523).  clip() passes gradient only inside
// the box, min() only to the winner, the range cap passes none.
// WANT_LOGP = false: the gradient alone (the leapfrog steps before the last one never use the density: mcmc.py:510-535
// evaluates only jax.grad there); the gradient's operations are the same either way.
template <int P, bool PAIRED, bool FULL, bool WANT_LOGP = true>
__device__ __forceinline__ float logp_grad(const LocK& w, float x, float y, float th, float mx,
                                           float my, float mth, const float (&inv_s)[3],
                                           const float (&log_s)[3], float (&g)[3]) {
  Rays<P, PAIRED, true> q;
  const float px = clampf(x, w.xlo, w.xhi), py = clampf(y, w.ylo, w.yhi);
  lidar_dev<P, PAIRED, true, FULL>(w, px, py, th, q);
  float meas = 0.0f, gpx = 0.0f, gpy = 0.0f, gth = 0.0f;
#pragma unroll
  for (int r = 0; r < P; ++r) {
    if (FULL || r < w.RH) {
      {
        const float t = q.bp[r];
        const float z = w.obs_s[r] - t * w.inv_sigma;
        if (WANT_LOGP) meas += -0.5f * (z * z) - w.lp_const;
        ray_grad_acc(w, q.kp[r], z * w.inv_sigma, t, q.dx[r], q.dy[r], q.rc[r], q.rs[r], gpx, gpy, gth);
      }
      if (PAIRED) {
        const float t = q.bm[r];
        const float z = w.obs_s[r + w.RH] - t * w.inv_sigma;
        if (WANT_LOGP) meas += -0.5f * (z * z) - w.lp_const;
        ray_grad_acc(w, q.km[r], z * w.inv_sigma, t, -q.dx[r], -q.dy[r], -q.rc[r], -q.rs[r], gpx, gpy, gth);
      }
    }
  }
  const float zx = (x - mx) * inv_s[0], zy = (y - my) * inv_s[1], zt = (th - mth) * inv_s[2];
  float logp = 0.0f;
  if (WANT_LOGP) {
    logp += -0.5f * (zx * zx) - (HALF_LOG_2PI_F + log_s[0]);
    logp += -0.5f * (zy * zy) - (HALF_LOG_2PI_F + log_s[1]);
    logp += -0.5f * (zt * zt) - (HALF_LOG_2PI_F + log_s[2]);
    logp += meas;
  }
  const bool in_x = (x >= w.xlo) & (x <= w.xhi), in_y = (y >= w.ylo) & (y <= w.yhi);
  g[0] = -zx * inv_s[0] + (in_x ? gpx : 0.0f);
  g[1] = -zy * inv_s[1] + (in_y ? gpy : 0.0f);
  g[2] = -zt * inv_s[2] + gth;
  return logp;
}

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------
#define GJX_FOR_PARTICLES(i, n)                                            \
  for (int64_t i = (int64_t)blockIdx.x * GJX_BLOCK + threadIdx.x; i < (n); \
       i += (int64_t)gridDim.x * GJX_BLOCK)

template <int P, bool PAIRED, bool FULL>
__global__ void __launch_bounds__(GJX_BLOCK)
    k_loc_lidar(LocK w, const float* __restrict__ px, const float* __restrict__ py,
                const float* __restrict__ th, int64_t n, float* __restrict__ out) {
  GJX_FOR_PARTICLES(i, n) {
    Rays<P, PAIRED, false> q;
    lidar_dev<P, PAIRED, false, FULL>(w, px[i], py[i], th[i], q);
#pragma unroll
    for (int r = 0; r < P; ++r)
      if (FULL || r < w.RH) {
        out[i * w.R + r] = q.bp[r];
        if (PAIRED) out[i * w.R + r + w.RH] = q.bm[r];
      }
  }
}

// Distances AND their partial derivatives w.r.t. (px, py, theta) per ray: the backward of distance_to_wall_lidar
// under autograd (generic hmc / mala on a model that calls it).  jac = 3 planes of n x R (d/dpx, d/dpy, d/dtheta);
// the derivative of a ray is that of its winning candidate (zero at the range cap), as jax.grad takes it through
// the reference's jnp.min / jnp.where (core.py:252-308).
template <int P, bool PAIRED, bool FULL>
__global__ void __launch_bounds__(GJX_BLOCK)
    k_loc_lidar_jac(LocK w, const float* __restrict__ px, const float* __restrict__ py,
                    const float* __restrict__ th, int64_t n, float* __restrict__ out, float* __restrict__ jac) {
  const int64_t plane = n * w.R;
  GJX_FOR_PARTICLES(i, n) {
    Rays<P, PAIRED, true> q;
    lidar_dev<P, PAIRED, true, FULL>(w, px[i], py[i], th[i], q);
#pragma unroll
    for (int r = 0; r < P; ++r)
      if (FULL || r < w.RH) {
        float gx = 0.0f, gy = 0.0f, gt = 0.0f;
        ray_grad_acc(w, q.kp[r], 1.0f, q.bp[r], q.dx[r], q.dy[r], q.rc[r], q.rs[r], gx, gy, gt);
        const int64_t o = i * w.R + r;
        out[o] = q.bp[r];
        jac[o] = gx;
        jac[plane + o] = gy;
        jac[2 * plane + o] = gt;
        if (PAIRED) {
          gx = gy = gt = 0.0f;
          ray_grad_acc(w, q.km[r], 1.0f, q.bm[r], -q.dx[r], -q.dy[r], -q.rc[r], -q.rs[r], gx, gy, gt);
          const int64_t om = o + w.RH;
          out[om] = q.bm[r];
          jac[om] = gx;
          jac[plane + om] = gy;
          jac[2 * plane + om] = gt;
        }
      }
  }
}

struct Key3 {
  uint32_t k[3][2];
};

__device__ __forceinline__ float draw_normal(const uint32_t (&k)[2], uint64_t gi, float sigma,
                                             float mean) {
  // tfd.Normal.sample: normal(key, shape) * scale + loc
  return gjx_add_rn(gjx_mul_rn(gjx_normal_at(k[0], k[1], gi), sigma), mean);
}

template <int P, bool PAIRED, bool FULL>
__global__ void __launch_bounds__(GJX_BLOCK) GJX_EXT_WAVES
    k_loc_extend(LocK w, Key3 keys, uint64_t offset, int64_t n, const float* px_in,
                 const float* py_in, const float* pth_in, const float* logw_in, float* x, float* y,
                 float* th, float* mx, float* my, float* mth, float* logw,
                 float* __restrict__ meas, int64_t meas_stride, float* __restrict__ max_partials,
                 float* __restrict__ max_cell) {
  float lw_max = -__builtin_inff();
  // the inputs of a thread's NEXT particle are loaded before the work on the current one: their latency hides behind
  // ~700 instructions instead of standing at the top of every iteration
  const int64_t stride = (int64_t)gridDim.x * GJX_BLOCK;
  const int64_t i_first = (int64_t)blockIdx.x * GJX_BLOCK + threadIdx.x;
  float nx_in = 0.0f, ny_in = 0.0f, nt_in = 0.0f, nlw_in = 0.0f;
  if (i_first < n) {
    if (!w.is_initial) {
      nx_in = px_in[i_first];
      ny_in = py_in[i_first];
      nt_in = pth_in[i_first];
    }
    nlw_in = logw_in[i_first];
  }
  GJX_FOR_PARTICLES(i, n) {
    const float cx_in = nx_in, cy_in = ny_in, ct_in = nt_in, lw_old = nlw_in;
    if (i + stride < n) {
      if (!w.is_initial) {
        nx_in = px_in[i + stride];
        ny_in = py_in[i + stride];
        nt_in = pth_in[i + stride];
      }
      nlw_in = logw_in[i + stride];
    }
    float m0, m1, m2;
    if (w.is_initial) {
      m0 = w.mean0[0];
      m1 = w.mean0[1];
      m2 = w.mean0[2];
    } else {
      m0 = clampf(cx_in, w.xlo, w.xhi);
      m1 = clampf(cy_in, w.ylo, w.yhi);
      m2 = ct_in;
    }
    const uint64_t gi = offset + (uint64_t)i;
    const float nx = draw_normal(keys.k[0], gi, w.sig[0], m0);
    const float ny = draw_normal(keys.k[1], gi, w.sig[1], m1);
    const float nt = draw_normal(keys.k[2], gi, w.sig[2], m2);
    // the six pose / mean stores go out BEFORE the lidar: the memory pipe gets them while the ALUs work, instead of
    // 19 stores per particle in one burst at the end
    x[i] = nx;
    y[i] = ny;
    th[i] = nt;
    mx[i] = m0;
    my[i] = m1;
    mth[i] = m2;
    float ep[P], em[PAIRED ? P : 1];
    const float wsum = meas_terms<P, PAIRED, FULL>(w, nx, ny, nt, ep, em);
    if (meas) {
      // the trace keeps the per-ray scores (Vmap sub-trace, core.py:1416-1492): ray r -> column r
#pragma unroll
      for (int r = 0; r < P; ++r)
        if (FULL || r < w.RH) {
          meas[(int64_t)r * meas_stride + i] = ep[r];
          if (PAIRED) meas[(int64_t)(r + w.RH) * meas_stride + i] = em[r];
        }
    }
    const float lw_new = lw_old + wsum;
    logw[i] = lw_new;
    lw_max = fmaxf(lw_max, lw_new);
  }
  // pass 1 of the normaliser (the block maxima of the new log-weights) rides on this kernel: the
  // fused step does not read the 4 N bytes again for it
  // (max_cell: straight into ONE running maximum by an atomic that most blocks skip, gjx_atomic_max_f32; the cell
  // is -inf on entry and the normaliser that reads it puts it back: no fold launch between the two)
  if (max_partials || max_cell) {
    __shared__ float smx[GJX_BLOCK / 64];
    lw_max = gjx_wave_max(lw_max);
    if ((threadIdx.x & 63) == 0) smx[threadIdx.x >> 6] = lw_max;
    __syncthreads();
    if (threadIdx.x == 0) {
      float r = smx[0];
      for (int wv = 1; wv < GJX_BLOCK / 64; ++wv) r = fmaxf(r, smx[wv]);
      if (max_partials) max_partials[blockIdx.x] = r;
      if (max_cell) gjx_atomic_max_f32(max_cell, r);
    }
  }
}

// Per-ray measurement scores of given poses, minus the constant: out[r * n + i] = 0.5 z_r^2 with
// z_r = (obs_r - lidar_r(pose_i)) / sigma (the column layout gjx_loc_extend_scores writes).
template <int P, bool PAIRED, bool FULL>
__global__ void __launch_bounds__(GJX_BLOCK)
    k_loc_ray_scores(LocK w, const float* __restrict__ x, const float* __restrict__ y,
                     const float* __restrict__ th, int64_t n, float* __restrict__ out, int64_t stride) {
  GJX_FOR_PARTICLES(i, n) {
    float ep[P], em[PAIRED ? P : 1];
    (void)meas_terms<P, PAIRED, FULL>(w, x[i], y[i], th[i], ep, em);
#pragma unroll
    for (int r = 0; r < P; ++r)
      if (FULL || r < w.RH) {
        out[(int64_t)r * stride + i] = ep[r];
        if (PAIRED) out[(int64_t)(r + w.RH) * stride + i] = em[r];
      }
  }
}

// jnp.linspace(lo, hi, g)[i]  [3P]: lo * (1 - i/div) + hi * (i/div) for i < div, hi at the end
__device__ __forceinline__ float linspace_at(float lo, float hi, int g, int i) {
  if (g == 1) return lo;
  if (i == g - 1) return hi;
  const float step = __fdiv_rn((float)i, (float)(g - 1));
  return gjx_add_rn(gjx_mul_rn(lo, gjx_sub_rn(1.0f, step)), gjx_mul_rn(hi, step));
}

struct GridK {
  int g;                       // grid points per dimension
  float radius, theta_radius;  // search radius in x / y, and in theta (pi / 6)
  float noise, noise_th;       // proposal sigmas around the selected grid pose
  float edge;                  // grid kept >= edge away from the world boundary (0.5)
};

// The end of the locally-optimal extend for one particle, given the selected grid pose (sx, sy, st): Gaussian noise
// around it, the proposal's score, the model's weight at the drawn pose, the stores.  Shared by both kernels below.
template <int P, bool PAIRED, bool FULL>
__device__ __forceinline__ void locopt_finish(const LocK& w, const Key3& keys, const GridK& gk, uint64_t gi, int64_t i,
                                              float sx, float sy, float st, float m0, float m1, float m2, float lw_old,
                                              const float (&inv_s)[3], const float (&log_s)[3], float* x, float* y,
                                              float* th, float* mx, float* my, float* mth, float* logw) {
  const float inv_q = 1.0f / gk.noise, inv_qt = 1.0f / gk.noise_th;
  const float log_q = logf(gk.noise), log_qt = logf(gk.noise_th);
  const float nx = draw_normal(keys.k[0], gi, gk.noise, sx);
  const float ny = draw_normal(keys.k[1], gi, gk.noise, sy);
  const float nt = draw_normal(keys.k[2], gi, gk.noise_th, st);
  const float qx = (nx - sx) * inv_q, qy = (ny - sy) * inv_q, qt = (nt - st) * inv_qt;
  const float pscore = -(((-0.5f * (qx * qx) - (HALF_LOG_2PI_F + log_q)) +
                          (-0.5f * (qy * qy) - (HALF_LOG_2PI_F + log_q))) +
                         (-0.5f * (qt * qt) - (HALF_LOG_2PI_F + log_qt)));
  const float zx = (nx - m0) * inv_s[0], zy = (ny - m1) * inv_s[1], zt = (nt - m2) * inv_s[2];
  float wt = ((-0.5f * (zx * zx) - (HALF_LOG_2PI_F + log_s[0])) +
              (-0.5f * (zy * zy) - (HALF_LOG_2PI_F + log_s[1]))) +
             (-0.5f * (zt * zt) - (HALF_LOG_2PI_F + log_s[2]));
  float ep[P], em[PAIRED ? P : 1];
  wt += meas_terms<P, PAIRED, FULL>(w, nx, ny, nt, ep, em);
  x[i] = nx;
  y[i] = ny;
  th[i] = nt;
  mx[i] = m0;
  my[i] = m1;
  mth[i] = m2;
  logw[i] = (lw_old + wt) + pscore;
}

// extend() with the locally-optimal grid proposal (examples/localization/core.py:730-809 via
// smc.py:493-509): assess() of the model at g^3 poses around the particle's previous choices,
// first argmax, Gaussian noise around it, weight = target weight + proposal score.  One thread
// = one particle; g^3 + 1 lidar evaluations per particle, VALU-bound like the MH kernel.
template <int P, bool PAIRED, bool FULL>
__global__ void __launch_bounds__(GJX_BLOCK)
    k_loc_extend_locopt(LocK w, Key3 keys, GridK gk, uint64_t offset, int64_t n, const float* px_in,
                        const float* py_in, const float* pth_in, const float* logw_in, float* x,
                        float* y, float* th, float* mx, float* my, float* mth, float* logw) {
  float inv_s[3], log_s[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    inv_s[k] = 1.0f / w.sig[k];
    log_s[k] = logf(w.sig[k]);
  }
  GJX_FOR_PARTICLES(i, n) {
    const float cx = px_in[i], cy = py_in[i], ct = pth_in[i];
    const float m0 = clampf(cx, w.xlo, w.xhi), m1 = clampf(cy, w.ylo, w.yhi), m2 = ct;
    const float lw_old = logw_in[i];
    const float xlo = fmaxf(gk.edge, cx - gk.radius), xhi = fminf(w.width - gk.edge, cx + gk.radius);
    const float ylo = fmaxf(gk.edge, cy - gk.radius), yhi = fminf(w.height - gk.edge, cy + gk.radius);
    const float tlo = ct - gk.theta_radius, thi = ct + gk.theta_radius;
    float best = -__builtin_inff(), sx = 0.0f, sy = 0.0f, st = 0.0f;
    bool have = false;
    for (int ix = 0; ix < gk.g; ++ix) {
      const float gx = linspace_at(xlo, xhi, gk.g, ix);
      const float zx = (gx - m0) * inv_s[0];
      const float lx = -0.5f * (zx * zx) - (HALF_LOG_2PI_F + log_s[0]);
      for (int iy = 0; iy < gk.g; ++iy) {
        const float gy = linspace_at(ylo, yhi, gk.g, iy);
        const float zy = (gy - m1) * inv_s[1];
        const float lxy = lx + (-0.5f * (zy * zy) - (HALF_LOG_2PI_F + log_s[1]));
        for (int it = 0; it < gk.g; ++it) {
          const float gt = linspace_at(tlo, thi, gk.g, it);
          const float zt = (gt - m2) * inv_s[2];
          float sc = lxy + (-0.5f * (zt * zt) - (HALF_LOG_2PI_F + log_s[2]));
          float ep[P], em[PAIRED ? P : 1];
          sc += meas_terms<P, PAIRED, FULL>(w, gx, gy, gt, ep, em);
          const bool better = !have | (sc > best);  // jnp.argmax: first maximum
          best = better ? sc : best;
          sx = better ? gx : sx;
          sy = better ? gy : sy;
          st = better ? gt : st;
          have = true;
        }
      }
    }
    locopt_finish<P, PAIRED, FULL>(w, keys, gk, offset + (uint64_t)i, i, sx, sy, st, m0, m1, m2, lw_old, inv_s, log_s, x,
                                   y, th, mx, my, mth, logw);
  }
}

// The same extend with ONE WORKGROUP per particle: the g^3 grid poses are spread over the 256 threads (the reference
// runs this proposal with 5 to a few thousand particles -- one thread per particle leaves the chip empty and walks
// up to 15 625 lidar solves in sequence).  Every pose is scored by the expressions of the kernel above; the argmax
// keeps jnp.argmax's "first maximum" (ties to the lower flat index ix g^2 + iy g + it, a NaN never wins unless
// it is pose 0, as in the sequential scan); thread 0 then finishes the particle with the same code.
template <int P, bool PAIRED, bool FULL>
__global__ void __launch_bounds__(GJX_BLOCK)
    k_loc_extend_locopt_coop(LocK w, Key3 keys, GridK gk, uint64_t offset, int64_t n, const float* px_in,
                             const float* py_in, const float* pth_in, const float* logw_in, float* x, float* y,
                             float* th, float* mx, float* my, float* mth, float* logw) {
  __shared__ float s_best[GJX_BLOCK];
  __shared__ int s_idx[GJX_BLOCK];
  __shared__ int s_first_nan;
  float inv_s[3], log_s[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    inv_s[k] = 1.0f / w.sig[k];
    log_s[k] = logf(w.sig[k]);
  }
  const int g = gk.g, g2 = g * g, g3 = g2 * g;
  for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
    const float cx = px_in[i], cy = py_in[i], ct = pth_in[i];
    const float m0 = clampf(cx, w.xlo, w.xhi), m1 = clampf(cy, w.ylo, w.yhi), m2 = ct;
    const float xlo = fmaxf(gk.edge, cx - gk.radius), xhi = fminf(w.width - gk.edge, cx + gk.radius);
    const float ylo = fmaxf(gk.edge, cy - gk.radius), yhi = fminf(w.height - gk.edge, cy + gk.radius);
    const float tlo = ct - gk.theta_radius, thi = ct + gk.theta_radius;
    float best = -__builtin_inff();
    int bidx = 0x7fffffff;
    for (int idx = threadIdx.x; idx < g3; idx += GJX_BLOCK) {
      const int ix = idx / g2, rem = idx - ix * g2, iy = rem / g, it = rem - iy * g;
      const float gx = linspace_at(xlo, xhi, g, ix);
      const float zx = (gx - m0) * inv_s[0];
      const float lx = -0.5f * (zx * zx) - (HALF_LOG_2PI_F + log_s[0]);
      const float gy = linspace_at(ylo, yhi, g, iy);
      const float zy = (gy - m1) * inv_s[1];
      const float lxy = lx + (-0.5f * (zy * zy) - (HALF_LOG_2PI_F + log_s[1]));
      const float gt = linspace_at(tlo, thi, g, it);
      const float zt = (gt - m2) * inv_s[2];
      float sc = lxy + (-0.5f * (zt * zt) - (HALF_LOG_2PI_F + log_s[2]));
      float ep[P], em[PAIRED ? P : 1];
      sc += meas_terms<P, PAIRED, FULL>(w, gx, gy, gt, ep, em);
      if (idx == 0) s_first_nan = (sc != sc) ? 1 : 0;
      // ascending idx within a thread: a strict > keeps the first maximum; the first finite-or-inf score is taken
      const bool better = (sc > best) | ((bidx == 0x7fffffff) & (sc == sc));
      best = better ? sc : best;
      bidx = better ? idx : bidx;
    }
    s_best[threadIdx.x] = best;
    s_idx[threadIdx.x] = bidx;
    __syncthreads();
    for (int o = GJX_BLOCK / 2; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) {
        const float b2 = s_best[threadIdx.x + o];
        const int i2 = s_idx[threadIdx.x + o];
        const float b1 = s_best[threadIdx.x];
        const int i1 = s_idx[threadIdx.x];
        // a candidate without a pose (index 0x7fffffff) never wins; equal scores: the lower flat index
        const bool take2 = (i2 != 0x7fffffff) & ((i1 == 0x7fffffff) | (b2 > b1) | ((b2 == b1) & (i2 < i1)));
        s_best[threadIdx.x] = take2 ? b2 : b1;
        s_idx[threadIdx.x] = take2 ? i2 : i1;
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      int win = s_idx[0];
      if (s_first_nan || win == 0x7fffffff) win = 0;   // the sequential scan starts from pose 0 and a NaN sticks
      const int ix = win / g2, rem = win - ix * g2, iy = rem / g, it = rem - iy * g;
      const float sx = linspace_at(xlo, xhi, g, ix), sy = linspace_at(ylo, yhi, g, iy), st = linspace_at(tlo, thi, g, it);
      locopt_finish<P, PAIRED, FULL>(w, keys, gk, offset + (uint64_t)i, i, sx, sy, st, m0, m1, m2, logw_in[i], inv_s,
                                     log_s, x, y, th, mx, my, mth, logw);
    }
    __syncthreads();
  }
}

// One proposal of mh(trace, selection) (mcmc.py:288-296): regenerate -- fresh prior draws of the
// selected sites at the stored args (core.py:1645-1648), unselected sites keep their value and
// contribute logpdf + old_score = 0 -- and its weight
//     wt = sum_r [logpdf_r(new) + old_score_r]   (core.py:1649-1657, 1474-1492)
// plus log(u) of the accept test.  Shared by the fused move kernel and gjx_loc_regenerate so that
// both evaluate the same expressions.
template <int P, bool PAIRED, bool FULL>
__device__ __forceinline__ void mh_propose(const LocK& w, const uint32_t (&k)[4][2], int sel_mask, uint64_t gi,
                                           float cx, float cy, float ct, float m0, float m1, float m2,
                                           const float (&ep_old)[P], const float (&em_old)[PAIRED ? P : 1],
                                           float& nx, float& ny, float& nt, float (&ep_new)[P],
                                           float (&em_new)[PAIRED ? P : 1], float& wt, float& log_u) {
  nx = cx;
  ny = cy;
  nt = ct;
  if (sel_mask & 1) nx = draw_normal(k[0], gi, w.sig[0], m0);
  if (sel_mask & 2) ny = draw_normal(k[1], gi, w.sig[1], m1);
  if (sel_mask & 4) nt = draw_normal(k[2], gi, w.sig[2], m2);
  const float u = gjx_uniform_at(k[3][0], k[3][1], gi, 0.0f, 1.0f);
  (void)meas_terms<P, PAIRED, FULL>(w, nx, ny, nt, ep_new, em_new);
  wt = 0.0f;
#pragma unroll
  for (int r = 0; r < P; ++r)
    if (FULL || r < w.RH) wt += ep_old[r] - ep_new[r];
  if (PAIRED) {
#pragma unroll
    for (int r = 0; r < P; ++r)
      if (FULL || r < w.RH) wt += em_old[r] - em_new[r];
  }
  log_u = __builtin_amdgcn_logf(u) * 0.69314718055994530942f;
}

// OVL (the overlapped sharded step, gjx_shard_moves_main / _fixup, and the one-GPU step with the decision on the
// device, gjx_loc_smc_step_auto).  Main launch (idx != null): the resampling
// decision is read from the device (split[2] == 0: every particle is its own ancestor and keeps its weight); the
// ancestor of a position comes straight from the owner-side search (idx / range), positions whose ancestor another
// rank owns are left to the fix-up launch, and the run [p0, p1) the host will be told is checked against what the
// particles see (split[3] |= 2).  Fix-up launch (idx == null): ancestors from `anc`, work item t -> position t or
// p1 + t - p0.  The default instantiation is the kernel as it was.
struct MhOverlap {
  int32_t* split;         // main launch: {p0 + 1 (0: no own-sourced position), p1, go, flags} on the device
  const int32_t* idx;     // main launch: ancestors of the candidate positions [range[0], range[1]); null: fix-up
  const int64_t* range;
  int64_t shard_lo;
  int64_t p0, p1;         // fix-up launch; p1 < 0: all positions
};

template <int P, bool PAIRED, bool FULL, bool OVL = false>
__global__ void __launch_bounds__(GJX_BLOCK) GJX_MH_WAVES
    k_loc_mh(LocK w, MoveKeys keys, int n_moves, int sel_mask, uint64_t offset, int64_t n,
             const float* x_in, const float* y_in, const float* th_in, const float* __restrict__ mx,
             const float* __restrict__ my, const float* __restrict__ mth, float* x, float* y,
             float* th, int64_t* __restrict__ accept_counts, const int32_t* __restrict__ anc,
             int64_t anc_base, float* __restrict__ mx_out, float* __restrict__ my_out,
             float* __restrict__ mth_out, const float* __restrict__ meas, int64_t meas_stride,
             float* __restrict__ logw_reset, MhOverlap ovl) {
  __shared__ uint32_t counts[GJX_MAX_MOVES];
  if (threadIdx.x < GJX_MAX_MOVES) counts[threadIdx.x] = 0u;
  __syncthreads();
  const int64_t nthreads = (int64_t)gridDim.x * GJX_BLOCK;
  const int64_t n_round = ((n + nthreads - 1) / nthreads) * nthreads;
  bool resampled = true;
  // main launch: candidate index of position 0, number of candidates, first global index of the shard, and the
  // own-sourced run as {start, length} -- 32-bit, wave-uniform (n_global < 2^31)
  int32_t ic0 = 0, q0 = 0;
  uint32_t m_cand = 0u, q_len = 0u, lo32 = 0u;
  if constexpr (OVL) {
    if (ovl.idx) {
      const int64_t ja = ovl.range[0];
      m_cand = (uint32_t)(ovl.range[1] - ja);
      ic0 = (int32_t)(ovl.shard_lo - ja);
      lo32 = (uint32_t)ovl.shard_lo;
      resampled = ovl.split[2] != 0;
      const int32_t s0 = ovl.split[0], s1 = ovl.split[1];
      q0 = s0 > 0 ? s0 - 1 : (int32_t)n;
      q_len = s0 > 0 ? (uint32_t)(s1 - q0) : 0u;
    }
  }
  for (int64_t i = (int64_t)blockIdx.x * GJX_BLOCK + threadIdx.x; i < n_round; i += nthreads) {
    bool live = i < n;
    int64_t ii = live ? i : 0;
    // anc != nullptr: the particle is the resampled copy of ancestor anc[i] (the gather of
    // resample_vectorized_trace, smc.py:170-174, fused into the move's loads); its stored means
    // move with it (mx_out / my_out / mth_out)
    int64_t src;
    if constexpr (OVL) {
      int64_t a;
      if (ovl.idx) {
        a = ii;
        if (resampled) {
          const uint32_t ic = (uint32_t)((int32_t)ii + ic0);
          int32_t g = -1;
          if (ic < m_cand) {
            const uint32_t gl = (uint32_t)ovl.idx[ic] - lo32;
            if (gl < (uint32_t)n) g = (int32_t)gl;
          }
          if (live && ((g >= 0) != ((uint32_t)((int32_t)ii - q0) < q_len))) atomicOr(ovl.split + 3, 2);
          a = g;
        }
      } else {
        if (ovl.p1 >= 0) ii = ii < ovl.p0 ? ii : ovl.p1 + (ii - ovl.p0);
        a = (int64_t)anc[ii] - anc_base;
      }
      live = live && a >= 0;
      if (__ballot(live) == 0ull) continue;   // the whole wave is left to the other launch (wave-uniform)
      src = a >= 0 ? a : 0;
    } else {
      src = anc ? (int64_t)anc[ii] - anc_base : ii;
    }
    float cx = x_in[src], cy = y_in[src], ct = th_in[src];
    const float m0 = mx[src], m1 = my[src], m2 = mth[src];
    const uint64_t gi = offset + (uint64_t)ii;
    float ep_old[P], em_old[PAIRED ? P : 1];
    if (meas) {
      // per-ray scores of the current pose as stored by extend (the trace's old_score,
      // core.py:1649-1657): the same bits meas_terms would recompute, one lidar solve saved
#pragma unroll
      for (int r = 0; r < P; ++r)
        if (FULL || r < w.RH) {
          ep_old[r] = meas[(int64_t)r * meas_stride + src];
          if (PAIRED) em_old[r] = meas[(int64_t)(r + w.RH) * meas_stride + src];
        }
    } else {
      (void)meas_terms<P, PAIRED, FULL>(w, cx, cy, ct, ep_old, em_old);
    }
    for (int m = 0; m < n_moves; ++m) {
      float nx, ny, nt, wt, log_u;
      float ep_new[P], em_new[PAIRED ? P : 1];
      mh_propose<P, PAIRED, FULL>(w, keys.k[m], sel_mask, gi, cx, cy, ct, m0, m1, m2, ep_old, em_old, nx, ny,
                                  nt, ep_new, em_new, wt, log_u);
      const float log_alpha = gjx_min0(wt);
      const bool accept = live && (log_u < log_alpha);
      cx = accept ? nx : cx;
      cy = accept ? ny : cy;
      ct = accept ? nt : ct;
#pragma unroll
      for (int r = 0; r < P; ++r)
        if (FULL || r < w.RH) {
          ep_old[r] = accept ? ep_new[r] : ep_old[r];
          if (PAIRED) em_old[r] = accept ? em_new[r] : em_old[r];
        }
      gjx_count_accepts(counts, m, accept);
    }
    if (live) {
      const int64_t o = OVL ? ii : i;
      x[o] = cx;
      y[o] = cy;
      th[o] = ct;
      if (OVL || anc) {
        mx_out[o] = m0;
        my_out[o] = m1;
        mth_out[o] = m2;
      }
      if (logw_reset && resampled) logw_reset[o] = 0.0f;  // resample(): log_weights := 0 (smc.py:646)
    }
  }
  gjx_flush_accepts(counts, n_moves, accept_counts);
}

// One proposal of hmc(trace, selection, step_size, n_steps) (mcmc.py:485-562): momenta, n_steps
// leapfrog steps with the analytic gradient, and the RAW log acceptance ratio
//     log_alpha = [logp(q_L) + log N(-p_L)] - [logp(q_0) + log N(p_0)]
// plus log(u) of the accept test.  Shared by the fused move kernel and gjx_loc_hmc_propose.
// Momenta are drawn in sorted-key order theta, x, y (mcmc.py:497; [3P] dict tree_map order);
// unselected coordinates get zero momentum and a masked gradient: they stay put and their
// (constant) kinetic terms cancel in log_alpha.
template <int P, bool PAIRED, bool FULL>
__device__ __forceinline__ void hmc_propose(const LocK& w, const uint32_t (&k)[4][2], int sel_mask,
                                            const float (&msk)[3], float eps, float half_eps, int n_steps,
                                            uint64_t gi, const float (&q)[3], const float (&g)[3],
                                            float cur_logp, float m0, float m1, float m2,
                                            const float (&inv_s)[3], const float (&log_s)[3], float (&qq)[3],
                                            float (&gg)[3], float& new_logp, float& log_alpha, float& log_u) {
  const float pth = (sel_mask & 4) ? gjx_normal_at(k[0][0], k[0][1], gi) : 0.0f;
  const float ppx = (sel_mask & 1) ? gjx_normal_at(k[1][0], k[1][1], gi) : 0.0f;
  const float ppy = (sel_mask & 2) ? gjx_normal_at(k[2][0], k[2][1], gi) : 0.0f;
  const float u = gjx_uniform_at(k[3][0], k[3][1], gi, 0.0f, 1.0f);
  float p[3] = {ppx, ppy, pth};
  const float prev_mom = ((-0.5f * (p[2] * p[2]) - HALF_LOG_2PI_F) +
                          (-0.5f * (p[0] * p[0]) - HALF_LOG_2PI_F)) +
                         (-0.5f * (p[1] * p[1]) - HALF_LOG_2PI_F);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    qq[c] = q[c];
    gg[c] = g[c];
  }
  new_logp = cur_logp;
  for (int l = 0; l < n_steps; ++l) {
#pragma unroll
    for (int c = 0; c < 3; ++c) p[c] = p[c] + half_eps * (gg[c] * msk[c]);
#pragma unroll
    for (int c = 0; c < 3; ++c) qq[c] = qq[c] + eps * p[c];
    if (l + 1 < n_steps)    // (wave-uniform) the density is only read at the end point
      (void)logp_grad<P, PAIRED, FULL, false>(w, qq[0], qq[1], qq[2], m0, m1, m2, inv_s, log_s, gg);
    else
      new_logp = logp_grad<P, PAIRED, FULL, true>(w, qq[0], qq[1], qq[2], m0, m1, m2, inv_s, log_s, gg);
#pragma unroll
    for (int c = 0; c < 3; ++c) p[c] = p[c] + half_eps * (gg[c] * msk[c]);
  }
  const float new_mom = ((-0.5f * (p[2] * p[2]) - HALF_LOG_2PI_F) +
                         (-0.5f * (p[0] * p[0]) - HALF_LOG_2PI_F)) +
                        (-0.5f * (p[1] * p[1]) - HALF_LOG_2PI_F);
  log_alpha = (new_logp + new_mom) - (cur_logp + prev_mom);
  log_u = __builtin_amdgcn_logf(u) * 0.69314718055994530942f;
}

template <int P, bool PAIRED, bool FULL>
__global__ void __launch_bounds__(GJX_BLOCK) GJX_HMC_WAVES
    k_loc_hmc(LocK w, MoveKeys keys, int n_moves, int sel_mask, float eps, float half_eps,
              int n_steps, uint64_t offset, int64_t n, const float* x_in, const float* y_in,
              const float* th_in, const float* __restrict__ mx, const float* __restrict__ my,
              const float* __restrict__ mth, float* x, float* y, float* th,
              int64_t* __restrict__ accept_counts) {
  __shared__ uint32_t counts[GJX_MAX_MOVES];
  if (threadIdx.x < GJX_MAX_MOVES) counts[threadIdx.x] = 0u;
  __syncthreads();
  float inv_s[3], log_s[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    inv_s[k] = 1.0f / w.sig[k];
    log_s[k] = logf(w.sig[k]);
  }
  const int64_t nthreads = (int64_t)gridDim.x * GJX_BLOCK;
  const int64_t n_round = ((n + nthreads - 1) / nthreads) * nthreads;
  for (int64_t i = (int64_t)blockIdx.x * GJX_BLOCK + threadIdx.x; i < n_round; i += nthreads) {
    const bool live = i < n;
    const int64_t ii = live ? i : 0;
    float q[3] = {x_in[ii], y_in[ii], th_in[ii]};
    const float msk[3] = {(sel_mask & 1) ? 1.0f : 0.0f, (sel_mask & 2) ? 1.0f : 0.0f,
                          (sel_mask & 4) ? 1.0f : 0.0f};
    const float m0 = mx[ii], m1 = my[ii], m2 = mth[ii];
    const uint64_t gi = offset + (uint64_t)ii;
    float g[3];
    float cur_logp = logp_grad<P, PAIRED, FULL>(w, q[0], q[1], q[2], m0, m1, m2, inv_s, log_s, g);
    for (int m = 0; m < n_moves; ++m) {
      float qq[3], gg[3], new_logp, log_alpha, log_u;
      hmc_propose<P, PAIRED, FULL>(w, keys.k[m], sel_mask, msk, eps, half_eps, n_steps, gi, q, g, cur_logp, m0, m1,
                                   m2, inv_s, log_s, qq, gg, new_logp, log_alpha, log_u);
      log_alpha = gjx_min0(log_alpha);
      const bool accept = live && (log_u < log_alpha);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        q[k] = accept ? qq[k] : q[k];
        g[k] = accept ? gg[k] : g[k];
      }
      cur_logp = accept ? new_logp : cur_logp;
      gjx_count_accepts(counts, m, accept);
    }
    if (live) {
      x[i] = q[0];
      y[i] = q[1];
      th[i] = q[2];
    }
  }
  gjx_flush_accepts(counts, n_moves, accept_counts);
}

struct Key4 {
  uint32_t k[4][2];
};

// regenerate(trace, selection) of localization_model for every particle (Fn.regenerate
// core.py:2200-2219 over Distribution.regenerate core.py:1638-1657 and Vmap.regenerate
// core.py:1474-1492), i.e. the first half of mh (mcmc.py:288): the proposed choices, the weight
// and -- for callers that finish the move themselves -- log(u) of mh's accept test and the
// per-ray scores of the proposal.
template <int P, bool PAIRED, bool FULL>
__global__ void __launch_bounds__(GJX_BLOCK)
    k_loc_regenerate(LocK w, Key4 keys, int sel_mask, uint64_t offset, int64_t n,
                     const float* __restrict__ x, const float* __restrict__ y,
                     const float* __restrict__ th, const float* __restrict__ mx,
                     const float* __restrict__ my, const float* __restrict__ mth,
                     const float* __restrict__ meas, int64_t meas_stride, float* __restrict__ nx_out,
                     float* __restrict__ ny_out, float* __restrict__ nt_out, float* __restrict__ weight,
                     float* __restrict__ log_u_out, float* __restrict__ new_meas, int64_t new_stride) {
  GJX_FOR_PARTICLES(i, n) {
    const float cx = x[i], cy = y[i], ct = th[i];
    float ep_old[P], em_old[PAIRED ? P : 1];
    if (meas) {
#pragma unroll
      for (int r = 0; r < P; ++r)
        if (FULL || r < w.RH) {
          ep_old[r] = meas[(int64_t)r * meas_stride + i];
          if (PAIRED) em_old[r] = meas[(int64_t)(r + w.RH) * meas_stride + i];
        }
    } else {
      (void)meas_terms<P, PAIRED, FULL>(w, cx, cy, ct, ep_old, em_old);
    }
    float nx, ny, nt, wt, log_u;
    float ep_new[P], em_new[PAIRED ? P : 1];
    mh_propose<P, PAIRED, FULL>(w, keys.k, sel_mask, offset + (uint64_t)i, cx, cy, ct, mx[i], my[i], mth[i],
                                ep_old, em_old, nx, ny, nt, ep_new, em_new, wt, log_u);
    nx_out[i] = nx;
    ny_out[i] = ny;
    nt_out[i] = nt;
    weight[i] = wt;
    if (log_u_out) log_u_out[i] = log_u;
    if (new_meas) {
#pragma unroll
      for (int r = 0; r < P; ++r)
        if (FULL || r < w.RH) {
          new_meas[(int64_t)r * new_stride + i] = ep_new[r];
          if (PAIRED) new_meas[(int64_t)(r + w.RH) * new_stride + i] = em_new[r];
        }
    }
  }
}

// The proposal half of hmc (mcmc.py:485-562) for every particle: end point of the leapfrog
// trajectory, the raw log acceptance ratio, log(u), and the model log-densities at both ends.
template <int P, bool PAIRED, bool FULL>
__global__ void __launch_bounds__(GJX_BLOCK)
    k_loc_hmc_propose(LocK w, Key4 keys, int sel_mask, float eps, float half_eps, int n_steps,
                      uint64_t offset, int64_t n, const float* __restrict__ x,
                      const float* __restrict__ y, const float* __restrict__ th,
                      const float* __restrict__ mx, const float* __restrict__ my,
                      const float* __restrict__ mth, float* __restrict__ qx, float* __restrict__ qy,
                      float* __restrict__ qth, float* __restrict__ log_alpha_out,
                      float* __restrict__ log_u_out, float* __restrict__ logp_cur,
                      float* __restrict__ logp_new) {
  float inv_s[3], log_s[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    inv_s[k] = 1.0f / w.sig[k];
    log_s[k] = logf(w.sig[k]);
  }
  const float msk[3] = {(sel_mask & 1) ? 1.0f : 0.0f, (sel_mask & 2) ? 1.0f : 0.0f,
                        (sel_mask & 4) ? 1.0f : 0.0f};
  GJX_FOR_PARTICLES(i, n) {
    const float q[3] = {x[i], y[i], th[i]};
    const float m0 = mx[i], m1 = my[i], m2 = mth[i];
    float g[3];
    const float cur = logp_grad<P, PAIRED, FULL>(w, q[0], q[1], q[2], m0, m1, m2, inv_s, log_s, g);
    float qq[3], gg[3], new_logp, log_alpha, log_u;
    hmc_propose<P, PAIRED, FULL>(w, keys.k, sel_mask, msk, eps, half_eps, n_steps, offset + (uint64_t)i, q, g,
                                 cur, m0, m1, m2, inv_s, log_s, qq, gg, new_logp, log_alpha, log_u);
    qx[i] = qq[0];
    qy[i] = qq[1];
    qth[i] = qq[2];
    log_alpha_out[i] = log_alpha;
    if (log_u_out) log_u_out[i] = log_u;
    if (logp_cur) logp_cur[i] = cur;
    if (logp_new) logp_new[i] = new_logp;
  }
}

template <int P, bool PAIRED, bool FULL>
__global__ void __launch_bounds__(GJX_BLOCK)
    k_loc_logp_grad(LocK w, int64_t n, const float* __restrict__ x, const float* __restrict__ y,
                    const float* __restrict__ th, const float* __restrict__ mx,
                    const float* __restrict__ my, const float* __restrict__ mth,
                    float* __restrict__ logp, float* __restrict__ grad) {
  float inv_s[3], log_s[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    inv_s[k] = 1.0f / w.sig[k];
    log_s[k] = logf(w.sig[k]);
  }
  GJX_FOR_PARTICLES(i, n) {
    float g[3];
    logp[i] = logp_grad<P, PAIRED, FULL>(w, x[i], y[i], th[i], mx[i], my[i], mth[i], inv_s, log_s, g);
    grad[i] = g[0];
    grad[n + i] = g[1];
    grad[2 * n + i] = g[2];
  }
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
// Even ray counts use the paired solve with P = R/2 slots; odd counts solve every ray.
#define GJX_DISPATCH_P(K, PPV, PRV, CALL)  \
  do {                                     \
    constexpr int PP = PPV;                \
    constexpr bool PR = PRV;               \
    if ((K).RH == PP) {                    \
      constexpr bool FL = true;            \
      CALL;                                \
    } else {                               \
      constexpr bool FL = false;           \
      CALL;                                \
    }                                      \
  } while (0)

#ifdef GJX_DEV_R8_ONLY  // development builds: instantiate only one variant (8 rays; -DGJX_DEV_PP=16: 32 rays; -DGJX_DEV_PAIRED=false: odd counts)
#ifndef GJX_DEV_PP
#define GJX_DEV_PP 4
#endif
#ifndef GJX_DEV_PAIRED
#define GJX_DEV_PAIRED true
#endif
#ifndef GJX_DEV_FULL
#define GJX_DEV_FULL true
#endif
#define GJX_DISPATCH_R(K, CALL)                                     \
  do {                                                              \
    if ((bool)(K).paired == GJX_DEV_PAIRED && (GJX_DEV_FULL ? (K).RH == GJX_DEV_PP : (K).RH < GJX_DEV_PP)) { \
      constexpr int PP = GJX_DEV_PP;                                \
      constexpr bool PR = GJX_DEV_PAIRED, FL = GJX_DEV_FULL;        \
      CALL;                                                         \
    } else {                                                        \
      gjx_set_error("development build: only one ray count is compiled"); \
      return GJX_ELIMIT;                                            \
    }                                                               \
  } while (0)
#else
#define GJX_DISPATCH_R(K, CALL)                          \
  do {                                                   \
    if ((K).paired) {                                    \
      if ((K).RH <= 4) GJX_DISPATCH_P(K, 4, true, CALL); \
      else if ((K).RH <= 8) GJX_DISPATCH_P(K, 8, true, CALL); \
      else GJX_DISPATCH_P(K, 16, true, CALL);            \
    } else {                                             \
      if ((K).RH <= 8) GJX_DISPATCH_P(K, 8, false, CALL); \
      else if ((K).RH <= 16) GJX_DISPATCH_P(K, 16, false, CALL); \
      else GJX_DISPATCH_P(K, 32, false, CALL);           \
    }                                                    \
  } while (0)
#endif

GJX_API int32_t gjx_loc_lidar(const gjx_loc_world* world, const float* d_px, const float* d_py,
                              const float* d_theta, int64_t n, float* d_out, void* stream) {
  if (n == 0) return GJX_OK;  // an empty particle set is a no-op
  GJX_REQUIRE(d_px && d_py && d_theta && d_out, GJX_EINVAL, "gjx_loc_lidar: null argument");
  GJX_REQUIRE(n > 0, GJX_EINVAL, "gjx_loc_lidar: n must not be negative");
  LocK k;
  int32_t rc = build_lock(world, nullptr, 0, &k, "gjx_loc_lidar");
  if (rc != GJX_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  GJX_DISPATCH_R(k, (k_loc_lidar<PP, PR, FL><<<gjx_grid_for(n), GJX_BLOCK, 0, s>>>(k, d_px, d_py, d_theta, n, d_out)));
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

GJX_API int32_t gjx_loc_lidar_jac(const gjx_loc_world* world, const float* d_px, const float* d_py,
                                  const float* d_theta, int64_t n, float* d_out, float* d_jac, void* stream) {
  if (n == 0) return GJX_OK;
  GJX_REQUIRE(d_px && d_py && d_theta && d_out && d_jac, GJX_EINVAL, "gjx_loc_lidar_jac: null argument");
  GJX_REQUIRE(n > 0, GJX_EINVAL, "gjx_loc_lidar_jac: n must not be negative");
  LocK k;
  int32_t rc = build_lock(world, nullptr, 0, &k, "gjx_loc_lidar_jac");
  if (rc != GJX_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  GJX_DISPATCH_R(k, (k_loc_lidar_jac<PP, PR, FL><<<gjx_grid_for(n), GJX_BLOCK, 0, s>>>(k, d_px, d_py, d_theta, n, d_out,
                                                                                      d_jac)));
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

GJX_API int32_t gjx_loc_ray_scores(const gjx_loc_world* world, const float* obs, const float* d_x,
                                   const float* d_y, const float* d_th, int64_t n, float* d_ray_scores,
                                   void* stream) {
  return gjx_loc_ray_scores_strided(world, obs, d_x, d_y, d_th, n, d_ray_scores, n, stream);
}

int32_t gjx_loc_ray_scores_strided(const gjx_loc_world* world, const float* obs, const float* d_x, const float* d_y,
                                   const float* d_th, int64_t n, float* d_ray_scores, int64_t stride, void* stream) {
  if (n == 0) return GJX_OK;  // an empty particle set is a no-op
  GJX_REQUIRE(obs && d_x && d_y && d_th && d_ray_scores, GJX_EINVAL, "gjx_loc_ray_scores: null argument");
  GJX_REQUIRE(n > 0 && stride >= n, GJX_EINVAL, "gjx_loc_ray_scores: n must not be negative, stride >= n");
  LocK k;
  int32_t rc = build_lock(world, obs, 0, &k, "gjx_loc_ray_scores");
  if (rc != GJX_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  GJX_DISPATCH_R(k, (k_loc_ray_scores<PP, PR, FL><<<gjx_grid_heavy(n), GJX_BLOCK, 0, s>>>(k, d_x, d_y, d_th, n,
                                                                                       d_ray_scores, stride)));
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

int32_t gjx_loc_extend_scores(gjx_ctx* ctx, const gjx_loc_world* world, const uint32_t keys[3][2],
                              int32_t is_initial, uint64_t offset, int64_t n, const float* obs,
                              const float* d_px, const float* d_py, const float* d_pth,
                              const float* d_logw_in, float* d_x, float* d_y, float* d_th, float* d_mx,
                              float* d_my, float* d_mth, float* d_logw, float* d_ray_scores,
                              int64_t scores_stride, void* stream);

GJX_API int32_t gjx_loc_extend(gjx_ctx* ctx, const gjx_loc_world* world, const uint32_t keys[3][2],
                               int32_t is_initial, uint64_t offset, int64_t n, const float* obs,
                               const float* d_px, const float* d_py, const float* d_pth,
                               const float* d_logw_in, float* d_x, float* d_y, float* d_th,
                               float* d_mx, float* d_my, float* d_mth, float* d_logw, void* stream) {
  return gjx_loc_extend_scores(ctx, world, keys, is_initial, offset, n, obs, d_px, d_py, d_pth, d_logw_in,
                               d_x, d_y, d_th, d_mx, d_my, d_mth, d_logw, nullptr, 0, stream);
}

GJX_API int32_t gjx_loc_extend_scores(gjx_ctx* ctx, const gjx_loc_world* world,
                                      const uint32_t keys[3][2], int32_t is_initial, uint64_t offset,
                                      int64_t n, const float* obs, const float* d_px,
                                      const float* d_py, const float* d_pth, const float* d_logw_in,
                                      float* d_x, float* d_y, float* d_th, float* d_mx, float* d_my,
                                      float* d_mth, float* d_logw, float* d_ray_scores,
                                      int64_t scores_stride, void* stream) {
  return gjx_loc_extend_max(ctx, world, keys, is_initial, offset, n, obs, d_px, d_py, d_pth, d_logw_in, d_x, d_y,
                            d_th, d_mx, d_my, d_mth, d_logw, d_ray_scores, scores_stride, nullptr, nullptr, nullptr, stream);
}

// gjx_loc_extend_scores that also writes the block maxima of the new log-weights (d_max_partials,
// *n_partials of them; capacity GJX_MAX_PARTIALS) and / or raises *d_max_cell to their maximum (the cell must hold
// -inf, or a lower bound, on entry): pass 1 of the normaliser for the fused step.
int32_t gjx_loc_extend_max(gjx_ctx* ctx, const gjx_loc_world* world, const uint32_t keys[3][2],
                           int32_t is_initial, uint64_t offset, int64_t n, const float* obs, const float* d_px,
                           const float* d_py, const float* d_pth, const float* d_logw_in, float* d_x,
                           float* d_y, float* d_th, float* d_mx, float* d_my, float* d_mth, float* d_logw,
                           float* d_ray_scores, int64_t scores_stride, float* d_max_partials,
                           int* n_partials, float* d_max_cell, void* stream) {
  (void)ctx;
  if (n == 0) return GJX_OK;  // an empty particle set is a no-op
  GJX_REQUIRE(!d_ray_scores || scores_stride >= n, GJX_EINVAL,
              "gjx_loc_extend_scores: scores_stride=%lld < n=%lld", (long long)scores_stride, (long long)n);
  GJX_REQUIRE(keys && obs && d_x && d_y && d_th && d_mx && d_my && d_mth && d_logw && d_logw_in,
              GJX_EINVAL, "gjx_loc_extend: null argument");
  GJX_REQUIRE(is_initial || (d_px && d_py && d_pth), GJX_EINVAL,
              "gjx_loc_extend: previous pose columns are required unless is_initial");
  GJX_REQUIRE(n > 0, GJX_EINVAL, "gjx_loc_extend: n must not be negative");
  LocK k;
  int32_t rc = build_lock(world, obs, is_initial != 0, &k, "gjx_loc_extend");
  if (rc != GJX_OK) return rc;
  Key3 k3;
  memcpy(k3.k, keys, sizeof(k3.k));
  hipStream_t s = (hipStream_t)stream;
  const int grid = gjx_grid_heavy(n);
  GJX_REQUIRE(!d_max_partials || grid <= GJX_MAX_PARTIALS, GJX_ELIMIT, "gjx_loc_extend: grid %d exceeds %d partials",
              grid, GJX_MAX_PARTIALS);
  if (n_partials) *n_partials = grid;
  GJX_DISPATCH_R(k, (k_loc_extend<PP, PR, FL><<<grid, GJX_BLOCK, 0, s>>>(
                        k, k3, offset, n, d_px, d_py, d_pth, d_logw_in, d_x, d_y, d_th, d_mx, d_my,
                        d_mth, d_logw, d_ray_scores, scores_stride, d_max_partials, d_max_cell)));
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

GJX_API int32_t gjx_loc_extend_locopt(gjx_ctx* ctx, const gjx_loc_world* world,
                                      const uint32_t keys[3][2], int32_t grid_size, float noise_std,
                                      float grid_radius, uint64_t offset, int64_t n, const float* obs,
                                      const float* d_px, const float* d_py, const float* d_pth,
                                      const float* d_logw_in, float* d_x, float* d_y, float* d_th,
                                      float* d_mx, float* d_my, float* d_mth, float* d_logw,
                                      void* stream) {
  (void)ctx;
  if (n == 0) return GJX_OK;  // an empty particle set is a no-op
  GJX_REQUIRE(keys && obs && d_px && d_py && d_pth && d_x && d_y && d_th && d_mx && d_my && d_mth &&
                  d_logw && d_logw_in,
              GJX_EINVAL, "gjx_loc_extend_locopt: null argument");
  GJX_REQUIRE(n > 0, GJX_EINVAL, "gjx_loc_extend_locopt: n must not be negative");
  GJX_REQUIRE(grid_size >= 1 && grid_size <= 64, GJX_ELIMIT,
              "gjx_loc_extend_locopt: grid_size=%d not in [1, 64]", grid_size);
  GJX_REQUIRE(noise_std > 0.0f && grid_radius >= 0.0f, GJX_EINVAL,
              "gjx_loc_extend_locopt: noise_std must be positive, grid_radius non-negative");
  LocK k;
  int32_t rc = build_lock(world, obs, 0, &k, "gjx_loc_extend_locopt");
  if (rc != GJX_OK) return rc;
  Key3 k3;
  memcpy(k3.k, keys, sizeof(k3.k));
  GridK gk;
  gk.g = grid_size;
  gk.radius = grid_radius;
  gk.theta_radius = (float)(3.14159265358979323846 / 6.0);  // jnp.pi / 6 (core.py:771)
  gk.noise = noise_std;
  gk.noise_th = (float)((double)noise_std * 0.5);            // noise_std * 0.5 (core.py:806)
  gk.edge = 0.5f;                                            // core.py:760-763
  hipStream_t s = (hipStream_t)stream;
  // a workgroup per particle until the particles alone fill the chip with threads (GJX_LOCOPT_COOP = 0 / 1 forces
  // the one or the other: tests compare them bit for bit)
  const char* e = getenv("GJX_LOCOPT_COOP");
  const int force = e ? atoi(e) : -1;
  const bool coop = force >= 0 ? force != 0 : n < ((int64_t)1 << 18);
  if (coop) {
    const int blocks = (int)(n < 65536 ? n : 65536);
    GJX_DISPATCH_R(k, (k_loc_extend_locopt_coop<PP, PR, FL><<<blocks, GJX_BLOCK, 0, s>>>(
                          k, k3, gk, offset, n, d_px, d_py, d_pth, d_logw_in, d_x, d_y, d_th, d_mx, d_my, d_mth,
                          d_logw)));
  } else {
    GJX_DISPATCH_R(k, (k_loc_extend_locopt<PP, PR, FL><<<gjx_grid_heavy(n), GJX_BLOCK, 0, s>>>(
                          k, k3, gk, offset, n, d_px, d_py, d_pth, d_logw_in, d_x, d_y, d_th, d_mx,
                          d_my, d_mth, d_logw)));
  }
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

static int32_t zero_accepts(int64_t* d_accept_counts, int n_moves, hipStream_t s) {
  if (d_accept_counts)
    GJX_CHECK_HIP(hipMemsetAsync(d_accept_counts, 0, sizeof(int64_t) * n_moves, s));
  return GJX_OK;
}

static int32_t launch_mh(const char* who, gjx_ctx* ctx, const gjx_loc_world* world,
                         const uint32_t (*keys)[4][2], int32_t n_moves, int32_t sel_mask,
                         int32_t is_initial, uint64_t offset, int64_t n, const float* obs,
                         const int32_t* d_anc, int64_t anc_base, const float* const* src6,
                         float* const* dst6, const float* d_ray_scores, int64_t scores_stride,
                         float* d_logw_reset, int64_t* d_accept_counts, void* stream,
                         const MhOverlap* ovl = nullptr, bool keep_counts = false) {
  GJX_REQUIRE(ctx && keys && obs, GJX_EINVAL, "%s: null argument", who);
  for (int c = 0; c < 6; ++c)
    GJX_REQUIRE(src6[c] && (dst6[c] || (c >= 3 && !d_anc && !ovl)), GJX_EINVAL, "%s: column %d is null", who, c);
  GJX_REQUIRE(sel_mask >= 1 && sel_mask <= 7, GJX_EINVAL, "%s: sel_mask=%d not in [1, 7]", who, sel_mask);
  GJX_REQUIRE(n > 0, GJX_EINVAL, "%s: n must not be negative", who);
  GJX_REQUIRE(n_moves >= 1 && n_moves <= GJX_MAX_MOVES, GJX_ELIMIT, "%s: n_moves=%d not in [1, %d]", who,
              n_moves, GJX_MAX_MOVES);
  LocK k;
  int32_t rc = build_lock(world, obs, is_initial != 0, &k, who);
  if (rc != GJX_OK) return rc;
  MoveKeys mk;
  memset(&mk, 0, sizeof(mk));
  memcpy(mk.k, keys, sizeof(uint32_t) * 8 * n_moves);
  hipStream_t s = (hipStream_t)stream;
  if (!keep_counts) {
    rc = zero_accepts(d_accept_counts, n_moves, s);
    if (rc != GJX_OK) return rc;
  }
  const int grid = gjx_grid_heavy(n);
  if (ovl) {
    GJX_DISPATCH_R(k, (k_loc_mh<PP, PR, FL, true><<<grid, GJX_BLOCK, 0, s>>>(
                          k, mk, n_moves, sel_mask, offset, n, src6[0], src6[1], src6[2], src6[3], src6[4],
                          src6[5], dst6[0], dst6[1], dst6[2], d_accept_counts, d_anc, anc_base, dst6[3],
                          dst6[4], dst6[5], d_ray_scores, scores_stride, d_logw_reset, *ovl)));
  } else {
    GJX_DISPATCH_R(k, (k_loc_mh<PP, PR, FL><<<grid, GJX_BLOCK, 0, s>>>(
                          k, mk, n_moves, sel_mask, offset, n, src6[0], src6[1], src6[2], src6[3], src6[4],
                          src6[5], dst6[0], dst6[1], dst6[2], d_accept_counts, d_anc, anc_base, dst6[3],
                          dst6[4], dst6[5], d_ray_scores, scores_stride, d_logw_reset, MhOverlap{})));
  }
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

// The move launches of the overlapped sharded step and of the one-GPU step decided on the device (gjx_smc.hip).  Main
// (d_split = {p0 + 1, p1, go, flags}; one GPU: {1, n, go, flags}): decision and run read on the device, ancestors from the search (d_idx, d_range), n_work = the shard's n positions.  Fix-up:
// ancestors from d_anc for the positions [0, p0) and [p1, ...), counters kept.
int32_t gjx_loc_mh_overlap(gjx_ctx* ctx, const gjx_loc_world* world, const uint32_t (*keys)[4][2],
                           int32_t n_moves, int32_t sel_mask, int32_t is_initial, uint64_t offset, int64_t n_work,
                           const float* obs, const int32_t* d_anc, const float* const* d_src6, float* const* d_dst6,
                           const float* d_ray_scores, int64_t scores_stride, float* d_logw_reset,
                           int64_t* d_accept_counts, int32_t* d_split, const int32_t* d_idx, const int64_t* d_range,
                           int64_t p0, int64_t p1, void* stream) {
  if (n_work == 0) return GJX_OK;
  GJX_REQUIRE(d_src6 && d_dst6, GJX_EINVAL, "gjx_loc_mh_overlap: null argument");
  GJX_REQUIRE(d_split ? (d_idx && d_range) : (d_anc != nullptr && !d_idx), GJX_EINVAL,
              "gjx_loc_mh_overlap: the main launch needs the search results, the fix-up launch its ancestors");
  GJX_REQUIRE(offset + (uint64_t)n_work < ((uint64_t)1 << 31), GJX_ELIMIT, "gjx_loc_mh_overlap: indices beyond 2^31");
  for (int c = 0; c < 6; ++c)
    GJX_REQUIRE(d_src6[c] != d_dst6[c], GJX_EINVAL, "gjx_loc_mh_overlap: column %d gathers in place", c);
  const MhOverlap ovl{d_split, d_idx, d_range, (int64_t)offset, p0, p1};
  return launch_mh("gjx_loc_mh_overlap", ctx, world, keys, n_moves, sel_mask, is_initial, offset, n_work, obs, d_anc,
                   0, d_src6, d_dst6, d_ray_scores, scores_stride, d_logw_reset, d_accept_counts, stream, &ovl,
                   /*keep_counts=*/true);
}

GJX_API int32_t gjx_loc_mh(gjx_ctx* ctx, const gjx_loc_world* world, const uint32_t (*keys)[4][2],
                           int32_t n_moves, int32_t sel_mask, int32_t is_initial, uint64_t offset,
                           int64_t n, const float* obs, const float* d_x_in, const float* d_y_in,
                           const float* d_th_in, const float* d_mx, const float* d_my,
                           const float* d_mth, float* d_x, float* d_y, float* d_th,
                           int64_t* d_accept_counts, void* stream) {
  if (n == 0) return GJX_OK;  // an empty particle set is a no-op
  const float* src6[6] = {d_x_in, d_y_in, d_th_in, d_mx, d_my, d_mth};
  float* dst6[6] = {d_x, d_y, d_th, nullptr, nullptr, nullptr};
  return launch_mh("gjx_loc_mh", ctx, world, keys, n_moves, sel_mask, is_initial, offset, n, obs, nullptr, 0,
                   src6, dst6, nullptr, 0, nullptr, d_accept_counts, stream);
}

GJX_API int32_t gjx_loc_mh_resampled(gjx_ctx* ctx, const gjx_loc_world* world,
                                     const uint32_t (*keys)[4][2], int32_t n_moves, int32_t sel_mask,
                                     int32_t is_initial, uint64_t offset, int64_t n, const float* obs,
                                     const int32_t* d_ancestors, int64_t anc_base,
                                     const float* const* d_src6, float* const* d_dst6,
                                     const float* d_ray_scores, int64_t scores_stride,
                                     float* d_logw_reset, int64_t* d_accept_counts, void* stream) {
  if (n == 0) return GJX_OK;  // an empty particle set is a no-op
  GJX_REQUIRE(d_src6 && d_dst6, GJX_EINVAL, "gjx_loc_mh_resampled: null argument");
  GJX_REQUIRE(d_ancestors || d_ray_scores, GJX_EINVAL,
              "gjx_loc_mh_resampled: neither ancestors nor stored scores given (use gjx_loc_mh)");
  GJX_REQUIRE(!d_ray_scores || scores_stride >= (d_ancestors ? 1 : n), GJX_EINVAL,
              "gjx_loc_mh_resampled: scores_stride=%lld too small", (long long)scores_stride);
  if (d_ancestors)
    for (int c = 0; c < 6; ++c)
      GJX_REQUIRE(d_src6[c] != d_dst6[c], GJX_EINVAL, "gjx_loc_mh_resampled: column %d gathers in place", c);
  return launch_mh("gjx_loc_mh_resampled", ctx, world, keys, n_moves, sel_mask, is_initial, offset, n, obs,
                   d_ancestors, anc_base, d_src6, d_dst6, d_ray_scores, scores_stride, d_logw_reset,
                   d_accept_counts, stream);
}

GJX_API int32_t gjx_loc_hmc(gjx_ctx* ctx, const gjx_loc_world* world, const uint32_t (*keys)[4][2],
                            int32_t n_moves, int32_t sel_mask, float step_size, int32_t n_steps,
                            int32_t is_initial, uint64_t offset, int64_t n, const float* obs,
                            const float* d_x_in, const float* d_y_in, const float* d_th_in,
                            const float* d_mx, const float* d_my, const float* d_mth, float* d_x,
                            float* d_y, float* d_th, int64_t* d_accept_counts, void* stream) {
  if (n == 0) return GJX_OK;  // an empty particle set is a no-op
  GJX_REQUIRE(ctx && keys && obs && d_x && d_y && d_th && d_x_in && d_y_in && d_th_in && d_mx &&
                  d_my && d_mth,
              GJX_EINVAL, "gjx_loc_hmc: null argument");
  GJX_REQUIRE(sel_mask >= 1 && sel_mask <= 7, GJX_EINVAL, "gjx_loc_hmc: sel_mask=%d not in [1, 7]",
              sel_mask);
  GJX_REQUIRE(n > 0, GJX_EINVAL, "gjx_loc_hmc: n must not be negative");
  GJX_REQUIRE(n_moves >= 1 && n_moves <= GJX_MAX_MOVES, GJX_ELIMIT,
              "gjx_loc_hmc: n_moves=%d not in [1, %d]", n_moves, GJX_MAX_MOVES);
  GJX_REQUIRE(n_steps >= 1 && n_steps <= 4096, GJX_ELIMIT, "gjx_loc_hmc: n_steps=%d out of range",
              n_steps);
  LocK k;
  int32_t rc = build_lock(world, obs, is_initial != 0, &k, "gjx_loc_hmc");
  if (rc != GJX_OK) return rc;
  MoveKeys mk;
  memset(&mk, 0, sizeof(mk));
  memcpy(mk.k, keys, sizeof(uint32_t) * 8 * n_moves);
  hipStream_t s = (hipStream_t)stream;
  rc = zero_accepts(d_accept_counts, n_moves, s);
  if (rc != GJX_OK) return rc;
  const int grid = gjx_grid_heavy(n);
  const float half_eps = (float)((double)step_size / 2.0);  // (step_size / 2.0) * g, mcmc.py:516
  GJX_DISPATCH_R(k, (k_loc_hmc<PP, PR, FL><<<grid, GJX_BLOCK, 0, s>>>(
                        k, mk, n_moves, sel_mask, step_size, half_eps, n_steps, offset, n, d_x_in,
                        d_y_in, d_th_in, d_mx, d_my, d_mth, d_x, d_y, d_th, d_accept_counts)));
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

GJX_API int32_t gjx_loc_regenerate(const gjx_loc_world* world, const uint32_t keys[4][2], int32_t sel_mask,
                                   int32_t is_initial, uint64_t offset, int64_t n, const float* obs,
                                   const float* d_x, const float* d_y, const float* d_th,
                                   const float* d_mx, const float* d_my, const float* d_mth,
                                   const float* d_ray_scores, int64_t scores_stride, float* d_nx,
                                   float* d_ny, float* d_nth, float* d_weight, float* d_log_u,
                                   float* d_new_scores, int64_t new_scores_stride, void* stream) {
  if (n == 0) return GJX_OK;  // an empty particle set is a no-op
  GJX_REQUIRE(keys && obs && d_x && d_y && d_th && d_mx && d_my && d_mth && d_nx && d_ny && d_nth && d_weight,
              GJX_EINVAL, "gjx_loc_regenerate: null argument");
  GJX_REQUIRE(sel_mask >= 1 && sel_mask <= 7, GJX_EINVAL, "gjx_loc_regenerate: sel_mask=%d not in [1, 7]",
              sel_mask);
  GJX_REQUIRE(n > 0, GJX_EINVAL, "gjx_loc_regenerate: n must not be negative");
  GJX_REQUIRE((!d_ray_scores || scores_stride >= n) && (!d_new_scores || new_scores_stride >= n), GJX_EINVAL,
              "gjx_loc_regenerate: a scores stride is smaller than n=%lld", (long long)n);
  LocK k;
  int32_t rc = build_lock(world, obs, is_initial != 0, &k, "gjx_loc_regenerate");
  if (rc != GJX_OK) return rc;
  Key4 k4;
  memcpy(k4.k, keys, sizeof(k4.k));
  hipStream_t s = (hipStream_t)stream;
  GJX_DISPATCH_R(k, (k_loc_regenerate<PP, PR, FL><<<gjx_grid_heavy(n), GJX_BLOCK, 0, s>>>(
                        k, k4, sel_mask, offset, n, d_x, d_y, d_th, d_mx, d_my, d_mth, d_ray_scores,
                        scores_stride, d_nx, d_ny, d_nth, d_weight, d_log_u, d_new_scores, new_scores_stride)));
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

GJX_API int32_t gjx_loc_hmc_propose(const gjx_loc_world* world, const uint32_t keys[4][2], int32_t sel_mask,
                                    float step_size, int32_t n_steps, int32_t is_initial, uint64_t offset,
                                    int64_t n, const float* obs, const float* d_x, const float* d_y,
                                    const float* d_th, const float* d_mx, const float* d_my,
                                    const float* d_mth, float* d_qx, float* d_qy, float* d_qth,
                                    float* d_log_alpha, float* d_log_u, float* d_logp_cur, float* d_logp_new,
                                    void* stream) {
  if (n == 0) return GJX_OK;  // an empty particle set is a no-op
  GJX_REQUIRE(keys && obs && d_x && d_y && d_th && d_mx && d_my && d_mth && d_qx && d_qy && d_qth && d_log_alpha,
              GJX_EINVAL, "gjx_loc_hmc_propose: null argument");
  GJX_REQUIRE(sel_mask >= 1 && sel_mask <= 7, GJX_EINVAL, "gjx_loc_hmc_propose: sel_mask=%d not in [1, 7]",
              sel_mask);
  GJX_REQUIRE(n > 0, GJX_EINVAL, "gjx_loc_hmc_propose: n must not be negative");
  GJX_REQUIRE(n_steps >= 1 && n_steps <= 4096, GJX_ELIMIT, "gjx_loc_hmc_propose: n_steps=%d out of range",
              n_steps);
  LocK k;
  int32_t rc = build_lock(world, obs, is_initial != 0, &k, "gjx_loc_hmc_propose");
  if (rc != GJX_OK) return rc;
  Key4 k4;
  memcpy(k4.k, keys, sizeof(k4.k));
  hipStream_t s = (hipStream_t)stream;
  const float half_eps = (float)((double)step_size / 2.0);  // (step_size / 2.0) * g, mcmc.py:516
  GJX_DISPATCH_R(k, (k_loc_hmc_propose<PP, PR, FL><<<gjx_grid_heavy(n), GJX_BLOCK, 0, s>>>(
                        k, k4, sel_mask, step_size, half_eps, n_steps, offset, n, d_x, d_y, d_th, d_mx, d_my,
                        d_mth, d_qx, d_qy, d_qth, d_log_alpha, d_log_u, d_logp_cur, d_logp_new)));
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

GJX_API int32_t gjx_loc_logp_grad(const gjx_loc_world* world, int32_t is_initial, int64_t n,
                                  const float* obs, const float* d_x, const float* d_y,
                                  const float* d_th, const float* d_mx, const float* d_my,
                                  const float* d_mth, float* d_logp, float* d_grad3,
                                  void* stream) {
  if (n == 0) return GJX_OK;  // an empty particle set is a no-op
  GJX_REQUIRE(obs && d_x && d_y && d_th && d_mx && d_my && d_mth && d_logp && d_grad3, GJX_EINVAL,
              "gjx_loc_logp_grad: null argument");
  GJX_REQUIRE(n > 0, GJX_EINVAL, "gjx_loc_logp_grad: n must not be negative");
  LocK k;
  int32_t rc = build_lock(world, obs, is_initial != 0, &k, "gjx_loc_logp_grad");
  if (rc != GJX_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  GJX_DISPATCH_R(k, (k_loc_logp_grad<PP, PR, FL><<<gjx_grid_for(n), GJX_BLOCK, 0, s>>>(
                        k, n, d_x, d_y, d_th, d_mx, d_my, d_mth, d_logp, d_grad3)));
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

// ---------------------------------------------------------------------------
// host-side restatement of distance_to_wall_lidar (core.py:252-308, 81-128) in unfused f32
// with libm sin/cos and IEEE division.  Used only by the synthetic-data generator
// (data.py:124-165), where a ray through a wall end point (the y = x waypoints graze the
// corner (7, 7)) makes the observation depend on the last ulp: the data set is built once on
// the host, as in the reference, and must not depend on device rounding.
// ---------------------------------------------------------------------------
#pragma clang fp contract(off)
GJX_API int32_t gjx_loc_lidar_host(const gjx_loc_world* w, const float* px, const float* py,
                                   const float* theta, int64_t n, float* out) {
  GJX_REQUIRE(w && px && py && theta && out, GJX_EINVAL, "gjx_loc_lidar_host: null argument");
  GJX_REQUIRE(w->n_walls >= 1 && w->n_walls <= GJX_MAX_WALLS && w->n_rays >= 1 &&
                  w->n_rays <= GJX_MAX_RAYS,
              GJX_ELIMIT, "gjx_loc_lidar_host: world out of limits");
  const float two_pi = 6.283185307179586f;
  for (int64_t i = 0; i < n; ++i) {
    for (int r = 0; r < w->n_rays; ++r) {
      const float step = (float)r / (float)w->n_rays;
      const float ang = 0.0f * (1.0f - step) + two_pi * step;
      const float wa = theta[i] + ang;
      const float dx = cosf(wa), dy = sinf(wa);
      const float t_right = dx > 0.0f ? (w->width - px[i]) / dx : INFINITY;
      const float t_left = dx < 0.0f ? -px[i] / dx : INFINITY;
      const float t_top = dy > 0.0f ? (w->height - py[i]) / dy : INFINITY;
      const float t_bottom = dy < 0.0f ? -py[i] / dy : INFINITY;
      float best = fminf(fminf(t_right, t_left), fminf(t_top, t_bottom));
      for (int k = 0; k < w->n_walls; ++k) {
        const float wdx = w->walls[k][2] - w->walls[k][0], wdy = w->walls[k][3] - w->walls[k][1];
        const float twx = w->walls[k][0] - px[i], twy = w->walls[k][1] - py[i];
        float denom = dx * wdy - dy * wdx;
        if (fabsf(denom) < 1e-10f) denom = 1e-10f;
        const float t = (twx * wdy - twy * wdx) / denom;
        const float s = (twx * dy - twy * dx) / denom;
        if (t >= 0.0f && s >= 0.0f && s <= 1.0f) best = fminf(best, t);
      }
      out[i * w->n_rays + r] = fminf(best, w->max_range);
    }
  }
  return GJX_OK;
}

// jax.random.normal(key) for a scalar draw, on the host (erf_inv as XLA's f32 polynomial).
GJX_API float gjx_host_normal(const uint32_t key[2]) {
  uint32_t a, b;
  gjx_tf2x32(key[0], key[1], 0u, 0u, a, b);
  const uint32_t bits = a ^ b;
  uint32_t fb = (bits >> 9) | 0x3F800000u;
  float f;
  memcpy(&f, &fb, 4);
  f -= 1.0f;
  const float lo = -0.99999994f;
  float u = f * 2.0f + lo;
  u = u > lo ? u : lo;
  float w = -log1pf(-u * u);
  float p;
  if (w < 5.0f) {
    w = w - 2.5f;
    p = 2.81022636e-08f;
    p = 3.43273939e-07f + p * w;
    p = -3.5233877e-06f + p * w;
    p = -4.39150654e-06f + p * w;
    p = 0.00021858087f + p * w;
    p = -0.00125372503f + p * w;
    p = -0.00417768164f + p * w;
    p = 0.246640727f + p * w;
    p = 1.50140941f + p * w;
  } else {
    w = sqrtf(w) - 3.0f;
    p = -0.000200214257f;
    p = 0.000100950558f + p * w;
    p = 0.00134934322f + p * w;
    p = -0.00367342844f + p * w;
    p = 0.00573950773f + p * w;
    p = -0.0076224613f + p * w;
    p = 0.00943887047f + p * w;
    p = 1.00167406f + p * w;
    p = 2.83297682f + p * w;
  }
  return 1.41421356237309504880f * (p * u);
}
