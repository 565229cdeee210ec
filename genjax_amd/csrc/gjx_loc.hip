// gjx_loc.hip -- fused kernels for the localization model family
// (examples/localization/core.py:81-128, 252-395) behind init/extend (smc.py:272-371, 449-530),
// mh (mcmc.py:267-309) and hmc (mcmc.py:434-578).
//
// One thread = one particle; the world (walls, ray angles, observation) is wave-uniform and
// travels in the kernel arguments, so the compiler keeps it in SGPRs (s_load).  The lidar
// loop runs walls outer / rays inner with the per-ray direction and running minimum in
// VGPRs.  PRNG draws are counter-based on the global particle index (no HBM traffic).
// Algorithmic HBM bytes per particle: extend 44 B, mh/hmc 24 B read + 12 B write for ALL
// fused moves; the kernels are VALU-bound (finding 5).
#include "gjx_common.h"

struct LocK {
  float width, height, xlo, xhi, ylo, yhi;
  int W, R;
  float x1[GJX_MAX_WALLS], y1[GJX_MAX_WALLS], wdx[GJX_MAX_WALLS], wdy[GJX_MAX_WALLS];
  float ang[GJX_MAX_RAYS];    // jnp.linspace(0, 2pi, R, endpoint=False)  (core.py:267)
  float obs_s[GJX_MAX_RAYS];  // obs_r / sigma
  float inv_sigma, lp_const, max_range;
  float mean0[3];             // initial means (core.py:331-333)
  float sig[3];               // prior sigmas of THIS step (core.py:351-353)
  int is_initial;
};

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
  for (int i = 0; i < w->n_walls; ++i) {
    k->x1[i] = w->walls[i][0];
    k->y1[i] = w->walls[i][1];
    k->wdx[i] = w->walls[i][2] - w->walls[i][0];
    k->wdy[i] = w->walls[i][3] - w->walls[i][1];
  }
  const float two_pi = 6.283185307179586f;
  for (int r = 0; r < w->n_rays; ++r) {
    // start*(1 - i/num) + stop*(i/num) with start = 0  ([3P] jnp.linspace, f32)
    float step = (float)r / (float)w->n_rays;
    k->ang[r] = 0.0f * (1.0f - step) + two_pi * step;
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

// distance_to_wall_lidar + compute_ray_wall_intersection for one pose (already clipped).
// best[r] = min(t_boundary, t_wall, max_range).  When GRAD, kbest[r] records the winner:
// 0..W-1 wall, -1 right, -2 left, -3 top, -4 bottom, -5 range cap.
template <int RMAX, bool GRAD>
__device__ __forceinline__ void lidar_dev(const LocK& w, float px, float py, float th,
                                          float (&dx)[RMAX], float (&dy)[RMAX],
                                          float (&best)[RMAX], int (&kbest)[RMAX]) {
  const float inf = __builtin_inff();
#pragma unroll
  for (int r = 0; r < RMAX; ++r) {
    if (r < w.R) {
      float s, c;
      sincosf(th + w.ang[r], &s, &c);
      dx[r] = c;
      dy[r] = s;
      const float rc = __frcp_rn(c), rs = __frcp_rn(s);
      const float t_right = c > 0.0f ? (w.width - px) * rc : inf;
      const float t_left = c < 0.0f ? -px * rc : inf;
      const float t_top = s > 0.0f ? (w.height - py) * rs : inf;
      const float t_bottom = s < 0.0f ? -py * rs : inf;
      // exactly one of right/left and one of top/bottom is finite
      const float tx = fminf(t_right, t_left), ty = fminf(t_top, t_bottom);
      float b = fminf(tx, ty);
      if (GRAD) {
        int kx = (c > 0.0f) ? -1 : -2, ky = (s > 0.0f) ? -3 : -4;
        kbest[r] = (ty < tx) ? ky : kx;
        if (!(b <= w.max_range)) kbest[r] = -5;
      }
      best[r] = fminf(b, w.max_range);
    }
  }
  for (int k = 0; k < w.W; ++k) {
    const float wdx = w.wdx[k], wdy = w.wdy[k];
    const float twx = w.x1[k] - px, twy = w.y1[k] - py;
    const float A = twx * wdy - twy * wdx;
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
      if (r < w.R) {
        float denom = dx[r] * wdy - dy[r] * wdx;
        denom = (fabsf(denom) < 1e-10f) ? 1e-10f : denom;
        const float rinv = __frcp_rn(denom);
        const float t = A * rinv;
        const float s = (twx * dy[r] - twy * dx[r]) * rinv;
        const bool valid = (t >= 0.0f) & (s >= 0.0f) & (s <= 1.0f) & (t < best[r]);
        best[r] = valid ? t : best[r];
        if (GRAD) kbest[r] = valid ? k : kbest[r];
      }
    }
  }
}

// sum_r log N(obs_r; d_r, sigma) for the (unclipped) choices x, y, th; lp[r] per ray.
template <int RMAX>
__device__ __forceinline__ float meas_logp(const LocK& w, float x, float y, float th,
                                           float (&lp)[RMAX]) {
  float dx[RMAX], dy[RMAX], best[RMAX];
  int kb[RMAX];
  const float px = clampf(x, w.xlo, w.xhi), py = clampf(y, w.ylo, w.yhi);
  lidar_dev<RMAX, false>(w, px, py, th, dx, dy, best, kb);
  float sum = 0.0f;
#pragma unroll
  for (int r = 0; r < RMAX; ++r) {
    if (r < w.R) {
      const float z = w.obs_s[r] - best[r] * w.inv_sigma;
      lp[r] = -0.5f * (z * z) - w.lp_const;
      sum += lp[r];
    }
  }
  return sum;
}

__device__ __forceinline__ float normal_lp(float x, float mu, float inv_s, float log_s) {
  const float z = (x - mu) * inv_s;
  return -0.5f * (z * z) - (HALF_LOG_2PI_F + log_s);
}

// assess() with all sites constrained + analytic gradient w.r.t. (x, y, theta).
//   d/dpx t_wall = -wdy/D, d/dpy t_wall = wdx/D, d/dphi t_wall = t (s wdy + c wdx)/D
//   right/left:  d/dpx = -1/c, d/dphi = t s/c;   top/bottom: d/dpy = -1/s, d/dphi = -t c/s
//   d logN(obs; d, sigma)/dd = z/sigma with z = (obs - d)/sigma; clip() passes gradient
//   only inside the box; min() passes it to the winner; the range cap passes none.
template <int RMAX>
__device__ __forceinline__ float logp_grad(const LocK& w, float x, float y, float th, float mx,
                                           float my, float mth, const float (&inv_s)[3],
                                           const float (&log_s)[3], float (&g)[3]) {
  float dx[RMAX], dy[RMAX], best[RMAX];
  int kb[RMAX];
  const float px = clampf(x, w.xlo, w.xhi), py = clampf(y, w.ylo, w.yhi);
  lidar_dev<RMAX, true>(w, px, py, th, dx, dy, best, kb);
  float meas = 0.0f, gpx = 0.0f, gpy = 0.0f, gth = 0.0f;
#pragma unroll
  for (int r = 0; r < RMAX; ++r) {
    if (r < w.R) {
      const float t = best[r];
      const float z = w.obs_s[r] - t * w.inv_sigma;
      meas += -0.5f * (z * z) - w.lp_const;
      const float gd = z * w.inv_sigma;
      const int k = kb[r];
      const float c = dx[r], s = dy[r];
      float tpx = 0.0f, tpy = 0.0f, tph = 0.0f;
      if (k >= 0) {
        const float wdx = w.wdx[k], wdy = w.wdy[k];
        float D = c * wdy - s * wdx;
        const bool tiny = fabsf(D) < 1e-10f;
        D = tiny ? 1e-10f : D;
        const float rD = __frcp_rn(D);
        tpx = -wdy * rD;
        tpy = wdx * rD;
        tph = tiny ? 0.0f : t * (s * wdy + c * wdx) * rD;
      } else if (k >= -2) {
        const float rc = __frcp_rn(c);
        tpx = -rc;
        tph = t * s * rc;
      } else if (k >= -4) {
        const float rs = __frcp_rn(s);
        tpy = -rs;
        tph = -t * c * rs;
      }
      gpx += gd * tpx;
      gpy += gd * tpy;
      gth += gd * tph;
    }
  }
  const float zx = (x - mx) * inv_s[0], zy = (y - my) * inv_s[1], zt = (th - mth) * inv_s[2];
  float logp = 0.0f;
  logp += -0.5f * (zx * zx) - (HALF_LOG_2PI_F + log_s[0]);
  logp += -0.5f * (zy * zy) - (HALF_LOG_2PI_F + log_s[1]);
  logp += -0.5f * (zt * zt) - (HALF_LOG_2PI_F + log_s[2]);
  logp += meas;
  const bool in_x = (x >= w.xlo) & (x <= w.xhi), in_y = (y >= w.ylo) & (y <= w.yhi);
  g[0] = -zx * inv_s[0] + (in_x ? gpx : 0.0f);
  g[1] = -zy * inv_s[1] + (in_y ? gpy : 0.0f);
  g[2] = -zt * inv_s[2] + gth;
  return logp;
}

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------
template <int RMAX>
__global__ void __launch_bounds__(GJX_BLOCK)
    k_loc_lidar(LocK w, const float* __restrict__ px, const float* __restrict__ py,
                const float* __restrict__ th, int64_t n, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * GJX_BLOCK + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * GJX_BLOCK) {
    float dx[RMAX], dy[RMAX], best[RMAX];
    int kb[RMAX];
    lidar_dev<RMAX, false>(w, px[i], py[i], th[i], dx, dy, best, kb);
#pragma unroll
    for (int r = 0; r < RMAX; ++r)
      if (r < w.R) out[i * w.R + r] = best[r];
  }
}

struct Key3 {
  uint32_t k[3][2];
};

template <int RMAX>
__global__ void __launch_bounds__(GJX_BLOCK)
    k_loc_extend(LocK w, Key3 keys, uint64_t offset, int64_t n, const float* px_in,
                 const float* py_in, const float* pth_in, const float* logw_in, float* x, float* y,
                 float* th, float* mx, float* my, float* mth, float* logw) {
  for (int64_t i = (int64_t)blockIdx.x * GJX_BLOCK + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * GJX_BLOCK) {
    float m0, m1, m2;
    if (w.is_initial) {
      m0 = w.mean0[0];
      m1 = w.mean0[1];
      m2 = w.mean0[2];
    } else {
      m0 = clampf(px_in[i], w.xlo, w.xhi);
      m1 = clampf(py_in[i], w.ylo, w.yhi);
      m2 = pth_in[i];
    }
    const float lw_old = logw_in[i];
    const uint64_t gi = offset + (uint64_t)i;
    // tfd.Normal.sample: normal(key, shape) * scale + loc
    const float nx = __fadd_rn(__fmul_rn(gjx_normal_at(keys.k[0][0], keys.k[0][1], gi), w.sig[0]), m0);
    const float ny = __fadd_rn(__fmul_rn(gjx_normal_at(keys.k[1][0], keys.k[1][1], gi), w.sig[1]), m1);
    const float nt = __fadd_rn(__fmul_rn(gjx_normal_at(keys.k[2][0], keys.k[2][1], gi), w.sig[2]), m2);
    float lp[RMAX];
    const float wsum = meas_logp<RMAX>(w, nx, ny, nt, lp);
    x[i] = nx;
    y[i] = ny;
    th[i] = nt;
    mx[i] = m0;
    my[i] = m1;
    mth[i] = m2;
    logw[i] = lw_old + wsum;
  }
}

// Block-level acceptance counts: wave ballot -> LDS -> one plain store per (block, move).
__device__ __forceinline__ void count_accepts(uint32_t* lds_counts, int move, bool accept) {
  const unsigned long long b = __ballot(accept);
  if ((threadIdx.x & 63) == 0) atomicAdd(&lds_counts[move], (uint32_t)__popcll(b));
}

template <int RMAX>
__global__ void __launch_bounds__(GJX_BLOCK)
    k_loc_mh(LocK w, MoveKeys keys, int n_moves, int sel_mask, uint64_t offset, int64_t n,
             const float* x_in, const float* y_in, const float* th_in, const float* __restrict__ mx,
             const float* __restrict__ my, const float* __restrict__ mth, float* x, float* y,
             float* th, uint32_t* __restrict__ accept_partials) {
  __shared__ uint32_t counts[GJX_MAX_MOVES];
  if (threadIdx.x < GJX_MAX_MOVES) counts[threadIdx.x] = 0u;
  __syncthreads();
  const int64_t nthreads = (int64_t)gridDim.x * GJX_BLOCK;
  const int64_t n_round = ((n + nthreads - 1) / nthreads) * nthreads;
  for (int64_t i = (int64_t)blockIdx.x * GJX_BLOCK + threadIdx.x; i < n_round; i += nthreads) {
    const bool live = i < n;
    const int64_t ii = live ? i : 0;
    float cx = x_in[ii], cy = y_in[ii], ct = th_in[ii];
    const float m0 = mx[ii], m1 = my[ii], m2 = mth[ii];
    const uint64_t gi = offset + (uint64_t)ii;
    float lp_old[RMAX];
    (void)meas_logp<RMAX>(w, cx, cy, ct, lp_old);
    for (int m = 0; m < n_moves; ++m) {
      // regenerate: fresh prior draws at the stored args (core.py:1645-1648)
      // unselected sites keep their value and contribute logpdf + old_score = 0
      float nx = cx, ny = cy, nt = ct;
      if (sel_mask & 1) nx = __fadd_rn(__fmul_rn(gjx_normal_at(keys.k[m][0][0], keys.k[m][0][1], gi), w.sig[0]), m0);
      if (sel_mask & 2) ny = __fadd_rn(__fmul_rn(gjx_normal_at(keys.k[m][1][0], keys.k[m][1][1], gi), w.sig[1]), m1);
      if (sel_mask & 4) nt = __fadd_rn(__fmul_rn(gjx_normal_at(keys.k[m][2][0], keys.k[m][2][1], gi), w.sig[2]), m2);
      const float u = gjx_uniform_at(keys.k[m][3][0], keys.k[m][3][1], gi, 0.0f, 1.0f);
      float lp_new[RMAX];
      (void)meas_logp<RMAX>(w, nx, ny, nt, lp_new);
      // weight = sum_r [logpdf_r(new) + old_score_r]  (core.py:1649-1657, 1474-1492)
      float wt = 0.0f;
#pragma unroll
      for (int r = 0; r < RMAX; ++r)
        if (r < w.R) wt += lp_new[r] - lp_old[r];
      const float log_alpha = fminf(0.0f, wt);
      const bool accept = live && (logf(u) < log_alpha);
      cx = accept ? nx : cx;
      cy = accept ? ny : cy;
      ct = accept ? nt : ct;
#pragma unroll
      for (int r = 0; r < RMAX; ++r)
        if (r < w.R) lp_old[r] = accept ? lp_new[r] : lp_old[r];
      count_accepts(counts, m, accept);
    }
    if (live) {
      x[i] = cx;
      y[i] = cy;
      th[i] = ct;
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < n_moves)
    accept_partials[(int64_t)blockIdx.x * GJX_MAX_MOVES + threadIdx.x] = counts[threadIdx.x];
}

template <int RMAX>
__global__ void __launch_bounds__(GJX_BLOCK)
    k_loc_hmc(LocK w, MoveKeys keys, int n_moves, int sel_mask, float eps, float half_eps,
              int n_steps, uint64_t offset, int64_t n, const float* x_in, const float* y_in,
              const float* th_in, const float* __restrict__ mx, const float* __restrict__ my,
              const float* __restrict__ mth, float* x, float* y, float* th,
              uint32_t* __restrict__ accept_partials) {
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
    float cur_logp = logp_grad<RMAX>(w, q[0], q[1], q[2], m0, m1, m2, inv_s, log_s, g);
    for (int m = 0; m < n_moves; ++m) {
      // momenta in sorted-key order theta, x, y (mcmc.py:497; [3P] dict tree_map order)
      // unselected coordinates get zero momentum and a masked gradient: they stay put and
      // their (constant) kinetic terms cancel in log_alpha
      const float pth = (sel_mask & 4) ? gjx_normal_at(keys.k[m][0][0], keys.k[m][0][1], gi) : 0.0f;
      const float ppx = (sel_mask & 1) ? gjx_normal_at(keys.k[m][1][0], keys.k[m][1][1], gi) : 0.0f;
      const float ppy = (sel_mask & 2) ? gjx_normal_at(keys.k[m][2][0], keys.k[m][2][1], gi) : 0.0f;
      const float u = gjx_uniform_at(keys.k[m][3][0], keys.k[m][3][1], gi, 0.0f, 1.0f);
      float p[3] = {ppx, ppy, pth};
      const float prev_mom = ((-0.5f * (p[2] * p[2]) - HALF_LOG_2PI_F) +
                              (-0.5f * (p[0] * p[0]) - HALF_LOG_2PI_F)) +
                             (-0.5f * (p[1] * p[1]) - HALF_LOG_2PI_F);
      float qq[3] = {q[0], q[1], q[2]};
      float gg[3] = {g[0], g[1], g[2]};
      float new_logp = cur_logp;
      for (int l = 0; l < n_steps; ++l) {
#pragma unroll
        for (int k = 0; k < 3; ++k) p[k] = p[k] + half_eps * (gg[k] * msk[k]);
#pragma unroll
        for (int k = 0; k < 3; ++k) qq[k] = qq[k] + eps * p[k];
        new_logp = logp_grad<RMAX>(w, qq[0], qq[1], qq[2], m0, m1, m2, inv_s, log_s, gg);
#pragma unroll
        for (int k = 0; k < 3; ++k) p[k] = p[k] + half_eps * (gg[k] * msk[k]);
      }
      const float new_mom = ((-0.5f * (p[2] * p[2]) - HALF_LOG_2PI_F) +
                             (-0.5f * (p[0] * p[0]) - HALF_LOG_2PI_F)) +
                            (-0.5f * (p[1] * p[1]) - HALF_LOG_2PI_F);
      float log_alpha = (new_logp + new_mom) - (cur_logp + prev_mom);
      log_alpha = fminf(0.0f, log_alpha);
      const bool accept = live && (logf(u) < log_alpha);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        q[k] = accept ? qq[k] : q[k];
        g[k] = accept ? gg[k] : g[k];
      }
      cur_logp = accept ? new_logp : cur_logp;
      count_accepts(counts, m, accept);
    }
    if (live) {
      x[i] = q[0];
      y[i] = q[1];
      th[i] = q[2];
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < n_moves)
    accept_partials[(int64_t)blockIdx.x * GJX_MAX_MOVES + threadIdx.x] = counts[threadIdx.x];
}

template <int RMAX>
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
  for (int64_t i = (int64_t)blockIdx.x * GJX_BLOCK + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * GJX_BLOCK) {
    float g[3];
    logp[i] = logp_grad<RMAX>(w, x[i], y[i], th[i], mx[i], my[i], mth[i], inv_s, log_s, g);
    grad[i] = g[0];
    grad[n + i] = g[1];
    grad[2 * n + i] = g[2];
  }
}

__global__ void __launch_bounds__(GJX_BLOCK)
    k_accept_reduce(const uint32_t* __restrict__ partials, int nblocks, int n_moves,
                    int64_t* __restrict__ out) {
  // one wave per move would do; a block-wide loop keeps it simple and deterministic
  __shared__ unsigned long long sm[GJX_BLOCK];
  for (int m = 0; m < n_moves; ++m) {
    unsigned long long s = 0;
    for (int b = threadIdx.x; b < nblocks; b += GJX_BLOCK) s += partials[(int64_t)b * GJX_MAX_MOVES + m];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int o = GJX_BLOCK / 2; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) out[m] = (int64_t)sm[0];
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
#define GJX_DISPATCH_R(R, CALL)             \
  do {                                      \
    if ((R) <= 8) {                         \
      constexpr int RM = 8;                 \
      CALL;                                 \
    } else if ((R) <= 16) {                 \
      constexpr int RM = 16;                \
      CALL;                                 \
    } else {                                \
      constexpr int RM = 32;                \
      CALL;                                 \
    }                                       \
  } while (0)

GJX_API int32_t gjx_loc_lidar(const gjx_loc_world* world, const float* d_px, const float* d_py,
                              const float* d_theta, int64_t n, float* d_out, void* stream) {
  GJX_REQUIRE(d_px && d_py && d_theta && d_out, GJX_EINVAL, "gjx_loc_lidar: null argument");
  GJX_REQUIRE(n > 0, GJX_EINVAL, "gjx_loc_lidar: n must be positive");
  LocK k;
  int32_t rc = build_lock(world, nullptr, 0, &k, "gjx_loc_lidar");
  if (rc != GJX_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  GJX_DISPATCH_R(k.R, (k_loc_lidar<RM><<<gjx_grid_for(n), GJX_BLOCK, 0, s>>>(k, d_px, d_py, d_theta, n, d_out)));
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

GJX_API int32_t gjx_loc_extend(gjx_ctx* ctx, const gjx_loc_world* world, const uint32_t keys[3][2],
                               int32_t is_initial, uint64_t offset, int64_t n, const float* obs,
                               const float* d_px, const float* d_py, const float* d_pth,
                               const float* d_logw_in, float* d_x, float* d_y, float* d_th,
                               float* d_mx, float* d_my, float* d_mth, float* d_logw, void* stream) {
  (void)ctx;
  GJX_REQUIRE(keys && obs && d_x && d_y && d_th && d_mx && d_my && d_mth && d_logw && d_logw_in,
              GJX_EINVAL, "gjx_loc_extend: null argument");
  GJX_REQUIRE(is_initial || (d_px && d_py && d_pth), GJX_EINVAL,
              "gjx_loc_extend: previous pose columns are required unless is_initial");
  GJX_REQUIRE(n > 0, GJX_EINVAL, "gjx_loc_extend: n must be positive");
  LocK k;
  int32_t rc = build_lock(world, obs, is_initial != 0, &k, "gjx_loc_extend");
  if (rc != GJX_OK) return rc;
  Key3 k3;
  memcpy(k3.k, keys, sizeof(k3.k));
  hipStream_t s = (hipStream_t)stream;
  GJX_DISPATCH_R(k.R, (k_loc_extend<RM><<<gjx_grid_for(n), GJX_BLOCK, 0, s>>>(
                          k, k3, offset, n, d_px, d_py, d_pth, d_logw_in, d_x, d_y, d_th, d_mx, d_my,
                          d_mth, d_logw)));
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

static int32_t finish_accepts(gjx_ctx* ctx, int grid, int n_moves, int64_t* d_accept_counts,
                              hipStream_t s) {
  if (!d_accept_counts) return GJX_OK;
  k_accept_reduce<<<1, GJX_BLOCK, 0, s>>>(ctx->d_accept_partials, grid, n_moves, d_accept_counts);
  GJX_LAUNCH_CHECK();
  return GJX_OK;
}

GJX_API int32_t gjx_loc_mh(gjx_ctx* ctx, const gjx_loc_world* world, const uint32_t (*keys)[4][2],
                           int32_t n_moves, int32_t sel_mask, int32_t is_initial, uint64_t offset,
                           int64_t n, const float* obs, const float* d_x_in, const float* d_y_in,
                           const float* d_th_in, const float* d_mx, const float* d_my,
                           const float* d_mth, float* d_x, float* d_y, float* d_th,
                           int64_t* d_accept_counts, void* stream) {
  GJX_REQUIRE(ctx && keys && obs && d_x && d_y && d_th && d_x_in && d_y_in && d_th_in && d_mx &&
                  d_my && d_mth,
              GJX_EINVAL, "gjx_loc_mh: null argument");
  GJX_REQUIRE(sel_mask >= 1 && sel_mask <= 7, GJX_EINVAL, "gjx_loc_mh: sel_mask=%d not in [1, 7]",
              sel_mask);
  GJX_REQUIRE(n > 0, GJX_EINVAL, "gjx_loc_mh: n must be positive");
  GJX_REQUIRE(n_moves >= 1 && n_moves <= GJX_MAX_MOVES, GJX_ELIMIT,
              "gjx_loc_mh: n_moves=%d not in [1, %d]", n_moves, GJX_MAX_MOVES);
  LocK k;
  int32_t rc = build_lock(world, obs, is_initial != 0, &k, "gjx_loc_mh");
  if (rc != GJX_OK) return rc;
  MoveKeys mk;
  memset(&mk, 0, sizeof(mk));
  memcpy(mk.k, keys, sizeof(uint32_t) * 8 * n_moves);
  hipStream_t s = (hipStream_t)stream;
  const int grid = gjx_grid_for(n);
  GJX_DISPATCH_R(k.R, (k_loc_mh<RM><<<grid, GJX_BLOCK, 0, s>>>(
                          k, mk, n_moves, sel_mask, offset, n, d_x_in, d_y_in, d_th_in, d_mx, d_my,
                          d_mth, d_x, d_y, d_th, ctx->d_accept_partials)));
  GJX_LAUNCH_CHECK();
  return finish_accepts(ctx, grid, n_moves, d_accept_counts, s);
}

GJX_API int32_t gjx_loc_hmc(gjx_ctx* ctx, const gjx_loc_world* world, const uint32_t (*keys)[4][2],
                            int32_t n_moves, int32_t sel_mask, float step_size, int32_t n_steps,
                            int32_t is_initial, uint64_t offset, int64_t n, const float* obs,
                            const float* d_x_in, const float* d_y_in, const float* d_th_in,
                            const float* d_mx, const float* d_my, const float* d_mth, float* d_x,
                            float* d_y, float* d_th, int64_t* d_accept_counts, void* stream) {
  GJX_REQUIRE(ctx && keys && obs && d_x && d_y && d_th && d_x_in && d_y_in && d_th_in && d_mx &&
                  d_my && d_mth,
              GJX_EINVAL, "gjx_loc_hmc: null argument");
  GJX_REQUIRE(sel_mask >= 1 && sel_mask <= 7, GJX_EINVAL, "gjx_loc_hmc: sel_mask=%d not in [1, 7]",
              sel_mask);
  GJX_REQUIRE(n > 0, GJX_EINVAL, "gjx_loc_hmc: n must be positive");
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
  const int grid = gjx_grid_for(n);
  const float half_eps = (float)((double)step_size / 2.0);  // (step_size / 2.0) * g, mcmc.py:516
  GJX_DISPATCH_R(k.R, (k_loc_hmc<RM><<<grid, GJX_BLOCK, 0, s>>>(
                          k, mk, n_moves, sel_mask, step_size, half_eps, n_steps, offset, n, d_x_in,
                          d_y_in, d_th_in, d_mx, d_my, d_mth, d_x, d_y, d_th,
                          ctx->d_accept_partials)));
  GJX_LAUNCH_CHECK();
  return finish_accepts(ctx, grid, n_moves, d_accept_counts, s);
}

GJX_API int32_t gjx_loc_logp_grad(const gjx_loc_world* world, int32_t is_initial, int64_t n,
                                  const float* obs, const float* d_x, const float* d_y,
                                  const float* d_th, const float* d_mx, const float* d_my,
                                  const float* d_mth, float* d_logp, float* d_grad3,
                                  void* stream) {
  GJX_REQUIRE(obs && d_x && d_y && d_th && d_mx && d_my && d_mth && d_logp && d_grad3, GJX_EINVAL,
              "gjx_loc_logp_grad: null argument");
  GJX_REQUIRE(n > 0, GJX_EINVAL, "gjx_loc_logp_grad: n must be positive");
  LocK k;
  int32_t rc = build_lock(world, obs, is_initial != 0, &k, "gjx_loc_logp_grad");
  if (rc != GJX_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  GJX_DISPATCH_R(k.R, (k_loc_logp_grad<RM><<<gjx_grid_for(n), GJX_BLOCK, 0, s>>>(
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
