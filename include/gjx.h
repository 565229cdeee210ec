/*
 * gjx.h -- C ABI of libgjx_hip.so: the MI355X (gfx950) backend for the
 * vectorized-inference hot path of probcomp/genjax.
 *
 * The reference has NO native boundary (it is pure Python over JAX/TFP, SURVEY.md
 * section 8b): each entry point below therefore cites the reference *Python*
 * function whose per-particle arithmetic it replaces (file:line relative to the
 * reference checkout).  The reference-side binding is a ctypes stub; see
 * INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 on success or a negative GJX_E* code and never
 *     throws; gjx_last_error() returns a thread-local description;
 *   - pointers named d_* are DEVICE pointers owned by the caller (e.g. the
 *     data_ptr() of a torch tensor); all other pointers are HOST pointers to
 *     small parameter blocks that are copied into the kernel arguments;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *     calls only enqueue work, they do not synchronise;
 *   - a gjx_ctx owns the per-GPU workspace (reduction partials, scan totals);
 *     one context per GPU and per stream, not thread-safe;
 *   - keys are jax.random-style Threefry keys: two uint32 words (hi, lo);
 *   - `offset` is the GLOBAL index of the first particle of this shard: draws
 *     are counter-based on the global index, so any sharding of the particle
 *     axis reproduces the single-device stream bit for bit;
 *   - per-particle entry points (draws, log-densities, extend / mh / hmc,
 *     gather, normalised weights) accept n == 0 as a no-op; reductions and
 *     resampling over an empty set are GJX_EINVAL (the reference's jnp.max
 *     of an empty array raises as well).
 */
#ifndef GJX_H
#define GJX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GJX_OK 0
#define GJX_EINVAL (-1)   /* bad argument (null pointer, size out of range) */
#define GJX_EHIP (-2)     /* a HIP runtime call failed */
#define GJX_ENOMEM (-3)   /* workspace allocation failed */
#define GJX_ELIMIT (-4)   /* size exceeds a compiled-in limit (rays, walls, moves) */

#define GJX_MAX_RAYS 32
#define GJX_MAX_WALLS 32
#define GJX_MAX_MOVES 64
#define GJX_MAX_COLS 16
#define GJX_MAX_PIECES 64   /* ranges per gjx_shard_pack call (= max ranks of a sharded run) */
#define GJX_MAX_POINTS 64

typedef struct gjx_ctx gjx_ctx;

int32_t gjx_version(void);
const char* gjx_last_error(void);

/* Workspace for up to n_max particles on `device`. */
int32_t gjx_ctx_create(int32_t device, int64_t n_max, gjx_ctx** out);
int32_t gjx_ctx_destroy(gjx_ctx* ctx);

/* ---- host-side key arithmetic ([3P] jax/_src/prng.py; call sites pjax.py:1337,1376) ---- */
void gjx_key(uint64_t seed, uint32_t key[2]);                         /* jax.random.key(seed) */
void gjx_threefry2x32(const uint32_t key[2], const uint32_t ctr[2], uint32_t out[2]);
void gjx_split(const uint32_t key[2], uint32_t n, uint32_t* out);    /* jax.random.split, n x 2 */
void gjx_fold_in(const uint32_t key[2], uint32_t data, uint32_t out[2]); /* jax.random.fold_in */

/* ---- counter-based draws: one key => n values laid out by global index ----
 * replaces the sampler side of tfp_distribution (core.py:1748-1750) as batched by
 * VmapBatchHandler (pjax.py:860-908): bits / uniform / normal of jax.random. */
int32_t gjx_random_bits_u32(const uint32_t key[2], uint64_t offset, int64_t n, uint32_t* d_out,
                            void* stream);
int32_t gjx_random_uniform_f32(const uint32_t key[2], uint64_t offset, int64_t n, float minval,
                               float maxval, float* d_out, void* stream);
/* out = normal(key)[i] * scale + loc  (tfd.Normal.sample; loc/scale are scalars) */
int32_t gjx_random_normal_f32(const uint32_t key[2], uint64_t offset, int64_t n, float loc,
                              float scale, float* d_out, void* stream);

/* The same samplers with per-particle (column) parameters, for the generic GFI path: a
 * parameter array has n elements when its stride is 1 and is a scalar when it is 0.
 *   normal:      normal(key)[i] * scale[i] + loc[i]            (tfd.Normal.sample)
 *   uniform:     low[i] + (high[i] - low[i]) * uniform(key)[i] (tfd.Uniform.sample)
 *   bernoulli:   uniform(key)[i] < p[i], as 0.0 / 1.0          (tfd.Bernoulli.sample; flip)
 *   categorical: Gumbel-max, [3P] TFP noise layout logits.shape + (n_samples,): per-row
 *                logits (stride 1, n x k, one draw per row) or shared logits (stride 0, n draws)
 *   mvnormal:    mu + chol(cov) @ normal(key, (n, d)), d <= 8  (MultivariateNormalFullCovariance) */
int32_t gjx_random_normal_cols_f32(const uint32_t key[2], uint64_t offset, int64_t n,
                                   const float* d_loc, int32_t loc_stride, const float* d_scale,
                                   int32_t scale_stride, float* d_out, void* stream);
int32_t gjx_random_uniform_cols_f32(const uint32_t key[2], uint64_t offset, int64_t n,
                                    const float* d_lo, int32_t lo_stride, const float* d_hi,
                                    int32_t hi_stride, float* d_out, void* stream);
int32_t gjx_random_bernoulli_f32(const uint32_t key[2], uint64_t offset, int64_t n, const float* d_p,
                                 int32_t p_stride, float* d_out, void* stream);
int32_t gjx_random_categorical_i32(const uint32_t key[2], uint64_t offset, int64_t n,
                                   const float* d_logits, int32_t logits_stride, int32_t k,
                                   int32_t* d_out, void* stream);
int32_t gjx_random_mvnormal_f32(const uint32_t key[2], uint64_t offset, int64_t n, int32_t d,
                                const float* d_mu, int32_t mu_stride, const float* d_cov,
                                int32_t cov_stride, float* d_out, void* stream);

/* beta(a, b).sample (distributions.py:62-65).  [3P] tfd.Beta's rejection sampler is not reproduced: the same
 * law as Ga / (Ga + Gb) with Marsaglia-Tsang gammas on counter-based draws of the site key (different bits).
 * bernoulli(logits).sample (distributions.py:22-25): uniform(key)[i] < sigmoid(logits[i]), as 0.0 / 1.0. */
int32_t gjx_random_beta_f32(const uint32_t key[2], uint64_t offset, int64_t n, const float* d_a,
                            int32_t a_stride, const float* d_b, int32_t b_stride, float* d_out, void* stream);
int32_t gjx_random_bernoulli_logits_f32(const uint32_t key[2], uint64_t offset, int64_t n,
                                        const float* d_logits, int32_t logits_stride, float* d_out,
                                        void* stream);

/* ---- element-wise log densities (distributions.py:22-25,51-54,62-65,92-95,159-162,194-197,
 * 298-301; log_prob call site core.py:1752-1754).  Parameter arrays have n elements when the
 * matching stride is 1 and are read at index 0 when it is 0 (scalar broadcast). */
int32_t gjx_normal_logpdf_f32(const float* d_x, const float* d_mu, int32_t mu_stride,
                              const float* d_sigma, int32_t sigma_stride, int64_t n, float* d_out,
                              void* stream);
int32_t gjx_uniform_logpdf_f32(const float* d_x, const float* d_lo, int32_t lo_stride,
                               const float* d_hi, int32_t hi_stride, int64_t n, float* d_out,
                               void* stream);
int32_t gjx_bernoulli_logpdf_f32(const float* d_x, const float* d_p, int32_t p_stride, int64_t n,
                                 float* d_out, void* stream);
int32_t gjx_beta_logpdf_f32(const float* d_x, const float* d_a, int32_t a_stride, const float* d_b,
                            int32_t b_stride, int64_t n, float* d_out, void* stream);
/* bernoulli(logits) (distributions.py:22-25; [3P] tfd.Bernoulli(logits=l).log_prob):
 * -x softplus(-l) - (1 - x) softplus(l), finite for any |l| */
int32_t gjx_bernoulli_logits_logpdf_f32(const float* d_x, const float* d_logits, int32_t logits_stride,
                                        int64_t n, float* d_out, void* stream);
/* Gradients of the log densities above, times an upstream gradient (what jax.grad of assess produces inside
 * hmc, mcmc.py:59-88, 508, 523).  kind 0 normal (p1 = mu, p2 = sigma), 1 uniform (lo, hi), 2 beta (a, b),
 * 3 bernoulli(p) (p1 = p, p2 unused).  Outputs (nullable): upstream x d/dx, x d/dp1, x d/dp2. */
int32_t gjx_logpdf_grad_f32(int32_t kind, const float* d_x, const float* d_p1, int32_t p1_stride,
                            const float* d_p2, int32_t p2_stride, const float* d_upstream,
                            int32_t upstream_stride, int64_t n, float* d_gx, float* d_g1, float* d_g2,
                            void* stream);
/* logits: n x k row-major when logits_stride = 1, one shared row of k when 0 */
int32_t gjx_categorical_logpdf_f32(const int32_t* d_x, const float* d_logits, int32_t logits_stride,
                                   int64_t n, int32_t k, float* d_out, void* stream);
/* x: n x d; mu: n x d or d (stride 0); cov: n x d x d or d x d (stride 0); d <= 8 */
int32_t gjx_mvnormal_logpdf_f32(const float* d_x, const float* d_mu, int32_t mu_stride,
                                const float* d_cov, int32_t cov_stride, int64_t n, int32_t d,
                                float* d_out, void* stream);

/* ---- Metropolis-Hastings accept / select for generic traces (mcmc.py:293-305) ----
 * accept[i] = log(uniform(key)[offset + i]) < jnp.minimum(0, log_weight[i]) (NaN rejects);
 * log_weight has n elements (stride 1) or is a scalar (stride 0). */
int32_t gjx_mh_accept(const uint32_t key[2], uint64_t offset, int64_t n, const float* d_log_weight,
                      int32_t weight_stride, uint8_t* d_out, void* stream);
/* jax.lax.select(accept, new_leaf, old_leaf) over leaves of n rows x inner 4-byte words */
int32_t gjx_select_rows_b32(const uint8_t* d_accept, const void* d_new, const void* d_old, int64_t n,
                            int64_t inner, void* d_out, void* stream);

/* ---- per-particle column operations of the inference surface ----
 * out = y + a x, multiply and add rounded separately: the leapfrog updates of hmc (mcmc.py:515-528) */
int32_t gjx_axpy_f32(float a, const float* d_x, int32_t x_stride, const float* d_y, int32_t y_stride,
                     int64_t n, float* d_out, void* stream);
/* out = (a + b) - (c + d): hmc's log_alpha (mcmc.py:559-561); d = NULL: (a + b) - c, mala's log_alpha
 * (mcmc.py:414); c = d = NULL: a + b */
int32_t gjx_sum_diff_f32(const float* d_a, const float* d_b, const float* d_c, const float* d_d, int64_t n,
                         float* d_out, void* stream);
/* out = a - *scalar: the diagnostic weights logw - logsumexp(logw) (smc.py:259-261) */
int32_t gjx_sub_scalar_f32(const float* d_a, const float* d_scalar, int64_t n, float* d_out, void* stream);
/* ParticleCollection.estimate (smc.py:206-245): out[k] = sum_i exp(logw[i] - *lse) v[i * inner + k] */
int32_t gjx_weighted_sum_f32(gjx_ctx* ctx, const float* d_logw, const float* d_lse, const float* d_values,
                             int64_t n, int32_t inner, float* d_out, void* stream);

/* ---- log-weight normalisation (smc.py:24-76 effective_sample_size; logsumexp call sites
 * smc.py:74,127,201,232,261,623,628).  Two-pass: max, then sum exp and sum exp^2.
 * d_out5 receives {max, logsumexp, ess, sum exp(a-max), sum exp(a-max)^2}. */
int32_t gjx_logsumexp_ess(gjx_ctx* ctx, const float* d_logw, int64_t n, float* d_out5,
                          void* stream);
/* The same two passes exposed separately for sharded runs: the caller reduces the
 * per-shard scalars across ranks (rank-ascending) between the passes. */
int32_t gjx_max_f32(gjx_ctx* ctx, const float* d_a, int64_t n, float* d_out1, void* stream);
int32_t gjx_sumexp_f32(gjx_ctx* ctx, const float* d_a, const float* d_max1, int64_t n,
                       float* d_out2 /* sum exp, sum exp^2 */, void* stream);

/* ---- resampling (smc.py:79-136 systematic_resample, :139-175 resample_vectorized_trace) ---- */
/* w[i] = exp(logw[i] - *d_lse)   (smc.py:127-128) */
int32_t gjx_normalized_weights(const float* d_logw, const float* d_lse, int64_t n, float* d_w,
                               void* stream);
/* cdf = jnp.cumsum(w) with the lax.associative_scan association (bit-exact). */
int32_t gjx_cumsum_f32(gjx_ctx* ctx, const float* d_w, int64_t n, float* d_cdf, void* stream);
/* Sharded form of the same scan (shard sizes must be multiples of 2048): per-shard balanced-tree
 * tile totals; the caller all-gathers the totals of every shard, scans them with
 * gjx_cumsum_f32, and applies the in-tile fold against the global scan. */
int32_t gjx_cumsum_tile_totals(const float* d_w, int64_t n, float* d_totals, void* stream);
int32_t gjx_cumsum_apply(const float* d_w, int64_t n, const float* d_scanned_totals,
                         int64_t tile_offset, float* d_cdf, void* stream);
/* idx[j] = min(searchsorted(cdf, (j0 + j + u) / n_total, side='left'), n_cdf - 1) for
 * j in [0, m); emulates jnp.searchsorted's fixed-depth bisection exactly. */
int32_t gjx_searchsorted_systematic(const float* d_cdf, int64_t n_cdf, float u, int64_t j0,
                                    int64_t m, int64_t n_total, int32_t* d_idx, void* stream);
/* Two-level form of the same bisection for a CDF that is only partly resident (sharded runs).
 * The first lc levels probe d_coarse: cdf[mid] of every node of the bisection tree's first lc
 * levels in heap order (node 1 = root, children 2i and 2i + 1; 2^lc entries, entry 0 unused);
 * gjx_bisect_plan (host) lists the mids so that the owners can fill the table.  The remaining
 * levels probe the slab d_cdf_slab = cdf[slab_base .. slab_base + slab_len), which must cover
 * [low, high) as reported by gjx_searchsorted_window for the same positions: the low bound of
 * the first position and the high bound of the last one after lc levels (positions are
 * increasing and every decision is monotone in the position, so all others lie between).
 * gjx_searchsorted_window reports n_ranges consecutive ranges of m positions starting at j0
 * (d_out: n_ranges x {low, high}), so one call gives every rank the windows of ALL ranks.
 * Results are global indices, identical to gjx_searchsorted_systematic on the full CDF. */
int32_t gjx_bisect_levels(int64_t n_cdf);   /* ceil(log2(n_cdf + 1)), the fixed depth */
int32_t gjx_bisect_plan(int64_t n_cdf, int32_t lc, int32_t* probe_idx /* host, 2^lc */);
int32_t gjx_searchsorted_window(const float* d_coarse, int32_t lc, int64_t n_cdf, float u,
                                int64_t j0, int64_t m, int32_t n_ranges, int64_t n_total,
                                int64_t* d_out, void* stream);
int32_t gjx_searchsorted_systematic_2l(const float* d_coarse, int32_t lc, const float* d_cdf_slab,
                                       int64_t slab_base, int64_t slab_len, int64_t n_cdf, float u,
                                       int64_t j0, int64_t m, int64_t n_total, int32_t* d_idx,
                                       void* stream);
/* cumsum(exp(logw - *d_lse)) with the normalisation fused into the scan's loads (same bits as
 * gjx_normalized_weights followed by gjx_cumsum_f32), and systematic_resample on top of it. */
int32_t gjx_cumsum_normalized_f32(gjx_ctx* ctx, const float* d_logw, const float* d_lse, int64_t n,
                                  float* d_cdf, void* stream);
int32_t gjx_resample_systematic_logw(gjx_ctx* ctx, const float* d_logw, const float* d_lse, int64_t n,
                                     float u, int32_t* d_ancestors, float* d_cdf_out /* nullable */,
                                     void* stream);
/* gjx_resample_systematic_logw for callers that CANNOT ALLOCATE: code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  No counterpart in the reference (which
 * has no allocation discipline to speak of: XLA owns its buffers).  It takes no gjx_ctx: the CDF
 * and the scan's tile totals live in a static device array of the library (16.0 MiB, held by
 * every GPU the library runs on until the process ends), so the call allocates nothing -- no
 * hipMalloc / hipFree, no allocating library routine.  Same kernels and same bits as
 * gjx_resample_systematic_logw.  n > GJX_STATIC_MAX_N is GJX_ELIMIT.  The stream's device must
 * be current.  Every call on a GPU shares that GPU's array: calls that overlap on one GPU
 * (different streams or threads) race; everything a call reads it has written itself, so
 * back-to-back calls need no clearing.  The array's address is resolved once per device, on the
 * first call there or by gjx_ctx_create on that device: make that first call outside a graph
 * capture. */
#define GJX_STATIC_MAX_N (1 << 22)
int32_t gjx_resample_systematic_logw_static(const float* d_logw, const float* d_lse, int64_t n, float u,
                                            int32_t* d_ancestors, void* stream);
/* systematic_resample given normalised weights and u: cumsum + searchsorted. */
int32_t gjx_resample_systematic(gjx_ctx* ctx, const float* d_weights, int64_t n, float u,
                                int32_t* d_ancestors, float* d_cdf_out /* nullable */,
                                void* stream);
/* categorical.sample(logw, sample_shape=(m,)) as Gumbel-max over an n x m noise matrix
 * (smc.py:160-162); O(n*m), intended for n <= 2^16. */
int32_t gjx_resample_multinomial(const uint32_t key[2], const float* d_logw, int64_t n, int64_t m,
                                 int32_t* d_ancestors, void* stream);
/* dst[c][j] = src[c][idx[j] - idx_base] for c < ncols, j < m  (leaf[indices], smc.py:170-174) */
int32_t gjx_gather_columns_f32(const int32_t* d_idx, int64_t idx_base, int64_t m,
                               const float* const* src_cols, float* const* dst_cols,
                               int32_t ncols, void* stream);

/* ---- localization model family (examples/localization/core.py:81-128,252-395) ---- */
typedef struct {
  float width, height;          /* World.width/height (core.py:49-50) */
  int32_t n_walls;              /* <= GJX_MAX_WALLS */
  int32_t n_rays;               /* <= GJX_MAX_RAYS */
  float walls[GJX_MAX_WALLS][4]; /* x1, y1, x2, y2 per wall (core.py:53-56) */
  float max_range;              /* 10.0 (core.py:253) */
  float sensor_sigma;           /* 0.3 (core.py:391) */
  float init_mean[3];           /* 1.5, 1.5, 0.0 (core.py:331-333) */
  float init_sigma[3];          /* 0.5, 0.5, 0.3 (core.py:341-343) */
  float drift_sigma[3];         /* 0.25, 0.25, 0.1 (core.py:336-338) */
  float clip_margin;            /* 0.1 (core.py:361-362) */
} gjx_loc_world;

/* distance_to_wall_lidar for n poses: d_out is n x n_rays row-major (core.py:252-308). */
int32_t gjx_loc_lidar(const gjx_loc_world* world, const float* d_px, const float* d_py,
                      const float* d_theta, int64_t n, float* d_out, void* stream);
/* gjx_loc_lidar plus the partial derivatives of every distance w.r.t. (px, py, theta): d_jac = 3 planes of n x n_rays
 * (d/dpx, d/dpy, d/dtheta), the derivative of a ray being that of its winning candidate (zero at the range cap) --
 * what jax.grad takes through distance_to_wall_lidar (core.py:252-308) inside a gradient move (mcmc.py:59-88). */
int32_t gjx_loc_lidar_jac(const gjx_loc_world* world, const float* d_px, const float* d_py, const float* d_theta,
                          int64_t n, float* d_out, float* d_jac, void* stream);

/* Host-side (no GPU) unfused-f32 restatement of the same function, used only to build the
 * synthetic data set (examples/localization/data.py:124-165) reproducibly, and the scalar
 * jax.random.normal(key) draw that data.py:156 makes per ray. */
int32_t gjx_loc_lidar_host(const gjx_loc_world* world, const float* px, const float* py,
                           const float* theta, int64_t n, float* out);
float gjx_host_normal(const uint32_t key[2]);

/* One particle-batched generate() of localization_model with only "measurements"
 * constrained: init (smc.py:272-317) when is_initial, else extend (smc.py:449-530).
 * In: the previous pose columns d_px,d_py,d_pth = model args prev_pose (ignored and nullable
 * when is_initial; they are clipped again, which is idempotent, so previous choices or previous
 * retvals both work) and d_logw_in.
 * Out: the prior means used (mx,my,mth = clipped previous pose, core.py:346-348,361-364),
 * the new choices (x,y,th, unclipped) and logw = logw_in + sum_r log N(obs_r; d_r, sigma).
 * Outputs may alias the matching inputs (element i is read before it is written).
 * keys: x, y, theta sub-keys in program order (core.py:356-358). obs: n_rays host floats. */
int32_t gjx_loc_extend(gjx_ctx* ctx, const gjx_loc_world* world, const uint32_t keys[3][2],
                       int32_t is_initial, uint64_t offset, int64_t n, const float* obs,
                       const float* d_px, const float* d_py, const float* d_pth,
                       const float* d_logw_in, float* d_x, float* d_y, float* d_th, float* d_mx,
                       float* d_my, float* d_mth, float* d_logw, void* stream);

/* extend() with the locally-optimal grid proposal (examples/localization/core.py:730-809 through
 * smc.py:493-509): per particle assess() of the model on a grid_size^3 grid of poses around its
 * previous choices (x, y within grid_radius and at least 0.5 inside the world, theta within
 * pi/6), first argmax, then x, y ~ N(sel, noise_std), theta ~ N(sel, noise_std / 2);
 * logw = logw_in + log p(new, obs | prev) - log q(new).  d_px, d_py, d_pth: the previous CHOICES
 * (old_choices; the model args are their clipped values).  keys: x, y, theta sites of the
 * proposal in program order.  Outputs as gjx_loc_extend; may alias the inputs. */
int32_t gjx_loc_extend_locopt(gjx_ctx* ctx, const gjx_loc_world* world, const uint32_t keys[3][2],
                              int32_t grid_size, float noise_std, float grid_radius, uint64_t offset,
                              int64_t n, const float* obs, const float* d_px, const float* d_py,
                              const float* d_pth, const float* d_logw_in, float* d_x, float* d_y,
                              float* d_th, float* d_mx, float* d_my, float* d_mth, float* d_logw,
                              void* stream);

/* n_moves fused mh(trace, selection) moves (mcmc.py:267-309 via rejuvenate smc.py:533-600).
 * sel_mask: bit 0 "x", bit 1 "y", bit 2 "theta" (sel("x")|sel("y")|sel("theta") = 7).
 * keys[m] = {x', y', theta', u} sub-keys of move m (slots of unselected sites are ignored).
 * d_*_in -> d_* may alias.  d_accept_counts (nullable): n_moves int64 acceptance counts. */
int32_t gjx_loc_mh(gjx_ctx* ctx, const gjx_loc_world* world, const uint32_t (*keys)[4][2],
                   int32_t n_moves, int32_t sel_mask, int32_t is_initial, uint64_t offset, int64_t n,
                   const float* obs, const float* d_x_in, const float* d_y_in, const float* d_th_in,
                   const float* d_mx, const float* d_my, const float* d_mth, float* d_x, float* d_y,
                   float* d_th, int64_t* d_accept_counts, void* stream);
/* gjx_loc_mh on freshly resampled particles: particle i starts from ancestor
 * d_ancestors[i] - anc_base of the source columns d_src6 = {x, y, th, mx, my, mth}
 * (the leaf[indices] gather of resample_vectorized_trace, smc.py:170-174, fused into the move
 * kernel's loads) and all six columns are written to d_dst6 (no column may alias its source). */
int32_t gjx_loc_mh_resampled(gjx_ctx* ctx, const gjx_loc_world* world, const uint32_t (*keys)[4][2],
                             int32_t n_moves, int32_t sel_mask, int32_t is_initial, uint64_t offset,
                             int64_t n, const float* obs, const int32_t* d_ancestors, int64_t anc_base,
                             const float* const* d_src6, float* const* d_dst6,
                             const float* d_ray_scores, int64_t scores_stride,
                             float* d_logw_reset /* nullable: n log-weights set to 0, smc.py:646 */,
                             int64_t* d_accept_counts, void* stream);
/* d_ray_scores (nullable): the per-ray measurement scores of the SOURCE particles for the same
 * observation, ray r at d_ray_scores[r * scores_stride + source index], as written by
 * gjx_loc_extend_scores -- the old_score the trace keeps per address (core.py:1649-1657,
 * 1416-1492).  With it the kernel skips the lidar solve of the starting pose (same bits).
 * d_ancestors may be NULL when scores are given (no resampling: source index = i, columns may
 * then be updated in place). */
/* The scores themselves for given poses: d_ray_scores[r * n + i] = 0.5 ((obs_r - d_r(pose_i)) / sigma)^2,
 * i.e. -log N(obs_r; d_r, sigma) minus the constant log(sigma sqrt(2 pi)) -- assess() of the
 * Vmap'd sensor_model_single_ray per ray (examples/localization/core.py:386-395, core.py:1434-1452). */
int32_t gjx_loc_ray_scores(const gjx_loc_world* world, const float* obs, const float* d_x,
                           const float* d_y, const float* d_th, int64_t n, float* d_ray_scores,
                           void* stream);
/* gjx_loc_extend that also stores those scores (nullable): ray r of particle i at
 * d_ray_scores[r * scores_stride + i], scores_stride >= n. */
int32_t gjx_loc_extend_scores(gjx_ctx* ctx, const gjx_loc_world* world, const uint32_t keys[3][2],
                              int32_t is_initial, uint64_t offset, int64_t n, const float* obs,
                              const float* d_px, const float* d_py, const float* d_pth,
                              const float* d_logw_in, float* d_x, float* d_y, float* d_th, float* d_mx,
                              float* d_my, float* d_mth, float* d_logw, float* d_ray_scores,
                              int64_t scores_stride, void* stream);


/* regenerate(trace, selection) of localization_model for every particle: Fn.regenerate
 * (core.py:2200-2219) over Distribution.regenerate (core.py:1638-1657) and Vmap.regenerate
 * (core.py:1474-1492), the first half of mh (mcmc.py:288).  keys = {x', y', theta', u} sub-keys
 * (slots of unselected sites are ignored).  Out: the proposed choices d_nx, d_ny, d_nth, the
 * weight d_weight = sum_r [logpdf_r(new) + old_score_r]; nullable: d_log_u = log(uniform(keys[3]))
 * as mh's accept test draws it (mcmc.py:295), d_new_scores = the proposal's per-ray scores
 * (layout of gjx_loc_extend_scores).  d_ray_scores (nullable): stored scores of the current pose. */
int32_t gjx_loc_regenerate(const gjx_loc_world* world, const uint32_t keys[4][2], int32_t sel_mask,
                           int32_t is_initial, uint64_t offset, int64_t n, const float* obs,
                           const float* d_x, const float* d_y, const float* d_th, const float* d_mx,
                           const float* d_my, const float* d_mth, const float* d_ray_scores,
                           int64_t scores_stride, float* d_nx, float* d_ny, float* d_nth, float* d_weight,
                           float* d_log_u, float* d_new_scores, int64_t new_scores_stride, void* stream);
/* The proposal half of hmc (mcmc.py:485-562) for every particle: momenta from keys =
 * {p_theta, p_x, p_y, u}, n_steps leapfrog steps; out: the trajectory's end point, the RAW
 * log_alpha = [logp(q_L) + log N(-p_L)] - [logp(q_0) + log N(p_0)] (mcmc.py:559-562, before the
 * minimum with 0); nullable: log(u), logp(q_0), logp(q_L). */
int32_t gjx_loc_hmc_propose(const gjx_loc_world* world, const uint32_t keys[4][2], int32_t sel_mask,
                            float step_size, int32_t n_steps, int32_t is_initial, uint64_t offset, int64_t n,
                            const float* obs, const float* d_x, const float* d_y, const float* d_th,
                            const float* d_mx, const float* d_my, const float* d_mth, float* d_qx, float* d_qy,
                            float* d_qth, float* d_log_alpha, float* d_log_u, float* d_logp_cur,
                            float* d_logp_new, void* stream);

/* n_moves fused hmc(trace, selection, step_size, n_steps) moves (mcmc.py:434-578).
 * keys[m] = {p_theta, p_x, p_y, u} sub-keys in draw order (sorted keys; unselected ignored). */
int32_t gjx_loc_hmc(gjx_ctx* ctx, const gjx_loc_world* world, const uint32_t (*keys)[4][2],
                    int32_t n_moves, int32_t sel_mask, float step_size, int32_t n_steps,
                    int32_t is_initial, uint64_t offset, int64_t n, const float* obs,
                    const float* d_x_in, const float* d_y_in, const float* d_th_in,
                    const float* d_mx, const float* d_my, const float* d_mth, float* d_x,
                    float* d_y, float* d_th, int64_t* d_accept_counts, void* stream);

/* ---- one smc_step (smc.py:731-767) of the localization family in two host calls ----
 * The step has one host decision (lax.cond on ESS < N // 2, smc.py:747-753); what precedes and
 * follows it is enqueued by one call each so that the stream is not starved by host overhead. */
/* host: sub-keys of the K rejuvenation moves (scan over moves, smc.py:756-765): move m runs on
 * fold_in(scan_key, m); kind 1 = mh, 2 = hmc; out = n_moves x 4 x 2 laid out as gjx_loc_mh /
 * gjx_loc_hmc expect */
int32_t gjx_loc_move_keys(const uint32_t scan_key[2], int32_t n_moves, int32_t kind, int32_t sel_mask,
                          uint32_t* out);
/* gjx_loc_extend followed by gjx_logsumexp_ess of the new log-weights (d_stats5) */
int32_t gjx_loc_smc_extend(gjx_ctx* ctx, const gjx_loc_world* world, const uint32_t keys[3][2],
                           int32_t is_initial, uint64_t offset, int64_t n, const float* obs,
                           const float* d_px, const float* d_py, const float* d_pth,
                           const float* d_logw_in, float* d_x, float* d_y, float* d_th, float* d_mx,
                           float* d_my, float* d_mth, float* d_logw, float* d_stats5,
                           float* d_ray_scores /* nullable: see gjx_loc_extend_scores */,
                           int64_t scores_stride, void* stream);
/* The scan and the search of the resampling a step may take (smc.py:127-134 on d_logw with the lse in d_stats5[1]),
 * enqueued BEFORE the host has read the ESS: its round trip for the lax.cond (smc.py:747-753) runs beside them.  The
 * ancestors stay in the context; gjx_loc_smc_resample_moves(do_resample = 2) uses them, (do_resample = 0) ignores them. */
int32_t gjx_loc_smc_presearch(gjx_ctx* ctx, const float* d_logw, const float* d_stats5, int64_t n, float u,
                              void* stream);
/* if do_resample: systematic resampling with uniform u (normalised weights from d_stats5[1],
 * CDF, bisection, gather of the six columns d_cols = {x, y, th, mx, my, mth} into d_cols_alt,
 * log-weights := 0); then n_moves fused moves (kind 0 none, 1 mh, 2 hmc) in place on the
 * columns that hold the particles (d_cols_alt after a resample, else d_cols).  do_resample = 2: as 1, with the
 * ancestors already searched by gjx_loc_smc_presearch on the same context and stream.  No counterpart in
 * the reference's API: it composes the entry points above exactly as resample() + rejuvenate() do. */
int32_t gjx_loc_smc_resample_moves(gjx_ctx* ctx, const gjx_loc_world* world, int32_t do_resample, float u,
                                   const float* d_stats5, int32_t kind, const uint32_t (*move_keys)[4][2],
                                   int32_t n_moves, int32_t sel_mask, float step_size, int32_t n_steps,
                                   int32_t is_initial, uint64_t offset, int64_t n, const float* obs,
                                   float* const* d_cols, float* const* d_cols_alt, float* d_logw,
                                   const float* d_ray_scores /* nullable: as written by gjx_loc_smc_extend
                                   for the SAME particles (before resampling) */,
                                   int64_t scores_stride,
                                   int64_t* d_accept_counts, void* ev_moves_begin, void* ev_moves_end,
                                   void* stream);
/* ev_moves_begin / ev_moves_end (nullable): hipEvent_t recorded on the stream right before /
 * after the move kernel, for callers that time that launch (bench.py).  Events come from: */
int32_t gjx_event_create(void** out);
/* Stage timing of gjx_loc_smc_extend / gjx_loc_smc_resample_moves: 7 events recorded on the launch stream
 * at the boundaries {before extend, after extend, after the normaliser, before the scan, after the scan,
 * after the search = before the moves, after the moves} of every following call on `ctx`; n = 0 ends it. */
int32_t gjx_ctx_set_stage_events(gjx_ctx* ctx, void* const* events, int32_t n);
int32_t gjx_event_record(void* ev, void* stream);
int32_t gjx_event_destroy(void* ev);
int32_t gjx_event_elapsed_ms(void* begin, void* end, float* ms);   /* waits for `end` */
/* one wave that occupies `stream` for about `microseconds` (<= 100000) and touches no memory; no counterpart in
 * the reference (a single-device library): the sharded driver's set-up probe uses it to find a second stream
 * that runs beside the main one */
int32_t gjx_spin_us(int32_t microseconds, void* stream);

/* ---- the same step with the particle axis sharded over ranks (new work: the
 * reference runs resample() on one device, smc.py:592-650).  Rank r holds the contiguous global
 * range [r n, (r + 1) n), n a multiple of 2048, n_ranks <= GJX_MAX_PIECES.  The host runs ONE
 * collective between consecutive calls (genjax_amd/parallel.py); no call here communicates.
 * Results (lse, ESS, ancestors, columns) have the bits of the single-device step for any n_ranks.
 *
 * Normaliser: all_gather(max) -> gjx_shard_tile_sums -> all_gather(tile sums) -> gjx_shard_stats_fold. */
/* gjx_loc_smc_extend without the sums: d_max1[0] = max of this shard's new log-weights */
int32_t gjx_loc_shard_extend(gjx_ctx* ctx, const gjx_loc_world* world, const uint32_t keys[3][2],
                             int32_t is_initial, uint64_t offset, int64_t n, const float* obs,
                             const float* d_px, const float* d_py, const float* d_pth, const float* d_logw_in,
                             float* d_x, float* d_y, float* d_th, float* d_mx, float* d_my, float* d_mth,
                             float* d_logw, float* d_max1, float* d_ray_scores, int64_t scores_stride,
                             void* stream);
/* {sum exp(a - M), sum exp(a - M)^2} of each of this shard's 2048-tiles, M = max(d_max_all[0..n_ranks)):
 * 2 n / 2048 floats, the same bits gjx_logsumexp_ess forms for these tiles on one device */
int32_t gjx_shard_tile_sums(const float* d_logw, int64_t n, const float* d_max_all, int32_t n_ranks,
                            float* d_tiles2, void* stream);
/* d_tiles_all: the tile sums of ALL shards in rank order (n_tiles_all = n_global / 2048), folded in
 * f64 in global tile order -> {max, lse, ess, sum exp, sum exp^2} of the global weight vector */
int32_t gjx_shard_stats_fold(const float* d_tiles_all, int64_t n_tiles_all, const float* d_max_all,
                             int32_t n_ranks, float* d_stats5, void* stream);
/* Resampling, owner-push: the rank that OWNS an ancestor finds the positions that select it and sends
 * every distinct selected particle once per destination rank (a rank receives <= one 28-byte row per
 * distinct remote ancestor of its n positions):
 *   gjx_shard_cdf_begin -> all_gather(d_msg) -> gjx_shard_cdf_finish -> gjx_shard_search ->
 *   all_gather(d_counts) -> gjx_shard_pack -> all_to_all(rows) -> gjx_shard_resample_moves
 * Probe lists (host, once per (n, n_ranks)): the nodes 1 .. 2^lc - 1 of the first lc bisection levels
 * (gjx_bisect_plan gives their mids) grouped by the rank whose shard holds the mid, p_cap slots per rank, node 0 =
 * empty slot: d_probe_nodes / d_probe_mids = this rank's row, *_all = all rows (n_ranks x p_cap). */
/* d_msg (n / 2048 + 4096 + 12 p_cap floats) = balanced-tree totals of the 2048-tiles of w = exp(logw - *d_lse)
 * (formed in the loads, never stored), the raw log-weights of the first and of the last tile, and one 12-float
 * record per owned probe: the addends that turn the (not yet known) carry of the probe's tile into cdf[mid] */
int32_t gjx_shard_cdf_begin(const float* d_logw, const float* d_lse, int64_t n, int32_t rank,
                            const int32_t* d_probe_nodes, const int32_t* d_probe_mids, int32_t p_cap, float* d_msg,
                            void* stream);
/* d_msg_all: the messages of all ranks in rank order.  Writes the tile totals of all shards
 * (d_all_totals, n_global / 2048), their scan (d_scanned), d_cdf_ext (2048 + n + 2048 floats) = the
 * slice [r n - 2048, (r + 1) n + 2048) of the GLOBAL jnp.cumsum (same association as one device; the
 * neighbours' tiles are recomputed from their raw log-weights with the same fold), and d_coarse
 * (2^lc floats, heap order as gjx_bisect_plan): cdf[mid] of EVERY probe, redone on every rank from the owners'
 * records against the scanned totals (the owner's operands in the owner's order: its bits; no reduction over ranks).
 * lc = gjx_bisect_levels(n_global) - 11: after lc levels a bisection interval spans <= 2048 entries. */
int32_t gjx_shard_cdf_finish(gjx_ctx* ctx, const float* d_logw, const float* d_lse, int64_t n,
                             const float* d_msg_all, int32_t n_ranks, int32_t rank,
                             const int32_t* d_probe_nodes_all, const int32_t* d_probe_mids_all, int32_t p_cap,
                             float* d_all_totals, float* d_scanned, float* d_cdf_ext, int32_t lc, float* d_coarse,
                             void* stream);
/* d_coarse: the complete table.  d_range[0..2) = the positions [ja, jb) that can select this shard,
 * d_cand_idx[j - ja] = their (global) ancestors, d_tile_counts = rows per 2048 candidates, d_counts[d] =
 * rows for destination rank d (0 for d = rank).  cand_cap (the capacity of d_cand_idx; d_tile_counts holds
 * cand_cap / 2048 + 1) must be >= n_global: all positions select one shard when it holds all the weight. */
int32_t gjx_shard_search(const float* d_coarse, int32_t lc, const float* d_cdf_ext, int32_t n_ranks,
                         int32_t rank, int64_t n, float u, int64_t cand_cap, int64_t* d_range,
                         int32_t* d_cand_idx, int32_t* d_tile_counts, int32_t* d_counts, void* stream);
/* The counting half of gjx_shard_search alone (d_tile_counts, d_counts from d_range / d_cand_idx): gjx_shard_front
 * leaves it to the caller, who runs it with the side work of the step -- the main move launch does not need it. */
int32_t gjx_shard_count(const int32_t* d_cand_idx, const int64_t* d_range, int32_t n_ranks, int32_t rank, int64_t n,
                        int32_t* d_tile_counts, int32_t* d_counts, void* stream);
/* d_rows: sum(d_counts) rows of 7 floats {first position that selected the particle (int32 bits), x, y,
 * th, mx, my, mth}, in position order = grouped by destination rank ascending */
int32_t gjx_shard_pack(const int32_t* d_cand_idx, const int64_t* d_range, const int32_t* d_tile_counts,
                       int32_t* d_tile_offsets, int32_t n_ranks, int32_t rank, int64_t n,
                       const float* const* d_cols6, float* d_rows, void* stream);
/* d_recv_rows: the n_recv rows received from all ranks (rank order).  They go to the halo
 * [n, n + n_recv) of the six CURRENT columns (each column has n + halo_cap elements; so have the score rows,
 * whose halo part is recomputed here); d_anc[jl] = local index of the ancestor of position rank n + jl;
 * then gather + log-weights := 0 + the K moves into d_cols_alt, exactly as gjx_loc_smc_resample_moves.
 * *d_err |= 1 if a position found no source (a host-side misuse; never clamped silently). */
int32_t gjx_shard_resample_moves(gjx_ctx* ctx, const gjx_loc_world* world, const int32_t* d_cand_idx,
                                 const int64_t* d_range, const float* d_recv_rows, int64_t n_recv, int32_t rank,
                                 int64_t n, int64_t halo_cap, int32_t* d_anc, int32_t* d_err, int32_t kind,
                                 const uint32_t (*move_keys)[4][2], int32_t n_moves, int32_t sel_mask,
                                 float step_size, int32_t n_steps, int32_t is_initial, const float* obs,
                                 float* const* d_cols, float* const* d_cols_alt, float* d_logw,
                                 float* d_ray_scores, int64_t scores_stride, int64_t* d_accept_counts,
                                 void* ev_moves_begin, void* ev_moves_end, void* stream);

/* One smc_step (smc.py:731-767) with mcmc_kernel = mh in ONE call and with the lax.cond (smc.py:747-753) taken on
 * the device: gjx_loc_smc_extend's work, then d_decision[2] = (ESS < n / 2) decided by the normaliser's fold kernel
 * (d_range = {0, take ? n : 0}; *d_lml += lse - log n when taken; d_trace2 = {ESS, take}), the scan and the search
 * (they return at once when the decision is "no"), and the move launch reading the decision: resampled from the
 * ancestors with log-weights := 0, or every particle in place with weights kept.  Output always in d_cols_alt (the
 * caller flips the column sets).  ev_stats (optional event) is recorded after the normaliser: the caller fetches
 * d_stats5 on another stream while the moves run, the host never stands between two kernels of a step.  Same
 * kernels on the same inputs as gjx_loc_smc_extend + gjx_loc_smc_resample_moves: same bits.  d_decision: int32[4] =
 * {1, n, take, flags}: the caller writes [0] = 1, [1] = n, [3] = 0 once; [3] |= 2 if a position disagrees with the
 * decision's range.  One GPU (PRNG offset 0). */
int32_t gjx_loc_smc_step_auto(gjx_ctx* ctx, const gjx_loc_world* world, const uint32_t keys[3][2],
                              int32_t is_initial, int64_t n, const float* obs, float u,
                              const uint32_t (*move_keys)[4][2], int32_t n_moves, int32_t sel_mask,
                              float* const* d_cols, float* const* d_cols_alt, float* d_logw, float* d_stats5,
                              float* d_ray_scores, int64_t scores_stride, int32_t* d_decision, int64_t* d_range,
                              float* d_lml, float* d_trace2, int64_t* d_accept_counts, void* ev_stats,
                              void* ev_moves_begin, void* ev_moves_end, void* stream);

/* The same resampling + mh moves with the moves of the own-sourced positions started before the host has sized
 * the all_to_all (the lax.cond of rejuvenation_smc, smc.py:752-760, taken on the device; results bit-identical to
 * gjx_shard_resample_moves, respectively to gjx_loc_smc_resample_moves without resampling):
 *   ... gjx_shard_search -> gjx_shard_split -> gjx_shard_moves_main   (one stream, no host decision in between)
 *   beside it: all_gather(d_counts) -> gjx_shard_pack; the host reads d_gstats / the row counts / d_split,
 *   all_to_all(rows) -> gjx_shard_moves_fixup (any stream)
 * gjx_shard_split: d_split[0..4) = {p0 + 1 (0: none), p1, go, flags}: [p0, p1) is the run of positions whose ancestor
 * is an own particle (positions and ancestors both increase: two bisections over the candidates' ancestors), go =
 * (d_gstats[2] = ESS) < n_global / 2; flags (d_split[3]) is left as it is: the caller zeroes it once and the move
 * launch ORs 2 into it; the n_moves acceptance counters are zeroed here (both move launches add). */
int32_t gjx_shard_split(const int32_t* d_cand_idx, const int64_t* d_range, int32_t rank, int64_t n,
                        const float* d_gstats, int64_t n_global, int32_t* d_split, int64_t* d_accept_counts,
                        int32_t n_moves, void* stream);
/* go: a position whose ancestor (d_cand_idx, as left by gjx_shard_search) is an own particle is resampled from it,
 * moved n_moves times and gets log-weight 0, the others are left untouched; d_split[3] |= 2 if a position's side
 * disagrees with [p0, p1).  Not go: every position is moved in place of itself, weights kept.  Output in
 * d_cols_alt either way (the caller flips the column sets). */
int32_t gjx_shard_moves_main(gjx_ctx* ctx, const gjx_loc_world* world, const int32_t* d_cand_idx,
                             const int64_t* d_range, int32_t rank, int64_t n, int32_t* d_split,
                             const uint32_t (*move_keys)[4][2], int32_t n_moves, int32_t sel_mask,
                             int32_t is_initial, const float* obs, float* const* d_cols, float* const* d_cols_alt,
                             float* d_logw, float* d_ray_scores, int64_t scores_stride, int64_t* d_accept_counts,
                             void* ev_moves_begin, void* ev_moves_end, void* stream);
/* Only when go: the received rows to the halo (their per-ray scores recomputed there), halo ancestors of the
 * positions [0, p0) and [p1, n) into d_anc_fix, and their moves.  May run on another stream than
 * gjx_shard_moves_main; the next step must wait for both. */
int32_t gjx_shard_moves_fixup(gjx_ctx* ctx, const gjx_loc_world* world, const float* d_recv_rows, int64_t n_recv,
                              int32_t rank, int64_t n, int64_t halo_cap, int64_t p0, int64_t p1,
                              int32_t* d_anc_fix, int32_t* d_err, const uint32_t (*move_keys)[4][2],
                              int32_t n_moves, int32_t sel_mask, int32_t is_initial, const float* obs,
                              float* const* d_cols, float* const* d_cols_alt, float* d_logw, float* d_ray_scores,
                              int64_t scores_stride, int64_t* d_accept_counts, void* stream);

/* ---- collectives from the C side: the front end of a sharded mh step as ONE call (the context
 * takes an RCCL communicator and the normaliser does its own exchange).  No counterpart in the reference (a
 * single-device library, src/genjax/core.py:1304-1317 is dead plumbing); semantics: resample(),
 * src/genjax/inference/smc.py:603-652, over a particle axis sharded by contiguous ranges.
 * gjx_comm: RCCL communicator bound at run time (dlopen librccl.so.1 -- in a process that has RCCL loaded already,
 * that copy); one rank needs none.  Bootstrap: rank 0 makes the id, the caller distributes its 128 bytes over
 * whatever channel the ranks already share, every rank calls gjx_comm_create with its GPU current. */
typedef struct gjx_comm gjx_comm;
int32_t gjx_comm_unique_id(uint8_t* out128);
int32_t gjx_comm_create(const uint8_t* id128 /* nullable when world == 1 */, int32_t rank, int32_t world,
                        gjx_comm** out);
int32_t gjx_comm_destroy(gjx_comm* comm);
/* d_recv[r * bytes .. (r + 1) * bytes) = rank r's d_send, enqueued on `stream` */
int32_t gjx_comm_all_gather(gjx_comm* comm, const void* d_send, void* d_recv, int64_t bytes_per_rank, void* stream);
/* the persistent buffers of one rank's sharded step (sizes as in the gjx_shard_* calls above) */
typedef struct {
  int64_t n, n_global;
  int32_t rank, n_ranks, lc, p_cap;
  float* cols[6];            /* the CURRENT column set {x, y, th, mx, my, mth}, each n + halo floats */
  float* logw;
  float* ray_scores;         /* nullable */
  int64_t scores_stride;
  float *max1, *max_all;     /* 1, n_ranks; *max1 = -inf between calls */
  float *tiles, *tiles_all;  /* 2 n / 2048, 2 n_global / 2048 */
  float* gstats;             /* 5 */
  float *msg, *msg_all;      /* msg_len, msg_len n_ranks */
  int64_t msg_len;           /* n / 2048 + 4096 + 12 p_cap */
  const int32_t *probe_nodes, *probe_mids;   /* n_ranks x p_cap */
  float *all_totals, *scanned, *cdf_ext, *table;
  int64_t* range;
  int32_t* cand_idx;
  int64_t cand_cap;
  int32_t *tile_counts, *counts, *split;
} gjx_shard_bufs;
/* gjx_loc_shard_extend -> all_gather -> gjx_shard_tile_sums -> all_gather -> gjx_shard_stats_fold ->
 * gjx_shard_cdf_begin -> all_gather -> gjx_shard_cdf_finish -> gjx_shard_search (WITHOUT its row counts: d_counts is
 * zeroed, gjx_shard_count is the caller's, after ev_split) -> gjx_shard_split, enqueued back
 * to back on `stream` (same kernels, same bits as the call-by-call form; the shard maximum is raised by the extend
 * kernel's own atomics instead of a fold launch: max is exact in any order).
 * move_keys (nullable): gjx_shard_moves_main follows in the same call, output in d_cols_alt (the caller flips the
 * column sets); ev_split (nullable hipEvent_t) is recorded between the split and the move launch: the side work of the
 * step (gjx_shard_count, the all_gather of the row counts, gjx_shard_pack on the columns in bufs->cols, the read-back)
 * waits for it, not for the moves. */
int32_t gjx_shard_front(gjx_ctx* ctx, gjx_comm* comm, const gjx_loc_world* world, const uint32_t keys[3][2],
                        int32_t is_initial, const float* obs, float u, const gjx_shard_bufs* bufs, int32_t n_moves,
                        int64_t* d_accept_counts, const uint32_t (*move_keys)[4][2], int32_t sel_mask,
                        float* const* d_cols_alt, void* ev_split, void* ev_moves_begin, void* ev_moves_end,
                        void* stream);

/* assess() of localization_model with every site constrained, plus its gradient w.r.t.
 * (x, y, theta): the function jax.grad differentiates inside hmc (mcmc.py:59-88,508). */
int32_t gjx_loc_logp_grad(const gjx_loc_world* world, int32_t is_initial, int64_t n,
                          const float* obs, const float* d_x, const float* d_y, const float* d_th,
                          const float* d_mx, const float* d_my, const float* d_mth, float* d_logp,
                          float* d_grad3 /* 3 x n: d/dx, d/dy, d/dtheta columns */, void* stream);

/* ---- curvefit polynomial importance sampling (examples/curvefit/core.py:57-136) ----
 * init(npoint_curve, (xs,), n, {"ys": {"obs": ys}}): a,b,c ~ N(0,1) from keys[0..2],
 * logw = sum_j log N(ys_j; a + b x_j + c x_j^2, sigma).  n_points <= GJX_MAX_POINTS. */
int32_t gjx_poly_is(const uint32_t keys[3][2], uint64_t offset, int64_t n, const float* xs,
                    const float* ys, int32_t n_points, float sigma, float* d_a, float* d_b,
                    float* d_c, float* d_logw, void* stream);

/* ---- linear-Gaussian state-space model (src/genjax/extras/state_space.py:377-431) ----
 * state ~ MVN(is_initial ? initial_mean : A prev, is_initial ? initial_cov : Q) @ "state",
 * obs ~ MVN(C state, R) @ "obs".  Matrices are row-major, d_state, d_obs <= 4. */
typedef struct {
  int32_t d_state, d_obs;
  float initial_mean[4];
  float initial_cov[16];
  float A[16];
  float Q[16];
  float C[16];   /* d_obs x d_state */
  float R[16];   /* d_obs x d_obs */
} gjx_lgssm_params;

/* Bootstrap-PF generate() with "obs" constrained: init (smc.py:272-317) when is_initial, else
 * extend (smc.py:449-530).  Columns are SoA: d_state rows of n floats.  d_prev (nullable when
 * is_initial) = previous state; outputs: the new state, the prior mean used (A prev or
 * initial_mean, needed by mh's regenerate) and logw = logw_in + log MVN(obs; C state, R).
 * key: the sub-key of the "state" site (draw index i * d_state + r). */
int32_t gjx_lgssm_extend(gjx_ctx* ctx, const gjx_lgssm_params* model, const uint32_t key[2],
                         int32_t is_initial, uint64_t offset, int64_t n, const float* obs,
                         const float* d_prev, const float* d_logw_in, float* d_state, float* d_mean,
                         float* d_logw, void* stream);
/* n_moves fused mh(trace, sel("state")) moves (mcmc.py:267-309); keys[m] = {state', u}. */
int32_t gjx_lgssm_mh(gjx_ctx* ctx, const gjx_lgssm_params* model, const uint32_t (*keys)[2][2],
                     int32_t n_moves, int32_t is_initial, uint64_t offset, int64_t n, const float* obs,
                     const float* d_state_in, const float* d_mean, float* d_state,
                     int64_t* d_accept_counts, void* stream);

/* seed(rejuvenation_smc)(key, linear_gaussian, None, mh(sel("state")) | None, {"obs": T rows}, ..., systematic)
 * (smc.py:655-785) in one call, with the Seed key schedule (pjax.py:1318-1401) walked on the host and the lax.cond
 * on ESS < N // 2 (smc.py:747-753) decided ON THE DEVICE, so that the host never waits: per step extend ->
 * normaliser (its last kernel takes the decision, adds lse - log N to *d_lml when it resamples and records
 * {ess, resampled}) -> CDF and ancestors (kernels that return at once when the decision is no) -> n_moves mh moves.
 * Without moves the gather of a resampling rides on the next step's extend.  interp_key: the caller's interpreter key, advanced exactly as the reference's interpreter would be.
 * obs_rows: host, T x d_obs.  d_trace: 2 T floats {ess_t, resampled_t}; d_accept_counts (nullable): T x n_moves.
 * *final_set = 0 if the final particles are in (d_state, d_mean), 1 if in the *_alt columns. */
int32_t gjx_lgssm_smc_run(gjx_ctx* ctx, const gjx_lgssm_params* model, uint32_t interp_key[2], int32_t T,
                          const float* obs_rows, int64_t n, int32_t n_moves, float* d_state, float* d_mean,
                          float* d_state_alt, float* d_mean_alt, float* d_logw, float* d_stats5, float* d_lml,
                          float* d_trace, int32_t* d_go, int64_t* d_range, int64_t* d_accept_counts,
                          int32_t* final_set, void* stream);

/* ---- outlier-mixture curve fitting (examples/curvefit/core.py:468-552, 934-1014) ----
 * npoint_curve_with_outliers_beta(xs, alpha, beta): outlier_rate ~ beta(alpha, beta);
 * a, b, c ~ N(0, 1); per point is_outlier ~ flip(rate), y ~ Cond(uniform(out_lo, out_hi),
 * normal(a + b x + c x^2, inlier_sigma)).  One chain / particle per element; every chain owns a
 * Threefry key (chain key = split(base_key, .)[offset + i], examples/curvefit/figs.py:939-941). */
typedef struct {
  int32_t n_points;               /* <= GJX_MAX_POINTS */
  float xs[GJX_MAX_POINTS], ys[GJX_MAX_POINTS];
  float alpha, beta;              /* Beta prior of outlier_rate (figs.py:929: 1, 3) */
  float inlier_sigma;             /* 0.1 (core.py:474) */
  float out_lo, out_hi;           /* -2, 2 (core.py:482) */
} gjx_outlier_model;

/* generate() with only ys/y/obs constrained, per chain (figs.py:926-932): writes the chain keys
 * (n x 2 uint32) and the prior draw (rate, a, b, c, is_outlier bit mask: bit j = point j).
 * [3P] tfd.Beta's rejection sampler is not reproduced: the rate is drawn as Ga / (Ga + Gb) with
 * Marsaglia-Tsang gammas on counter-based draws of the site key (same law, different bits). */
int32_t gjx_outlier_init(const gjx_outlier_model* model, const uint32_t base_key[2], uint64_t offset,
                         int64_t n, uint32_t* d_chain_keys, float* d_a, float* d_b, float* d_c,
                         float* d_rate, uint64_t* d_outliers, void* stream);
/* n_iters fused applications of mixed_gibbs_hmc_kernel (core.py:988-1014) per chain: enumerative
 * Gibbs over the indicators (core.py:934-985), then hmc(step_size, n_steps) on
 * (curve.a, curve.b, curve.c, outlier_rate) (mcmc.py:434-578).  The chain keys are advanced
 * (key, sub = split(key) per iteration, figs.py:928), so successive calls continue the chains.
 * d_accept_count (nullable): total number of accepted HMC moves. */
int32_t gjx_outlier_gibbs_hmc(const gjx_outlier_model* model, int32_t n_iters, float step_size,
                              int32_t n_steps, int64_t n, uint32_t* d_chain_keys, float* d_a,
                              float* d_b, float* d_c, float* d_rate, uint64_t* d_outliers,
                              int64_t* d_accept_count, void* stream);
/* ONE application of the same kernel (core.py:988-1014: enumerative_gibbs_outliers :934-985, then hmc
 * mcmc.py:434-578) that also reports the quantities it decided on: d_diag = (10 + n_points) columns of n floats,
 * 0..3 the HMC proposal q_L (mcmc.py:510-535), 4 log alpha before min(0, .) (:559-562), 5 log u (:565), 6 / 7
 * assess() at q_0 / q_L, 8 / 9 the momentum log-densities there, 10 + j the Gumbel-max margin of point j in the
 * Gibbs step (the indicator is 1 iff > 0).  The fused launch above equals n_iters of these bit for bit. */
int32_t gjx_outlier_gibbs_hmc_diag(const gjx_outlier_model* model, float step_size, int32_t n_steps, int64_t n,
                                   uint32_t* d_chain_keys, float* d_a, float* d_b, float* d_c, float* d_rate,
                                   uint64_t* d_outliers, float* d_diag, int64_t* d_accept_count, void* stream);

/* ---- Game-of-Life inversion by single-site Gibbs sweeps (examples/gol/core.py) ----
 * Boards: n_chains x n x n cells, one byte (0 / 1) per cell, row-major, square (the
 * reference's range mask, core.py:141-149, mixes the axes; all its boards are square).
 * d_future: the observed next state, shared by all chains (future_stride = 0) or per chain
 * (future_stride = n * n).  Keys: n_chains x 2 uint32 on the device, one key per chain. */
/* proposal.simulate (core.py:240-249): cell c = uniform(key)[c] < 0.5 */
int32_t gjx_gol_init(const uint32_t* d_keys, int64_t n_chains, int32_t n, uint8_t* d_boards,
                     void* stream);
/* gibbs_move_on_all_cells_at_offset (core.py:108-185): cells (oy + 3a, ox + 3b) in place;
 * draw of cell (a, b) = uniform(pass key, (ny', nx'))[a, b] < 1 - softmax(scores)[0] */
int32_t gjx_gol_gibbs_pass(const uint32_t* d_pass_keys, int64_t n_chains, int32_t n, int32_t oy,
                           int32_t ox, float p_flip, const uint8_t* d_future, int64_t future_stride,
                           uint8_t* d_boards, void* stream);
/* full_gibbs_sweep (core.py:188-204): the nine colour passes; d_pass_keys9 = 9 x n_chains x 2 */
int32_t gjx_gol_sweep(const uint32_t* d_pass_keys9, int64_t n_chains, int32_t n, float p_flip,
                      const uint8_t* d_future, int64_t future_stride, uint8_t* d_boards, void* stream);
/* The same passes on BIT-PACKED boards (one bit per cell, row pitch gjx_gol_words_per_row(n) = ceil(n / 64)
 * 64-bit words, bit c of word w of a row = column 64 w + c, bits beyond n zero; chains contiguous).  Same
 * conditionals, same draws, same new boards as the byte form bit for bit; the 18 rule outcomes of a cell come
 * from two bit-parallel Life steps per tile and the first six terms of each score from a table built with the
 * reference's order of additions.  future_stride_words: 0 (shared target) or n * words_per_row. */
int32_t gjx_gol_words_per_row(int32_t n);
int32_t gjx_gol_pack(const uint8_t* d_boards, int64_t n_chains, int32_t n, uint64_t* d_bits, void* stream);
int32_t gjx_gol_unpack(const uint64_t* d_bits, int64_t n_chains, int32_t n, uint8_t* d_boards, void* stream);
int32_t gjx_gol_gibbs_pass_packed(const uint32_t* d_pass_keys, int64_t n_chains, int32_t n, int32_t oy, int32_t ox,
                                  float p_flip, const uint64_t* d_future_bits, int64_t future_stride_words,
                                  uint64_t* d_board_bits, void* stream);
int32_t gjx_gol_sweep_packed(const uint32_t* d_pass_keys9, int64_t n_chains, int32_t n, float p_flip,
                             const uint64_t* d_future_bits, int64_t future_stride_words, uint64_t* d_board_bits,
                             void* stream);
/* generate_next_state(board, p_flip = 0) (core.py:68-83, 229-230): the rule of Life */
int32_t gjx_gol_step(const uint8_t* d_boards, int64_t n_chains, int32_t n, uint8_t* d_next, void* stream);
/* per chain: bits of step(board) that differ from the target (core.py:328-329) and the
 * predictive posterior score = sum of the step bits' log-densities (core.py:214-221) */
int32_t gjx_gol_summary(const uint8_t* d_boards, const uint8_t* d_future, int64_t future_stride,
                        int64_t n_chains, int32_t n, float p_flip, int64_t* d_n_incorrect,
                        double* d_pred_score, void* stream);
/* host: key schedule of seed(_run_sampler_and_get_summary)(key, sampler) (core.py:312-364) for
 * one chain: the init site key and pass_keys[n_frames][per_frame][9][2] */
void gjx_gol_keys(const uint32_t key[2], int32_t n_frames, int32_t per_frame, uint32_t init_key[2],
                  uint32_t* pass_keys);

#ifdef __cplusplus
}
#endif
#endif /* GJX_H */
