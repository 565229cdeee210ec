/*
 * gjx_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement ("oracle") of the GenJAX vectorized-inference hot path
 * (reference: probcomp/genjax v1.0.13, file:line citations are relative to
 * /root/reference).  Nothing in the product package (genjax_amd/) may import,
 * link or call this file; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py do, and only as the checker / baseline.
 *
 * PARITY STATUS
 *   - The reference is pure Python on top of JAX 0.7.2 / TFP 0.25.0, neither of
 *     which is installed in the build container, so the
 *     reference itself could not be executed to generate vectors.
 *   - Pinned by known-answer tests (tests/test_oracle_*.py): Random123
 *     Threefry-2x32-20 KATs, the published jax.random.split(PRNGKey(0)) value,
 *     analytic answers taken from the reference's own tests (Kalman closed forms
 *     tests/test_smc.py:1539-1596,1721-1778; hierarchical-normal marginal
 *     tests/test_smc.py:52-65), SciPy log-densities (formulas pinned by
 *     tests/test_distributions.py:81-105) and finite differences for gradients.
 *   - "parity unpinned" (no reference test fixes them): the [3P] draw layout of
 *     jax.random under jax_threefry_partitionable, the Seed key schedule, the
 *     cumsum association and the searchsorted procedure.  Each is restated from
 *     the published algorithm and isolated in one function below.
 *
 * All arithmetic is IEEE binary32 ("float"), compiled with -ffp-contract=off so
 * that no FMA contraction happens (JAX default dtype is float32).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* 1. Threefry-2x32 and the jax.random key operations  [3P: jax/_src/prng.py] */
/* ------------------------------------------------------------------------- */

/* jnp.minimum(0.0, x): unlike fminf it PROPAGATES NaN, so a NaN acceptance ratio rejects
 * (log_u < nan is False)   mcmc.py:293-296, 563-566 */
static inline float min0(float x) { return (x > 0.0f) ? 0.0f : x; }

static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

/* Threefry-2x32, 20 rounds (Salmon et al. 2011, Random123), with jax's key
 * schedule constant 0x1BD11BDA.  KATs: tests/test_oracle_prng.py. */
static inline void tf2x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1,
                          uint32_t* o0, uint32_t* o1) {
  const uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
  static const int R0[4] = {13, 15, 26, 6};
  static const int R1[4] = {17, 29, 16, 24};
  uint32_t x0 = c0 + ks[0], x1 = c1 + ks[1];
  for (int g = 0; g < 5; ++g) {
    const int* R = (g & 1) ? R1 : R0;
    for (int i = 0; i < 4; ++i) {
      x0 += x1;
      x1 = rotl32(x1, R[i]);
      x1 ^= x0;
    }
    x0 += ks[(g + 1) % 3];
    x1 += ks[(g + 2) % 3] + (uint32_t)(g + 1);
  }
  *o0 = x0;
  *o1 = x1;
}

ORC_API void orc_threefry2x32(const uint32_t key[2], const uint32_t ctr[2], uint32_t out[2]) {
  tf2x32(key[0], key[1], ctr[0], ctr[1], &out[0], &out[1]);
}

/* jax.random.key(seed) / PRNGKey(seed): (hi32, lo32) of the integer seed. */
ORC_API void orc_key(uint64_t seed, uint32_t key[2]) {
  key[0] = (uint32_t)(seed >> 32);
  key[1] = (uint32_t)(seed & 0xFFFFFFFFu);
}

/* jax.random.split under jax_threefry_partitionable=True (default since jax 0.5):
 * child j = both words of threefry(key, counter=(0, j))   [threefry_split_foldlike] */
ORC_API void orc_split(const uint32_t key[2], uint32_t n, uint32_t* out /* n x 2 */) {
  for (uint32_t j = 0; j < n; ++j) tf2x32(key[0], key[1], 0u, j, &out[2 * j], &out[2 * j + 1]);
}

/* Legacy (pre-0.5 default) split: threefry over iota(2n) split in two halves.
 * Only used to check the block function against the value printed in the JAX
 * documentation: split(PRNGKey(0)) = [[4146024105, 967050713], [2718843009, 1272950319]]. */
ORC_API void orc_split_legacy(const uint32_t key[2], uint32_t n, uint32_t* out /* n x 2 */) {
  uint32_t* flat = (uint32_t*)malloc(sizeof(uint32_t) * 2 * n);
  for (uint32_t j = 0; j < n; ++j) tf2x32(key[0], key[1], j, n + j, &flat[j], &flat[n + j]);
  memcpy(out, flat, sizeof(uint32_t) * 2 * n);
  free(flat);
}

/* jax.random.fold_in(key, data): threefry(key, counter=(0, data)). */
ORC_API void orc_fold_in(const uint32_t key[2], uint32_t data, uint32_t out[2]) {
  tf2x32(key[0], key[1], 0u, data, &out[0], &out[1]);
}

/* random_bits, 32-bit, partitionable layout: element i uses counter (hi32(i), lo32(i))
 * and returns word0 ^ word1.  `offset` lets a shard start at a global index. */
static inline uint32_t bits_at(const uint32_t key[2], uint64_t i) {
  uint32_t a, b;
  tf2x32(key[0], key[1], (uint32_t)(i >> 32), (uint32_t)(i & 0xFFFFFFFFu), &a, &b);
  return a ^ b;
}

ORC_API void orc_random_bits(const uint32_t key[2], uint64_t offset, uint64_t n, uint32_t* out) {
  for (uint64_t i = 0; i < n; ++i) out[i] = bits_at(key, offset + i);
}

static inline float u01_from_bits(uint32_t bits) {
  uint32_t fb = (bits >> 9) | 0x3F800000u;
  float f;
  memcpy(&f, &fb, 4);
  return f - 1.0f;
}

/* jax.random.uniform(key, shape, f32, minval, maxval):
 *   max(minval, u01 * (maxval - minval) + minval) */
static inline float uniform_at(const uint32_t key[2], uint64_t i, float minval, float maxval) {
  float f = u01_from_bits(bits_at(key, i));
  float v = f * (maxval - minval) + minval;
  return v > minval ? v : minval;
}

ORC_API void orc_uniform(const uint32_t key[2], uint64_t offset, uint64_t n, float minval,
                         float maxval, float* out) {
  for (uint64_t i = 0; i < n; ++i) out[i] = uniform_at(key, offset + i, minval, maxval);
}

/* XLA's f32 erf_inv (Giles 2010 single-precision polynomial on w = -log1p(-x*x)). */
ORC_API float orc_erfinv(float x) {
  static const float A[9] = {2.81022636e-08f, 3.43273939e-07f, -3.5233877e-06f,
                             -4.39150654e-06f, 0.00021858087f, -0.00125372503f,
                             -0.00417768164f, 0.246640727f, 1.50140941f};
  static const float B[9] = {-0.000200214257f, 0.000100950558f, 0.00134934322f,
                             -0.00367342844f, 0.00573950773f, -0.0076224613f,
                             0.00943887047f, 1.00167406f, 2.83297682f};
  if (fabsf(x) == 1.0f) return x * INFINITY;
  float w = -log1pf(-x * x);
  const float* c;
  if (w < 5.0f) {
    w = w - 2.5f;
    c = A;
  } else {
    w = sqrtf(w) - 3.0f;
    c = B;
  }
  float p = c[0];
  for (int i = 1; i < 9; ++i) p = c[i] + p * w;
  return p * x;
}

/* jax.random.normal: sqrt(2) * erf_inv(uniform(key, minval=nextafter(-1,0), maxval=1)). */
static inline float normal_at(const uint32_t key[2], uint64_t i) {
  const float lo = nextafterf(-1.0f, 0.0f);
  float u = uniform_at(key, i, lo, 1.0f);
  return 1.41421356237309504880f * orc_erfinv(u);
}

ORC_API void orc_normal(const uint32_t key[2], uint64_t offset, uint64_t n, float* out) {
  for (uint64_t i = 0; i < n; ++i) out[i] = normal_at(key, offset + i);
}

/* ------------------------------------------------------------------------- */
/* 2. Log densities  [3P: TFP 0.25 log_prob; call site core.py:1752-1754]      */
/* ------------------------------------------------------------------------- */

#define HALF_LOG_2PI 0.9189385332046727f

/* tfd.Normal._log_prob: -0.5*(x/s - m/s)^2 - (0.5*log(2pi) + log s)   distributions.py:159-162 */
static inline float normal_logpdf(float x, float mu, float sigma) {
  float z = x / sigma - mu / sigma;
  return -0.5f * (z * z) - (HALF_LOG_2PI + logf(sigma));
}

ORC_API void orc_normal_logpdf(const float* x, const float* mu, const float* sigma, int64_t n,
                               float* out) {
  for (int64_t i = 0; i < n; ++i) out[i] = normal_logpdf(x[i], mu[i], sigma[i]);
}

/* tfd.Uniform._log_prob: -log(high-low) inside [low, high], -inf outside   distributions.py:194-197 */
ORC_API void orc_uniform_logpdf(const float* x, const float* lo, const float* hi, int64_t n,
                                float* out) {
  for (int64_t i = 0; i < n; ++i)
    out[i] = (x[i] < lo[i] || x[i] > hi[i]) ? -INFINITY : -logf(hi[i] - lo[i]);
}

/* tfd.Bernoulli(probs=p)._log_prob: x*log(p) + (1-x)*log1p(-p), with 0*(-inf) := 0
 * (multiply_no_nan)   distributions.py:22-25,51-54 */
ORC_API void orc_bernoulli_logpdf(const float* x, const float* p, int64_t n, float* out) {
  for (int64_t i = 0; i < n; ++i) {
    float a = (x[i] == 0.0f) ? 0.0f : x[i] * logf(p[i]);
    float b = (1.0f - x[i] == 0.0f) ? 0.0f : (1.0f - x[i]) * log1pf(-p[i]);
    out[i] = a + b;
  }
}

/* tfd.Categorical(logits)._log_prob: log_softmax(logits)[x]   distributions.py:92-95
 * logits: n x k row-major, x: n int32 */
ORC_API void orc_categorical_logpdf(const int32_t* x, const float* logits, int64_t n, int32_t k,
                                    float* out) {
  for (int64_t i = 0; i < n; ++i) {
    const float* l = logits + i * k;
    float m = l[0];
    for (int j = 1; j < k; ++j) m = l[j] > m ? l[j] : m;
    float s = 0.0f;
    for (int j = 0; j < k; ++j) s += expf(l[j] - m);
    out[i] = (l[x[i]] - m) - logf(s);
  }
}

/* tfd.Beta._log_prob: (a-1)log x + (b-1)log1p(-x) - lbeta(a,b)   distributions.py:62-65 */
ORC_API void orc_beta_logpdf(const float* x, const float* a, const float* b, int64_t n, float* out) {
  for (int64_t i = 0; i < n; ++i) {
    float lbeta = lgammaf(a[i]) + lgammaf(b[i]) - lgammaf(a[i] + b[i]);
    float t1 = (a[i] - 1.0f == 0.0f) ? 0.0f : (a[i] - 1.0f) * logf(x[i]);
    float t2 = (b[i] - 1.0f == 0.0f) ? 0.0f : (b[i] - 1.0f) * log1pf(-x[i]);
    out[i] = (x[i] < 0.0f || x[i] > 1.0f) ? -INFINITY : t1 + t2 - lbeta;
  }
}

/* tfd.MultivariateNormalFullCovariance: Cholesky L of cov, then
 * -0.5*|L^-1 (x-mu)|^2 - sum log L_ii - d/2 log(2pi)   distributions.py:298-301
 * x: n x d, mu: n x d, cov: n x d x d (d <= 8) */
ORC_API int orc_mvnormal_logpdf(const float* x, const float* mu, const float* cov, int64_t n,
                                int32_t d, float* out) {
  if (d > 8) return -1;
  for (int64_t i = 0; i < n; ++i) {
    float L[64];
    const float* C = cov + i * d * d;
    for (int r = 0; r < d; ++r)
      for (int c = 0; c <= r; ++c) {
        float s = C[r * d + c];
        for (int k = 0; k < c; ++k) s -= L[r * d + k] * L[c * d + k];
        L[r * d + c] = (r == c) ? sqrtf(s) : s / L[c * d + c];
      }
    float z[8], q = 0.0f, logdet = 0.0f;
    for (int r = 0; r < d; ++r) {
      float s = x[i * d + r] - mu[i * d + r];
      for (int k = 0; k < r; ++k) s -= L[r * d + k] * z[k];
      z[r] = s / L[r * d + r];
      q += z[r] * z[r];
      logdet += logf(L[r * d + r]);
    }
    out[i] = -0.5f * q - logdet - (float)d * HALF_LOG_2PI;
  }
  return 0;
}

/* ------------------------------------------------------------------------- */
/* 3. Weight normalisation  (smc.py:24-76, 193-204; [3P] jax logsumexp)         */
/* ------------------------------------------------------------------------- */

/* jax.scipy.special.logsumexp: m = max; m = isfinite(m) ? m : 0; log(sum(exp(a-m))) + m.
 * XLA's reduction order is unspecified; the restatement accumulates in double
 * and rounds once, i.e. it returns the best f32 answer. */
ORC_API float orc_logsumexp(const float* a, int64_t n) {
  float m = -INFINITY;
  for (int64_t i = 0; i < n; ++i) m = a[i] > m ? a[i] : m;
  if (!isfinite(m)) m = 0.0f;
  double s = 0.0;
  for (int64_t i = 0; i < n; ++i) s += (double)expf(a[i] - m);
  return logf((float)s) + m;
}

/* effective_sample_size (smc.py:74-76): 1 / sum(exp(logw - lse)^2). */
ORC_API float orc_ess(const float* logw, int64_t n) {
  float lse = orc_logsumexp(logw, n);
  double s = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    float w = expf(logw[i] - lse);
    s += (double)(w * w);
  }
  return 1.0f / (float)s;
}

/* ------------------------------------------------------------------------- */
/* 4. Resampling  (smc.py:79-175; [3P] jnp.cumsum / jnp.searchsorted)           */
/* ------------------------------------------------------------------------- */

/* jnp.cumsum on CPU/GPU lowers to lax.associative_scan(add):
 *   reduced = e[0:-1:2] + e[1::2]; odd = scan(reduced);
 *   even[i+1] = odd[i] + e[2i+2]; even[0] = e[0]; interleave(even, odd). */
static void assoc_scan_add(const float* e, float* out, int64_t n) {
  if (n <= 0) return;
  if (n == 1) {
    out[0] = e[0];
    return;
  }
  int64_t nr = n / 2;
  float* r = (float*)malloc(sizeof(float) * nr);
  float* odd = (float*)malloc(sizeof(float) * nr);
  for (int64_t i = 0; i < nr; ++i) r[i] = e[2 * i] + e[2 * i + 1];
  assoc_scan_add(r, odd, nr);
  out[0] = e[0];
  for (int64_t i = 0; 2 * i + 2 < n; ++i) out[2 * i + 2] = odd[i] + e[2 * i + 2];
  for (int64_t i = 0; i < nr; ++i) out[2 * i + 1] = odd[i];
  free(r);
  free(odd);
}

ORC_API void orc_cumsum(const float* w, int64_t n, float* out) { assoc_scan_add(w, out, n); }

/* jnp.searchsorted(a, v, side='left', method='scan'): fixed ceil(log2(n+1))
 * bisection steps on (low, high) = (0, n); go_left iff v <= a[mid]; returns high.
 * For an array that is not monotone (f32 cumsum with the association above is
 * not guaranteed to be) the answer depends on this exact procedure. */
ORC_API void orc_searchsorted_left(const float* a, int64_t n, const float* v, int64_t m,
                                   int32_t* out) {
  int levels = 0;
  while (((int64_t)1 << levels) < n + 1) ++levels;
  for (int64_t j = 0; j < m; ++j) {
    uint32_t low = 0, high = (uint32_t)n;
    for (int l = 0; l < levels; ++l) {
      uint32_t mid = (low + high) / 2u;
      int go_left = v[j] <= a[mid];
      if (go_left)
        high = mid;
      else
        low = mid;
    }
    out[j] = (int32_t)high;
  }
}

/* systematic_resample given normalised weights and the offset u (smc.py:132-135):
 * positions = (arange(n) + u) / n; idx = searchsorted(cumsum(w), positions);
 * the gather leaf[idx] clamps idx to n-1 ([3P] jnp indexing). */
ORC_API void orc_systematic_indices(const float* weights, int64_t n, float u, int32_t* idx,
                                    float* cdf_out /* nullable */) {
  float* cdf = (float*)malloc(sizeof(float) * n);
  float* pos = (float*)malloc(sizeof(float) * n);
  assoc_scan_add(weights, cdf, n);
  for (int64_t j = 0; j < n; ++j) pos[j] = ((float)(int32_t)j + u) / (float)n;
  orc_searchsorted_left(cdf, n, pos, n, idx);
  for (int64_t j = 0; j < n; ++j)
    if (idx[j] > (int32_t)(n - 1)) idx[j] = (int32_t)(n - 1);
  if (cdf_out) memcpy(cdf_out, cdf, sizeof(float) * n);
  free(cdf);
  free(pos);
}

/* weights = exp(logw - logsumexp(logw))   (smc.py:127-128) */
ORC_API void orc_normalized_weights(const float* logw, int64_t n, float* w) {
  float lse = orc_logsumexp(logw, n);
  for (int64_t i = 0; i < n; ++i) w[i] = expf(logw[i] - lse);
}

/* categorical.sample(logw, sample_shape=(m,)) (smc.py:160-162) = TFP Categorical
 * -> [3P] gumbel(key, shape=(1, n, m)) + logits[..., None], argmax over axis -2.
 * Noise element (category i, sample j) has flat counter i*m + j.  parity unpinned. */
ORC_API void orc_categorical_indices(const uint32_t key[2], const float* logw, int64_t n,
                                     int64_t m, int32_t* idx) {
  const float tiny = 1.17549435e-38f;
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < m; ++j) {
    float best = -INFINITY;
    int32_t arg = 0;
    for (int64_t i = 0; i < n; ++i) {
      float u = uniform_at(key, (uint64_t)(i * m + j), tiny, 1.0f);
      float g = -logf(-logf(u));
      float v = logw[i] + g;
      if (i == 0 || v > best) { /* jnp.argmax: first maximal index */
        best = v;
        arg = (int32_t)i;
      }
    }
    idx[j] = arg;
  }
}

/* ------------------------------------------------------------------------- */
/* 5. Localization model  (examples/localization/core.py)                      */
/* ------------------------------------------------------------------------- */

typedef struct {
  float width, height;
  int32_t W;
  const float *x1, *y1, *x2, *y2;
} orc_world;

#define ORC_MAX_RAYS 64
#define TWO_PI_F 6.283185307179586f

static inline float clipf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* compute_ray_wall_intersection (core.py:81-128) */
static float ray_walls(float px, float py, float dx, float dy, const orc_world* w) {
  float best = INFINITY;
  for (int k = 0; k < w->W; ++k) {
    float wdx = w->x2[k] - w->x1[k];
    float wdy = w->y2[k] - w->y1[k];
    float twx = w->x1[k] - px;
    float twy = w->y1[k] - py;
    float denom = dx * wdy - dy * wdx;
    if (fabsf(denom) < 1e-10f) denom = 1e-10f;
    float t = (twx * wdy - twy * wdx) / denom;
    float s = (twx * dy - twy * dx) / denom;
    int valid = (t >= 0.0f) && (s >= 0.0f) && (s <= 1.0f);
    float dist = valid ? t : INFINITY;
    best = dist < best ? dist : best;
  }
  return best;
}

/* distance_to_wall_lidar (core.py:252-308), max_range = 10 */
static void lidar(float px, float py, float theta, const orc_world* w, int R, float* d) {
  for (int r = 0; r < R; ++r) {
    /* jnp.linspace(0, 2pi, R, endpoint=False)[r] = 0*(1-r/R) + 2pi*(r/R)  [3P] */
    float step = (float)r / (float)R;
    float ang = 0.0f * (1.0f - step) + TWO_PI_F * step;
    float wa = theta + ang;
    float dx = cosf(wa), dy = sinf(wa);
    float t_right = dx > 0.0f ? (w->width - px) / dx : INFINITY;
    float t_left = dx < 0.0f ? -px / dx : INFINITY;
    float t_top = dy > 0.0f ? (w->height - py) / dy : INFINITY;
    float t_bottom = dy < 0.0f ? -py / dy : INFINITY;
    float tb = fminf(fminf(t_right, t_left), fminf(t_top, t_bottom));
    float tw = ray_walls(px, py, dx, dy, w);
    float tm = fminf(tb, tw);
    d[r] = fminf(tm, 10.0f);
  }
}

ORC_API void orc_lidar(const float* px, const float* py, const float* theta, int64_t n,
                       float width, float height, int32_t W, const float* x1, const float* y1,
                       const float* x2, const float* y2, int32_t R, float* d /* n x R */) {
  orc_world w = {width, height, W, x1, y1, x2, y2};
  for (int64_t i = 0; i < n; ++i) lidar(px[i], py[i], theta[i], &w, R, d + i * R);
}

/* Prior parameters of localization_model (core.py:328-353). */
static inline void loc_sigmas(int is_initial, float s[3]) {
  s[0] = is_initial ? 0.5f : 0.25f;
  s[1] = is_initial ? 0.5f : 0.25f;
  s[2] = is_initial ? 0.3f : 0.1f;
}

/* Per-ray sensor log-densities of one pose: choices are clipped for the sensor
 * (core.py:361-368), obs_r ~ N(d_r, 0.3) (core.py:386-395). */
static void loc_ray_logpdfs(float x, float y, float th, const float* obs, const orc_world* w,
                            int R, float* lp) {
  float d[ORC_MAX_RAYS];
  float px = clipf(x, 0.1f, w->width - 0.1f);
  float py = clipf(y, 0.1f, w->height - 0.1f);
  lidar(px, py, th, w, R, d);
  for (int r = 0; r < R; ++r) lp[r] = normal_logpdf(obs[r], d[r], 0.3f);
}

/* per-ray sensor log-densities of n poses (the Vmap trace's per-site scores, negated) */
ORC_API void orc_loc_ray_logpdfs_n(const float* x, const float* y, const float* th, int64_t n,
                                   const float* obs, int32_t R, float width, float height, int32_t W,
                                   const float* x1, const float* y1, const float* x2,
                                   const float* y2, float* lp) {
  orc_world w = {width, height, W, x1, y1, x2, y2};
  for (int64_t i = 0; i < n; ++i) loc_ray_logpdfs(x[i], y[i], th[i], obs, &w, R, lp + i * R);
}

/* One vectorised generate() of localization_model with only "measurements"
 * constrained (smc.py:306 for init, smc.py:488-491 for extend; Generate handler
 * core.py:1944-1975; Vmap.generate sums the ray weights core.py:1416-1433).
 *
 *   is_initial: means are (1.5, 1.5, 0); otherwise the previous retval pose =
 *   clip of the previous choices (core.py:346-348,361-364,383; smc.py:734-742).
 * In/out columns: x,y,th hold the previous choices on entry and the new choices
 * on exit; mx,my,mth receive the means used; lp (n x R) the per-ray log-densities. */
ORC_API void orc_loc_extend(const uint32_t kx[2], const uint32_t ky[2], const uint32_t kth[2],
                            int32_t is_initial, int64_t n, uint64_t offset, const float* obs,
                            int32_t R, float width, float height, int32_t W, const float* x1,
                            const float* y1, const float* x2, const float* y2, float* x, float* y,
                            float* th, float* mx, float* my, float* mth, float* logw, float* lp) {
  orc_world w = {width, height, W, x1, y1, x2, y2};
  float s[3];
  loc_sigmas(is_initial, s);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    float m0, m1, m2;
    if (is_initial) {
      m0 = 1.5f;
      m1 = 1.5f;
      m2 = 0.0f;
    } else {
      m0 = clipf(x[i], 0.1f, width - 0.1f);
      m1 = clipf(y[i], 0.1f, height - 0.1f);
      m2 = th[i];
    }
    /* tfd.Normal.sample: normal(key, shape) * scale + loc  [3P] */
    float nx = normal_at(kx, offset + (uint64_t)i) * s[0] + m0;
    float ny = normal_at(ky, offset + (uint64_t)i) * s[1] + m1;
    float nt = normal_at(kth, offset + (uint64_t)i) * s[2] + m2;
    float* lpi = lp + i * R;
    loc_ray_logpdfs(nx, ny, nt, obs, &w, R, lpi);
    float wsum = 0.0f;
    for (int r = 0; r < R; ++r) wsum += lpi[r];
    float winc = ((0.0f + 0.0f) + 0.0f) + 0.0f; /* three unconstrained sites weigh 0 */
    winc += wsum;
    x[i] = nx;
    y[i] = ny;
    th[i] = nt;
    mx[i] = m0;
    my[i] = m1;
    mth[i] = m2;
    logw[i] = logw[i] + winc;
  }
}

/* jnp.linspace(lo, hi, g)[i]  [3P]: lo * (1 - i/div) + hi * (i/div) for i < div, hi at the end */
static inline float linspace_at(float lo, float hi, int g, int i) {
  if (g == 1) return lo;
  if (i == g - 1) return hi;
  const float step = (float)i / (float)(g - 1);
  return lo * (1.0f - step) + hi * step;
}

/* extend() with the locally-optimal grid proposal (examples/localization/core.py:730-809,
 * smc.py:493-509): grid of g^3 poses around the particle's previous choices, model assess() at
 * every grid pose, first argmax, Gaussian noise around it; weight = target weight + proposal
 * score.  x, y, th hold the previous CHOICES on entry (old_choices; the model args are their
 * clipped values, smc.py:734, core.py:361-364). */
ORC_API void orc_loc_extend_locopt(const uint32_t kx[2], const uint32_t ky[2], const uint32_t kth[2],
                                   int32_t grid, float radius, float noise, int64_t n,
                                   uint64_t offset, const float* obs, int32_t R, float width,
                                   float height, int32_t W, const float* x1, const float* y1,
                                   const float* x2, const float* y2, float* x, float* y, float* th,
                                   float* mx, float* my, float* mth, float* logw, float* lp) {
  orc_world w = {width, height, W, x1, y1, x2, y2};
  float s[3];
  loc_sigmas(0, s);
  const float theta_radius = (float)(3.14159265358979323846 / 6.0);
  const float noise_th = (float)((double)noise * 0.5);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    const float cx = x[i], cy = y[i], ct = th[i];
    const float m0 = clipf(cx, 0.1f, width - 0.1f), m1 = clipf(cy, 0.1f, height - 0.1f), m2 = ct;
    const float xlo = fmaxf(0.5f, cx - radius), xhi = fminf(width - 0.5f, cx + radius);
    const float ylo = fmaxf(0.5f, cy - radius), yhi = fminf(height - 0.5f, cy + radius);
    const float tlo = ct - theta_radius, thi = ct + theta_radius;
    float best = -INFINITY, sx = 0.0f, sy = 0.0f, st = 0.0f;
    int have = 0;
    float rays[ORC_MAX_RAYS];
    for (int ix = 0; ix < grid; ++ix)
      for (int iy = 0; iy < grid; ++iy)
        for (int it = 0; it < grid; ++it) {
          const float gx = linspace_at(xlo, xhi, grid, ix), gy = linspace_at(ylo, yhi, grid, iy);
          const float gt = linspace_at(tlo, thi, grid, it);
          float sc = (normal_logpdf(gx, m0, s[0]) + normal_logpdf(gy, m1, s[1])) +
                     normal_logpdf(gt, m2, s[2]);
          loc_ray_logpdfs(gx, gy, gt, obs, &w, R, rays);
          float ms = 0.0f;
          for (int r = 0; r < R; ++r) ms += rays[r];
          sc += ms;
          if (!have || sc > best) { /* jnp.argmax: first maximum */
            best = sc;
            sx = gx;
            sy = gy;
            st = gt;
            have = 1;
          }
        }
    const float nx = normal_at(kx, offset + (uint64_t)i) * noise + sx;
    const float ny = normal_at(ky, offset + (uint64_t)i) * noise + sy;
    const float nt = normal_at(kth, offset + (uint64_t)i) * noise_th + st;
    const float pscore = -((normal_logpdf(nx, sx, noise) + normal_logpdf(ny, sy, noise)) +
                           normal_logpdf(nt, st, noise_th));
    float* lpi = lp + i * R;
    loc_ray_logpdfs(nx, ny, nt, obs, &w, R, lpi);
    float ms = 0.0f;
    for (int r = 0; r < R; ++r) ms += lpi[r];
    const float wt = ((normal_logpdf(nx, m0, s[0]) + normal_logpdf(ny, m1, s[1])) +
                      normal_logpdf(nt, m2, s[2])) + ms;
    x[i] = nx;
    y[i] = ny;
    th[i] = nt;
    mx[i] = m0;
    my[i] = m1;
    mth[i] = m2;
    logw[i] = (logw[i] + wt) + pscore;
  }
}

/* The g^3 grid scores of ONE particle of orc_loc_extend_locopt (same loops, same order: index = (ix g + iy) g + it)
 * and the sum of the magnitudes of the terms of each score, for the argmax-tie listing of the parity test. */
ORC_API void orc_loc_locopt_grid_scores(int32_t grid, float radius, float cx, float cy, float ct, const float* obs,
                                        int32_t R, float width, float height, int32_t W, const float* x1,
                                        const float* y1, const float* x2, const float* y2, float* scores,
                                        float* terms, float* poses /* g^3 x 3 */) {
  orc_world w = {width, height, W, x1, y1, x2, y2};
  float s[3];
  loc_sigmas(0, s);
  const float theta_radius = (float)(3.14159265358979323846 / 6.0);
  const float m0 = clipf(cx, 0.1f, width - 0.1f), m1 = clipf(cy, 0.1f, height - 0.1f), m2 = ct;
  const float xlo = fmaxf(0.5f, cx - radius), xhi = fminf(width - 0.5f, cx + radius);
  const float ylo = fmaxf(0.5f, cy - radius), yhi = fminf(height - 0.5f, cy + radius);
  const float tlo = ct - theta_radius, thi = ct + theta_radius;
  float rays[ORC_MAX_RAYS];
  int64_t k = 0;
  for (int ix = 0; ix < grid; ++ix)
    for (int iy = 0; iy < grid; ++iy)
      for (int it = 0; it < grid; ++it, ++k) {
        const float gx = linspace_at(xlo, xhi, grid, ix), gy = linspace_at(ylo, yhi, grid, iy);
        const float gt = linspace_at(tlo, thi, grid, it);
        const float a0 = normal_logpdf(gx, m0, s[0]), a1 = normal_logpdf(gy, m1, s[1]), a2 = normal_logpdf(gt, m2, s[2]);
        float sc = (a0 + a1) + a2;
        loc_ray_logpdfs(gx, gy, gt, obs, &w, R, rays);
        float ms = 0.0f, mag = fabsf(a0) + fabsf(a1) + fabsf(a2);
        for (int r = 0; r < R; ++r) {
          ms += rays[r];
          mag += fabsf(rays[r]);
        }
        sc += ms;
        scores[k] = sc;
        terms[k] = mag;
        poses[3 * k] = gx;
        poses[3 * k + 1] = gy;
        poses[3 * k + 2] = gt;
      }
}

/* mh(trace, sel("x")|sel("y")|sel("theta"))  (mcmc.py:267-309):
 * regenerate -> fresh prior draws at the stored args (core.py:1638-1648), weight =
 * sum_r [logpdf_r(new) + old_score_r] (core.py:1649-1657, 1474-1492), accept iff
 * log(u) < min(0, w).  Returns the number of accepted particles. */
/* diag (nullable): n x 6 floats per particle {x', y', theta', weight (before min0), log(u),
 * sum_r (|logpdf_r(new)| + |old_score_r|)}: the proposal of the move and the quantities of its
 * accept test, for the teacher-forced per-move parity tests (test infrastructure only). */
static int64_t loc_mh_impl(const uint32_t kx[2], const uint32_t ky[2], const uint32_t kth[2],
                           const uint32_t ku[2], int32_t is_initial, int64_t n, uint64_t offset,
                           const float* obs, int32_t R, float width, float height, int32_t W,
                           const float* x1, const float* y1, const float* x2, const float* y2,
                           float* x, float* y, float* th, const float* mx, const float* my,
                           const float* mth, float* lp, uint8_t* accept_out /* nullable */,
                           float* diag /* nullable */) {
  orc_world w = {width, height, W, x1, y1, x2, y2};
  float s[3];
  loc_sigmas(is_initial, s);
  int64_t acc = 0;
#pragma omp parallel for schedule(static) reduction(+ : acc)
  for (int64_t i = 0; i < n; ++i) {
    float nx = normal_at(kx, offset + (uint64_t)i) * s[0] + mx[i];
    float ny = normal_at(ky, offset + (uint64_t)i) * s[1] + my[i];
    float nt = normal_at(kth, offset + (uint64_t)i) * s[2] + mth[i];
    float u = uniform_at(ku, offset + (uint64_t)i, 0.0f, 1.0f);
    u = 0.0f + (1.0f - 0.0f) * u; /* tfd.Uniform: low + (high-low)*samples */
    float nlp[ORC_MAX_RAYS];
    float* lpi = lp + i * R;
    loc_ray_logpdfs(nx, ny, nt, obs, &w, R, nlp);
    float wsum = 0.0f;
    for (int r = 0; r < R; ++r) wsum += nlp[r] + (-lpi[r]);
    float wt = ((0.0f + 0.0f) + 0.0f) + 0.0f;
    wt += wsum;
    float log_alpha = min0(wt);
    int accept = logf(u) < log_alpha;
    if (diag) {
      float scale = 0.0f;
      for (int r = 0; r < R; ++r) scale += fabsf(nlp[r]) + fabsf(lpi[r]);
      float* d = diag + 6 * i;
      d[0] = nx;
      d[1] = ny;
      d[2] = nt;
      d[3] = wt;
      d[4] = logf(u);
      d[5] = scale;
    }
    if (accept) {
      x[i] = nx;
      y[i] = ny;
      th[i] = nt;
      for (int r = 0; r < R; ++r) lpi[r] = nlp[r];
      acc += 1;
    }
    if (accept_out) accept_out[i] = (uint8_t)accept;
  }
  return acc;
}

ORC_API int64_t orc_loc_mh(const uint32_t kx[2], const uint32_t ky[2], const uint32_t kth[2],
                           const uint32_t ku[2], int32_t is_initial, int64_t n, uint64_t offset,
                           const float* obs, int32_t R, float width, float height, int32_t W,
                           const float* x1, const float* y1, const float* x2, const float* y2,
                           float* x, float* y, float* th, const float* mx, const float* my,
                           const float* mth, float* lp, uint8_t* accept_out /* nullable */) {
  return loc_mh_impl(kx, ky, kth, ku, is_initial, n, offset, obs, R, width, height, W, x1, y1, x2, y2, x, y,
                     th, mx, my, mth, lp, accept_out, NULL);
}

ORC_API int64_t orc_loc_mh_diag(const uint32_t kx[2], const uint32_t ky[2], const uint32_t kth[2],
                                const uint32_t ku[2], int32_t is_initial, int64_t n, uint64_t offset,
                                const float* obs, int32_t R, float width, float height, int32_t W,
                                const float* x1, const float* y1, const float* x2, const float* y2,
                                float* x, float* y, float* th, const float* mx, const float* my,
                                const float* mth, float* lp, uint8_t* accept_out, float* diag) {
  return loc_mh_impl(kx, ky, kth, ku, is_initial, n, offset, obs, R, width, height, W, x1, y1, x2, y2, x, y,
                     th, mx, my, mth, lp, accept_out, diag);
}

/* ---- forward-mode duals for the HMC gradient (jax.grad of assess, mcmc.py:508,523) ---- */
typedef struct {
  float v, d[3];
} dual;
static inline dual dconst(float v) {
  dual r = {v, {0.0f, 0.0f, 0.0f}};
  return r;
}
static inline dual dvar(float v, int k) {
  dual r = dconst(v);
  r.d[k] = 1.0f;
  return r;
}
static inline dual dadd(dual a, dual b) {
  dual r = {a.v + b.v, {a.d[0] + b.d[0], a.d[1] + b.d[1], a.d[2] + b.d[2]}};
  return r;
}
static inline dual dsub(dual a, dual b) {
  dual r = {a.v - b.v, {a.d[0] - b.d[0], a.d[1] - b.d[1], a.d[2] - b.d[2]}};
  return r;
}
static inline dual dmul(dual a, dual b) {
  dual r;
  r.v = a.v * b.v;
  for (int k = 0; k < 3; ++k) r.d[k] = a.d[k] * b.v + a.v * b.d[k];
  return r;
}
static inline dual ddiv(dual a, dual b) {
  dual r;
  r.v = a.v / b.v;
  for (int k = 0; k < 3; ++k) r.d[k] = (a.d[k] - r.v * b.d[k]) / b.v;
  return r;
}
static inline dual dscale(dual a, float c) {
  dual r = {a.v * c, {a.d[0] * c, a.d[1] * c, a.d[2] * c}};
  return r;
}
static inline dual dmin(dual a, dual b) { return (b.v < a.v) ? b : a; }

/* log density of (x, y, theta) under localization_model with all sites
 * constrained (Assess handler core.py:1978-2003), value and gradient. */
static float loc_logp_grad_abs(float x, float y, float th, float m0, float m1, float m2,
                               const float s[3], const float* obs, const orc_world* w, int R,
                               float g[3], float* gabs /* nullable: [3] sum of |terms| of g, [1] of logp */) {
  dual X = dvar(x, 0), Y = dvar(y, 1), T = dvar(th, 2);
  /* priors */
  float lpx = normal_logpdf(x, m0, s[0]);
  float lpy = normal_logpdf(y, m1, s[1]);
  float lpt = normal_logpdf(th, m2, s[2]);
  float gx = -(x / s[0] - m0 / s[0]) / s[0];
  float gy = -(y / s[1] - m1 / s[1]) / s[1];
  float gt = -(th / s[2] - m2 / s[2]) / s[2];
  /* jnp.clip: identity gradient inside, zero outside (core.py:361-362) */
  float xlo = 0.1f, xhi = w->width - 0.1f, ylo = 0.1f, yhi = w->height - 0.1f;
  dual PX = (x < xlo) ? dconst(xlo) : ((x > xhi) ? dconst(xhi) : X);
  dual PY = (y < ylo) ? dconst(ylo) : ((y > yhi) ? dconst(yhi) : Y);
  float meas = 0.0f;
  float gm[3] = {0.0f, 0.0f, 0.0f};
  if (gabs) gabs[0] = gabs[1] = gabs[2] = gabs[3] = 0.0f;
  for (int r = 0; r < R; ++r) {
    float step = (float)r / (float)R;
    float ang = 0.0f * (1.0f - step) + TWO_PI_F * step;
    dual WA = dadd(T, dconst(ang));
    float c = cosf(WA.v), sn = sinf(WA.v);
    dual DX = {c, {-sn * WA.d[0], -sn * WA.d[1], -sn * WA.d[2]}};
    dual DY = {sn, {c * WA.d[0], c * WA.d[1], c * WA.d[2]}};
    dual INF = dconst(INFINITY);
    dual t_right = DX.v > 0.0f ? ddiv(dsub(dconst(w->width), PX), DX) : INF;
    dual t_left = DX.v < 0.0f ? ddiv(dscale(PX, -1.0f), DX) : INF;
    dual t_top = DY.v > 0.0f ? ddiv(dsub(dconst(w->height), PY), DY) : INF;
    dual t_bottom = DY.v < 0.0f ? ddiv(dscale(PY, -1.0f), DY) : INF;
    dual tb = dmin(dmin(t_right, t_left), dmin(t_top, t_bottom));
    dual tw = INF;
    for (int k = 0; k < w->W; ++k) {
      float wdx = w->x2[k] - w->x1[k];
      float wdy = w->y2[k] - w->y1[k];
      dual TWX = dsub(dconst(w->x1[k]), PX);
      dual TWY = dsub(dconst(w->y1[k]), PY);
      dual denom = dsub(dscale(DX, wdy), dscale(DY, wdx));
      if (fabsf(denom.v) < 1e-10f) denom = dconst(1e-10f);
      dual tt = ddiv(dsub(dscale(TWX, wdy), dscale(TWY, wdx)), denom);
      dual ss = ddiv(dsub(dmul(TWX, DY), dmul(TWY, DX)), denom);
      int valid = (tt.v >= 0.0f) && (ss.v >= 0.0f) && (ss.v <= 1.0f);
      if (valid) tw = dmin(tw, tt);
    }
    dual tm = dmin(tb, tw);
    dual dd = (tm.v > 10.0f) ? dconst(10.0f) : tm;
    float z = obs[r] / 0.3f - dd.v / 0.3f;
    meas += -0.5f * (z * z) - (HALF_LOG_2PI + logf(0.3f));
    for (int k = 0; k < 3; ++k) gm[k] += (z / 0.3f) * dd.d[k];
    if (gabs) {
      /* z = obs/sigma - d/sigma cancels: the term's rounding error scales with the operands of that
       * subtraction, not with z */
      for (int k = 0; k < 3; ++k) gabs[k] += ((fabsf(obs[r]) + fabsf(dd.v)) / 0.3f / 0.3f) * fabsf(dd.d[k]);
      gabs[3] += fabsf(-0.5f * (z * z) - (HALF_LOG_2PI + logf(0.3f)));
    }
  }
  float logp = 0.0f;
  logp += lpx;
  logp += lpy;
  logp += lpt;
  logp += meas;
  g[0] = gx + gm[0];
  g[1] = gy + gm[1];
  g[2] = gt + gm[2];
  if (gabs) {
    gabs[0] += fabsf(gx);
    gabs[1] += fabsf(gy);
    gabs[2] += fabsf(gt);
    gabs[3] += fabsf(lpx) + fabsf(lpy) + fabsf(lpt);
  }
  return logp;
}

static float loc_logp_grad(float x, float y, float th, float m0, float m1, float m2,
                           const float s[3], const float* obs, const orc_world* w, int R,
                           float g[3]) {
  return loc_logp_grad_abs(x, y, th, m0, m1, m2, s, obs, w, R, g, NULL);
}

/* orc_loc_logp_grad plus, per particle, the scale a 1e-5 relative bar on each SUM has to refer to
 * (gabs: n x 4): for gradient component k, sum_r (|obs_r| + d_r) / sigma^2 |d d_r / d q_k| + |prior
 * term| -- every ray term is the product ((obs_r - d_r) / sigma^2) (d d_r / d q_k), whose first factor is a
 * cancelling difference of the operands |obs_r|, d_r; for logp, sum of |log N| over the sites
 * (test infrastructure). */
ORC_API void orc_loc_logp_grad_abs(const float* x, const float* y, const float* th, const float* mx,
                                   const float* my, const float* mth, int32_t is_initial, int64_t n,
                                   const float* obs, int32_t R, float width, float height, int32_t W,
                                   const float* x1, const float* y1, const float* x2, const float* y2,
                                   float* logp, float* grad /* n x 3 */, float* gabs /* n x 4 */) {
  orc_world w = {width, height, W, x1, y1, x2, y2};
  float s[3];
  loc_sigmas(is_initial, s);
  for (int64_t i = 0; i < n; ++i)
    logp[i] = loc_logp_grad_abs(x[i], y[i], th[i], mx[i], my[i], mth[i], s, obs, &w, R, grad + 3 * i,
                                gabs + 4 * i);
}

ORC_API void orc_loc_logp_grad(const float* x, const float* y, const float* th, const float* mx,
                               const float* my, const float* mth, int32_t is_initial, int64_t n,
                               const float* obs, int32_t R, float width, float height, int32_t W,
                               const float* x1, const float* y1, const float* x2, const float* y2,
                               float* logp, float* grad /* n x 3 */) {
  orc_world w = {width, height, W, x1, y1, x2, y2};
  float s[3];
  loc_sigmas(is_initial, s);
  for (int64_t i = 0; i < n; ++i)
    logp[i] = loc_logp_grad(x[i], y[i], th[i], mx[i], my[i], mth[i], s, obs, &w, R, grad + 3 * i);
}

/* hmc(trace, sel("x")|sel("y")|sel("theta"), step_size, n_steps)  (mcmc.py:434-578).
 * Momenta are drawn in sorted-key order theta, x, y ([3P] tree_map over a dict);
 * kp = {k_theta, k_x, k_y}.  The leapfrog scan consumes one split but draws
 * nothing (Appendix A), so it does not appear here. */
/* diag (nullable): n x 9 floats per particle {x_L, y_L, theta_L, log_alpha (before min0), log(u),
 * logp(q_0), logp(q_L), log N(p_0), log N(-p_L)} (test infrastructure, as orc_loc_mh_diag). */
static int64_t loc_hmc_impl(const uint32_t kpth[2], const uint32_t kpx[2], const uint32_t kpy[2],
                            const uint32_t ku[2], float step_size, int32_t n_steps,
                            int32_t is_initial, int64_t n, uint64_t offset, const float* obs,
                            int32_t R, float width, float height, int32_t W, const float* x1,
                            const float* y1, const float* x2, const float* y2, float* x, float* y,
                            float* th, const float* mx, const float* my, const float* mth,
                            float* lp, uint8_t* accept_out /* nullable */, float* diag /* nullable */) {
  orc_world w = {width, height, W, x1, y1, x2, y2};
  float s[3];
  loc_sigmas(is_initial, s);
  const float half_eps = (float)((double)step_size / 2.0);
  int64_t acc = 0;
#pragma omp parallel for schedule(static) reduction(+ : acc)
  for (int64_t i = 0; i < n; ++i) {
    float q[3] = {x[i], y[i], th[i]};
    float g[3];
    float prev_model = loc_logp_grad(q[0], q[1], q[2], mx[i], my[i], mth[i], s, obs, &w, R, g);
    float pth = normal_at(kpth, offset + (uint64_t)i) * 1.0f + 0.0f;
    float ppx = normal_at(kpx, offset + (uint64_t)i) * 1.0f + 0.0f;
    float ppy = normal_at(kpy, offset + (uint64_t)i) * 1.0f + 0.0f;
    float u = uniform_at(ku, offset + (uint64_t)i, 0.0f, 1.0f);
    float p[3] = {ppx, ppy, pth};
    /* tree_reduce(add) over sorted keys: (theta + x) + y */
    float prev_mom = (normal_logpdf(p[2], 0.0f, 1.0f) + normal_logpdf(p[0], 0.0f, 1.0f)) +
                     normal_logpdf(p[1], 0.0f, 1.0f);
    for (int l = 0; l < n_steps; ++l) {
      for (int k = 0; k < 3; ++k) p[k] = p[k] + half_eps * g[k];
      for (int k = 0; k < 3; ++k) q[k] = q[k] + step_size * p[k];
      (void)loc_logp_grad(q[0], q[1], q[2], mx[i], my[i], mth[i], s, obs, &w, R, g);
      for (int k = 0; k < 3; ++k) p[k] = p[k] + half_eps * g[k];
    }
    float gdummy[3];
    float new_model = loc_logp_grad(q[0], q[1], q[2], mx[i], my[i], mth[i], s, obs, &w, R, gdummy);
    float new_mom = (normal_logpdf(-p[2], 0.0f, 1.0f) + normal_logpdf(-p[0], 0.0f, 1.0f)) +
                    normal_logpdf(-p[1], 0.0f, 1.0f);
    float log_alpha = (new_model + new_mom) - (prev_model + prev_mom);
    if (diag) {
      float* d = diag + 9 * i;
      d[0] = q[0];
      d[1] = q[1];
      d[2] = q[2];
      d[3] = log_alpha;
      d[4] = logf(u);
      d[5] = prev_model;
      d[6] = new_model;
      d[7] = prev_mom;
      d[8] = new_mom;
    }
    log_alpha = min0(log_alpha);
    int accept = logf(u) < log_alpha;
    if (accept) {
      x[i] = q[0];
      y[i] = q[1];
      th[i] = q[2];
      loc_ray_logpdfs(q[0], q[1], q[2], obs, &w, R, lp + i * R);
      acc += 1;
    }
    if (accept_out) accept_out[i] = (uint8_t)accept;
  }
  return acc;
}

ORC_API int64_t orc_loc_hmc(const uint32_t kpth[2], const uint32_t kpx[2], const uint32_t kpy[2],
                            const uint32_t ku[2], float step_size, int32_t n_steps,
                            int32_t is_initial, int64_t n, uint64_t offset, const float* obs,
                            int32_t R, float width, float height, int32_t W, const float* x1,
                            const float* y1, const float* x2, const float* y2, float* x, float* y,
                            float* th, const float* mx, const float* my, const float* mth,
                            float* lp, uint8_t* accept_out /* nullable */) {
  return loc_hmc_impl(kpth, kpx, kpy, ku, step_size, n_steps, is_initial, n, offset, obs, R, width, height, W,
                      x1, y1, x2, y2, x, y, th, mx, my, mth, lp, accept_out, NULL);
}

ORC_API int64_t orc_loc_hmc_diag(const uint32_t kpth[2], const uint32_t kpx[2], const uint32_t kpy[2],
                                 const uint32_t ku[2], float step_size, int32_t n_steps,
                                 int32_t is_initial, int64_t n, uint64_t offset, const float* obs,
                                 int32_t R, float width, float height, int32_t W, const float* x1,
                                 const float* y1, const float* x2, const float* y2, float* x, float* y,
                                 float* th, const float* mx, const float* my, const float* mth,
                                 float* lp, uint8_t* accept_out, float* diag) {
  return loc_hmc_impl(kpth, kpx, kpy, ku, step_size, n_steps, is_initial, n, offset, obs, R, width, height, W,
                      x1, y1, x2, y2, x, y, th, mx, my, mth, lp, accept_out, diag);
}

/* ------------------------------------------------------------------------- */
/* 6. rejuvenation_smc driver for the localization model (smc.py:655-785)      */
/*    with the Seed key schedule (pjax.py:1318-1401)     */
/* ------------------------------------------------------------------------- */

typedef struct {
  uint32_t k[2];
} okey;

/* self.key, sub_key = jrand.split(self.key)   (pjax.py:1337) */
static okey seed_next(okey* state) {
  uint32_t out[4];
  orc_split(state->k, 2, out);
  okey sub = {{out[2], out[3]}};
  state->k[0] = out[0];
  state->k[1] = out[1];
  return sub;
}

static void gather_f32(float* col, const int32_t* idx, int64_t n, int64_t stride) {
  float* tmp = (float*)malloc(sizeof(float) * n * stride);
  for (int64_t j = 0; j < n; ++j)
    for (int64_t c = 0; c < stride; ++c) tmp[j * stride + c] = col[(int64_t)idx[j] * stride + c];
  memcpy(col, tmp, sizeof(float) * n * stride);
  free(tmp);
}

/* kernel: 0 none, 1 mh, 2 hmc.  method: 0 categorical, 1 systematic.
 * Outputs: final columns, lml (final log_marginal_likelihood()), per-step ESS
 * (before the resampling decision), resampled flags, accept counts (T x K). */
ORC_API int orc_loc_smc(const uint32_t key0[2], int64_t n, int32_t T, const float* obs /* T x R */,
                        int32_t R, float width, float height, int32_t W, const float* x1,
                        const float* y1, const float* x2, const float* y2, int32_t kernel,
                        int32_t K, float step_size, int32_t n_steps, int32_t method, float* x,
                        float* y, float* th, float* mx, float* my, float* mth, float* logw,
                        float* lml_out, float* ess_trace, int32_t* resampled,
                        int64_t* accept_counts /* T x K, nullable */, float* lml_acc_out) {
  if (R > ORC_MAX_RAYS) return -1;
  float* lp = (float*)malloc(sizeof(float) * n * R);
  float* wts = (float*)malloc(sizeof(float) * n);
  int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * n);
  for (int64_t i = 0; i < n; ++i) {
    x[i] = y[i] = th[i] = 0.0f; /* dummy pose, ignored at t = 0 (core.py:494) */
    logw[i] = 0.0f;
  }
  float lml_acc = 0.0f;
  okey top = {{key0[0], key0[1]}};
  okey scan_key = {{0, 0}};
  for (int32_t t = 0; t < T; ++t) {
    okey ks;
    if (t == 0) {
      ks = top; /* init + cond + rejuvenation scan run in the top-level interpreter */
    } else {
      /* body of lax.scan(smc_step): fresh interpreter on fold_in(sub, t-1) (pjax.py:1376-1377) */
      orc_fold_in(scan_key.k, (uint32_t)(t - 1), ks.k);
    }
    okey kx = seed_next(&ks), ky = seed_next(&ks), kth = seed_next(&ks);
    orc_loc_extend(kx.k, ky.k, kth.k, t == 0, n, 0, obs + (int64_t)t * R, R, width, height, W, x1,
                   y1, x2, y2, x, y, th, mx, my, mth, logw, lp);
    /* cond_p: one split whether or not the branch is taken (pjax.py:1344) */
    okey kc = seed_next(&ks);
    float ess = orc_ess(logw, n);
    ess_trace[t] = ess;
    int do_resample = ess < (float)(n / 2);
    resampled[t] = do_resample;
    if (do_resample) {
      okey inner = kc; /* fresh interpreter on the cond sub-key */
      okey kdraw = seed_next(&inner);
      float lse = orc_logsumexp(logw, n);
      lml_acc = lml_acc + (lse - logf((float)n)); /* smc.py:628-633 */
      if (method == 1) {
        float u = uniform_at(kdraw.k, 0, 0.0f, 1.0f);
        for (int64_t i = 0; i < n; ++i) wts[i] = expf(logw[i] - lse);
        orc_systematic_indices(wts, n, u, idx, NULL);
      } else {
        orc_categorical_indices(kdraw.k, logw, n, n, idx);
      }
      gather_f32(x, idx, n, 1);
      gather_f32(y, idx, n, 1);
      gather_f32(th, idx, n, 1);
      gather_f32(mx, idx, n, 1);
      gather_f32(my, idx, n, 1);
      gather_f32(mth, idx, n, 1);
      gather_f32(lp, idx, n, R);
      for (int64_t i = 0; i < n; ++i) logw[i] = 0.0f;
    }
    if (kernel != 0) {
      okey kr = seed_next(&ks); /* scan_p: one split (pjax.py:1381) */
      for (int32_t m = 0; m < K; ++m) {
        okey km;
        orc_fold_in(kr.k, (uint32_t)m, km.k);
        int64_t acc;
        if (kernel == 1) {
          okey a = seed_next(&km), b = seed_next(&km), c = seed_next(&km), d = seed_next(&km);
          acc = orc_loc_mh(a.k, b.k, c.k, d.k, t == 0, n, 0, obs + (int64_t)t * R, R, width,
                           height, W, x1, y1, x2, y2, x, y, th, mx, my, mth, lp, NULL);
        } else {
          okey a = seed_next(&km), b = seed_next(&km), c = seed_next(&km);
          (void)seed_next(&km); /* leapfrog scan_p */
          okey d = seed_next(&km);
          acc = orc_loc_hmc(a.k, b.k, c.k, d.k, step_size, n_steps, t == 0, n, 0,
                            obs + (int64_t)t * R, R, width, height, W, x1, y1, x2, y2, x, y, th,
                            mx, my, mth, lp, NULL);
        }
        if (accept_counts) accept_counts[(int64_t)t * K + m] = acc;
      }
    }
    if (t == 0) scan_key = seed_next(&ks); /* scan_p over obs[1:]: one split (pjax.py:1381) */
  }
  float lse = orc_logsumexp(logw, n);
  *lml_out = lml_acc + (lse - logf((float)n)); /* smc.py:193-204 */
  if (lml_acc_out) *lml_acc_out = lml_acc;
  free(lp);
  free(wts);
  free(idx);
  return 0;
}

/* ------------------------------------------------------------------------- */
/* 7. Linear-Gaussian state-space model (src/genjax/extras/state_space.py:377-431) */
/* ------------------------------------------------------------------------- */

static int chol_f32(const float* cov, int d, float* L) {
  for (int i = 0; i < d * d; ++i) L[i] = 0.0f;
  for (int r = 0; r < d; ++r)
    for (int c = 0; c <= r; ++c) {
      float s = cov[r * d + c];
      for (int k = 0; k < c; ++k) s -= L[r * d + k] * L[c * d + k];
      if (r == c) {
        if (!(s > 0.0f)) return -1;
        L[r * d + c] = sqrtf(s);
      } else {
        L[r * d + c] = s / L[c * d + c];
      }
    }
  return 0;
}

/* log MVN(obs; C x, R) through chol(R), TFP's MultivariateNormalFullCovariance.log_prob */
static float lg_obs_lp(int ds, int dobs, const float* Cm, const float* LR, const float* obs,
                       const float* x) {
  float z[8], q = 0.0f, logdet = 0.0f;
  for (int r = 0; r < dobs; ++r) {
    float m = 0.0f;
    for (int c = 0; c < ds; ++c) m += Cm[r * ds + c] * x[c];
    float s = obs[r] - m;
    for (int k = 0; k < r; ++k) s -= LR[r * dobs + k] * z[k];
    z[r] = s / LR[r * dobs + r];
    q += z[r] * z[r];
    logdet += logf(LR[r * dobs + r]);
  }
  return -0.5f * q - logdet - (float)dobs * HALF_LOG_2PI;
}

/* generate() of linear_gaussian with "obs" constrained (state_space.py:413-427 under
 * smc.py:306 / 488-491).  state / mean are SoA: ds rows of n.  One key for the "state" site:
 * MVN sample = mean + chol(cov) @ normal(key, (n, ds)), flat index i*ds + r ([3P] TFP). */
ORC_API int orc_lgssm_extend(const uint32_t key[2], int32_t is_initial, int64_t n, int32_t ds,
                             int32_t dobs, const float* mu0, const float* cov0, const float* A,
                             const float* Q, const float* Cm, const float* R, const float* obs,
                             float* state, float* mean, float* logw) {
  float Lx[64], LR[64];
  if (ds > 8 || dobs > 8) return -1;
  if (chol_f32(is_initial ? cov0 : Q, ds, Lx) || chol_f32(R, dobs, LR)) return -2;
  for (int64_t i = 0; i < n; ++i) {
    float m[8], z[8], x[8];
    for (int r = 0; r < ds; ++r) {
      if (is_initial) {
        m[r] = mu0[r];
      } else {
        float s = 0.0f;
        for (int c = 0; c < ds; ++c) s += A[r * ds + c] * state[(int64_t)c * n + i];
        m[r] = s;
      }
    }
    for (int r = 0; r < ds; ++r) z[r] = normal_at(key, (uint64_t)i * (uint64_t)ds + (uint64_t)r);
    for (int r = 0; r < ds; ++r) {
      float s = 0.0f;
      for (int c = 0; c <= r; ++c) s += Lx[r * ds + c] * z[c];
      x[r] = m[r] + s;
    }
    for (int r = 0; r < ds; ++r) {
      state[(int64_t)r * n + i] = x[r];
      mean[(int64_t)r * n + i] = m[r];
    }
    logw[i] = logw[i] + lg_obs_lp(ds, dobs, Cm, LR, obs, x);
  }
  return 0;
}

/* mh(trace, sel("state")) (mcmc.py:267-309): regenerate "state" from its prior at the stored
 * mean; weight = logpdf_obs(new) + old_score_obs. */
/* diag (nullable): n x (ds + 3) floats per particle {proposal (ds), weight before min0, log u,
 * |logpdf_obs(new)| + |logpdf_obs(old)|} for the teacher-forced per-move parity test (test infrastructure). */
static int64_t lgssm_mh_impl(const uint32_t kstate[2], const uint32_t ku[2], int32_t is_initial,
                             int64_t n, int32_t ds, int32_t dobs, const float* cov0, const float* Q,
                             const float* Cm, const float* R, const float* obs, float* state,
                             const float* mean, float* diag);

ORC_API int64_t orc_lgssm_mh(const uint32_t kstate[2], const uint32_t ku[2], int32_t is_initial,
                             int64_t n, int32_t ds, int32_t dobs, const float* cov0, const float* Q,
                             const float* Cm, const float* R, const float* obs, float* state,
                             const float* mean) {
  return lgssm_mh_impl(kstate, ku, is_initial, n, ds, dobs, cov0, Q, Cm, R, obs, state, mean, NULL);
}

ORC_API int64_t orc_lgssm_mh_diag(const uint32_t kstate[2], const uint32_t ku[2], int32_t is_initial,
                                  int64_t n, int32_t ds, int32_t dobs, const float* cov0, const float* Q,
                                  const float* Cm, const float* R, const float* obs, float* state,
                                  const float* mean, float* diag) {
  return lgssm_mh_impl(kstate, ku, is_initial, n, ds, dobs, cov0, Q, Cm, R, obs, state, mean, diag);
}

static int64_t lgssm_mh_impl(const uint32_t kstate[2], const uint32_t ku[2], int32_t is_initial,
                             int64_t n, int32_t ds, int32_t dobs, const float* cov0, const float* Q,
                             const float* Cm, const float* R, const float* obs, float* state,
                             const float* mean, float* diag) {
  float Lx[64], LR[64];
  if (chol_f32(is_initial ? cov0 : Q, ds, Lx) || chol_f32(R, dobs, LR)) return -2;
  int64_t acc = 0;
  for (int64_t i = 0; i < n; ++i) {
    float cur[8], z[8], x[8];
    for (int r = 0; r < ds; ++r) cur[r] = state[(int64_t)r * n + i];
    for (int r = 0; r < ds; ++r) z[r] = normal_at(kstate, (uint64_t)i * (uint64_t)ds + (uint64_t)r);
    for (int r = 0; r < ds; ++r) {
      float s = 0.0f;
      for (int c = 0; c <= r; ++c) s += Lx[r * ds + c] * z[c];
      x[r] = mean[(int64_t)r * n + i] + s;
    }
    float u = uniform_at(ku, (uint64_t)i, 0.0f, 1.0f);
    const float lp_new = lg_obs_lp(ds, dobs, Cm, LR, obs, x), lp_old = lg_obs_lp(ds, dobs, Cm, LR, obs, cur);
    float wt = lp_new + (-lp_old);
    if (diag) {
      float* d = diag + (int64_t)(ds + 3) * i;
      for (int r = 0; r < ds; ++r) d[r] = x[r];
      d[ds] = wt;
      d[ds + 1] = logf(u);
      d[ds + 2] = fabsf(lp_new) + fabsf(lp_old);
    }
    if (logf(u) < min0(wt)) {
      for (int r = 0; r < ds; ++r) state[(int64_t)r * n + i] = x[r];
      acc += 1;
    }
  }
  return acc;
}

/* ------------------------------------------------------------------------- */
/* 9. Outlier-mixture curve fitting: enumerative Gibbs + HMC per chain        */
/*    (examples/curvefit/core.py:468-552 model, :934-1014 kernel;              */
/*     harness examples/curvefit/figs.py:907-950)                              */
/* ------------------------------------------------------------------------- */
/* Model npoint_curve_with_outliers_beta(xs, alpha, beta):
 *   outlier_rate ~ beta(alpha, beta); a, b, c ~ normal(0, 1);
 *   per point: is_outlier ~ flip(rate);  y ~ Cond(uniform(-2, 2), normal(a + b x + c x^2, 0.1)).
 * Cond evaluates both branches and selects with jnp.where (core.py:2721-2730): no cond_p, no
 * key split.  The Gibbs code calls point_with_outliers.assess with five arguments although the
 * model takes three (core.py:953-960 vs :483); restated from the model.
 * Per-chain key schedule (figs.py:926-941; pjax.py:1333-1338): chain key = split(key, n)[i];
 * iteration: key, sub = split(key); Seed(sub): S -> Gibbs categorical noise (P, 2); S x 4 -> HMC
 * momenta in tree order curve.a, curve.b, curve.c, outlier_rate; S -> leapfrog scan (unused);
 * S -> u.  Scalar draws use counter 0 of their sub-key.
 * [3P, parity unpinned] tfd.Beta.sample is a rejection sampler over gamma draws whose draw count
 * depends on the data; it is NOT reproduced.  The init draw of outlier_rate below is
 * X = Ga / (Ga + Gb) with Marsaglia-Tsang gammas on counter-based draws of the site key
 * (same law, different bits). */

typedef struct {
  int32_t n_points;
  float xs[64], ys[64];
  float alpha, beta, inlier_sigma, out_lo, out_hi;
} orc_outlier_model;

static float mt_gamma(const uint32_t key[2], uint64_t stream, float shape) {
  const float a = shape < 1.0f ? shape + 1.0f : shape;
  const float d = a - 1.0f / 3.0f, c = 1.0f / sqrtf(9.0f * d);
  float g = d;
  for (uint64_t t = 0; t < 64; ++t) {
    const float x = normal_at(key, (stream << 16) + 2 * t);
    const float u = uniform_at(key, (stream << 16) + 2 * t + 1, 0.0f, 1.0f);
    float v = 1.0f + c * x;
    if (v <= 0.0f) continue;
    v = v * v * v;
    g = d * v;
    if (logf(u) < 0.5f * x * x + d - d * v + d * logf(v)) break;
  }
  if (shape < 1.0f) {
    const float u = uniform_at(key, (stream << 16) + 200, 0.0f, 1.0f);
    g = g * powf(u, 1.0f / shape);
  }
  return g;
}

static float bern_logpdf(float x, float p) {
  float a = (x == 0.0f) ? 0.0f : x * logf(p);
  float b = (1.0f - x == 0.0f) ? 0.0f : (1.0f - x) * log1pf(-p);
  return a + b;
}

static float beta_logpdf1(float x, float a, float b) {
  float lbeta = lgammaf(a) + lgammaf(b) - lgammaf(a + b);
  float t1 = (a - 1.0f == 0.0f) ? 0.0f : (a - 1.0f) * logf(x);
  float t2 = (b - 1.0f == 0.0f) ? 0.0f : (b - 1.0f) * log1pf(-x);
  return (x < 0.0f || x > 1.0f) ? -INFINITY : t1 + t2 - lbeta;
}

/* assess() of the whole model at (a, b, c, rate, outliers) and its gradient w.r.t. the four
 * continuous choices (the function jax.grad differentiates inside hmc, mcmc.py:59-88,508). */
static float outlier_logp_grad(const orc_outlier_model* m, const float q[4], uint64_t mask,
                               float g[4]) {
  const float a = q[0], b = q[1], c = q[2], rate = q[3];
  float lp = beta_logpdf1(rate, m->alpha, m->beta);
  lp += normal_logpdf(a, 0.0f, 1.0f);
  lp += normal_logpdf(b, 0.0f, 1.0f);
  lp += normal_logpdf(c, 0.0f, 1.0f);
  float ga = -a, gb = -b, gc = -c;
  float grate = (m->alpha - 1.0f == 0.0f ? 0.0f : (m->alpha - 1.0f) / rate) -
                (m->beta - 1.0f == 0.0f ? 0.0f : (m->beta - 1.0f) / (1.0f - rate));
  const float inv_var = 1.0f / (m->inlier_sigma * m->inlier_sigma);
  const float lp_out = -logf(m->out_hi - m->out_lo);
  float ys_lp = 0.0f;
  for (int j = 0; j < m->n_points; ++j) {
    const float x = m->xs[j], y = m->ys[j];
    const int o = (int)((mask >> j) & 1u);
    const float mean = a + b * x + c * (x * x);
    float pt = bern_logpdf((float)o, rate);
    if (o) {
      pt += (y < m->out_lo || y > m->out_hi) ? -INFINITY : lp_out;
      grate += 1.0f / rate;
    } else {
      pt += normal_logpdf(y, mean, m->inlier_sigma);
      const float r = (y - mean) * inv_var;
      ga += r;
      gb += r * x;
      gc += r * (x * x);
      grate -= 1.0f / (1.0f - rate);
    }
    ys_lp += pt;
  }
  lp += ys_lp;
  g[0] = ga;
  g[1] = gb;
  g[2] = gc;
  g[3] = grate;
  return lp;
}

static void key_split2(const uint32_t k[2], uint32_t k0[2], uint32_t k1[2]) {
  uint32_t out[4];
  orc_split(k, 2, out);
  k0[0] = out[0];
  k0[1] = out[1];
  k1[0] = out[2];
  k1[1] = out[3];
}

/* generate(constraints = {"ys": {"y": {"obs": ys}}}) per chain: the prior draw of
 * (rate, a, b, c, is_outlier) under Seed(chain_key) in program order (figs.py:926-932). */
ORC_API void orc_outlier_init(const orc_outlier_model* m, const uint32_t base_key[2],
                              uint64_t offset, int64_t n, uint32_t* chain_keys, float* a, float* b,
                              float* c, float* rate, uint64_t* outliers) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    uint32_t ck[2], st[2], sub[2], nx[2];
    const uint32_t ctr[2] = {(uint32_t)((offset + (uint64_t)i) >> 32), (uint32_t)(offset + (uint64_t)i)};
    orc_threefry2x32(base_key, ctr, ck); /* split(key, n)[i] */
    chain_keys[2 * i] = ck[0];
    chain_keys[2 * i + 1] = ck[1];
    st[0] = ck[0];
    st[1] = ck[1];
    key_split2(st, nx, sub); /* outlier_rate ~ beta */
    const float ga = mt_gamma(sub, 0, m->alpha), gb = mt_gamma(sub, 1, m->beta);
    rate[i] = ga / (ga + gb);
    st[0] = nx[0]; st[1] = nx[1];
    key_split2(st, nx, sub);
    a[i] = normal_at(sub, 0) * 1.0f + 0.0f;
    st[0] = nx[0]; st[1] = nx[1];
    key_split2(st, nx, sub);
    b[i] = normal_at(sub, 0) * 1.0f + 0.0f;
    st[0] = nx[0]; st[1] = nx[1];
    key_split2(st, nx, sub);
    c[i] = normal_at(sub, 0) * 1.0f + 0.0f;
    st[0] = nx[0]; st[1] = nx[1];
    key_split2(st, nx, sub); /* is_outlier ~ flip(rate), sample_shape (P,) */
    uint64_t mask = 0;
    for (int j = 0; j < m->n_points; ++j)
      if (uniform_at(sub, (uint64_t)j, 0.0f, 1.0f) < rate[i]) mask |= (uint64_t)1 << j;
    outliers[i] = mask;
  }
}

/* n_iters applications of mixed_gibbs_hmc_kernel per chain (core.py:988-1014). */
ORC_API int64_t orc_outlier_gibbs_hmc(const orc_outlier_model* m, int32_t n_iters, float step_size,
                                      int32_t n_steps, int64_t n, uint32_t* chain_keys, float* a,
                                      float* b, float* c, float* rate, uint64_t* outliers) {
  const float half_eps = (float)((double)step_size / 2.0);
  const float tiny = 1.17549435e-38f;
  const float lp_out = -logf(m->out_hi - m->out_lo);
  int64_t acc = 0;
#pragma omp parallel for schedule(static) reduction(+ : acc)
  for (int64_t i = 0; i < n; ++i) {
    uint32_t key[2] = {chain_keys[2 * i], chain_keys[2 * i + 1]};
    float q[4] = {a[i], b[i], c[i], rate[i]};
    uint64_t mask = outliers[i];
    for (int it = 0; it < n_iters; ++it) {
      uint32_t nk[2], st[2], sub[2], nx[2];
      key_split2(key, nk, st); /* key, subkey = jrand.split(key)   figs.py:928 */
      key[0] = nk[0];
      key[1] = nk[1];
      /* --- enumerative Gibbs over the outlier indicators (core.py:934-985) --- */
      key_split2(st, nx, sub);
      st[0] = nx[0]; st[1] = nx[1];
      uint64_t nm = 0;
      for (int j = 0; j < m->n_points; ++j) {
        const float x = m->xs[j], y = m->ys[j];
        const float mean = q[0] + q[1] * x + q[2] * (x * x);
        const float l0 = bern_logpdf(0.0f, q[3]) + normal_logpdf(y, mean, m->inlier_sigma);
        const float l1 = bern_logpdf(1.0f, q[3]) +
                         ((y < m->out_lo || y > m->out_hi) ? -INFINITY : lp_out);
        /* categorical.sample(logits (P, 2)) = argmax(logits + gumbel(key, (P, 2)))  [3P] */
        const float u0 = uniform_at(sub, 2 * (uint64_t)j, tiny, 1.0f);
        const float u1 = uniform_at(sub, 2 * (uint64_t)j + 1, tiny, 1.0f);
        const float v0 = l0 + (-logf(-logf(u0))), v1 = l1 + (-logf(-logf(u1)));
        if (v1 > v0) nm |= (uint64_t)1 << j;
      }
      mask = nm;
      /* --- hmc on (curve.a, curve.b, curve.c, outlier_rate)  (mcmc.py:434-578) --- */
      float p[4], g[4], gd[4];
      const float prev_model = outlier_logp_grad(m, q, mask, g);
      for (int k = 0; k < 4; ++k) {
        key_split2(st, nx, sub);
        st[0] = nx[0]; st[1] = nx[1];
        p[k] = normal_at(sub, 0) * 1.0f + 0.0f;
      }
      key_split2(st, nx, sub); /* leapfrog scan_p: one split, unused (pjax.py:1381) */
      st[0] = nx[0]; st[1] = nx[1];
      key_split2(st, nx, sub);
      const float u = uniform_at(sub, 0, 0.0f, 1.0f);
      const float prev_mom = ((normal_logpdf(p[0], 0.0f, 1.0f) + normal_logpdf(p[1], 0.0f, 1.0f)) +
                              normal_logpdf(p[2], 0.0f, 1.0f)) +
                             normal_logpdf(p[3], 0.0f, 1.0f);
      float qq[4] = {q[0], q[1], q[2], q[3]};
      for (int l = 0; l < n_steps; ++l) {
        for (int k = 0; k < 4; ++k) p[k] = p[k] + half_eps * g[k];
        for (int k = 0; k < 4; ++k) qq[k] = qq[k] + step_size * p[k];
        (void)outlier_logp_grad(m, qq, mask, g);
        for (int k = 0; k < 4; ++k) p[k] = p[k] + half_eps * g[k];
      }
      const float new_model = outlier_logp_grad(m, qq, mask, gd);
      const float new_mom = ((normal_logpdf(-p[0], 0.0f, 1.0f) + normal_logpdf(-p[1], 0.0f, 1.0f)) +
                             normal_logpdf(-p[2], 0.0f, 1.0f)) +
                            normal_logpdf(-p[3], 0.0f, 1.0f);
      float log_alpha = (new_model + new_mom) - (prev_model + prev_mom);
      log_alpha = min0(log_alpha);
      if (logf(u) < log_alpha) {
        for (int k = 0; k < 4; ++k) q[k] = qq[k];
        acc += 1;
      }
    }
    chain_keys[2 * i] = key[0];
    chain_keys[2 * i + 1] = key[1];
    a[i] = q[0];
    b[i] = q[1];
    c[i] = q[2];
    rate[i] = q[3];
    outliers[i] = mask;
  }
  return acc;
}

/* ONE application of mixed_gibbs_hmc_kernel per chain (core.py:988-1014) with the quantities of its decisions,
 * for the teacher-forced parity test (test infrastructure).  diag: (10 + P) COLUMNS of n floats:
 *   0..3  the HMC proposal (end point of the leapfrog trajectory), 4  log alpha before min(0, .), 5  log u,
 *   6  assess() at the start, 7  assess() at the end point, 8 / 9  the momentum log-densities at start / end,
 *   10 + j  v1 - v0 of point j in the Gibbs step (the indicator is 1 iff it is > 0),
 *   10 + P + j  |l0| + |l1| + |gumbel0| + |gumbel1| of point j: the terms v1 - v0 is a sum of (its bar). */
ORC_API int64_t orc_outlier_gibbs_hmc_diag(const orc_outlier_model* m, float step_size, int32_t n_steps, int64_t n,
                                           uint32_t* chain_keys, float* a, float* b, float* c, float* rate,
                                           uint64_t* outliers, float* diag) {
  const float half_eps = (float)((double)step_size / 2.0);
  const float tiny = 1.17549435e-38f;
  const float lp_out = -logf(m->out_hi - m->out_lo);
  int64_t acc = 0;
#pragma omp parallel for schedule(static) reduction(+ : acc)
  for (int64_t i = 0; i < n; ++i) {
    uint32_t key[2] = {chain_keys[2 * i], chain_keys[2 * i + 1]};
    float q[4] = {a[i], b[i], c[i], rate[i]};
    uint32_t nk[2], st[2], sub[2], nx[2];
    key_split2(key, nk, st);
    key_split2(st, nx, sub);
    st[0] = nx[0]; st[1] = nx[1];
    uint64_t nm = 0;
    for (int j = 0; j < m->n_points; ++j) {
      const float x = m->xs[j], y = m->ys[j];
      const float mean = q[0] + q[1] * x + q[2] * (x * x);
      const float l0 = bern_logpdf(0.0f, q[3]) + normal_logpdf(y, mean, m->inlier_sigma);
      const float l1 = bern_logpdf(1.0f, q[3]) + ((y < m->out_lo || y > m->out_hi) ? -INFINITY : lp_out);
      const float u0 = uniform_at(sub, 2 * (uint64_t)j, tiny, 1.0f);
      const float u1 = uniform_at(sub, 2 * (uint64_t)j + 1, tiny, 1.0f);
      const float v0 = l0 + (-logf(-logf(u0))), v1 = l1 + (-logf(-logf(u1)));
      if (v1 > v0) nm |= (uint64_t)1 << j;
      diag[(int64_t)(10 + j) * n + i] = v1 - v0;
      diag[(int64_t)(10 + m->n_points + j) * n + i] =
          (fabsf(l0) + fabsf(isinf(l1) ? 0.0f : l1)) + (fabsf(logf(-logf(u0))) + fabsf(logf(-logf(u1))));
    }
    const uint64_t mask = nm;
    float p[4], g[4], gd[4];
    const float prev_model = outlier_logp_grad(m, q, mask, g);
    for (int k = 0; k < 4; ++k) {
      key_split2(st, nx, sub);
      st[0] = nx[0]; st[1] = nx[1];
      p[k] = normal_at(sub, 0) * 1.0f + 0.0f;
    }
    key_split2(st, nx, sub);
    st[0] = nx[0]; st[1] = nx[1];
    key_split2(st, nx, sub);
    const float u = uniform_at(sub, 0, 0.0f, 1.0f);
    const float prev_mom = ((normal_logpdf(p[0], 0.0f, 1.0f) + normal_logpdf(p[1], 0.0f, 1.0f)) +
                            normal_logpdf(p[2], 0.0f, 1.0f)) + normal_logpdf(p[3], 0.0f, 1.0f);
    float qq[4] = {q[0], q[1], q[2], q[3]};
    for (int l = 0; l < n_steps; ++l) {
      for (int k = 0; k < 4; ++k) p[k] = p[k] + half_eps * g[k];
      for (int k = 0; k < 4; ++k) qq[k] = qq[k] + step_size * p[k];
      (void)outlier_logp_grad(m, qq, mask, g);
      for (int k = 0; k < 4; ++k) p[k] = p[k] + half_eps * g[k];
    }
    const float new_model = outlier_logp_grad(m, qq, mask, gd);
    const float new_mom = ((normal_logpdf(-p[0], 0.0f, 1.0f) + normal_logpdf(-p[1], 0.0f, 1.0f)) +
                           normal_logpdf(-p[2], 0.0f, 1.0f)) + normal_logpdf(-p[3], 0.0f, 1.0f);
    const float log_alpha = (new_model + new_mom) - (prev_model + prev_mom);
    for (int k = 0; k < 4; ++k) diag[(int64_t)k * n + i] = qq[k];
    diag[4 * n + i] = log_alpha;
    diag[5 * n + i] = logf(u);
    diag[6 * n + i] = prev_model;
    diag[7 * n + i] = new_model;
    diag[8 * n + i] = prev_mom;
    diag[9 * n + i] = new_mom;
    if (logf(u) < min0(log_alpha)) {
      for (int k = 0; k < 4; ++k) q[k] = qq[k];
      acc += 1;
    }
    chain_keys[2 * i] = nk[0];
    chain_keys[2 * i + 1] = nk[1];
    a[i] = q[0];
    b[i] = q[1];
    c[i] = q[2];
    rate[i] = q[3];
    outliers[i] = mask;
  }
  return acc;
}

/* The HMC half of the same iteration evaluated in DOUBLE precision from the same f32 state, f32 draws and the
 * indicators the f32 Gibbs step chose: the exact value of the reference's formulas on these inputs, up to 1e-16.
 * Test infrastructure: a chain whose f32 evaluations disagree beyond 1e-5 is judged by each evaluation's distance
 * from this value (oracle/parity.py check_outlier_iteration).  State is NOT advanced.  diag: 10 columns of n
 * doubles, rows as in orc_outlier_gibbs_hmc_diag. */
static double outlier_logp_grad64(const orc_outlier_model* m, const double q[4], uint64_t mask, double g[4]) {
  const double a = q[0], b = q[1], c = q[2], rate = q[3];
  const double al = (double)m->alpha, be = (double)m->beta, sg = (double)m->inlier_sigma;
  const double lbeta = lgamma(al) + lgamma(be) - lgamma(al + be);
  double lp = (rate < 0.0 || rate > 1.0) ? -INFINITY
                                         : ((al - 1.0 == 0.0 ? 0.0 : (al - 1.0) * log(rate)) +
                                            (be - 1.0 == 0.0 ? 0.0 : (be - 1.0) * log1p(-rate)) - lbeta);
  const double hl2pi = 0.91893853320467274178;
  lp += -0.5 * a * a - hl2pi;
  lp += -0.5 * b * b - hl2pi;
  lp += -0.5 * c * c - hl2pi;
  double ga = -a, gb = -b, gc = -c;
  double grate = (al - 1.0 == 0.0 ? 0.0 : (al - 1.0) / rate) - (be - 1.0 == 0.0 ? 0.0 : (be - 1.0) / (1.0 - rate));
  const double lp_out = -log((double)m->out_hi - (double)m->out_lo);
  for (int j = 0; j < m->n_points; ++j) {
    const double x = (double)m->xs[j], y = (double)m->ys[j];
    const int o = (int)((mask >> j) & 1u);
    const double mean = a + b * x + c * (x * x);
    if (o) {
      lp += log(rate) + ((y < m->out_lo || y > m->out_hi) ? -INFINITY : lp_out);
      grate += 1.0 / rate;
    } else {
      const double z = (y - mean) / sg;
      lp += log1p(-rate) + (-0.5 * z * z - log(sg) - hl2pi);
      const double r = (y - mean) / (sg * sg);
      ga += r;
      gb += r * x;
      gc += r * (x * x);
      grate -= 1.0 / (1.0 - rate);
    }
  }
  g[0] = ga;
  g[1] = gb;
  g[2] = gc;
  g[3] = grate;
  return lp;
}

/* noise_seed != 0: every stored q, p and gradient component of the trajectory is multiplied by 1 + d, d uniform in
 * +-2^-24 (q, p: one f32 rounding) or +-2^-22 (gradient: a sum of ~20 rounded terms), from a counter-based
 * stream of (noise_seed, chain, step): a randomised model of WHERE f32 roundings enter, to measure how far the
 * dynamics of this chain carry them (stochastic-arithmetic style error estimate). */
static inline double f64_noise(uint32_t seed, uint64_t i, uint32_t slot, double scale) {
  uint32_t o0, o1;
  tf2x32(seed, 0x9E3779B9u, (uint32_t)i ^ (uint32_t)(i >> 32), slot, &o0, &o1);
  return 1.0 + scale * ((double)o0 / 2147483648.0 - 1.0);
}

ORC_API void orc_outlier_hmc_f64(const orc_outlier_model* m, float step_size, int32_t n_steps, int64_t n,
                                 const uint32_t* chain_keys, const float* a, const float* b, const float* c,
                                 const float* rate, const uint64_t* masks_after_gibbs, uint32_t noise_seed,
                                 double* diag) {
  const double eps = (double)step_size, half_eps = (double)step_size / 2.0, hl2pi = 0.91893853320467274178;
  const double e24 = noise_seed ? 5.9604644775390625e-08 : 0.0, e22 = 4.0 * e24;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    uint32_t key[2] = {chain_keys[2 * i], chain_keys[2 * i + 1]};
    uint32_t nk[2], st[2], sub[2], nx[2];
    key_split2(key, nk, st);
    key_split2(st, nx, sub); /* the Gibbs site */
    st[0] = nx[0]; st[1] = nx[1];
    double q[4] = {a[i], b[i], c[i], rate[i]}, p[4], g[4], gd[4];
    const uint64_t mask = masks_after_gibbs[i];
    const double prev_model = outlier_logp_grad64(m, q, mask, g);
    double prev_mom = 0.0;
    for (int k = 0; k < 4; ++k) {
      key_split2(st, nx, sub);
      st[0] = nx[0]; st[1] = nx[1];
      p[k] = (double)(normal_at(sub, 0) * 1.0f + 0.0f);
      prev_mom += -0.5 * p[k] * p[k] - hl2pi;
    }
    key_split2(st, nx, sub);
    st[0] = nx[0]; st[1] = nx[1];
    key_split2(st, nx, sub);
    const float u = uniform_at(sub, 0, 0.0f, 1.0f);
    if (noise_seed)
      for (int k = 0; k < 4; ++k) g[k] *= f64_noise(noise_seed, (uint64_t)i, 4000u + k, e22);
    for (int l = 0; l < n_steps; ++l) {
      const uint32_t s0 = 16u * (uint32_t)l;
      for (int k = 0; k < 4; ++k) p[k] = (p[k] + half_eps * g[k]) * f64_noise(noise_seed, (uint64_t)i, s0 + k, e24);
      for (int k = 0; k < 4; ++k) q[k] = (q[k] + eps * p[k]) * f64_noise(noise_seed, (uint64_t)i, s0 + 4 + k, e24);
      (void)outlier_logp_grad64(m, q, mask, g);
      if (noise_seed)
        for (int k = 0; k < 4; ++k) g[k] *= f64_noise(noise_seed, (uint64_t)i, s0 + 8 + k, e22);
      for (int k = 0; k < 4; ++k) p[k] = (p[k] + half_eps * g[k]) * f64_noise(noise_seed, (uint64_t)i, s0 + 12 + k, e24);
    }
    const double new_model = outlier_logp_grad64(m, q, mask, gd);
    double new_mom = 0.0;
    for (int k = 0; k < 4; ++k) new_mom += -0.5 * p[k] * p[k] - hl2pi;
    for (int k = 0; k < 4; ++k) diag[(int64_t)k * n + i] = q[k];
    diag[4 * n + i] = (new_model + new_mom) - (prev_model + prev_mom);
    diag[5 * n + i] = log((double)u);
    diag[6 * n + i] = prev_model;
    diag[7 * n + i] = new_model;
    diag[8 * n + i] = prev_mom;
    diag[9 * n + i] = new_mom;
  }
}

/* the model's log density / gradient, exposed for finite-difference tests */
ORC_API float orc_outlier_logp_grad(const orc_outlier_model* m, const float q[4], uint64_t mask,
                                    float g[4]) {
  return outlier_logp_grad(m, q, mask, g);
}

/* ------------------------------------------------------------------------- */
/* 10. Game-of-Life inversion: single-site Gibbs sweeps (examples/gol/core.py)  */
/* ------------------------------------------------------------------------- */
/* get_cell_from_window (core.py:24-45): rule of Life on a 3x3 window, then
 * bit ~ flip(det ? 1 - p_flip : p_flip);  flip(p).logpdf(x) = x log p + (1 - x) log1p(-p)
 * (tfd.Bernoulli, multiply_no_nan).  Boards are n x n bytes, zero padded (core.py:56). */
static inline int gol_rule(int centre, int nb) {
  return centre ? (nb == 2 || nb == 3) : (nb == 3);
}

static inline int gol_at(const uint8_t* b, int n, int y, int x) {
  return (y >= 0 && y < n && x >= 0 && x < n) ? b[(int64_t)y * n + x] : 0;
}

ORC_API void orc_gol_step(const uint8_t* board, int32_t n, uint8_t* next) {
  for (int y = 0; y < n; ++y)
    for (int x = 0; x < n; ++x) {
      int nb = 0;
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx)
          if (dy || dx) nb += gol_at(board, n, y + dy, x + dx);
      next[(int64_t)y * n + x] = (uint8_t)gol_rule(board[(int64_t)y * n + x], nb);
    }
}

/* logpdf table [bit][det] of flip(det ? 1 - f : f) */
static void gol_table(float f, float t[2][2]) {
  const float one_minus_f = 1.0f - f;
  t[1][1] = logf(one_minus_f);      /* bit 1, p = 1 - f */
  t[1][0] = logf(f);                /* bit 1, p = f     */
  t[0][1] = log1pf(-one_minus_f);   /* bit 0, p = 1 - f */
  t[0][0] = log1pf(-f);             /* bit 0, p = f     */
}

/* get_gibbs_probs_fast (core.py:108-153) for cell (i, j): P(cell = 0 | rest). */
static float gol_p_zero(const uint8_t* cur, const uint8_t* fut, int n, int i, int j,
                        const float t[2][2]) {
  float score[2];
  for (int val = 0; val < 2; ++val) {
    float s = 0.0f;
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        const int fy = i + dy, fx = j + dx;
        /* the reference's mask compares next_y with shape[1] and next_x with shape[0]
         * (core.py:141-149); boards are square here */
        if (!(fy >= 0 && fy < n && fx >= 0 && fx < n)) continue;
        int nb = 0, centre = 0;
        for (int wy = -1; wy <= 1; ++wy)
          for (int wx = -1; wx <= 1; ++wx) {
            const int y = fy + wy, x = fx + wx;
            const int v = (y == i && x == j) ? val : gol_at(cur, n, y, x);
            if (wy == 0 && wx == 0) centre = v; else nb += v;
          }
        s += t[fut[(int64_t)fy * n + fx]][gol_rule(centre, nb)];
      }
    score[val] = s;
  }
  const float m = score[0] > score[1] ? score[0] : score[1];
  const float e0 = expf(score[0] - m), e1 = expf(score[1] - m);
  return e0 / (e0 + e1);  /* jax.nn.softmax(scores)[0] */
}

/* gibbs_move_on_all_cells_at_offset (core.py:166-185): cells (oy + 3a, ox + 3b); the flip draw
 * of cell (a, b) is uniform(key, (ny', nx'))[a, b] < 1 - p_zero  [3P tfd.Bernoulli.sample]. */
ORC_API void orc_gol_gibbs_pass(uint8_t* cur, const uint8_t* fut, int32_t n, int32_t oy, int32_t ox,
                                float p_flip, const uint32_t key[2]) {
  float t[2][2];
  gol_table(p_flip, t);
  const int nyp = (n - oy + 2) / 3, nxp = (n - ox + 2) / 3;
  uint8_t* vals = (uint8_t*)malloc((size_t)nyp * nxp);
#pragma omp parallel for schedule(static)
  for (int a = 0; a < nyp; ++a)
    for (int b = 0; b < nxp; ++b) {
      const float p_zero = gol_p_zero(cur, fut, n, oy + 3 * a, ox + 3 * b, t);
      const float u = uniform_at(key, (uint64_t)a * (uint64_t)nxp + (uint64_t)b, 0.0f, 1.0f);
      vals[(int64_t)a * nxp + b] = (uint8_t)(u < 1.0f - p_zero);
    }
  for (int a = 0; a < nyp; ++a)
    for (int b = 0; b < nxp; ++b) cur[(int64_t)(oy + 3 * a) * n + (ox + 3 * b)] = vals[(int64_t)a * nxp + b];
  free(vals);
}

/* orc_gol_gibbs_pass that also reports, per updated cell (a, b) in row-major order, the margin of its decision
 * u - (1 - p_zero) (the new bit is 1 iff it is < 0): test infrastructure for the per-cell parity listing. */
ORC_API void orc_gol_gibbs_pass_margins(uint8_t* cur, const uint8_t* fut, int32_t n, int32_t oy, int32_t ox,
                                        float p_flip, const uint32_t key[2], float* margins /* nyp x nxp */) {
  float t[2][2];
  gol_table(p_flip, t);
  const int nyp = (n - oy + 2) / 3, nxp = (n - ox + 2) / 3;
  uint8_t* vals = (uint8_t*)malloc((size_t)nyp * nxp);
#pragma omp parallel for schedule(static)
  for (int a = 0; a < nyp; ++a)
    for (int b = 0; b < nxp; ++b) {
      const float p_zero = gol_p_zero(cur, fut, n, oy + 3 * a, ox + 3 * b, t);
      const float u = uniform_at(key, (uint64_t)a * (uint64_t)nxp + (uint64_t)b, 0.0f, 1.0f);
      vals[(int64_t)a * nxp + b] = (uint8_t)(u < 1.0f - p_zero);
      margins[(int64_t)a * nxp + b] = u - (1.0f - p_zero);
    }
  for (int a = 0; a < nyp; ++a)
    for (int b = 0; b < nxp; ++b) cur[(int64_t)(oy + 3 * a) * n + (ox + 3 * b)] = vals[(int64_t)a * nxp + b];
  free(vals);
}

/* full_gibbs_sweep (core.py:188-204): the nine colour offsets in row-major order, one sample
 * site (one split) each. */
ORC_API void orc_gol_sweep(uint8_t* cur, const uint8_t* fut, int32_t n, float p_flip,
                           const uint32_t sweep_key[2]) {
  okey ks = {{sweep_key[0], sweep_key[1]}};
  for (int oy = 0; oy < 3; ++oy)
    for (int ox = 0; ox < 3; ++ox) {
      okey sub = seed_next(&ks);
      orc_gol_gibbs_pass(cur, fut, n, oy, ox, p_flip, sub.k);
    }
}

/* proposal.simulate (core.py:240-249): init ~ flip(0.5) per cell, one site of shape (n, n). */
ORC_API void orc_gol_init(const uint32_t key[2], int32_t n, uint8_t* board) {
  for (int64_t c = 0; c < (int64_t)n * n; ++c)
    board[c] = (uint8_t)(uniform_at(key, (uint64_t)c, 0.0f, 1.0f) < 0.5f);
}

/* GibbsSamplerState summaries (core.py:207-230): predictive_posterior_score = sum of the step
 * bits' log-densities; n_incorrect = bits of step(board, p_flip = 0) that differ from the target. */
ORC_API void orc_gol_summary(const uint8_t* cur, const uint8_t* fut, int32_t n, float p_flip,
                             int64_t* n_incorrect, double* pred_score) {
  float t[2][2];
  gol_table(p_flip, t);
  uint8_t* nx = (uint8_t*)malloc((size_t)n * n);
  orc_gol_step(cur, n, nx);
  int64_t bad = 0;
  double sc = 0.0;
  for (int64_t c = 0; c < (int64_t)n * n; ++c) {
    bad += nx[c] != fut[c];
    sc += (double)t[fut[c]][nx[c]];
  }
  free(nx);
  *n_incorrect = bad;
  *pred_score = sc;
}

/* seed(_run_sampler_and_get_summary)(key, sampler) (core.py:312-364) key schedule:
 * S -> init flips; S -> outer scan (frames), iteration f on fold_in(., f): S -> inner scan,
 * iteration g on fold_in(., g): the nine colour passes.  Writes the board after every frame. */
ORC_API void orc_gol_run(const uint32_t key[2], const uint8_t* target, int32_t n, float p_flip,
                         int32_t n_frames, int32_t per_frame, uint8_t* boards /* (n_frames+1) n n */) {
  okey ks = {{key[0], key[1]}};
  okey s_init = seed_next(&ks);
  uint8_t* cur = boards;
  orc_gol_init(s_init.k, n, cur);
  okey s_outer = seed_next(&ks);
  for (int f = 0; f < n_frames; ++f) {
    uint8_t* nxt = boards + (int64_t)(f + 1) * n * n;
    memcpy(nxt, cur, (size_t)n * n);
    okey kf;
    orc_fold_in(s_outer.k, (uint32_t)f, kf.k);
    okey s_inner = seed_next(&kf);
    for (int g = 0; g < per_frame; ++g) {
      uint32_t kg[2];
      orc_fold_in(s_inner.k, (uint32_t)g, kg);
      orc_gol_sweep(nxt, target, n, p_flip, kg);
    }
    cur = nxt;
  }
}
