"""CPU tests that pin the oracle (oracle/gjx_oracle.c): third-party known answers, analytic
answers taken from the reference's own tests, SciPy log-densities, finite differences, and the
committed golden vectors."""
import json
import math
import os

import numpy as np
import pytest
from scipy import stats
from scipy.special import erfinv as sp_erfinv

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def gold(name):
    return json.load(open(os.path.join(GOLD, name)))


def from_bits(v):
    return np.asarray(v, np.uint32).view(np.float32)


# ---------------------------------------------------------------- third-party known answers
def test_threefry_random123_kats(oracle):
    for kat in gold("kat.json")["threefry2x32_20"]:
        assert oracle.threefry2x32(kat["key"], kat["ctr"]).tolist() == kat["out"]


def test_jax_documented_split_value(oracle):
    """jax.random.split(PRNGKey(0)) as printed in the JAX docs (legacy, non-partitionable layout)
    pins the block function, the key constant and jax's word order."""
    assert oracle.split_legacy(oracle.key(0), 2).tolist() == gold("kat.json")["jax_docs_legacy_split_prngkey0"]


def test_key_layout(oracle):
    assert oracle.key(42).tolist() == [0, 42]
    assert oracle.key((7 << 32) + 5).tolist() == [7, 5]
    k = oracle.key(1)
    # split_foldlike: child j = threefry(key, (0, j)); fold_in(key, d) = threefry(key, (0, d))
    assert oracle.split(k, 3)[2].tolist() == oracle.threefry2x32(k, [0, 2]).tolist()
    assert oracle.fold_in(k, 2).tolist() == oracle.threefry2x32(k, [0, 2]).tolist()


def test_uniform_and_normal_construction(oracle):
    k = oracle.key(3)
    b = oracle.random_bits(k, 1000)
    u = oracle.uniform(k, 1000)
    ref = ((b >> 9) | 0x3F800000).view(np.float32) - np.float32(1.0)
    assert np.array_equal(u, ref)
    assert u.min() >= 0.0 and u.max() < 1.0
    # 64-bit counter: element i of an offset stream equals element offset + i
    assert np.array_equal(oracle.random_bits(k, 4, offset=(1 << 32) - 2)[2:], oracle.random_bits(k, 2, offset=1 << 32))
    z = oracle.normal(k, 200000)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1.0) < 0.01
    assert abs(stats.skew(z)) < 0.03 and abs(stats.kurtosis(z)) < 0.06


def test_erfinv_against_scipy(oracle):
    xs = np.linspace(-0.9999, 0.9999, 20001).astype(np.float32)
    got = np.array([oracle.erfinv(x) for x in xs])
    ref = sp_erfinv(xs.astype(np.float64))
    assert np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-30)) < 2e-5  # Giles' f32 polynomial
    assert oracle.erfinv(1.0) == np.inf and oracle.erfinv(-1.0) == -np.inf


# ---------------------------------------------------------------- log densities (test_distributions.py:81-105)
def test_logpdfs_against_scipy(oracle):
    rng = np.random.default_rng(0)
    x = rng.normal(size=500).astype(np.float32)
    np.testing.assert_allclose(oracle.normal_logpdf(x, 0.3, 1.7), stats.norm(0.3, 1.7).logpdf(x), atol=1e-4)
    np.testing.assert_allclose(oracle.uniform_logpdf(np.array([0.5, 3.0], np.float32), -1.0, 2.0),
                               [-math.log(3.0), -np.inf], atol=1e-6)
    xb = rng.uniform(0.01, 0.99, 500).astype(np.float32)
    np.testing.assert_allclose(oracle.beta_logpdf(xb, 2.0, 5.0), stats.beta(2.0, 5.0).logpdf(xb), atol=1e-4)
    np.testing.assert_allclose(oracle.beta_logpdf(xb, 1.0, 3.0), stats.beta(1.0, 3.0).logpdf(xb), atol=1e-4)
    b = (rng.uniform(size=500) < 0.3).astype(np.float32)
    np.testing.assert_allclose(oracle.bernoulli_logpdf(b, 0.3), stats.bernoulli(0.3).logpmf(b), atol=1e-5)
    assert oracle.bernoulli_logpdf(np.array([1.0], np.float32), 1.0)[0] == 0.0  # multiply_no_nan
    logits = rng.normal(size=(500, 4)).astype(np.float32)
    xi = rng.integers(0, 4, 500).astype(np.int32)
    ref = logits[np.arange(500), xi] - np.log(np.exp(logits.astype(np.float64)).sum(1))
    np.testing.assert_allclose(oracle.categorical_logpdf(xi, logits), ref, atol=1e-5)
    A = rng.normal(size=(3, 3))
    cov = (A @ A.T + np.eye(3)).astype(np.float32)
    mu = rng.normal(size=3).astype(np.float32)
    xv = rng.normal(size=(500, 3)).astype(np.float32)
    np.testing.assert_allclose(oracle.mvnormal_logpdf(xv, mu, cov),
                               stats.multivariate_normal(mu, cov).logpdf(xv), atol=2e-4)


# ---------------------------------------------------------------- normaliser / resampling
def test_logsumexp_ess(oracle):
    rng = np.random.default_rng(1)
    a = (rng.normal(size=10000) * 20 - 300).astype(np.float32)
    from scipy.special import logsumexp

    np.testing.assert_allclose(oracle.logsumexp(a), logsumexp(a.astype(np.float64)), rtol=1e-6)
    w = np.exp(a.astype(np.float64) - logsumexp(a.astype(np.float64)))
    np.testing.assert_allclose(oracle.ess(a), 1.0 / np.sum(w * w), rtol=1e-4)
    assert oracle.ess(np.zeros(128, np.float32)) == pytest.approx(128.0)
    assert oracle.logsumexp(np.full(5, -np.inf, np.float32)) == -np.inf


def _assoc_scan_reference(e):
    """lax.associative_scan(add) written directly from its published recursion."""
    e = list(np.asarray(e, np.float32))
    n = len(e)
    if n < 2:
        return e
    red = [np.float32(e[2 * i] + e[2 * i + 1]) for i in range(n // 2)]
    odd = _assoc_scan_reference(red)
    out = [None] * n
    out[0] = e[0]
    for i in range(n // 2):
        out[2 * i + 1] = odd[i]
    for i in range((n - 1) // 2):
        out[2 * i + 2] = np.float32(odd[i] + e[2 * i + 2])
    return out


@pytest.mark.parametrize("n", [1, 2, 3, 7, 8, 37, 100, 1000])
def test_cumsum_association(oracle, n):
    rng = np.random.default_rng(n)
    e = rng.exponential(size=n).astype(np.float32)
    got = oracle.cumsum(e)
    ref = np.array(_assoc_scan_reference(e), np.float32)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    np.testing.assert_allclose(got, np.cumsum(e.astype(np.float64)), rtol=1e-5)


def test_cumsum_dyadic_fold_property(oracle):
    """prefix(k) = fold-left over the dyadic blocks of k+1 (MSB first), each a balanced tree."""
    rng = np.random.default_rng(7)
    e = rng.exponential(size=64).astype(np.float32)

    def tree(lo, n):
        return e[lo] if n == 1 else np.float32(tree(lo, n // 2) + tree(lo + n // 2, n // 2))

    got = oracle.cumsum(e)
    for k in range(64):
        cnt, pos, acc = k + 1, 0, None
        for b in range(6, -1, -1):
            if cnt & (1 << b):
                blk = tree(pos, 1 << b)
                acc = blk if acc is None else np.float32(acc + blk)
                pos += 1 << b
        assert got[k] == acc


def test_searchsorted_and_systematic(oracle):
    rng = np.random.default_rng(2)
    a = np.sort(rng.uniform(size=1000)).astype(np.float32)
    v = rng.uniform(size=500).astype(np.float32)
    assert np.array_equal(oracle.searchsorted_left(a, v), np.searchsorted(a, v, side="left"))
    w = rng.exponential(size=4096).astype(np.float32)
    w /= w.sum()
    idx = oracle.systematic_indices(w, 0.5)
    counts = np.bincount(idx, minlength=4096)
    assert np.all(np.abs(counts - 4096 * w) <= 1.0 + 1e-3)        # systematic: |N_i - N w_i| < 1
    assert np.all(np.diff(idx) >= 0)
    # all mass on one particle
    w = np.zeros(100, np.float32)
    w[37] = 1.0
    assert np.all(oracle.systematic_indices(w, 0.3) == 37)


def test_categorical_indices_distribution(oracle):
    logw = np.log(np.array([0.1, 0.2, 0.3, 0.4], np.float32))
    idx = oracle.categorical_indices(oracle.key(5), logw, m=40000)
    freq = np.bincount(idx, minlength=4) / 40000
    np.testing.assert_allclose(freq, [0.1, 0.2, 0.3, 0.4], atol=0.01)


# ---------------------------------------------------------------- localization
def test_lidar_geometry(oracle):
    w = oracle.basic_world()
    d = oracle.lidar([2.0], [5.0], [0.0], w, 8)[0]
    # from (2, 5) heading 0: east ray passes the doorway at x=4 (y in [3,5]: 5 is on its edge) ...
    assert d[2] == pytest.approx(5.0, abs=1e-5)      # north: boundary y = 10
    assert d[4] == pytest.approx(2.0, abs=1e-5)      # west: boundary x = 0
    assert d[6] == pytest.approx(5.0, abs=1e-5)      # south: boundary y = 0
    d = oracle.lidar([2.0], [1.0], [0.0], w, 8)[0]
    assert d[0] == pytest.approx(2.0, abs=1e-5)      # east: wall x = 4 (y in [0, 3])
    # reference quirk: a ray collinear with a wall has its denominator clamped to 1e-10 and
    # reports distance 0 (core.py:114-128): from (2, 2) the east ray runs along the obstacle y = 2
    assert oracle.lidar([2.0], [2.0], [0.0], w, 8)[0][0] == 0.0
    d = oracle.lidar([6.0], [1.0], [0.0], w, 4)[0]
    assert d[0] == pytest.approx(2.0, abs=1e-5) and d[2] == pytest.approx(2.0, abs=1e-5)
    assert np.all(oracle.lidar([6.0], [8.0], [0.3], w, 16) <= 10.0)


def test_gradient_finite_differences(oracle):
    w = oracle.complex_world()
    obs, k = oracle.localization_observations(w)
    n = 200
    st = oracle.LocState(n, 8)
    oracle.loc_extend(st, oracle.split(k, 3), True, obs[0], w)
    logp, grad = oracle.loc_logp_grad(st, True, obs[0], w)
    h = 1e-3
    ok = 0
    for j, name in enumerate(("x", "y", "th")):
        plus, minus = st.copy(), st.copy()
        getattr(plus, name)[:] += np.float32(h)
        getattr(minus, name)[:] -= np.float32(h)
        fp, _ = oracle.loc_logp_grad(plus, True, obs[0], w)
        fm, _ = oracle.loc_logp_grad(minus, True, obs[0], w)
        fd = (fp.astype(np.float64) - fm) / (2 * h)
        good = np.isclose(grad[:, j], fd, rtol=5e-2, atol=0.5)
        ok += good.mean()
    assert ok / 3 > 0.9          # a ray that changes its winning wall inside +-h breaks the FD


def test_mh_and_rejuvenate_invariants(oracle):
    """rejuvenate leaves weights untouched (test_smc.py:970-995); same key => same chain
    (test_mcmc.py:760-784)."""
    w = oracle.basic_world()
    obs, k = oracle.localization_observations(w)
    st = oracle.LocState(2000, 8)
    oracle.loc_extend(st, oracle.split(k, 3), True, obs[0], w)
    logw0 = st.logw.copy()
    a, b = st.copy(), st.copy()
    mk = oracle.split(oracle.fold_in(k, 9), 4)
    na, _ = oracle.loc_mh(a, mk, True, obs[0], w)
    nb, _ = oracle.loc_mh(b, mk, True, obs[0], w)
    assert na == nb and np.array_equal(a.x, b.x)
    assert np.array_equal(a.logw, logw0)
    assert 0 < na < 2000


def test_smc_resample_invariants(oracle):
    """resample zeroes the weights and preserves log_marginal_likelihood (test_smc.py:777-805)."""
    w = oracle.basic_world()
    obs, k = oracle.localization_observations(w)
    r1 = oracle.loc_smc(k, 512, obs[:1], w, kernel="none")
    assert r1["resampled"][0] == 1 and np.all(r1["state"].logw == 0.0)
    st = oracle.LocState(512, 8)
    oracle.loc_extend(st, [oracle.Seed(k).next()] * 0 or _first3(oracle, k), True, obs[0], w)
    lml_before = oracle.logsumexp(st.logw) - math.log(512)
    assert r1["lml"] == pytest.approx(lml_before, rel=1e-6)


def _first3(oracle, k):
    s = oracle.Seed(k)
    return [s.next() for _ in range(3)]


def test_observations_match_reference_shape(oracle):
    obs, k = oracle.localization_observations(oracle.basic_world())
    assert obs.shape == (17, 8) and np.all(obs >= 0)


# ---------------------------------------------------------------- golden vectors (regression pins)
def test_golden_prng(oracle):
    g = gold("prng.json")
    k = oracle.key(42)
    assert oracle.split(k, 3).tolist() == g["split42_3"]
    assert oracle.fold_in(k, 7).tolist() == g["fold_in42_7"]
    assert oracle.random_bits(k, 8).tolist() == g["bits42_8"]
    assert oracle.random_bits(k, 4, (1 << 32) - 2).tolist() == g["bits42_offset_2p32m2_4"]
    assert np.array_equal(oracle.uniform(k, 8), from_bits(g["uniform42_8_bits"]))
    np.testing.assert_allclose(oracle.normal(k, 8), g["normal42_8"], rtol=1e-6)


def test_golden_localization(oracle):
    g = gold("localization16.json")
    w = oracle.basic_world()
    obs, k = oracle.localization_observations(w)
    assert k.tolist() == g["smc_key"]
    np.testing.assert_array_equal(obs, np.array(g["obs"], np.float32))
    st = oracle.LocState(16, 8)
    oracle.loc_extend(st, np.array(g["keys_init"], np.uint32), True, obs[0], w)
    np.testing.assert_allclose(st.logw, g["init"]["logw"], rtol=1e-6)
    oracle.loc_extend(st, np.array(g["keys_step"], np.uint32), False, obs[1], w)
    np.testing.assert_allclose(st.logw, g["step"]["logw"], rtol=1e-6)
    wts = from_bits(g["weights_bits"])
    anc, cdf = oracle.systematic_indices(wts, g["u"], return_cdf=True)
    assert anc.tolist() == g["ancestors"]
    assert np.array_equal(cdf, from_bits(g["cdf_bits"]))


def test_golden_cumsum(oracle):
    g = gold("cumsum37.json")
    assert np.array_equal(oracle.cumsum(from_bits(g["in_bits"])), from_bits(g["out_bits"]))


# ------------------------------------------------------------------ config 3: outlier mixture
def test_outlier_model_gradient_and_nan_semantics(oracle):
    """assess() gradient of npoint_curve_with_outliers_beta (curvefit/core.py:468-552) against
    finite differences; outside the Beta support the density is -inf / NaN and the move must
    reject (jnp.minimum propagates NaN, mcmc.py:563-566)."""
    xs, ys, idx = oracle.outlier_dataset()
    m = oracle.outlier_model(xs, ys, alpha=1.0, beta=3.0)
    mask = sum(1 << j for j in idx)
    q = np.array([-0.2, -0.4, 0.6, 0.3], np.float32)
    lp, g = oracle.outlier_logp_grad(m, q, mask)
    for k in range(4):
        d = np.zeros(4, np.float32)
        d[k] = 1e-3
        fd = (oracle.outlier_logp_grad(m, q + d, mask)[0] - oracle.outlier_logp_grad(m, q - d, mask)[0]) / 2e-3
        assert abs(g[k] - fd) <= 2e-3 * max(1.0, abs(fd)), (k, g[k], fd)
    # closed form of the log density at this point (SciPy)
    from scipy import stats

    mean = q[0] + q[1] * xs + q[2] * xs * xs
    o = np.array([(mask >> j) & 1 for j in range(18)])
    ref = stats.beta.logpdf(0.3, 1, 3) + stats.norm.logpdf(q[:3]).sum()
    ref += np.sum(np.where(o == 1, np.log(0.3) - np.log(4.0), np.log(0.7) + stats.norm.logpdf(ys, mean, 0.1)))
    assert abs(lp - ref) <= 1e-4 * abs(ref)
    assert not np.isfinite(oracle.outlier_logp_grad(m, np.array([0, 0, 0, 1.5], np.float32), mask)[0])


def test_outlier_gibbs_hmc_recovers_outliers(oracle):
    """Gibbs + HMC chains (figs.py:907-950) on the reference's 18-point data set find the five
    planted outliers; the rate never leaves (0, 1); same key => same chains."""
    xs, ys, idx = oracle.outlier_dataset()
    m = oracle.outlier_model(xs, ys, alpha=1.0, beta=3.0)
    n = 512
    st = oracle.outlier_init(m, oracle.key(12345), n)
    assert np.array_equal(st["keys"], oracle.split(oracle.key(12345), n))
    # sharded init == full init (chain keys depend on the global chain index only)
    part = oracle.outlier_init(m, oracle.key(12345), 100, offset=200)
    for k_ in st:
        assert np.array_equal(part[k_], st[k_][200:300])
    st2 = {k_: v.copy() for k_, v in st.items()}
    acc = oracle.outlier_gibbs_hmc(m, st, 300, 0.01, 20)
    acc2 = oracle.outlier_gibbs_hmc(m, st2, 300, 0.01, 20)
    assert acc == acc2 and all(np.array_equal(st[k_], st2[k_]) for k_ in st)
    assert 0.85 < acc / (n * 300) <= 1.0
    probs = np.array([((st["outliers"] >> np.uint64(j)) & np.uint64(1)).mean() for j in range(18)])
    assert np.all(probs[idx] > 0.9) and np.all(np.delete(probs, idx) < 0.3)
    assert st["rate"].min() > 0.0 and st["rate"].max() < 1.0


# ------------------------------------------------------------------ config 5: Game-of-Life Gibbs
def test_gol_rule_and_gibbs_conditional(oracle):
    """Rule of Life known answers (blinker, block) and the single-cell conditional
    (gol/core.py:108-163) against brute-force enumeration of the joint of the 9 next cells."""
    b = np.zeros((5, 5), np.uint8)
    b[2, 1:4] = 1
    assert np.array_equal(oracle.gol_step(b), b.T)                     # blinker oscillates
    blk = np.zeros((4, 4), np.uint8)
    blk[1:3, 1:3] = 1
    assert np.array_equal(oracle.gol_step(blk), blk)                   # block is a still life
    rng = np.random.default_rng(5)
    n, f = 7, 0.03
    cur = (rng.random((n, n)) < 0.4).astype(np.uint8)
    fut = (rng.random((n, n)) < 0.4).astype(np.uint8)

    def loglik(board):
        nxt = oracle.gol_step(board)
        return np.where(nxt == fut, np.log(1 - f), np.log(f)).sum()

    # a pass over colour (oy, ox) with u ~ key: cell set to 1 iff u < P(1 | rest)
    for (i, j) in ((0, 0), (3, 3), (6, 6), (0, 3)):
        c0, c1 = cur.copy(), cur.copy()
        c0[i, j], c1[i, j] = 0, 1
        l0, l1 = loglik(c0), loglik(c1)
        p_one = 1.0 / (1.0 + np.exp(l0 - l1))
        hits = 0
        trials = 400
        for t in range(trials):
            c = cur.copy()
            oracle.gol_gibbs_pass(c, fut, i % 3, j % 3, f, oracle.fold_in(oracle.key(3), t))
            hits += int(c[i, j])
        assert abs(hits / trials - p_one) < 4 * np.sqrt(p_one * (1 - p_one) / trials) + 0.02, (i, j, p_one)


def test_gol_run_reduces_reconstruction_error(oracle):
    rng = np.random.default_rng(0)
    n = 24
    target = oracle.gol_step((rng.random((n, n)) < 0.3).astype(np.uint8))
    boards = oracle.gol_run(oracle.key(1), target, 0.03, 6, 4)
    errs = [oracle.gol_summary(b, target, 0.03)[0] for b in boards]
    assert errs[-1] < 0.35 * errs[0]
    assert np.array_equal(boards, oracle.gol_run(oracle.key(1), target, 0.03, 6, 4))   # same key, same run
    assert np.array_equal(boards[0], oracle.gol_init(oracle.Seed(oracle.key(1)).next(), n))


def test_golden_outlier_and_gol(oracle):
    """Regression vectors of configs 3 and 5 (outputs of the restatement, see make_golden.py)."""
    g = gold("outlier8.json")
    xs, ys, idx = oracle.outlier_dataset()
    assert np.array_equal(ys, from_bits(g["ys_bits"])) and idx == g["outlier_idx"]
    m = oracle.outlier_model(xs, ys, alpha=1.0, beta=3.0)
    st = oracle.outlier_init(m, oracle.key(12345), 8)
    assert st["keys"].tolist() == g["init"]["keys"] and [int(v) for v in st["outliers"]] == g["init"]["outliers"]
    np.testing.assert_allclose(st["rate"], g["init"]["rate"], rtol=1e-6)
    assert oracle.outlier_gibbs_hmc(m, st, 3, 0.01, 20) == g["accepted"]
    assert st["keys"].tolist() == g["after3"]["keys"]
    assert [int(v) for v in st["outliers"]] == g["after3"]["outliers"]
    for k_ in ("a", "b", "c", "rate"):
        np.testing.assert_allclose(st[k_], g["after3"][k_], rtol=1e-5)
    g = gold("gol12.json")
    target = np.array(g["target"], np.uint8)
    boards = oracle.gol_run(oracle.key(1), target, 0.03, 2, 2)
    assert np.array_equal(boards, np.array(g["boards"], np.uint8))
    assert [oracle.gol_summary(b, target, 0.03)[0] for b in boards] == g["n_incorrect"]


def test_reference_fixture_if_present(oracle):
    """tests/golden/reference_jax.json is written by benchmarks/reference_jax_cpu.py on a machine that has
    the reference's JAX / TFP environment (this container and the GPU boxes do not).  When
    somebody commits it, the restatement is pinned against the reference itself: integer streams bit for
    bit, floats to the tolerance north_star states (1e-5 relative)."""
    import json
    import os

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_jax.json")
    if not os.path.exists(path):
        pytest.skip("no reference-generated fixture committed yet (parity with the reference itself is unpinned)")
    g = json.load(open(path))
    k = oracle.key(42)
    assert [int(v) for v in k] == g["key42"]
    assert oracle.split(k, 3).tolist() == g["split42_3"]
    assert [int(v) for v in oracle.fold_in(k, 7)] == g["fold_in42_7"]
    assert [int(v) for v in oracle.random_bits(k, 8)] == g["bits42_8"]
    as_bits = lambda a: np.asarray(a, np.float32).view(np.uint32).astype(np.int64)  # noqa: E731
    assert as_bits(oracle.uniform(k, 8)).tolist() == g["uniform42_8_bits"]
    assert np.abs(as_bits(oracle.normal(k, 8)) - np.asarray(g["normal42_8_bits"], np.int64)).max() <= 2
    assert as_bits(oracle.cumsum(oracle.uniform(oracle.key(7), 37))).tolist() == g["cumsum37_bits"]
    w = oracle.basic_world()
    obs, smc_key = oracle.localization_observations(w)
    assert as_bits(obs).tolist() == g["observations_bits"]
    loc = g["localization16"]
    r = oracle.loc_smc(smc_key, loc["n"], obs[:loc["stages"]], w, kernel="mh", K=loc["moves"], method="systematic")
    as_f32 = lambda b: np.asarray(b, np.uint32).view(np.float32)  # noqa: E731
    for name, col in (("x_bits", r["state"].x), ("y_bits", r["state"].y), ("theta_bits", r["state"].th)):
        np.testing.assert_allclose(col, as_f32(loc[name]), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(r["state"].logw, np.asarray(loc["log_weights"], np.float32), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(r["lml"], loc["log_ml"], rtol=1e-5)


def test_stage_stepper_equals_the_c_driver_bitwise(oracle):
    """oracle.LocSMCStepper (stages composed in Python, used by the teacher-forced GPU checks) ==
    orc_loc_smc (one C call) on the same key: ESS trace, decisions, accept counts, final columns."""
    w = oracle.basic_world()
    obs, k = oracle.localization_observations(w)
    for kernel, K in (("mh", 3), ("hmc", 2)):
        n = 512
        ref = oracle.loc_smc(k, n, obs[:6], w, kernel=kernel, K=K, method="systematic")
        sp = oracle.LocSMCStepper(k, n, obs[:6], w, kernel=kernel, K=K)
        outs = [sp.stage(t) for t in range(6)]
        assert np.array_equal(np.float32([o["ess"] for o in outs]), ref["ess"])
        assert [o["resampled"] for o in outs] == [bool(r) for r in ref["resampled"]]
        assert np.array_equal(np.array([o["accept"] for o in outs]), ref["accept"])
        for name in ("x", "y", "th", "mx", "my", "mth", "logw"):
            assert np.array_equal(getattr(sp.st, name), getattr(ref["state"], name)), name
        assert sp.log_marginal_likelihood() == ref["lml"]


# ----------------------------------------------------------------------------------------------
# Fixtures the REFERENCE's own tests hold for this path (analytic answers), applied to the oracle's
# and to the product's host-side Kalman filter.
# ----------------------------------------------------------------------------------------------
def _kalman_impls(oracle):
    from genjax_amd.families.state_space import kalman_filter as product_kalman   # host-side float64, no GPU

    return (("oracle", oracle.kalman_filter), ("product", product_kalman))


def test_reference_fixture_kalman_single_step_closed_form(oracle):
    """reference tests/test_smc.py:1539-1596: T = 1, prior N(0, 1), y = 1.5 ~ N(x, 0.5): posterior and
    log marginal in closed form (Bayesian linear regression), atol 1e-5 there, 1e-9 here (float64)."""
    from scipy.stats import multivariate_normal

    mu0, P0, A, Q, C, R = [0.0], [[1.0]], [[1.0]], [[0.1]], [[1.0]], [[0.5]]
    y = np.array([[1.5]])
    prec_post = 1.0 / 0.5 + 1.0 / 1.0
    mean_post = (1.5 / 0.5 + 0.0) / prec_post
    lml = multivariate_normal.logpdf(y[0], [0.0], [[1.0 + 0.5]])
    for name, kf in _kalman_impls(oracle):
        means, covs, got = kf(y, mu0, P0, A, Q, C, R)
        assert abs(means[0, 0] - mean_post) < 1e-9, name
        assert abs(covs[0, 0, 0] - 1.0 / prec_post) < 1e-9, name
        assert abs(got - lml) < 1e-9, name


def test_reference_fixture_kalman_two_step_hand_computation(oracle):
    """reference tests/test_smc.py:1721-1778: random walk, unit noises, y = (1, 2): filtered means 0.5, 1.4
    and variances 0.5, 0.6 (hand computation; atol 1e-6 there)."""
    for name, kf in _kalman_impls(oracle):
        means, covs, _ = kf(np.array([[1.0], [2.0]]), [0.0], [[1.0]], [[1.0]], [[1.0]], [[1.0]], [[1.0]])
        assert np.allclose(means[:, 0], [0.5, 1.4], atol=1e-12), name
        assert np.allclose(covs[:, 0, 0], [0.5, 0.6], atol=1e-12), name


def test_reference_fixture_conjugate_posterior_moments(oracle):
    """reference tests/test_mcmc.py:118-163: the exact Beta-Bernoulli and Normal-Normal posterior moments the GPU
    chains are compared with (tests/test_gpu_surface.py).  The closed forms of tests_fixtures are checked against the
    MODEL, not against themselves: prior density x likelihood from the restatement's own log-densities (the ones the
    kernels are held to), integrated numerically, must have those moments -- and the reference's quoted numbers."""
    from scipy.integrate import quad

    from tests_fixtures import beta_bernoulli_posterior, normal_normal_posterior

    def moments(unnorm, lo, hi):
        z = quad(unnorm, lo, hi, epsabs=1e-9, epsrel=1e-8, limit=200)[0]
        m1 = quad(lambda t: t * unnorm(t), lo, hi, epsabs=1e-9, epsrel=1e-8, limit=200)[0] / z
        m2 = quad(lambda t: t * t * unnorm(t), lo, hi, epsabs=1e-9, epsrel=1e-8, limit=200)[0] / z
        return m1, m2 - m1 * m1

    f32 = lambda v: np.asarray([v], np.float32)                              # noqa: E731
    for obs in (True, False):
        def post(p, obs=obs):
            lp = float(oracle.beta_logpdf(f32(p), f32(2.0), f32(5.0))[0])
            lp += float(oracle.bernoulli_logpdf(f32(1.0 if obs else 0.0), f32(p))[0])
            return math.exp(lp)
        mean, var = moments(post, 1e-6, 1.0 - 1e-6)
        want = beta_bernoulli_posterior(obs)
        assert abs(mean - want[0]) < 2e-6 and abs(var - want[1]) < 2e-6, (obs, mean, var, want)   # f32 log-densities
    assert np.allclose(beta_bernoulli_posterior(True)[:2], (3.0 / 8.0, 15.0 / (64.0 * 9.0)))       # test_mcmc.py:1356-1360

    def post_mu(mu):
        return math.exp(float(oracle.normal_logpdf(f32(mu), f32(0.0), f32(1.0))[0])
                        + float(oracle.normal_logpdf(f32(1.5), f32(mu), f32(0.5))[0]))
    mean, var = moments(post_mu, -6.0, 8.0)
    want = normal_normal_posterior(1.5)
    assert abs(mean - want[0]) < 2e-6 and abs(var - want[1]) < 2e-6, (mean, var, want)
    assert np.allclose(want, (1.2, 0.2))   # prior N(0, 1), y ~ N(mu, 0.25): 1.5 * 4 / 5, 1 / 5
