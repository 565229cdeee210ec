"""Host-side exact companions of the state-space models (reference src/genjax/extras/state_space.py:103-163,
251-292, 434-609; the reference checks them against TFP's LinearGaussianStateSpaceModel in
tests/test_linear_gaussian.py:83-150, 301-338 -- TFP is not installed here, so the joint Gaussian of the whole
sequence, built by hand, stands in): Kalman log marginal likelihood and filtered / smoothed moments against the
joint Gaussian's density and conditionals, the discrete forward filter and path log-probability against
brute-force enumeration.  No GPU needed."""
import itertools
import math

import numpy as np
import pytest
import scipy.stats as st

from genjax_amd.families import state_space as S


def _simple():      # tests/test_linear_gaussian.py:28-42
    return np.array([0.0]), np.array([[1.0]]), np.array([[0.9]]), np.array([[0.1]]), np.array([[1.0]]), np.array([[0.2]])


def _complex():     # tests/test_linear_gaussian.py:44-58
    return (np.array([0.0, 0.0]), np.eye(2), np.array([[0.9, 0.1], [-0.1, 0.8]]), 0.1 * np.eye(2),
            np.array([[1.0, 0.5]]), np.array([[0.3]]))


def _joint(params, T):
    """Mean and covariance of (x_0..x_{T-1}, y_0..y_{T-1}) stacked."""
    m0, P0, A, Q, C, R = params
    d, o = m0.size, C.shape[0]
    means, covs = [m0], [[None] * T for _ in range(T)]
    P = P0
    covs[0][0] = P0
    for t in range(1, T):
        means.append(A @ means[-1])
        P = A @ P @ A.T + Q
        covs[t][t] = P
    for s in range(T):
        for t in range(s + 1, T):
            covs[s][t] = covs[s][t - 1] @ A.T
            covs[t][s] = covs[s][t].T
    Sx = np.block(covs)
    mx = np.concatenate(means)
    Cb = np.kron(np.eye(T), C)
    Sy = Cb @ Sx @ Cb.T + np.kron(np.eye(T), R)
    return mx, Sx, Cb @ mx, Sy, Sx @ Cb.T


@pytest.mark.parametrize("params,lengths", [(_simple(), [1, 2, 3, 5]), (_complex(), [5, 10, 15])], ids=["simple", "complex"])
def test_kalman_marginal_log_prob_against_the_joint_gaussian(params, lengths):
    rng = np.random.default_rng(42)
    for T in lengths:
        y = rng.standard_normal((T, params[4].shape[0]))
        _, _, lml = S.kalman_filter(y, *params)
        _, _, my, Sy, _ = _joint(params, T)
        np.testing.assert_allclose(lml, st.multivariate_normal(my, Sy).logpdf(y.reshape(-1)), rtol=1e-9, atol=1e-9)
        assert S.linear_gaussian_exact_log_marginal(y, *params) == lml


@pytest.mark.parametrize("params", [_simple(), _complex()], ids=["simple", "complex"])
def test_filtered_and_smoothed_moments_are_the_joint_gaussians_conditionals(params):
    T = 6
    d = params[0].size
    y = np.random.default_rng(7).standard_normal((T, params[4].shape[0]))
    mx, Sx, my, Sy, Sxy = _joint(params, T)
    fm, fc, _ = S.kalman_filter(y, *params)
    sm, sc = S.kalman_smoother(y, *params)
    # smoothing: condition on all of y
    gain = Sxy @ np.linalg.inv(Sy)
    post_mean = mx + gain @ (y.reshape(-1) - my)
    post_cov = Sx - gain @ Sxy.T
    for t in range(T):
        sl = slice(t * d, (t + 1) * d)
        np.testing.assert_allclose(sm[t], post_mean[sl], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(sc[t], post_cov[sl, sl], rtol=1e-8, atol=1e-10)
    # filtering at t: condition on y_0..y_t only
    o = params[4].shape[0]
    for t in range(T):
        k = (t + 1) * o
        g = Sxy[:, :k] @ np.linalg.inv(Sy[:k, :k])
        m = mx + g @ (y.reshape(-1)[:k] - my[:k])
        c = Sx - g @ Sxy[:, :k].T
        sl = slice(t * d, (t + 1) * d)
        np.testing.assert_allclose(fm[t], m[sl], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(fc[t], c[sl, sl], rtol=1e-8, atol=1e-10)
    # the last smoothed step is the last filtered step; smoothing never widens (test_linear_gaussian.py:301-338)
    np.testing.assert_allclose(sm[-1], fm[-1])
    np.testing.assert_allclose(sc[-1], fc[-1])
    assert all(np.trace(sc[t]) <= np.trace(fc[t]) + 1e-12 for t in range(T))


def test_kalman_single_timestep_and_stability():
    """test_linear_gaussian.py:261-298, 340-373: T = 1 is the Gaussian evidence of one observation; tiny and huge
    noise scales stay finite."""
    m0, P0, A, Q, C, R = _simple()
    y = np.array([[0.7]])
    means, covs, lml = S.kalman_filter(y, m0, P0, A, Q, C, R)
    np.testing.assert_allclose(lml, st.norm(0.0, math.sqrt(1.0 + 0.2)).logpdf(0.7), rtol=1e-12)
    np.testing.assert_allclose(means[0], [0.7 * 1.0 / 1.2], rtol=1e-12)
    np.testing.assert_allclose(covs[0], [[1.0 - 1.0 / 1.2]], rtol=1e-12)
    for scale in (1e-6, 1e6):
        _, _, lml = S.kalman_filter(np.ones((4, 1)), m0, P0, A, Q * scale, C, R * scale)
        assert math.isfinite(lml)


def test_discrete_forward_filter_and_path_log_prob_against_enumeration():
    pi = np.array([0.5, 0.3, 0.2])
    A = np.array([[0.6, 0.3, 0.1], [0.2, 0.6, 0.2], [0.1, 0.4, 0.5]])
    E = np.array([[0.7, 0.2, 0.05, 0.05], [0.1, 0.6, 0.2, 0.1], [0.05, 0.1, 0.3, 0.55]])     # test_smc.py:88-110
    obs = [0, 3, 1, 2]
    total, marg = 0.0, np.zeros((len(obs), 3))
    for path in itertools.product(range(3), repeat=len(obs)):
        lp = S.compute_sequence_log_prob(path, obs, pi, A, E)
        p = pi[path[0]] * E[path[0], obs[0]]
        for t in range(1, len(obs)):
            p *= A[path[t - 1], path[t]] * E[path[t], obs[t]]
        np.testing.assert_allclose(lp, math.log(p), rtol=1e-12)
        total += p
    alpha, lml = S.forward_filter(obs, pi, A, E)
    np.testing.assert_allclose(lml, math.log(total), rtol=1e-12)
    assert S.discrete_hmm_exact_log_marginal(obs, pi, A, E) == lml
    # filtering marginals: p(x_t | y_0..t) by enumerating the prefix
    for t in range(len(obs)):
        m = np.zeros(3)
        for path in itertools.product(range(3), repeat=t + 1):
            m[path[-1]] += math.exp(S.compute_sequence_log_prob(path, obs[:t + 1], pi, A, E))
        np.testing.assert_allclose(np.exp(alpha[t]), m / m.sum(), rtol=1e-10)
    # impossible observations give -inf, not NaN
    E0 = np.array([[1.0, 0.0], [1.0, 0.0]])
    assert S.forward_filter([1], [0.5, 0.5], [[0.5, 0.5], [0.5, 0.5]], E0)[1] == -math.inf
