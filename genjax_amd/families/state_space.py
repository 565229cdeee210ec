"""Linear-Gaussian state-space family (reference src/genjax/extras/state_space.py:377-530) on
fused kernels: `linear_gaussian` keeps the reference signature
(prev_state, time_index, initial_mean, initial_cov, A, Q, C, R) and addresses "state" / "obs";
`kalman_filter` is the exact-answer companion (host, O(T d^3), no particles)."""
from __future__ import annotations

from typing import NamedTuple

import numpy as np
import torch

from .. import backend as B
from ..core import Selection, gen as _gen
from .._native import GjxError
from .base import FusedFamily, FusedTrace, MoveSpec


def _np(a):
    return a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)


class LgTrace(FusedTrace):
    """Batched trace: state / mean are SoA (d_state, n)."""

    def __init__(self, family, state, mean, obs, time_index, params):
        self.family, self.state, self.mean = family, state, mean
        self.obs = np.asarray(obs, np.float32).reshape(-1)
        self.time_index, self.params = int(time_index), params

    @property
    def n(self):
        return self.state.shape[1]

    @property
    def x(self):  # device handle used by generic helpers
        return self.state[0]

    def columns(self):
        return {"state": self.state, "mean": self.mean}

    def gather(self, ancestors, idx_base=0):
        d = self.state.shape[0]
        cols = [self.state[r] for r in range(d)] + [self.mean[r] for r in range(d)]
        out = B.gather_columns(ancestors, cols, idx_base=idx_base)
        return LgTrace(self.family, torch.stack(out[:d]), torch.stack(out[d:]), self.obs, self.time_index,
                       self.params)

    def get_choices(self):
        obs = torch.as_tensor(self.obs, device=self.state.device).expand(self.n, -1)
        return {"state": self.state.t(), "obs": obs}

    def get_retval(self):
        return LgRetval((self.state.t(), self.time_index + 1) + tuple(self.params), self)

    def get_args(self):
        return ((self.mean.t(), self.time_index) + tuple(self.params), {})


class LgRetval(tuple):
    def __new__(cls, items, trace):
        self = super().__new__(cls, items)
        self.trace = trace
        return self


class LinearGaussian(FusedFamily):
    """linear_gaussian (state_space.py:377-431)."""

    name = "linear_gaussian"
    addresses = ("state",)

    @staticmethod
    def _model(params):
        return B.make_lgssm(*[_np(p) for p in params])

    @staticmethod
    def _obs_of(constraints, dobs):
        try:
            obs = _np(constraints["obs"]).astype(np.float32).reshape(-1)
        except (KeyError, TypeError):
            raise GjxError('linear_gaussian: constraints must be {"obs": y_t}')
        if set(constraints) - {"obs"}:
            raise GjxError("linear_gaussian (fused): only the 'obs' site may be constrained")
        if obs.size != dobs:
            raise GjxError(f"linear_gaussian: observation has {obs.size} entries, d_obs={dobs}")
        return obs

    def _key(self):
        from ..core import _next_key

        return _next_key()

    def _step(self, params, t, n, prev, logw, constraints, offset):
        model = self._model(params)
        obs = self._obs_of(constraints, model.d_obs)
        state, mean, logw = B.lgssm_extend(model, self._key(), t == 0, obs, prev, logw, offset=offset)
        return LgTrace(self, state, mean, obs, t, params), logw

    def fused_init(self, args, n, constraints, offset=0):
        prev_state, t, *params = args
        t = int(_np(t).reshape(-1)[0])
        logw0 = torch.zeros(n, dtype=torch.float32, device=B._dev())
        prev = None
        if t != 0:
            prev = torch.as_tensor(_np(prev_state), dtype=torch.float32, device=B._dev()).reshape(-1, 1).expand(-1, n).contiguous()
        return self._step(params, t, n, prev, logw0, constraints, offset)

    def fused_extend(self, trace, log_weights, args, constraints, offset=0):
        n = log_weights.numel()
        if isinstance(args, LgRetval):
            prev, t, params = args.trace.state, args[1], args[2:]
        else:
            prev_state, t, *params = args
            prev = prev_state.t().contiguous() if isinstance(prev_state, torch.Tensor) and prev_state.dim() == 2 \
                else torch.as_tensor(_np(prev_state), dtype=torch.float32, device=B._dev()).reshape(-1, 1).expand(-1, n).contiguous()
        t = int(_np(t).reshape(-1)[0]) if not isinstance(t, int) else t
        return self._step(tuple(params), t, n, prev, log_weights, constraints, offset)

    def sel_mask(self, selection: Selection) -> int:
        if selection.match("obs")[0]:
            self.unsupported("regenerating the observed 'obs' site")
        return 1 if selection.match("state")[0] else 0

    def move_keys(self, spec: MoveSpec, key) -> np.ndarray:
        if spec.kind != "mh":
            self.unsupported("hmc")
        seq = B.KeySeq(key)
        out = np.zeros((2, 2), np.uint32)
        if self.sel_mask(spec.selection):
            out[0] = seq.next()
        out[1] = seq.next()
        return out

    def fused_moves(self, trace: LgTrace, spec: MoveSpec, move_keys, offset=0, accept_counts=None):
        if not self.sel_mask(spec.selection):
            return trace
        model = self._model(trace.params)
        state = B.lgssm_mh(model, move_keys, trace.time_index == 0, trace.obs, trace.state, trace.mean,
                           offset=offset, accept_counts=accept_counts)
        return LgTrace(self, state, trace.mean, trace.obs, trace.time_index, trace.params)

    def fused_smc_run(self, observations, initial_args, n, n_moves, spec):
        """The whole seeded rejuvenation_smc run in one host call with the resampling decisions taken on the
        device (gjx_lgssm_smc_run); None when this run is not of that shape (the caller then takes the
        stage-by-stage driver).  Same key schedule, same kernels, same bits as the stage-by-stage driver."""
        from ..core import _Ambient, const
        from ..inference.smc import ParticleCollection

        prev_state, t0, *params = initial_args
        if int(_np(t0).reshape(-1)[0]) != 0 or not _Ambient.keys:
            return None
        if n_moves > 0 and (spec is None or spec.kind != "mh" or not self.sel_mask(spec.selection)):
            return None
        if n_moves > B.N.MAX_MOVES:
            return None
        model = self._model(params)
        try:
            obs = _np(observations["obs"]).astype(np.float32)
        except (KeyError, TypeError):
            raise GjxError('linear_gaussian: observations must be {"obs": (T, d_obs) rows}')
        obs = obs.reshape(obs.shape[0], -1)
        if obs.shape[1] != model.d_obs:
            raise GjxError(f"linear_gaussian: observations have {obs.shape[1]} columns, d_obs={model.d_obs}")
        seq = _Ambient.keys[-1]
        r = B.lgssm_smc_run(model, seq.key, obs, n, n_moves, want_accepts=n_moves > 0)
        seq.key = r["key"]
        T = obs.shape[0]
        trace = LgTrace(self, r["state"], r["mean"], obs[-1], T - 1, tuple(params))
        p = ParticleCollection(trace, r["logw"], const(n), r["lml"].reshape(()))
        host = torch.stack([r["ess"], r["resampled"]]).cpu().numpy()      # the run's ONE read-back
        p.diagnostics = {"ess": [float(v) for v in host[0]], "resampled": [bool(v) for v in host[1]],
                         "accept_counts": [r["accept_counts"][t] for t in range(T)] if n_moves > 0 else []}
        return p

    def generate(self, x, *args, **kwargs):
        """Fused when only "obs" is constrained and the previous state is the fused layout's; the @gen twin
        otherwise (a constrained "state": the generate call behind a user-written proposal, smc.py:332-350,
        494-509)."""
        from ..core import batch_size

        x = x or {}
        prev = args[0] if args else None
        row_major = isinstance(prev, torch.Tensor) and prev.dim() == 2        # (n, d) rows of a generic trace
        if set(x) - {"obs"} or "obs" not in x or row_major:
            return self.generic.generate(x, *args, **kwargs)
        return self.fused_init(args, batch_size() or 1, x)


def _time_is_zero(time_index) -> bool:
    return int(_np(time_index).reshape(-1)[0]) == 0


def _linear_gaussian(prev_state, time_index, initial_mean, initial_cov, A, Q, C, R):
    """The model as the reference writes it (state_space.py:377-431) over columns: states are rows (n, d_state),
    so A @ x becomes x @ A^T.  time_index is shared by the particles (a host-side lax.select)."""
    from ..core import as_col
    from ..distributions import multivariate_normal

    A_, C_ = as_col(A), as_col(C)
    if _time_is_zero(time_index):
        mean, cov = as_col(initial_mean), as_col(initial_cov)
    else:
        mean, cov = as_col(prev_state) @ A_.t(), as_col(Q)
    state = multivariate_normal(mean, cov) @ "state"
    multivariate_normal(state @ C_.t(), as_col(R)) @ "obs"
    return state, time_index + 1, initial_mean, initial_cov, A, Q, C, R


linear_gaussian = LinearGaussian()
linear_gaussian.generic = _gen(_linear_gaussian)


def kalman_filter(observations, initial_mean, initial_cov, A, Q, C, R):
    """Filtered means / covariances and the exact log marginal likelihood
    (state_space.py:434-530).  Host-side float64."""
    y = _np(observations).astype(np.float64)
    mu = _np(initial_mean).astype(np.float64).reshape(-1)
    d = mu.size
    P = _np(initial_cov).astype(np.float64).reshape(d, d)
    A, Q = _np(A).astype(np.float64).reshape(d, d), _np(Q).astype(np.float64).reshape(d, d)
    C = _np(C).astype(np.float64).reshape(-1, d)
    o = C.shape[0]
    R = _np(R).astype(np.float64).reshape(o, o)
    y = y.reshape(-1, o)
    means, covs, lml = [], [], 0.0
    for t in range(y.shape[0]):
        if t > 0:
            mu, P = A @ mu, A @ P @ A.T + Q
        innovation = y[t] - C @ mu
        S = C @ P @ C.T + R
        gain = P @ C.T @ np.linalg.inv(S)
        sign, logdet = np.linalg.slogdet(S)
        lml += -0.5 * (innovation @ np.linalg.solve(S, innovation) + logdet + o * np.log(2 * np.pi))
        mu, P = mu + gain @ innovation, P - gain @ C @ P
        means.append(mu)
        covs.append(P)
    return np.stack(means), np.stack(covs), float(lml)


# --------------------------------------------------------------------------- #
# discrete HMM (reference src/genjax/extras/state_space.py:52-100, 103-163, 295-363): the kernel model on the
# generic column path -- categorical draws and log-densities are library launches, the table look-ups are
# row gathers -- with the exact forward filter as its test companion (host, f64)
# --------------------------------------------------------------------------- #
def _discrete_hmm(prev_state, time_index, initial_probs, transition_matrix, emission_matrix):
    """state ~ categorical(log initial_probs) at time 0, categorical(log transition_matrix[prev_state]) after;
    obs ~ categorical(log emission_matrix[state]).  Returns the next step's arguments."""
    from ..core import as_col
    from ..distributions import categorical

    if _time_is_zero(time_index):
        logits = torch.log(as_col(initial_probs))
    else:
        logits = torch.log(as_col(transition_matrix))[torch.as_tensor(prev_state, device=B._dev()).long()]
    state = categorical(logits) @ "state"
    categorical(torch.log(as_col(emission_matrix))[torch.as_tensor(state, device=B._dev()).long()]) @ "obs"
    return state, time_index + 1, initial_probs, transition_matrix, emission_matrix


discrete_hmm = _gen(_discrete_hmm)


def forward_filter(observations, initial_probs, transition_matrix, emission_matrix):
    """Exact filtering distributions log p(x_t | y_{0:t}) (T, K) and the log marginal likelihood
    (state_space.py:103-163), log-space recursion in float64 on the host."""
    y = _np(observations).astype(np.int64).reshape(-1)
    with np.errstate(divide="ignore"):
        log_pi = np.log(_np(initial_probs).astype(np.float64))
        log_a = np.log(_np(transition_matrix).astype(np.float64))
        log_e = np.log(_np(emission_matrix).astype(np.float64))

    def lse(a, axis):
        m = np.max(a, axis=axis, keepdims=True)
        m = np.where(np.isfinite(m), m, 0.0)
        return np.squeeze(m, axis) + np.log(np.sum(np.exp(a - m), axis=axis))

    with np.errstate(divide="ignore", invalid="ignore"):      # an impossible observation: -inf, as in the reference
        alpha = np.zeros((y.size, log_pi.size))
        alpha[0] = log_e[:, y[0]] + log_pi
        for t in range(1, y.size):
            alpha[t] = log_e[:, y[t]] + lse(alpha[t - 1][:, None] + log_a, 0)
        return alpha - lse(alpha, 1)[:, None], float(lse(alpha[-1], 0))


def _stack_sequences(who, k, draw_one):
    """K > 1 of the dataset samplers (state_space.py:345-357, 662-674): K independent sequences with a leading axis
    K.  The reference draws them under modular_vmap; here they are K passes over the ambient key stream, so the
    draws are independent and identically distributed but not the reference's bits for K > 1 (K = 1 is)."""
    if k < 1:
        raise GjxError(f"{who}: K must be >= 1 (got {k})")
    runs = [draw_one() for _ in range(k)]
    states, obs = torch.stack([r[0] for r in runs]), torch.stack([r[1] for r in runs])
    return states, obs, {"obs": obs}


def sample_hmm_dataset(initial_probs, transition_matrix, emission_matrix, T, K=None):
    """One state / observation sequence of length T drawn by iterating discrete_hmm.simulate on its own return
    value (state_space.py:295-363); -> (states (T,), observations (T,), {"obs": observations}), with a leading axis
    K when K > 1.  Runs under seed(...) like any other sampling code."""
    from ..core import Const

    steps = T.value if isinstance(T, Const) else int(T)
    k = 1 if K is None else (K.value if isinstance(K, Const) else int(K))
    if k != 1:
        return _stack_sequences("sample_hmm_dataset", k,
                                lambda: sample_hmm_dataset(initial_probs, transition_matrix, emission_matrix, steps))
    carry = (torch.zeros((), dtype=torch.int32, device=B._dev()), 0, initial_probs, transition_matrix, emission_matrix)
    states, obs = [], []
    for _ in range(steps):
        tr = discrete_hmm.simulate(*carry)
        carry = tr.get_retval()
        ch = tr.get_choices()
        states.append(ch["state"].reshape(()))
        obs.append(ch["obs"].reshape(()))
    states, obs = torch.stack(states), torch.stack(obs)
    return states, obs, {"obs": obs}


# --------------------------------------------------------------------------- #
# exact companions and dataset helpers of the two state-space models
# (reference extras/state_space.py:166-292, 533-862); host float64 unless they sample
# --------------------------------------------------------------------------- #
def kalman_smoother(observations, initial_mean, initial_cov, A, Q, C, R):
    """Rauch-Tung-Striebel smoothed means (T, d) and covariances (T, d, d) (state_space.py:533-609)."""
    means, covs, _ = kalman_filter(observations, initial_mean, initial_cov, A, Q, C, R)
    d = means.shape[1]
    A_ = _np(A).astype(np.float64).reshape(d, d)
    Q_ = _np(Q).astype(np.float64).reshape(d, d)
    sm, sc = means.copy(), covs.copy()
    for t in range(means.shape[0] - 2, -1, -1):
        pred_mean, pred_cov = A_ @ means[t], A_ @ covs[t] @ A_.T + Q_
        gain = covs[t] @ A_.T @ np.linalg.inv(pred_cov)
        sm[t] = means[t] + gain @ (sm[t + 1] - pred_mean)
        sc[t] = covs[t] + gain @ (sc[t + 1] - pred_cov) @ gain.T
    return sm, sc


def sample_linear_gaussian_dataset(initial_mean, initial_cov, A, Q, C, R, T, K=None):
    """One state / observation sequence drawn by iterating linear_gaussian.simulate on its own return value
    (state_space.py:612-679): -> (states (T, d_state), observations (T, d_obs), {"obs": observations}), with a
    leading axis K when K > 1."""
    from ..core import Const

    steps = T.value if isinstance(T, Const) else int(T)
    k = 1 if K is None else (K.value if isinstance(K, Const) else int(K))
    if k != 1:
        return _stack_sequences("sample_linear_gaussian_dataset", k, lambda: sample_linear_gaussian_dataset(
            initial_mean, initial_cov, A, Q, C, R, steps))
    d = _np(initial_mean).reshape(-1).size
    carry = (torch.zeros(d, device=B._dev()), 0, initial_mean, initial_cov, A, Q, C, R)
    states, obs = [], []
    for _ in range(steps):
        tr = linear_gaussian.generic.simulate(*carry)
        carry = tr.get_retval()
        ch = tr.get_choices()
        states.append(ch["state"].reshape(-1))
        obs.append(ch["obs"].reshape(-1))
    states, obs = torch.stack(states), torch.stack(obs)
    return states, obs, {"obs": obs}


def compute_sequence_log_prob(states, observations, initial_probs, transition_matrix, emission_matrix) -> float:
    """log p(states, observations) of a discrete HMM path (state_space.py:251-292)."""
    z, y = _np(states).astype(np.int64).reshape(-1), _np(observations).astype(np.int64).reshape(-1)
    with np.errstate(divide="ignore"):
        lp, la, le = (np.log(_np(m).astype(np.float64)) for m in (initial_probs, transition_matrix, emission_matrix))
    total = lp[z[0]] + le[z[0], y[0]]
    for t in range(1, z.size):
        total += la[z[t - 1], z[t]] + le[z[t], y[t]]
    return float(total)


def backward_sample(alpha, transition_matrix):
    """A state path from the smoothing distribution given the (log, normalised) forward filter
    (state_space.py:166-207): x_T ~ alpha_T, then x_t ~ alpha_t(.) * p(x_{t+1} | .).  Draws through this
    package's categorical (ambient key), last step first."""
    from ..distributions import categorical

    la = np.asarray(alpha, np.float64)
    with np.errstate(divide="ignore"):
        lt = np.log(_np(transition_matrix).astype(np.float64))
    dev = B._dev()
    path = [int(categorical.sample(torch.as_tensor(la[-1], dtype=torch.float32, device=dev)))]
    for t in range(la.shape[0] - 2, -1, -1):
        logits = la[t] + lt[:, path[-1]]
        path.append(int(categorical.sample(torch.as_tensor(logits, dtype=torch.float32, device=dev))))
    return torch.tensor(path[::-1], dtype=torch.int32, device=dev)


class DiscreteHMMTrace(NamedTuple):
    """The record forward_filtering_backward_sampling returns (state_space.py:44-49): the sampled path, the
    observations it was conditioned on and log p(states, observations) -- the JOINT of that path, not the marginal."""

    states: object
    observations: object
    log_prob: float


class LinearGaussianTrace(NamedTuple):
    """Same three fields for the linear-Gaussian model (state_space.py:369-374)."""

    states: object
    observations: object
    log_prob: float


def forward_filtering_backward_sampling(observations, initial_probs, transition_matrix, emission_matrix):
    """An exact posterior sample of the state path (state_space.py:210-248): forward filter, backward sample, then
    the joint log-probability of the sampled path with the observations -> DiscreteHMMTrace."""
    alpha, _ = forward_filter(observations, initial_probs, transition_matrix, emission_matrix)
    states = backward_sample(alpha, transition_matrix)
    log_prob = compute_sequence_log_prob(states, observations, initial_probs, transition_matrix, emission_matrix)
    return DiscreteHMMTrace(states=states, observations=observations, log_prob=log_prob)


def discrete_hmm_test_dataset(initial_probs, transition_matrix, emission_matrix, T) -> dict:
    states, observations, _ = sample_hmm_dataset(initial_probs, transition_matrix, emission_matrix, T)
    return {"z": states, "obs": observations}


def discrete_hmm_exact_log_marginal(observations, initial_probs, transition_matrix, emission_matrix) -> float:
    return forward_filter(observations, initial_probs, transition_matrix, emission_matrix)[1]


def linear_gaussian_test_dataset(initial_mean, initial_cov, A, Q, C, R, T) -> dict:
    states, observations, _ = sample_linear_gaussian_dataset(initial_mean, initial_cov, A, Q, C, R, T)
    return {"z": states, "obs": observations}


def linear_gaussian_exact_log_marginal(observations, initial_mean, initial_cov, A, Q, C, R) -> float:
    return kalman_filter(observations, initial_mean, initial_cov, A, Q, C, R)[2]


def discrete_hmm_inference_problem(initial_probs, transition_matrix, emission_matrix, T):
    """-> (dataset {"z", "obs"}, exact log marginal) (state_space.py:797-826)."""
    dataset = discrete_hmm_test_dataset(initial_probs, transition_matrix, emission_matrix, T)
    return dataset, discrete_hmm_exact_log_marginal(dataset["obs"], initial_probs, transition_matrix, emission_matrix)


def linear_gaussian_inference_problem(initial_mean, initial_cov, A, Q, C, R, T):
    """-> (dataset {"z", "obs"}, exact log marginal) (state_space.py:829-862)."""
    dataset = linear_gaussian_test_dataset(initial_mean, initial_cov, A, Q, C, R, T)
    return dataset, linear_gaussian_exact_log_marginal(dataset["obs"], initial_mean, initial_cov, A, Q, C, R)
