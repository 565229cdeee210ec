"""ctypes face of the CPU restatement (oracle/gjx_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libgjx_oracle.so")
_SO_FMA = os.path.join(_HERE, "_build", "libgjx_oracle_fma.so")

u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    """Compile the restatement with gcc (see oracle/Makefile)."""
    src = os.path.join(_HERE, "gjx_oracle.c")
    stale = lambda so: not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src)      # noqa: E731
    if force or stale(_SO) or stale(_SO_FMA):
        subprocess.check_call(["make", "-C", _HERE, "-s"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None
_lib_fma = None


class contracted:
    """with oracle.contracted(): every oracle call inside runs the SAME restatement compiled with fused
    multiply-add contraction (oracle/Makefile, second build).  For justifying a difference only: a quantity the
    strict build holds outside the bar, and this build inside it, differs by the rounding of a * b + c alone."""

    def __enter__(self):
        global _lib, _lib_fma
        if _lib_fma is None:
            if not os.path.exists(_SO_FMA):
                build()
            _lib_fma = C.CDLL(_SO_FMA)
            _declare(_lib_fma)
        self._saved, _lib = lib(), _lib_fma
        return self

    def __exit__(self, *exc):
        global _lib
        _lib = self._saved


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _declare(_lib)
    return _lib


def _declare(L):
    L.orc_loc_ray_logpdfs_n.argtypes = [f32p, f32p, f32p, C.c_int64, f32p, C.c_int32, C.c_float, C.c_float,
                                        C.c_int32, f32p, f32p, f32p, f32p, f32p]
    L.orc_threefry2x32.argtypes = [u32p, u32p, u32p]
    L.orc_key.argtypes = [C.c_uint64, u32p]
    L.orc_split.argtypes = [u32p, C.c_uint32, u32p]
    L.orc_split_legacy.argtypes = [u32p, C.c_uint32, u32p]
    L.orc_fold_in.argtypes = [u32p, C.c_uint32, u32p]
    L.orc_random_bits.argtypes = [u32p, C.c_uint64, C.c_uint64, u32p]
    L.orc_uniform.argtypes = [u32p, C.c_uint64, C.c_uint64, C.c_float, C.c_float, f32p]
    L.orc_normal.argtypes = [u32p, C.c_uint64, C.c_uint64, f32p]
    L.orc_erfinv.argtypes = [C.c_float]
    L.orc_erfinv.restype = C.c_float
    L.orc_normal_logpdf.argtypes = [f32p, f32p, f32p, C.c_int64, f32p]
    L.orc_uniform_logpdf.argtypes = [f32p, f32p, f32p, C.c_int64, f32p]
    L.orc_bernoulli_logpdf.argtypes = [f32p, f32p, C.c_int64, f32p]
    L.orc_categorical_logpdf.argtypes = [i32p, f32p, C.c_int64, C.c_int32, f32p]
    L.orc_beta_logpdf.argtypes = [f32p, f32p, f32p, C.c_int64, f32p]
    L.orc_mvnormal_logpdf.argtypes = [f32p, f32p, f32p, C.c_int64, C.c_int32, f32p]
    L.orc_mvnormal_logpdf.restype = C.c_int
    L.orc_logsumexp.argtypes = [f32p, C.c_int64]
    L.orc_logsumexp.restype = C.c_float
    L.orc_ess.argtypes = [f32p, C.c_int64]
    L.orc_ess.restype = C.c_float
    L.orc_cumsum.argtypes = [f32p, C.c_int64, f32p]
    L.orc_searchsorted_left.argtypes = [f32p, C.c_int64, f32p, C.c_int64, i32p]
    L.orc_systematic_indices.argtypes = [f32p, C.c_int64, C.c_float, i32p, C.c_void_p]
    L.orc_normalized_weights.argtypes = [f32p, C.c_int64, f32p]
    L.orc_categorical_indices.argtypes = [u32p, f32p, C.c_int64, C.c_int64, i32p]
    world = [C.c_float, C.c_float, C.c_int32, f32p, f32p, f32p, f32p]
    L.orc_lidar.argtypes = [f32p, f32p, f32p, C.c_int64] + world + [C.c_int32, f32p]
    L.orc_loc_extend.argtypes = (
        [u32p, u32p, u32p, C.c_int32, C.c_int64, C.c_uint64, f32p, C.c_int32]
        + world
        + [f32p] * 8
    )
    L.orc_loc_mh.argtypes = (
        [u32p] * 4
        + [C.c_int32, C.c_int64, C.c_uint64, f32p, C.c_int32]
        + world
        + [f32p] * 7
        + [C.c_void_p]
    )
    L.orc_loc_mh.restype = C.c_int64
    L.orc_loc_mh_diag.argtypes = L.orc_loc_mh.argtypes + [f32p]
    L.orc_loc_mh_diag.restype = C.c_int64
    L.orc_loc_logp_grad.argtypes = (
        [f32p] * 6 + [C.c_int32, C.c_int64, f32p, C.c_int32] + world + [f32p, f32p]
    )
    L.orc_loc_hmc.argtypes = (
        [u32p] * 4
        + [C.c_float, C.c_int32, C.c_int32, C.c_int64, C.c_uint64, f32p, C.c_int32]
        + world
        + [f32p] * 7
        + [C.c_void_p]
    )
    L.orc_loc_hmc.restype = C.c_int64
    L.orc_loc_hmc_diag.argtypes = L.orc_loc_hmc.argtypes + [f32p]
    L.orc_loc_hmc_diag.restype = C.c_int64
    L.orc_loc_smc.argtypes = (
        [u32p, C.c_int64, C.c_int32, f32p, C.c_int32]
        + world
        + [C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_int32]
        + [f32p] * 7
        + [f32p, f32p, i32p, C.c_void_p, f32p]
    )
    L.orc_loc_smc.restype = C.c_int
    if hasattr(L, "orc_set_threads"):
        L.orc_set_threads.argtypes = [C.c_int]


# --------------------------------------------------------------------------- #
# PRNG
# --------------------------------------------------------------------------- #
def _k(key):
    return np.ascontiguousarray(np.asarray(key, dtype=np.uint32).reshape(2))


def threefry2x32(key, ctr):
    out = np.zeros(2, np.uint32)
    lib().orc_threefry2x32(_k(key), _k(ctr), out)
    return out


def key(seed: int):
    out = np.zeros(2, np.uint32)
    lib().orc_key(int(seed), out)
    return out


def split(k, n: int = 2):
    out = np.zeros((n, 2), np.uint32)
    lib().orc_split(_k(k), n, out.reshape(-1))
    return out


def split_legacy(k, n: int = 2):
    out = np.zeros((n, 2), np.uint32)
    lib().orc_split_legacy(_k(k), n, out.reshape(-1))
    return out


def fold_in(k, data: int):
    out = np.zeros(2, np.uint32)
    lib().orc_fold_in(_k(k), int(data), out)
    return out


def random_bits(k, n: int, offset: int = 0):
    out = np.zeros(n, np.uint32)
    lib().orc_random_bits(_k(k), offset, n, out)
    return out


def uniform(k, n: int, minval=0.0, maxval=1.0, offset: int = 0):
    out = np.zeros(n, np.float32)
    lib().orc_uniform(_k(k), offset, n, minval, maxval, out)
    return out


def normal(k, n: int, offset: int = 0):
    out = np.zeros(n, np.float32)
    lib().orc_normal(_k(k), offset, n, out)
    return out


def erfinv(x: float) -> float:
    return float(lib().orc_erfinv(float(x)))


class Seed:
    """The Seed interpreter's key state (pjax.py:1316,1337): next() = one split."""

    def __init__(self, k):
        self.key = _k(k).copy()

    def next(self):
        ks = split(self.key, 2)
        self.key = ks[0].copy()
        return ks[1].copy()


# --------------------------------------------------------------------------- #
# log densities
# --------------------------------------------------------------------------- #
def _f(a, n=None):
    a = np.asarray(a, dtype=np.float32)
    if n is not None:
        a = np.broadcast_to(a, (n,))
    return np.ascontiguousarray(a)


def normal_logpdf(x, mu, sigma):
    x = _f(x)
    n = x.size
    out = np.zeros(n, np.float32)
    lib().orc_normal_logpdf(x.reshape(-1), _f(mu, n), _f(sigma, n), n, out)
    return out


def uniform_logpdf(x, lo, hi):
    x = _f(x)
    n = x.size
    out = np.zeros(n, np.float32)
    lib().orc_uniform_logpdf(x.reshape(-1), _f(lo, n), _f(hi, n), n, out)
    return out


def bernoulli_logpdf(x, p):
    x = _f(x)
    n = x.size
    out = np.zeros(n, np.float32)
    lib().orc_bernoulli_logpdf(x.reshape(-1), _f(p, n), n, out)
    return out


def categorical_logpdf(x, logits):
    logits = np.ascontiguousarray(np.asarray(logits, np.float32))
    n, k = logits.shape
    out = np.zeros(n, np.float32)
    lib().orc_categorical_logpdf(np.ascontiguousarray(np.asarray(x, np.int32)), logits.reshape(-1), n, k, out)
    return out


def beta_logpdf(x, a, b):
    x = _f(x)
    n = x.size
    out = np.zeros(n, np.float32)
    lib().orc_beta_logpdf(x.reshape(-1), _f(a, n), _f(b, n), n, out)
    return out


def mvnormal_logpdf(x, mu, cov):
    x = np.ascontiguousarray(np.asarray(x, np.float32))
    n, d = x.shape
    mu = np.ascontiguousarray(np.broadcast_to(np.asarray(mu, np.float32), (n, d)))
    cov = np.ascontiguousarray(np.broadcast_to(np.asarray(cov, np.float32), (n, d, d)))
    out = np.zeros(n, np.float32)
    rc = lib().orc_mvnormal_logpdf(x.reshape(-1), mu.reshape(-1), cov.reshape(-1), n, d, out)
    assert rc == 0
    return out


# --------------------------------------------------------------------------- #
# normalisation + resampling
# --------------------------------------------------------------------------- #
def logsumexp(a):
    a = _f(a)
    return float(lib().orc_logsumexp(a, a.size))


def ess(logw):
    a = _f(logw)
    return float(lib().orc_ess(a, a.size))


def cumsum(w):
    w = _f(w)
    out = np.zeros_like(w)
    lib().orc_cumsum(w, w.size, out)
    return out


def searchsorted_left(a, v):
    a = _f(a)
    v = _f(v)
    out = np.zeros(v.size, np.int32)
    lib().orc_searchsorted_left(a, a.size, v, v.size, out)
    return out


def systematic_indices(weights, u, return_cdf=False):
    w = _f(weights)
    idx = np.zeros(w.size, np.int32)
    cdf = np.zeros(w.size, np.float32) if return_cdf else None
    lib().orc_systematic_indices(w, w.size, np.float32(u), idx, cdf.ctypes.data if return_cdf else None)
    return (idx, cdf) if return_cdf else idx


def normalized_weights(logw):
    a = _f(logw)
    out = np.zeros_like(a)
    lib().orc_normalized_weights(a, a.size, out)
    return out


def categorical_indices(k, logw, m=None):
    a = _f(logw)
    m = a.size if m is None else m
    idx = np.zeros(m, np.int32)
    lib().orc_categorical_indices(_k(k), a, a.size, m, idx)
    return idx


# --------------------------------------------------------------------------- #
# localization
# --------------------------------------------------------------------------- #
class World:
    """examples/localization/core.py:43-57 (width, height, wall endpoint arrays)."""

    def __init__(self, width, height, walls):
        walls = np.asarray(walls, np.float32).reshape(-1, 4)
        self.width = float(width)
        self.height = float(height)
        self.x1 = np.ascontiguousarray(walls[:, 0])
        self.y1 = np.ascontiguousarray(walls[:, 1])
        self.x2 = np.ascontiguousarray(walls[:, 2])
        self.y2 = np.ascontiguousarray(walls[:, 3])
        self.W = walls.shape[0]

    def args(self):
        return (self.width, self.height, self.W, self.x1, self.y1, self.x2, self.y2)


# create_basic_multi_room_world (examples/localization/core.py:1006-1043)
BASIC_WALLS = [
    [4.0, 0.0, 4.0, 3.0],
    [4.0, 5.0, 4.0, 10.0],
    [8.0, 0.0, 8.0, 4.0],
    [8.0, 6.0, 8.0, 10.0],
    [5.0, 7.0, 7.0, 7.0],
    [9.0, 2.0, 10.0, 2.0],
    [10.0, 2.0, 10.0, 3.0],
    [10.0, 3.0, 9.0, 3.0],
    [9.0, 3.0, 9.0, 2.0],
]

# create_complex_multi_room_world (examples/localization/core.py:1046-1092)
COMPLEX_WALLS = BASIC_WALLS[:5] + [
    [9.0, 2.0, 10.5, 2.5],
    [10.5, 2.5, 10.0, 3.5],
    [10.0, 3.5, 8.5, 3.0],
    [8.5, 3.0, 9.0, 2.0],
    [1.0, 1.0, 2.5, 0.5],
    [3.0, 8.5, 3.5, 9.5],
    [5.5, 2.0, 6.5, 1.0],
    [6.0, 4.5, 7.0, 5.5],
    [8.5, 0.5, 9.5, 1.5],
    [10.5, 8.0, 11.0, 9.0],
]

# create_waypoint_trajectory_room1_to_room3 (examples/localization/data.py:75-99)
WAYPOINTS = [
    (1.2, 1.2), (1.5, 1.5), (2.5, 2.5), (3.5, 3.5), (3.9, 4.0), (4.1, 4.0), (4.5, 4.0),
    (5.5, 4.2), (6.5, 4.5), (7.5, 5.0), (7.9, 5.0), (8.1, 5.0), (8.5, 5.0), (9.5, 6.0),
    (10.5, 7.5), (11.0, 8.5), (11.5, 9.5),
]


def basic_world():
    return World(12.0, 10.0, BASIC_WALLS)


def complex_world():
    return World(12.0, 10.0, COMPLEX_WALLS)


def lidar(px, py, theta, world: World, R: int = 8):
    px = _f(px).reshape(-1)
    n = px.size
    out = np.zeros((n, R), np.float32)
    lib().orc_lidar(px, _f(py, n), _f(theta, n), n, *world.args(), R, out.reshape(-1))
    return out


def localization_observations(world: World, seed: int = 42, R: int = 8, noise_std=0.15):
    """main.py:141-153 + data.py:124-165: key(seed) -> split -> subkey; key1 = split(subkey)[0];
    noise_keys = split(key1, T); angle_keys = split(noise_keys[i], R); obs = max(0, d + n*0.15).
    Returns (obs[T, R], smc_key) where smc_key is the next subkey main.py:180 would hand on."""
    k = key(seed)
    ks = split(k, 2)
    k, sub = ks[0], ks[1]
    key1 = split(sub, 2)[0]
    T = len(WAYPOINTS)
    noise_keys = split(key1, T)
    xs = np.array([w[0] for w in WAYPOINTS], np.float32)
    ys = np.array([w[1] for w in WAYPOINTS], np.float32)
    true_d = lidar(xs, ys, np.zeros(T, np.float32), world, R)
    obs = np.zeros((T, R), np.float32)
    for i in range(T):
        angle_keys = split(noise_keys[i], R)
        for j in range(R):
            nz = normal(angle_keys[j], 1)[0]
            obs[i, j] = max(np.float32(0.0), np.float32(true_d[i, j] + np.float32(nz * np.float32(noise_std))))
    smc_key = split(k, 2)[1]
    return obs, smc_key


class LocState:
    """SoA particle state used by the restatement."""

    def __init__(self, n, R):
        self.n, self.R = n, R
        for name in ("x", "y", "th", "mx", "my", "mth", "logw"):
            setattr(self, name, np.zeros(n, np.float32))
        self.lp = np.zeros(n * R, np.float32)

    def cols(self):
        return (self.x, self.y, self.th, self.mx, self.my, self.mth)

    def copy(self):
        s = LocState(self.n, self.R)
        for name in ("x", "y", "th", "mx", "my", "mth", "logw", "lp"):
            setattr(s, name, getattr(self, name).copy())
        return s


def loc_extend(st: LocState, keys, is_initial, obs, world: World, offset=0):
    obs = _f(obs)
    lib().orc_loc_extend(_k(keys[0]), _k(keys[1]), _k(keys[2]), int(is_initial), st.n, offset, obs,
                         st.R, *world.args(), st.x, st.y, st.th, st.mx, st.my, st.mth, st.logw, st.lp)


def loc_extend_locopt(st: LocState, keys, grid, radius, noise, obs, world: World, offset=0):
    """extend() with the locally-optimal grid proposal (localization/core.py:730-809)."""
    L = lib()
    L.orc_loc_extend_locopt.argtypes = [u32p, u32p, u32p, C.c_int32, C.c_float, C.c_float, C.c_int64, C.c_uint64,
                                        f32p, C.c_int32, C.c_float, C.c_float, C.c_int32, f32p, f32p, f32p, f32p] + \
        [f32p] * 8
    L.orc_loc_extend_locopt(_k(keys[0]), _k(keys[1]), _k(keys[2]), grid, radius, noise, st.n, offset, _f(obs),
                            st.R, *world.args(), st.x, st.y, st.th, st.mx, st.my, st.mth, st.logw, st.lp)


def loc_locopt_grid_scores(grid, radius, pose, obs, world: World, R=8):
    """The grid^3 scores of one particle of the locally-optimal proposal, in argmax order:
    -> (scores, sum of |terms| per score, grid poses (grid^3, 3))."""
    L = lib()
    L.orc_loc_locopt_grid_scores.argtypes = [C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, f32p, C.c_int32,
                                             C.c_float, C.c_float, C.c_int32, f32p, f32p, f32p, f32p, f32p, f32p, f32p]
    g3 = grid ** 3
    sc, tm, ps = np.zeros(g3, np.float32), np.zeros(g3, np.float32), np.zeros((g3, 3), np.float32)
    L.orc_loc_locopt_grid_scores(grid, radius, float(pose[0]), float(pose[1]), float(pose[2]), _f(obs), R,
                                 *world.args(), sc, tm, ps.reshape(-1))
    return sc, tm, ps


def loc_mh(st: LocState, keys, is_initial, obs, world: World, offset=0):
    obs = _f(obs)
    acc = np.zeros(st.n, np.uint8)
    n_acc = lib().orc_loc_mh(_k(keys[0]), _k(keys[1]), _k(keys[2]), _k(keys[3]), int(is_initial), st.n,
                             offset, obs, st.R, *world.args(), st.x, st.y, st.th, st.mx, st.my, st.mth,
                             st.lp, acc.ctypes.data)
    return int(n_acc), acc


def loc_mh_diag(st: LocState, keys, is_initial, obs, world: World, offset=0):
    """loc_mh that also reports the proposal: -> (n_acc, accept flags, diag (n, 6) =
    {x', y', theta', weight before min0, log u, sum_r |logpdf_r(new)| + |old_score_r|})."""
    obs = _f(obs)
    acc = np.zeros(st.n, np.uint8)
    diag = np.zeros((st.n, 6), np.float32)
    n_acc = lib().orc_loc_mh_diag(_k(keys[0]), _k(keys[1]), _k(keys[2]), _k(keys[3]), int(is_initial), st.n,
                                  offset, obs, st.R, *world.args(), st.x, st.y, st.th, st.mx, st.my, st.mth,
                                  st.lp, acc.ctypes.data, diag.reshape(-1))
    return int(n_acc), acc, diag


def loc_hmc_diag(st: LocState, keys, step_size, n_steps, is_initial, obs, world: World, offset=0):
    """loc_hmc that also reports the proposal: diag (n, 9) = {x_L, y_L, theta_L, log_alpha before
    min0, log u, logp(q_0), logp(q_L), log N(p_0), log N(-p_L)}."""
    obs = _f(obs)
    acc = np.zeros(st.n, np.uint8)
    diag = np.zeros((st.n, 9), np.float32)
    n_acc = lib().orc_loc_hmc_diag(_k(keys[0]), _k(keys[1]), _k(keys[2]), _k(keys[3]), step_size, n_steps,
                                   int(is_initial), st.n, offset, obs, st.R, *world.args(), st.x, st.y, st.th,
                                   st.mx, st.my, st.mth, st.lp, acc.ctypes.data, diag.reshape(-1))
    return int(n_acc), acc, diag


def loc_logp_grad(st: LocState, is_initial, obs, world: World):
    obs = _f(obs)
    logp = np.zeros(st.n, np.float32)
    grad = np.zeros((st.n, 3), np.float32)
    lib().orc_loc_logp_grad(st.x, st.y, st.th, st.mx, st.my, st.mth, int(is_initial), st.n, obs, st.R,
                            *world.args(), logp, grad.reshape(-1))
    return logp, grad


def loc_logp_grad_abs(st: LocState, is_initial, obs, world: World):
    """loc_logp_grad plus gabs (n, 4): sum of |terms| of each gradient component and of logp."""
    L = lib()
    L.orc_loc_logp_grad_abs.argtypes = L.orc_loc_logp_grad.argtypes + [f32p]
    obs = _f(obs)
    logp = np.zeros(st.n, np.float32)
    grad = np.zeros((st.n, 3), np.float32)
    gabs = np.zeros((st.n, 4), np.float32)
    L.orc_loc_logp_grad_abs(st.x, st.y, st.th, st.mx, st.my, st.mth, int(is_initial), st.n, obs, st.R,
                            *world.args(), logp, grad.reshape(-1), gabs.reshape(-1))
    return logp, grad, gabs


def loc_hmc(st: LocState, keys, step_size, n_steps, is_initial, obs, world: World, offset=0):
    """keys = (k_p_theta, k_p_x, k_p_y, k_u) in draw order."""
    obs = _f(obs)
    acc = np.zeros(st.n, np.uint8)
    n_acc = lib().orc_loc_hmc(_k(keys[0]), _k(keys[1]), _k(keys[2]), _k(keys[3]), step_size, n_steps,
                              int(is_initial), st.n, offset, obs, st.R, *world.args(), st.x, st.y, st.th,
                              st.mx, st.my, st.mth, st.lp, acc.ctypes.data)
    return int(n_acc), acc


def loc_smc(k, n, obs, world: World, kernel="none", K=0, step_size=0.01, n_steps=10,
            method="systematic"):
    """seed(rejuvenation_smc)(key, localization_model, ...) restated (smc.py:655-785)."""
    obs = np.ascontiguousarray(np.asarray(obs, np.float32))
    T, R = obs.shape
    st = LocState(n, R)
    lml = np.zeros(1, np.float32)
    lml_acc = np.zeros(1, np.float32)
    ess_trace = np.zeros(T, np.float32)
    resampled = np.zeros(T, np.int32)
    kk = {"none": 0, "mh": 1, "hmc": 2}[kernel]
    acc = np.zeros(max(1, T * max(K, 1)), np.int64)
    rc = lib().orc_loc_smc(_k(k), n, T, obs.reshape(-1), R, *world.args(), kk, K, step_size, n_steps,
                           {"categorical": 0, "systematic": 1}[method], st.x, st.y, st.th, st.mx, st.my,
                           st.mth, st.logw, lml, ess_trace, resampled, acc.ctypes.data, lml_acc)
    assert rc == 0
    return dict(state=st, lml=float(lml[0]), lml_acc=float(lml_acc[0]), ess=ess_trace,
                resampled=resampled, accept=acc.reshape(T, max(K, 1)))


# --------------------------------------------------------------------------- #
# linear-Gaussian state-space model + exact Kalman filter
# --------------------------------------------------------------------------- #
def _declare_lg(L):
    if getattr(L, "_lg_declared", False):
        return
    L.orc_lgssm_extend.argtypes = [u32p, C.c_int32, C.c_int64, C.c_int32, C.c_int32] + [f32p] * 7 + [f32p] * 3
    L.orc_lgssm_extend.restype = C.c_int
    L.orc_lgssm_mh.argtypes = [u32p, u32p, C.c_int32, C.c_int64, C.c_int32, C.c_int32] + [f32p] * 7
    L.orc_lgssm_mh.restype = C.c_int64
    L._lg_declared = True


class LgModel:
    def __init__(self, initial_mean, initial_cov, A, Q, Cm, R):
        f = lambda a: np.ascontiguousarray(np.asarray(a, np.float32))
        self.mu0 = f(initial_mean).reshape(-1)
        self.ds = self.mu0.size
        self.C = f(Cm).reshape(-1, self.ds)
        self.dobs = self.C.shape[0]
        self.cov0, self.A, self.Q = (f(a).reshape(self.ds, self.ds) for a in (initial_cov, A, Q))
        self.R = f(R).reshape(self.dobs, self.dobs)


def lgssm_extend(model: LgModel, k, is_initial, obs, state, mean, logw):
    """In place on SoA arrays state/mean (ds, n) and logw (n)."""
    _declare_lg(lib())
    n = logw.size
    rc = lib().orc_lgssm_extend(_k(k), int(is_initial), n, model.ds, model.dobs, model.mu0,
                                model.cov0.reshape(-1), model.A.reshape(-1), model.Q.reshape(-1),
                                model.C.reshape(-1), model.R.reshape(-1), _f(obs).reshape(-1),
                                state.reshape(-1), mean.reshape(-1), logw)
    assert rc == 0


def lgssm_mh(model: LgModel, kstate, ku, is_initial, obs, state, mean):
    _declare_lg(lib())
    n = state.shape[1]
    return int(lib().orc_lgssm_mh(_k(kstate), _k(ku), int(is_initial), n, model.ds, model.dobs,
                                  model.cov0.reshape(-1), model.Q.reshape(-1), model.C.reshape(-1),
                                  model.R.reshape(-1), _f(obs).reshape(-1), state.reshape(-1), mean.reshape(-1)))


def lgssm_mh_diag(model: LgModel, kstate, ku, is_initial, obs, state, mean):
    """lgssm_mh that also returns, per particle, {proposal (ds), weight before min0, log u, sum of the
    magnitudes of the two observation log-densities}: -> (accepted, diag (n, ds + 3))."""
    L = lib()
    _declare_lg(L)
    L.orc_lgssm_mh_diag.argtypes = list(L.orc_lgssm_mh.argtypes) + [f32p]
    L.orc_lgssm_mh_diag.restype = C.c_int64
    n = state.shape[1]
    diag = np.zeros((n, model.ds + 3), np.float32)
    acc = int(L.orc_lgssm_mh_diag(_k(kstate), _k(ku), int(is_initial), n, model.ds, model.dobs,
                                  model.cov0.reshape(-1), model.Q.reshape(-1), model.C.reshape(-1),
                                  model.R.reshape(-1), _f(obs).reshape(-1), state.reshape(-1), mean.reshape(-1),
                                  diag.reshape(-1)))
    return acc, diag


def kalman_filter(observations, initial_mean, initial_cov, A, Q, Cm, R):
    """Exact filter + log marginal likelihood (state_space.py:434-530), in float64."""
    y = np.asarray(observations, np.float64)
    T = y.shape[0]
    mu = np.asarray(initial_mean, np.float64).reshape(-1)
    d = mu.size
    P = np.asarray(initial_cov, np.float64).reshape(d, d)
    A, Q = np.asarray(A, np.float64).reshape(d, d), np.asarray(Q, np.float64).reshape(d, d)
    Cm = np.asarray(Cm, np.float64).reshape(-1, d)
    o = Cm.shape[0]
    R = np.asarray(R, np.float64).reshape(o, o)
    means, covs, lml = np.zeros((T, d)), np.zeros((T, d, d)), 0.0
    for t in range(T):
        if t > 0:
            mu, P = A @ mu, A @ P @ A.T + Q
        innov = y[t].reshape(o) - Cm @ mu
        S = Cm @ P @ Cm.T + R
        K = P @ Cm.T @ np.linalg.inv(S)
        lml += -0.5 * (innov @ np.linalg.solve(S, innov) + np.log(np.linalg.det(S)) + o * np.log(2 * np.pi))
        mu, P = mu + K @ innov, P - K @ Cm @ P
        means[t], covs[t] = mu, P
    return means, covs, float(lml)


# --------------------------------------------------------------------------- #
# outlier-mixture curve fitting: per-chain enumerative Gibbs + HMC (gjx_oracle.c section 9)
# --------------------------------------------------------------------------- #
class OutlierModel(C.Structure):
    _fields_ = [("n_points", C.c_int32), ("xs", C.c_float * 64), ("ys", C.c_float * 64), ("alpha", C.c_float),
                ("beta", C.c_float), ("inlier_sigma", C.c_float), ("out_lo", C.c_float), ("out_hi", C.c_float)]


def outlier_model(xs, ys, alpha=1.0, beta=3.0, inlier_sigma=0.1, out_lo=-2.0, out_hi=2.0):
    xs, ys = _f(xs).reshape(-1), _f(ys).reshape(-1)
    m = OutlierModel()
    m.n_points = xs.size
    for j in range(xs.size):
        m.xs[j], m.ys[j] = float(xs[j]), float(ys[j])
    m.alpha, m.beta, m.inlier_sigma, m.out_lo, m.out_hi = alpha, beta, inlier_sigma, out_lo, out_hi
    return m


def outlier_dataset(n_points=18, seed=42):
    """figs.py:657-688: 13 inliers on the true curve + 5 outliers near 1.2 (key(42) -> split chain)."""
    xs = np.linspace(0, 1, n_points).astype(np.float32)
    y_true = (np.float32(-0.211) + np.float32(-0.395) * xs + np.float32(0.673) * xs * xs).astype(np.float32)
    k = key(seed)
    k, sub = split(k, 2)
    ys = (y_true + np.float32(0.1) * normal(sub, n_points)).astype(np.float32)
    idx = [3, 6, 9, 13, 16]
    k, sub = split(k, 2)
    vals = (np.float32(1.2) + np.float32(0.2) * normal(sub, len(idx))).astype(np.float32)
    ys[idx] = vals
    return xs, ys, idx


def outlier_init(model, base_key, n, offset=0):
    L = lib()
    st = {"keys": np.zeros((n, 2), np.uint32), "outliers": np.zeros(n, np.uint64)}
    for k_ in ("a", "b", "c", "rate"):
        st[k_] = np.zeros(n, np.float32)
    L.orc_outlier_init.argtypes = [C.POINTER(OutlierModel), u32p, C.c_uint64, C.c_int64, u32p, f32p, f32p, f32p,
                                   f32p, np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")]
    L.orc_outlier_init(C.byref(model), _k(base_key), offset, n, st["keys"].reshape(-1), st["a"], st["b"], st["c"],
                       st["rate"], st["outliers"])
    return st


def outlier_gibbs_hmc(model, st, n_iters, step_size, n_steps):
    L = lib()
    L.orc_outlier_gibbs_hmc.argtypes = [C.POINTER(OutlierModel), C.c_int32, C.c_float, C.c_int32, C.c_int64, u32p,
                                        f32p, f32p, f32p, f32p,
                                        np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")]
    L.orc_outlier_gibbs_hmc.restype = C.c_int64
    n = st["a"].size
    return int(L.orc_outlier_gibbs_hmc(C.byref(model), n_iters, step_size, n_steps, n, st["keys"].reshape(-1),
                                       st["a"], st["b"], st["c"], st["rate"], st["outliers"]))


def outlier_gibbs_hmc_diag(model, st, step_size, n_steps):
    """ONE iteration of the Gibbs + HMC kernel on `st` (advanced in place) with the quantities of its decisions:
    -> (accepted, diag (10 + P, n)); rows: 0..3 proposal, 4 log alpha (before min0), 5 log u, 6 / 7 assess at the
    start / end point, 8 / 9 momentum log-densities at start / end, 10 + j the Gibbs margin v1 - v0 of point j,
    10 + P + j the sum of the magnitudes of the four terms of that margin."""
    L = lib()
    L.orc_outlier_gibbs_hmc_diag.argtypes = [C.POINTER(OutlierModel), C.c_float, C.c_int32, C.c_int64, u32p,
                                             f32p, f32p, f32p, f32p,
                                             np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS"), f32p]
    L.orc_outlier_gibbs_hmc_diag.restype = C.c_int64
    n = st["a"].size
    diag = np.zeros((10 + 2 * int(model.n_points), n), np.float32)
    acc = int(L.orc_outlier_gibbs_hmc_diag(C.byref(model), step_size, n_steps, n, st["keys"].reshape(-1),
                                           st["a"], st["b"], st["c"], st["rate"], st["outliers"], diag.reshape(-1)))
    return acc, diag


def outlier_hmc_f64(model, st, masks_after_gibbs, step_size, n_steps, noise_seed=0):
    """The HMC half of one iteration from state `st` (not advanced) in double precision, on the f32 draws and the
    given post-Gibbs indicators: -> (10, n) float64, rows 0..9 as in outlier_gibbs_hmc_diag.  noise_seed != 0:
    with f32-sized relative perturbations injected at every update of the trajectory (a randomised model of the
    roundings of an f32 evaluation)."""
    L = lib()
    u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
    L.orc_outlier_hmc_f64.argtypes = [C.POINTER(OutlierModel), C.c_float, C.c_int32, C.c_int64, u32p, f32p, f32p, f32p,
                                      f32p, u64p, C.c_uint32, np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")]
    n = st["a"].size
    diag = np.zeros((10, n), np.float64)
    L.orc_outlier_hmc_f64(C.byref(model), step_size, n_steps, n, np.ascontiguousarray(st["keys"]).reshape(-1),
                          st["a"], st["b"], st["c"], st["rate"], np.ascontiguousarray(masks_after_gibbs, np.uint64),
                          int(noise_seed), diag.reshape(-1))
    return diag


def outlier_logp_grad(model, q, mask):
    L = lib()
    L.orc_outlier_logp_grad.argtypes = [C.POINTER(OutlierModel), f32p, C.c_uint64, f32p]
    L.orc_outlier_logp_grad.restype = C.c_float
    g = np.zeros(4, np.float32)
    lp = L.orc_outlier_logp_grad(C.byref(model), _f(q).reshape(4), int(mask), g)
    return float(lp), g


# --------------------------------------------------------------------------- #
# Game-of-Life inversion by Gibbs sweeps (gjx_oracle.c section 10; examples/gol/core.py)
# --------------------------------------------------------------------------- #
def gol_step(board):
    b = np.ascontiguousarray(board, np.uint8)
    out = np.zeros_like(b)
    lib().orc_gol_step.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    lib().orc_gol_step(b.ctypes.data, b.shape[0], out.ctypes.data)
    return out


def gol_init(k, n):
    out = np.zeros((n, n), np.uint8)
    lib().orc_gol_init.argtypes = [u32p, C.c_int32, C.c_void_p]
    lib().orc_gol_init(_k(k), n, out.ctypes.data)
    return out


def gol_gibbs_pass(cur, fut, oy, ox, p_flip, k):
    """In place on `cur` (n x n uint8)."""
    L = lib()
    L.orc_gol_gibbs_pass.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, u32p]
    L.orc_gol_gibbs_pass(cur.ctypes.data, np.ascontiguousarray(fut, np.uint8).ctypes.data, cur.shape[0], oy, ox,
                         p_flip, _k(k))


def gol_gibbs_pass_margins(cur, fut, oy, ox, p_flip, k):
    """gol_gibbs_pass (in place on `cur`) that also returns u - (1 - p_zero) of every updated cell, shape
    (ceil((n - oy) / 3), ceil((n - ox) / 3)): the new bit is 1 iff the margin is negative."""
    L = lib()
    L.orc_gol_gibbs_pass_margins.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, u32p,
                                             f32p]
    n = cur.shape[0]
    m = np.zeros(((n - oy + 2) // 3, (n - ox + 2) // 3), np.float32)
    L.orc_gol_gibbs_pass_margins(cur.ctypes.data, np.ascontiguousarray(fut, np.uint8).ctypes.data, n, oy, ox, p_flip,
                                 _k(k), m.reshape(-1))
    return m


def gol_sweep(cur, fut, p_flip, sweep_key):
    L = lib()
    L.orc_gol_sweep.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_float, u32p]
    L.orc_gol_sweep(cur.ctypes.data, np.ascontiguousarray(fut, np.uint8).ctypes.data, cur.shape[0], p_flip,
                    _k(sweep_key))


def gol_summary(cur, fut, p_flip):
    L = lib()
    L.orc_gol_summary.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.POINTER(C.c_int64),
                                  C.POINTER(C.c_double)]
    bad, sc = C.c_int64(), C.c_double()
    cur, fut = np.ascontiguousarray(cur, np.uint8), np.ascontiguousarray(fut, np.uint8)
    L.orc_gol_summary(cur.ctypes.data, fut.ctypes.data, cur.shape[0], p_flip, C.byref(bad), C.byref(sc))
    return int(bad.value), float(sc.value)


def gol_run(k, target, p_flip, n_frames, per_frame):
    """Boards after init and after every frame: (n_frames + 1, n, n) uint8."""
    L = lib()
    L.orc_gol_run.argtypes = [u32p, C.c_void_p, C.c_int32, C.c_float, C.c_int32, C.c_int32, C.c_void_p]
    target = np.ascontiguousarray(target, np.uint8)
    n = target.shape[0]
    out = np.zeros((n_frames + 1, n, n), np.uint8)
    L.orc_gol_run(_k(k), target.ctypes.data, n, p_flip, n_frames, per_frame, out.ctypes.data)
    return out


# --------------------------------------------------------------------------- #
# rejuvenation_smc for the localization model, one stage at a time (the C driver orc_loc_smc
# composed in Python from the same C primitives and the same key schedule; tests/test_oracle.py
# checks that the two agree bit for bit).  Used by the teacher-forced parity checks, which need the
# oracle's state between the stages of a step.
# --------------------------------------------------------------------------- #
class LocSMCStepper:
    """Stages of smc_step (smc.py:731-767): extend -> ESS / cond(resample) -> K moves."""

    def __init__(self, k, n, obs, world: World, kernel="mh", K=0, step_size=0.01, n_steps=10):
        self.obs = np.ascontiguousarray(np.asarray(obs, np.float32))
        self.T, self.R = self.obs.shape
        self.n, self.world, self.kernel, self.K = n, world, kernel, K
        self.eps, self.L = step_size, n_steps
        self.st = LocState(n, self.R)
        self.top = Seed(k)
        self.scan_key = None
        self.lml_acc = np.float32(0.0)
        self.t = -1

    def stage_keys(self, t):
        """Interpreter of stage t: the top-level one for t = 0, fold_in(scan_key, t - 1) after."""
        return self.top if t == 0 else Seed(fold_in(self.scan_key, t - 1))

    def extend(self, t, ks):
        keys = [ks.next() for _ in range(3)]
        loc_extend(self.st, keys, t == 0, self.obs[t], self.world)
        return keys

    def normalise(self):
        return logsumexp(self.st.logw), ess(self.st.logw)

    def resample(self, kc, lse):
        """Systematic resampling in the cond's interpreter (fresh Seed on kc):
        -> (u, ancestors, normalised weights, cdf)."""
        kdraw = Seed(kc).next()
        u = float(uniform(kdraw, 1)[0])
        w = normalized_weights(self.st.logw)         # exp(logw - logsumexp(logw)), smc.py:127-128
        idx, cdf = systematic_indices(w, u, return_cdf=True)
        for name in ("x", "y", "th", "mx", "my", "mth"):
            setattr(self.st, name, np.ascontiguousarray(getattr(self.st, name)[idx]))
        self.st.lp = np.ascontiguousarray(self.st.lp.reshape(self.n, self.R)[idx].reshape(-1))
        self.lml_acc = np.float32(self.lml_acc + (np.float32(lse) - np.float32(np.log(np.float32(self.n)))))
        self.st.logw[:] = 0.0
        return u, idx, w, cdf

    def move_keys(self, kr, m):
        km = Seed(fold_in(kr, m))
        if self.kernel == "mh":
            return [km.next() for _ in range(4)]
        a, b, c = km.next(), km.next(), km.next()
        km.next()                                    # leapfrog scan_p: one split, no draw
        return [a, b, c, km.next()]

    def move(self, t, keys4, diag=False):
        if self.kernel == "mh":
            fn = loc_mh_diag if diag else loc_mh
            return fn(self.st, keys4, t == 0, self.obs[t], self.world)
        fn = loc_hmc_diag if diag else loc_hmc
        return fn(self.st, keys4, self.eps, self.L, t == 0, self.obs[t], self.world)

    def stage(self, t):
        """One whole stage, as orc_loc_smc runs it: -> dict(ess, resampled, accept counts)."""
        ks = self.stage_keys(t)
        self.extend(t, ks)
        kc = ks.next()                               # cond_p: one split whether or not it is taken
        lse, e = self.normalise()
        take = e < float(self.n // 2)
        if take:
            self.resample(kc, lse)
        acc = []
        if self.kernel != "none":
            kr = ks.next()                           # scan_p over the moves: one split
            for m in range(self.K):
                acc.append(self.move(t, self.move_keys(kr, m))[0])
        if t == 0:
            self.scan_key = ks.next()                # scan_p over observations[1:]
        self.t = t
        return dict(ess=e, resampled=bool(take), accept=acc)

    def log_marginal_likelihood(self):
        lse = np.float32(logsumexp(self.st.logw))
        return float(np.float32(self.lml_acc + (lse - np.float32(np.log(np.float32(self.n))))))
