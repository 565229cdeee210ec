"""Teacher-forced parity of the localization SMC path against the CPU restatement: every stage
and every move starts BOTH sides from the oracle's state, so every quantity is held at
north_star's bar (1e-5 relative; bit-exact for indices).  The bars, and how a particle outside one
is justified individually (lidar discontinuities), are stated in oracle/parity.py, which holds
the checks (smoke() uses them too).  The fused production kernels are tied to the verified
single-move entry points bit for bit (GPU against GPU: no tolerance)."""
import numpy as np
import pytest
import torch

from oracle import parity as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def B():
    from genjax_amd import backend

    assert torch.cuda.is_available()
    return backend


def _world(B, oracle, kind="basic", R=8):
    ow = oracle.basic_world() if kind == "basic" else oracle.complex_world()
    walls = np.stack([ow.x1, ow.y1, ow.x2, ow.y2], axis=1)
    return ow, B.make_loc_world(ow.width, ow.height, walls, n_rays=R)


def _start_state(oracle, ow, obs, k, n, is_initial, R=8):
    seq = oracle.Seed(k)
    st = oracle.LocState(n, R)
    oracle.loc_extend(st, [seq.next() for _ in range(3)], True, obs[0], ow)
    if is_initial:
        return st, obs[0]
    oracle.loc_extend(st, [seq.next() for _ in range(3)], False, obs[1], ow)
    return st, obs[1]


# (16 rays: the 8-slot paired instantiations of the kernels, P = 8; 32 rays: 16 slots; 11 rays: 16 slots unpaired and
# not full -- the reference's own setting is 8 rays, P = 4)
@pytest.mark.parametrize("kind,R", [("basic", 8), ("complex", 8), ("basic", 16), ("basic", 32), ("complex", 11)])
def test_extend_teacher_forced(B, oracle, kind, R):
    """init and three extend steps, each from the oracle's state."""
    ow, gw = _world(B, oracle, kind, R)
    obs, k = oracle.localization_observations(ow, R=R)
    n = 60000
    seq = oracle.Seed(k)
    st = oracle.LocState(n, R)
    total = sum(P.check_extend(torch, B, oracle, gw, ow, st, [seq.next() for _ in range(3)], t == 0, obs[t],
                               tag=f"extend t={t}") for t in range(4))
    print(f"\nextend teacher-forced ({kind}, {R} rays): 4 stages x {n} particles, {total} discontinuity particles, all justified")


@pytest.mark.parametrize("kind,is_initial,R", [("basic", True, 8), ("basic", False, 8), ("complex", False, 8),
                                                ("complex", False, 16), ("basic", False, 32), ("basic", False, 11)])
def test_mh_teacher_forced_per_move(B, oracle, kind, is_initial, R):
    ow, gw = _world(B, oracle, kind, R)
    obs, k = oracle.localization_observations(ow, R=R)
    n, K = 50000, 6
    st, t_obs = _start_state(oracle, ow, obs, k, n, is_initial, R)
    tot = np.zeros(2, int)
    for m in range(K):
        tot += P.check_mh_move(torch, B, oracle, gw, ow, st, oracle.split(oracle.fold_in(k, m), 4), is_initial, t_obs,
                               tag=f"mh move {m}")
    print(f"\nmh teacher-forced ({kind}, is_initial={is_initial}, {R} rays): {K} moves x {n} particles, "
          f"{tot[0]} discontinuity particles, {tot[1]} accept flips, all justified")


@pytest.mark.parametrize("kind,R", [("basic", 8), ("complex", 8), ("basic", 16), ("basic", 32)])
def test_hmc_teacher_forced_per_move(B, oracle, kind, R):
    ow, gw = _world(B, oracle, kind, R)
    obs, k = oracle.localization_observations(ow, R=R)
    n, K = 20000, 3
    st, t_obs = _start_state(oracle, ow, obs, k, n, False, R)
    tot = np.zeros(2, int)
    # 32 rays make the posterior four times as stiff: at the reference's step (0.01) the leapfrog is unstable there and
    # trajectories amplify the last bit of the start to O(1) -- nothing to compare.  A step the integrator is stable at.
    eps = 0.01 if R <= 16 else 0.0025
    for m in range(K):
        tot += P.check_hmc_move(torch, B, oracle, gw, ow, st, oracle.split(oracle.fold_in(k, 100 + m), 4), eps, 10,
                                False, t_obs, tag=f"hmc move {m}")
    print(f"\nhmc teacher-forced ({kind}, {R} rays): {K} moves x {n} particles, {tot[0]} particles outside the bar, "
          f"{tot[1]} accept flips, all justified")


@pytest.mark.parametrize("kind,R", [("basic", 8), ("complex", 8), ("complex", 16), ("basic", 32)])
def test_logp_grad_at_1e5_of_its_terms(B, oracle, kind, R):
    """assess() + gradient (the function hmc differentiates): log-density within 1e-5 of sum |log N|, every
    gradient component within 1e-5 of sum_r (|obs_r| + d_r) / sigma^2 |d d_r / d q| (the ray terms are products
    whose first factor (obs_r - d_r) / sigma^2 is a cancelling difference); a particle outside lies within the
    movement of the oracle's own gradient within 3e-6 of the pose (a ray on the edge between two winners)."""
    ow, gw = _world(B, oracle, kind, R)
    obs, k = oracle.localization_observations(ow, R=R)
    n = 40000
    st, t_obs = _start_state(oracle, ow, obs, k, n, False, R)
    g = P.gpu_state(torch, st)
    logp, grad = B.loc_logp_grad(gw, False, t_obs, g["x"], g["y"], g["th"], g["mx"], g["my"], g["mth"])
    logp, grad = logp.cpu().numpy(), grad.cpu().numpy().T
    rlogp, rgrad, gabs = oracle.loc_logp_grad_abs(st, False, t_obs, ow)
    bad = (np.abs(logp - rlogp) > P.RTOL * gabs[:, 3]) | (np.abs(grad - rgrad) > P.RTOL * gabs[:, :3]).any(axis=1)
    flagged = np.flatnonzero(bad)
    lines = []
    for i in flagged:
        J = P.JITTER
        s1 = oracle.LocState(len(J), R)
        s1.x[:], s1.y[:], s1.th[:] = st.x[i] + J[:, 0], st.y[i] + J[:, 1], st.th[i] + J[:, 2]
        s1.mx[:], s1.my[:], s1.mth[:] = st.mx[i], st.my[i], st.mth[i]
        jl, jg, ja = oracle.loc_logp_grad_abs(s1, False, t_obs, ow)
        ok, spread = P.within_spread(np.concatenate([jl[:, None], jg], axis=1), np.concatenate([[logp[i]], grad[i]]),
                                     P.RTOL * np.concatenate([[gabs[i, 3]], gabs[i, :3]]))
        lines.append(f"i={i} grad kernel {grad[i]} oracle {rgrad[i]} (bar {P.RTOL * gabs[i, :3]}); the oracle's own "
                     f"gradient moves by {spread[1:]} within 3e-6 of the pose: difference within that movement = {ok}")
        assert ok, "unjustified gradient difference: " + lines[-1]
    P.report("logp_grad: outside the bar, each within the movement of the oracle's own gradient next to the pose", lines)
    assert len(flagged) <= 2e-3 * n, len(flagged)
    print(f"\nlogp_grad ({kind}): {n} particles, {len(flagged)} on a winner edge, all justified")


@pytest.mark.parametrize("kind", ["basic", "complex"])
def test_logp_grad_on_the_lanes_redone_with_the_reference_formula(B, oracle, kind):
    """Headings that are exact multiples of pi / 2 put every other ray parallel to the axis-aligned walls: the lidar
    kernel redoes such lanes with the reference's formula verbatim and the winner is a wall BY ID, whose gradient
    takes the general wall formula (not the class formulas of the fast path).  Same bars as above; and the gradient
    of the rays that hit a wall head-on must be there (a zero d/dx for a pose facing a wall is the failure mode)."""
    ow, gw = _world(B, oracle, kind)
    obs, k = oracle.localization_observations(ow)
    n = 4096
    rng = np.random.default_rng(17)
    st = oracle.LocState(n, 8)
    st.x[:] = rng.uniform(0.3, ow.width - 0.3, n).astype(np.float32)
    st.y[:] = rng.uniform(0.3, ow.height - 0.3, n).astype(np.float32)
    st.th[:] = (np.float32(np.pi / 2) * rng.integers(-2, 3, n)).astype(np.float32)
    st.mx[:], st.my[:], st.mth[:] = st.x + np.float32(0.05), st.y - np.float32(0.03), st.th + np.float32(0.01)
    t_obs = obs[1]
    g = P.gpu_state(torch, st)
    logp, grad = B.loc_logp_grad(gw, False, t_obs, g["x"], g["y"], g["th"], g["mx"], g["my"], g["mth"])
    logp, grad = logp.cpu().numpy(), grad.cpu().numpy().T
    rlogp, rgrad, gabs = oracle.loc_logp_grad_abs(st, False, t_obs, ow)
    bad = (np.abs(logp - rlogp) > P.RTOL * gabs[:, 3]) | (np.abs(grad - rgrad) > P.RTOL * gabs[:, :3]).any(axis=1)
    flagged = np.flatnonzero(bad)
    for i in flagged:
        J = P.JITTER
        s1 = oracle.LocState(len(J), 8)
        s1.x[:], s1.y[:], s1.th[:] = st.x[i] + J[:, 0], st.y[i] + J[:, 1], st.th[i] + J[:, 2]
        s1.mx[:], s1.my[:], s1.mth[:] = st.mx[i], st.my[i], st.mth[i]
        jl, jg, ja = oracle.loc_logp_grad_abs(s1, False, t_obs, ow)
        ok, spread = P.within_spread(np.concatenate([jl[:, None], jg], axis=1), np.concatenate([[logp[i]], grad[i]]),
                                     P.RTOL * np.concatenate([[gabs[i, 3]], gabs[i, :3]]))
        assert ok, f"unjustified gradient difference on an exact lane: i={i} kernel {grad[i]} oracle {rgrad[i]}"
    assert len(flagged) <= 0.02 * n, len(flagged)
    # the sensor part of d/dx, d/dy (prior part removed): present on both sides for the same particles
    prior = np.stack([-(st.x - st.mx), -(st.y - st.my)], axis=1) / np.float32(0.25) ** 2
    sens_k, sens_o = np.abs(grad[:, :2] - prior).max(axis=1), np.abs(rgrad[:, :2] - prior).max(axis=1)
    assert ((sens_o > 1.0) & (sens_k < 1e-3)).sum() == 0, "kernel gradient lost the wall terms"
    print(f"\nlogp_grad on exact lanes ({kind}): {n} axis-parallel poses, {len(flagged)} on a winner edge, all justified")


@pytest.mark.parametrize("K,R", [(7, 8), (64, 8), (5, 32), (5, 11)])
def test_fused_mh_moves_equal_single_moves_bitwise(B, oracle, K, R):
    """K fused moves (one launch; also with the stored scores and the resampling gather in the launch) == K
    launches of one move: the production kernel is the composition of the moves verified one by one.  K = 64 is the
    most one launch takes (GJX_MAX_MOVES)."""
    ow, gw = _world(B, oracle, "basic", R)
    obs, k = oracle.localization_observations(ow, R=R)
    n = 30000
    st, t_obs = _start_state(oracle, ow, obs, k, n, False, R)
    mk = np.stack([oracle.split(oracle.fold_in(k, 50 + m), 4) for m in range(K)])
    g = P.gpu_state(torch, st)
    means = (g["mx"], g["my"], g["mth"])
    pose = (g["x"], g["y"], g["th"])
    step, c1 = pose, []
    for m in range(K):
        c = torch.zeros(1, dtype=torch.int64, device="cuda")
        step = B.loc_mh(gw, mk[m][None], False, t_obs, step, means, accept_counts=c)
        c1.append(int(c.item()))
    cK = torch.zeros(K, dtype=torch.int64, device="cuda")
    fused = B.loc_mh(gw, mk, False, t_obs, pose, means, accept_counts=cK)
    for a, b in zip(step, fused):
        assert torch.equal(a, b)
    assert cK.cpu().tolist() == c1
    scores = B.loc_ray_scores(gw, t_obs, *pose)              # stored scores + an identity "resampling"
    anc = torch.arange(n, dtype=torch.int32, device="cuda")
    src6 = [g[c].clone() for c in ("x", "y", "th", "mx", "my", "mth")]
    dst6 = B.loc_mh_resampled(gw, mk, False, t_obs, anc, src6, ray_scores=scores)
    for a, b in zip(fused, dst6[:3]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("R", [8, 32])
def test_fused_hmc_moves_equal_single_moves_bitwise(B, oracle, R):
    ow, gw = _world(B, oracle, "basic", R)
    obs, k = oracle.localization_observations(ow, R=R)
    n, K = 8192, 4
    st, t_obs = _start_state(oracle, ow, obs, k, n, False, R)
    mk = np.stack([oracle.split(oracle.fold_in(k, 70 + m), 4) for m in range(K)])
    g = P.gpu_state(torch, st)
    means = (g["mx"], g["my"], g["mth"])
    pose = (g["x"], g["y"], g["th"])
    step = pose
    for m in range(K):
        step = B.loc_hmc(gw, mk[m][None], 0.01, 10, False, t_obs, step, means)
    fused = B.loc_hmc(gw, mk, 0.01, 10, False, t_obs, pose, means)
    for a, b in zip(step, fused):
        assert torch.equal(a, b)


@pytest.mark.parametrize("kernel,n,K,stages", [("mh", 4096, 5, 6), ("mh", 1 << 16, 3, 4), ("hmc", 4096, 2, 4)])
def test_rejuvenation_smc_surface_teacher_forced(oracle, kernel, n, K, stages):
    """The reference's surface (init / extend / resample / rejuvenate with mh | hmc) walked stage by stage
    from the oracle's state: log-weights, lse, ESS, the resampling decision at 1e-5; ancestors and
    gathered columns bit-exact given the oracle's weights and uniform; every move as above."""
    import genjax_amd as G

    s = P.teacher_forced_smc(torch, G, oracle, n=n, K=K, stages=stages, kernel=kernel)
    assert any(s["resampled"])          # the run exercises the resampling branch
