"""Teacher-forced parity checks of the localization SMC path: HIP library (through the C ABI /
the Python surface) against the CPU restatement, one stage and one move at a time.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke()).  The product never imports this
module; the checks take the product's modules from the caller (teacher_forced_smc, which walks the
product's Python surface, imports it itself).

Every stage starts BOTH sides from the oracle's state, so nothing accumulates and every quantity
is held at north_star's bar, 1e-5 relative:
  choices / proposals       rtol 1e-5, atol 1e-6 (positions live in a 12 x 10 world)
  log-weights               1e-5 of (|logw_in| + sum_r |log N_r|): a log-weight is a SUM of log-densities
  log alpha (mh)            1e-5 of sum_r (|log N_r(new)| + |log N_r(old)|): a difference of log-weights
  lse, ESS                  rtol 1e-5, given the oracle's log-weights
  ancestors                 bit-exact, given the oracle's log-weights and uniform
  accept decisions          every flip is listed with its margin |log u - log alpha|, which must lie inside
                            the numerical difference of the two sides
The lidar likelihood is DISCONTINUOUS in the pose in f32 (a ray grazing a wall end point hits in one
evaluation order and misses in another).  A particle outside a bar is therefore checked
individually: the oracle itself, evaluated within a few 1e-6 of the pose (the size of the f32
roundings of the pose and of the ray angle), must reproduce the kernel's per-ray values
(justify_rays) or move by at least the difference (within_spread).  Each such particle is printed.
A ray that grazes a wall at a shallow angle is continuous but steep (with 16 rays some do): justify_rays then
accepts a value that lies between two neighbouring oracle samples, and says so in the line it prints.
"""
import itertools

import numpy as np

RTOL = 1e-5
ATOL_POS = 1e-6
LP_CONST = np.float32(0.9189385332046727) + np.float32(np.log(np.float32(0.3)))   # log(sigma sqrt(2 pi))

# pose perturbations of the size of the f32 roundings involved: ulp(12) = 9.5e-7 for x / y,
# 2.4e-7 for theta + 2 pi r / R, plus the <= 1.5 ulp of the kernel's sincos / reciprocal
_DXY = (0.0, -1e-6, 1e-6, -3e-6, 3e-6)
_DTH = (0.0, -3e-7, 3e-7, -1e-6, 1e-6)
JITTER = np.array([j for j in itertools.product(_DXY, _DXY, _DTH)], np.float32)   # row 0 = no perturbation


def _min0(w):
    return np.where(w > 0, np.float32(0.0), w)


def report(title, lines, limit=12):
    if lines:
        print(f"\n[{title}] {len(lines)} particle(s):")
        for ln in lines[:limit]:
            print("   ", ln)
        if len(lines) > limit:
            print(f"    ... and {len(lines) - limit} more")


def oracle_ray_scores(O, ow, obs, x, y, th, R=8):
    """The kernel's score convention, 0.5 ((obs_r - d_r) / sigma)^2 = -(log N) - log(sigma sqrt(2 pi)),
    from the oracle's per-ray log-densities: (n, R), plus the log-densities themselves."""
    x, y, th = (np.ascontiguousarray(np.asarray(v, np.float32).reshape(-1)) for v in (x, y, th))
    n = x.size
    lp = np.zeros((n, R), np.float32)
    O.lib().orc_loc_ray_logpdfs_n(x, y, th, n, np.ascontiguousarray(obs, np.float32), R, ow.width, ow.height,
                                  ow.W, ow.x1, ow.y1, ow.x2, ow.y2, lp.reshape(-1))
    return (-lp.astype(np.float64) - np.float64(LP_CONST)).astype(np.float32), lp


def justify_rays(O, ow, obs, pose, gpu_scores, rtol=RTOL, R=8):
    """Is every per-ray score the kernel computed for `pose` reproduced by the oracle at some pose
    within JITTER of it?  -> (ok, text).  gpu_scores: (R,) in the kernel's convention."""
    px = np.float32(pose[0]) + JITTER[:, 0]
    py = np.float32(pose[1]) + JITTER[:, 1]
    pt = np.float32(pose[2]) + JITTER[:, 2]
    sc, lp = oracle_ray_scores(O, ow, obs, px, py, pt, R)                # (J, R)
    g = np.asarray(gpu_scores, np.float32).reshape(1, R)
    # a ray's score is a summand of the particle's log-weight: its bar is 1e-5 of sum_r |log N_r|
    bar = rtol * np.abs(lp[0]).sum()
    close = np.abs(sc - g) <= bar
    per_ray = close.any(axis=0)
    # A ray that GRAZES a wall is continuous but steep (the score moves by many bars within 3e-6 of the pose): no
    # sample need land within one bar of the kernel's value.  Then the value must lie between two NEIGHBOURING samples
    # of the oracle (sorted) that are close together -- at most 4 bars or a fifth of the spread apart; at a jump (a
    # wall end point: the samples sit on two levels) the gap between the levels is the whole spread and this fails.
    steep = []
    for r in np.flatnonzero(~per_ray):
        v = np.sort(sc[:, r].astype(np.float64))
        j = int(np.searchsorted(v, g[0, r]))
        if 0 < j < v.size and (v[j] - v[j - 1]) <= max(4.0 * bar, 0.2 * (v[-1] - v[0])):
            per_ray[r] = True
            steep.append(int(r))
    moved = [r for r in range(R) if not close[0, r]]
    txt = (f"pose=({pose[0]:.7f}, {pose[1]:.7f}, {pose[2]:.7f}) rays off at the pose itself: {moved}; "
           + "; ".join(f"ray {r}: kernel {g[0, r]:.6g}, oracle {sc[0, r]:.6g}, oracle within 3e-6 of the pose "
                       f"spans [{sc[:, r].min():.6g}, {sc[:, r].max():.6g}]"
                       + (" (steep, not a jump: the kernel's value lies between two neighbouring oracle samples)"
                          if r in steep else "") for r in moved))
    return bool(per_ray.all()), txt


def within_spread(jittered, target, tol, sample_margin=0.0):
    """jittered (J, k): the oracle's own result at input + JITTER (row 0 = the input itself).  Is the
    kernel's `target` (k,) no further from the oracle's result than that result itself moves within
    JITTER (max - min per component), plus the bar `tol` (k,)?  True means: by the oracle's own account
    the function is discontinuous / ill-conditioned at this input to the extent of the difference.
    sample_margin: the J perturbations are a finite sample of the function's local variation and the extremes of a
    sample underestimate its range; a caller whose function is a long chaotic composition (an HMC trajectory) may
    allow that fraction of the measured spread on top (it says so in its report)."""
    jittered = np.asarray(jittered, np.float64)
    spread = np.nanmax(jittered, axis=0) - np.nanmin(jittered, axis=0)
    t = np.asarray(target, np.float64)
    return bool(np.all(np.abs(t - jittered[0]) <= spread * (1.0 + sample_margin) + tol)), spread


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def gpu_state(torch, st):
    return {name: _dev(torch, getattr(st, name).copy()) for name in ("x", "y", "th", "mx", "my", "mth")}


# --------------------------------------------------------------------------- #
# stage checks.  B = genjax_amd.backend, O = oracle, gw / ow = the world on either side.
# Each returns the number of individually justified particles (and flips); anything unjustified raises.
# --------------------------------------------------------------------------- #
def check_extend(torch, B, O, gw, ow, st, keys3, is_initial, obs_t, tag="extend", max_frac=2e-3):
    """One extend / init from the oracle's state `st` (advanced in place by the oracle)."""
    n = st.n
    prev = None if is_initial else tuple(_dev(torch, getattr(st, c).copy()) for c in ("x", "y", "th"))
    logw_in = st.logw.copy()
    scores = torch.empty((st.R, n), dtype=torch.float32, device="cuda")
    g = B.loc_extend(gw, keys3, is_initial, obs_t, prev, _dev(torch, logw_in), ray_scores=scores)
    O.loc_extend(st, keys3, is_initial, obs_t, ow)
    for name in ("mx", "my", "mth"):
        assert np.array_equal(g[name].cpu().numpy(), getattr(st, name)), (tag, name)
    got = {c: g[c].cpu().numpy() for c in ("x", "y", "th", "logw")}
    for c in ("x", "y", "th"):
        np.testing.assert_allclose(got[c], getattr(st, c), rtol=RTOL, atol=ATOL_POS, err_msg=f"{tag} {c}")
    lp = st.lp.reshape(n, st.R)
    tol = RTOL * (np.abs(logw_in) + np.abs(lp).sum(axis=1))
    d = np.abs(got["logw"].astype(np.float64) - st.logw.astype(np.float64))
    flagged = np.flatnonzero(np.isnan(d) | (d > tol))
    sc = scores.cpu().numpy()
    lines = []
    for i in flagged:
        ok, txt = justify_rays(O, ow, obs_t, (got["x"][i], got["y"][i], got["th"][i]), sc[:, i], R=st.R)
        lines.append(f"i={i} logw kernel {got['logw'][i]:.6g} oracle {st.logw[i]:.6g} (bar {tol[i]:.2g}); {txt}")
        assert ok, f"{tag}: unjustified log-weight difference: " + lines[-1]
    report(f"{tag}: log-weights outside 1e-5, each a lidar discontinuity", lines)
    assert len(flagged) <= max(2, max_frac * n), (tag, len(flagged))
    return len(flagged)


def check_normaliser(torch, B, O, logw, tag="normaliser"):
    """lse / ESS of the ORACLE's log-weights: -> (lse, ess) of the oracle."""
    stats = B.logsumexp_ess(_dev(torch, logw)).cpu().numpy()
    lse, ess = O.logsumexp(logw), O.ess(logw)
    assert stats[0] == logw.max(), tag
    np.testing.assert_allclose(stats[1], lse, rtol=RTOL, err_msg=f"{tag} lse")
    np.testing.assert_allclose(stats[2], ess, rtol=RTOL, err_msg=f"{tag} ess")
    return lse, ess


def check_mh_move(torch, B, O, gw, ow, st, keys4, is_initial, obs_t, tag="mh", max_frac=2e-3):
    """One mh move from the oracle's state (advanced in place by the oracle): proposals, weight,
    log u, accept decisions; then the fused production kernel == regenerate + accept + select bit
    for bit.  -> (particles justified as discontinuities, accept flips)."""
    n, R = st.n, st.R
    g = gpu_state(torch, st)
    pose, means = (g["x"], g["y"], g["th"]), (g["mx"], g["my"], g["mth"])
    old = {c: getattr(st, c).copy() for c in ("x", "y", "th")}
    lp_before = st.lp.reshape(n, R).copy()
    old_gpu = B.loc_ray_scores(gw, obs_t, *pose).cpu().numpy().T                     # (n, R)
    old_orc = (-lp_before.astype(np.float64) - np.float64(LP_CONST)).astype(np.float32)
    out = {k_: v.cpu().numpy() for k_, v in
           B.loc_regenerate(gw, keys4, is_initial, obs_t, pose, means, want_scores=True).items()}
    fused = B.loc_mh(gw, np.asarray(keys4)[None], is_initial, obs_t, pose, means)    # production kernel, K = 1
    n_acc, acc_o, diag = O.loc_mh_diag(st, keys4, is_initial, obs_t, ow)

    for c, name in enumerate(("x", "y", "th")):                                     # same Threefry bits
        np.testing.assert_allclose(out[name], diag[:, c], rtol=RTOL, atol=ATOL_POS, err_msg=f"{tag} proposal {name}")
    np.testing.assert_allclose(out["log_u"], diag[:, 4], rtol=RTOL, atol=1e-6, err_msg=f"{tag} log u")

    tol = RTOL * diag[:, 5]
    dw = np.abs(out["weight"].astype(np.float64) - diag[:, 3].astype(np.float64))
    bad_w = np.isnan(dw) | (dw > tol)
    flagged = np.flatnonzero(bad_w)
    lines = []
    for i in flagged:
        ok_new, txt_new = justify_rays(O, ow, obs_t, (out["x"][i], out["y"][i], out["th"][i]), out["scores"][:, i], R=R)
        ok_old, txt_old = justify_rays(O, ow, obs_t, (old["x"][i], old["y"][i], old["th"][i]), old_gpu[i], R=R)
        lines.append(f"i={i} weight kernel {out['weight'][i]:.6g} oracle {diag[i, 3]:.6g} (bar {tol[i]:.2g}); "
                     f"proposal: {txt_new} | current: {txt_old}")
        assert ok_new and ok_old, f"{tag}: unjustified weight difference: " + lines[-1]
    report(f"{tag}: weights outside 1e-5, each a lidar discontinuity", lines)
    assert len(flagged) <= max(2, max_frac * n), (tag, len(flagged))
    # the per-ray scores (what extend stores for the move kernel): each a summand of the log-weight
    off_old = np.abs(old_gpu - old_orc) > RTOL * np.abs(lp_before).sum(axis=1, keepdims=True)
    assert off_old.any(axis=1).sum() <= max(2, max_frac * n), (tag, int(off_old.any(axis=1).sum()))

    acc_g = out["log_u"] < _min0(out["weight"])
    flips = np.flatnonzero(acc_g != acc_o.astype(bool))
    lines = []
    for i in flips:
        margin_o = abs(float(diag[i, 4]) - float(_min0(diag[i:i + 1, 3])[0]))
        margin_g = abs(float(out["log_u"][i]) - float(_min0(out["weight"][i:i + 1])[0]))
        slack = float(dw[i]) + abs(float(out["log_u"][i]) - float(diag[i, 4]))
        lines.append(f"i={i} |log u - log alpha| oracle {margin_o:.3g} kernel {margin_g:.3g}, difference of the two "
                     f"sides {slack:.3g}, discontinuity={bool(bad_w[i])}")
        assert bad_w[i] or margin_o <= slack * (1 + 1e-6) + 1e-12, f"{tag}: unjustified accept flip: " + lines[-1]
    report(f"{tag}: accept flips (decision inside the numerical difference)", lines)

    for j, name in enumerate(("x", "y", "th")):            # production kernel == regenerate + accept + select
        want = np.where(acc_g, out[name], old[name])
        got = fused[j].cpu().numpy()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (tag, name)
    return len(flagged), len(flips)


def _hmc_spread(O, ow, obs_t, is_initial, keys4, i, pose0, means, target, tol, eps, L, R):
    rows = []
    for j in JITTER:
        s1 = O.LocState(1, R)
        s1.x[0], s1.y[0], s1.th[0] = pose0[0] + j[0], pose0[1] + j[1], pose0[2] + j[2]
        s1.mx[0], s1.my[0], s1.mth[0] = means
        _, _, d = O.loc_hmc_diag(s1, keys4, eps, L, is_initial, obs_t, ow, offset=int(i))
        rows.append(d[0, :4])
    rows = np.stack(rows)
    strict, spread = within_spread(rows, target, tol)
    if strict:
        return True, spread, False
    # a trajectory that rides a discontinuity takes one of (at least) two branches, and inside each branch the result
    # keeps moving with the start: 125 perturbed starts sample that family, the kernel's rounding is one more draw
    # from it and may fall just beyond the sample's extremes
    ok, _ = within_spread(rows, target, tol, sample_margin=HMC_SAMPLE_MARGIN)
    return ok, spread, ok


HMC_SAMPLE_MARGIN = 0.25


def check_hmc_move(torch, B, O, gw, ow, st, keys4, eps, L, is_initial, obs_t, tag="hmc", max_frac=0.25):
    """One hmc move from the oracle's state.  Along a trajectory a ray changes its winning wall; the
    leapfrog step at which it does depends on the last bits of the pose, and the gradient jumps there:
    some particles end outside the bar (measured: 3 % at a step stage, 13 % at the initial stage, whose
    particles are spread over a whole room), each checked against the oracle's own sensitivity to 3e-6 of
    its start."""
    n, R = st.n, st.R
    g = gpu_state(torch, st)
    pose, means = (g["x"], g["y"], g["th"]), (g["mx"], g["my"], g["mth"])
    old = np.stack([st.x, st.y, st.th], axis=1).copy()
    mu = np.stack([st.mx, st.my, st.mth], axis=1).copy()
    out = {k_: v.cpu().numpy() for k_, v in B.loc_hmc_propose(gw, keys4, eps, L, is_initial, obs_t, pose, means).items()}
    fused = B.loc_hmc(gw, np.asarray(keys4)[None], eps, L, is_initial, obs_t, pose, means)
    n_acc, acc_o, diag = O.loc_hmc_diag(st, keys4, eps, L, is_initial, obs_t, ow)

    np.testing.assert_allclose(out["log_u"], diag[:, 4], rtol=RTOL, atol=1e-6, err_msg=f"{tag} log u")
    bad0 = np.abs(out["logp_cur"] - diag[:, 5]) > RTOL * np.abs(diag[:, 5])         # same pose on both sides
    # log alpha = [logp(q_L) + log N(-p_L)] - [logp(q_0) + log N(p_0)].  The kinetic terms are QUADRATIC in the
    # momentum, itself a sum of gradient terms held at 1e-5: they carry twice the relative error.
    tol = RTOL * (np.abs(diag[:, 5]) + np.abs(diag[:, 6]) + 2.0 * (np.abs(diag[:, 7]) + np.abs(diag[:, 8])))
    q_g = np.stack([out["x"], out["y"], out["th"]], axis=1)
    bad_q = ~np.isclose(q_g, diag[:, :3], rtol=RTOL, atol=ATOL_POS).all(axis=1)
    dla = np.abs(out["log_alpha"].astype(np.float64) - diag[:, 3].astype(np.float64))
    bad = bad0 | bad_q | np.isnan(dla) | (dla > tol) | (np.abs(out["logp_new"] - diag[:, 6]) > RTOL * np.abs(diag[:, 6]))
    flagged = np.flatnonzero(bad)
    lines, n_margin = [], 0
    for i in flagged:
        target = np.array([q_g[i, 0], q_g[i, 1], q_g[i, 2], out["log_alpha"][i]], np.float64)
        tol4 = np.array([ATOL_POS + RTOL * abs(t_) for t_ in target[:3]] + [tol[i]])
        ok, spread, margin_used = _hmc_spread(O, ow, obs_t, is_initial, keys4, i, old[i], mu[i], target, tol4, eps, L, R)
        n_margin += bool(margin_used)
        lines.append(f"i={i} log alpha kernel {out['log_alpha'][i]:.6g} oracle {diag[i, 3]:.6g} (bar {tol[i]:.2g}), end "
                     f"point off by {np.abs(q_g[i] - diag[i, :3]).max():.2g}; the oracle's own result moves by "
                     f"{spread[3]:.3g} (log alpha) / {spread[:3].max():.2g} (end point) within 3e-6 of the start: "
                     f"difference within that movement = {ok}"
                     + (f" (beyond the 125 sampled starts' extremes, inside +{HMC_SAMPLE_MARGIN:.0%} of their range)"
                        if margin_used else ""))
        assert ok, f"{tag}: unjustified hmc difference: " + lines[-1]
    report(f"{tag}: outside 1e-5, each within the movement of the oracle's own result within 3e-6 of the start "
           f"({n_margin} of them only with the sampling margin)", lines)
    assert len(flagged) <= max(2, max_frac * n), (tag, len(flagged))
    assert n_margin <= max(1, 0.01 * len(flagged)), (tag, "particles that needed the sampling margin", n_margin)

    acc_g = out["log_u"] < _min0(out["log_alpha"])
    flips = np.flatnonzero(acc_g != acc_o.astype(bool))
    lines = []
    for i in flips:
        margin_o = abs(float(diag[i, 4]) - float(_min0(diag[i:i + 1, 3])[0]))
        slack = float(dla[i]) + abs(float(out["log_u"][i]) - float(diag[i, 4]))
        lines.append(f"i={i} |log u - log alpha| oracle {margin_o:.3g}, difference of the two sides {slack:.3g}, "
                     f"trajectory through a discontinuity={bool(bad[i])}")
        assert bad[i] or margin_o <= slack * (1 + 1e-6) + 1e-12, f"{tag}: unjustified accept flip: " + lines[-1]
    report(f"{tag}: accept flips", lines)

    want = np.where(acc_g[:, None], q_g, old)
    got = np.stack([t.cpu().numpy() for t in fused], axis=1)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), tag
    return len(flagged), len(flips)


# --------------------------------------------------------------------------- #
# the same through the reference's Python surface: init / extend / resample / rejuvenate(mh | hmc)
# --------------------------------------------------------------------------- #
def teacher_forced_smc(torch, G, O, n=4096, K=5, stages=2, kernel="mh", world_kind="basic", verbose=True):
    """seed(rejuvenation_smc)(key, localization_model, mcmc_kernel = mh | hmc, systematic resampling)
    walked stage by stage through genjax_amd's init / extend / resample / rejuvenate, every stage
    started from the oracle's state (oracle.LocSMCStepper).  -> summary dict."""
    from genjax_amd import backend as B                     # the caller's product (this module is test infrastructure)
    from genjax_amd.families import localization as LOC
    from genjax_amd.inference import ParticleCollection, extend, hmc, init, mh, rejuvenate, resample
    const, seed, sel = G.const, G.seed, G.sel

    ow = O.basic_world() if world_kind == "basic" else O.complex_world()
    world = LOC.create_basic_multi_room_world() if world_kind == "basic" else LOC.create_complex_multi_room_world()
    gw = world.native(8)
    obs, k = O.localization_observations(ow)
    sp = O.LocSMCStepper(k, n, obs, ow, kernel=kernel, K=K)
    selection = sel("x") | sel("y") | sel("theta")
    move = (lambda tr: mh(tr, selection)) if kernel == "mh" else (lambda tr: hmc(tr, selection, sp.eps, sp.L))
    summary = dict(justified=0, flips=0, resampled=[], ess=[])

    def collection(t_index, logw):
        st = sp.st
        cols = [_dev(torch, getattr(st, c).copy()) for c in ("x", "y", "th", "mx", "my", "mth")]
        tr = LOC.LocTrace(LOC.localization_model, *cols, obs[t_index], t_index, world, 8)
        return ParticleCollection(tr, _dev(torch, logw.copy()), const(n))

    for t in range(stages):
        tag = f"stage {t}"
        ks = sp.stage_keys(t)
        key_state = ks.key.copy()
        cons = {"measurements": {"distance": obs[t]}}
        # ---- init / extend through the surface, from the oracle's state
        if t == 0:
            p = seed(init)(key_state, LOC.localization_model, (LOC.Pose(0.0, 0.0, 0.0), 0, world, const(8)), const(n), cons)
        else:
            pc = collection(t - 1, sp.st.logw)
            p = seed(extend)(key_state, pc, LOC.localization_model, pc.traces.get_retval(), cons)
        logw_in = sp.st.logw.copy()
        sp.extend(t, ks)                                   # the oracle's extend (advances ks by three splits)
        st = sp.st
        for c, name in (("x", "x"), ("y", "y"), ("th", "th")):
            np.testing.assert_allclose(getattr(p.traces, c).cpu().numpy(), getattr(st, name), rtol=RTOL, atol=ATOL_POS,
                                       err_msg=f"{tag} {name}")
        for c in ("mx", "my", "mth"):
            assert np.array_equal(getattr(p.traces, c).cpu().numpy(), getattr(st, c)), (tag, c)
        lp = st.lp.reshape(n, 8)
        tol = RTOL * (np.abs(logw_in) + np.abs(lp).sum(axis=1))
        got_logw = p.log_weights.cpu().numpy()
        d = np.abs(got_logw.astype(np.float64) - st.logw.astype(np.float64))
        flagged = np.flatnonzero(np.isnan(d) | (d > tol))
        sc = p.traces.ray_scores.cpu().numpy()
        lines = []
        for i in flagged:
            pose = tuple(float(getattr(p.traces, c)[i].item()) for c in ("x", "y", "th"))
            ok, txt = justify_rays(O, ow, obs[t], pose, sc[:, i])
            lines.append(f"i={i} logw kernel {got_logw[i]:.6g} oracle {st.logw[i]:.6g} (bar {tol[i]:.2g}); {txt}")
            assert ok, f"{tag}: unjustified log-weight difference: " + lines[-1]
        report(f"{tag}: log-weights outside 1e-5, each a lidar discontinuity", lines)
        assert len(flagged) <= max(2, 2e-3 * n)
        summary["justified"] += len(flagged)
        # ---- normaliser and the lax.cond decision, on the oracle's log-weights
        kc = ks.next()
        pc = collection(t, st.logw)
        lse, e = sp.normalise()
        np.testing.assert_allclose(float(pc.effective_sample_size().item()), e, rtol=RTOL, err_msg=f"{tag} ess")
        np.testing.assert_allclose(float(pc._norm_stats()[1].item()), lse, rtol=RTOL, err_msg=f"{tag} lse")
        take = e < float(n // 2)
        assert (float(pc.effective_sample_size().item()) < (n // 2)) == take, tag
        summary["ess"].append(e)
        summary["resampled"].append(bool(take))
        # ---- resample: ancestors and the gathered columns bit-exact given the oracle's weights and uniform
        if take:
            q = seed(resample)(kc, pc, "systematic")
            pre = {c: getattr(sp.st, c).copy() for c in ("x", "y", "th", "mx", "my", "mth")}
            lml_before = sp.lml_acc
            u, idx, w_o, cdf_o = sp.resample(kc, lse)
            # the contract: ancestors bit-exact GIVEN the reference's normalised weights and uniform
            anc_c = B.resample_systematic(_dev(torch, w_o), u).cpu().numpy()
            assert np.array_equal(anc_c, idx), (tag, "ancestors given the oracle's weights")
            # through the surface the weights are exp(logw - lse) with the LIBRARY's lse and exp (last-ulp
            # differences from libm): an ancestor may move where a position sits on a CDF step.  Each is listed
            # with its margin and must lie within 1e-5 (relative) of the step it crossed.
            anc_g = q.traces._pending[0].cpu().numpy()
            pos = ((np.arange(n, dtype=np.float32) + np.float32(u)) / np.float32(n)).astype(np.float32)
            lines = []
            for j in np.flatnonzero(anc_g != idx):
                lo, hi = sorted((int(anc_g[j]), int(idx[j])))
                margin = np.abs(cdf_o[lo:hi].astype(np.float64) - float(pos[j])).max()
                lines.append(f"j={j} ancestor surface {anc_g[j]} oracle {idx[j]}, |position - cdf step| {margin:.3g} "
                             f"(bar {RTOL * float(pos[j]):.3g})")
                assert margin <= RTOL * float(pos[j]), f"{tag}: unjustified ancestor difference: " + lines[-1]
            report(f"{tag}: ancestors that differ through the surface (position on a CDF step)", lines)
            # (expected count ~ n^2 x ulp: an lse off by one ulp scales every weight by 1 + 1.2e-7, which moves
            # the CDF by up to 1.2e-7 against a step spacing of 1 / n; 305 of 65536 measured)
            assert len(lines) <= max(2, 2e-2 * n), (tag, len(lines))
            summary["justified"] += len(lines)
            for c in ("x", "y", "th", "mx", "my", "mth"):             # leaf[indices]: an exact gather
                a, b = getattr(q.traces, c).cpu().numpy(), pre[c][anc_g]
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (tag, "gather", c)
            assert float(q.log_weights.abs().max().item()) == 0.0
            np.testing.assert_allclose(float(q.log_marginal_estimate.item()), float(sp.lml_acc - lml_before),
                                       rtol=RTOL, err_msg=f"{tag} lml increment")
        # ---- K moves, one at a time, each from the oracle's state; the surface's move == the fused kernel
        if kernel != "none" and K > 0:
            kr = ks.next()
            for m in range(K):
                keys4 = sp.move_keys(kr, m)
                pc = collection(t, sp.st.logw)
                out = seed(rejuvenate)(O.fold_in(kr, m), pc, move)          # surface: rejuvenate -> mh / hmc
                g = gpu_state(torch, sp.st)
                pose, means = (g["x"], g["y"], g["th"]), (g["mx"], g["my"], g["mth"])
                if kernel == "mh":
                    ref = B.loc_mh(gw, np.asarray(keys4)[None], t == 0, obs[t], pose, means)
                else:
                    ref = B.loc_hmc(gw, np.asarray(keys4)[None], sp.eps, sp.L, t == 0, obs[t], pose, means)
                for c, r in zip(("x", "y", "th"), ref):
                    assert torch.equal(getattr(out.traces, c), r), (tag, "move", m, c)
                if kernel == "mh":
                    j, f = check_mh_move(torch, B, O, gw, ow, sp.st, keys4, t == 0, obs[t], tag=f"{tag} mh move {m}")
                else:
                    j, f = check_hmc_move(torch, B, O, gw, ow, sp.st, keys4, sp.eps, sp.L, t == 0, obs[t],
                                          tag=f"{tag} hmc move {m}")
                summary["justified"] += j
                summary["flips"] += f
        if t == 0:
            sp.scan_key = ks.next()
    if verbose:
        print(f"teacher-forced {kernel} smc: {stages} stages x {n} particles x {K} moves at 1e-5: "
              f"{summary['justified']} particles individually justified, {summary['flips']} accept flips; "
              f"ESS {np.round(summary['ess'], 3)}, resampled {summary['resampled']}")
    return summary


# --------------------------------------------------------------------------- #
# secondary families, one kernel application at a time from the oracle's state
# --------------------------------------------------------------------------- #
def _ulp_jitter(v, k):
    """v moved by k units in the last place (f32)."""
    v = np.float32(v)
    for _ in range(abs(int(k))):
        v = np.nextafter(v, np.float32(np.inf if k > 0 else -np.inf), dtype=np.float32)
    return v


def check_outlier_iteration(torch, B, O, gm, om, st, eps, L, tag="gibbs+hmc", max_frac=3e-2):
    """One mixed_gibbs_hmc_kernel application (examples/curvefit/core.py:988-1014) per chain from the oracle's
    state `st` (advanced in place by the oracle).  Held at 1e-5 of their terms: the Gumbel-max margins of the
    Gibbs step, assess() at both ends of the trajectory, log alpha, the end point; chain keys bit-exact.  Every
    indicator flip and accept flip is listed with its margin, which must lie inside the numerical difference of
    the two sides; a chain outside a bar is judged against the SAME formulas in double precision on the same f32
    inputs and draws: the kernel may be no further from that value than 4 x the deviation of 16 double-precision
    replicas with f32-sized noise injected at every update, plus the bar (a randomised rounding-error estimate of
    THAT chain; the oracle's f32 build is itself up to 1.9e-4 relative away from f64 on 0.1 % of these chains.
    Chain 21 of the test, rate 0.11 -> 0.76 in one trajectory: log alpha -7.7142 in f64, -7.7116 oracle, -7.7163
    kernel and oracle built with fused multiply-adds).  The fused production launch must
    equal this single application bit for bit.  -> (justified chains, indicator flips, accept flips)."""
    n, P = st["a"].size, int(om.n_points)
    up = lambda: {"a": _dev(torch, st["a"].copy()), "b": _dev(torch, st["b"].copy()), "c": _dev(torch, st["c"].copy()),
                  "rate": _dev(torch, st["rate"].copy()), "keys": _dev(torch, st["keys"].copy().view(np.int32)),
                  "outliers": _dev(torch, st["outliers"].copy().view(np.int64))}                      # noqa: E731
    old = np.stack([st[k_].copy() for k_ in ("a", "b", "c", "rate")])                     # (4, n)
    before = {k_: v.copy() for k_, v in st.items()}
    g1, g2 = up(), up()
    dg = B.outlier_gibbs_hmc_diag(gm, g1, eps, L).cpu().numpy()
    B.outlier_gibbs_hmc(gm, g2, 1, eps, L)                                               # the production launch
    for k_ in g1:
        assert torch.equal(g1[k_], g2[k_]), (tag, "fused launch != single application", k_)
    _, do = O.outlier_gibbs_hmc_diag(om, st, eps, L)
    assert np.array_equal(g1["keys"].cpu().numpy().view(np.uint32), st["keys"]), (tag, "chain keys")

    # ---- Gibbs: margins, indicator flips
    mg, mo, mt = dg[10:10 + P], do[10:10 + P], do[10 + P:10 + 2 * P]
    both_inf = np.isinf(mg) & np.isinf(mo) & (np.sign(mg) == np.sign(mo))
    dm = np.where(both_inf, 0.0, np.abs(mg.astype(np.float64) - mo.astype(np.float64)))
    assert np.all(dm <= RTOL * mt + 1e-30), (tag, "Gibbs margin", float(np.nanmax(dm / np.maximum(mt, 1e-30))))
    mask_g = g1["outliers"].cpu().numpy().view(np.uint64)
    flips_ind = np.flatnonzero(mask_g != st["outliers"])
    lines = []
    for i in flips_ind:
        for j in range(P):
            if ((int(mask_g[i]) >> j) & 1) != ((int(st["outliers"][i]) >> j) & 1):
                lines.append(f"chain {i} point {j}: v1 - v0 oracle {mo[j, i]:.3g} kernel {mg[j, i]:.3g} "
                             f"(bar {RTOL * mt[j, i]:.2g})")
                assert abs(float(mo[j, i])) <= dm[j, i] * (1 + 1e-6) + 1e-12, f"{tag}: unjustified indicator flip: " + lines[-1]
    report(f"{tag}: indicator flips (Gumbel-max margin inside the numerical difference)", lines)
    same_mask = mask_g == st["outliers"]

    # ---- HMC: assess at both ends, log alpha, end point
    # a trajectory that leaves 0 < rate < 1 has log alpha = nan (log of a negative number) on both sides and is
    # rejected on both sides: nothing else of it is comparable
    degenerate = ~np.isfinite(do[4])
    assert not np.isfinite(dg[4][degenerate]).any(), (tag, "log alpha finite where the oracle's is not")
    terms = np.abs(do[6]) + np.abs(do[7]) + 2.0 * (np.abs(do[8]) + np.abs(do[9]))
    with np.errstate(invalid="ignore"):
        bad = np.abs(dg[4].astype(np.float64) - do[4]) > RTOL * terms
        bad |= np.abs(dg[6].astype(np.float64) - do[6]) > RTOL * np.abs(do[6])
        bad |= np.abs(dg[7].astype(np.float64) - do[7]) > RTOL * np.abs(do[7])
        bad |= ~np.isclose(dg[0:4], do[0:4], rtol=RTOL, atol=ATOL_POS).all(axis=0)
        bad |= ~np.isfinite(dg[4])
    bad &= ~degenerate
    np.testing.assert_allclose(dg[5], do[5], rtol=RTOL, atol=1e-6, err_msg=f"{tag} log u")
    flagged = np.flatnonzero(bad & same_mask)
    lines = []
    if flagged.size:
        # Judge of a chain that two f32 evaluations place on either side of the bar: the reference's formulas on
        # the same f32 inputs and draws in DOUBLE precision (the value every f32 evaluation approximates), and the
        # same double-precision trajectory with f32-sized relative perturbations injected at every update (16
        # randomised replicas: how far THIS chain's dynamics carry one rounding per operation).  The kernel passes
        # when it is no further from the clean f64 value than 4 x the largest deviation of those replicas, plus the
        # bar; the oracle's own f32 build is printed beside it (it meets the same criterion on every chain).
        # Such chains run into a wall of the rate's potential (1 / rate, 1 / (1 - rate)): ill-conditioned.
        js = {k_: before[k_][flagged].copy() for k_ in before}
        d64 = O.outlier_hmc_f64(om, js, st["outliers"][flagged], eps, L)
        with np.errstate(invalid="ignore"):
            env = np.abs(np.stack([O.outlier_hmc_f64(om, js, st["outliers"][flagged], eps, L, noise_seed=r_ + 1)
                                   for r_ in range(16)]) - d64[None]).max(axis=0)
        env[~np.isfinite(env)] = np.inf       # a replica left 0 < rate < 1: the chain grazes the edge of the domain
    for c_, i in enumerate(flagged):
        rows = [0, 1, 2, 3, 4, 7]
        t64 = abs(d64[6, c_]) + abs(d64[7, c_]) + 2.0 * (abs(d64[8, c_]) + abs(d64[9, c_]))
        bars = np.array([ATOL_POS + RTOL * abs(d64[r_, c_]) for r_ in rows[:4]] + [RTOL * t64, RTOL * abs(d64[7, c_])])
        e_kernel = np.abs(dg[rows, i].astype(np.float64) - d64[rows, c_])
        e_strict = np.abs(do[rows, i].astype(np.float64) - d64[rows, c_])
        ok = bool(np.all(e_kernel <= 4.0 * env[rows, c_] + bars))
        lines.append(f"chain {i} rate {before['rate'][i]:.6g} -> {d64[3, c_]:.6g}: log alpha in f64 {d64[4, c_]:.7g}, kernel "
                     f"{dg[4, i]:.7g}, oracle {do[4, i]:.7g} (bar {bars[4]:.2g}); distance from f64: kernel {e_kernel[4]:.2g}, "
                     f"oracle {e_strict[4]:.2g}, f64 replicas with f32-sized noise up to {env[4, c_]:.2g}; end point: "
                     f"{e_kernel[:4].max():.2g}, {e_strict[:4].max():.2g}, {env[:4, c_].max():.2g}: within 4x the replicas' "
                     f"deviation + bar = {ok}")
        assert ok, f"{tag}: unjustified hmc difference: " + lines[-1]
    report(f"{tag}: chains outside 1e-5 of the strict oracle build, each judged against the f64 evaluation", lines)
    assert len(flagged) <= max(2, max_frac * n), (tag, len(flagged))

    # ---- accept decisions and the selected state
    with np.errstate(invalid="ignore"):
        acc_g = dg[5] < _min0(dg[4])
        acc_o = do[5] < _min0(do[4])
    flips = np.flatnonzero((acc_g != acc_o) & same_mask)
    lines = []
    for i in flips:
        margin_o = abs(float(do[5, i]) - float(_min0(do[4, i:i + 1])[0]))
        slack = abs(float(dg[4, i]) - float(do[4, i])) + abs(float(dg[5, i]) - float(do[5, i]))
        lines.append(f"chain {i} |log u - log alpha| oracle {margin_o:.3g}, difference of the two sides {slack:.3g}, "
                     f"outside a bar = {bool(bad[i])}")
        assert bad[i] or margin_o <= slack * (1 + 1e-6) + 1e-12, f"{tag}: unjustified accept flip: " + lines[-1]
    report(f"{tag}: accept flips", lines)
    got = np.stack([g1[k_].cpu().numpy() for k_ in ("a", "b", "c", "rate")])
    want = np.where(acc_g[None, :], dg[0:4], old)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (tag, "state != select(accept, proposal, old)")
    return len(flagged), len(flips_ind), len(flips)


def check_lgssm_mh_move(torch, B, O, gm, om, kstate, ku, is_initial, obs_t, state, mean, tag="lgssm mh"):
    """One mh(trace, sel("state")) move (mcmc.py:267-309 on extras/state_space.py:377-431) from the oracle's state
    (`state` (ds, n), advanced in place by the oracle): proposals at 1e-5, every accept flip listed with its
    margin |log u - min(0, w)|, which must lie within 1e-5 of the two observation log-densities the weight is the
    difference of.  -> accept flips."""
    ds, n = state.shape
    cur = state.copy()
    mk = np.stack([np.asarray(kstate, np.uint32), np.asarray(ku, np.uint32)])[None]
    out = B.lgssm_mh(gm, mk, is_initial, obs_t, _dev(torch, cur), _dev(torch, mean.copy())).cpu().numpy()
    _, d = O.lgssm_mh_diag(om, kstate, ku, is_initial, obs_t, state, mean)
    prop, w, logu, terms = d[:, :ds].T, d[:, ds], d[:, ds + 1], d[:, ds + 2]
    acc_o = logu < _min0(w)
    moved = (out.view(np.uint32) != cur.view(np.uint32)).any(axis=0)
    # an accepted proposal that equals the current state bit for bit cannot be told from a rejection; none occurs
    acc_g = moved
    ok_prop = np.isclose(out, prop, rtol=RTOL, atol=ATOL_POS).all(axis=0)
    assert np.all(ok_prop[acc_g]), (tag, "accepted state is not the proposal", int((~ok_prop[acc_g]).sum()))
    flips = np.flatnonzero(acc_g != acc_o)
    lines = []
    for i in flips:
        margin = abs(float(logu[i]) - float(_min0(w[i:i + 1])[0]))
        lines.append(f"i={i} |log u - log alpha| oracle {margin:.3g} (bar {RTOL * terms[i]:.2g}), weight {w[i]:.6g}")
        assert margin <= RTOL * terms[i] + 1e-7 * abs(float(logu[i])), f"{tag}: unjustified accept flip: " + lines[-1]
    report(f"{tag}: accept flips (margin inside 1e-5 of the weight's terms)", lines)
    rejected_same = (out.view(np.uint32) == cur.view(np.uint32)).all(axis=0)
    assert np.all(rejected_same[~acc_g])
    return len(flips)


def check_gol_pass(torch, B, O, boards, target, keys_c, oy, ox, p_flip, tag="gol pass", ulps=4.0):
    """One colour pass (examples/gol/core.py:156-185) on every chain from the oracle's boards (`boards`
    (chains, n, n) uint8, advanced in place by the oracle): every cell that differs is listed with the margin
    of its decision |u - (1 - p_zero)|, which must be within a few ulp of 1 (softmax through different exp / divide
    roundings); all other cells bit-exact.  -> mismatching cells."""
    g = _dev(torch, boards.copy())
    B.gol_gibbs_pass(g, _dev(torch, target), keys_c, oy, ox, p_flip)
    got = g.cpu().numpy()
    lines = []
    for c in range(boards.shape[0]):
        m = O.gol_gibbs_pass_margins(boards[c], target, oy, ox, p_flip, keys_c[c])
        for (y, x) in np.argwhere(got[c] != boards[c]):
            assert (y - oy) % 3 == 0 and (x - ox) % 3 == 0, (tag, "a cell outside the colour class changed", c, y, x)
            mar = float(m[(y - oy) // 3, (x - ox) // 3])
            lines.append(f"chain {c} cell ({y}, {x}): u - (1 - p_zero) = {mar:.3g}")
            assert abs(mar) <= ulps * 1.1920929e-07, f"{tag}: unjustified cell difference: " + lines[-1]
    report(f"{tag}: cells that differ (decision within {ulps:g} ulp of its threshold)", lines)
    return len(lines)


def check_locopt_extend(torch, B, O, gw, ow, st, keys3, grid, radius, noise, obs_t, offset=0, tag="locally optimal"):
    """extend() with the locally-optimal grid proposal (examples/localization/core.py:730-809) from the oracle's
    state (advanced in place by the oracle).  A particle whose pose differs has picked another grid pose: it is
    listed with the two grid scores (the oracle's winner and the kernel's), whose gap must lie within 1e-5 of the
    terms the scores are sums of (an argmax tie in f32); every other particle is held at 1e-5.
    -> (argmax ties, log-weights justified as lidar discontinuities)."""
    n = st.n
    prev = np.stack([st.x, st.y, st.th]).copy()
    g = {name: _dev(torch, getattr(st, name).copy()) for name in ("x", "y", "th", "logw")}
    logw_in = st.logw.copy()
    out = B.loc_extend_locopt(gw, keys3, grid, noise, radius, obs_t, (g["x"], g["y"], g["th"]), g["logw"], offset=offset)
    O.loc_extend_locopt(st, keys3, grid, radius, noise, obs_t, ow, offset=offset)
    for name in ("mx", "my", "mth"):
        assert np.array_equal(out[name].cpu().numpy(), getattr(st, name)), (tag, name)
    got = {c: out[c].cpu().numpy() for c in ("x", "y", "th", "logw")}
    same = np.ones(n, bool)
    for c in ("x", "y", "th"):
        same &= np.isclose(got[c], getattr(st, c), rtol=RTOL, atol=2 * ATOL_POS)
    lines = []
    for i in np.flatnonzero(~same):
        sc, tm, ps = O.loc_locopt_grid_scores(grid, radius, prev[:, i], obs_t, ow, R=st.R)
        w_o = int(np.argmax(sc))
        shift = np.array([st.x[i], st.y[i], st.th[i]], np.float64) - ps[w_o]           # the noise the oracle added
        w_g = int(np.argmin(np.abs(ps - (np.array([got["x"][i], got["y"][i], got["th"][i]]) - shift)).sum(axis=1)))
        gap, bar = float(sc[w_o]) - float(sc[w_g]), RTOL * float(tm[w_o] + tm[w_g])
        lines.append(f"i={i}: grid pose oracle #{w_o} score {sc[w_o]:.7g}, kernel #{w_g} score {sc[w_g]:.7g}: gap "
                     f"{gap:.3g} (bar {bar:.2g})")
        assert w_g != w_o and 0.0 <= gap <= bar, f"{tag}: unjustified argmax difference: " + lines[-1]
    report(f"{tag}: particles on another grid pose (argmax tie within 1e-5 of the scores' terms)", lines)
    lp = st.lp.reshape(n, st.R)
    def absn(v, m, sd):
        return np.abs(-0.5 * ((v - m) / sd) ** 2 - np.log(sd) - 0.9189385332046727)

    prior = absn(st.x, st.mx, 0.25) + absn(st.y, st.my, 0.25) + absn(st.th, st.mth, 0.1)
    # the three proposal log-densities: |log N(z sd; 0, sd)| <= z^2 / 2 + |log sd| + 0.92 with |z| < 5.5
    tol = RTOL * (np.abs(logw_in) + np.abs(lp).sum(axis=1) + prior + 3.0 * (15.2 + abs(np.log(noise)) + 1.7))
    d = np.abs(got["logw"].astype(np.float64) - st.logw.astype(np.float64))
    off = np.flatnonzero(same & (np.isnan(d) | (d > tol)))
    jl = []
    for i in off:
        sc_i = B.loc_ray_scores(gw, obs_t, out["x"][i:i + 1], out["y"][i:i + 1], out["th"][i:i + 1]).cpu().numpy()[:, 0]
        ok, txt = justify_rays(O, ow, obs_t, (got["x"][i], got["y"][i], got["th"][i]), sc_i, R=st.R)
        jl.append(f"i={i} logw kernel {got['logw'][i]:.6g} oracle {st.logw[i]:.6g} (bar {tol[i]:.2g}); {txt}")
        assert ok, f"{tag}: unjustified log-weight difference: " + jl[-1]
    report(f"{tag}: log-weights outside 1e-5, each a lidar discontinuity", jl)
    return len(lines), len(jl)
