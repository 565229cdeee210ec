#!/usr/bin/env python
"""Timings of the BASELINE.json configs other than the headline one (bench.py covers configs[3]
with MH).  One JSON line per config on stdout; run on one MI355X:  python benchmarks/configs.py

  configs[0]  curvefit importance sampling (8 points): N = 1e5 (the reference's CPU case) and 2^24
  configs[1]  linear-Gaussian bootstrap PF, T = 100, N = 2^20, systematic resampling, vs Kalman
  configs[2]  outlier-mixture curvefit, enumerative Gibbs + HMC, 2^20 chains x 20 iterations
  configs[3'] localization rejuvenation-SMC with the HMC kernel (eps 0.01, L = 10, K = 25)
  configs[4]  Game-of-Life inversion, 1024 x 1024 board, 64 batched chains, Gibbs sweeps
Timing protocol = src/genjax/timing.py:37-137: warm-up runs, then repeats, synchronised.
"""
import gc
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import genjax_amd as G  # noqa: E402
from genjax_amd import const, seed  # noqa: E402


def timed(fn, repeats=5, warmup=2):
    for _ in range(warmup):
        fn()
    gc.collect()
    gc.freeze()          # a generation-2 GC pass costs 35-40 ms once torch is loaded; keep it out of the timings
    torch.cuda.synchronize()
    ts = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return out, float(np.mean(ts)), float(np.std(ts))


def emit(**kw):
    print(json.dumps(kw), flush=True)


def config0():
    from genjax_amd.families.curvefit import npoint_curve
    from genjax_amd.inference import init

    xs = np.linspace(0.0, 1.0, 8).astype(np.float32)
    ys = (0.3 - 0.5 * xs + 0.7 * xs**2).astype(np.float32)
    cons = {"ys": {"obs": torch.from_numpy(ys).cuda()}}
    for n in (100000, 1 << 24):
        p, mean, std = timed(lambda: seed(init)(G.key(42), npoint_curve, (xs,), const(n), cons))
        emit(config="curvefit importance sampling, 8 points", n_particles=n, ms=mean * 1e3, ms_std=std * 1e3,
             particles_per_s=n / mean, log_ml=float(p.log_marginal_likelihood()))


def config1():
    from genjax_amd.families.state_space import kalman_filter, linear_gaussian
    from genjax_amd.inference import rejuvenation_smc

    T, n = 100, 1 << 20
    rng = np.random.default_rng(1)
    x, ys = rng.normal(0.0, 1.0), []
    for t in range(T):
        if t > 0:
            x = 0.9 * x + rng.normal(0.0, math.sqrt(0.1))
        ys.append(x + rng.normal(0.0, math.sqrt(0.2)))
    ys = np.asarray(ys, np.float32)
    f = lambda a: np.asarray(a, np.float32)  # noqa: E731
    params = (f([0.0]), f([[1.0]]), f([[0.9]]), f([[0.1]]), f([[1.0]]), f([[0.2]]))
    args = (np.zeros(1, np.float32), 0) + params

    def run():
        return seed(rejuvenation_smc)(G.key(7), linear_gaussian, None, None, {"obs": ys}, args, const(n),
                                      const(False), const(0), const("systematic"))

    p, mean, std = timed(run, repeats=3, warmup=1)
    _, _, exact = kalman_filter(ys.reshape(-1, 1), *params)
    emit(config="linear-Gaussian bootstrap PF, systematic resampling", T=T, n_particles=n, ms=mean * 1e3,
         ms_std=std * 1e3, particle_steps_per_s=n * T / mean, log_ml=float(p.log_marginal_likelihood()),
         kalman_log_ml=float(exact), resampled_steps=int(sum(p.diagnostics["resampled"])))


def config2():
    from genjax_amd import backend as B
    from genjax_amd.families.curvefit import run_gibbs_hmc

    n, iters = 1 << 20, 20
    xs = np.linspace(0, 1, 18).astype(np.float32)
    ys = (-0.211 - 0.395 * xs + 0.673 * xs * xs).astype(np.float32)
    k = G.key(42)
    k, sub = G.split(k, 2)
    ys = ys + 0.1 * B.random_normal(sub, 18).cpu().numpy()
    idx = [3, 6, 9, 13, 16]
    k, sub = G.split(k, 2)
    ys[idx] = 1.2 + 0.2 * B.random_normal(sub, 5).cpu().numpy()
    res, mean, std = timed(lambda: run_gibbs_hmc(xs, ys, G.key(12345), n, iters), repeats=3, warmup=1)
    emit(config="outlier-mixture curvefit, enumerative Gibbs + HMC (eps 0.01, L 20)", n_chains=n,
         iterations=iters, ms=mean * 1e3, ms_std=std * 1e3, chain_iterations_per_s=n * iters / mean,
         accept_rate=res.accept_rate,
         outlier_probabilities=[round(float(v), 3) for v in res.outlier_probabilities().cpu()])


def config3_hmc():
    from genjax_amd.families.localization import create_basic_multi_room_world, generate_ground_truth_data
    from genjax_amd.parallel import ShardedLocalizationSMC, ShardInfo

    n, K = int(os.environ.get("GJX_CFG3_N", 1 << 24)), 25
    wld = create_basic_multi_room_world()
    key = G.key(42)
    key, sub = G.split(key, 2)
    _, _, obs = generate_ground_truth_data(wld, sub, n_rays=8)
    obs = np.stack(obs)
    smc = ShardedLocalizationSMC(wld.native(8), obs, n, G.split(key, 2)[1], kernel="hmc", n_moves=K,
                                 step_size=0.01, n_steps=10, shard=ShardInfo(0, 1, n),
                                 device=torch.device("cuda", 0))
    smc.init()
    smc.step(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 4
    for t in range(2, 2 + steps):
        smc.step(t)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    emit(config="localization rejuvenation-SMC, HMC kernel (eps 0.01, L 10, K 25), basic world, 8 rays",
         n_particles=n, ms_per_step=dt * 1e3, particle_steps_per_s=n / dt,
         lidar_gradient_evaluations_per_s=n * K * 11 / dt)


def config3_api():
    """The headline workload (bench.py) driven through the drop-in Python surface instead of the
    step driver: seed(rejuvenation_smc)(key, localization_model, mcmc_kernel=mh, ...), T = 17."""
    from genjax_amd import sel
    from genjax_amd.families.localization import (Pose, create_basic_multi_room_world, generate_ground_truth_data,
                                                  localization_model)
    from genjax_amd.inference import mh, rejuvenation_smc

    n, K = int(os.environ.get("GJX_CFG3API_N", 1 << 24)), 10     # the reference runs it with 200 ... 5000 particles
    wld = create_basic_multi_room_world()
    key = G.key(42)
    key, sub = G.split(key, 2)
    _, _, obs = generate_ground_truth_data(wld, sub, n_rays=8)
    obs = np.stack(obs)
    s = sel("x") | sel("y") | sel("theta")
    smc_key = G.split(key, 2)[1]

    def run():
        return seed(rejuvenation_smc)(smc_key, localization_model, mcmc_kernel=const(lambda tr: mh(tr, s)),
                                      observations={"measurements": {"distance": obs}},
                                      initial_model_args=(Pose(0.0, 0.0, 0.0), 0, wld, const(8)),
                                      n_particles=const(n), n_rejuvenation_moves=const(K),
                                      resample_method=const("systematic"))

    p, mean, std = timed(run, repeats=3, warmup=1)
    T = obs.shape[0]
    emit(config="localization rejuvenation-SMC, MH K 10, through seed(rejuvenation_smc) (Python surface), T 17",
         n_particles=n, ms_total=mean * 1e3, ms_std=std * 1e3, ms_per_stage=mean * 1e3 / T,
         particle_steps_per_s=n * T / mean, log_ml=float(p.log_marginal_likelihood()))


def config4_gol():
    from genjax_amd import backend as B
    from genjax_amd.families.gol import GibbsSampler, chain_keys

    n, chains = 1024, 64
    rng = np.random.default_rng(1)
    truth = torch.from_numpy((rng.random((n, n)) < 0.3).astype(np.uint8)).cuda()
    target = B.gol_step(truth[None])[0]
    sampler = GibbsSampler(target, 0.03)
    ck = chain_keys(G.key(1), chains)
    sweeps = 10
    sched = [B.gol_keys(k, 1, sweeps) for k in ck]
    state = sampler.get_initial_state(np.stack([s_[0] for s_ in sched]))
    pk = np.stack([s_[1] for s_ in sched], axis=3)[0]             # (sweeps, 9, B, 2)
    pk_dev = torch.from_numpy(np.ascontiguousarray(pk).view(np.int32)).cuda()
    first = state.n_incorrect_bits().double().mean().item()

    def run():
        for g in range(sweeps):
            sampler.update_state(state, pk_dev[g])

    _, mean, std = timed(run, repeats=3, warmup=1)
    last = state.n_incorrect_bits().double().mean().item()
    cells = chains * n * n
    emit(config="Game-of-Life inversion, 1024 x 1024, 9-colour Gibbs sweeps, batched chains", n_chains=chains,
         sweeps=sweeps, ms_per_sweep=mean * 1e3 / sweeps, ms_std=std * 1e3 / sweeps,
         cell_updates_per_s=cells * sweeps / mean,
         bound="valu (bit-packed boards: ~0.8 B of HBM traffic per cell and sweep; profiles/r02_config4_gol_counters.txt)",
         mean_incorrect_bits_initial=first, mean_incorrect_bits_after=last)


if __name__ == "__main__":
    torch.cuda.set_device(0)
    which = sys.argv[1:] or ["0", "1", "2", "3", "3api", "4"]
    for w in which:
        {"0": config0, "1": config1, "2": config2, "3": config3_hmc, "3api": config3_api, "4": config4_gol}[w]()
