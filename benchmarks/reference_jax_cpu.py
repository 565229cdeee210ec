#!/usr/bin/env python
"""Run on a machine that HAS the reference's environment (Python >= 3.12, jax 0.7.2, tensorflow-probability
0.25.0, the probcomp/genjax checkout on PYTHONPATH together with its `examples/` directory).  It cannot run
in the build container or on the GPU boxes and nothing in this repository imports it;
it has therefore never been executed -- treat the first run as its test.

It does the two things BASELINE.md section 2 leaves open:

  1. times the reference's own JAX/XLA CPU path on the headline workload, calling
     jax.jit(seed(rejuvenation_smc)) DIRECTLY (the run_smc_* wrappers of examples/localization/core.py
     post-process in O(N T) Python) with systematic resampling patched in (rejuvenation_smc hard-codes
     resample(p) = categorical, smc.py:714,750: an N x N Gumbel matrix), and prints one JSON line per N in
     the units of bench.py (particle-steps/s) with the core count;
  2. writes tests/golden/reference_jax.json: PRNG words, uniforms, normals, split / fold_in for key(42), and
     a 16-particle localization run (choices, log-weights, ESS trace, log-ML) under the same key schedule
     tests/golden/localization16.json pins for the restatement.  tests/test_oracle.py compares the
     restatement with that file when it exists (test_reference_fixture_if_present) -- committing it turns
     "parity unpinned" into pinned without a code change.

usage:  python benchmarks/reference_jax_cpu.py [--sizes 4096 65536 1048576] [--moves 10] [--out tests/golden]
"""
import argparse
import functools
import json
import os
import time


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", type=int, nargs="*", default=[1 << 12, 1 << 16, 1 << 20])
    ap.add_argument("--moves", type=int, default=10)
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                  "tests", "golden"))
    a = ap.parse_args()

    import jax
    import jax.numpy as jnp
    import jax.random as jrand
    import numpy as np
    from genjax import const, seed, sel
    from genjax.inference import mh, rejuvenation_smc
    import genjax.inference.smc as ref_smc
    from examples.localization.core import Pose, create_basic_multi_room_world, localization_model
    from examples.localization.data import generate_ground_truth_data

    # systematic resampling inside rejuvenation_smc (the `method` argument exists, smc.py:603-606)
    ref_smc.resample = functools.partial(ref_smc.resample, method="systematic")

    world = create_basic_multi_room_world()
    key = jrand.key(42)
    key, sub = jrand.split(key)
    _poses, _controls, observations = generate_ground_truth_data(world, sub, n_rays=8)
    obs = jnp.stack([jnp.asarray(o) for o in observations])
    smc_key = jrand.split(key)[1]
    selection = sel("x") | sel("y") | sel("theta")

    def run(n, k_moves):
        return seed(rejuvenation_smc)(
            smc_key, localization_model, mcmc_kernel=const(lambda tr: mh(tr, selection)),
            observations={"measurements": {"distance": obs}},
            initial_model_args=(Pose(x=0.0, y=0.0, theta=0.0), jnp.array(0), world, const(8)),
            n_particles=const(n), n_rejuvenation_moves=const(k_moves))

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    T = int(obs.shape[0])
    for n in a.sizes:
        fn = jax.jit(functools.partial(run, n, a.moves))
        out = fn()
        jax.block_until_ready(out.log_weights)              # compile + first run
        times = []
        for _ in range(a.repeats):
            t0 = time.perf_counter()
            out = fn()
            jax.block_until_ready(out.log_weights)
            times.append(time.perf_counter() - t0)
        best = min(times)
        print(json.dumps({
            "metric": "particle-steps/sec, localization rejuvenation-SMC (extend+ESS+resample+%d MH moves)" % a.moves,
            "kind": "reference JAX/XLA CPU", "jax": jax.__version__, "backend": jax.default_backend(),
            "cores": cores, "n_particles": n, "stages": T, "seconds": best,
            "value": n * T / best, "unit": "particle-steps/s",
            "log_ml": float(out.log_marginal_likelihood())}), flush=True)

    # ---- golden vectors -------------------------------------------------------------------------
    def bits_of(x):
        return [int(v) for v in np.asarray(x, np.float32).view(np.uint32).reshape(-1)]

    k42 = jrand.key(42)
    gold = {
        "versions": {"jax": jax.__version__},
        "key42": [int(v) for v in np.asarray(jrand.key_data(k42))],
        "split42_3": np.asarray(jrand.key_data(jrand.split(k42, 3))).tolist(),
        "fold_in42_7": [int(v) for v in np.asarray(jrand.key_data(jrand.fold_in(k42, 7)))],
        "bits42_8": [int(v) for v in np.asarray(jrand.bits(k42, (8,), dtype=jnp.uint32))],
        "uniform42_8_bits": bits_of(jrand.uniform(k42, (8,))),
        "normal42_8_bits": bits_of(jrand.normal(k42, (8,))),
        "cumsum37_bits": bits_of(jnp.cumsum(jrand.uniform(jrand.key(7), (37,)))),
        "observations_bits": bits_of(obs),
    }
    n16 = 16
    p = jax.jit(lambda: seed(rejuvenation_smc)(
        smc_key, localization_model, mcmc_kernel=const(lambda tr: mh(tr, selection)),
        observations={"measurements": {"distance": obs[:4]}},
        initial_model_args=(Pose(x=0.0, y=0.0, theta=0.0), jnp.array(0), world, const(8)),
        n_particles=const(n16), n_rejuvenation_moves=const(2)))()
    ch = p.traces.get_choices()
    gold["localization16"] = {
        "n": n16, "stages": 4, "moves": 2, "resampling": "systematic",
        "x_bits": bits_of(ch["x"]), "y_bits": bits_of(ch["y"]), "theta_bits": bits_of(ch["theta"]),
        "log_weights": [float(v) for v in np.asarray(p.log_weights)],
        "log_ml": float(p.log_marginal_likelihood()),
        "ess": float(p.effective_sample_size()),
    }
    os.makedirs(a.out, exist_ok=True)
    path = os.path.join(a.out, "reference_jax.json")
    with open(path, "w") as f:
        json.dump(gold, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
