"""Writes tests/golden/*.json.

The reference (JAX 0.7.2 + TFP 0.25.0, Python >= 3.12) cannot be imported in the build container
(SURVEY.md section 8c), so these vectors are produced by the CPU restatement in oracle/ -- they pin
the restatement (regression) and give the GPU tests fixed inputs/outputs; they are NOT outputs of
the reference itself.  The only third-party known answers are in kat.json (Random123 Threefry
KATs and the jax.random.split(PRNGKey(0)) value printed in the JAX documentation).

Run from the repository root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle as O  # noqa: E402


def f32list(a):
    return [float(np.float32(v)) for v in np.asarray(a).reshape(-1)]


def bits(a):
    return [int(v) for v in np.asarray(a, np.float32).view(np.uint32).reshape(-1)]


def main():
    kat = {
        "threefry2x32_20": [
            {"key": [0, 0], "ctr": [0, 0], "out": [0x6B200159, 0x99BA4EFE]},
            {"key": [0xFFFFFFFF, 0xFFFFFFFF], "ctr": [0xFFFFFFFF, 0xFFFFFFFF], "out": [0x1CB996FC, 0xBB002BE7]},
            {"key": [0x13198A2E, 0x03707344], "ctr": [0x243F6A88, 0x85A308D3], "out": [0xC4923A9C, 0x483DF7A0]},
        ],
        "jax_docs_legacy_split_prngkey0": [[4146024105, 967050713], [2718843009, 1272950319]],
    }
    json.dump(kat, open(os.path.join(HERE, "kat.json"), "w"), indent=1)

    k = O.key(42)
    prng = {
        "key42": [int(v) for v in k],
        "split42_3": O.split(k, 3).tolist(),
        "fold_in42_7": [int(v) for v in O.fold_in(k, 7)],
        "bits42_8": [int(v) for v in O.random_bits(k, 8)],
        "bits42_offset_2p32m2_4": [int(v) for v in O.random_bits(k, 4, (1 << 32) - 2)],
        "uniform42_8_bits": bits(O.uniform(k, 8)),
        "normal42_8": f32list(O.normal(k, 8)),
    }
    json.dump(prng, open(os.path.join(HERE, "prng.json"), "w"), indent=1)

    # one 16-particle localization run-in: init, extend, normaliser, systematic resampling
    w = O.basic_world()
    obs, smc_key = O.localization_observations(w)
    seq = O.Seed(smc_key)
    st = O.LocState(16, 8)
    keys0 = [seq.next() for _ in range(3)]
    O.loc_extend(st, keys0, True, obs[0], w)
    init = {n: f32list(getattr(st, n)) for n in ("x", "y", "th", "logw")}
    keys1 = [seq.next() for _ in range(3)]
    O.loc_extend(st, keys1, False, obs[1], w)
    step = {n: f32list(getattr(st, n)) for n in ("x", "y", "th", "mx", "my", "mth", "logw")}
    lse, ess = O.logsumexp(st.logw), O.ess(st.logw)
    wts = O.normalized_weights(st.logw)
    u = 0.37454012
    anc, cdf = O.systematic_indices(wts, u, return_cdf=True)
    mk = O.split(O.fold_in(smc_key, 1000), 4)
    st_mh = st.copy()
    n_acc, acc = O.loc_mh(st_mh, mk, False, obs[1], w)
    hk = O.split(O.fold_in(smc_key, 2000), 4)
    st_hmc = st.copy()
    n_acc_h, acc_h = O.loc_hmc(st_hmc, hk, 0.01, 10, False, obs[1], w)
    logp, grad = O.loc_logp_grad(st, False, obs[1], w)
    loc = {
        "smc_key": [int(v) for v in smc_key],
        "obs": [f32list(o) for o in obs],
        "keys_init": np.asarray(keys0).tolist(), "keys_step": np.asarray(keys1).tolist(),
        "init": init, "step": step, "lse": lse, "ess": ess,
        "weights_bits": bits(wts), "u": u, "cdf_bits": bits(cdf), "ancestors": anc.tolist(),
        "mh_keys": mk.tolist(), "mh_accept": acc.tolist(), "mh_x": f32list(st_mh.x),
        "hmc_keys": hk.tolist(), "hmc_accept": acc_h.tolist(), "hmc_x": f32list(st_hmc.x),
        "logp": f32list(logp), "grad": f32list(grad),
    }
    json.dump(loc, open(os.path.join(HERE, "localization16.json"), "w"), indent=1)

    # cumsum association on an awkward length
    rng = np.random.default_rng(1)
    e = rng.exponential(size=37).astype(np.float32)
    json.dump({"in_bits": bits(e), "out_bits": bits(O.cumsum(e))}, open(os.path.join(HERE, "cumsum37.json"), "w"),
              indent=1)
    # config 3: eight outlier-mixture chains, init + 3 Gibbs/HMC iterations
    xs, ys, idx = O.outlier_dataset()
    m = O.outlier_model(xs, ys, alpha=1.0, beta=3.0)
    st = O.outlier_init(m, O.key(12345), 8)
    init3 = {k_: f32list(st[k_]) for k_ in ("a", "b", "c", "rate")}
    init3["outliers"] = [int(v) for v in st["outliers"]]
    init3["keys"] = st["keys"].tolist()
    acc = O.outlier_gibbs_hmc(m, st, 3, 0.01, 20)
    fin3 = {k_: f32list(st[k_]) for k_ in ("a", "b", "c", "rate")}
    fin3["outliers"] = [int(v) for v in st["outliers"]]
    fin3["keys"] = st["keys"].tolist()
    json.dump({"xs": f32list(xs), "ys_bits": bits(ys), "outlier_idx": idx, "init": init3, "after3": fin3,
               "accepted": acc}, open(os.path.join(HERE, "outlier8.json"), "w"), indent=1)

    # config 5: one 12 x 12 Game-of-Life chain, init + 2 frames of 2 sweeps
    rng = np.random.default_rng(4)
    target = O.gol_step((rng.random((12, 12)) < 0.35).astype(np.uint8))
    boards = O.gol_run(O.key(1), target, 0.03, 2, 2)
    json.dump({"target": target.tolist(), "boards": boards.tolist(),
               "n_incorrect": [O.gol_summary(b, target, 0.03)[0] for b in boards]},
              open(os.path.join(HERE, "gol12.json"), "w"), indent=1)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
