import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import oracle
from genjax_amd import backend as B
from test_gpu_teacher_forced import _world, _start_state, _gpu_state, _min0
for kind in ("basic", "complex"):
    ow, gw = _world(B, oracle, kind)
    obs, k = oracle.localization_observations(ow)
    n, K, eps, L = 20000, 3, 0.01, 10
    st, t_obs = _start_state(oracle, ow, obs, k, n, False)
    mk = np.stack([oracle.split(oracle.fold_in(k, 100 + m), 4) for m in range(K)])
    for m in range(K):
        g = _gpu_state(st)
        pose, means = (g["x"], g["y"], g["th"]), (g["mx"], g["my"], g["mth"])
        out = {k_: v.cpu().numpy() for k_, v in B.loc_hmc_propose(gw, mk[m], eps, L, False, t_obs, pose, means).items()}
        n_acc, acc_o, diag = oracle.loc_hmc_diag(st, mk[m], eps, L, False, t_obs, ow)
        scale = np.abs(diag[:, 5]) + np.abs(diag[:, 6]) + 2.0 * (np.abs(diag[:, 7]) + np.abs(diag[:, 8]))
        dla = np.abs(out["log_alpha"].astype(np.float64) - diag[:, 3])
        r = dla / scale
        q_g = np.stack([out["x"], out["y"], out["th"]], axis=1)
        dq = np.abs(q_g - diag[:, :3]).max(axis=1)
        print(kind, m, "acc", n_acc, "rel log_alpha err pct [50,90,99,99.9,max]", np.nanpercentile(r, [50, 90, 99, 99.9, 100]),
              "n>1e-5", (r > 1e-5).sum(), "n>1e-4", (r > 1e-4).sum(), "dq pct", np.percentile(dq, [50, 99, 99.9, 100]))
        big = np.flatnonzero(r > 1e-5)
        for i in big[:8]:
            print("   i", i, "la", out["log_alpha"][i], diag[i, 3], "logp0", diag[i, 5], "logpL", diag[i, 6], "ke0", diag[i, 7], "keL", diag[i, 8], "dq", dq[i], "acc_o", acc_o[i])
        acc_g = out["log_u"] < _min0(out["log_alpha"])
        print("   flips", (acc_g != acc_o.astype(bool)).sum())
