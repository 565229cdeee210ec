import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import oracle as O
import genjax_amd as G
from genjax_amd import const, seed, sel
from genjax_amd.inference import rejuvenation_smc, mh, hmc
from genjax_amd.families.localization import *

ow = O.basic_world()
obs_o, k = O.localization_observations(ow)
world = create_basic_multi_room_world()
key = G.key(42)
key, sub = G.split(key, 2)
poses, _, obs = generate_ground_truth_data(world, sub)
obs = np.stack(obs)
print("obs max diff vs oracle", np.abs(obs - obs_o).max())
key, smc_key = G.split(key, 2)
assert np.array_equal(smc_key, k)
for N_ in (4096, 65536):
    for kern, K in (("none", 0), ("mh", 10), ("hmc", 3)):
        mk = None
        if kern == "mh":
            mk = const(lambda tr: mh(tr, sel("x") | sel("y") | sel("theta")))
        if kern == "hmc":
            mk = const(lambda tr: hmc(tr, sel("x") | sel("y") | sel("theta"), 0.01, 10))
        t0 = time.time()
        p = seed(rejuvenation_smc)(smc_key, localization_model, mcmc_kernel=mk,
              observations={"measurements": {"distance": obs_o}},
              initial_model_args=(Pose(0.0, 0.0, 0.0), 0, world, const(8)),
              n_particles=const(N_), n_rejuvenation_moves=const(K), resample_method=const("systematic"))
        lml = float(p.log_marginal_likelihood().item())
        t1 = time.time()
        r = O.loc_smc(k, N_, obs_o, ow, kernel=kern, K=K, n_steps=10, method="systematic")
        t2 = time.time()
        print(N_, kern, "gpu lml", lml, "oracle", r["lml"], "gpu %.2fs cpu %.2fs" % (t1 - t0, t2 - t1))
        print("  ess gpu", np.array(p.diagnostics["ess"])[:6])
        print("  ess orc", r["ess"][:6])
        if kern != "none":
            print("  acc gpu", p.diagnostics["accept_counts"][1].cpu().numpy()[:5], "orc", r["accept"][1][:5])
