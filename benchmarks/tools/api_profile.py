"""cProfile of seed(rejuvenation_smc) on the localization model at a small particle count (host-side cost of a run).

    python benchmarks/tools/api_profile.py [n_particles]
"""
import cProfile
import os
import pstats
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    import genjax_amd as G
    from genjax_amd import const, seed, sel
    from genjax_amd.families.localization import (Pose, create_basic_multi_room_world, generate_ground_truth_data,
                                                  localization_model)
    from genjax_amd.inference import mh, rejuvenation_smc

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
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
                                      n_particles=const(n), n_rejuvenation_moves=const(10),
                                      resample_method=const("systematic"))

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(45)


if __name__ == "__main__":
    main()
