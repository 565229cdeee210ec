"""One rejuvenation-SMC step of the localization family at other ray counts than the headline's 8:
    python benchmarks/tools/rays_sweep.py [N] [rays,rays,...]        (GJX_LIB_PATH picks a variant library)
mh (K 10) and hmc (K 5, L 10), 8 / 16 rays by default, one GPU.  Prints one JSON line per case."""
import json
import sys
import time

import numpy as np
import torch

import genjax_amd as G
from genjax_amd.families.localization import create_basic_multi_room_world, generate_ground_truth_data
from genjax_amd.parallel import ShardedLocalizationSMC, ShardInfo


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    wld = create_basic_multi_room_world()
    ray_counts = [int(r) for r in sys.argv[2].split(",")] if len(sys.argv) > 2 else [8, 16]
    for rays in ray_counts:
        key = G.key(42)
        key, sub = G.split(key, 2)
        _, _, obs = generate_ground_truth_data(wld, sub, n_rays=rays)
        obs = np.stack(obs)
        for kernel, kw in (("mh", dict(n_moves=10)), ("hmc", dict(n_moves=5, step_size=0.01, n_steps=10))):
            smc = ShardedLocalizationSMC(wld.native(rays), obs, n, G.split(key, 2)[1], kernel=kernel,
                                         shard=ShardInfo(0, 1, n), device=torch.device("cuda", 0), **kw)
            smc.init()
            smc.step(1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            steps = 4
            for t in range(2, 2 + steps):
                smc.step(t)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            print(json.dumps({"rays": rays, "kernel": kernel, "n_particles": n, "ms_per_step": round(dt * 1e3, 4),
                              "log_ml": float(smc.log_marginal_likelihood())}), flush=True)


if __name__ == "__main__":
    main()
