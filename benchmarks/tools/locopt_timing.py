"""Time of gjx_loc_extend_locopt (the locally-optimal grid proposal, examples/localization/core.py:730-809) in its two
forms -- a workgroup per particle, a thread per particle -- over the particle counts the reference uses it with.

    python benchmarks/tools/locopt_timing.py
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    import genjax_amd as G
    from genjax_amd import backend as B
    from genjax_amd.families.localization import create_basic_multi_room_world, generate_ground_truth_data

    dev = torch.device("cuda", 0)
    wld = create_basic_multi_room_world()
    key = G.key(42)
    key, sub = G.split(key, 2)
    _, _, obs = generate_ground_truth_data(wld, sub, n_rays=8)
    gw = wld.native(8)
    keys3 = G.split(key, 3)
    rng = np.random.default_rng(0)
    for n in (5, 100, 1000, 10000, 100000, 1 << 20):
        prev = [torch.from_numpy(c.astype(np.float32)).to(dev) for c in
                (rng.uniform(1.0, 11.0, n), rng.uniform(1.0, 9.0, n), rng.uniform(-3.0, 3.0, n))]
        logw = torch.zeros(n, device=dev)
        for g in (15, 25):
            if n * g ** 3 > 3e10:
                continue
            row = {"n_particles": n, "grid": g, "lidar_solves": n * (g ** 3 + 1)}
            for form, env in (("workgroup_per_particle", "1"), ("thread_per_particle", "0")):
                if env == "0" and n * g ** 3 > 4e9 and n < 50000:
                    continue                                   # minutes of one-wave work: not worth the GPU time
                os.environ["GJX_LOCOPT_COOP"] = env
                B.loc_extend_locopt(gw, keys3, g, 0.15, 2.0, obs[3], prev, logw)
                torch.cuda.synchronize()
                reps = 3
                t0 = time.perf_counter()
                for _ in range(reps):
                    B.loc_extend_locopt(gw, keys3, g, 0.15, 2.0, obs[3], prev, logw)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / reps * 1e3
                row[form + "_ms"] = ms
                row[form + "_solves_per_s"] = row["lidar_solves"] / (ms * 1e-3)
            print(json.dumps(row), flush=True)
    os.environ.pop("GJX_LOCOPT_COOP", None)


if __name__ == "__main__":
    main()
