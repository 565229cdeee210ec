"""Host-side cost of every call of the sharded step (one rank, RCCL group of one): wall time the Python thread
spends inside each backend / collective call, averaged over the steps of a few runs.  The GPU is kept busy
(calls only enqueue), so these are enqueue costs, not kernel times.

    python benchmarks/tools/host_cost.py [--particles 2097152] [--runs 6]
"""
import argparse
import collections
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--particles", type=int, default=1 << 21)
    ap.add_argument("--runs", type=int, default=6)
    ap.add_argument("--no-overlap", action="store_true")
    a = ap.parse_args()
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    import genjax_amd as G
    from genjax_amd import backend as B
    from genjax_amd import parallel as P
    from genjax_amd.families.localization import create_basic_multi_room_world, generate_ground_truth_data

    wld = create_basic_multi_room_world()
    key = G.key(42)
    key, sub = G.split(key, 2)
    _, _, obs = generate_ground_truth_data(wld, sub, n_rays=8)
    obs = np.stack(obs)
    smc = P.ShardedLocalizationSMC(wld.native(8), obs, a.particles, G.split(key, 2)[1], kernel="mh", n_moves=10,
                                   shard=P.ShardInfo(0, 1, a.particles), device=dev)
    smc.force_sharded = True
    smc.overlap = not a.no_overlap
    smc.run()                                           # buffers, communicator
    cost, calls = collections.defaultdict(float), collections.defaultdict(int)
    on = {"on": False}

    def wrap(owner, name, label):
        fn = getattr(owner, name)

        def timed(*args, **kw):
            t0 = time.perf_counter()
            try:
                return fn(*args, **kw)
            finally:
                if on["on"]:
                    cost[label] += time.perf_counter() - t0
                    calls[label] += 1

        setattr(owner, name, timed)

    for name in ("loc_shard_extend", "shard_tile_sums", "shard_stats_fold", "shard_cdf_begin", "shard_cdf_finish",
                 "shard_search", "shard_pack", "shard_split", "shard_moves_main", "shard_moves_fixup",
                 "shard_resample_moves", "loc_move_keys", "shard_send_rows", "shard_recv_rows"):
        if hasattr(B, name):
            wrap(B, name, "B." + name)
    for name in ("all_gather_into", "all_to_all"):
        wrap(smc.comm, name, "comm." + name)
    for name in ("_read_begin", "_read_end", "_read2", "_host_u", "_global_stats", "_stage_sharded"):
        wrap(smc, name, "smc." + name)
    on["on"] = True
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.runs):
        smc.reset()
        smc.run()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    stages = a.runs * obs.shape[0]
    print(f"wall per stage {wall / stages * 1e6:8.1f} us   ({stages} stages, overlap={smc.overlap})")
    for k_, v in sorted(cost.items(), key=lambda kv: -kv[1]):
        print(f"{k_:28s} {v / stages * 1e6:8.1f} us per stage   {calls[k_] / stages:5.2f} calls per stage   "
              f"{v / max(1, calls[k_]) * 1e6:7.1f} us per call")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
