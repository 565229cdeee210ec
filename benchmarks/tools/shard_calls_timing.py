"""Time of every front-end call of the sharded step for ONE rank of a G-rank run, on one GPU (the other ranks' shares of
the gathered buffers are copies of this rank's: sizes and launch shapes are those of the real run, values plausible).
HIP events around each call, median of 25: launch overheads included (an empty launch is ~6 us here).

    python benchmarks/tools/shard_calls_timing.py [log2 n per rank = 21]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genjax_amd import backend as B  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 21
n = 1 << lg


def med(fn, reps=30):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return float(np.median(ts[5:]))


for G in (1, 2, 4, 8):
    if n * G >= 1 << 31:
        break
    rank = G // 2
    buf = B.new_shard_buffers(n, n * G, G, rank, torch.device("cuda", 0), n_rays=8)
    g = torch.Generator(device="cuda").manual_seed(1)
    buf.logw.copy_(-torch.rand(n, device="cuda", generator=g).log_().abs_() * 3 - 4)   # a few heavy particles
    mx = B.shard_local_max(buf)
    buf.max_all.copy_(mx.reshape(-1)[:1].expand(G))
    out = {}
    out["tile_sums"] = med(lambda: B.shard_tile_sums(buf))
    buf.tiles_all.view(G, -1).copy_(buf.tiles.view(1, -1).expand(G, -1))
    out["stats_fold"] = med(lambda: B.shard_stats_fold(buf))
    out["cdf_begin"] = med(lambda: B.shard_cdf_begin(buf))
    buf.msg_all.view(G, -1).copy_(buf.msg.view(1, -1).expand(G, -1))
    out["cdf_finish"] = med(lambda: B.shard_cdf_finish(buf))
    out["search+count"] = med(lambda: B.shard_search(buf, 0.37))
    out["split"] = med(lambda: B.shard_split(buf, 10))
    print(f"G = {G} (rank {rank}, n = 2^{lg} per rank): " + ", ".join(f"{k} {v:.1f}" for k, v in out.items())
          + f"  | sum {sum(out.values()):.1f} us")
    del buf
