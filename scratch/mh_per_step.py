"""Move-launch duration per step for the fused step (gather + scores in the launch) and without (diagnostic)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genjax_amd as G
from genjax_amd import backend as B
from genjax_amd.families.localization import create_basic_multi_room_world, generate_ground_truth_data
from genjax_amd.parallel import ShardedLocalizationSMC, ShardInfo
n = 1 << 24
torch.cuda.set_device(0)
wld = create_basic_multi_room_world()
key = G.key(42); key, sub = G.split(key, 2)
_, _, obs = generate_ground_truth_data(wld, sub, n_rays=8); obs = np.stack(obs)
for mode in ("fused", "noscores", "forced_sharded_kernels"):
    smc = ShardedLocalizationSMC(wld.native(8), obs, n, G.split(key, 2)[1], kernel="mh", n_moves=10,
                                 shard=ShardInfo(0, 1, n), device=torch.device("cuda", 0))
    smc.store_scores = mode == "fused"
    ev = []
    def mk():
        p = (B.new_event(), B.new_event()); ev.append(p); return p
    smc.move_events = mk
    if mode == "forced_sharded_kernels":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
        dist.init_process_group("gloo", rank=0, world_size=1)
        smc.force_sharded = True
    smc.init()
    for t in range(1, 17):
        smc.step(t)
    torch.cuda.synchronize()
    print(mode, " ".join("%.2f" % B.event_elapsed_ms(a, b) for a, b in ev), " ess:", " ".join("%.2g" % e for e in smc.state.ess))
