"""Per-step wall times of the single-GPU step driver (diagnostic)."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genjax_amd as G
from genjax_amd.families.localization import create_basic_multi_room_world, generate_ground_truth_data
from genjax_amd.parallel import ShardedLocalizationSMC, ShardInfo
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 21
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 80
torch.cuda.set_device(0)
wld = create_basic_multi_room_world()
key = G.key(42); key, sub = G.split(key, 2)
_, _, obs = generate_ground_truth_data(wld, sub, n_rays=8); obs = np.stack(obs)
smc = ShardedLocalizationSMC(wld.native(8), obs, n, G.split(key, 2)[1], kernel="mh", n_moves=10,
                             shard=ShardInfo(0, 1, n), device=torch.device("cuda", 0))
smc.init(); torch.cuda.synchronize()
time.sleep(float(os.environ.get('SETTLE_SLEEP', '0')))
T = obs.shape[0]
ts = []
for s in range(steps):
    t0 = time.perf_counter()
    smc.step(1 + (s % (T - 1)), scan_index=s)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("ms/step:", " ".join(f"{t:.2f}" for t in ts))
print("ess:", " ".join(f"{e:.3g}" for e in smc.state.ess[1:]))
