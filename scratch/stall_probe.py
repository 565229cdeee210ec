"""Which host call stalls?  Times every backend call of the fused step (diagnostic)."""
import sys, time, os, gc
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genjax_amd as G
from genjax_amd import backend as B
from genjax_amd.families.localization import create_basic_multi_room_world, generate_ground_truth_data
from genjax_amd.parallel import ShardedLocalizationSMC, ShardInfo
n, steps = 1 << 21, 80
torch.cuda.set_device(0)
wld = create_basic_multi_room_world()
key = G.key(42); key, sub = G.split(key, 2)
_, _, obs = generate_ground_truth_data(wld, sub, n_rays=8); obs = np.stack(obs)
VAR = os.environ.get("VAR", "base")
n = int(os.environ.get("NP", n))
smc = ShardedLocalizationSMC(wld.native(8), obs, n, G.split(key, 2)[1], kernel="hmc" if VAR == "hmc" else "mh",
                             n_moves=0 if VAR == "nomh" else (1 if VAR == "hmc" else 10), n_steps=2,
                             shard=ShardInfo(0, 1, n), device=torch.device("cuda", 0))
smc.store_scores = VAR != "noscores"
smc.fused = VAR != "granular"
log = []
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); log.append((name, (time.perf_counter() - t0) * 1e3)); return r
    setattr(obj, name, g)
for nm in ("loc_smc_extend", "loc_move_keys", "loc_smc_resample_moves"):
    wrap(B, nm)
wrap(smc, "_read")
smc.init(); torch.cuda.synchronize()
T = obs.shape[0]
gc_counts = []
for s in range(steps):
    log.clear()
    t0 = time.perf_counter()
    smc.step(1 + (s % (T - 1)), scan_index=s)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if (t2 - t0) > float(os.environ.get('THRESH_MS', '2')) * 1e-3:
        print("step", s, "total %.2f ms, step() %.2f ms, final sync %.2f ms" % ((t2 - t0) * 1e3, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
        print("   calls:", ", ".join("%s %.2f" % x for x in log), " gc:", gc.get_count())
print("done", VAR, n)
