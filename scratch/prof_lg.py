"""cProfile of the LG-SSM particle filter through the Python surface (diagnostic)."""
import cProfile, pstats, sys, os, math
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genjax_amd as G
from genjax_amd import const, seed
from genjax_amd.families.state_space import linear_gaussian
from genjax_amd.inference import rejuvenation_smc
torch.cuda.set_device(0)
T, n = 100, 1 << 20
rng = np.random.default_rng(1)
ys = rng.normal(size=T).astype(np.float32)
f = lambda a: np.asarray(a, np.float32)
params = (f([0.0]), f([[1.0]]), f([[0.9]]), f([[0.1]]), f([[1.0]]), f([[0.2]]))
args = (np.zeros(1, np.float32), 0) + params
def run():
    p = seed(rejuvenation_smc)(G.key(7), linear_gaussian, None, None, {"obs": ys}, args, const(n), const(False), const(0), const("systematic"))
    torch.cuda.synchronize(); return p
run(); run()
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
