import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import oracle
from genjax_amd import backend as B
def ulp_diff(a, b):
    a = np.asarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.asarray(b, np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a); b = np.where(b < 0, -(b & 0x7FFFFFFF), b)
    return np.abs(a - b)
tot = 0; mx = 0; hist = np.zeros(16, int)
for s in range(16):
    k = oracle.split(oracle.key(7 + s), 2)[1]
    n = 1 << 22
    got = B.random_normal(k, n).cpu().numpy(); ref = oracle.normal(k, n)
    d = ulp_diff(got, ref); mx = max(mx, d.max()); tot += n
    hist += np.bincount(np.minimum(d, 15), minlength=16)
    big = np.flatnonzero(d > 2)
    if len(big): print("seed", s, "examples", [(float(got[i]), float(ref[i]), int(d[i])) for i in big[:5]])
print("n", tot, "max ulp", mx, "hist", hist)
