import sys; sys.path.insert(0,'.')
import numpy as np, torch, oracle as O
from genjax_amd import backend as B
k = O.split(O.key(7), 2)[1]
n = 1<<18
got = B.random_normal(k, n).cpu().numpy(); ref = O.normal(k, n)
lo = np.nextafter(np.float32(-1), np.float32(0))
u = O.uniform(k, n, float(lo), 1.0)
a = got.view(np.int32).astype(np.int64); b = ref.view(np.int32).astype(np.int64)
d = np.abs(a-b)
idx = np.argsort(-d)[:12]
for i in idx:
    x = np.float64(u[i]); w = -np.log1p(-np.float32(u[i]*u[i]).astype(np.float64))
    print(i, "u=%.9g" % u[i], "w=%.7f" % w, "gpu=%.9g ref=%.9g ulp=%d" % (got[i], ref[i], d[i]))
print("hist", np.bincount(np.minimum(d, 10)))
