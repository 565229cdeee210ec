"""One-time stall vs kernarg bytes: loops a launch with a 1.7 KB by-value argument block (diagnostic)."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genjax_amd import backend as B
from genjax_amd.families.localization import create_basic_multi_room_world
torch.cuda.set_device(0)
w = create_basic_multi_room_world().native(8)
n = 1 << 16
x = torch.full((n,), 3.0, device="cuda"); y = torch.full((n,), 3.0, device="cuda"); th = torch.zeros(n, device="cuda")
B.loc_lidar(w, x, y, th); torch.cuda.synchronize()
big = []
t_start = time.perf_counter()
for i in range(4000):
    t0 = time.perf_counter()
    B.loc_lidar(w, x, y, th)
    if i % 8 == 7:
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    if dt > 2.0:
        big.append((i, round(dt, 2), round((t0 - t_start) * 1e3)))
print("stalls (launch, ms, at ms):", big)
