"""Is there a one-time host stall after some number of launches in any process? (diagnostic)"""
import sys, time, torch
torch.cuda.set_device(0)
x = torch.zeros(1 << 20, device="cuda")
torch.cuda.synchronize()
mode = sys.argv[1] if len(sys.argv) > 1 else "async"
big = []
for i in range(3000):
    t0 = time.perf_counter()
    x.add_(1.0)
    if mode == "sync" or i % 15 == 14:
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    if dt > 2.0:
        big.append((i, round(dt, 2)))
print(mode, "stalls (launch, ms):", big)
