"""One-time stall vs runtime blit operations (memset / D2H copies), in a warm process (diagnostic)."""
import sys, time, ctypes, torch
torch.cuda.set_device(0)
mode = sys.argv[1]
hip = ctypes.CDLL("libamdhip64.so")
x = torch.zeros(1 << 21, device="cuda"); s5 = torch.zeros(5, device="cuda")
pin = torch.empty(5, pin_memory=True)
torch.cuda.synchronize()
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
big = []
for i in range(3000):
    t0 = time.perf_counter()
    if mode == "memset":
        hip.hipMemsetAsync(ctypes.c_void_p(x.data_ptr()), 0, ctypes.c_size_t(x.numel() * 4), stream)
    elif mode == "d2h_pinned":
        pin.copy_(s5, non_blocking=True)
    elif mode == "d2h_pageable":
        s5.cpu()
    elif mode == "mix":
        x.add_(1.0)
        hip.hipMemsetAsync(ctypes.c_void_p(x.data_ptr()), 0, ctypes.c_size_t(x.numel() * 4), stream)
        x.add_(1.0)
        s5.cpu()
    if i % 4 == 3:
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    if dt > 2.0:
        big.append((i, round(dt, 2)))
print(mode, "stalls (iteration, ms):", big)
