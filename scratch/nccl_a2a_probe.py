"""Does all_to_all_single accept empty tensors / zero splits under NCCL? (one rank; diagnostic)"""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
buf_s, buf_r = torch.arange(64, dtype=torch.float32, device="cuda"), torch.zeros(64, device="cuda")
dist.all_to_all_single(buf_r[:0], buf_s[:0], [0], [0]); torch.cuda.synchronize(); print("empty ok")
dist.all_to_all_single(buf_r[:15], buf_s[:15], [15], [15]); torch.cuda.synchronize(); print("self ok", float(buf_r[:15].sum()))
out = torch.zeros(5, device="cuda"); dist.all_gather_into_tensor(out, buf_s[:5]); print("all_gather ok", out.tolist())
dist.all_reduce(buf_r); torch.cuda.synchronize(); print("all_reduce ok")
dist.barrier(); dist.destroy_process_group(); print("done")
