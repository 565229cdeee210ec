#!/usr/bin/env python
"""bench.py -- particle-steps/s of the localization rejuvenation-SMC hot path (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

Workload (BASELINE.json configs[3], the config the metric is quoted on): examples/localization
model, basic 9-wall world, R = 8 lidar rays, 17 waypoint observations (seed 42), N = 2^24
particles IN TOTAL (sharded over the ranks: strong scaling), systematic resampling when
ESS < N/2, K = 10 fused MH rejuvenation moves (examples/localization/core.py:611).
One "step" = one smc_step (smc.py:731-767) over all particles: extend -> normalise/ESS ->
conditional resample (cumsum, bisection, gather) -> 10 MH moves.  The particle set is resident
in HBM when the timed region starts; observations cycle through t = 1..16.
value = total particles * steps / max-over-ranks wall time.

Also reported on the same JSON line:
  roofline     -- the dominant kernel (k_loc_mh, all 10 moves in one launch), HIP-event timed on
                  the launch stream inside the timed region; algorithmic bytes = 36 B/particle
                  (read x,y,theta + 3 means, write x,y,theta).  The kernel is VALU-bound
                  (SURVEY.md section 0 finding 5), so the HBM fraction is small by construction;
                  the `valu` sub-object prices the same launch against the f32 vector peak.
  cpu_baseline -- the CPU restatement (oracle/, kind "port") of the same SMC run on a bounded
                  sample, OpenMP over the host cores available to this process (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import gc
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_TOTAL_DEFAULT = 1 << 24
MH_BYTES_PER_PARTICLE = 36.0   # read x,y,th,mx,my,mth (24 B) + write x,y,th (12 B), all moves fused
# single-GPU mh launches that follow a resampling carry the gather: + ancestor index (4 B) + write of the means (12 B)
MH_GATHER_BYTES_PER_PARTICLE = 52.0
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
VALU_PEAK_TFLOPS = 157.3       # f32 vector peak
# static VALU instruction estimate per particle for one launch is measured by rocprof
# (profiles/); bench reports lane-op throughput from the counted flops below.
FLOPS_PER_LIDAR = 8 * (9 * 19 + 40)      # R x (W x ~19 + boundary/sincos ~40) f32 ops
FLOPS_PER_MOVE = FLOPS_PER_LIDAR + 4 * 110 + 60


def pmc_traffic(kernel_prefix: str, n_local: int):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/pmc_traffic.json, written by scratch/pmc_summary.py: FETCH_SIZE x2 per the gfx950
    correction + WRITE_SIZE; collected at N = 2^24 on one GPU).  None when the sizes differ."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if n_local != N_TOTAL_DEFAULT or not os.path.exists(path):
        return None
    try:
        with open(path) as f:
            d = json.load(f)
        for name, v in d.items():
            if name.startswith(kernel_prefix):
                return float(v["bytes"])
    except Exception:
        return None
    return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=8,
                    help="untimed steps; the GPU clocks need ~8 move launches (~25 ms) to ramp after the idle set-up")
    ap.add_argument("--particles", type=int, default=N_TOTAL_DEFAULT, help="TOTAL particles over all GPUs")
    ap.add_argument("--moves", type=int, default=10)
    ap.add_argument("--kernel", default="mh", choices=["mh", "hmc"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-particles", type=int, default=1 << 21)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the measured configuration); gloo = rehearsal of the multi-rank "
                         "flow on fewer GPUs than ranks (ranks share devices, transport staged through the host)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="diagnostic: run the sharded step (collectives included) even with one rank")
    return ap.parse_args()


def cpu_baseline(n_cpu: int, moves: int):
    """Time the CPU restatement on a bounded sample of the same workload."""
    import oracle as O

    O.lib()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = int(os.environ.get("OMP_NUM_THREADS", min(cores, 16)))
    threads = max(1, min(threads, cores))
    os.environ["OMP_NUM_THREADS"] = str(threads)
    w = O.basic_world()
    obs, k = O.localization_observations(w)
    O.loc_smc(k, 1024, obs, w, kernel="mh", K=moves, method="systematic")  # warm-up
    t0 = time.perf_counter()
    O.loc_smc(k, n_cpu, obs, w, kernel="mh", K=moves, method="systematic")
    dt = time.perf_counter() - t0
    steps = obs.shape[0] - 1
    return {
        "value": n_cpu * steps / dt,
        "unit": "particle-steps/s",
        "cores": threads,
        "kind": "port",
        "sample": f"oracle/ C restatement, full rejuvenation_smc run: N={n_cpu} particles x {steps} smc_steps "
                  f"(+init), MH K={moves}, systematic resampling, {dt:.1f} s wall",
    }


def main():
    a = parse()
    # stdout carries exactly ONE line, the JSON result: libraries that print to the C-level stdout
    # (RCCL writes a five-line version banner there when a communicator is created) are diverted to
    # stderr for the whole run, and the line is written to the saved descriptor at the end
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    if a.backend == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist

    if world > 1 or a.force_sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    import genjax_amd as G
    from genjax_amd.families.localization import create_basic_multi_room_world, generate_ground_truth_data
    from genjax_amd.parallel import ShardedLocalizationSMC, ShardInfo

    n_total = a.particles
    if n_total % (world * 2048) != 0:
        raise SystemExit("--particles must be a multiple of 2048 * gpus")
    wld = create_basic_multi_room_world()
    key = G.key(42)
    key, sub = G.split(key, 2)
    _, _, obs = generate_ground_truth_data(wld, sub, n_rays=8)
    obs = np.stack(obs)
    smc_key = G.split(key, 2)[1]
    T = obs.shape[0]
    shard = ShardInfo(rank, world, n_total)
    smc = ShardedLocalizationSMC(wld.native(8), obs, n_total, smc_key, kernel=a.kernel, n_moves=a.moves,
                                 shard=shard, device=dev)

    # time the dominant kernel with events on the launch stream (torch's current stream)
    from genjax_amd import backend as B

    mh_events = []
    orig_mh = B.loc_mh
    timing_on = {"on": False}

    def timed_mh(*args, **kw):
        # sharded path (world > 1): the move kernel is launched through B.loc_mh / B.loc_hmc
        if not timing_on["on"]:
            return orig_mh(*args, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()                      # torch's current stream IS the launch stream (backend._stream)
        out = orig_mh(*args, **kw)
        e1.record()
        mh_events.append((e0, e1))
        return out

    def fused_events():
        # single-GPU path: gjx_loc_smc_resample_moves records the pair around the move kernel
        if not timing_on["on"]:
            return None
        pair = (B.new_event(), B.new_event())
        mh_events.append(pair)
        return pair

    if a.kernel == "mh":
        B.loc_mh = timed_mh
    else:
        orig_mh = B.loc_hmc
        B.loc_hmc = timed_mh
    smc.move_events = fused_events
    smc.force_sharded = a.force_sharded

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    smc.init()                        # particles resident in HBM before the timed region
    # A full (generation-2) pass of CPython's cyclic GC over the ~10^6 objects torch / numpy leave alive
    # takes 35-40 ms and fires once, a fixed number of allocations into the run (measured: step 28 of
    # every process but the first on a box) -- ten steps' worth of GPU idle at N = 2^24.  Standard
    # practice for latency-bound loops: collect now, then freeze the survivors out of the GC's reach.
    gc.collect()
    gc.freeze()
    step_no = 0

    def one_step():
        nonlocal step_no
        t = 1 + (step_no % (T - 1))
        smc.step(t, scan_index=step_no)
        step_no += 1

    for _ in range(a.warmup):
        one_step()
    sync_all()
    timing_on["on"] = True
    t0 = time.perf_counter()
    for _ in range(a.steps):
        one_step()
    sync_all()
    dt = time.perf_counter() - t0
    timing_on["on"] = False
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    n_local = shard.n_local
    # rows a rank fetched from other shards per resampling: mean over the timed steps, max over ranks
    rows_mean = float(np.mean(smc.remote_rows[-max(1, a.steps):])) if smc.remote_rows else 0.0
    rows_max_rank = rows_mean
    if world > 1:
        t_rows = torch.tensor([rows_mean], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(t_rows, op=dist.ReduceOp.MAX)
        rows_max_rank = float(t_rows.item())
    mh_ms = [e0.elapsed_time(e1) if isinstance(e0, torch.cuda.Event) else B.event_elapsed_ms(e0, e1)
             for e0, e1 in mh_events]
    mh_avg_s = (sum(mh_ms) / max(1, len(mh_ms))) * 1e-3
    resampled = sum(1 for r in smc.state.resampled[-a.steps:] if r)

    if rank == 0:
        # bytes of the timed launches.  Single GPU, mh: a launch that follows a resampling carries the
        # gather (+ ancestor index 4 B + write of the means 12 B), and the launch reads the per-ray scores
        # extend stored (R x 4 B) instead of re-solving the lidar of the starting pose
        sharded = world > 1 or a.force_sharded
        fused_gather = a.kernel == "mh" and smc.fused and (not sharded or smc.fast_resamples > 0)
        stored_scores = fused_gather and smc.store_scores
        frac_res = resampled / max(1, a.steps)
        if sharded:   # share of the resamplings that took the halo path (the others gather separately)
            frac_res *= smc.fast_resamples / max(1, sum(1 for r in smc.state.resampled if r))
        bytes_pp = MH_BYTES_PER_PARTICLE
        if fused_gather:
            bytes_pp += (MH_GATHER_BYTES_PER_PARTICLE - MH_BYTES_PER_PARTICLE) * frac_res
        if stored_scores:
            bytes_pp += 4.0 * 8
        achieved = bytes_pp * n_local / mh_avg_s / 1e9 if mh_avg_s > 0 else 0.0
        flops = (0 if stored_scores else FLOPS_PER_LIDAR) + a.moves * FLOPS_PER_MOVE
        valu_t = flops * n_local / mh_avg_s / 1e12 if mh_avg_s > 0 else 0.0
        line = {
            "metric": "particle-steps/sec, localization rejuvenation-SMC (extend+ESS+resample+10 MH moves)",
            "value": n_total * a.steps / dt,
            "unit": "particle-steps/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "examples/localization rejuvenation_smc, basic world (9 walls), 8 rays, 17 waypoint "
                            f"observations cycled, N={n_total} particles total, systematic resampling, "
                            f"{a.kernel} K={a.moves}",
                "particles_total": n_total,
                "particles_per_gpu": n_local,
                "parallelism": f"particle-shard x{world}",
                "resampled_steps": resampled,
                "collectives": ("none" if world == 1 else
                                f"{a.backend}: per step all_gather(5 f32), and when resampling all_gather(N/2048 f32), "
                                "all_reduce(coarse table), all_to_all_single(rows outside the own shard)"),
                "halo_path_resamplings": smc.fast_resamples if sharded else None,
                "rows_from_other_shards_per_resample_rank0": rows_mean,
                "rows_from_other_shards_per_resample_max_over_ranks": rows_max_rank,
            },
            "roofline": {
                "kernel": f"k_loc_{a.kernel}<8> ({a.moves} fused moves per launch"
                          + (", resampling gather fused into its loads" if fused_gather else "")
                          + (", starting scores read from extend's store)" if stored_scores else ")"),
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic(f"k_loc_{a.kernel}<", n_local) if a.moves == 10 else None,
                "traffic_note": "HBM bytes per launch, rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, "
                                "separate passes, profiles/r01_bench_mh_pmc_fetch_write.csv",
                "algorithmic_bytes_per_launch": bytes_pp * n_local,
                "avg_launch_ms": mh_avg_s * 1e3,
                "algorithmic_bytes_per_particle": bytes_pp,
                "valu": {"note": "kernel is VALU-bound: f32 ops of the REFERENCE formulation (R x (W x 19 + 40) per lidar, "
                                 "4 draws per move) / launch time vs the 157.3 TFLOP/s vector peak; the kernel itself "
                                 "executes fewer (paired rays, axis-aligned line solve)",
                         "achieved_tflops": valu_t, "peak_tflops": VALU_PEAK_TFLOPS,
                         "frac": valu_t / VALU_PEAK_TFLOPS},
            },
        }
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(a.cpu_particles, a.moves)
        os.write(result_fd, (json.dumps(line) + "\n").encode())
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
