#!/usr/bin/env python
"""bench.py -- particle-steps/s of the localization rejuvenation-SMC hot path (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, or
   started plain, in which case it launches its own N ranks that way before touching a GPU: launch_ranks())

Workload (BASELINE.json configs[3], the config the metric is quoted on): examples/localization
model, basic 9-wall world, R = 8 lidar rays, the reference's 17 waypoint observations (seed 42),
N = 2^24 particles IN TOTAL (sharded over the ranks: strong scaling), systematic resampling when
ESS < N/2, K = 10 fused MH rejuvenation moves (examples/localization/core.py:611).

One "step" = one smc_step (smc.py:731-767) over all particles: extend -> normalise/ESS ->
conditional resample (cumsum, bisection, gather) -> 10 MH moves.  The timed region runs WHOLE
reference-shaped runs back to back, each one seed(rejuvenation_smc) on the same key: the init stage
(init -> ESS -> resample -> moves, smc.py:705-729) followed by the 16 smc_steps of the scan.  K counts
the smc_steps; the init stages inside the region are timed but NOT counted (the metric is
N (T - 1) / wall).  Every run is also timed on its own and aggregated as the
reference's timing() does (src/genjax/timing.py:37-137: trimmed mean, 20 % off either end).
`--steady-state` is the older mode: no re-initialisation, observations t = 1..16 cycled.
value = total particles * K / max-over-ranks wall time; particles are resident in HBM throughout.

Also on the JSON line:
  roofline        the dominant kernel (k_loc_mh, all moves of a step in one launch), HIP-event timed on the
                  launch stream inside the timed region.  It is VALU-bound (SURVEY.md section 0 finding 5):
                  achieved = executed VALU wave-instructions (SQ_INSTS_VALU of the committed rocprofv3 pass,
                  profiles/r03_sq_counters.json) x 64 lanes x 2 flop / launch time against the 157.3 TFLOP/s
                  f32 vector peak, i.e. frac = share of the VALU issue slots (one wave-instruction per 2
                  cycles per SIMD) the launch uses; `hbm` prices the same launch's algorithmic bytes.
  roofline_stages the other stages of a step, HIP-event timed in one extra untimed run after the region
                  (gjx_ctx_set_stage_events): normalise / scan are HBM-bound, priced on algorithmic bytes.
  cpu_baseline    the CPU restatement (oracle/, kind "port") of the same run on a bounded sample,
                  OpenMP over the host cores available to this process (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_TOTAL_DEFAULT = 1 << 24
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
VALU_PEAK_TFLOPS = 157.3       # f32 vector peak = 1024 SIMDs x 32 lanes x 2 flop x 2.4 GHz
VALU_PEAK_WAVE_INSTS = 1024 * 2.4e9 / 2.0   # one wave64 instruction per 2 cycles per SIMD
# algorithmic HBM bytes per particle
MH_BYTES = 36.0                # read x,y,th,mx,my,mth (24 B) + write x,y,th (12 B), all moves fused
MH_GATHER_EXTRA = 16.0         # after a resampling: + ancestor index (4 B) + the means written (12 B)
STAGE_BYTES = {"extend": 16.0 + 28.0, "normalise": 4.0, "scan": 12.0, "search": 4.0}
PROFILE_FILES = {"sq": "profiles/r03_sq_counters.json", "pmc": "profiles/r03_pmc_traffic.json"}


def load_profile(which: str):
    path = os.path.join(ROOT, PROFILE_FILES[which])
    try:
        with open(path) as f:
            return json.load(f)
    except Exception:
        return None


def profile_entry(d, prefix):
    if not d:
        return None
    for name, v in d.get("kernels", d).items():
        if name.startswith(prefix):
            return v
    return None


def trimmed_mean(samples, trim=0.2):
    """_aggregate of the reference's timing() (src/genjax/timing.py:24-34)."""
    a = np.sort(np.asarray(samples, np.float64))
    k = int(a.size * trim)
    if 0.0 < trim < 0.5 and a.size > 2 and k > 0 and 2 * k < a.size:
        a = a[k:-k]
    return float(a.mean())


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=160, help="timed smc_steps (16 per reference-shaped run)")
    ap.add_argument("--warmup", type=int, default=17,
                    help="untimed stages (17 = one whole run); the GPU clocks need ~8 move launches to ramp")
    ap.add_argument("--particles", type=int, default=N_TOTAL_DEFAULT, help="TOTAL particles over all GPUs")
    ap.add_argument("--moves", type=int, default=10)
    ap.add_argument("--kernel", default="mh", choices=["mh", "hmc"])
    ap.add_argument("--hmc-steps", type=int, default=10)
    ap.add_argument("--steady-state", action="store_true",
                    help="no re-initialisation: observations t = 1..16 cycled (round-1 mode)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stage-profile", action="store_true")
    ap.add_argument("--cpu-particles", type=int, default=1 << 21)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the measured configuration); gloo = rehearsal of the multi-rank "
                         "flow on fewer GPUs than ranks (ranks share devices, transport staged through the host)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="diagnostic: run the sharded step (collectives included) even with one rank")
    ap.add_argument("--device-decision", dest="device_decision", action="store_true", default=None,
                    help="one-GPU step as ONE call with the ESS decision taken on the device (default: on for at most "
                         "2^22 particles per GPU, where the host round trip shows; off above)")
    ap.add_argument("--host-decision", dest="device_decision", action="store_false",
                    help="one-GPU step as two calls around the ESS read-back, whatever the size")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="diagnostic (runs without a GPU): the ranks only meet over gloo and rank 0 prints a stub line; "
                         "checks that `bench.py --gpus N` starts its own ranks")
    ap.add_argument("--no-overlap", action="store_true",
                    help="diagnostic: sharded mh step in its two-phase form (all moves after the row exchange)")
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


def launch_ranks(a) -> int:
    """`python bench.py --gpus N` started WITHOUT a launcher (no WORLD_SIZE in the environment): start the N ranks
    ourselves, as child processes of `torch.distributed.run` on 127.0.0.1, BEFORE this process touches a GPU; pass
    rank 0's JSON line through as our only line of stdout and the children's exit status as ours."""
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    elif p.returncode == 0:
        sys.stderr.write("bench.py: the ranks exited with status 0 but printed no result line\n")
        return 1
    return p.returncode


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        raise SystemExit(launch_ranks(a))
    if os.environ.get("GJX_BENCH_WATCHDOG"):      # diagnostic: dump every thread's stack and exit after N seconds
        import faulthandler

        faulthandler.dump_traceback_later(float(os.environ["GJX_BENCH_WATCHDOG"]), exit=True)
    # stdout carries exactly ONE line, the JSON result: libraries that print to the C-level stdout
    # (RCCL writes a five-line version banner there when a communicator is created) are diverted to
    # stderr for the whole run, and the line is written to the saved descriptor at the end
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world == 1 and a.gpus > 1:
        raise SystemExit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    if a.rendezvous_only:
        import torch.distributed as dist

        dist.init_process_group("gloo")
        t = torch.tensor([rank], dtype=torch.int64)
        dist.all_reduce(t)
        if rank == 0:
            os.write(result_fd, (json.dumps({"metric": "rendezvous only (no measurement)", "value": None,
                                             "n_gpus": world, "rank_sum": int(t.item())}) + "\n").encode())
        dist.barrier()
        dist.destroy_process_group()
        return
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
    from genjax_amd import backend as B
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
                                 n_steps=a.hmc_steps, shard=shard, device=dev)
    smc.force_sharded = a.force_sharded
    smc.overlap = not a.no_overlap
    smc.device_decision = a.device_decision
    sharded = world > 1 or a.force_sharded

    # ---- the dominant kernel is timed with events on the launch stream (torch's current stream); a pool of
    # pairs is created up front and destroyed at the end
    n_pairs = a.steps + a.steps // (T - 1) + 4
    pool = [(B.new_event(), B.new_event()) for _ in range(n_pairs)] if a.moves > 0 else []
    used, timing_on = [], {"on": False}

    def move_events():
        if not timing_on["on"] or len(used) >= len(pool):
            return None
        used.append(pool[len(used)])
        return used[-1]

    smc.move_events = move_events if a.moves > 0 else None

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- stage sequence: whole runs (init + 16 steps), or the cycled steady state
    cursor = {"t": 0, "scan": 0}

    def next_stage():
        """Runs the next stage; returns True when it was an smc_step (False: an init stage)."""
        if a.steady_state:
            if smc.state is None:
                smc.init()
                return False
            smc.step(1 + (cursor["scan"] % (T - 1)), scan_index=cursor["scan"])
            cursor["scan"] += 1
            return True
        t = cursor["t"]
        cursor["t"] = (t + 1) % T
        if t == 0:
            smc.reset()
            smc.init()
            return False
        smc.step(t)
        return True

    if a.steady_state:
        next_stage()                  # particles resident in HBM before anything is timed
    # A full (generation-2) pass of CPython's cyclic GC over the ~10^6 objects torch / numpy leave alive
    # takes 35-40 ms and fires a fixed number of allocations into the run: collect now, then freeze the survivors
    gc.collect()
    gc.freeze()
    for _ in range(a.warmup):
        next_stage()
    if not a.steady_state:
        while cursor["t"] != 0:       # the timed region starts at a run boundary
            next_stage()

    sync_all()
    timing_on["on"] = True
    run_ms, steps_done, stages_done = [], 0, 0
    t0 = time.perf_counter()
    t_run = t0
    while steps_done < a.steps:
        steps_done += int(next_stage())
        stages_done += 1
        if not a.steady_state and (cursor["t"] == 0 or steps_done == a.steps):
            torch.cuda.synchronize()                 # as timing(): every run is synchronised and timed on its own
            now = time.perf_counter()
            if cursor["t"] == 0:
                run_ms.append((now - t_run) * 1e3)
            t_run = now
    sync_all()
    dt = time.perf_counter() - t0
    timing_on["on"] = False
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    n_local = shard.n_local
    # overlapped sharded step: the timed launch moves the own-sourced positions only (the rest follow the exchange)
    timed_positions = float(np.mean(smc.main_positions[-max(1, a.steps):])) if smc.main_positions else float(n_local)
    resampled_flags = list(smc.state.resampled)
    rows_mean = float(np.mean(smc.remote_rows[-max(1, a.steps):])) if smc.remote_rows else 0.0
    rows_max_rank = rows_mean
    if world > 1:
        t_rows = torch.tensor([rows_mean], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(t_rows, op=dist.ReduceOp.MAX)
        rows_max_rank = float(t_rows.item())
    # bytes this rank received from other ranks, per collective, over the whole run (warm-up included), and the
    # largest row exchange of any single resampling; maxima over ranks
    xfer = None
    if sharded and smc.comm is not None:
        mine = [float(smc.comm.bytes_in[k_]) for k_ in ("all_gather", "all_to_all")]
        mine.append(28.0 * max(smc.remote_rows + [0]))
        t_x = torch.tensor(mine, dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(t_x, op=dist.ReduceOp.MAX)
        n_res = max(1, len(smc.remote_rows))
        xfer = {"resamplings": len(smc.remote_rows),
                "all_gather_bytes_per_stage": t_x[0].item() / max(1, smc.comm.normalisers),
                "all_to_all_bytes_per_resampling_mean": t_x[1].item() / n_res,
                "all_to_all_bytes_largest_resampling": t_x[2].item(),
                "bound_28B_x_n_local": 28 * n_local}
    mh_ms = [B.event_elapsed_ms(e0, e1) for e0, e1 in used]
    mh_avg_s = (sum(mh_ms) / max(1, len(mh_ms))) * 1e-3

    # ---- one more (untimed) run with stage events: the other kernels of the step, measured live
    stage_ms = None
    if not sharded and not a.no_stage_profile and smc.buf is not None:
        evs = [B.new_event() for _ in range(B.N_STAGE_EVENTS)]
        B.set_stage_events(smc.buf.ctx, evs)
        acc = {s: [] for s in B.STAGES}
        cursor["t"] = 0
        for _ in range(T if not a.steady_state else T - 1):
            was_step = next_stage()
            torch.cuda.synchronize()
            if not was_step or not smc.state.resampled[-1]:
                continue                              # price the full step: stages that resampled
            for k_, s in enumerate(B.STAGES):
                acc[s].append(B.event_elapsed_ms(evs[k_], evs[k_ + 1]))
        B.set_stage_events(smc.buf.ctx, None)
        for e in evs:
            B.event_destroy(e)
        stage_ms = {s: (float(np.mean(v)) if v else None) for s, v in acc.items()}
    for e0, e1 in pool:
        B.event_destroy(e0)
        B.event_destroy(e1)

    if rank == 0:
        fused_gather = a.kernel == "mh" and smc.fused      # sharded too: received rows sit in the columns' halo
        stored_scores = fused_gather and smc.store_scores
        frac_res = sum(1 for r in resampled_flags if r) / max(1, len(resampled_flags))
        bytes_pp = MH_BYTES + (MH_GATHER_EXTRA * frac_res if fused_gather else 0.0) + (4.0 * 8 if stored_scores else 0.0)
        hbm_gbs = bytes_pp * timed_positions / mh_avg_s / 1e9 if mh_avg_s > 0 else 0.0
        # executed VALU instructions of the launch from the committed SQ pass of this same command
        sq = profile_entry(load_profile("sq"), f"k_loc_{a.kernel}<")
        pmc = profile_entry(load_profile("pmc"), f"k_loc_{a.kernel}<")
        same_cfg = sq is not None and sq.get("moves") == a.moves and a.kernel == "mh"
        valu_insts = sq["SQ_INSTS_VALU"] * (timed_positions / sq["particles"]) if same_cfg else None
        salu_insts = sq["SQ_INSTS_SALU"] * (timed_positions / sq["particles"]) if same_cfg else None
        issue_frac = (valu_insts / mh_avg_s / VALU_PEAK_WAVE_INSTS) if (valu_insts and mh_avg_s > 0) else None
        roofline = {
            "kernel": f"k_loc_{a.kernel}<8 rays> ({a.moves} fused moves per launch"
                      + (", resampling gather fused into its loads" if fused_gather else "")
                      + (", starting scores read from extend's store)" if stored_scores else ")"),
            "bound": "valu",
            "achieved": (issue_frac * VALU_PEAK_TFLOPS) if issue_frac is not None else None,
            "peak": VALU_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": issue_frac,
            "traffic": (pmc["bytes"] * (timed_positions / pmc["particles"])) if (pmc and pmc.get("moves") == a.moves) else None,
            "avg_launch_ms": mh_avg_s * 1e3,
            "launches_timed": len(mh_ms),
            "valu_wave_instructions_per_launch": valu_insts,
            "salu_instructions_per_launch": salu_insts,
            "formula": "frac = SQ_INSTS_VALU per launch (profiles/r03_sq_counters.json, rocprofv3 --pmc on this command, "
                       "scaled by particles per GPU) x 2 issue cycles / (1024 SIMDs x 2.4e9 Hz x avg_launch_ms); "
                       "achieved = frac x 157.3 TFLOP/s (every VALU lane-op priced as an fma).  The instruction mix "
                       "bounds frac below 1: v_alignbit (Threefry rotations), min/max/compare/select take ~4 issue "
                       "cycles, v_rcp/v_log/v_exp/v_sqrt ~8 (profiles/r01_valu_rate_ubench.txt), and the scalar pipe is "
                       "shared by the 4 SIMDs of a CU",
            "traffic_note": "HBM bytes per launch: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, separate passes "
                            "(profiles/r03_pmc_traffic.json); below the algorithmic bytes when resampled ancestors repeat (L2 hits)",
            "hbm": {"algorithmic_bytes_per_particle": bytes_pp, "algorithmic_bytes_per_launch": bytes_pp * timed_positions,
                    "achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_gbs / HBM_PEAK_GBS},
        }
        stages = None
        if stage_ms:
            stages = []
            for s in ("extend", "normalise", "scan", "search", "moves"):
                ms = stage_ms.get(s)
                if ms is None:
                    continue
                ent = {"stage": s, "avg_ms": ms}
                if s in STAGE_BYTES:
                    b = STAGE_BYTES[s] + (4.0 * 8 if (s == "extend" and stored_scores) else 0.0)
                    gbs = b * n_local / (ms * 1e-3) / 1e9
                    ent.update({"bound": "valu" if s == "extend" else ("l2-latency" if s == "search" else "hbm"),
                                "algorithmic_bytes_per_particle": b, "achieved": gbs, "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS})
                stages.append(ent)
            stages.append({"stage": "host_decision", "avg_ms": stage_ms.get("host_decision"),
                           "note": "ESS read-back for the lax.cond (smc.py:747-753) between the two fused calls"})
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
                "workload": "examples/localization rejuvenation_smc, basic world (9 walls), 8 rays, the 17 waypoint "
                            f"observations, N={n_total} particles total, systematic resampling, {a.kernel} K={a.moves}; "
                            + ("steady state: observations 1..16 cycled without re-initialisation" if a.steady_state else
                               "whole runs back to back (init stage + 16 smc_steps each; init stages timed, not counted)"),
                "particles_total": n_total,
                "particles_per_gpu": n_local,
                "parallelism": f"particle-shard x{world}",
                "stages_in_timed_region": stages_done,
                "resampled_stages_of_last_run": int(sum(1 for r in resampled_flags[-T:] if r)),
                "ess_over_n_of_last_run": [round(float(e) / n_total, 5) for e in list(smc.state.ess)[-T:]],
                "steps_with_the_decision_on_the_device": smc.auto_steps if world == 1 and not a.force_sharded else None,
                "collectives": ("none" if world == 1 else
                                f"{a.backend}: per step all_gather(max, 1 f32) + all_gather(tile sums, 2 f32 per 2048 "
                                "particles); when resampling all_gather(tile totals + two raw tiles + one 12-float record per "
                                "owned coarse-table probe), all_gather(row counts), all_to_all_single(one 28-byte row per "
                                "distinct remote ancestor and destination)"),
                "rows_received_per_resampling": rows_mean if sharded else None,
                "rows_received_per_resampling_max_over_ranks": rows_max_rank if sharded else None,
                "bytes_per_row": 28,
                "overlapped_steps": smc.overlapped_steps if sharded else None,
                "front_end_calls_from_c": smc.front_steps if sharded else None,
                "positions_in_timed_launch_mean": timed_positions if sharded else None,
                "bytes_received_max_over_ranks": xfer,
            },
            "timing_protocol": None if not run_ms else {
                "reference": "src/genjax/timing.py:37-137 (every run synchronised and timed on its own; trimmed mean, 20 % "
                             "off either end)",
                "runs": len(run_ms), "run_ms_trimmed_mean": trimmed_mean(run_ms), "run_ms_std": float(np.std(run_ms)),
                "run_ms_min": float(np.min(run_ms)),
                "particle_steps_per_s_trimmed": n_total * (T - 1) / (trimmed_mean(run_ms) * 1e-3),
            },
            "roofline": roofline,
            "roofline_stages": stages,
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
