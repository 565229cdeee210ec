"""Particle-axis sharding of the SMC hot path across the GPUs of one node
(SURVEY.md section 8e): one process per GPU, `torch.distributed` (backend "nccl" = RCCL over
xGMI), contiguous global index ranges per rank (rank r holds [r n, (r + 1) n)).

What crosses GPUs per SMC step of the localization driver (`ShardedLocalizationSMC`, the
gjx_shard_* entry points):
  * normaliser: all_gather of every rank's maximum (1 float), all_gather of the per-tile sums
    {sum exp(a - M), sum exp(a - M)^2} of its 2048-tiles (2 n / 2048 floats) -- every rank then folds
    ALL tile sums in global tile order, which is what one device does: lse and ESS have the same
    bits for 1, 2, 4, 8 ranks;
  * resampling (only in steps that resample): all_gather of the balanced-tree tile totals, one raw tile from
    either end of every shard, and one 12-float record per coarse-table probe the shard owns
    (n / 2048 + 4096 + 12 * ~2^lc / G floats) for the association-exact global CDF and the COMPLETE coarse
    bisection table on every rank (the owner sends the addends that turn the carry of the probe's tile into
    cdf[mid]; everyone redoes those few additions against the scanned totals: no all_reduce), all_gather of the
    rows every rank will send to every other (G int32), and ONE all_to_all_single of rows
    {first position, x, y, th, mx, my, mth} (28 bytes): the rank that OWNS an ancestor finds the
    positions that select it (positions and ancestors are both increasing, so they form a range) and
    sends each distinct selected particle once per destination rank.  A rank receives at most one row
    per distinct remote ancestor of its n positions, never a window of rows nobody selected.
PRNG draws are counter-based on the GLOBAL particle index, so every G gives the same particles bit
for bit as G = 1.

`backend` is injectable so that the collective choreography can be exercised by gloo tests on
CPU with a stand-in (tests/test_parallel_gloo.py); the product path always uses
genjax_amd.backend (HIP kernels).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
import torch
import torch.distributed as dist

SCAN_TILE = 2048
ROW_BYTES = 28
_SIDE_STREAMS = {}     # device index -> the process's side stream there (ShardedLocalizationSMC._streams)


def _concurrent_stream(device, candidates: int = 4):
    """A normal-priority stream that runs beside the current stream: the first of a few fresh streams on which a
    tiny operation finishes while a ~1 ms spin (the library's one idle wave, gjx_spin_us) is still running on the
    current stream; if none does, the last candidate: correct either way, only without overlap.  The candidates
    not chosen are dropped here (torch hands streams out of a per-device pool it never destroys; the driver keeps
    ONE side stream per device and process, whatever the number of drivers)."""
    from . import backend as B

    main = torch.cuda.current_stream(device)
    probe = torch.zeros(1, dtype=torch.float32, device=device)
    main.synchronize()
    chosen = None
    for _ in range(candidates):
        chosen = torch.cuda.Stream(device=device)          # normal priority: never a priority queue
        main_done, side_done = torch.cuda.Event(), torch.cuda.Event()
        with torch.cuda.device(device):
            B.spin_us(1000)                                # on the current (main) stream
        main_done.record(main)
        with torch.cuda.stream(chosen):
            probe.add_(1.0)
            side_done.record(chosen)
        side_done.synchronize()
        beside = not main_done.query()
        main.synchronize()
        if beside:
            break
    return chosen


def buf_is_device(buf) -> bool:
    return getattr(buf, "logw", None) is not None and getattr(buf.logw, "is_cuda", False)


@dataclass
class ShardInfo:
    rank: int = 0
    world: int = 1
    n_global: int = 0

    @property
    def n_local(self):
        return self.n_global // self.world

    @property
    def offset(self):
        return self.rank * self.n_local


class Comm:
    """The collectives of the sharded step.  Backend "nccl" (RCCL over xGMI) takes device tensors
    directly; with "gloo" (CPU rehearsals, several ranks on one GPU in tests) device tensors are staged
    through host memory -- transport only, the kernels are the same.  With one rank and no process group
    every collective is the identity.  bytes_in counts what this rank RECEIVED from other ranks."""

    def __init__(self, group=None, world: int | None = None):
        self.group = group
        self.live = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.live else 1
        if world is not None and world != self.world:
            raise ValueError(f"shard.world = {world} but the process group has {self.world} ranks")
        self.rank = dist.get_rank(group) if self.live else 0
        self.gloo = self.live and dist.get_backend(group) == "gloo"
        self.bytes_in = {"all_gather": 0, "all_to_all": 0}
        self.normalisers = 0         # global normalisers computed (one per stage)
        # RCCL: the process group's own method with an option object built once -- what the torch.distributed
        # wrapper ends up calling, without its per-call argument checks (4 all_gathers per stage)
        self._pg = None
        if self.live and not self.gloo:
            try:
                pg = group or dist.distributed_c10d._get_default_group()
                ag = dist.distributed_c10d.AllgatherOptions()
                ag.asyncOp = False
                if hasattr(pg, "_allgather_base"):
                    self._pg, self._ag_opts = pg, ag
            except (AttributeError, RuntimeError, TypeError):
                self._pg = None
        self.chandle = None          # the C side's communicator (backend.CommHandle): see c_side()

    def c_side(self, device):
        """The C side's own communicator, so that the front end of a sharded step is ONE host call with its
        collectives issued from C (gjx_shard_front).  RCCL ranks: rank 0's unique id travels over this process
        group, every rank creates its RCCL communicator, and one all_gather through it is checked against the same
        all_gather through torch.distributed before it is trusted.  One rank without a process group: the identity
        communicator.  gloo ranks (CPU rehearsals, several ranks on one GPU): None, the step stays call by call."""
        import os
        import sys

        from . import backend as B

        if self.chandle is not None or os.environ.get("GJX_SHARD_FRONT", "1") == "0" or not hasattr(B, "CommHandle"):
            return self.chandle
        if self.gloo:
            return None
        try:
            if not self.live:
                self.chandle = B.CommHandle(0, 1)        # one rank, no process group: the identity communicator
                return self.chandle
            idt = torch.zeros(128, dtype=torch.uint8, device=device)
            if self.rank == 0:
                idt.copy_(torch.frombuffer(bytearray(B.comm_unique_id()), dtype=torch.uint8))
            dist.broadcast(idt, src=0, group=self.group)
            h = B.CommHandle(self.rank, self.world, bytes(idt.cpu().numpy().tobytes()))
            mine = torch.full((4,), float(self.rank), dtype=torch.float32, device=device)
            got = torch.empty(4 * self.world, dtype=torch.float32, device=device)
            want = torch.empty(4 * self.world, dtype=torch.float32, device=device)
            h.all_gather(got, mine)
            dist.all_gather_into_tensor(want, mine, group=self.group)
            torch.cuda.synchronize(device)
            ok = torch.tensor([int(torch.equal(got, want))], dtype=torch.int32, device=device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)      # every rank takes the same path
            if int(ok.item()) == 1:
                self.chandle = h
            else:
                h.close()
                print("genjax_amd: the C-side RCCL communicator disagreed with torch.distributed; the sharded step "
                      "stays call by call", file=sys.stderr)
        except Exception as e:                                   # no RCCL, no peer access...: the Python-driven step
            print(f"genjax_amd: C-side communicator not available ({e}); the sharded step stays call by call",
                  file=sys.stderr)
            self.chandle = None
        return self.chandle

    def _staged(self, t):
        return self.gloo and t.is_cuda

    def all_gather_into(self, out: torch.Tensor, inp: torch.Tensor):
        self.bytes_in["all_gather"] += (self.world - 1) * inp.numel() * inp.element_size()
        if not self.live:
            out.copy_(inp.reshape(-1))
        elif self._pg is not None:
            work = self._pg._allgather_base(out, inp, self._ag_opts)
            if work is not None:
                work.wait()
        elif self._staged(inp):
            parts = [torch.empty(inp.shape, dtype=inp.dtype) for _ in range(self.world)]
            dist.all_gather(parts, inp.cpu(), group=self.group)
            out.copy_(torch.cat(parts))
        elif self.gloo:
            parts = list(out.view(self.world, -1).unbind(0))
            dist.all_gather(parts, inp, group=self.group)
        else:
            dist.all_gather_into_tensor(out, inp, group=self.group)

    def all_to_all(self, out: torch.Tensor, inp: torch.Tensor, out_splits, in_splits):
        self.bytes_in["all_to_all"] += out.numel() * out.element_size()
        if not self.live:
            return
        if self._staged(inp):
            ho = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(ho, inp.cpu(), list(out_splits), list(in_splits), group=self.group)
            out.copy_(ho)
        else:
            dist.all_to_all_single(out, inp, list(out_splits), list(in_splits), group=self.group)

    def warm(self, device):
        """One tiny call of every collective the step uses, so that RCCL's lazy set-up (communicator,
        the peer-to-peer channels all_to_all opens per pair) happens here and not inside a step."""
        G = self.world
        a = torch.zeros(G, dtype=torch.float32, device=device)
        b = torch.empty(G, dtype=torch.float32, device=device)
        self.all_to_all(b, a, [1] * G, [1] * G)
        self.all_gather_into(b, a[:1].contiguous())
        i = torch.zeros(G * G, dtype=torch.int32, device=device)
        self.all_gather_into(i, i[:G].contiguous())
        self.bytes_in = {k_: 0 for k_ in self.bytes_in}


@dataclass
class LocSMCState:
    x: torch.Tensor
    y: torch.Tensor
    th: torch.Tensor
    mx: torch.Tensor
    my: torch.Tensor
    mth: torch.Tensor
    logw: torch.Tensor
    lml: float = 0.0
    t: int = 0
    ess: list = field(default_factory=list)
    resampled: list = field(default_factory=list)


class ShardedLocalizationSMC:
    """rejuvenation_smc(localization_model, mcmc_kernel = mh | hmc, systematic resampling) over a
    sharded particle set; with world = 1 it is the single-GPU hot path step by step.
    Key schedule identical to genjax_amd.inference.rejuvenation_smc."""

    def __init__(self, world_native, observations, n_global, key, kernel="mh", n_moves=10, step_size=0.01,
                 n_steps=10, shard: ShardInfo | None = None, group=None, backend=None, device=None):
        from . import backend as B

        self.B = backend or B
        self.Bkeys = B
        self.w = world_native
        self.obs = np.ascontiguousarray(np.asarray(observations, np.float32))
        self.shard = shard or ShardInfo(0, 1, n_global)
        self.shard.n_global = n_global
        self.group = group
        self.kernel, self.K, self.eps, self.L = kernel, n_moves, step_size, n_steps
        self.key0 = np.asarray(key, np.uint32).reshape(2).copy()
        self.top = B.KeySeq(key)
        self.scan_key = None
        self.device = device
        self.state = None
        self.fused = True            # single GPU: two fused host calls per step
        self.buf = None
        self.move_events = None      # optional callable -> (begin, end) events around the move kernel
        self.comm = None
        self.speculate = True        # sharded step: enqueue the resampling front end before the ESS read-back
        self.force_sharded = False   # run the sharded choreography even with one rank (overhead measurements)
        self.store_scores = True     # mh: extend keeps the per-ray scores for the move kernel of the same step
        self.remote_rows = []        # rows received from other shards per resampling (diagnostic)
        # one-GPU mh step as ONE C call with the ESS decision taken on the device (gjx_loc_smc_step_auto).  None =
        # by size: on up to 2^22 particles, where the host round trip shows (n = 2^21: 0.499 against 0.514 ms per
        # step); off above (N = 2^24: 3.483 / 3.489 against 3.488 / 3.491 ms -- the decision-reading instantiation
        # of the move kernel is 0.6 % slower, which eats the 0.03 ms round trip it saves).  True / False force it.
        self.device_decision = None
        self.overlap = True          # sharded mh step: own-sourced moves start before the host sizes the exchange
        self.overlapped_steps = 0    # steps that took that form (diagnostic)
        self.auto_steps = 0          # one-GPU steps that ran with the decision on the device (diagnostic)
        self._auto_run = 0           # ... of the current run
        self.main_positions = []     # positions the main move launch of such a step covered (diagnostic)
        self._pinned = None
        self._pinned3 = None
        self._read_ev = None         # event behind an asynchronous read-back when no side stream is in use
        self._sched = {}             # (interpreter key, t) -> keys of the stage (a pure function of both)
        self._side = None            # (side stream, read event, join event) of the overlapped step
        self.c_front = True          # sharded mh step: front end as ONE C call, collectives from the C side (RCCL)
        self.front_steps = 0         # steps that took that form (diagnostic)

    # -- key schedule helpers
    def _move_keys(self, sr):
        from .core import sel
        from .families.base import MoveSpec
        from .families.localization import localization_model

        spec = MoveSpec(self.kernel, sel("x") | sel("y") | sel("theta"), self.eps, self.L)
        return np.stack([localization_model.move_keys(spec, self.Bkeys.fold_in(sr, m)) for m in range(self.K)])

    def _stage_fused(self, seq, t, st):
        """Single-GPU form of _stage: the same key schedule and kernels through the two fused
        entry points (gjx_loc_smc_extend, gjx_loc_smc_resample_moves), with all host work that
        does not depend on the ESS done BEFORE the one read-back of the step."""
        Bk, sh = self.B, self.shard
        n = sh.n_local
        if self.buf is None:
            keep = self.store_scores and self.kernel == "mh" and self.K > 0
            self.buf = Bk.LocStepBuffers(n, self.device, n_rays=self.w.n_rays if keep else 0)
        buf = self.buf
        on_device = self.device_decision if self.device_decision is not None else n <= (1 << 22)
        if (on_device and self.kernel == "mh" and self.K > 0 and sh.offset == 0 and sh.world == 1
                and hasattr(Bk, "loc_smc_step_auto") and buf.logw.is_cuda):
            return self._stage_auto(seq, t, st)
        keys3 = [seq.next() for _ in range(3)]
        if t == 0:
            buf.logw.zero_()
        Bk.loc_smc_extend(self.w, buf, keys3, t == 0, self.obs[t], offset=sh.offset)
        # host work overlapped with the kernels above: cond key, resampling uniform, move keys
        kc = seq.next()                                     # cond: one split
        u = self._host_u(self.Bkeys.KeySeq(kc).next())
        mk = None
        if self.kernel in ("mh", "hmc") and self.K > 0:
            mk = Bk.loc_move_keys(seq.next(), self.K, self.kernel)   # scan: one split
        # the lax.cond decision needs the ESS.  When the last stage resampled this one probably will: its scan and
        # search (they need the lse, which is on the device) are enqueued behind the read-back copy, so the host's
        # round trip runs beside them instead of between the normaliser and the scan; dropped if the ESS says no
        ahead = (self.speculate and sh.world == 1 and buf.logw.is_cuda and hasattr(Bk, "loc_smc_presearch")
                 and (st is None or not st.resampled or st.resampled[-1]))
        if ahead:
            pending = self._read_begin(buf.stats)
            Bk.loc_smc_presearch(buf, u)
            host = self._read_end(pending)[0]
        else:
            host = self._read(buf.stats)
        ess = float(host[2])
        take = ess < (sh.n_global // 2)
        lml = st.lml if st else 0.0
        if take:
            lml = float(np.float32(lml) + (np.float32(host[1]) - np.float32(math.log(sh.n_global))))
        Bk.loc_smc_resample_moves(self.w, buf, take, u, self.kernel if mk is not None else "none",
                                  mk if mk is not None else np.zeros((0, 4, 2), np.uint32), t == 0, self.obs[t],
                                  step_size=self.eps, n_steps=self.L, offset=sh.offset,
                                  events=self.move_events() if self.move_events else None, presearched=ahead)
        c = buf.cols
        new = LocSMCState(c[0], c[1], c[2], c[3], c[4], c[5], buf.logw, lml=lml, t=t,
                          ess=st.ess if st else [], resampled=st.resampled if st else [])
        new.ess.append(ess)
        new.resampled.append(take)
        return new

    def _stage_auto(self, seq, t, st):
        """_stage_fused without a host decision: ONE C call enqueues the whole step (the device takes the lax.cond
        itself, gjx_loc_smc_step_auto); the statistics come back on a side stream while the moves run, for the
        log-ML / ESS bookkeeping only.  Same kernels on the same inputs: same bits."""
        Bk, sh, buf = self.B, self.shard, self.buf
        keys3, u, mk = self._stage_keys(seq)
        if t == 0:
            buf.logw.zero_()
            buf.dev_lml.zero_()
            self._auto_run = 0
        self._auto_run += 1
        side, ev_stats = self._streams(buf.stats)[:2]
        Bk.loc_smc_step_auto(self.w, buf, keys3, t == 0, self.obs[t], u, mk, ev_stats=ev_stats.cuda_event,
                             events=self.move_events() if self.move_events else None)      # flips buf.cur
        side.wait_event(ev_stats)
        with torch.cuda.stream(side):
            pending = self._read_begin(buf.stats)
        host = self._read_end(pending)[0]
        ess = float(host[2])
        take = ess < (sh.n_global // 2)
        lml = st.lml if st else 0.0
        if take:
            lml = float(np.float32(lml) + (np.float32(host[1]) - np.float32(math.log(sh.n_global))))
        self.auto_steps += 1
        return self._new_state(st, t, lml, ess, take)

    def _global_stats(self):
        """buf.max1 (this shard's maximum) -> buf.gstats, the five numbers of the GLOBAL weight vector:
        two all_gathers (1 float, then 2 floats per 2048-tile)."""
        Bk, buf, comm = self.B, self.buf, self.comm
        comm.normalisers += 1
        comm.all_gather_into(buf.max_all, buf.max1)
        Bk.shard_tile_sums(buf)
        comm.all_gather_into(buf.tiles_all, buf.tiles)
        return Bk.shard_stats_fold(buf)

    def _stage_sharded(self, seq, t, st):
        """Sharded form of _stage_fused; collectives as listed in the module docstring (two small all_gathers per
        step; all_gather, all_gather, all_to_all_single in the steps that resample).  mh steps whose front end is
        enqueued speculatively take the overlapped form (_resample_moves_overlapped: the device decides, no host
        decision before the moves); the others read the ESS, decide on the host, and run the two-phase form."""
        Bk, sh = self.B, self.shard
        n, G = sh.n_local, sh.world
        if self.buf is None:
            keep = self.store_scores and self.kernel == "mh" and self.K > 0
            self.buf = Bk.new_shard_buffers(n, sh.n_global, G, sh.rank, self.device,
                                            n_rays=self.w.n_rays if keep else 0)
            self.comm = Comm(self.group, G)
            self.comm.warm(self.device)
            if self.c_front and buf_is_device(self.buf):
                self.comm.c_side(self.device)
        buf, comm = self.buf, self.comm
        keys3, u, mk = self._stage_keys(seq)
        if t == 0:
            buf.logw.zero_()
        lml = st.lml if st else 0.0
        if (self.c_front and comm.chandle is not None and self.overlap and self.kernel == "mh" and mk is not None
                and hasattr(Bk, "shard_front")):
            # the whole front end AND the main move launch in ONE call, collectives from the C side; the device takes
            # the lax.cond; the side work of the step forks on the event the call records after the split
            ev_fork = self._streams(buf.gstats)[1]
            Bk.shard_front(self.w, buf, comm.chandle, keys3, t == 0, self.obs[t], u, mk, ev_split=ev_fork.cuda_event,
                           events=self.move_events() if self.move_events else None)      # flips buf.cur
            comm.normalisers += 1
            comm.bytes_in["all_gather"] += (G - 1) * 4 * (1 + buf.tiles.numel() + buf.msg.numel())
            self.front_steps += 1
            take, ess, lml = self._resample_moves_overlapped(t, mk, lml, launched=True)
            return self._new_state(st, t, lml, ess, take)
        Bk.loc_shard_extend(self.w, buf, keys3, t == 0, self.obs[t])
        self._global_stats()

        def enqueue_front_end(rows=True):
            Bk.shard_cdf_begin(buf)
            comm.all_gather_into(buf.msg_all, buf.msg)
            Bk.shard_cdf_finish(buf)                        # the complete coarse table on every rank
            Bk.shard_search(buf, u)
            if rows:
                comm.all_gather_into(buf.counts_all, buf.counts)
                Bk.shard_pack(buf)                          # rows at device-computed offsets: no host count needed

        # When the previous step resampled, the resampling front end (it only needs the lse, which is
        # on the device) is enqueued BEFORE the ESS is read back: one host round trip per step
        # instead of two; if the ESS then says "no resampling" the work is simply dropped.  Every rank
        # takes the same branch (the decision history is identical on all ranks).
        speculate = self.speculate and (st is None or not st.resampled or st.resampled[-1])
        if speculate and self.overlap and self.kernel == "mh" and mk is not None and hasattr(Bk, "shard_split"):
            enqueue_front_end(rows=False)                   # the moves need the search, not the rows
            take, ess, lml = self._resample_moves_overlapped(t, mk, lml)
            return self._new_state(st, t, lml, ess, take)
        if speculate:
            enqueue_front_end()
        # the lax.cond decision needs the ESS; with the front end already enqueued the row counts come back in the
        # same synchronisation
        host, counts = self._read2(buf.gstats, buf.counts_all if speculate else None)
        ess = float(host[2])
        take = ess < (sh.n_global // 2)
        no_keys = np.zeros((0, 4, 2), np.uint32)
        kind = self.kernel if mk is not None else "none"
        if take:
            lml = float(np.float32(lml) + (np.float32(host[1]) - np.float32(math.log(sh.n_global))))
            if not speculate:
                enqueue_front_end()
                counts = self._read(buf.counts_all)
            counts = counts.reshape(G, G)                           # [source, destination]: sizes the exchange
            to_peer, from_peer = counts[sh.rank].tolist(), counts[:, sh.rank].tolist()
            send = Bk.shard_send_rows(buf, sum(to_peer))
            recv = Bk.shard_recv_rows(buf, sum(from_peer))
            if int(counts.sum()) > 0:                               # the same matrix on every rank
                row = Bk.SHARD_ROW
                comm.all_to_all(recv, send, [r * row for r in from_peer], [r * row for r in to_peer])
            self.remote_rows.append(sum(from_peer))
            Bk.shard_resample_moves(self.w, buf, recv, kind, mk if mk is not None else no_keys, t == 0, self.obs[t],
                                    step_size=self.eps, n_steps=self.L,
                                    events=self.move_events() if self.move_events else None)
        else:
            Bk.loc_smc_resample_moves(self.w, buf, False, u, kind, mk if mk is not None else no_keys, t == 0,
                                      self.obs[t], step_size=self.eps, n_steps=self.L, offset=sh.offset,
                                      events=self.move_events() if self.move_events else None)
        return self._new_state(st, t, lml, ess, take)

    def _stage_keys(self, seq):
        """(extend keys, resampling uniform, move keys) of a stage, in the interpreter's order: three sites, the
        cond's split (whose first split seeds the systematic offset), the scan's split.  A pure function of the
        interpreter key on entry: computed once, then looked up (runs repeated on one key, as timing() does)."""
        entry = seq.key.tobytes()
        hit = self._sched.get(entry)
        if hit is None:
            keys3 = [seq.next() for _ in range(3)]
            kc = seq.next()                                     # cond: one split
            u = self._host_u(self.Bkeys.KeySeq(kc).next())
            mk = None
            if self.kernel in ("mh", "hmc") and self.K > 0:
                mk = self.Bkeys.loc_move_keys(seq.next(), self.K, self.kernel)   # scan: one split
            if len(self._sched) < 4096:
                self._sched[entry] = (keys3, u, mk, seq.key.copy())
            return keys3, u, mk
        seq.key = hit[3].copy()
        return hit[0], hit[1], hit[2]

    def _new_state(self, st, t, lml, ess, take):
        c = self.buf.cols
        new = LocSMCState(c[0], c[1], c[2], c[3], c[4], c[5], self.buf.logw, lml=lml, t=t,
                          ess=st.ess if st else [], resampled=st.resampled if st else [])
        new.ess.append(ess)
        new.resampled.append(take)
        return new

    def _resample_moves_overlapped(self, t, mk, lml, launched=False):
        """The cond(resample) + K mh moves of a step whose CDF and search are already enqueued, with no host decision
        and no collective between the search and the moves: the device takes the ESS decision itself
        (gjx_shard_split) and moves every position whose ancestor is an own particle.  Beside that launch, on a
        side stream, the row counts are gathered, the rows packed, and the ESS, the counts and the own-sourced run
        [p0, p1) come back in ONE read-back; the host then exchanges the rows and moves the positions [0, p0) and
        [p1, n), still on the side stream.  Same bits as the two-phase form (every particle goes through the same
        kernel code with the same counters); the next step waits for both streams."""
        Bk, sh, buf, comm = self.B, self.shard, self.buf, self.comm
        n, G = sh.n_local, sh.world
        one = getattr(buf, "report", None)                  # statistics, split and counts in one buffer: one copy
        if launched:
            # gjx_shard_front has enqueued everything up to and including the main move launch and recorded the fork
            # event after the split; the particles as extended sit in the set the launch read (now the other one)
            side = self._streams(buf.gstats)[0]
            side.wait_event(self._streams(buf.gstats)[1])
            with torch.cuda.stream(side):
                Bk.shard_count(buf)                         # the front end's search left the row counts to this stream
                comm.all_gather_into(buf.counts_all, buf.counts)
                Bk.shard_pack(buf, cols_set=1 - buf.cur)
                pending = self._read_begin(*((one,) if one is not None else (buf.gstats, buf.counts_all, buf.split)))
        else:
            Bk.shard_split(buf, self.K)
            with self._on_side(buf.gstats, fork=True):
                comm.all_gather_into(buf.counts_all, buf.counts)
                Bk.shard_pack(buf)                          # reads the current columns: before the main launch flips
                pending = self._read_begin(*((one,) if one is not None else (buf.gstats, buf.counts_all, buf.split)))
            Bk.shard_moves_main(self.w, buf, mk, t == 0, self.obs[t],
                                events=self.move_events() if self.move_events else None)      # flips buf.cur
        got = self._read_end(pending)
        host, counts, split = buf.parse_report(got[0]) if one is not None else got
        self.overlapped_steps += 1
        ess = float(host[2])
        take = ess < (sh.n_global // 2)
        if bool(split[2]) != take:
            raise RuntimeError(f"sharded step: device and host disagree on the resampling decision (ess={ess})")
        if take:
            lml = float(np.float32(lml) + (np.float32(host[1]) - np.float32(math.log(sh.n_global))))
            p0, p1 = (int(split[0]) - 1, int(split[1])) if split[0] > 0 else (n, n)
            self.main_positions.append(p1 - p0)
            counts = counts.reshape(G, G)                       # [source, destination]: sizes the exchange
            to_peer, from_peer = counts[sh.rank].tolist(), counts[:, sh.rank].tolist()
            send = Bk.shard_send_rows(buf, sum(to_peer))
            recv = Bk.shard_recv_rows(buf, sum(from_peer))
            self.remote_rows.append(sum(from_peer))
            exchange = int(counts.sum()) > 0                    # the same matrix on every rank
            if exchange or p0 + (n - p1) > 0:
                with self._on_side(buf.gstats, fork=False):
                    if exchange:
                        row = Bk.SHARD_ROW
                        comm.all_to_all(recv, send, [r * row for r in from_peer], [r * row for r in to_peer])
                    Bk.shard_moves_fixup(self.w, buf, recv, p0, p1, mk, t == 0, self.obs[t])
        else:
            self.main_positions.append(n)
        self._join_side(buf.gstats)
        return take, ess, lml

    def _read_begin(self, *tensors):
        """Asynchronous read-back of a few small tensors on the current stream: copies into pinned staging and an
        event behind them (host tensors: nothing to start)."""
        if not tensors[0].is_cuda:
            return tensors, None
        if self._pinned3 is None or [p.numel() for p in self._pinned3] != [a.numel() for a in tensors]:
            self._pinned3 = [torch.empty(a.numel(), dtype=a.dtype).pin_memory() for a in tensors]
        for p, a in zip(self._pinned3, tensors):
            p.copy_(a, non_blocking=True)
        if self._side is not None:
            ev = self._side[2]
        else:                                               # no side stream in use (one-GPU two-call step): an event of its own
            if self._read_ev is None:
                self._read_ev = torch.cuda.Event()
            ev = self._read_ev
        ev.record(torch.cuda.current_stream())
        return tensors, ev

    def _read_end(self, pending):
        tensors, ev = pending
        if ev is None:
            return [a.numpy().copy() for a in tensors]
        ev.synchronize()
        return [p.numpy().copy() for p in self._pinned3]

    def _streams(self, like):
        """(side stream, fork event, read event, join event) of the overlapped step."""
        if self._side is None:
            # The side stream must sit on ANOTHER hardware queue than the main stream (HIP spreads streams over a
            # few queues round-robin; on a shared queue the side work simply runs in submission order with the move
            # launch and nothing overlaps): normal priority, picked by a probe that shows it runs beside the main
            # stream.  Never a priority stream: DESIGN.md section 6 records the hangs those produced.
            key_ = like.device.index
            if key_ not in _SIDE_STREAMS:          # ONE side stream per device for the whole process
                _SIDE_STREAMS[key_] = _concurrent_stream(like.device)
            self._side = (_SIDE_STREAMS[key_], torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event())
            for ev in self._side[1:]:
                ev.record()                                 # creates the raw handle (torch creates events lazily)
        return self._side

    def _on_side(self, like, fork):
        """Context: work enqueued inside runs on the side stream; fork=True: after everything enqueued on the main
        stream so far (host tensors: a no-op)."""
        import contextlib

        if not like.is_cuda:
            return contextlib.nullcontext()
        side, ev_fork = self._streams(like)[:2]
        if fork:
            ev_fork.record(torch.cuda.current_stream())
            side.wait_event(ev_fork)
        return torch.cuda.stream(side)

    def _join_side(self, like):
        """The main stream waits for everything enqueued on the side stream."""
        if like.is_cuda:
            side, _f, _r, ev_join = self._streams(like)
            ev_join.record(side)
            torch.cuda.current_stream().wait_event(ev_join)

    def _stage(self, seq, t, st):
        """extend/init -> normaliser -> cond(resample) -> K moves, all under interpreter `seq`."""
        single = self.shard.world == 1 and not self.force_sharded
        if self.fused:
            return self._stage_fused(seq, t, st) if single else self._stage_sharded(seq, t, st)
        if not single:
            raise ValueError("the kernel-by-kernel form of the step exists for one rank only (tests); sharded runs "
                             "use the fused entry points")
        Bk, sh = self.B, self.shard
        n = sh.n_local
        keys3 = [seq.next() for _ in range(3)]
        prev = None if t == 0 else (st.x, st.y, st.th)
        logw_in = st.logw if st is not None else torch.zeros(n, dtype=torch.float32, device=self.device)
        out = Bk.loc_extend(self.w, keys3, t == 0, self.obs[t], prev, logw_in, offset=sh.offset)
        new = LocSMCState(out["x"], out["y"], out["th"], out["mx"], out["my"], out["mth"], out["logw"],
                          lml=st.lml if st else 0.0, t=t, ess=st.ess if st else [],
                          resampled=st.resampled if st else [])
        kc = seq.next()                                     # cond: one split
        stats = Bk.logsumexp_ess(new.logw)
        host = stats.cpu().numpy()                          # the lax.cond decision needs the ESS
        ess = float(host[2])
        new.ess.append(ess)
        take = ess < (sh.n_global // 2)
        new.resampled.append(take)
        if take:
            u = self._host_u(self.Bkeys.KeySeq(kc).next())
            new.lml = float(np.float32(new.lml) + (np.float32(host[1]) - np.float32(math.log(sh.n_global))))
            anc = Bk.resample_systematic(Bk.normalized_weights(new.logw, stats[1:2]), u)
            new.x, new.y, new.th, new.mx, new.my, new.mth = Bk.gather_columns(
                anc, [new.x, new.y, new.th, new.mx, new.my, new.mth])
            new.logw = torch.zeros_like(new.logw)
        if self.kernel in ("mh", "hmc") and self.K > 0:
            sr = seq.next()                                 # scan: one split
            mk = self._move_keys(sr)
            pose, means = (new.x, new.y, new.th), (new.mx, new.my, new.mth)
            if self.kernel == "mh":
                pose = Bk.loc_mh(self.w, mk, t == 0, self.obs[t], pose, means, pose_out=pose, offset=sh.offset)
            else:
                pose = Bk.loc_hmc(self.w, mk, self.eps, self.L, t == 0, self.obs[t], pose, means, pose_out=pose,
                                  offset=sh.offset)
            new.x, new.y, new.th = pose
        return new

    @staticmethod
    def _read(t: torch.Tensor) -> np.ndarray:
        """Device -> host read-back of a few scalars (the step's one or two host decisions)."""
        return t.cpu().numpy().copy()

    def _read2(self, a: torch.Tensor, b: torch.Tensor | None):
        """Two small read-backs under ONE synchronisation (pinned staging, asynchronous copies)."""
        if b is None or not a.is_cuda:
            return self._read(a), (None if b is None else self._read(b))
        if self._pinned is None or self._pinned[1].numel() != b.numel():
            self._pinned = (torch.empty(a.numel(), dtype=a.dtype).pin_memory(),
                            torch.empty(b.numel(), dtype=b.dtype).pin_memory())
        pa, pb = self._pinned
        pa.copy_(a, non_blocking=True)
        pb.copy_(b, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return pa.numpy().copy(), pb.numpy().copy()

    @staticmethod
    def _host_u(key):
        from .inference.smc import _host_uniform01

        return _host_uniform01(key)

    def reset(self):
        """Back to the state before init(): the next init() repeats the run on the same key (the device
        buffers are kept)."""
        self.top = self.Bkeys.KeySeq(self.key0)
        self.scan_key, self.state = None, None

    def init(self):
        self.state = self._stage(self.top, 0, None)
        self.scan_key = self.top.next()                     # scan over observations[1:]
        return self.state

    def step(self, t: int, scan_index: int | None = None):
        """One smc_step (smc.py:731-767) with observation t; scan_index = t - 1 by default."""
        i = (t - 1) if scan_index is None else scan_index
        ck = (self.scan_key.tobytes(), i)
        start = self._sched.get(ck)
        if start is None:
            start = self.Bkeys.fold_in(self.scan_key, i)
            if len(self._sched) < 4096:
                self._sched[ck] = start
        seq = self.Bkeys.KeySeq(start)
        self.state = self._stage(seq, t, self.state)
        return self.state

    def run(self):
        from .inference.smc import _no_gc_pauses

        with _no_gc_pauses():
            self.init()
            for t in range(1, self.obs.shape[0]):
                self.step(t)
        return self.state

    def check(self):
        """Raises if a resampling found a position without a source (the kernels flag it, they never clamp)."""
        if self.comm is None and self.state is not None and self._auto_run == len(self.state.ess) > 0:
            # every stage of this run took the lax.cond on the device: its flag and its own log-ML accumulator
            dec = self.buf.decision.cpu().numpy()
            dev_lml, host_lml = np.float32(self.buf.dev_lml.cpu().item()), np.float32(self.state.lml)
            if int(dec[3]) != 0 or not (dev_lml == host_lml or (np.isnan(dev_lml) and np.isnan(host_lml))):
                raise RuntimeError(f"one-GPU step: device and host bookkeeping disagree (flag {int(dec[3])}, log-ML "
                                   f"{dev_lml} on the device, {self.state.lml} on the host)")
        if self.comm is not None and (int(self.buf.err.cpu().item()) != 0 or
                                      (hasattr(self.buf, "split") and int(self.buf.split[3].cpu().item()) != 0)):
            raise RuntimeError("sharded resampling: a position found no source particle, or disagreed with the "
                               "own-sourced run the host was told (inconsistent search / exchange)")

    def log_marginal_likelihood(self) -> float:
        self.check()
        if self.comm is not None:                           # sharded path: the normaliser of the step
            self.B.shard_local_max(self.buf)
            stats = self._global_stats().cpu().numpy()
            if getattr(self.buf.max1, "is_cuda", False):
                self.buf.max1.fill_(-math.inf)              # the cell the one-call front end raises by atomics
        else:
            stats = self.B.logsumexp_ess(self.state.logw).cpu().numpy()
        return float(np.float32(self.state.lml) + (np.float32(stats[1]) - np.float32(math.log(self.shard.n_global))))
