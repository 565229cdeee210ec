"""Particle-axis sharding of the SMC hot path across the GPUs of one node
(SURVEY.md section 8e): one process per GPU, `torch.distributed` (backend "nccl" = RCCL over
xGMI), contiguous global index ranges per rank.

What crosses GPUs per SMC step of the localization driver (`ShardedLocalizationSMC`, the
gjx_shard_* entry points):
  * normaliser: ONE all_gather of every rank's {max, lse, ess, sum exp, sum exp^2} (5 floats),
    folded rank-ascending in f64 so that the result does not depend on the collective's ring order;
  * resampling: all_gather of the per-shard balanced-tree tile totals (N / 2048 floats) for the
    association-exact global CDF, all_reduce of the coarse bisection table (<= 2^14 floats), and
    one all_to_all_single of the rows {cdf, 6 columns} of a rank's window that lie in OTHER shards
    (the own shard is read in place; the remaining bisection levels run on window + shard).
The generic functions `global_normaliser` / `global_systematic_resample` (any list of columns,
used by genjax_amd.inference for other models) do the same with the granular entry points: two
small all_gathers for the normaliser and a grouped send/recv of contiguous slabs.
PRNG draws are counter-based on the GLOBAL particle index, so every G gives the same
particles bit for bit as G = 1 given the same lse.

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


def _all_gather_cat(t: torch.Tensor, world: int, group=None) -> torch.Tensor:
    if world == 1:
        return t
    if t.is_cuda and dist.get_backend(group) == "gloo":      # rehearsal transport: stage through the host
        out = [torch.empty(t.shape, dtype=t.dtype) for _ in range(world)]
        dist.all_gather(out, t.cpu(), group=group)
        return torch.cat(out).to(t.device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t.contiguous(), group=group)
    return torch.cat(out)


def global_normaliser(Bk, logw: torch.Tensor, shard: ShardInfo, group=None) -> torch.Tensor:
    """{max, lse, ess, sum exp, sum exp^2} of the GLOBAL weight vector (device tensor)."""
    if shard.world == 1:
        return Bk.logsumexp_ess(logw)
    m_loc = Bk.max_f32(logw)
    m_all = _all_gather_cat(m_loc, shard.world, group)
    m = m_all.max().reshape(1)
    s_loc = Bk.sumexp_f32(logw, m)
    s_all = _all_gather_cat(s_loc, shard.world, group).reshape(shard.world, 2)
    # rank-ascending fold in f64, rounded once (order-independent of the ring)
    s = s_all.double().sum(dim=0).float()
    mm = torch.where(torch.isfinite(m), m, torch.zeros_like(m))
    lse = torch.log(s[0:1]) + mm
    ess = (s[0:1] * s[0:1]) / s[1:2]
    return torch.cat([m, lse, ess, s[0:1], s[1:2]])


def slab_plan(lo: int, hi: int, n_local: int, world: int):
    """Which [start, stop) of every source rank's local columns covers global indices [lo, hi]."""
    plan = []
    for s in range(world):
        a, b = max(lo, s * n_local), min(hi + 1, (s + 1) * n_local)
        plan.append((a - s * n_local, b - s * n_local) if b > a else (0, 0))
    return plan


def exchange_slabs(cols: torch.Tensor, wants: np.ndarray, shard: ShardInfo, group=None) -> torch.Tensor:
    """cols: (C, n_local) local columns.  wants: (G, 2) host array, row r = the inclusive global
    index range [lo, hi] rank r needs (every rank holds the same matrix).  Returns the global
    columns [lo, hi] of THIS rank, (C, hi - lo + 1), assembled from the owning ranks by one
    grouped send/recv (all-to-all-v of contiguous slabs)."""
    G, n = shard.world, shard.n_local
    lo, hi = int(wants[shard.rank, 0]), int(wants[shard.rank, 1])
    if G == 1:
        return cols[:, lo:hi + 1]
    my_plan = slab_plan(lo, hi, n, G)
    recv = [torch.empty((cols.shape[0], b - a), dtype=cols.dtype, device=cols.device) for a, b in my_plan]
    ops, keep = [], []
    for peer in range(G):
        a, b = slab_plan(int(wants[peer, 0]), int(wants[peer, 1]), n, G)[shard.rank]
        if peer == shard.rank:
            if b > a:
                recv[peer].copy_(cols[:, a:b])
            continue
        if b > a:                                   # what `peer` needs from me
            buf = cols[:, a:b].contiguous()
            keep.append(buf)
            ops.append(dist.P2POp(dist.isend, buf, peer, group=group))
        if recv[peer].numel() > 0:                  # what I need from `peer`
            ops.append(dist.P2POp(dist.irecv, recv[peer], peer, group=group))
    if ops:                                         # grouped send/recv = all-to-all-v
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return torch.cat(recv, dim=1)


class Comm:
    """The four collectives of the sharded step.  Backend "nccl" (RCCL over xGMI) takes device
    tensors directly; with "gloo" (CPU rehearsals, several ranks on one GPU in tests) device
    tensors are staged through host memory -- transport only, the kernels are the same."""

    def __init__(self, group=None):
        self.group = group
        self.gloo = dist.get_backend(group) == "gloo"

    def _staged(self, t):
        return self.gloo and t.is_cuda

    def all_gather_into(self, out: torch.Tensor, inp: torch.Tensor):
        if self._staged(inp):
            parts = [torch.empty(inp.shape, dtype=inp.dtype) for _ in range(dist.get_world_size(self.group))]
            dist.all_gather(parts, inp.cpu(), group=self.group)
            out.copy_(torch.cat(parts))
        elif self.gloo:
            parts = list(out.view(dist.get_world_size(self.group), -1).unbind(0))
            dist.all_gather(parts, inp, group=self.group)
        else:
            dist.all_gather_into_tensor(out, inp, group=self.group)

    def all_reduce_sum(self, t: torch.Tensor):
        if self._staged(t):
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def all_to_all(self, out: torch.Tensor, inp: torch.Tensor, out_splits, in_splits):
        if self._staged(inp):
            ho = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(ho, inp.cpu(), list(out_splits), list(in_splits), group=self.group)
            out.copy_(ho)
        else:
            dist.all_to_all_single(out, inp, list(out_splits), list(in_splits), group=self.group)


    def warm(self, device):
        """One tiny call of every collective the step uses, so that RCCL's lazy set-up (communicator,
        the peer-to-peer channels all_to_all opens per pair) happens here and not inside a step."""
        G = dist.get_world_size(self.group)
        a = torch.zeros(G, dtype=torch.float32, device=device)
        b = torch.empty(G, dtype=torch.float32, device=device)
        self.all_to_all(b, a, [1] * G, [1] * G)
        self.all_gather_into(b, a[:1].contiguous())
        self.all_reduce_sum(a)


def exchange_plan(wants: np.ndarray, rank: int, n_local: int, world: int):
    """Host plan of the row exchange from the windows of ALL ranks (wants: world x {lo, hi}
    inclusive, identical on every rank).  A rank reads the part of its window inside its own
    shard in place; the rest arrives as rows from the owners.
    -> (pieces, in_rows, out_rows, left_rows, any_remote): local ranges to pack for each peer
    (ascending peer, own entry empty), rows sent to / received from each peer, how many received
    rows lie below the own shard, and whether ANY rank needs remote rows (else every rank skips
    the collective)."""
    # plain ints: at 8 ranks this is ~10 us of host time on the step's critical path
    w = wants.tolist()
    my0, my1 = rank * n_local, (rank + 1) * n_local
    lo_r, hi1_r = w[rank][0], w[rank][1] + 1
    pieces, in_rows, out_rows = [], [], []
    any_remote = False
    for peer in range(world):
        lo, hi1 = w[peer][0], w[peer][1] + 1
        s0 = peer * n_local
        any_remote = any_remote or lo < s0 or hi1 > s0 + n_local
        if peer == rank:
            pieces.append((0, 0))
            in_rows.append(0)
            out_rows.append(0)
            continue
        a, b = max(lo, my0), min(hi1, my1)                  # what `peer` takes from my shard
        pieces.append((a - my0, b - my0) if b > a else (0, 0))
        in_rows.append(max(b - a, 0))
        out_rows.append(max(min(hi1_r, s0 + n_local) - max(lo_r, s0), 0))   # what I take from `peer`
    left_rows = sum(out_rows[:rank])
    return pieces, in_rows, out_rows, left_rows, any_remote


FINE_LEVELS = 11   # bisection levels left to the slab: intervals of <= 2^11 CDF entries

_plan_dev_cache: dict = {}


def _owned_plan(Bk, n_global: int, lc: int, shard: ShardInfo, device):
    """Device index tensors of the coarse-table nodes whose probe index this rank owns."""
    key_ = (n_global, lc, shard.rank, shard.world, str(device))
    if key_ not in _plan_dev_cache:
        plan = np.asarray(Bk.bisect_plan(n_global, lc)).astype(np.int64)
        nodes = np.arange(plan.size)
        own = (plan >= shard.offset) & (plan < shard.offset + shard.n_local) & (nodes > 0)
        _plan_dev_cache[key_] = (torch.from_numpy(nodes[own]).to(device),
                                 torch.from_numpy((plan[own] - shard.offset).astype(np.int32)).to(device))
    return _plan_dev_cache[key_]


def global_systematic_resample(Bk, logw, lse, u: float, cols, shard: ShardInfo, group=None):
    """Systematic resampling of the global particle set.  cols: list of local f32 columns.
    Returns the list of resampled local columns (outputs [offset, offset + n_local)).

    Sharded form (bit-identical to one device for any world size):
      1. local tile totals -> all_gather (N/2048 floats) -> every rank scans them with the same
         association -> local in-tile fold = this rank's slice of the GLOBAL CDF;
      2. coarse table: the CDF at the probe indices of the first `lc` bisection levels
         (<= 2^14 values), each filled by its owner and summed over ranks (x + 0 is exact);
      3. the [low, high) ranges the positions of EVERY rank reach after the coarse levels, computed
         redundantly on each rank from the shared table (one 16 G-byte read-back, no collective),
         then ONE grouped exchange of contiguous slabs of {CDF, columns} over those ranges:
         systematic positions are increasing, so each range is contiguous and, for balanced
         weights, mostly local;
      4. the remaining bisection levels and the ancestor gather run on the slab."""
    G, n = shard.world, shard.n_local
    w = Bk.normalized_weights(logw, lse)
    if G == 1:
        anc = Bk.resample_systematic(w, u)
        return Bk.gather_columns(anc, cols)
    if n % SCAN_TILE != 0:
        raise ValueError(f"sharded resampling needs n_local ({n}) to be a multiple of {SCAN_TILE}")
    N = shard.n_global
    totals = Bk.cumsum_tile_totals(w)
    all_totals = _all_gather_cat(totals, G, group)
    scanned = Bk.cumsum(all_totals)
    cdf_loc = Bk.cumsum_apply(w, scanned, shard.rank * (n // SCAN_TILE))
    lc = max(0, Bk.bisect_levels(N) - FINE_LEVELS)
    nodes, local_idx = _owned_plan(Bk, N, lc, shard, cdf_loc.device)
    table = torch.zeros(1 << lc, dtype=torch.float32, device=cdf_loc.device)
    if nodes.numel() > 0:
        table.index_copy_(0, nodes, Bk.gather_columns(local_idx, [cdf_loc])[0])
    dist.all_reduce(table, op=dist.ReduceOp.SUM, group=group)
    wants = Bk.searchsorted_window(table, lc, N, u, 0, n, N, n_ranges=G).cpu().numpy().reshape(G, 2).copy()
    wants[:, 1] = np.minimum(wants[:, 1], N - 1)
    lo = int(wants[shard.rank, 0])
    slab = exchange_slabs(torch.stack([cdf_loc] + list(cols)), wants, shard, group)
    anc = Bk.searchsorted_systematic_2l(table, lc, slab[0], lo, N, u, shard.offset, n, N)
    return Bk.gather_columns(anc, [slab[c] for c in range(1, slab.shape[0])], idx_base=lo)


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
        self.remote_rows = []        # rows fetched from other shards per resampling (diagnostic)
        self.fast_resamples = 0      # sharded resamplings that took the halo path (diagnostic)
        self.halo_path = True        # False: always the general resampling tail (tests)

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
        host = self._read(buf.stats)                        # the lax.cond decision needs the ESS
        ess = float(host[2])
        take = ess < (sh.n_global // 2)
        lml = st.lml if st else 0.0
        if take:
            lml = float(np.float32(lml) + (np.float32(host[1]) - np.float32(math.log(sh.n_global))))
        Bk.loc_smc_resample_moves(self.w, buf, take, u, self.kernel if mk is not None else "none",
                                  mk if mk is not None else np.zeros((0, 4, 2), np.uint32), t == 0, self.obs[t],
                                  step_size=self.eps, n_steps=self.L, offset=sh.offset,
                                  events=self.move_events() if self.move_events else None)
        c = buf.cols
        new = LocSMCState(c[0], c[1], c[2], c[3], c[4], c[5], buf.logw, lml=lml, t=t,
                          ess=st.ess if st else [], resampled=st.resampled if st else [])
        new.ess.append(ess)
        new.resampled.append(take)
        return new

    def _stage_sharded(self, seq, t, st):
        """Sharded form of _stage_fused: the same two host decisions, four collectives
        (all_gather of 5 floats, all_gather of the tile totals, all_reduce of the coarse table,
        all_to_all_single of the rows a rank's window needs from other shards)."""
        Bk, sh = self.B, self.shard
        n, G = sh.n_local, sh.world
        if self.buf is None:
            keep = self.store_scores and self.kernel == "mh" and self.K > 0
            self.buf = Bk.new_shard_buffers(n, sh.n_global, G, sh.rank, self.device,
                                            n_rays=self.w.n_rays if keep else 0)
            self.comm = Comm(self.group)
            self.comm.warm(self.device)
        buf, comm = self.buf, self.comm
        keys3 = [seq.next() for _ in range(3)]
        if t == 0:
            buf.logw.zero_()
        Bk.loc_smc_extend(self.w, buf, keys3, t == 0, self.obs[t], offset=sh.offset)
        comm.all_gather_into(buf.stats_all, buf.stats)
        Bk.shard_stats_fold(buf)
        kc = seq.next()                                     # cond: one split
        u = self._host_u(self.Bkeys.KeySeq(kc).next())
        mk = None
        if self.kernel in ("mh", "hmc") and self.K > 0:
            mk = self.Bkeys.loc_move_keys(seq.next(), self.K, self.kernel)   # scan: one split
        def enqueue_cdf_and_windows():
            Bk.shard_cdf_begin(buf)
            comm.all_gather_into(buf.all_totals, buf.totals)
            Bk.shard_cdf_finish(buf)
            comm.all_reduce_sum(buf.table)
            return Bk.shard_windows(buf, u)

        # When the previous step resampled, the resampling front end (it only needs the lse, which is
        # on the device) is enqueued BEFORE the ESS is read back: one host round trip per step
        # instead of two; if the ESS then says "no resampling" the work is simply dropped.  Every rank
        # takes the same branch (the decision history is identical on all ranks).
        speculate = self.speculate and (st is None or not st.resampled or st.resampled[-1])
        wants_dev = enqueue_cdf_and_windows() if speculate else None
        host = self._read(buf.gstats)                       # the lax.cond decision needs the ESS
        ess = float(host[2])
        take = ess < (sh.n_global // 2)
        lml = st.lml if st else 0.0
        if take:
            lml = float(np.float32(lml) + (np.float32(host[1]) - np.float32(math.log(sh.n_global))))
            if wants_dev is None:
                wants_dev = enqueue_cdf_and_windows()
            wants = self._read(wants_dev).reshape(G, 2)     # sizes the exchange
            wants[:, 1] = np.minimum(wants[:, 1], sh.n_global - 1)
            pieces, in_rows, out_rows, left_rows, any_remote = exchange_plan(wants, sh.rank, n, G)
            # A rank whose window does not fit its halos takes the general path, which has no use for
            # the stored scores: its rows travel without them (28 B instead of 28 + 4 R).  Those are the
            # large windows, the ones that bound the exchange.  Every rank evaluates the same predicate
            # for every rank from the same numbers.
            extra = {}
            if getattr(buf, "n_scores", 0) > 0 and hasattr(Bk, "window_fits_halo"):
                halo_ok = self.halo_path and mk is not None and self.kernel == "mh"
                widths = [buf.row if halo_ok and Bk.window_fits_halo(int(wants[p, 0]), int(wants[p, 1]), p * n, n,
                                                                     buf.halo) else Bk.SHARD_ROW for p in range(G)]
                if any(wd != buf.row for wd in widths):
                    extra = {"widths": widths}
            my_row = extra["widths"][sh.rank] if extra else buf.row
            recv = Bk.shard_recv_rows(buf, sum(out_rows), my_row) if extra else Bk.shard_recv_rows(buf, sum(out_rows))
            if any_remote:
                send = Bk.shard_pack(buf, pieces, **extra)
                in_w = extra["widths"] if extra else [buf.row] * G
                comm.all_to_all(recv, send, [r * my_row for r in out_rows], [r * wd for r, wd in zip(in_rows, in_w)])
            self.remote_rows.append(sum(out_rows))
            no_keys = np.zeros((0, 4, 2), np.uint32)
            fast = Bk.shard_resample_moves(self.w, buf, u, int(wants[sh.rank, 0]), int(wants[sh.rank, 1]), recv,
                                           left_rows, self.kernel if mk is not None else "none",
                                           mk if mk is not None else no_keys, t == 0, self.obs[t],
                                           step_size=self.eps, n_steps=self.L,
                                           events=self.move_events() if self.move_events else None,
                                           allow_fast=self.halo_path, **({"recv_row": my_row} if extra else {}))
            self.fast_resamples += int(bool(fast))
        else:
            Bk.loc_smc_resample_moves(self.w, buf, False, u, self.kernel if mk is not None else "none",
                                      mk if mk is not None else np.zeros((0, 4, 2), np.uint32), t == 0, self.obs[t],
                                      step_size=self.eps, n_steps=self.L, offset=sh.offset,
                                      events=self.move_events() if self.move_events else None)
        c = buf.cols
        new = LocSMCState(c[0], c[1], c[2], c[3], c[4], c[5], buf.logw, lml=lml, t=t,
                          ess=st.ess if st else [], resampled=st.resampled if st else [])
        new.ess.append(ess)
        new.resampled.append(take)
        return new

    def _stage(self, seq, t, st):
        """extend/init -> normaliser -> cond(resample) -> K moves, all under interpreter `seq`."""
        if self.fused and hasattr(self.B, "loc_smc_extend"):
            single = self.shard.world == 1 and not self.force_sharded
            return self._stage_fused(seq, t, st) if single else self._stage_sharded(seq, t, st)
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
        stats = global_normaliser(Bk, new.logw, sh, self.group)
        host = stats.cpu().numpy()                          # the lax.cond decision needs the ESS
        ess = float(host[2])
        new.ess.append(ess)
        take = ess < (sh.n_global // 2)
        new.resampled.append(take)
        if take:
            u = self._host_u(self.Bkeys.KeySeq(kc).next())
            new.lml = float(np.float32(new.lml) + (np.float32(host[1]) - np.float32(math.log(sh.n_global))))
            cols = global_systematic_resample(Bk, new.logw, stats[1:2], u,
                                              [new.x, new.y, new.th, new.mx, new.my, new.mth], sh, self.group)
            new.x, new.y, new.th, new.mx, new.my, new.mth = cols
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
        seq = self.Bkeys.KeySeq(self.Bkeys.fold_in(self.scan_key, i))
        self.state = self._stage(seq, t, self.state)
        return self.state

    def run(self):
        from .inference.smc import _no_gc_pauses

        with _no_gc_pauses():
            self.init()
            for t in range(1, self.obs.shape[0]):
                self.step(t)
        return self.state

    def log_marginal_likelihood(self) -> float:
        if self.comm is not None:                           # sharded fused path: same fold as the step
            self.buf.stats.copy_(self.B.logsumexp_ess(self.state.logw))
            self.comm.all_gather_into(self.buf.stats_all, self.buf.stats)
            stats = self.B.shard_stats_fold(self.buf).cpu().numpy()
        else:
            stats = global_normaliser(self.B, self.state.logw, self.shard, self.group).cpu().numpy()
        return float(np.float32(self.state.lml) + (np.float32(stats[1]) - np.float32(math.log(self.shard.n_global))))
