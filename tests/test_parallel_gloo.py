"""world_size-2 gloo tests (CPU) of the sharded SMC choreography in genjax_amd/parallel.py:
rank-ordered normaliser, association-exact sharded CDF, exact global bisection, slab exchange.
The kernels are replaced by a numpy stand-in built on the oracle (test infrastructure); the
product path always runs them in HIP."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

TILE = 2048


class OracleBackend:
    """numpy/oracle stand-in for genjax_amd.backend, CPU tensors in and out."""

    def __init__(self):
        import oracle

        self.O = oracle

    @staticmethod
    def _np(t):
        return t.detach().cpu().numpy().astype(np.float32)

    def logsumexp_ess(self, logw):
        a = self._np(logw)
        m = a.max()
        mm = m if np.isfinite(m) else 0.0
        e = np.exp((a - np.float32(mm)).astype(np.float32))
        s1, s2 = np.float32(e.astype(np.float64).sum()), np.float32((e * e).astype(np.float64).sum())
        return torch.tensor([m, np.log(s1) + mm, s1 * s1 / s2, s1, s2], dtype=torch.float32)

    def max_f32(self, a):
        return torch.tensor([self._np(a).max()], dtype=torch.float32)

    def sumexp_f32(self, a, m):
        mm = float(m[0]) if np.isfinite(float(m[0])) else 0.0
        e = np.exp((self._np(a) - np.float32(mm)).astype(np.float32))
        return torch.tensor([e.astype(np.float64).sum(), (e * e).astype(np.float64).sum()], dtype=torch.float32)

    def normalized_weights(self, logw, lse):
        return torch.from_numpy(np.exp(self._np(logw) - np.float32(lse[0])).astype(np.float32))

    def cumsum(self, w):
        return torch.from_numpy(self.O.cumsum(self._np(w)))

    def cumsum_tile_totals(self, w):
        a = self._np(w)
        return torch.from_numpy(np.array([self.O.cumsum(a[i:i + TILE])[-1] for i in range(0, a.size, TILE)],
                                         np.float32))

    def cumsum_apply(self, w, scanned, tile_offset):
        a, S = self._np(w), self._np(scanned)
        out = np.zeros_like(a)
        for ti in range(a.size // TILE):
            tile = a[ti * TILE:(ti + 1) * TILE]
            levels = [tile]
            while levels[-1].size > 1:
                levels.append((levels[-1][0::2] + levels[-1][1::2]).astype(np.float32))
            gt = ti + tile_offset
            carry = S[gt - 1] if gt > 0 else np.float32(0)
            for i in range(TILE):
                cnt = i + 1
                if cnt == TILE:
                    out[ti * TILE + i] = S[gt]
                    continue
                acc = carry
                for b in range(10, -1, -1):
                    if cnt & (1 << b):
                        acc = np.float32(acc + levels[b][(cnt >> (b + 1)) << 1])
                out[ti * TILE + i] = acc
        return torch.from_numpy(out)

    def searchsorted_systematic(self, cdf, u, j0, m, n_total):
        pos = ((np.arange(j0, j0 + m, dtype=np.int32).astype(np.float32) + np.float32(u)) /
               np.float32(n_total)).astype(np.float32)
        idx = self.O.searchsorted_left(self._np(cdf), pos)
        return torch.from_numpy(np.minimum(idx, cdf.numel() - 1).astype(np.int32))

    # -- two-level bisection (numpy emulation, independent of the kernels)
    @staticmethod
    def bisect_levels(n):
        levels = 0
        while (1 << levels) < n + 1:
            levels += 1
        return levels

    @staticmethod
    def bisect_plan(n, lc):
        from genjax_amd import backend as B   # host-only function of the C ABI (no GPU needed)

        return B.bisect_plan(n, lc)

    @staticmethod
    def _coarse(table, lc, n, pos):
        low, high, node = 0, n, 1
        for _ in range(lc):
            mid = (low + high) >> 1
            if pos <= table[node]:
                high, node = mid, 2 * node
            else:
                low, node = mid, 2 * node + 1
        return low, high, node

    def _positions(self, u, j0, m, n_total):
        return ((np.arange(j0, j0 + m, dtype=np.int32).astype(np.float32) + np.float32(u)) /
                np.float32(n_total)).astype(np.float32)

    def searchsorted_window(self, table, lc, n, u, j0, m, n_total, n_ranges=1):
        t = self._np(table)
        out = []
        for r in range(n_ranges):
            pos = self._positions(u, j0 + r * m, m, n_total)
            out.append([self._coarse(t, lc, n, pos[0])[0], self._coarse(t, lc, n, pos[-1])[1]])
        return torch.tensor(out, dtype=torch.int64)

    def searchsorted_systematic_2l(self, table, lc, slab, slab_base, n, u, j0, m, n_total):
        t, c = self._np(table), self._np(slab)
        levels = self.bisect_levels(n)
        out = np.zeros(m, np.int32)
        for k, pos in enumerate(self._positions(u, j0, m, n_total)):
            low, high, _ = self._coarse(t, lc, n, pos)
            for _ in range(levels - lc):
                mid = (low + high) >> 1
                assert 0 <= mid - slab_base < c.size, "probe outside the exchanged slab"
                if pos <= c[mid - slab_base]:
                    high = mid
                else:
                    low = mid
            out[k] = min(high, n - 1)
        return torch.from_numpy(out)

    def resample_systematic(self, w, u):
        return torch.from_numpy(self.O.systematic_indices(self._np(w), u))

    def gather_columns(self, idx, cols, idx_base=0):
        i = idx.long() - idx_base
        return [c[i] for c in cols]

    # -- localization stages (oracle restatement; `world` is the product's gjx_loc_world struct)
    def _oworld(self, w):
        walls = np.array([[w.walls[i][j] for j in range(4)] for i in range(w.n_walls)], np.float32)
        return self.O.World(w.width, w.height, walls)

    def _ostate(self, n, R, cols, logw=None):
        st = self.O.LocState(n, R)
        for name, c in cols.items():
            getattr(st, name)[:] = self._np(c)
        if logw is not None:
            st.logw[:] = self._np(logw)
        return st

    def loc_extend(self, world, keys3, is_initial, obs, prev, logw_in, out=None, offset=0):
        n = logw_in.numel()
        cols = {} if prev is None else dict(zip(("x", "y", "th"), prev))
        st = self._ostate(n, world.n_rays, cols, logw_in)
        self.O.loc_extend(st, keys3, is_initial, obs, self._oworld(world), offset=offset)
        return {k_: torch.from_numpy(getattr(st, k_).copy()) for k_ in ("x", "y", "th", "mx", "my", "mth", "logw")}

    def loc_mh(self, world, move_keys, is_initial, obs, pose_in, means, pose_out=None, sel_mask=7, offset=0,
               accept_counts=None):
        n = pose_in[0].numel()
        cols = dict(zip(("x", "y", "th", "mx", "my", "mth"), tuple(pose_in) + tuple(means)))
        st = self._ostate(n, world.n_rays, cols)
        ow = self._oworld(world)
        # per-ray scores of the current pose (the oracle keeps them as trace state)
        self.O.lib().orc_loc_ray_logpdfs_n(st.x, st.y, st.th, n, self.O._f(obs), st.R, *ow.args(), st.lp)
        for mk in np.asarray(move_keys):
            self.O.loc_mh(st, mk, is_initial, obs, ow, offset=offset)
        return tuple(torch.from_numpy(getattr(st, k_).copy()) for k_ in ("x", "y", "th"))


    # -- fused / sharded step API (buffer-oriented, mirrors genjax_amd.backend.ShardStepBuffers).  Every stage is
    # restated from its DEFINITION (brute force over all positions where the kernels walk trees), so that the
    # choreography of genjax_amd/parallel.py is checked against an independent statement of what must come out.
    SHARD_ROW = 7

    class _Buf:
        @property
        def cols(self):
            return self.sets[self.cur]

    def new_shard_buffers(self, n, n_global, world, rank, device, n_rays=0):
        b = self._Buf()
        z = lambda k, dt=torch.float32: torch.zeros(k, dtype=dt)  # noqa: E731
        b.n, b.n_global, b.world, b.rank, b.offset, b.halo = n, n_global, world, rank, rank * n, n
        nt = n // TILE
        b.sets, b.cur = [[z(n) for _ in range(6)] for _ in range(2)], 0
        b.logw, b.stats = z(n), z(5)
        b.max1, b.max_all, b.tiles, b.tiles_all, b.gstats = z(1), z(world), z(2 * nt), z(2 * nt * world), z(5)
        b.msg, b.msg_all = z(nt + 2 * TILE + n), z((nt + 2 * TILE + n) * world)
        b.cdf_ext = z(n + 2 * TILE)
        b.lc = max(0, self.bisect_levels(n_global) - 11)
        b.table = z(1 << b.lc)
        b.counts, b.counts_all = z(world, torch.int32), z(world * world, torch.int32)
        b.err = z(1, torch.int32)
        return b

    def loc_shard_extend(self, world, buf, keys3, is_initial, obs):
        prev = None if is_initial else tuple(buf.cols[:3])
        out = self.loc_extend(world, keys3, is_initial, obs, prev, buf.logw, offset=buf.offset)
        for c, k_ in zip(buf.cols, ("x", "y", "th", "mx", "my", "mth")):
            c.copy_(out[k_])
        buf.logw.copy_(out["logw"])
        self.shard_local_max(buf)

    def shard_local_max(self, buf):
        buf.max1.copy_(self.max_f32(buf.logw))

    def shard_tile_sums(self, buf):
        m = self._np(buf.max_all).max()
        mm = np.float32(m if np.isfinite(m) else 0.0)
        e = np.exp((self._np(buf.logw) - mm).astype(np.float32)).reshape(-1, TILE).astype(np.float64)
        t = np.stack([e.sum(axis=1), (e * e).sum(axis=1)], axis=1).astype(np.float32)   # one rounding per tile
        buf.tiles.copy_(torch.from_numpy(t.reshape(-1)))

    def shard_stats_fold(self, buf):
        t = self._np(buf.tiles_all).reshape(-1, 2).astype(np.float64)
        m = self._np(buf.max_all).max()
        mm = np.float32(m if np.isfinite(m) else 0.0)
        f1, f2 = np.float32(t[:, 0].sum()), np.float32(t[:, 1].sum())                   # global tile order
        lse = np.float32(np.log(f1)) + mm
        ess = np.float32(np.exp(2.0 * (np.float64(lse) - np.float64(mm))) / np.float64(f2))
        buf.gstats.copy_(torch.tensor([m, lse, ess, f1, f2], dtype=torch.float32))
        return buf.gstats

    def shard_cdf_begin(self, buf):
        # (the stand-in's message carries the shard's whole weight vector: every rank can then state the coarse
        # table from its definition, cdf[mid] of the global cumsum -- the product sends 12 floats per probe)
        w = self.normalized_weights(buf.logw, buf.gstats[1:2])
        buf.msg.copy_(torch.cat([self.cumsum_tile_totals(w), buf.logw[:TILE], buf.logw[-TILE:], w]))

    def shard_cdf_finish(self, buf):
        nt = buf.n // TILE
        msgs = buf.msg_all.reshape(buf.world, -1)
        scanned = self.cumsum(msgs[:, :nt].reshape(-1).contiguous())
        lse = buf.gstats[1:2]
        own = self.cumsum_apply(self.normalized_weights(buf.logw, lse), scanned, buf.rank * nt)
        left = right = torch.zeros(TILE)
        if buf.rank > 0:
            left = self.cumsum_apply(self.normalized_weights(msgs[buf.rank - 1, nt + TILE:nt + 2 * TILE], lse), scanned,
                                     buf.rank * nt - 1)
        if buf.rank + 1 < buf.world:
            right = self.cumsum_apply(self.normalized_weights(msgs[buf.rank + 1, nt:nt + TILE], lse), scanned,
                                      (buf.rank + 1) * nt)
        buf.cdf_ext.copy_(torch.cat([left, own, right]))
        plan = np.asarray(self.bisect_plan(buf.n_global, buf.lc)).astype(np.int64)
        cdf = self._np(self.cumsum(msgs[:, nt + 2 * TILE:].reshape(-1).contiguous()))   # the global cumsum
        assert np.array_equal(cdf[buf.offset:buf.offset + buf.n], self._np(own))
        t = np.zeros(1 << buf.lc, np.float32)
        t[1:] = cdf[plan[1:]]
        buf.table.copy_(torch.from_numpy(t))

    def shard_search(self, buf, u):
        """Every position walks the coarse levels; the ones whose interval can reach this shard walk the rest on
        the resident CDF slice; rows = distinct (ancestor here, destination elsewhere) pairs."""
        N, n, r = buf.n_global, buf.n, buf.rank
        t, c = self._np(buf.table), self._np(buf.cdf_ext)
        pos = self._positions(u, 0, N, N)
        low, high, node = np.zeros(N, np.int64), np.full(N, N, np.int64), np.ones(N, np.int64)
        for _ in range(buf.lc):
            mid = (low + high) >> 1
            left = pos <= t[node]
            high, low, node = np.where(left, mid, high), np.where(left, low, mid), 2 * node + (~left)
        cand = np.flatnonzero((high >= r * n) & (high < (r + 1) * n + TILE))
        assert cand.size == 0 or np.array_equal(cand, np.arange(cand[0], cand[-1] + 1))   # one range of positions
        lo, hi = low[cand], high[cand]
        for _ in range(self.bisect_levels(N) - buf.lc):
            mid = (lo + hi) >> 1
            k = mid - (r * n - TILE)
            assert ((k >= 0) & (k < c.size)).all(), "probe outside the resident slice"
            left = pos[cand] <= c[k]
            hi, lo = np.where(left, mid, hi), np.where(left, lo, mid)
        anc = np.minimum(hi, N - 1)
        buf.cand_j, buf.cand_anc = cand, anc
        own = (anc >= r * n) & (anc < (r + 1) * n)
        dest = cand // n
        first = np.ones(cand.size, bool)
        first[1:] = (anc[1:] != anc[:-1]) | (dest[1:] != dest[:-1])
        buf.flag = own & (dest != r) & first
        buf.counts.copy_(torch.from_numpy(np.bincount(dest[buf.flag], minlength=buf.world).astype(np.int32)))
        return buf.counts

    def shard_pack(self, buf):
        j, a = buf.cand_j[buf.flag], buf.cand_anc[buf.flag] - buf.offset
        n_rows = j.size
        rows = np.zeros((n_rows, self.SHARD_ROW), np.float32)
        rows[:, 0] = j.astype(np.int32).view(np.float32)
        for c_, col in enumerate(buf.cols):
            rows[:, 1 + c_] = self._np(col)[a]
        buf._send = torch.from_numpy(rows.reshape(-1))
        return buf._send

    def shard_send_rows(self, buf, n_rows):
        assert buf._send.numel() == n_rows * self.SHARD_ROW
        return buf._send

    def shard_recv_rows(self, buf, rows):
        assert rows <= buf.halo
        return torch.zeros(rows * self.SHARD_ROW, dtype=torch.float32)

    def shard_resample_moves(self, world, buf, recv, kind, move_keys, is_initial, obs, step_size=0.0, n_steps=1,
                             sel_mask=7, events=None, accept_counts=None):
        n, r = buf.n, buf.rank
        rows = self._np(recv).reshape(-1, self.SHARD_ROW)
        starts = rows[:, 0].copy().view(np.int32).astype(np.int64)
        assert np.all(np.diff(starts) > 0)                       # rows arrive in position order
        ext = [np.concatenate([self._np(c), rows[:, 1 + k]]) for k, c in enumerate(buf.cols)]
        anc = np.full(n, -1, np.int64)
        own = (buf.cand_anc >= r * n) & (buf.cand_anc < (r + 1) * n) & (buf.cand_j // n == r)
        anc[buf.cand_j[own] - r * n] = buf.cand_anc[own] - r * n
        rest = np.flatnonzero(anc < 0)
        slot = np.searchsorted(starts, rest + r * n, side="right") - 1
        if (slot < 0).any():
            buf.err.fill_(1)
        anc[rest] = n + np.maximum(slot, 0)
        for k, dst in enumerate(buf.sets[1 - buf.cur]):
            dst.copy_(torch.from_numpy(ext[k][anc]))
        buf.logw.zero_()
        buf.cur = 1 - buf.cur
        self.loc_smc_resample_moves(world, buf, False, 0.0, kind, move_keys, is_initial, obs, offset=buf.offset)

    # -- the overlapped form of the same step: split by definition (which positions have an own ancestor), every
    # particle moved by the same oracle move, each launch keeping only the positions it is responsible for
    def _local_anc(self, buf):
        n, r = buf.n, buf.rank
        anc = np.full(n, -1, np.int64)
        own = (buf.cand_anc >= r * n) & (buf.cand_anc < (r + 1) * n) & (buf.cand_j // n == r)
        anc[buf.cand_j[own] - r * n] = buf.cand_anc[own] - r * n
        return anc

    def shard_split(self, buf, n_moves, accept_counts=None):
        anc = self._local_anc(buf)
        loc = np.flatnonzero(anc >= 0)
        runs = 0 if loc.size == 0 else 1 + int((np.diff(loc) > 1).sum())
        go = float(buf.gstats[2]) < buf.n_global // 2
        assert runs <= 1                                          # positions and ancestors both increase
        buf.split = torch.tensor([loc[0] + 1 if loc.size else 0, loc[-1] + 1 if loc.size else 0, int(go), 0],
                                 dtype=torch.int32)
        return buf.split

    def _moved(self, world, buf, ext, anc, move_keys, is_initial, obs):
        """The K moves of the copies ext[anc] at this rank's positions -> six columns."""
        cols = [torch.from_numpy(e[anc]) for e in ext]
        pose = self.loc_mh(world, move_keys, is_initial, obs, tuple(cols[:3]), tuple(cols[3:]), offset=buf.offset)
        return list(pose) + cols[3:]

    def shard_moves_main(self, world, buf, move_keys, is_initial, obs, sel_mask=7, events=None, accept_counts=None):
        go, anc = bool(buf.split[2]), self._local_anc(buf)
        buf.anc_main = anc
        ext = [self._np(c) for c in buf.cols]
        src = np.where(anc >= 0, anc, 0) if go else np.arange(buf.n)
        keep = torch.from_numpy(anc >= 0) if go else torch.ones(buf.n, dtype=torch.bool)
        out = self._moved(world, buf, ext, src, move_keys, is_initial, obs)
        for dst, o in zip(buf.sets[1 - buf.cur], out):
            dst[keep] = o[keep]
        if go:
            buf.logw[keep] = 0.0
        buf.cur = 1 - buf.cur

    def shard_moves_fixup(self, world, buf, recv, p0, p1, move_keys, is_initial, obs, sel_mask=7, accept_counts=None):
        n, r = buf.n, buf.rank
        rows = self._np(recv).reshape(-1, self.SHARD_ROW)
        starts = rows[:, 0].copy().view(np.int32).astype(np.int64)
        assert np.all(np.diff(starts) > 0)
        rest = np.concatenate([np.arange(0, p0), np.arange(p1, n)])
        assert np.array_equal(rest, np.flatnonzero(buf.anc_main < 0))
        if rest.size == 0:
            return
        prev = buf.sets[1 - buf.cur]                              # shard_moves_main flipped the sets
        ext = [np.concatenate([self._np(c), rows[:, 1 + k]]) for k, c in enumerate(prev)]
        slot = np.searchsorted(starts, rest + r * n, side="right") - 1
        if (slot < 0).any():
            buf.err.fill_(1)
        anc = np.zeros(n, np.int64)
        anc[rest] = n + np.maximum(slot, 0)
        out = self._moved(world, buf, ext, anc, move_keys, is_initial, obs)
        idx = torch.from_numpy(rest)
        for dst, o in zip(buf.cols, out):
            dst[idx] = o[idx]
        buf.logw[idx] = 0.0

    def loc_smc_resample_moves(self, world, buf, do_resample, u, kind, move_keys, is_initial, obs, step_size=0.0,
                               n_steps=1, sel_mask=7, offset=0, accept_counts=None, events=None):
        assert not do_resample, "the sharded driver resamples through shard_resample_moves"
        if kind == "mh" and len(move_keys) > 0:
            pose = self.loc_mh(world, move_keys, is_initial, obs, tuple(buf.cols[:3]), tuple(buf.cols[3:]),
                               offset=offset)
            for c, p_ in zip(buf.cols[:3], pose):
                c.copy_(p_)
        else:
            assert kind == "none" or len(move_keys) == 0

    def loc_move_keys(self, scan_key, n_moves, kind):
        from genjax_amd import backend as B   # host-only function of the C ABI

        return B.loc_move_keys(scan_key, n_moves, kind)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _resample_worker(rank, world, port, kind, tiles, out_dir):
    """One resampling of a synthetic weight vector through the sharded choreography (normaliser included)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from genjax_amd.parallel import Comm, ShardInfo

    Bk = OracleBackend()
    n = TILE * tiles
    n_global = n * world
    logw, cols = _synthetic(kind, n_global, n)
    shard = ShardInfo(rank, world, n_global)
    lo, hi = shard.offset, shard.offset + n
    buf = Bk.new_shard_buffers(n, n_global, world, rank, None)
    for dst, c in zip(buf.cols, cols):
        dst.copy_(torch.from_numpy(c[lo:hi]))
    buf.logw.copy_(torch.from_numpy(logw[lo:hi]))
    comm = Comm(None, world)
    Bk.shard_local_max(buf)
    comm.all_gather_into(buf.max_all, buf.max1)
    Bk.shard_tile_sums(buf)
    comm.all_gather_into(buf.tiles_all, buf.tiles)
    stats = Bk.shard_stats_fold(buf).clone()
    Bk.shard_cdf_begin(buf)
    comm.all_gather_into(buf.msg_all, buf.msg)
    Bk.shard_cdf_finish(buf)
    Bk.shard_search(buf, 0.4321)
    comm.all_gather_into(buf.counts_all, buf.counts)
    counts = buf.counts_all.numpy().reshape(world, world)
    to_peer, from_peer = counts[rank].tolist(), counts[:, rank].tolist()
    Bk.shard_pack(buf)
    send, recv = Bk.shard_send_rows(buf, sum(to_peer)), Bk.shard_recv_rows(buf, sum(from_peer))
    before = comm.bytes_in["all_to_all"]
    if counts.sum() > 0:
        comm.all_to_all(recv, send, [7 * r for r in from_peer], [7 * r for r in to_peer])
    received = comm.bytes_in["all_to_all"] - before
    Bk.shard_resample_moves(None, buf, recv, "none", [], False, None)
    np.save(os.path.join(out_dir, f"stats_{rank}.npy"), stats.numpy())
    np.save(os.path.join(out_dir, f"cols_{rank}.npy"), np.stack([c.numpy() for c in buf.cols]))
    np.save(os.path.join(out_dir, f"meta_{rank}.npy"), np.array([received, int(buf.err.item()), sum(from_peer)]))
    dist.barrier()
    dist.destroy_process_group()


def _synthetic(kind, n_global, n):
    rng = np.random.default_rng(11)
    scale = {"flat": 0.05, "peaked": 6.0, "spike": 0.5}[kind]
    logw = (rng.normal(size=n_global) * scale - 50.0).astype(np.float32)
    if kind == "spike":
        logw[n + 17:n + 40] += 25.0
    cols = [rng.normal(size=n_global).astype(np.float32) for _ in range(6)]
    return logw, cols


@pytest.mark.parametrize("kind,world,tiles", [("flat", 2, 2), ("peaked", 2, 2), ("peaked", 4, 1), ("spike", 4, 1),
                                              ("spike", 8, 1), ("peaked", 3, 1)])
def test_sharded_resampling_equals_single_process(tmp_path, kind, world, tiles, oracle):
    """world ranks over gloo: same lse / ESS bits on every rank and for every world size; the gathered columns are
    the single-process systematic resampling of the same weights; a rank receives one 28-byte row per distinct
    remote ancestor of its positions (<= 28 B x n_local), also when one shard holds all the weight."""
    port = _free_port()
    mp.spawn(_resample_worker, args=(world, port, kind, tiles, str(tmp_path)), nprocs=world, join=True)
    n = TILE * tiles
    n_global = n * world
    logw, cols = _synthetic(kind, n_global, n)
    stats = [np.load(tmp_path / f"stats_{r}.npy") for r in range(world)]
    for s_ in stats[1:]:
        assert np.array_equal(s_, stats[0])                          # every rank holds the same scalars
    np.testing.assert_allclose(stats[0][1], oracle.logsumexp(logw), rtol=1e-6)
    np.testing.assert_allclose(stats[0][2], oracle.ess(logw), rtol=1e-4)
    # G-invariance of the normaliser: the stand-in's one-rank fold of the same tiles
    Bk = OracleBackend()
    one = Bk.new_shard_buffers(n_global, n_global, 1, 0, None)
    one.logw.copy_(torch.from_numpy(logw))
    Bk.shard_local_max(one)
    one.max_all.copy_(one.max1)
    Bk.shard_tile_sums(one)
    one.tiles_all.copy_(one.tiles)
    assert np.array_equal(Bk.shard_stats_fold(one).numpy(), stats[0])
    # single-process reference with the SAME lse (the weights are an input of the ancestor contract)
    w = np.exp(logw - np.float32(stats[0][1])).astype(np.float32)
    anc = oracle.systematic_indices(w, 0.4321)
    got = np.concatenate([np.load(tmp_path / f"cols_{r}.npy") for r in range(world)], axis=1)
    for c in range(6):
        assert np.array_equal(got[c], cols[c][anc])
    for r in range(world):
        received, err, rows = np.load(tmp_path / f"meta_{r}.npy")
        a_r = anc[r * n:(r + 1) * n]
        remote = np.unique(a_r[(a_r < r * n) | (a_r >= (r + 1) * n)]).size
        assert err == 0 and rows == remote and received == 28 * remote <= 28 * n, (r, received, remote)


def _smc_worker(rank, world, port, n_global, K, fused, out_dir, overlap=True):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle as O
    from genjax_amd import backend as B
    from genjax_amd.parallel import ShardedLocalizationSMC, ShardInfo

    ow = O.basic_world()
    obs, k = O.localization_observations(ow)
    walls = np.stack([ow.x1, ow.y1, ow.x2, ow.y2], axis=1)
    gw = B.make_loc_world(ow.width, ow.height, walls, n_rays=8)
    smc = ShardedLocalizationSMC(gw, obs[:6], n_global, k, kernel="mh", n_moves=K,
                                 shard=ShardInfo(rank, world, n_global), backend=OracleBackend())
    smc.fused = fused
    smc.force_sharded = fused
    smc.overlap = overlap
    st = smc.run()
    lml = smc.log_marginal_likelihood()
    assert (smc.overlapped_steps > 0) == (overlap and fused), smc.overlapped_steps
    if world == 1:                                     # a second run on the same key: per-stage keys looked up
        first = (np.stack([st.x.numpy(), st.y.numpy(), st.th.numpy()]).copy(), lml, list(st.ess))
        n_cached = len(smc._sched)
        smc.reset()
        st = smc.run()
        lml = smc.log_marginal_likelihood()
        assert len(smc._sched) == n_cached > 0
        assert np.array_equal(first[0], np.stack([st.x.numpy(), st.y.numpy(), st.th.numpy()]))
        assert first[1] == lml and first[2] == list(st.ess)
    np.save(os.path.join(out_dir, f"smc_{rank}.npy"), np.stack([st.x.numpy(), st.y.numpy(), st.th.numpy()]))
    np.save(os.path.join(out_dir, f"meta_{rank}.npy"),
            np.array([lml] + st.ess + [float(r) for r in st.resampled] +
                     [float(max(smc.remote_rows + [0])) if fused else 0.0]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,overlap", [(1, True), (2, True), (4, True), (8, True), (2, False), (4, False)])
def test_sharded_localization_smc_equals_single_process_driver(tmp_path, oracle, world, overlap):
    """ShardedLocalizationSMC on 1, 2, 4, 8 gloo ranks (stages by the oracle stand-in) reproduces the
    oracle's own single-process rejuvenation_smc driver: same key schedule, same ESS trace and
    resampling decisions, same particles (up to the last-ulp difference between the stand-in's tile-wise
    logsumexp and the oracle's, which can move an ancestor at a CDF boundary), and the SAME BITS for every
    world size (checked against the one-rank run by the 2-, 4- and 8-rank cases)."""
    n_global, K, T = 16384, 2, 6
    port = _free_port()
    mp.spawn(_smc_worker, args=(world, port, n_global, K, True, str(tmp_path), overlap), nprocs=world, join=True)
    ow = oracle.basic_world()
    obs, k = oracle.localization_observations(ow)
    ref = oracle.loc_smc(k, n_global, obs[:T], ow, kernel="mh", K=K, method="systematic")
    got = np.concatenate([np.load(tmp_path / f"smc_{r}.npy") for r in range(world)], axis=1)
    meta = [np.load(tmp_path / f"meta_{r}.npy") for r in range(world)]
    for m in meta:
        assert np.array_equal(m[:-1], meta[0][:-1])
        assert m[-1] <= n_global // world                            # rows received per resampling <= n_local
    np.testing.assert_allclose(meta[0][1:1 + T], ref["ess"][:T], rtol=1e-4)
    assert np.array_equal(meta[0][1 + T:1 + 2 * T].astype(int), ref["resampled"][:T])
    np.testing.assert_allclose(meta[0][0], ref["lml"], rtol=1e-5)
    same = (got[0] == ref["state"].x) & (got[1] == ref["state"].y) & (got[2] == ref["state"].th)
    assert same.mean() > 0.99, same.mean()
    if world > 1:                                                    # bit-identical to the one-rank run
        sub = tmp_path / "one"
        sub.mkdir()
        # (the one-rank run takes the OTHER form of the step: overlapped and two-phase give the same bits)
        mp.spawn(_smc_worker, args=(1, _free_port(), n_global, K, True, str(sub), not overlap), nprocs=1, join=True)
        one = np.load(sub / "smc_0.npy")
        assert np.array_equal(got.view(np.uint32), one.view(np.uint32))
        assert np.array_equal(np.load(sub / "meta_0.npy")[:-1], meta[0][:-1])


def test_kernel_by_kernel_step_is_single_rank_only():
    from genjax_amd.parallel import ShardedLocalizationSMC, ShardInfo

    smc = ShardedLocalizationSMC(None, np.zeros((2, 8), np.float32), 4096, np.array([0, 1], np.uint32),
                                 shard=ShardInfo(0, 2, 4096), backend=OracleBackend())
    smc.fused = False
    with pytest.raises(ValueError, match="one rank only"):
        smc.init()


def test_bisect_plan_matches_python():
    """Host function of the C ABI: probe index of every node of the coarse bisection tree."""
    from genjax_amd import backend as B

    for n, lc in ((8192, 2), (8192, 3), (1000003, 9), (1 << 24, 14), (5, 3)):
        plan = B.bisect_plan(n, lc)
        state = {1: (0, n)}
        for i in range(1, 1 << lc):
            low, high = state[i]
            mid = (low + high) >> 1
            assert plan[i] == mid, (n, lc, i)
            state[2 * i], state[2 * i + 1] = (low, mid), (mid, high)
    assert B.bisect_levels(1 << 24) == 25 and B.bisect_levels((1 << 24) - 1) == 24
