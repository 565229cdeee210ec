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


    # -- fused / sharded step API (buffer-oriented, mirrors genjax_amd.backend.ShardStepBuffers)
    SHARD_ROW = 7

    class _Buf:
        @property
        def cols(self):
            return self.sets[self.cur]

    def new_shard_buffers(self, n, n_global, world, rank, device, n_rays=0):
        b = self._Buf()
        b.row = self.SHARD_ROW
        z = lambda k, dt=torch.float32: torch.zeros(k, dtype=dt)  # noqa: E731
        b.n, b.n_global, b.world, b.rank, b.offset = n, n_global, world, rank, rank * n
        b.sets, b.cur = [[z(n) for _ in range(6)] for _ in range(2)], 0
        b.logw, b.stats, b.stats_all, b.gstats = z(n), z(5), z(5 * world), z(5)
        b.w, b.cdf = z(n), z(n)
        b.totals, b.all_totals, b.scanned = z(n // TILE), z(n_global // TILE), z(n_global // TILE)
        b.lc = max(0, self.bisect_levels(n_global) - 11)
        b.table, b.wants = z(1 << b.lc), z(2 * world, torch.int64)
        return b

    def loc_smc_extend(self, world, buf, keys3, is_initial, obs, offset=0):
        prev = None if is_initial else tuple(buf.cols[:3])
        out = self.loc_extend(world, keys3, is_initial, obs, prev, buf.logw, offset=offset)
        for c, k_ in zip(buf.cols, ("x", "y", "th", "mx", "my", "mth")):
            c.copy_(out[k_])
        buf.logw.copy_(out["logw"])
        buf.stats.copy_(self.logsumexp_ess(buf.logw))

    def shard_stats_fold(self, buf):
        a = self._np(buf.stats_all).reshape(buf.world, 5)
        m = a[:, 0].max()
        mm = m if np.isfinite(m) else np.float32(0)
        s = q = 0.0
        for r in range(buf.world):
            mr = a[r, 0] if np.isfinite(a[r, 0]) else np.float32(0)
            sc = np.exp(np.float64(mr) - np.float64(mm))
            s += 0.0 if a[r, 3] == 0 else np.float64(a[r, 3]) * sc
            q += 0.0 if a[r, 4] == 0 else np.float64(a[r, 4]) * sc * sc
        f1, f2 = np.float32(s), np.float32(q)
        buf.gstats.copy_(torch.tensor([m, np.float32(np.log(f1)) + mm, f1 * f1 / f2, f1, f2], dtype=torch.float32))
        return buf.gstats

    def shard_cdf_begin(self, buf):
        buf.w.copy_(self.normalized_weights(buf.logw, buf.gstats[1:2]))
        buf.totals.copy_(self.cumsum_tile_totals(buf.w))

    def shard_cdf_finish(self, buf):
        buf.scanned.copy_(self.cumsum(buf.all_totals))
        buf.cdf.copy_(self.cumsum_apply(buf.w, buf.scanned, buf.offset // TILE))
        plan = np.asarray(self.bisect_plan(buf.n_global, buf.lc)).astype(np.int64)
        t = np.zeros(1 << buf.lc, np.float32)
        own = (plan >= buf.offset) & (plan < buf.offset + buf.n) & (np.arange(plan.size) > 0)
        t[own] = self._np(buf.cdf)[plan[own] - buf.offset]
        buf.table.copy_(torch.from_numpy(t))

    def shard_windows(self, buf, u):
        buf.wants.copy_(self.searchsorted_window(buf.table, buf.lc, buf.n_global, u, 0, buf.n, buf.n_global,
                                                 n_ranges=buf.world).reshape(-1))
        return buf.wants

    def shard_pack(self, buf, pieces):
        rows = [torch.stack([buf.cdf[a:b]] + [c[a:b] for c in buf.cols], dim=1) for a, b in pieces]
        return torch.cat(rows).reshape(-1).contiguous()

    def shard_recv_rows(self, buf, rows):
        return torch.zeros(rows * self.SHARD_ROW, dtype=torch.float32)

    def shard_resample_finish(self, buf, u, lo, hi, recv, left_rows):
        rows = recv.reshape(-1, self.SHARD_ROW)
        loc0, loc1 = buf.offset, buf.offset + buf.n
        a, b = max(lo, loc0), min(hi + 1, loc1)                       # the local part of the window
        assert left_rows == max(0, min(hi + 1, loc0) - lo)
        assert rows.shape[0] - left_rows == max(0, hi + 1 - max(lo, loc1))
        local = torch.stack([buf.cdf] + list(buf.cols), dim=1)[max(a - loc0, 0):max(b - loc0, 0)]
        slab = torch.cat([rows[:left_rows], local, rows[left_rows:]])
        assert slab.shape[0] == hi - lo + 1
        anc = self.searchsorted_systematic_2l(buf.table, buf.lc, slab[:, 0].contiguous(), lo, buf.n_global, u,
                                              buf.offset, buf.n, buf.n_global)
        idx = anc.long() - lo
        for k_, dst in enumerate(buf.sets[1 - buf.cur]):
            dst.copy_(slab[idx, k_ + 1])
        buf.logw.zero_()
        buf.cur = 1 - buf.cur

    def shard_resample_moves(self, world, buf, u, lo, hi, recv, left_rows, kind, move_keys, is_initial, obs,
                             step_size=0.0, n_steps=1, sel_mask=7, events=None, allow_fast=True):
        self.shard_resample_finish(buf, u, lo, hi, recv, left_rows)
        self.loc_smc_resample_moves(world, buf, False, u, kind, move_keys, is_initial, obs, offset=buf.offset)
        return False

    def loc_smc_resample_moves(self, world, buf, do_resample, u, kind, move_keys, is_initial, obs, step_size=0.0,
                               n_steps=1, sel_mask=7, offset=0, accept_counts=None, events=None):
        assert not do_resample, "the sharded driver resamples through shard_resample_finish"
        if kind == "mh" and len(move_keys) > 0:
            pose = self.loc_mh(world, move_keys, is_initial, obs, tuple(buf.cols[:3]), tuple(buf.cols[3:]),
                               offset=offset)
            for c, p_ in zip(buf.cols[:3], pose):
                c.copy_(p_)
        else:
            assert kind == "none" or len(move_keys) == 0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, kind, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from genjax_amd.parallel import ShardInfo, global_normaliser, global_systematic_resample

    Bk = OracleBackend()
    n_global = TILE * 4
    rng = np.random.default_rng(11)
    scale = 0.05 if kind == "flat" else 6.0
    logw = (rng.normal(size=n_global) * scale - 50.0).astype(np.float32)
    cols = [rng.normal(size=n_global).astype(np.float32) for _ in range(3)]
    shard = ShardInfo(rank, world, n_global)
    lo, hi = shard.offset, shard.offset + shard.n_local
    stats = global_normaliser(Bk, torch.from_numpy(logw[lo:hi]), shard)
    out = global_systematic_resample(Bk, torch.from_numpy(logw[lo:hi]), stats[1:2], 0.4321,
                                     [torch.from_numpy(c[lo:hi]) for c in cols], shard)
    np.save(os.path.join(out_dir, f"stats_{rank}.npy"), stats.numpy())
    np.save(os.path.join(out_dir, f"cols_{rank}.npy"), np.stack([o.numpy() for o in out]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,world", [("flat", 2), ("peaked", 2), ("peaked", 4)])
def test_sharded_resampling_equals_single_process(tmp_path, kind, world, oracle):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, kind, str(tmp_path)), nprocs=world, join=True)
    n_global = TILE * 4
    rng = np.random.default_rng(11)
    scale = 0.05 if kind == "flat" else 6.0
    logw = (rng.normal(size=n_global) * scale - 50.0).astype(np.float32)
    cols = [rng.normal(size=n_global).astype(np.float32) for _ in range(3)]
    stats = [np.load(tmp_path / f"stats_{r}.npy") for r in range(world)]
    assert np.array_equal(stats[0], stats[1])                       # every rank holds the same scalars
    np.testing.assert_allclose(stats[0][1], oracle.logsumexp(logw), rtol=1e-6)
    np.testing.assert_allclose(stats[0][2], oracle.ess(logw), rtol=1e-4)
    # single-process reference with the SAME lse (the weights are an input of the ancestor contract)
    w = np.exp(logw - np.float32(stats[0][1])).astype(np.float32)
    anc = oracle.systematic_indices(w, 0.4321)
    got = np.concatenate([np.load(tmp_path / f"cols_{r}.npy") for r in range(world)], axis=1)
    for c in range(3):
        assert np.array_equal(got[c], cols[c][anc])


def _smc_worker(rank, world, port, n_global, K, fused, out_dir):
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
    st = smc.run()
    lml = smc.log_marginal_likelihood()
    np.save(os.path.join(out_dir, f"smc_{rank}.npy"), np.stack([st.x.numpy(), st.y.numpy(), st.th.numpy()]))
    np.save(os.path.join(out_dir, f"meta_{rank}.npy"), np.array([lml] + st.ess + [float(r) for r in st.resampled]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,fused", [(2, True), (4, True), (2, False)])
def test_sharded_localization_smc_equals_single_process_driver(tmp_path, oracle, world, fused):
    """ShardedLocalizationSMC on 2 gloo ranks (stages by the oracle stand-in) reproduces the
    oracle's own single-process rejuvenation_smc driver: same key schedule, same ESS trace and
    resampling decisions, same particles (up to the last-ulp difference of the two logsumexp
    summation orders, which can move an ancestor at a CDF boundary)."""
    n_global, K = 8192, 2
    port = _free_port()
    mp.spawn(_smc_worker, args=(world, port, n_global, K, fused, str(tmp_path)), nprocs=world, join=True)
    ow = oracle.basic_world()
    obs, k = oracle.localization_observations(ow)
    ref = oracle.loc_smc(k, n_global, obs[:6], ow, kernel="mh", K=K, method="systematic")
    got = np.concatenate([np.load(tmp_path / f"smc_{r}.npy") for r in range(world)], axis=1)
    meta = [np.load(tmp_path / f"meta_{r}.npy") for r in range(world)]
    assert np.array_equal(meta[0], meta[1])
    T = 6
    np.testing.assert_allclose(meta[0][1:1 + T], ref["ess"][:T], rtol=1e-4)
    assert np.array_equal(meta[0][1 + T:1 + 2 * T].astype(int), ref["resampled"][:T])
    np.testing.assert_allclose(meta[0][0], ref["lml"], rtol=1e-5)
    same = (got[0] == ref["state"].x) & (got[1] == ref["state"].y) & (got[2] == ref["state"].th)
    assert same.mean() > 0.99, same.mean()


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


def test_exchange_plan():
    """Rows a rank needs beyond its own shard come from the owners, ascending; rows inside the
    own shard never travel."""
    from genjax_amd.parallel import exchange_plan

    n, G = 10, 4
    wants = np.array([[0, 12], [8, 19], [25, 27], [5, 39]])
    # rank 1 owns [10, 20): sends [10, 13) to rank 0, nothing to 2, everything to rank 3
    pieces, in_rows, out_rows, left, any_remote = exchange_plan(wants, 1, n, G)
    assert pieces == [(0, 3), (0, 0), (0, 0), (0, 10)] and in_rows == [3, 0, 0, 10]
    assert out_rows == [2, 0, 0, 0] and left == 2 and any_remote
    # rank 3 owns [30, 40): receives [5, 30) from ranks 0..2, all of it left of its shard
    pieces, in_rows, out_rows, left, _ = exchange_plan(wants, 3, n, G)
    assert out_rows == [5, 10, 10, 0] and left == 25 and in_rows == [0, 0, 0, 0]
    # rank 2's window lies inside its shard
    assert exchange_plan(wants, 2, n, G)[2] == [0, 0, 0, 0]
    local_only = np.array([[0, 9], [10, 19], [22, 25], [30, 39]])
    assert exchange_plan(local_only, 0, n, G)[4] is False


def test_slab_plan():
    from genjax_amd.parallel import slab_plan

    assert slab_plan(5, 20, 10, 4) == [(5, 10), (0, 10), (0, 1), (0, 0)]
    assert slab_plan(0, 39, 10, 4) == [(0, 10)] * 4
    assert slab_plan(17, 17, 10, 4) == [(0, 0), (7, 8), (0, 0), (0, 0)]
