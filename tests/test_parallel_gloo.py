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


def _smc_worker(rank, world, port, n_global, K, out_dir):
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
    st = smc.run()
    lml = smc.log_marginal_likelihood()
    np.save(os.path.join(out_dir, f"smc_{rank}.npy"), np.stack([st.x.numpy(), st.y.numpy(), st.th.numpy()]))
    np.save(os.path.join(out_dir, f"meta_{rank}.npy"), np.array([lml] + st.ess + [float(r) for r in st.resampled]))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_localization_smc_equals_single_process_driver(tmp_path, oracle):
    """ShardedLocalizationSMC on 2 gloo ranks (stages by the oracle stand-in) reproduces the
    oracle's own single-process rejuvenation_smc driver: same key schedule, same ESS trace and
    resampling decisions, same particles (up to the last-ulp difference of the two logsumexp
    summation orders, which can move an ancestor at a CDF boundary)."""
    world, n_global, K = 2, 4096, 2
    port = _free_port()
    mp.spawn(_smc_worker, args=(world, port, n_global, K, str(tmp_path)), nprocs=world, join=True)
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


def test_slab_plan():
    from genjax_amd.parallel import slab_plan

    assert slab_plan(5, 20, 10, 4) == [(5, 10), (0, 10), (0, 1), (0, 0)]
    assert slab_plan(0, 39, 10, 4) == [(0, 10)] * 4
    assert slab_plan(17, 17, 10, 4) == [(0, 0), (7, 8), (0, 0), (0, 0)]
