"""GPU tests of the sharded smc step (genjax_amd/parallel.py, gjx_shard_* entry points).

There is one GPU per test box, so (1) the shard kernels are driven for G ranks from ONE process
with the collectives done by hand (concatenate / add / slice): ancestors and gathered columns
must equal the single-device resampling bit for bit given the same lse; (2) the whole driver runs
as 2 and 4 gloo ranks sharing the GPU (device tensors staged through the host for transport; the
kernels are the product's) and must reproduce the 1-rank run."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TILE = 2048


@pytest.fixture(scope="module")
def B():
    from genjax_amd import backend

    assert torch.cuda.is_available()
    return backend


def _fill(buf, cols, logw, lo, hi):
    for dst, c in zip(buf.cols, cols):
        dst.copy_(c[lo:hi])
    buf.logw.copy_(logw[lo:hi])


@pytest.mark.parametrize("kind,G,tiles", [("flat", 2, 4), ("peaked", 4, 2), ("spike", 4, 3), ("peaked", 8, 16)])
def test_shard_kernels_equal_single_device_resampling(B, kind, G, tiles):
    from genjax_amd.parallel import exchange_plan

    dev = torch.device("cuda", 0)
    n = TILE * tiles
    N = n * G
    rng = np.random.default_rng(5 + G)
    scale = {"flat": 0.05, "peaked": 6.0, "spike": 0.5}[kind]
    lw = (rng.normal(size=N) * scale - 30.0).astype(np.float32)
    if kind == "spike":                      # nearly all mass in one shard: windows far from home
        lw[n + 17: n + 40] += 25.0
    logw = torch.from_numpy(lw).to(dev)
    cols = [torch.from_numpy(rng.normal(size=N).astype(np.float32)).to(dev) for _ in range(6)]
    u = 0.3711
    stats = B.logsumexp_ess(logw)
    w = B.normalized_weights(logw, stats[1:2])
    anc, cdf = B.resample_systematic(w, u, return_cdf=True)
    ref = B.gather_columns(anc, cols)

    bufs = [B.new_shard_buffers(n, N, G, r, dev) for r in range(G)]
    for r, b in enumerate(bufs):
        _fill(b, cols, logw, r * n, (r + 1) * n)
        b.gstats.copy_(stats)
        B.shard_cdf_begin(b)
    all_totals = torch.cat([b.totals for b in bufs])          # all_gather
    for b in bufs:
        b.all_totals.copy_(all_totals)
        B.shard_cdf_finish(b)
    table = torch.stack([b.table for b in bufs]).sum(0)       # all_reduce: x + 0 is exact
    for r, b in enumerate(bufs):
        b.table.copy_(table)
        assert torch.equal(b.cdf, cdf[r * n:(r + 1) * n])     # global jnp.cumsum association
    plan_mid = torch.from_numpy(np.asarray(B.bisect_plan(N, bufs[0].lc)).astype(np.int64)).to(dev)
    assert torch.equal(table[1:], cdf[plan_mid[1:]])
    wants = B.shard_windows(bufs[0], u).cpu().numpy().reshape(G, 2).copy()
    wants[:, 1] = np.minimum(wants[:, 1], N - 1)
    plans = [exchange_plan(wants, r, n, G) for r in range(G)]
    sends = [B.shard_pack(b, plans[r][0]).clone() for r, b in enumerate(bufs)]
    R = B.SHARD_ROW
    for r, b in enumerate(bufs):                              # all_to_all_single by hand
        pieces, in_rows, out_rows, left_rows, _ = plans[r]
        parts = []
        for p in range(G):
            start = sum(plans[p][1][:r]) * R
            parts.append(sends[p][start:start + plans[p][1][r] * R])
            assert plans[p][1][r] == out_rows[p]
        recv = B.shard_recv_rows(b, sum(out_rows))
        if recv.numel():
            recv.copy_(torch.cat(parts))
        B.shard_resample_finish(b, u, int(wants[r, 0]), int(wants[r, 1]), recv, left_rows)
        for got, want in zip(b.cols, ref):
            assert torch.equal(got, want[r * n:(r + 1) * n]), (kind, G, r)
        assert float(b.logw.abs().max()) == 0.0
    if kind == "spike":
        assert any(sum(p[2]) > 0 for p in plans)              # remote rows were really needed


def test_shard_stats_fold_matches_global_normaliser(B):
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(3)
    G, n = 4, TILE
    lw = (rng.normal(size=G * n) * 4 - 200.0).astype(np.float32)
    lw[n:2 * n] = -np.inf                                      # a shard without any weight
    logw = torch.from_numpy(lw).to(dev)
    want = B.logsumexp_ess(logw).cpu().numpy()
    b = B.new_shard_buffers(n, G * n, G, 0, dev)
    b.stats_all.copy_(torch.cat([B.logsumexp_ess(logw[r * n:(r + 1) * n].contiguous()) for r in range(G)]))
    got = B.shard_stats_fold(b).cpu().numpy()
    assert got[0] == want[0]
    np.testing.assert_allclose(got[1:3], want[1:3], rtol=2e-6)


# ------------------------------------------------------------------ the driver on gloo ranks sharing the GPU
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, n_global, kern, K, T, out_dir):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle as O
    from genjax_amd import backend as Bk
    from genjax_amd.parallel import ShardedLocalizationSMC, ShardInfo

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ow = O.basic_world()
    obs, k = O.localization_observations(ow)
    walls = np.stack([ow.x1, ow.y1, ow.x2, ow.y2], axis=1)
    gw = Bk.make_loc_world(ow.width, ow.height, walls, n_rays=8)
    general = kern.endswith("_general")                       # force the general resampling tail
    smc = ShardedLocalizationSMC(gw, obs[:T], n_global, k, kernel=kern.replace("_general", ""), n_moves=K,
                                 shard=ShardInfo(rank, world, n_global), device=dev)
    smc.halo_path = not general
    st = smc.run()
    lml = smc.log_marginal_likelihood()
    torch.cuda.synchronize()
    tag = f"{world}g" if general else f"{world}"
    np.save(os.path.join(out_dir, f"cols_{tag}_{rank}.npy"),
            np.stack([c.cpu().numpy() for c in (st.x, st.y, st.th, st.mx, st.my, st.mth, st.logw)]))
    np.save(os.path.join(out_dir, f"meta_{tag}_{rank}.npy"),
            np.array([lml] + st.ess + [float(r) for r in st.resampled] + [float(smc.fast_resamples)] +
                     [float(sum(smc.remote_rows))]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kern,K", [("mh", 3), ("hmc", 1)])
def test_sharded_driver_on_shared_gpu_equals_one_rank(tmp_path, kern, K):
    import torch.multiprocessing as mp

    n_global, T = 16384, 6
    runs = {}
    for world in (1, 2, 4):
        mp.spawn(_rank_main, args=(world, _free_port(), n_global, kern, K, T, str(tmp_path)), nprocs=world,
                 join=True)
        cols = np.concatenate([np.load(tmp_path / f"cols_{world}_{r}.npy") for r in range(world)], axis=1)
        metas = [np.load(tmp_path / f"meta_{world}_{r}.npy") for r in range(world)]
        for m in metas[1:]:
            assert np.array_equal(m[:-2], metas[0][:-2])       # every rank holds the same scalars
        if world > 1 and kern == "mh":                         # the halo path (gather + scores in the move launch)
            n_res = int(metas[0][1 + T:1 + 2 * T].sum())       # a window overhanging by more than the halo takes the
            assert all(0 < m[-2] <= n_res for m in metas), ([m[-2] for m in metas], n_res)   # general path
        runs[world] = (cols, metas[0])
    one_cols, one_meta = runs[1]
    for world in (2, 4):
        cols, meta = runs[world]
        np.testing.assert_allclose(meta[0], one_meta[0], rtol=1e-5)                  # log-ML
        np.testing.assert_allclose(meta[1:1 + T], one_meta[1:1 + T], rtol=1e-4)      # ESS trace
        assert np.array_equal(meta[1 + T:1 + 2 * T], one_meta[1 + T:1 + 2 * T])      # resampling decisions
        # the fold of the per-shard sums differs from one device in the last ulp of lse, which can move an
        # ancestor at a CDF boundary; everything else is the same particle for particle
        same = np.all(cols[:6] == one_cols[:6], axis=0)
        assert same.mean() > 0.99, (world, same.mean())


def test_halo_path_equals_general_path_bitwise(tmp_path):
    """Two ranks, same collectives and lse: the halo path (rows unpacked into the column halos, gather and
    stored scores inside the move launch) and the general path (piecewise gather, separate move launch that
    re-solves the starting lidar) produce the same particles bit for bit."""
    import torch.multiprocessing as mp

    n_global, T, K, world = 16384, 6, 3, 2
    for kern in ("mh", "mh_general"):
        mp.spawn(_rank_main, args=(world, _free_port(), n_global, kern, K, T, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        fast, general = np.load(tmp_path / f"cols_2_{r}.npy"), np.load(tmp_path / f"cols_2g_{r}.npy")
        assert np.array_equal(fast.view(np.uint32), general.view(np.uint32))
        mf, mg = np.load(tmp_path / f"meta_2_{r}.npy"), np.load(tmp_path / f"meta_2g_{r}.npy")
        assert np.array_equal(mf[:-2], mg[:-2]) and mf[-2] > 0 and mg[-2] == 0


def test_bench_runs_the_sharded_step_over_nccl_with_one_rank():
    """bench.py --force-sharded: the sharded choreography through a ONE-rank NCCL (RCCL) process group --
    the only RCCL exercise a one-GPU box allows (all_gather_into_tensor and all_reduce run, the row
    exchange has nothing to move).  Checks the bench contract's JSON line on the way."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-sharded", "--no-cpu-baseline",
                          "--particles", str(1 << 20), "--steps", "6", "--warmup", "2"],
                         capture_output=True, text=True, timeout=170, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    assert len(out.stdout.strip().splitlines()) == 1, out.stdout[:500]   # RCCL's banner must not reach stdout
    line = json.loads(out.stdout.strip().splitlines()[-1])
    for key_ in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key_ in line, key_
    assert line["value"] > 0 and line["steps"] == 6 and line["n_gpus"] == 1 and line["vs_baseline"] is None
    assert line["config"]["halo_path_resamplings"] > 0
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(line["roofline"])
