"""GPU tests of the sharded smc step (genjax_amd/parallel.py, gjx_shard_* entry points).

There is one GPU per test box, so (1) the shard kernels are driven for G = 1, 2, 3, 4, 8 ranks from ONE
process with the collectives done by hand (concatenate / add / slice): lse, ESS, CDF, ancestors and gathered
columns must equal the single-device step bit for bit; (2) the whole driver runs as 1, 2, 4 and 8 gloo ranks
sharing the GPU (device tensors staged through the host for transport; the kernels are the product's) and
must reproduce the single-device fused step bit for bit."""
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


def _weights(kind, N, n, rng):
    scale = {"flat": 0.05, "peaked": 6.0, "spike": 0.5, "empty_shard": 4.0}[kind]
    lw = (rng.normal(size=N) * scale - 30.0).astype(np.float32)
    if kind == "spike":                      # nearly all mass in a few particles of one shard
        lw[n + 17: n + 40] += 25.0
    if kind == "empty_shard":                # a shard without any weight
        lw[n:2 * n] = -np.inf
    return lw


@pytest.mark.parametrize("kind,G,tiles", [("flat", 1, 4), ("flat", 2, 4), ("peaked", 4, 2), ("spike", 4, 3),
                                          ("peaked", 8, 16), ("spike", 8, 2), ("peaked", 3, 5), ("empty_shard", 4, 2),
                                          ("flat", 8, 1), ("peaked", 2, 2112),    # > 2048 tile totals
                                          ("peaked", 1, 1), ("peaked", 2, 1)])     # no / one coarse level
def test_sharded_step_equals_single_device_bitwise(B, kind, G, tiles):
    """The shard kernels driven for G ranks from one process, collectives by hand (concatenate / add / slice):
    lse, ESS, the CDF slices, the ancestors of every candidate position and the gathered columns have the bits
    of the single-device normaliser and resampling; the exchange carries one 28-byte row per distinct remote
    ancestor and destination, never more than one per receiving position."""
    dev = torch.device("cuda", 0)
    n = TILE * tiles
    N = n * G
    rng = np.random.default_rng(5 + G)
    lw = _weights(kind, N, n, rng)
    logw = torch.from_numpy(lw).to(dev)
    cols = [torch.from_numpy(rng.normal(size=N).astype(np.float32)).to(dev) for _ in range(6)]
    u = 0.3711
    stats = B.logsumexp_ess(logw)                                  # one device
    anc, cdf = B.resample_systematic_logw(logw, stats[1:2], u, return_cdf=True)
    ref = B.gather_columns(anc, cols)
    anc_h = anc.cpu().numpy().astype(np.int64)

    bufs = [B.new_shard_buffers(n, N, G, r, dev) for r in range(G)]
    # -- normaliser: all_gather(max), all_gather(tile sums), fold
    for r, b in enumerate(bufs):
        _fill(b, cols, logw, r * n, (r + 1) * n)
        B.shard_local_max(b)
    max_all = torch.cat([b.max1 for b in bufs])
    for b in bufs:
        b.max_all.copy_(max_all)
        B.shard_tile_sums(b)
    tiles_all = torch.cat([b.tiles for b in bufs])
    for b in bufs:
        b.tiles_all.copy_(tiles_all)
        got = B.shard_stats_fold(b)
        assert torch.equal(got.view(torch.int32), stats.view(torch.int32)), (got, stats)   # the bits of one device
    # -- CDF: all_gather(message)
    for b in bufs:
        B.shard_cdf_begin(b)
    msg_all = torch.cat([b.msg for b in bufs])
    for b in bufs:
        b.msg_all.copy_(msg_all)
        B.shard_cdf_finish(b)
    plan_mid = torch.from_numpy(np.asarray(B.bisect_plan(N, bufs[0].lc)).astype(np.int64)).to(dev)
    for r, b in enumerate(bufs):
        lo, hi = max(0, r * n - TILE), min(N, (r + 1) * n + TILE)
        ext = b.cdf_ext[lo - (r * n - TILE): hi - (r * n - TILE)]
        assert torch.equal(ext, cdf[lo:hi]), (kind, G, r)          # own slice + one tile of either neighbour
        # the COMPLETE coarse table on every rank, from the owners' probe records: no reduction over ranks
        assert torch.equal(b.table[1:], cdf[plan_mid[1:]]), (kind, G, r)
    # -- owner-side search, rows per destination
    counts = np.zeros((G, G), np.int64)
    for r, b in enumerate(bufs):
        counts[r] = B.shard_search(b, u).cpu().numpy()
        ja, jb = b.range.cpu().tolist()
        own = np.flatnonzero((anc_h >= r * n) & (anc_h < (r + 1) * n))
        if own.size:
            assert ja <= own[0] and own[-1] < jb, (kind, G, r, ja, jb, own[0], own[-1])   # nothing missed
        assert 0 <= ja <= jb <= N
        assert np.array_equal(b.cand_idx[:jb - ja].cpu().numpy(), anc_h[ja:jb])              # same bisection
        # expected rows: distinct (ancestor, destination) pairs with the ancestor here and the position elsewhere
        want = np.zeros(G, np.int64)
        for d in range(G):
            if d != r:
                a_d = anc_h[d * n:(d + 1) * n]
                want[d] = np.unique(a_d[(a_d >= r * n) & (a_d < (r + 1) * n)]).size
        assert np.array_equal(counts[r], want), (kind, G, r, counts[r], want)
    # -- all_to_all of the rows, by hand
    sends = [B.shard_pack(b)[:int(counts[r].sum()) * B.SHARD_ROW].clone() for r, b in enumerate(bufs)]
    R = B.SHARD_ROW
    for r, b in enumerate(bufs):
        parts = [sends[p][int(counts[p, :r].sum()) * R: int(counts[p, :r + 1].sum()) * R] for p in range(G)]
        n_recv = int(counts[:, r].sum())
        recv = B.shard_recv_rows(b, n_recv)
        if n_recv:
            recv.copy_(torch.cat(parts))
        a_r = anc_h[r * n:(r + 1) * n]
        assert n_recv == np.unique(a_r[(a_r < r * n) | (a_r >= (r + 1) * n)]).size <= n   # <= 28 B x n_local received
        B.shard_resample_moves(_any_world(B), b, recv, "none", np.zeros((0, 4, 2), np.uint32), False,
                               np.zeros(8, np.float32))
        for got_c, want_c in zip(b.cols, ref):
            assert torch.equal(got_c, want_c[r * n:(r + 1) * n]), (kind, G, r)
        assert float(b.logw.abs().max()) == 0.0 and int(b.err.item()) == 0
    if kind == "spike" and G > 1:
        assert counts.sum() > 0                                    # remote rows were really needed
        assert counts.sum() <= 23 * G                              # ... and only the selected particles travelled


_WORLD = {}


def _any_world(B):
    if "w" not in _WORLD:
        import oracle as O

        ow = O.basic_world()
        _WORLD["w"] = B.make_loc_world(ow.width, ow.height, np.stack([ow.x1, ow.y1, ow.x2, ow.y2], axis=1), n_rays=8)
    return _WORLD["w"]


def _front_end_by_hand(B, bufs, cols, logw, u):
    """Normaliser, CDF, search, pack and the row exchange of all ranks from one process (collectives by hand);
    returns the rows every rank received."""
    G, n = len(bufs), bufs[0].n
    for r, b in enumerate(bufs):
        _fill(b, cols, logw, r * n, (r + 1) * n)
        B.shard_local_max(b)
    max_all = torch.cat([b.max1 for b in bufs])
    for b in bufs:
        b.max_all.copy_(max_all)
        B.shard_tile_sums(b)
    tiles_all = torch.cat([b.tiles for b in bufs])
    for b in bufs:
        b.tiles_all.copy_(tiles_all)
        B.shard_stats_fold(b)
        B.shard_cdf_begin(b)
    msg_all = torch.cat([b.msg for b in bufs])
    for b in bufs:
        b.msg_all.copy_(msg_all)
        B.shard_cdf_finish(b)
    counts = np.zeros((G, G), np.int64)
    for r, b in enumerate(bufs):
        counts[r] = B.shard_search(b, u).cpu().numpy()
    R = B.SHARD_ROW
    sends = [B.shard_pack(b)[:int(counts[r].sum()) * R].clone() for r, b in enumerate(bufs)]
    recvs = []
    for r, b in enumerate(bufs):
        parts = [sends[p][int(counts[p, :r].sum()) * R: int(counts[p, :r + 1].sum()) * R] for p in range(G)]
        recv = B.shard_recv_rows(b, int(counts[:, r].sum())).clone()
        if recv.numel():
            recv.copy_(torch.cat(parts))
        recvs.append(recv)
    return recvs


@pytest.mark.parametrize("kind,G,tiles,stored", [("peaked", 4, 2, True), ("spike", 4, 3, True), ("empty_shard", 4, 2, True),
                                                 ("flat", 2, 2, True), ("peaked", 3, 5, False), ("peaked", 1, 4, True),
                                                 ("spike", 8, 1, False), ("peaked", 2, 1024, True)])   # 2^21 per rank
def test_overlapped_moves_equal_two_phase_bitwise(B, kind, G, tiles, stored):
    """gjx_shard_split -> gjx_shard_moves_main -> (rows) -> gjx_shard_moves_fixup against gjx_shard_resample_moves
    (or the un-resampled in-place moves when the ESS says so) on the same front end: particles, means, weights and
    acceptance counters bit for bit; the decision taken on the device is the host's; the own-sourced positions
    are one run."""
    import oracle as O

    dev = torch.device("cuda", 0)
    n = TILE * tiles
    N = n * G
    rng = np.random.default_rng(17 + G)
    logw = torch.from_numpy(_weights(kind, N, n, rng)).to(dev)
    ow = O.basic_world()
    obs = O.localization_observations(ow)[0][1]
    world = _any_world(B)
    pose = [rng.uniform(0.5, ow.width - 0.5, N), rng.uniform(0.5, ow.height - 0.5, N), rng.uniform(-3.1, 3.1, N)]
    cols = [torch.from_numpy(c.astype(np.float32)).to(dev) for c in pose + [c + rng.normal(size=N) * 0.2 for c in pose]]
    K, u = 3, 0.3711
    mk = B.loc_move_keys(np.array([3, 9], np.uint32), K, "mh")
    out = []
    for form in ("two-phase", "overlapped"):
        bufs = [B.new_shard_buffers(n, N, G, r, dev, n_rays=8 if stored else 0) for r in range(G)]
        recvs = _front_end_by_hand(B, bufs, cols, logw, u)
        res = []
        for r, (b, recv) in enumerate(zip(bufs, recvs)):
            if stored:
                b.ray_scores.copy_(B.loc_ray_scores(world, obs, *b.cols[:3]))
            acc = torch.full((K,), -1, dtype=torch.int64, device=dev)
            go = float(b.gstats[2]) < N // 2
            assert go == (kind != "flat")
            if form == "two-phase":
                if go:
                    B.shard_resample_moves(world, b, recv, "mh", mk, False, obs, accept_counts=acc)
                else:
                    B.loc_smc_resample_moves(world, b, False, u, "mh", mk, False, obs, offset=b.offset, accept_counts=acc)
            else:
                split = B.shard_split(b, K, acc)
                B.shard_moves_main(world, b, mk, False, obs, accept_counts=acc)
                p0p, p1, dgo, runs = split.cpu().tolist()
                assert bool(dgo) == go and runs == 0
                if go:
                    p0, p1 = (p0p - 1, p1) if p0p > 0 else (n, n)
                    n_fix = p0 + n - p1
                    assert (n_fix == 0) == (recv.numel() == 0)
                    side = torch.cuda.Stream(device=dev)            # as the driver does: another stream
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        B.shard_moves_fixup(world, b, recv, p0, p1, mk, False, obs, accept_counts=acc)
                    torch.cuda.current_stream().wait_stream(side)
            assert int(b.err.item()) == 0
            res.append([c.clone() for c in b.cols] + [b.logw.clone(), acc.clone()])
        out.append(res)
    for r in range(G):
        for a, c in zip(out[0][r], out[1][r]):
            assert torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a, c.view(torch.int32) if c.dtype == torch.float32 else c), (kind, G, r)
        assert int(out[0][r][-1].min()) >= 0


def test_sharded_search_error_paths(B):
    dev = torch.device("cuda", 0)
    b = B.new_shard_buffers(TILE, 2 * TILE, 2, 0, dev)
    b.cand_cap = TILE                                               # too small for n_global positions
    with pytest.raises(B.GjxError, match="n_global"):
        B.shard_search(b, 0.5)
    with pytest.raises(B.GjxError, match="multiple of 2048"):
        B.new_shard_buffers(1000, 2000, 2, 0, dev)
    with pytest.raises(B.GjxError, match="rows announced"):
        B.shard_recv_rows(b, TILE + 1)


# ------------------------------------------------------------------ the driver on gloo ranks sharing the GPU
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, n_global, kern, K, T, out_dir, overlap=True):
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
    smc = ShardedLocalizationSMC(gw, obs[:T], n_global, k, kernel=kern, n_moves=K,
                                 shard=ShardInfo(rank, world, n_global), device=dev)
    smc.force_sharded = True                                  # also with one rank: the same choreography
    smc.overlap = overlap
    st = smc.run()
    lml = smc.log_marginal_likelihood()                       # raises if a position found no source
    assert (smc.overlapped_steps > 0) == (overlap and kern == "mh" and K > 0), smc.overlapped_steps
    torch.cuda.synchronize()
    tag = f"{world}"
    np.save(os.path.join(out_dir, f"cols_{tag}_{rank}.npy"),
            np.stack([c.cpu().numpy() for c in (st.x, st.y, st.th, st.mx, st.my, st.mth, st.logw)]))
    np.save(os.path.join(out_dir, f"meta_{tag}_{rank}.npy"),
            np.array([lml] + st.ess + [float(r) for r in st.resampled] + [float(max(smc.remote_rows + [0]))] +
                     [float(smc.comm.bytes_in["all_to_all"])]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kern,K,worlds,overlap", [("mh", 3, (2, 4, 8), True), ("mh", 3, (2, 4), False),
                                                   ("hmc", 1, (2,), True), ("none", 0, (4,), True)])
def test_sharded_driver_on_shared_gpu_equals_one_device_bitwise(tmp_path, kern, K, worlds, overlap):
    """The whole driver as 2, 4 and 8 gloo ranks sharing the GPU against the single-device fused step (the
    product's G = 1 path, no sharding code at all): every particle, every log-weight, the ESS trace and the
    log-ML have the same bits; a rank never receives more than 28 B x n_local per resampling.  mh steps take the
    overlapped form (own-sourced moves before the exchange) unless overlap is off."""
    import torch.multiprocessing as mp

    import oracle as O
    from genjax_amd import backend as Bk
    from genjax_amd.parallel import ShardedLocalizationSMC, ShardInfo

    n_global, T = 16384, 6
    ow = O.basic_world()
    obs, k = O.localization_observations(ow)
    gw = Bk.make_loc_world(ow.width, ow.height, np.stack([ow.x1, ow.y1, ow.x2, ow.y2], axis=1), n_rays=8)
    one = ShardedLocalizationSMC(gw, obs[:T], n_global, k, kernel=kern, n_moves=K, shard=ShardInfo(0, 1, n_global),
                                 device=torch.device("cuda", 0))
    st = one.run()
    one_cols = np.stack([c.cpu().numpy() for c in (st.x, st.y, st.th, st.mx, st.my, st.mth, st.logw)])
    one_meta = np.array([one.log_marginal_likelihood()] + st.ess + [float(r) for r in st.resampled])
    assert sum(st.resampled) >= 2
    for world in (1,) + tuple(worlds):
        mp.spawn(_rank_main, args=(world, _free_port(), n_global, kern, K, T, str(tmp_path), overlap), nprocs=world,
                 join=True)
        cols = np.concatenate([np.load(tmp_path / f"cols_{world}_{r}.npy") for r in range(world)], axis=1)
        metas = [np.load(tmp_path / f"meta_{world}_{r}.npy") for r in range(world)]
        for m in metas:
            assert np.array_equal(m[:-2], one_meta), (world, m[:-2], one_meta)       # log-ML, ESS, decisions: bits
            assert m[-2] <= n_global // world                                       # rows received per resampling
        assert np.array_equal(cols.view(np.uint32), one_cols.view(np.uint32)), world
        if world > 1:
            assert sum(m[-1] for m in metas) > 0                                    # rows really crossed ranks


@pytest.mark.parametrize("n_global,K,R", [(16384, 3, 8), (1 << 18, 2, 8), (16384, 2, 32), (16384, 2, 11)])
def test_one_call_front_end_equals_single_device_bitwise(n_global, K, R):
    """The sharded mh step whose front end is ONE C call with the collectives issued from the C side
    (gjx_shard_front; here the one-rank identity communicator) against the single-device fused step: every
    particle, log-weight, the ESS trace, the decisions and the log-ML have the same bits; and against the
    call-by-call form of the same sharded step (c_front off)."""
    import oracle as O
    from genjax_amd import backend as Bk
    from genjax_amd.parallel import ShardedLocalizationSMC, ShardInfo

    T = 6
    ow = O.basic_world()
    obs, k = O.localization_observations(ow, R=R)   # 32 / 11 rays: the 16-slot instantiations of extend and of the moves
    gw = Bk.make_loc_world(ow.width, ow.height, np.stack([ow.x1, ow.y1, ow.x2, ow.y2], axis=1), n_rays=R)
    dev = torch.device("cuda", 0)

    def run(force, c_front):
        smc = ShardedLocalizationSMC(gw, obs[:T], n_global, k, kernel="mh", n_moves=K, shard=ShardInfo(0, 1, n_global),
                                     device=dev)
        smc.force_sharded, smc.c_front = force, c_front
        st = smc.run()
        lml = smc.log_marginal_likelihood()
        cols = np.stack([c.cpu().numpy() for c in (st.x, st.y, st.th, st.mx, st.my, st.mth, st.logw)])
        return cols, np.array([lml] + st.ess + [float(r) for r in st.resampled]), smc

    one_cols, one_meta, _ = run(False, False)
    assert sum(one_meta[1 + T:]) >= 2                                   # it did resample
    for c_front in (True, False):
        cols, meta, smc = run(True, c_front)
        assert (smc.front_steps == T) == c_front and smc.overlapped_steps == T
        assert np.array_equal(meta, one_meta), (c_front, meta, one_meta)
        assert np.array_equal(cols.view(np.uint32), one_cols.view(np.uint32)), c_front
    # a second run on the same buffers' cell invariant: log_marginal_likelihood() leaves the shard-maximum cell usable
    cols2, meta2, _ = run(True, True)
    assert np.array_equal(cols2.view(np.uint32), one_cols.view(np.uint32))


def test_bench_runs_the_sharded_step_over_nccl_with_one_rank():
    """bench.py --force-sharded: the sharded choreography through a ONE-rank NCCL (RCCL) process group --
    the only RCCL exercise a one-GPU box allows (all_gather_into_tensor runs, the row
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
    assert line["config"]["rows_received_per_resampling"] == 0          # one rank: nothing crosses
    assert line["config"]["front_end_calls_from_c"] > 0                 # ... through the C side's RCCL communicator
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(line["roofline"])
