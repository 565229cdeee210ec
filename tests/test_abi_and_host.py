"""CPU tests: the C-ABI library loads and exports every symbol include/gjx.h declares, the
host-side key arithmetic matches the oracle, and the host logic (selections, Const, key schedule,
data generator, argument checks) behaves like the reference.  No GPU compute is called."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    from genjax_amd import _native

    return _native, ctypes.CDLL(_native.build())


def test_header_symbols_are_exported():
    native, lib = _lib()
    header = open(os.path.join(ROOT, "include", "gjx.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = sorted(set(re.findall(r"\b(gjx_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 35
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in gjx.h but not exported"
    for name in native.EXPORTED_SYMBOLS:
        assert name in declared, f"{name} bound in _native.py but not declared in gjx.h"


def test_no_torch_types_in_abi():
    header = open(os.path.join(ROOT, "include", "gjx.h")).read()
    assert "torch" not in header.lower().replace("torch tensor", "") and "at::" not in header


def test_host_key_ops_match_oracle(oracle):
    import genjax_amd as G

    for seed in (0, 1, 42, 12345, (3 << 32) + 9):
        k = G.key(seed)
        assert np.array_equal(k, oracle.key(seed))
        assert np.array_equal(G.split(k, 4), oracle.split(k, 4))
        assert np.array_equal(G.fold_in(k, 17), oracle.fold_in(k, 17))
    from genjax_amd.backend import KeySeq

    a, b = KeySeq(oracle.key(42)), oracle.Seed(oracle.key(42))
    for _ in range(5):
        assert np.array_equal(a.next(), b.next())


def test_host_uniform_matches_oracle(oracle):
    from genjax_amd.inference.smc import _host_uniform01

    for s in range(20):
        k = oracle.fold_in(oracle.key(7), s)
        assert np.float32(_host_uniform01(k)) == oracle.uniform(k, 1)[0]


def test_data_generator_matches_oracle_bitwise(oracle):
    """generate_ground_truth_data (data.py:168-216) is host-side and bit-identical to the oracle's."""
    import genjax_amd as G
    from genjax_amd.families.localization import create_basic_multi_room_world, generate_ground_truth_data

    world = create_basic_multi_room_world()
    key = G.key(42)
    key, sub = G.split(key, 2)
    poses, controls, obs = generate_ground_truth_data(world, sub)
    ref, smc_key = oracle.localization_observations(oracle.basic_world())
    assert len(poses) == 17 and len(controls) == 16
    assert np.array_equal(np.stack(obs), ref)
    assert np.array_equal(G.split(key, 2)[1], smc_key)


def test_world_validation():
    from genjax_amd.families.localization import World, create_multi_room_world

    with pytest.raises(ValueError, match="at least 1 internal wall"):
        World(12.0, 10.0, np.zeros(0), np.zeros(0), np.zeros(0), np.zeros(0), 0)
    with pytest.raises(ValueError, match="expected"):
        World(12.0, 10.0, np.zeros(2), np.zeros(3), np.zeros(2), np.zeros(2), 2)
    with pytest.raises(ValueError, match="Unknown world_type"):
        create_multi_room_world("nope")
    assert create_multi_room_world("complex").num_walls == 15


def test_selection_algebra():
    """sel / | / ^ / ~ / tuple paths / dict (reference core.py:781-1018)."""
    from genjax_amd import sel

    s = sel("x") | sel("y")
    assert s.match("x")[0] and s.match("y")[0] and not s.match("theta")[0]
    assert () in s.match("x")[1] and () not in s.match("theta")[1]
    assert (~s).match("theta")[0] and not (~s).match("x")[0]
    both = (sel("x") | sel("y")) ^ (sel("y") | sel("z"))
    assert both.match("y")[0] and not both.match("x")[0]
    hit, rest = sel(("curve", "a")).match("curve")
    assert hit and rest.match("a")[0] and not rest.match("b")[0]
    assert sel(()).match("anything")[0] and not sel().match("x")[0] and not sel(None).match("x")[0]
    d = sel({"outer": sel("inner")})
    hit, rest = d.match("outer")
    assert hit and rest.match("inner")[0]


def test_const():
    from genjax_amd import Const, const

    c = const(8)
    assert isinstance(c, Const) and c.value == 8 and (c + 1).value == 9 and (c // 2).value == 4
    assert const(3) == const(3)


def test_move_key_schedule_matches_appendix_a(oracle):
    """mh: S x3 (x', y', theta') then u; hmc: S x3 momenta (theta, x, y), one S for the leapfrog scan,
    then u -- the schedule the oracle's driver follows."""
    from genjax_amd import sel
    from genjax_amd.families.base import MoveSpec
    from genjax_amd.families.localization import localization_model as m

    k = oracle.fold_in(oracle.key(5), 3)
    all_sel = sel("x") | sel("y") | sel("theta")
    s = oracle.Seed(k)
    ref = [s.next() for _ in range(4)]
    assert np.array_equal(m.move_keys(MoveSpec("mh", all_sel), k), np.stack(ref))
    s = oracle.Seed(k)
    ref = [s.next() for _ in range(3)]
    s.next()
    ref.append(s.next())
    assert np.array_equal(m.move_keys(MoveSpec("hmc", all_sel, 0.01, 10), k), np.stack(ref))
    # partial selection draws only the selected sites
    s = oracle.Seed(k)
    kx, ku = s.next(), s.next()
    got = m.move_keys(MoveSpec("mh", sel("x")), k)
    assert np.array_equal(got[0], kx) and np.array_equal(got[3], ku) and not got[1].any()


def test_native_move_key_schedule_matches_host(oracle):
    """gjx_loc_move_keys (host function of the C ABI; fold_in(scan_key, m) then the per-kind splits,
    smc.py:756-765 + mcmc.py:267-309 / 434-578) against the Python schedule, all selections."""
    from genjax_amd import backend as B, sel
    from genjax_amd.families.base import MoveSpec
    from genjax_amd.families.localization import localization_model as m

    sk = oracle.fold_in(oracle.key(11), 7)
    sites = {1: sel("x"), 2: sel("y"), 4: sel("theta")}
    for mask in range(1, 8):
        s = None
        for bit, one in sites.items():
            if mask & bit:
                s = one if s is None else (s | one)
        for kind in ("mh", "hmc"):
            spec = MoveSpec(kind, s, 0.01, 10)
            ref = np.stack([m.move_keys(spec, B.fold_in(sk, i)) for i in range(6)])
            got = B.loc_move_keys(sk, 6, kind, sel_mask=mask)
            assert np.array_equal(got, ref), (mask, kind)
    assert B.loc_move_keys(sk, 0, "mh").shape == (0, 4, 2)


def test_argument_errors_are_reported_not_thrown():
    native, _ = _lib()
    L = native.lib()
    rc = L.gjx_ctx_create(0, -5, ctypes.byref(ctypes.c_void_p()))
    assert rc == -1 and b"n_max" in L.gjx_last_error()
    key = (ctypes.c_uint32 * 2)(0, 1)
    assert L.gjx_random_normal_f32(key, 0, 10, 0.0, 1.0, None, None) == -1
    assert L.gjx_random_normal_f32(key, 0, 0, 0.0, 1.0, None, None) == 0   # empty input is fine
    assert L.gjx_resample_multinomial(key, ctypes.c_void_p(8), 1 << 20, 1 << 20, ctypes.c_void_p(8), None) == -4
    w = native.LocWorld()
    w.n_walls, w.n_rays, w.sensor_sigma = 0, 8, 0.3
    out = (ctypes.c_float * 8)()
    z = (ctypes.c_float * 1)(0.0)
    assert L.gjx_loc_lidar_host(ctypes.byref(w), z, z, z, 1, out) == -4


def test_step_and_shard_entry_points_validate_before_touching_the_gpu():
    """The fused-step and sharded-step entry points reject bad arguments with a status code and a
    message (no GPU needed: validation comes first)."""
    native, _ = _lib()
    L = native.lib()
    p8 = ctypes.c_void_p(8)                       # a non-null dummy pointer, never dereferenced
    cols = (ctypes.c_void_p * 6)(*([8] * 6))
    key = (ctypes.c_uint32 * 2)(1, 2)
    out = (ctypes.c_uint32 * 8)()
    assert L.gjx_loc_move_keys(key, 1, 3, 7, out) == -1 and b"bad argument" in L.gjx_last_error()
    assert L.gjx_loc_move_keys(key, 1, 1, 0, out) == -1
    assert L.gjx_shard_stats_fold(None, 2, p8, 2, p8, None) == -1
    assert L.gjx_shard_stats_fold(p8, 0, p8, 2, p8, None) == -1
    assert L.gjx_shard_stats_fold(p8, 4, p8, 65, p8, None) == -1                       # more ranks than GJX_MAX_PIECES
    assert L.gjx_shard_tile_sums(p8, 1000, p8, 2, p8, None) == -1 and b"multiple of 2048" in L.gjx_last_error()
    assert L.gjx_shard_cdf_begin(p8, p8, 1000, 0, p8, p8, 1, p8, None) == -1 and b"multiple of 2048" in L.gjx_last_error()
    assert L.gjx_shard_cdf_begin(p8, p8, 2048, 0, None, None, 1, p8, None) == -1 and b"probe" in L.gjx_last_error()
    assert L.gjx_shard_cdf_finish(p8, p8, p8, 2048, p8, 2, 2, p8, p8, 2, p8, p8, p8, 1, p8, None) == -1   # rank >= n_ranks
    assert L.gjx_shard_cdf_finish(p8, p8, p8, 1 << 30, p8, 4, 0, p8, p8, 2, p8, p8, p8, 2, p8, None) == -1   # n_global >= 2^31
    # 2^lc - 1 probes must fit the n_ranks x p_cap slots
    assert L.gjx_shard_cdf_finish(p8, p8, p8, 2048 * 64, p8, 2, 0, p8, p8, 1, p8, p8, p8, 3, p8, None) == -1
    assert b"probe" in L.gjx_last_error()
    # the candidate buffers must hold n_global positions; the coarse table must leave <= 2048-entry intervals
    assert L.gjx_shard_search(p8, 2, p8, 2, 0, 2048, 0.5, 2048, p8, p8, p8, p8, None) == -4
    assert b"n_global" in L.gjx_last_error()
    assert L.gjx_shard_search(p8, 0, p8, 2, 0, 2048, 0.5, 4096, p8, p8, p8, p8, None) == -1
    assert L.gjx_shard_pack(p8, p8, p8, p8, 2, 0, 2048, None, p8, None) == -1
    w8 = ctypes.cast(p8, ctypes.POINTER(native.LocWorld))
    f8 = ctypes.cast(p8, ctypes.POINTER(ctypes.c_float))
    assert L.gjx_shard_resample_moves(None, w8, p8, p8, p8, 0, 0, 2048, 2048, p8, p8, 0, None, 0, 7, 0.0, 1, 0, f8, cols, cols,
                                      p8, None, 0, None, None, None, None) == -1
    assert L.gjx_event_record(None, None) == -1
    assert L.gjx_event_elapsed_ms(None, None, None) == -1


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under genjax_amd/ may reference it."""
    pkg = os.path.join(ROOT, "genjax_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "gjx_oracle" not in text and "orc_" not in text, f


def test_missing_gpu_fails_loudly():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from genjax_amd import GjxError, backend

    with pytest.raises(GjxError, match="no CPU fallback"):
        backend.random_normal(backend.key(0), 4)


def test_rhat_and_ess_formulas():
    """compute_rhat / compute_ess (mcmc.py:91-247) against their numpy definitions (host tensors)."""
    import numpy as np
    import torch

    from genjax_amd.inference.mcmc import compute_ess, compute_rhat

    rng = np.random.default_rng(0)
    x = rng.normal(size=(4, 200)).cumsum(axis=1) * 0.05 + rng.normal(size=(4, 1))
    n = x.shape[1]
    b = n * x.mean(axis=1).var(ddof=1)
    w = x.var(axis=1, ddof=1).mean()
    ref = np.sqrt(((n - 1) * w + b) / n / w)
    assert abs(float(compute_rhat(torch.from_numpy(x))) - ref) < 1e-5
    flat = x.reshape(-1)
    rho = np.clip(np.corrcoef(flat[:-1], flat[1:])[0, 1], 0.0, 0.99)
    assert abs(float(compute_ess(torch.from_numpy(x), "bulk")) - 4 * n / (1 + 2 * rho)) < 1e-2
    q = np.quantile(x, [0.05, 0.95], axis=1)
    d = q[1] - q[0]
    rho_t = np.clip(np.corrcoef(d[:-1], d[1:])[0, 1], 0.0, 0.99)
    assert abs(float(compute_ess(torch.from_numpy(x), "tail")) - n / (1 + 2 * rho_t)) < 1e-2
    assert np.isnan(float(compute_rhat(torch.from_numpy(x[:1]))))


def test_localization_export_format_round_trip(tmp_path):
    """On-disk export (examples/localization/export.py:15-137): file names, headers and number
    formatting as the reference writes them; columnar history in, columnar history back."""
    import torch
    from genjax_amd.families import localization_export as E
    from genjax_amd.families.localization import Pose

    d = str(tmp_path / "run")
    f32 = lambda *v: np.asarray(v, np.float32)  # noqa: E731
    truth = [Pose(1.5, 1.5, 0.0), Pose(np.float32(2.1), np.float32(1.25), np.float32(0.1))]
    obs = [f32(1.0, 2.5, 0.75), f32(0.5, 0.25, 3.0)]
    E.save_ground_truth_data(d, truth, obs)
    assert open(os.path.join(d, "ground_truth_poses.csv")).read().splitlines() == [
        "timestep,x,y,theta", "0,1.5,1.5,0.0", "1,2.0999999046325684,1.25,0.10000000149011612"]
    assert open(os.path.join(d, "ground_truth_observations.csv")).read().splitlines() == [
        "timestep,ray_0,ray_1,ray_2", "0,1.0,2.5,0.75", "1,0.5,0.25,3.0"]
    poses, observations = E.load_ground_truth_data(d)
    assert poses[1].y == 1.25 and observations[1] == [0.5, 0.25, 3.0]

    hist = [Pose(torch.tensor([0.5, 1.0, 1.5]), torch.tensor([2.0, 2.5, 3.0]), torch.tensor([0.0, 0.25, -0.5])),
            Pose(f32(0.75, 1.25, 1.75), f32(2.0, 2.5, 3.0), f32(0.1, 0.2, 0.3))]
    wts = [torch.tensor([0.25, 0.5, 0.25]), f32(0.125, 0.375, 0.5)]
    diag = np.log(np.stack([w.numpy() if hasattr(w, "numpy") else w for w in wts])).astype(np.float32)
    res = {"particle_history": hist, "weight_history": wts, "diagnostic_weights": diag, "timing_stats": (0.5, 0.125)}
    E.save_benchmark_results(d, {"smc_basic": res}, {"n_particles": 3, "n_rays": 3})
    assert open(os.path.join(d, "smc_basic", "particles", "timestep_000.csv")).read().splitlines() == [
        "particle_id,x,y,theta,weight", "0,0.5,2.0,0.0,0.25", "1,1.0,2.5,0.25,0.5", "2,1.5,3.0,-0.5,0.25"]
    assert open(os.path.join(d, "benchmark_summary.csv")).read().splitlines() == [
        "method,mean_time_sec,std_time_sec,n_particles,n_timesteps", "smc_basic,0.5,0.125,3,2"]
    assert open(os.path.join(d, "smc_basic", "timing.csv")).read().splitlines() == [
        "mean_time_sec,std_time_sec", "0.5,0.125"]
    assert E.load_experiment_metadata(d) == {"n_particles": 3, "n_rays": 3}
    back = E.load_benchmark_results(d)["smc_basic"]
    assert back["timing_stats"] == (0.5, 0.125)
    assert np.array_equal(back["particle_history"][1].theta, f32(0.1, 0.2, 0.3))
    assert np.array_equal(back["weight_history"][0], f32(0.25, 0.5, 0.25))
    assert np.array_equal(back["diagnostic_weights"], diag)
    # a list of scalar poses (the reference's in-memory shape) is accepted too
    res2 = dict(res, particle_history=[[Pose(0.5, 2.0, 0.0), Pose(1.0, 2.5, 0.25), Pose(1.5, 3.0, -0.5)]],
                weight_history=[f32(0.25, 0.5, 0.25)], diagnostic_weights=None)
    E.save_smc_results(d, "listed", res2)
    assert (open(os.path.join(d, "listed", "particles", "timestep_000.csv")).read() ==
            open(os.path.join(d, "smc_basic", "particles", "timestep_000.csv")).read())
    assert E.load_smc_results(d, "listed")["diagnostic_weights"] is None


def test_export_layout_is_the_reference_writers_byte_for_byte_and_reads_a_hand_written_run(tmp_path):
    """The strings below are what examples/localization/export.py writes (file names :17,28,36,64,75,92,117; header
    rows :31,42,46,67,78,97,120-122; csv.writer's CRLF row ends; json.dump(indent=2)), typed in by hand from that
    file -- not produced by this package.  (1) The package's writer must produce exactly these bytes.  (2) A run
    directory written by hand in that layout (as the reference would leave it) must load."""
    import torch
    from genjax_amd.families import localization_export as E
    from genjax_amd.families.localization import Pose

    want = {
        "experiment_metadata.json": '{\n  "n_particles": 2,\n  "n_rays": 2,\n  "world": "basic"\n}',
        "ground_truth_poses.csv": "timestep,x,y,theta\r\n0,1.5,1.5,0.0\r\n1,2.0,1.25,0.5\r\n",
        "ground_truth_observations.csv": "timestep,ray_0,ray_1\r\n0,1.0,2.5\r\n1,0.5,0.25\r\n",
        "benchmark_summary.csv": "method,mean_time_sec,std_time_sec,n_particles,n_timesteps\r\nsmc_hmc,0.25,0.5,2,2\r\n",
        "smc_hmc/timing.csv": "mean_time_sec,std_time_sec\r\n0.25,0.5\r\n",
        "smc_hmc/particles/timestep_000.csv": "particle_id,x,y,theta,weight\r\n0,0.5,2.0,0.0,0.25\r\n1,1.0,2.5,0.25,0.75\r\n",
        "smc_hmc/particles/timestep_001.csv": "particle_id,x,y,theta,weight\r\n0,0.75,2.0,-0.5,0.5\r\n1,1.25,3.0,1.0,0.5\r\n",
        "smc_hmc/diagnostic_weights.csv": "timestep,particle_0,particle_1\r\n0,-1.5,-0.25\r\n1,-0.75,-0.5\r\n",
    }
    f32 = lambda *v: np.asarray(v, np.float32)  # noqa: E731
    # (1) the writer
    d = str(tmp_path / "written")
    E.save_ground_truth_data(d, [Pose(1.5, 1.5, 0.0), Pose(2.0, 1.25, 0.5)], [f32(1.0, 2.5), f32(0.5, 0.25)])
    res = {"particle_history": [Pose(torch.tensor([0.5, 1.0]), torch.tensor([2.0, 2.5]), torch.tensor([0.0, 0.25])),
                                Pose(f32(0.75, 1.25), f32(2.0, 3.0), f32(-0.5, 1.0))],
           "weight_history": [f32(0.25, 0.75), f32(0.5, 0.5)],
           "diagnostic_weights": f32(-1.5, -0.25, -0.75, -0.5).reshape(2, 2), "timing_stats": (0.25, 0.5)}
    E.save_benchmark_results(d, {"smc_hmc": res}, {"n_particles": 2, "n_rays": 2, "world": "basic"})
    for rel, text in want.items():
        with open(os.path.join(d, rel), newline="") as f:
            assert f.read() == text, rel
    assert sorted(os.listdir(d)) == ["benchmark_summary.csv", "experiment_metadata.json",
                                     "ground_truth_observations.csv", "ground_truth_poses.csv", "smc_hmc"]
    # the single-observation form (export.py:45-48)
    E.save_ground_truth_data(str(tmp_path / "single"), [Pose(0.0, 0.0, 0.0)], [f32(1.25)])
    with open(os.path.join(str(tmp_path / "single"), "ground_truth_observations.csv"), newline="") as f:
        assert f.read() == "timestep,observation\r\n0,1.25\r\n"
    assert E.load_ground_truth_data(str(tmp_path / "single"))[1] == [1.25]
    # (2) a directory typed in by hand
    h = str(tmp_path / "by_hand")
    for rel, text in want.items():
        os.makedirs(os.path.dirname(os.path.join(h, rel)), exist_ok=True)
        with open(os.path.join(h, rel), "w", newline="") as f:
            f.write(text)
    assert E.load_experiment_metadata(h) == {"n_particles": 2, "n_rays": 2, "world": "basic"}
    poses, observations = E.load_ground_truth_data(h)
    assert [(q.x, q.y, q.theta) for q in poses] == [(1.5, 1.5, 0.0), (2.0, 1.25, 0.5)]
    assert observations == [[1.0, 2.5], [0.5, 0.25]]
    back = E.load_benchmark_results(h)
    assert list(back) == ["smc_hmc"] and back["smc_hmc"]["timing_stats"] == (0.25, 0.5)
    hist, wts = back["smc_hmc"]["particle_history"], back["smc_hmc"]["weight_history"]
    assert len(hist) == 2 and np.array_equal(hist[1].theta, f32(-0.5, 1.0)) and np.array_equal(hist[0].y, f32(2.0, 2.5))
    assert np.array_equal(wts[0], f32(0.25, 0.75))
    assert np.array_equal(back["smc_hmc"]["diagnostic_weights"], f32(-1.5, -0.25, -0.75, -0.5).reshape(2, 2))


def test_move_key_block_equals_the_per_move_interpreter_schedule():
    """The K move keys of a rejuvenation scan from ONE host call (gjx_loc_move_keys, what the API path and the step
    driver use) against the per-move restatement of the interpreter -- a fresh KeySeq on fold_in(scan_key, m), one
    split per selected site in the kernel's order, one for hmc's leapfrog scan, one for the accept uniform -- for
    both kernels and every non-empty selection of the three pose sites."""
    from genjax_amd import backend as B
    from genjax_amd.core import sel
    from genjax_amd.families.base import MoveSpec
    from genjax_amd.families.localization import localization_model

    scan_key = np.array([0x1234ABCD, 0x0F0F1E1E], np.uint32)
    sites = ("x", "y", "theta")
    for kind in ("mh", "hmc"):
        for mask in range(1, 8):
            chosen = [a for b, a in enumerate(sites) if mask & (1 << b)]
            selection = sel(chosen[0])
            for a in chosen[1:]:
                selection = selection | sel(a)
            spec = MoveSpec(kind, selection, 0.01, 10)
            assert localization_model.sel_mask(selection) == mask
            block = localization_model.move_keys_block(spec, scan_key, 5)
            per_move = np.stack([localization_model.move_keys(spec, B.fold_in(scan_key, m)) for m in range(5)])
            assert block.shape == (5, 4, 2) and np.array_equal(block, per_move), (kind, mask)
