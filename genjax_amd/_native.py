"""ctypes binding of libgjx_hip.so (include/gjx.h).

The HIP library is the only compute path of this package: there is no CPU fallback.
If the shared object is missing or a call fails, a GjxError is raised.
"""
from __future__ import annotations

import ctypes as C
import glob
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_SO = os.environ.get("GJX_LIB_PATH") or os.path.join(_HERE, "_lib", "libgjx_hip.so")  # override: kernel-variant tuning only

MAX_RAYS, MAX_WALLS, MAX_MOVES, MAX_COLS, MAX_POINTS = 32, 32, 64, 16, 64
SCAN_TILE = 2048


class GjxError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """Compile every .hip source for gfx950 into genjax_amd/_lib/libgjx_hip.so (in-tree).
    One object per source (rebuilt only when it or a header changed, compiled in parallel),
    then one link."""
    from concurrent.futures import ThreadPoolExecutor

    srcs = sorted(glob.glob(os.path.join(_CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(_CSRC, "*.h")) + [os.path.join(_HERE, "..", "include", "gjx.h")]
    hdr_m = max(os.path.getmtime(h) for h in hdrs)
    objdir = os.path.join(os.path.dirname(_SO), "obj")
    os.makedirs(objdir, exist_ok=True)
    # -ffp-contract=on: a * b + c fuses only where the SOURCE writes it as one expression, so a device function
    # rounds the same way in every kernel it is inlined into (the fused move kernels == their single-move
    # entry points bit for bit; hipcc's default, fast-honor-pragmas, contracts across statements depending on
    # the surrounding code).  Measured: the move launch 2.75 ms against 2.81 ms with the default.
    # -instcombine-max-copied-from-constant-users: the world (LocK) and the move keys are by-value kernel arguments;
    # LLVM reads such a struct straight from the kernarg segment only if the by-value copy has at most 300 users
    # (the default of that cap) -- above it the copy stays, i.e. every lane writes ~2.5 KB of world tables to scratch
    # at kernel entry and indexes them there.  The 16- and 32-ray move kernels were over the cap (scratch 2.0-4.1 KB
    # per lane against 0.2-0.5 KB with the cap lifted).
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-ffp-contract=on",
             "-mllvm", "-instcombine-max-copied-from-constant-users=1000000"] + \
        os.environ.get("GJX_EXTRA_HIPCC_FLAGS", "").split()
    stamp = os.path.join(objdir, "flags.txt")
    if not os.path.exists(stamp) or open(stamp).read() != " ".join(flags):
        force = True
    jobs, objs = [], []
    for src in srcs:
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_m):
            jobs.append(["hipcc"] + flags + ["-c", src, "-o", obj])
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), 6)) as ex:
            list(ex.map(subprocess.check_call, jobs))
        with open(stamp, "w") as f:
            f.write(" ".join(flags))
    if jobs or not os.path.exists(_SO) or any(os.path.getmtime(_SO) < os.path.getmtime(o) for o in objs):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", _SO] + objs)
    return _SO


class LocWorld(C.Structure):
    """gjx_loc_world (include/gjx.h)."""

    _fields_ = [
        ("width", C.c_float),
        ("height", C.c_float),
        ("n_walls", C.c_int32),
        ("n_rays", C.c_int32),
        ("walls", (C.c_float * 4) * MAX_WALLS),
        ("max_range", C.c_float),
        ("sensor_sigma", C.c_float),
        ("init_mean", C.c_float * 3),
        ("init_sigma", C.c_float * 3),
        ("drift_sigma", C.c_float * 3),
        ("clip_margin", C.c_float),
    ]


class LgssmParams(C.Structure):
    """gjx_lgssm_params (include/gjx.h)."""

    _fields_ = [
        ("d_state", C.c_int32),
        ("d_obs", C.c_int32),
        ("initial_mean", C.c_float * 4),
        ("initial_cov", C.c_float * 16),
        ("A", C.c_float * 16),
        ("Q", C.c_float * 16),
        ("C", C.c_float * 16),
        ("R", C.c_float * 16),
    ]


class OutlierModel(C.Structure):
    """gjx_outlier_model (include/gjx.h)."""

    _fields_ = [
        ("n_points", C.c_int32),
        ("xs", C.c_float * MAX_POINTS),
        ("ys", C.c_float * MAX_POINTS),
        ("alpha", C.c_float),
        ("beta", C.c_float),
        ("inlier_sigma", C.c_float),
        ("out_lo", C.c_float),
        ("out_hi", C.c_float),
    ]


class ShardBufs(C.Structure):
    """gjx_shard_bufs (include/gjx.h)."""

    _fields_ = [
        ("n", C.c_int64), ("n_global", C.c_int64),
        ("rank", C.c_int32), ("n_ranks", C.c_int32), ("lc", C.c_int32), ("p_cap", C.c_int32),
        ("cols", C.c_void_p * 6),
        ("logw", C.c_void_p),
        ("ray_scores", C.c_void_p),
        ("scores_stride", C.c_int64),
        ("max1", C.c_void_p), ("max_all", C.c_void_p),
        ("tiles", C.c_void_p), ("tiles_all", C.c_void_p),
        ("gstats", C.c_void_p),
        ("msg", C.c_void_p), ("msg_all", C.c_void_p),
        ("msg_len", C.c_int64),
        ("probe_nodes", C.c_void_p), ("probe_mids", C.c_void_p),
        ("all_totals", C.c_void_p), ("scanned", C.c_void_p), ("cdf_ext", C.c_void_p), ("table", C.c_void_p),
        ("range", C.c_void_p),
        ("cand_idx", C.c_void_p),
        ("cand_cap", C.c_int64),
        ("tile_counts", C.c_void_p), ("counts", C.c_void_p), ("split", C.c_void_p),
    ]


_lib = None

_vp, _i32, _i64, _u64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float
_u32p = C.POINTER(C.c_uint32)
_f32p = C.POINTER(C.c_float)

_SIGNATURES = {
    "gjx_version": ([], _i32),
    "gjx_last_error": ([], C.c_char_p),
    "gjx_ctx_create": ([_i32, _i64, C.POINTER(_vp)], _i32),
    "gjx_ctx_destroy": ([_vp], _i32),
    "gjx_key": ([_u64, _u32p], None),
    "gjx_threefry2x32": ([_u32p, _u32p, _u32p], None),
    "gjx_split": ([_u32p, C.c_uint32, _u32p], None),
    "gjx_fold_in": ([_u32p, C.c_uint32, _u32p], None),
    "gjx_random_bits_u32": ([_u32p, _u64, _i64, _vp, _vp], _i32),
    "gjx_random_uniform_f32": ([_u32p, _u64, _i64, _f32, _f32, _vp, _vp], _i32),
    "gjx_random_normal_f32": ([_u32p, _u64, _i64, _f32, _f32, _vp, _vp], _i32),
    "gjx_random_normal_cols_f32": ([_u32p, _u64, _i64, _vp, _i32, _vp, _i32, _vp, _vp], _i32),
    "gjx_random_uniform_cols_f32": ([_u32p, _u64, _i64, _vp, _i32, _vp, _i32, _vp, _vp], _i32),
    "gjx_random_bernoulli_f32": ([_u32p, _u64, _i64, _vp, _i32, _vp, _vp], _i32),
    "gjx_random_categorical_i32": ([_u32p, _u64, _i64, _vp, _i32, _i32, _vp, _vp], _i32),
    "gjx_random_mvnormal_f32": ([_u32p, _u64, _i64, _i32, _vp, _i32, _vp, _i32, _vp, _vp], _i32),
    "gjx_random_beta_f32": ([_u32p, _u64, _i64, _vp, _i32, _vp, _i32, _vp, _vp], _i32),
    "gjx_random_bernoulli_logits_f32": ([_u32p, _u64, _i64, _vp, _i32, _vp, _vp], _i32),
    "gjx_bernoulli_logits_logpdf_f32": ([_vp, _vp, _i32, _i64, _vp, _vp], _i32),
    "gjx_logpdf_grad_f32": ([_i32, _vp, _vp, _i32, _vp, _i32, _vp, _i32, _i64, _vp, _vp, _vp, _vp], _i32),
    "gjx_axpy_f32": ([_f32, _vp, _i32, _vp, _i32, _i64, _vp, _vp], _i32),
    "gjx_sum_diff_f32": ([_vp, _vp, _vp, _vp, _i64, _vp, _vp], _i32),
    "gjx_sub_scalar_f32": ([_vp, _vp, _i64, _vp, _vp], _i32),
    "gjx_weighted_sum_f32": ([_vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp], _i32),
    "gjx_normal_logpdf_f32": ([_vp, _vp, _i32, _vp, _i32, _i64, _vp, _vp], _i32),
    "gjx_uniform_logpdf_f32": ([_vp, _vp, _i32, _vp, _i32, _i64, _vp, _vp], _i32),
    "gjx_bernoulli_logpdf_f32": ([_vp, _vp, _i32, _i64, _vp, _vp], _i32),
    "gjx_beta_logpdf_f32": ([_vp, _vp, _i32, _vp, _i32, _i64, _vp, _vp], _i32),
    "gjx_categorical_logpdf_f32": ([_vp, _vp, _i32, _i64, _i32, _vp, _vp], _i32),
    "gjx_mvnormal_logpdf_f32": ([_vp, _vp, _i32, _vp, _i32, _i64, _i32, _vp, _vp], _i32),
    "gjx_mh_accept": ([_u32p, _u64, _i64, _vp, _i32, _vp, _vp], _i32),
    "gjx_select_rows_b32": ([_vp, _vp, _vp, _i64, _i64, _vp, _vp], _i32),
    "gjx_logsumexp_ess": ([_vp, _vp, _i64, _vp, _vp], _i32),
    "gjx_max_f32": ([_vp, _vp, _i64, _vp, _vp], _i32),
    "gjx_sumexp_f32": ([_vp, _vp, _vp, _i64, _vp, _vp], _i32),
    "gjx_normalized_weights": ([_vp, _vp, _i64, _vp, _vp], _i32),
    "gjx_cumsum_f32": ([_vp, _vp, _i64, _vp, _vp], _i32),
    "gjx_cumsum_tile_totals": ([_vp, _i64, _vp, _vp], _i32),
    "gjx_cumsum_apply": ([_vp, _i64, _vp, _i64, _vp, _vp], _i32),
    "gjx_searchsorted_systematic": ([_vp, _i64, _f32, _i64, _i64, _i64, _vp, _vp], _i32),
    "gjx_bisect_levels": ([_i64], _i32),
    "gjx_bisect_plan": ([_i64, _i32, C.POINTER(C.c_int32)], _i32),
    "gjx_searchsorted_window": ([_vp, _i32, _i64, _f32, _i64, _i64, _i32, _i64, _vp, _vp], _i32),
    "gjx_searchsorted_systematic_2l": ([_vp, _i32, _vp, _i64, _i64, _i64, _f32, _i64, _i64, _i64, _vp, _vp], _i32),
    "gjx_resample_systematic": ([_vp, _vp, _i64, _f32, _vp, _vp, _vp], _i32),
    "gjx_resample_systematic_logw_static": ([_vp, _vp, _i64, _f32, _vp, _vp], _i32),
    "gjx_resample_multinomial": ([_u32p, _vp, _i64, _i64, _vp, _vp], _i32),
    "gjx_gather_columns_f32": ([_vp, _i64, _i64, C.POINTER(_vp), C.POINTER(_vp), _i32, _vp], _i32),
    "gjx_loc_lidar": ([C.POINTER(LocWorld), _vp, _vp, _vp, _i64, _vp, _vp], _i32),
    "gjx_loc_lidar_jac": ([C.POINTER(LocWorld), _vp, _vp, _vp, _i64, _vp, _vp, _vp], _i32),
    "gjx_loc_lidar_host": ([C.POINTER(LocWorld), _f32p, _f32p, _f32p, _i64, _f32p], _i32),
    "gjx_host_normal": ([_u32p], _f32),
    "gjx_loc_extend": ([_vp, C.POINTER(LocWorld), _u32p, _i32, _u64, _i64, _f32p] + [_vp] * 11 + [_vp], _i32),
    "gjx_loc_extend_locopt": ([_vp, C.POINTER(LocWorld), _u32p, _i32, _f32, _f32, _u64, _i64, _f32p] + [_vp] * 11 + [_vp], _i32),
    "gjx_loc_move_keys": ([_u32p, _i32, _i32, _i32, _u32p], _i32),
    "gjx_loc_smc_extend": ([_vp, C.POINTER(LocWorld), _u32p, _i32, _u64, _i64, _f32p] + [_vp] * 11 + [_vp, _vp, _i64, _vp], _i32),
    "gjx_loc_ray_scores": ([C.POINTER(LocWorld), _f32p, _vp, _vp, _vp, _i64, _vp, _vp], _i32),
    "gjx_loc_extend_scores": ([_vp, C.POINTER(LocWorld), _u32p, _i32, _u64, _i64, _f32p] + [_vp] * 11 + [_vp, _i64, _vp], _i32),
    "gjx_loc_smc_presearch": ([_vp, _vp, _vp, _i64, _f32, _vp], _i32),
    "gjx_loc_smc_resample_moves": ([_vp, C.POINTER(LocWorld), _i32, _f32, _vp, _i32, _u32p, _i32, _i32, _f32, _i32,
                                    _i32, _u64, _i64, _f32p, C.POINTER(_vp), C.POINTER(_vp), _vp, _vp, _i64, _vp, _vp, _vp, _vp], _i32),
    "gjx_cumsum_normalized_f32": ([_vp, _vp, _vp, _i64, _vp, _vp], _i32),
    "gjx_resample_systematic_logw": ([_vp, _vp, _vp, _i64, _f32, _vp, _vp, _vp], _i32),
    "gjx_loc_shard_extend": ([_vp, C.POINTER(LocWorld), _u32p, _i32, _u64, _i64, _f32p] + [_vp] * 11 + [_vp, _vp, _i64, _vp], _i32),
    "gjx_shard_tile_sums": ([_vp, _i64, _vp, _i32, _vp, _vp], _i32),
    "gjx_shard_stats_fold": ([_vp, _i64, _vp, _i32, _vp, _vp], _i32),
    "gjx_shard_cdf_begin": ([_vp, _vp, _i64, _i32, _vp, _vp, _i32, _vp, _vp], _i32),
    "gjx_shard_cdf_finish": ([_vp, _vp, _vp, _i64, _vp, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp],
                             _i32),
    "gjx_shard_search": ([_vp, _i32, _vp, _i32, _i32, _i64, _f32, _i64, _vp, _vp, _vp, _vp, _vp], _i32),
    "gjx_shard_count": ([_vp, _vp, _i32, _i32, _i64, _vp, _vp, _vp], _i32),
    "gjx_shard_pack": ([_vp, _vp, _vp, _vp, _i32, _i32, _i64, C.POINTER(_vp), _vp, _vp], _i32),
    "gjx_shard_resample_moves": ([_vp, C.POINTER(LocWorld), _vp, _vp, _vp, _i64, _i32, _i64, _i64, _vp, _vp, _i32,
                                  _u32p, _i32, _i32, _f32, _i32, _i32, _f32p, C.POINTER(_vp),
                                  C.POINTER(_vp), _vp, _vp, _i64, _vp, _vp, _vp, _vp], _i32),
    "gjx_loc_smc_step_auto": ([_vp, C.POINTER(LocWorld), _u32p, _i32, _i64, _f32p, _f32, _u32p, _i32, _i32,
                               C.POINTER(_vp), C.POINTER(_vp), _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                               _vp, _vp], _i32),
    "gjx_shard_split": ([_vp, _vp, _i32, _i64, _vp, _i64, _vp, _vp, _i32, _vp], _i32),
    "gjx_shard_moves_main": ([_vp, C.POINTER(LocWorld), _vp, _vp, _i32, _i64, _vp, _u32p, _i32, _i32, _i32, _f32p,
                              C.POINTER(_vp), C.POINTER(_vp), _vp, _vp, _i64, _vp, _vp, _vp, _vp], _i32),
    "gjx_shard_moves_fixup": ([_vp, C.POINTER(LocWorld), _vp, _i64, _i32, _i64, _i64, _i64, _i64, _vp, _vp, _u32p,
                               _i32, _i32, _i32, _f32p, C.POINTER(_vp), C.POINTER(_vp), _vp, _vp, _i64, _vp, _vp],
                              _i32),
    "gjx_comm_unique_id": ([C.POINTER(C.c_uint8)], _i32),
    "gjx_comm_create": ([C.POINTER(C.c_uint8), _i32, _i32, C.POINTER(_vp)], _i32),
    "gjx_comm_destroy": ([_vp], _i32),
    "gjx_comm_all_gather": ([_vp, _vp, _vp, _i64, _vp], _i32),
    "gjx_shard_front": ([_vp, _vp, C.POINTER(LocWorld), _u32p, _i32, _f32p, _f32, C.POINTER(ShardBufs), _i32, _vp,
                         _u32p, _i32, C.POINTER(_vp), _vp, _vp, _vp, _vp], _i32),
    "gjx_lgssm_smc_run": ([_vp, C.POINTER(LgssmParams), _u32p, _i32, _f32p, _i64, _i32] + [_vp] * 11
                          + [C.POINTER(C.c_int32), _vp], _i32),
    "gjx_ctx_set_stage_events": ([_vp, C.POINTER(_vp), _i32], _i32),
    "gjx_event_create": ([C.POINTER(_vp)], _i32),
    "gjx_event_record": ([_vp, _vp], _i32),
    "gjx_event_destroy": ([_vp], _i32),
    "gjx_event_elapsed_ms": ([_vp, _vp, C.POINTER(C.c_float)], _i32),
    "gjx_spin_us": ([_i32, _vp], _i32),
    "gjx_loc_mh": ([_vp, C.POINTER(LocWorld), _u32p, _i32, _i32, _i32, _u64, _i64, _f32p] + [_vp] * 9 + [_vp, _vp], _i32),
    "gjx_loc_mh_resampled": ([_vp, C.POINTER(LocWorld), _u32p, _i32, _i32, _i32, _u64, _i64, _f32p, _vp, _i64,
                              C.POINTER(_vp), C.POINTER(_vp), _vp, _i64, _vp, _vp, _vp], _i32),
    "gjx_loc_hmc": ([_vp, C.POINTER(LocWorld), _u32p, _i32, _i32, _f32, _i32, _i32, _u64, _i64, _f32p] + [_vp] * 9 + [_vp, _vp], _i32),
    "gjx_loc_regenerate": ([C.POINTER(LocWorld), _u32p, _i32, _i32, _u64, _i64, _f32p] + [_vp] * 7 + [_i64] + [_vp] * 6
                           + [_i64, _vp], _i32),
    "gjx_loc_hmc_propose": ([C.POINTER(LocWorld), _u32p, _i32, _f32, _i32, _i32, _u64, _i64, _f32p] + [_vp] * 13
                            + [_vp], _i32),
    "gjx_loc_logp_grad": ([C.POINTER(LocWorld), _i32, _i64, _f32p] + [_vp] * 8 + [_vp], _i32),
    "gjx_lgssm_extend": ([_vp, C.POINTER(LgssmParams), _u32p, _i32, _u64, _i64, _f32p] + [_vp] * 5 + [_vp], _i32),
    "gjx_lgssm_mh": ([_vp, C.POINTER(LgssmParams), _u32p, _i32, _i32, _u64, _i64, _f32p] + [_vp] * 3 + [_vp, _vp], _i32),
    "gjx_outlier_init": ([C.POINTER(OutlierModel), _u32p, _u64, _i64] + [_vp] * 6 + [_vp], _i32),
    "gjx_outlier_gibbs_hmc": ([C.POINTER(OutlierModel), _i32, _f32, _i32, _i64] + [_vp] * 7 + [_vp], _i32),
    "gjx_outlier_gibbs_hmc_diag": ([C.POINTER(OutlierModel), _f32, _i32, _i64] + [_vp] * 8 + [_vp], _i32),
    "gjx_gol_init": ([_vp, _i64, _i32, _vp, _vp], _i32),
    "gjx_gol_gibbs_pass": ([_vp, _i64, _i32, _i32, _i32, _f32, _vp, _i64, _vp, _vp], _i32),
    "gjx_gol_sweep": ([_vp, _i64, _i32, _f32, _vp, _i64, _vp, _vp], _i32),
    "gjx_gol_words_per_row": ([_i32], _i32),
    "gjx_gol_pack": ([_vp, _i64, _i32, _vp, _vp], _i32),
    "gjx_gol_unpack": ([_vp, _i64, _i32, _vp, _vp], _i32),
    "gjx_gol_gibbs_pass_packed": ([_vp, _i64, _i32, _i32, _i32, _f32, _vp, _i64, _vp, _vp], _i32),
    "gjx_gol_sweep_packed": ([_vp, _i64, _i32, _f32, _vp, _i64, _vp, _vp], _i32),
    "gjx_gol_step": ([_vp, _i64, _i32, _vp, _vp], _i32),
    "gjx_gol_summary": ([_vp, _vp, _i64, _i64, _i32, _f32, _vp, _vp, _vp], _i32),
    "gjx_gol_keys": ([_u32p, _i32, _i32, _u32p, _u32p], None),
    "gjx_poly_is": ([_u32p, _u64, _i64, _f32p, _f32p, _i32, _f32] + [_vp] * 4 + [_vp], _i32),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def lib():
    """Load libgjx_hip.so; raise GjxError if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise GjxError(
                f"{_SO} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). genjax_amd has no CPU fallback."
            )
        L = C.CDLL(_SO)
        for name, (argtypes, restype) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = argtypes
            fn.restype = restype
        _lib = L
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().gjx_last_error().decode("utf-8", "replace")
        raise GjxError(f"{what or 'gjx call'} failed (code {rc}): {msg}")
