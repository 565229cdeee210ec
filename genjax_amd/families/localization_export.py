"""On-disk export format of the localization experiments.

File names, CSV headers and row layout are those of the reference's exporter
(examples/localization/export.py:15-137) so that its plotting scripts read a run produced here:

    <dir>/experiment_metadata.json
    <dir>/ground_truth_poses.csv          timestep,x,y,theta
    <dir>/ground_truth_observations.csv   timestep,ray_0..ray_{R-1}   (or timestep,observation)
    <dir>/benchmark_summary.csv           method,mean_time_sec,std_time_sec,n_particles,n_timesteps
    <dir>/<method>/timing.csv             mean_time_sec,std_time_sec
    <dir>/<method>/particles/timestep_TTT.csv   particle_id,x,y,theta,weight
    <dir>/<method>/diagnostic_weights.csv       timestep,particle_0..particle_{N-1}

The store here is columnar: a timestep of the particle history is a `Pose` whose fields are
length-N columns (device tensors or arrays), not a list of N scalar poses; columns are copied to
the host once per timestep and formatted in bulk.  Numbers are written as the shortest text that
round-trips the f32 value widened to f64 (what `float(x)` of an f32 prints in the reference).
"""
from __future__ import annotations

import csv
import json
import os
from typing import Any, Dict, Sequence

import numpy as np

from .localization import Pose


def _host(a) -> np.ndarray:
    """Column -> 1-d float64 host array (f32 values widened exactly)."""
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    return np.atleast_1d(np.asarray(a)).astype(np.float64)


def _pose_columns(p) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
    """A timestep of history as three columns: a Pose of columns, or a sequence of scalar Poses."""
    if isinstance(p, Pose) or hasattr(p, "theta"):
        return _host(p.x), _host(p.y), _host(p.theta)
    return (np.array([float(q.x) for q in p]), np.array([float(q.y) for q in p]),
            np.array([float(q.theta) for q in p]))


def save_experiment_metadata(data_dir: str, config: Dict[str, Any]) -> str:
    os.makedirs(data_dir, exist_ok=True)
    path = os.path.join(data_dir, "experiment_metadata.json")
    with open(path, "w") as f:
        json.dump(config, f, indent=2)
    return path


def save_ground_truth_data(data_dir: str, true_poses: Sequence[Pose], observations: Sequence, controls=None):
    """Ground-truth trajectory and lidar readings (export.py:23-50); `controls` is accepted and
    not written, as in the reference."""
    os.makedirs(data_dir, exist_ok=True)
    poses_path = os.path.join(data_dir, "ground_truth_poses.csv")
    with open(poses_path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["timestep", "x", "y", "theta"])
        for t, p in enumerate(true_poses):
            w.writerow([t, float(_host(p.x)[0]), float(_host(p.y)[0]), float(_host(p.theta)[0])])
    obs_path = os.path.join(data_dir, "ground_truth_observations.csv")
    rows = [_host(o) for o in observations]
    with open(obs_path, "w", newline="") as f:
        w = csv.writer(f)
        if rows and rows[0].size > 1:
            w.writerow(["timestep"] + [f"ray_{i}" for i in range(rows[0].size)])
            for t, o in enumerate(rows):
                w.writerow([t] + o.tolist())
        else:
            w.writerow(["timestep", "observation"])
            for t, o in enumerate(rows):
                w.writerow([t, float(o[0])])
    return poses_path, obs_path


def save_smc_results(data_dir: str, method_name: str, result: Dict[str, Any]) -> str:
    """One method's particles, weights and diagnostics (export.py:53-102).  result keys:
    particle_history (T entries, each a Pose of N-columns), weight_history (T x N),
    diagnostic_weights ((T, N) or None), timing_stats (mean_sec, std_sec)."""
    method_dir = os.path.join(data_dir, method_name)
    particles_dir = os.path.join(method_dir, "particles")
    os.makedirs(particles_dir, exist_ok=True)
    mean_t, std_t = result["timing_stats"]
    with open(os.path.join(method_dir, "timing.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["mean_time_sec", "std_time_sec"])
        w.writerow([float(mean_t), float(std_t)])
    for t, (particles, weights) in enumerate(zip(result["particle_history"], result["weight_history"])):
        x, y, th = _pose_columns(particles)
        wt = _host(weights)
        if not (x.size == y.size == th.size == wt.size):
            raise ValueError(f"timestep {t}: column lengths differ ({x.size}, {y.size}, {th.size}, {wt.size})")
        with open(os.path.join(particles_dir, f"timestep_{t:03d}.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["particle_id", "x", "y", "theta", "weight"])
            w.writerows(zip(range(x.size), x.tolist(), y.tolist(), th.tolist(), wt.tolist()))
    diag = result.get("diagnostic_weights")
    if diag is not None and hasattr(diag, "shape"):
        d = diag.detach().cpu().numpy() if hasattr(diag, "detach") else np.asarray(diag)
        d = d.astype(np.float64)
        with open(os.path.join(method_dir, "diagnostic_weights.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["timestep"] + [f"particle_{i}" for i in range(d.shape[1])])
            for t in range(d.shape[0]):
                w.writerow([t] + d[t].tolist())
    return method_dir


def save_benchmark_results(data_dir: str, benchmark_results: Dict[str, Dict[str, Any]], config: Dict[str, Any]) -> str:
    """Metadata, every method's results and the summary table (export.py:105-137)."""
    save_experiment_metadata(data_dir, config)
    for name, result in benchmark_results.items():
        save_smc_results(data_dir, name, result)
    path = os.path.join(data_dir, "benchmark_summary.csv")
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["method", "mean_time_sec", "std_time_sec", "n_particles", "n_timesteps"])
        for name, result in benchmark_results.items():
            hist = result["particle_history"]
            w.writerow([name, float(result["timing_stats"][0]), float(result["timing_stats"][1]),
                        _pose_columns(hist[0])[0].size, len(hist)])
    return path


# ----------------------------------------------------------------------------- loading
def load_experiment_metadata(data_dir: str) -> Dict[str, Any]:
    with open(os.path.join(data_dir, "experiment_metadata.json")) as f:
        return json.load(f)


def load_ground_truth_data(data_dir: str):
    """-> (list of scalar Poses, list of observations: R-lists or floats) (export.py:147-172)."""
    poses = []
    with open(os.path.join(data_dir, "ground_truth_poses.csv")) as f:
        for row in csv.DictReader(f):
            poses.append(Pose(float(row["x"]), float(row["y"]), float(row["theta"])))
    observations = []
    with open(os.path.join(data_dir, "ground_truth_observations.csv")) as f:
        for row in csv.DictReader(f):
            if "ray_0" in row:
                observations.append([float(row[f"ray_{i}"]) for i in range(len(row) - 1)])
            else:
                observations.append(float(row["observation"]))
    return poses, observations


def load_smc_results(data_dir: str, method_name: str) -> Dict[str, Any]:
    """Inverse of save_smc_results; history comes back columnar (Pose of f32 columns per
    timestep), weights as f32 arrays, diagnostic_weights as a (T, N) f32 array or None."""
    method_dir = os.path.join(data_dir, method_name)
    with open(os.path.join(method_dir, "timing.csv")) as f:
        row = next(csv.DictReader(f))
        timing = (float(row["mean_time_sec"]), float(row["std_time_sec"]))
    particles_dir = os.path.join(method_dir, "particles")
    history, weights = [], []
    for name in sorted(n for n in os.listdir(particles_dir) if n.startswith("timestep_")):
        data = np.loadtxt(os.path.join(particles_dir, name), delimiter=",", skiprows=1, ndmin=2)
        history.append(Pose(data[:, 1].astype(np.float32), data[:, 2].astype(np.float32),
                            data[:, 3].astype(np.float32)))
        weights.append(data[:, 4].astype(np.float32))
    diag = None
    diag_path = os.path.join(method_dir, "diagnostic_weights.csv")
    if os.path.exists(diag_path):
        data = np.loadtxt(diag_path, delimiter=",", skiprows=1, ndmin=2)
        if data.size:
            diag = np.zeros((data.shape[0], data.shape[1] - 1), np.float32)
            diag[data[:, 0].astype(int)] = data[:, 1:]
    return {"timing_stats": timing, "particle_history": history, "weight_history": weights,
            "diagnostic_weights": diag}


def load_benchmark_results(data_dir: str) -> Dict[str, Dict[str, Any]]:
    with open(os.path.join(data_dir, "benchmark_summary.csv")) as f:
        names = [row["method"] for row in csv.DictReader(f)]
    return {name: load_smc_results(data_dir, name) for name in names}


def collection_history(particles, weights_normalised: bool = True):
    """(particle_history, weight_history, diagnostic_weights) of a `return_all_particles=True`
    ParticleCollection (smc.py:775-783): per timestep a Pose of columns and the normalised
    weights exp(logw - logsumexp(logw)) the reference's plots use."""
    from .. import backend as B

    ch = particles.traces.get_choices()
    x, y, th = ch["x"], ch["y"], ch["theta"]
    lw = particles.log_weights
    T = x.shape[0]
    history = [Pose(x[t], y[t], th[t]) for t in range(T)]
    if weights_normalised:
        wts = []
        for t in range(T):
            row = lw[t].contiguous()
            wts.append(B.normalized_weights(row, B.logsumexp_ess(row)[1:2]))     # exp(logw - lse), smc.py:127-128
    else:
        wts = [lw[t] for t in range(T)]
    return history, wts, getattr(particles, "diagnostic_weights", None)
