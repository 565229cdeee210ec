"""Timing protocol of the case studies (reference src/genjax/timing.py:23-166): every call is synchronised and
timed on its own, an outer run aggregates its inner samples with a trimmed mean (20 % off either end), and the
outer runs give mean and standard deviation.  Synchronisation is the device's: torch.cuda.synchronize()."""
from __future__ import annotations

import time
from typing import Any, Callable, Optional

import numpy as np
import torch


def _block_until_ready(result: Any) -> None:
    if torch.cuda.is_available() and torch.cuda.is_initialized():
        torch.cuda.synchronize()


def _aggregate(samples: np.ndarray, trim_fraction: float) -> float:
    """Trimmed mean of per-iteration timings (timing.py:23-34)."""
    samples = np.asarray(samples, np.float64)
    if samples.size == 0:
        return 0.0
    trimmed = samples
    if 0.0 < trim_fraction < 0.5 and samples.size > 2:
        k = int(samples.size * trim_fraction)
        if k > 0 and 2 * k < samples.size:
            trimmed = np.sort(samples)[k:-k]
    return float(np.mean(trimmed))


def timing(fn: Callable[[], Any], repeats: int = 20, inner_repeats: int = 20, auto_sync: bool = True,
           trim_fraction: float = 0.2):
    """-> (times (repeats,), (mean_time, std_time)) in seconds (timing.py:37-94)."""
    times = []
    for _ in range(repeats):
        samples = []
        for _ in range(inner_repeats):
            start = time.perf_counter_ns()
            result = fn()
            if auto_sync:
                _block_until_ready(result)
            samples.append((time.perf_counter_ns() - start) / 1e9)
        times.append(_aggregate(np.array(samples), trim_fraction))
    times = np.array(times)
    return times, (float(np.mean(times)) if times.size else 0.0, float(np.std(times)) if times.size else 0.0)


def benchmark_with_warmup(fn: Callable[[], Any], warmup_runs: int = 2, repeats: int = 10, inner_repeats: int = 10,
                          auto_sync: bool = True, trim_fraction: float = 0.2):
    """Warm-up calls (first-launch set-up: module load, workspace allocation) and then timing() (timing.py:97-137)."""
    for _ in range(warmup_runs):
        result = fn()
        if auto_sync:
            _block_until_ready(result)
    return timing(fn, repeats=repeats, inner_repeats=inner_repeats, auto_sync=auto_sync, trim_fraction=trim_fraction)


def compare_timings(*timing_results, labels: Optional[list] = None) -> None:
    """Print mean / std in ms and the percentage of the first entry's mean (timing.py:140-166)."""
    if labels is None:
        labels = [f"Method {i + 1}" for i in range(len(timing_results))]
    print(f"{'Method':<15} {'Mean (ms)':<12} {'Std (ms)':<12} {'Relative':<12}")
    print("-" * 55)
    base = timing_results[0][1][0] if timing_results else 1.0
    for (_, (mean, std)), label in zip(timing_results, labels):
        rel = mean / base * 100 if base > 0 else float("nan")
        print(f"{label:<15} {mean * 1000:<12.3f} {std * 1000:<12.3f} {rel:<12.1f}%")
