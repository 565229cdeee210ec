"""Game-of-Life inversion case study (reference examples/gol/core.py): infer the previous board
from an observed next board by single-site Gibbs sweeps, nine colour passes per sweep.

The reference runs ONE chain (`run_sampler_and_get_summary`, core.py:353-364); BASELINE.json's
config asks for batched chains, so `run_sampler_and_get_summary` here takes `n_chains`: chain c
runs the reference's program under the key `split(key, n_chains_total)[chain_offset + c]`
(n_chains = 1 keeps the reference's single-key program).  Boards live on the device as bytes;
every colour pass is one launch of gjx_gol_gibbs_pass over all chains."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

from .. import backend as B
from .._native import GjxError


def generate_next_state(prev_state: torch.Tensor) -> torch.Tensor:
    """generate_next_state(prev_state, p_flip = 0) (core.py:68-83): the rule of Life."""
    single = prev_state.dim() == 2
    out = B.gol_step(prev_state[None] if single else prev_state)
    return out[0] if single else out


class GibbsSamplerState:
    """Boards of all chains = choices["init"] (core.py:207-230).  The sweeps run on a bit-packed copy
    (gjx_gol_sweep_packed); `boards` ((B, n, n) uint8) is unpacked from it on access."""

    def __init__(self, boards: torch.Tensor, target: torch.Tensor, p_flip_value: float):
        self._bytes, self._bits = boards, None
        self.target, self.p_flip_value = target, p_flip_value
        self.n = int(boards.shape[-1])

    @property
    def boards(self) -> torch.Tensor:
        if self._bytes is None:
            self._bytes = B.gol_unpack(self._bits, self.n)
        return self._bytes

    def packed(self) -> torch.Tensor:
        """The bit-packed boards, for in-place sweeps: the byte copy is dropped (it would go stale)."""
        if self._bits is None:
            self._bits = B.gol_pack(self._bytes)
        self._bytes = None
        return self._bits

    def inferred_prev_board(self):
        return self.boards

    def inferred_reconstructed_target(self):
        return B.gol_step(self.boards)

    def predictive_posterior_score(self):
        return B.gol_summary(self.boards, self.target, self.p_flip_value)[1]

    def trace_score(self):
        """-log p(init, step) = -(n^2 log 0.5 + predictive score) (Tr.get_score, core.py:673-677)."""
        n = self.boards.shape[-1]
        return -(n * n * float(np.log(0.5)) + self.predictive_posterior_score())

    def n_incorrect_bits(self):
        return B.gol_summary(self.boards, self.target, self.p_flip_value)[0]

    def p_flip(self):
        return self.p_flip_value


class GibbsSampler:
    """GibbsSampler(target_image, p_flip) (core.py:233-287)."""

    def __init__(self, target_image, p_flip: float):
        t = target_image if isinstance(target_image, torch.Tensor) else torch.as_tensor(np.asarray(target_image))
        if t.dim() != 2 or t.shape[0] != t.shape[1]:
            raise GjxError("GibbsSampler: the target must be a square 2-D board")
        self.target_image = t.to(torch.uint8).cuda().contiguous()
        self.target_bits = B.gol_pack(self.target_image)[0]
        self.p_flip = float(p_flip)

    def get_initial_state(self, init_keys) -> GibbsSamplerState:
        boards = B.gol_init(np.asarray(init_keys, np.uint32).reshape(-1, 2), self.target_image.shape[0],
                            device=self.target_image.device)
        return GibbsSamplerState(boards, self.target_image, self.p_flip)

    def update_state(self, state: GibbsSamplerState, pass_keys9) -> GibbsSamplerState:
        """One full_gibbs_sweep (core.py:266-287) in place; pass_keys9: (9, B, 2)."""
        B.gol_sweep_packed(state.packed(), state.n, self.target_bits, pass_keys9, self.p_flip)
        return state


@dataclass
class GibbsRunSummary:
    """core.py:320-329, with a leading frame axis and a chain axis."""

    predictive_posterior_scores: torch.Tensor   # (n_frames + 1, B) float64
    n_incorrect_bits: torch.Tensor              # (n_frames + 1, B) int64
    inferred_prev_boards: torch.Tensor          # (B, n, n) final boards
    inferred_reconstructed_targets: torch.Tensor

    def n_incorrect_bits_in_reconstructed_image(self, target_image=None):
        return self.n_incorrect_bits[-1]


def chain_keys(key, n_chains: int, chain_offset: int = 0, n_chains_total: int | None = None) -> np.ndarray:
    """Per-chain keys: the key itself for a single chain (the reference's program), else
    split(key, total)[offset : offset + n_chains] (sharding-invariant)."""
    total = n_chains_total or (chain_offset + n_chains)
    if total == 1:
        return np.asarray(key, np.uint32).reshape(1, 2)
    return B.split(key, total)[chain_offset:chain_offset + n_chains]


def run_sampler_and_get_summary(key, sampler: GibbsSampler, n_steps: int, n_steps_per_summary_frame: int,
                                n_chains: int = 1, chain_offset: int = 0, n_chains_total: int | None = None
                                ) -> GibbsRunSummary:
    """seed(_run_sampler_and_get_summary)(key, sampler) (core.py:312-364) for n_chains chains."""
    n_frames = n_steps // n_steps_per_summary_frame
    ck = chain_keys(key, n_chains, chain_offset, n_chains_total)
    sched = [B.gol_keys(k, n_frames, n_steps_per_summary_frame) for k in ck]
    init_keys = np.stack([s[0] for s in sched])
    pass_keys = np.stack([s[1] for s in sched], axis=3)          # (frames, per_frame, 9, B, 2)
    dev = sampler.target_image.device
    pk_dev = torch.from_numpy(np.ascontiguousarray(pass_keys).view(np.int32)).to(dev)
    state = sampler.get_initial_state(init_keys)
    scores, bad = [], []

    def record():
        b_, s_ = B.gol_summary(state.boards, sampler.target_image, sampler.p_flip)
        bad.append(b_)
        scores.append(s_)

    record()
    for f in range(n_frames):
        for g in range(n_steps_per_summary_frame):
            sampler.update_state(state, pk_dev[f, g])
        record()
    return GibbsRunSummary(torch.stack(scores), torch.stack(bad), state.boards, B.gol_step(state.boards))
