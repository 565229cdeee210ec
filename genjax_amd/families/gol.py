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


def generate_next_state(prev_state: torch.Tensor, p_flip=None) -> torch.Tensor:
    """generate_next_state(prev_state, p_flip) (core.py:66-83): every cell of the next board is
    flip(1 - p_flip) of its Life outcome (one site key for the whole board); without p_flip (or 0) the rule
    of Life itself."""
    single = prev_state.dim() == 2
    out = B.gol_step(prev_state[None] if single else prev_state)
    out = out[0] if single else out
    if p_flip is None or float(p_flip) == 0.0:
        return out
    from ..distributions import flip

    p_is_one = torch.where(out != 0, 1.0 - float(p_flip), float(p_flip)).to(torch.float32)
    return flip.sample(p_is_one).to(torch.uint8)


def generate_state_pair(n_y: int, n_x: int, p_flip):
    """A uniformly random board and its noisy successor (core.py:85-100): -> (init, step)."""
    from ..distributions import flip

    if int(n_y) != int(n_x):
        raise GjxError("Game-of-Life boards must be square")
    init = flip.sample(torch.full((int(n_y), int(n_x)), 0.5, device=B._dev())).to(torch.uint8)
    return init, generate_next_state(init, p_flip)


def get_windows(state: torch.Tensor) -> torch.Tensor:
    """(n_y, n_x, 3, 3) windows of the zero-padded board, one centred on every cell (core.py:49-63)."""
    padded = torch.nn.functional.pad(state, (1, 1, 1, 1))
    return padded.unfold(0, 3, 1).unfold(1, 3, 1)


def get_gibbs_probs_fast(i: int, j: int, current_state: torch.Tensor, future_state: torch.Tensor, p_flip):
    """[P(cell = 0), P(cell = 1)] of the single-site conditional at (i, j) (core.py:108-153): for either value
    the sum, over the up to nine next-step cells whose window contains (i, j), of the log-probability of the
    observed future bit; softmax of the two sums.  Host f64 on a 5 x 5 patch -- the pass kernels evaluate the
    same conditional for a third of the rows and columns at once."""
    cur = current_state.detach().to("cpu", torch.int64).numpy()
    fut = future_state.detach().to("cpu", torch.int64).numpy()
    n_y, n_x = cur.shape
    pad = np.pad(cur, 2)
    scores = []
    for val in (0, 1):
        patch = pad[i:i + 5, j:j + 5].copy()
        patch[2, 2] = val
        total = 0.0
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                y, x = i + dy, j + dx
                if not (0 <= y < n_y and 0 <= x < n_x):
                    continue
                win = patch[1 + dy:4 + dy, 1 + dx:4 + dx]
                nb = int(win.sum() - win[1, 1])
                alive = 1 if ((win[1, 1] == 1 and nb in (2, 3)) or (win[1, 1] == 0 and nb == 3)) else 0
                p_one = 1.0 - float(p_flip) if alive else float(p_flip)
                total += float(np.log(p_one if fut[y, x] else 1.0 - p_one))
        scores.append(total)
    m = max(scores)
    e = np.exp(np.asarray(scores) - m)
    return torch.tensor(e / e.sum(), dtype=torch.float64)


def gibbs_move_on_all_cells_at_offset(oy: int, ox: int, current_state, future_state, p_flip):
    """One colour pass: every cell (oy + 3a, ox + 3b) redrawn from its conditional, one site key for the pass
    (core.py:167-185).  Returns the updated board; the input is left as it was."""
    from ..core import _next_key

    boards = current_state.to(torch.uint8).clone()
    single = boards.dim() == 2
    boards = boards[None] if single else boards
    key = np.asarray(_next_key(), np.uint32).reshape(1, 2)
    out = B.gol_gibbs_pass(boards, future_state.to(torch.uint8), np.repeat(key, boards.shape[0], axis=0), int(oy), int(ox),
                           float(p_flip))
    return out[0] if single else out


def full_gibbs_sweep(current_state, future_state, p_flip):
    """The nine colour passes in the reference's order (core.py:188-204)."""
    for oy in range(3):
        for ox in range(3):
            current_state = gibbs_move_on_all_cells_at_offset(oy, ox, current_state, future_state, p_flip)
    return current_state


# the case study's small known patterns (examples/gol/data.py:6-38); the logo images of data.py:41-268 are
# picture files read with matplotlib and are not part of this package
def get_blinker_4x4() -> torch.Tensor:
    return torch.tensor([[0, 0, 0, 0], [0, 1, 1, 1], [0, 0, 0, 0], [0, 0, 0, 0]], dtype=torch.uint8, device=B._dev())


def get_blinker_10x10() -> torch.Tensor:
    board = torch.zeros((10, 10), dtype=torch.uint8, device=B._dev())
    board[2, 4:7] = 1
    return board


def get_blinker_n(n: int) -> torch.Tensor:
    """zeros((n, n)).at[n - 2, n - 2 : n + 1].set(1): the slice is clipped to the board (data.py:36-38)."""
    board = torch.zeros((n, n), dtype=torch.uint8, device=B._dev())
    board[n - 2, n - 2:n + 1] = 1
    return board


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
