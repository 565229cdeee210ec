"""Localization model family: the north-star case study (reference examples/localization/
core.py and data.py) on fused HIP kernels.

`localization_model` keeps the reference's signature `(prev_pose, time_index, world,
n_rays=Const(8))` and choice addresses "x", "y", "theta", "measurements"/"distance"
(core.py:312-395).  Particle state is the minimal SoA of SURVEY.md Appendix B:
choices x,y,theta (unclipped), the prior means mx,my,mth used at this step, and the
log-weights kept by the ParticleCollection.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any

import numpy as np
import torch

from .. import backend as B
from ..core import Const, Selection, const, gen as _gen
from .._native import GjxError
from .base import FusedFamily, FusedTrace, MoveSpec


# --------------------------------------------------------------------------- #
# data structures (core.py:15-77)
# --------------------------------------------------------------------------- #
@dataclass
class Pose:
    x: Any
    y: Any
    theta: Any


@dataclass
class Control:
    velocity: float
    angular_velocity: float


@dataclass
class World:
    """Multi-room world: boundary + internal wall segments (core.py:43-77)."""

    __gjx_static__ = True       # shared by all particles: the inference library's tree walks leave it alone

    width: float
    height: float
    walls_x1: Any = None
    walls_y1: Any = None
    walls_x2: Any = None
    walls_y2: Any = None
    num_walls: int = 0

    def __post_init__(self):
        if self.num_walls < 1:
            raise ValueError(f"World must have at least 1 internal wall, got {self.num_walls}")
        for name in ("walls_x1", "walls_y1", "walls_x2", "walls_y2"):
            arr = np.asarray(getattr(self, name), dtype=np.float32)
            if arr.shape != (self.num_walls,):
                raise ValueError(f"Array {name} has shape {arr.shape}, expected {(self.num_walls,)}")
            setattr(self, name, arr)
        self._native = {}

    def walls(self) -> np.ndarray:
        return np.stack([self.walls_x1, self.walls_y1, self.walls_x2, self.walls_y2], axis=1)

    def native(self, n_rays: int):
        if n_rays not in self._native:
            self._native[n_rays] = B.make_loc_world(self.width, self.height, self.walls(), n_rays=n_rays)
        return self._native[n_rays]


_BASIC = [[4.0, 0.0, 4.0, 3.0], [4.0, 5.0, 4.0, 10.0], [8.0, 0.0, 8.0, 4.0], [8.0, 6.0, 8.0, 10.0],
          [5.0, 7.0, 7.0, 7.0], [9.0, 2.0, 10.0, 2.0], [10.0, 2.0, 10.0, 3.0], [10.0, 3.0, 9.0, 3.0],
          [9.0, 3.0, 9.0, 2.0]]
_COMPLEX = _BASIC[:5] + [[9.0, 2.0, 10.5, 2.5], [10.5, 2.5, 10.0, 3.5], [10.0, 3.5, 8.5, 3.0],
                         [8.5, 3.0, 9.0, 2.0], [1.0, 1.0, 2.5, 0.5], [3.0, 8.5, 3.5, 9.5],
                         [5.5, 2.0, 6.5, 1.0], [6.0, 4.5, 7.0, 5.5], [8.5, 0.5, 9.5, 1.5],
                         [10.5, 8.0, 11.0, 9.0]]


def _world_from(coords) -> World:
    a = np.asarray(coords, dtype=np.float32)
    return World(12.0, 10.0, a[:, 0], a[:, 1], a[:, 2], a[:, 3], len(coords))


def create_basic_multi_room_world() -> World:
    """3 rooms, 9 axis-aligned wall segments (core.py:1006-1043)."""
    return _world_from(_BASIC)


def create_complex_multi_room_world() -> World:
    """15 segments including slanted ones (core.py:1046-1092)."""
    return _world_from(_COMPLEX)


def create_multi_room_world(world_type="basic") -> World:
    if world_type == "basic":
        return create_basic_multi_room_world()
    if world_type == "complex":
        return create_complex_multi_room_world()
    raise ValueError(f"Unknown world_type: {world_type}. Use 'basic' or 'complex'")


def create_waypoint_trajectory_room1_to_room3():
    """17 waypoints from room 1 to room 3 (data.py:62-101)."""
    return [(1.2, 1.2), (1.5, 1.5), (2.5, 2.5), (3.5, 3.5), (3.9, 4.0), (4.1, 4.0), (4.5, 4.0), (5.5, 4.2),
            (6.5, 4.5), (7.5, 5.0), (7.9, 5.0), (8.1, 5.0), (8.5, 5.0), (9.5, 6.0), (10.5, 7.5), (11.0, 8.5),
            (11.5, 9.5)]


def distance_to_wall_lidar(pose: Pose, world: World, n_angles: int = 128, max_range: float = 10.0):
    """LIDAR distances (core.py:252-308) for a pose or a column of poses -> (n_angles,) or (N, n_angles)."""
    px, py, th = (torch.as_tensor(np.asarray(v, dtype=np.float32) if not isinstance(v, torch.Tensor) else v)
                  for v in (pose.x, pose.y, pose.theta))
    single = px.dim() == 0
    dev = B._dev()
    w = world.native(n_angles)
    if max_range != 10.0:
        w = B.make_loc_world(world.width, world.height, world.walls(), n_rays=n_angles, max_range=max_range)
    if torch.is_grad_enabled() and any(t.requires_grad for t in (px, py, th)):
        # under autograd (generic hmc / mala on a model that calls the sensor): the library's Jacobian kernel is the
        # backward, as jax.grad differentiates the reference's function (mcmc.py:59-88)
        n = max(t.numel() for t in (px, py, th))
        cols = [t.to(dev).to(torch.float32).reshape(-1).expand(n) for t in (px, py, th)]
        d = _LidarFn.apply(w, *cols)
        return d[0] if single else d
    px, py, th = (t.reshape(-1).to(dev).contiguous() for t in (px, py, th))
    d = B.loc_lidar(w, px, py, th)
    return d[0] if single else d


class _LidarFn(torch.autograd.Function):
    """distance_to_wall_lidar under autograd: forward = gjx_loc_lidar_jac (distances and their partials in one
    launch), backward = the chain rule over the rays (a sum the user's graph would otherwise hold)."""

    @staticmethod
    def forward(ctx, w, px, py, th):
        d, jac = B.loc_lidar_jac(w, px.contiguous(), py.contiguous(), th.contiguous())
        ctx.save_for_backward(jac)
        return d

    @staticmethod
    def backward(ctx, g):
        (jac,) = ctx.saved_tensors
        gp = (jac * g.unsqueeze(0)).sum(dim=2)          # (3, n): d/dpx, d/dpy, d/dtheta
        return None, gp[0], gp[1], gp[2]


def generate_synthetic_data_from_waypoints(waypoints, world, key, noise_std=0.15, n_rays=8):
    """Noisy LIDAR observations at the waypoints (data.py:124-165): noise_keys = split(key, T),
    angle_keys = split(noise_keys[i], n_rays), obs = max(0, true + normal(angle_key) * noise_std).
    Built on the host (as the reference builds it outside the jitted program)."""
    poses = [Pose(x=x, y=y, theta=0.0) for x, y in waypoints]
    xs = np.array([p.x for p in poses], np.float32)
    ys = np.array([p.y for p in poses], np.float32)
    true_d = B.loc_lidar_host(world.native(n_rays), xs, ys, np.zeros_like(xs))
    noise_keys = B.split(key, len(waypoints))
    observations = []
    for i, nk in enumerate(noise_keys):
        row = np.zeros(n_rays, np.float32)
        for j, ak in enumerate(B.split(nk, n_rays)):
            noise = np.float32(B.host_normal(ak) * np.float32(noise_std))
            row[j] = max(np.float32(0.0), np.float32(true_d[i, j] + noise))
        observations.append(row)
    return poses, observations


def create_waypoint_trajectory_exploration_room1():
    """Ten waypoints exploring room 1 (data.py:104-121)."""
    return [(2.0, 2.0), (1.0, 1.0), (3.0, 1.0), (3.0, 3.0), (1.0, 3.0), (2.0, 5.0), (1.0, 7.0), (3.0, 8.0), (2.0, 9.0),
            (2.0, 6.0)]


def generate_ground_truth_data(world, key, trajectory_type="room_navigation", n_steps=16, n_rays=8):
    """(poses, controls, observations) as data.py:168-216."""
    if trajectory_type == "room_navigation":
        waypoints = create_waypoint_trajectory_room1_to_room3()
    elif trajectory_type == "exploration":
        waypoints = create_waypoint_trajectory_exploration_room1()
    else:
        raise ValueError(f"Unknown trajectory_type: {trajectory_type}. Supported types: 'room_navigation', 'exploration'")
    key1 = B.split(key, 2)[0]
    poses, observations = generate_synthetic_data_from_waypoints(waypoints, world, key1, 0.15, n_rays)
    controls = [Control(1.0, 0.0) for _ in range(len(poses) - 1)]
    return poses, controls, observations


def generate_multiple_trajectories(world, key, n_trajectories=5, trajectory_types=None):
    """[(trajectory_type, (poses, controls, observations)), ...] under split(key, n_trajectories), cycling
    through the trajectory types (data.py:219-244)."""
    trajectory_types = trajectory_types or ["room_navigation", "exploration"]
    keys = B.split(key, n_trajectories)
    return [(trajectory_types[i % len(trajectory_types)],
             generate_ground_truth_data(world, keys[i], trajectory_type=trajectory_types[i % len(trajectory_types)]))
            for i in range(n_trajectories)]


# --------------------------------------------------------------------------- #
# fused trace
# --------------------------------------------------------------------------- #
class LocTrace(FusedTrace):
    """Batched trace of localization_model over SoA columns.

    ray_scores: the per-ray measurement scores of these particles for `obs` as extend stored them
    (the Vmap sub-trace's scores, core.py:1416-1492), or None.
    A trace returned by gather() is LAZY: it keeps (ancestors, source trace) and only gathers when a
    column is read.  mh on such a trace (resample -> rejuvenate, smc.py:754-765) never reads one:
    the move kernel loads through the ancestor indices (gjx_loc_mh_resampled)."""

    _NAMES = ("x", "y", "th", "mx", "my", "mth")

    def __init__(self, family, x, y, th, mx, my, mth, obs, time_index, world, n_rays, ray_scores=None,
                 pending=None):
        self.family = family
        self._cols = None if pending is not None else [x, y, th, mx, my, mth]
        self._pending = pending            # (ancestors, idx_base, source LocTrace)
        self.ray_scores = ray_scores
        self.obs = np.asarray(obs, dtype=np.float32).reshape(-1)
        self.time_index, self.world, self.n_rays = int(time_index), world, int(n_rays)

    def _materialise(self):
        if self._cols is None:
            anc, base, src = self._pending
            self._cols = B.gather_columns(anc, src._materialise(), idx_base=base)
            self._pending = None
        return self._cols

    x = property(lambda self: self._materialise()[0])
    y = property(lambda self: self._materialise()[1])
    th = property(lambda self: self._materialise()[2])
    mx = property(lambda self: self._materialise()[3])
    my = property(lambda self: self._materialise()[4])
    mth = property(lambda self: self._materialise()[5])

    @property
    def n(self):
        return self._pending[0].numel() if self._cols is None else self._cols[0].numel()

    @property
    def is_initial(self):
        return self.time_index == 0

    def columns(self):
        return dict(zip(self._NAMES, self._materialise()))

    def _like(self, **cols):
        c = self.columns()
        c.update(cols)
        return LocTrace(self.family, c["x"], c["y"], c["th"], c["mx"], c["my"], c["mth"], self.obs,
                        self.time_index, self.world, self.n_rays)

    def gather(self, ancestors, idx_base=0):
        src = self
        if self._cols is None:             # gather of a lazy gather: settle the inner one first
            self._materialise()
        return LocTrace(self.family, None, None, None, None, None, None, self.obs, self.time_index, self.world,
                        self.n_rays, pending=(ancestors, idx_base, src))

    # reference accessors (core.py:656-677)
    def get_choices(self):
        obs = torch.as_tensor(self.obs, device=self.x.device).expand(self.n, -1)
        return {"x": self.x, "y": self.y, "theta": self.th, "measurements": {"distance": obs}}

    def get_retval(self):
        return LocRetval((ClippedPose(self), self.time_index + 1, self.world, const(self.n_rays)), self)

    def get_args(self):
        return ((Pose(self.mx, self.my, self.mth), self.time_index, self.world, const(self.n_rays)), {})

    def get_score(self):
        logp, _ = B.loc_logp_grad(self.world.native(self.n_rays), self.is_initial, self.obs, self.x, self.y,
                                  self.th, self.mx, self.my, self.mth)
        return -logp


class ClippedPose:
    """The model's return pose (core.py:361-364): clip(x, 0.1, W - 0.1), clip(y, 0.1, H - 0.1), theta of
    a trace, computed when a field is read (extend() feeds the choice columns to the kernel, which
    re-clips, so the common path never reads one)."""

    def __init__(self, trace):
        self._trace = trace

    @property
    def x(self):
        return torch.clamp(self._trace.x, 0.1, self._trace.world.width - 0.1)

    @property
    def y(self):
        return torch.clamp(self._trace.y, 0.1, self._trace.world.height - 0.1)

    @property
    def theta(self):
        return self._trace.th


class LocRetval(tuple):
    """Return value (pose, time_index + 1, world, n_rays) that remembers its trace, so that
    extend() can feed the choice columns straight back without re-materialising the pose."""

    def __new__(cls, items, trace):
        self = super().__new__(cls, items)
        self.trace = trace
        return self


# --------------------------------------------------------------------------- #
# the family
# --------------------------------------------------------------------------- #
class LocalizationModel(FusedFamily):
    """localization_model(prev_pose, time_index, world, n_rays=Const(8))  (core.py:312-383)."""

    name = "localization_model"
    addresses = ("x", "y", "theta")
    _mask_cache: dict = {}          # id(selection) -> (selection, site mask): Selection.match walks a tree per site

    @staticmethod
    def _obs_of(constraints, n_rays):
        try:
            obs = constraints["measurements"]["distance"]
        except (KeyError, TypeError):
            raise GjxError('localization_model: constraints must be {"measurements": {"distance": obs}}')
        extra = set(constraints) - {"measurements"}
        if extra:
            raise GjxError(f"localization_model (fused): constraints on {sorted(extra)} are not supported")
        obs = obs.detach().cpu().numpy() if isinstance(obs, torch.Tensor) else np.asarray(obs)
        obs = obs.astype(np.float32).reshape(-1)
        if obs.size != n_rays:
            raise GjxError(f"localization_model: observation has {obs.size} rays, n_rays={n_rays}")
        return obs

    @staticmethod
    def _unpack(args):
        if len(args) == 3:
            prev_pose, time_index, world = args
            n_rays = Const(8)
        else:
            prev_pose, time_index, world, n_rays = args
        n_rays = n_rays.value if isinstance(n_rays, Const) else int(n_rays)
        if isinstance(time_index, torch.Tensor):
            time_index = int(time_index.reshape(-1)[0].item())
        return prev_pose, int(time_index), world, n_rays

    def _keys(self, count):
        from ..core import _next_key

        return [_next_key() for _ in range(count)]

    def fused_init(self, args, n, constraints, offset=0):
        prev_pose, t, world, R = self._unpack(args)
        obs = self._obs_of(constraints, R)
        logw0 = torch.zeros(n, dtype=torch.float32, device=B._dev())
        prev = None
        if t != 0:
            prev = tuple(self._col(v, n) for v in (prev_pose.x, prev_pose.y, prev_pose.theta))
        scores = self._scores_buffer(R, n, logw0.device)
        out = B.loc_extend(world.native(R), self._keys(3), t == 0, obs, prev, logw0, offset=offset, ray_scores=scores)
        tr = LocTrace(self, out["x"], out["y"], out["th"], out["mx"], out["my"], out["mth"], obs, t, world, R,
                      ray_scores=scores)
        return tr, out["logw"]

    keep_ray_scores = True     # extend stores the per-ray scores for the mh launch of the same step

    def _scores_buffer(self, n_rays, n, device):
        if not self.keep_ray_scores:
            return None
        return torch.empty((n_rays, n), dtype=torch.float32, device=device)

    @staticmethod
    def _col(v, n):
        if isinstance(v, torch.Tensor) and v.numel() == n:
            return v.reshape(n).contiguous()
        return torch.full((n,), float(v), dtype=torch.float32, device=B._dev())

    def fused_extend(self, trace, log_weights, args, constraints, offset=0):
        n = log_weights.numel()
        if isinstance(args, LocRetval):
            # previous retval: clip(choices) -- the kernel re-clips, so feed the choice columns
            src = args.trace
            prev = (src.x, src.y, src.th)
            _, t, world, R = self._unpack(tuple(args))
        else:
            prev_pose, t, world, R = self._unpack(tuple(args))
            prev = tuple(self._col(v, n) for v in (prev_pose.x, prev_pose.y, prev_pose.theta))
        obs = self._obs_of(constraints, R)
        scores = self._scores_buffer(R, n, log_weights.device)
        out = B.loc_extend(world.native(R), self._keys(3), t == 0, obs, prev, log_weights, offset=offset,
                           ray_scores=scores)
        tr = LocTrace(self, out["x"], out["y"], out["th"], out["mx"], out["my"], out["mth"], obs, t, world, R,
                      ray_scores=scores)
        return tr, out["logw"]

    # -- moves
    def sel_mask(self, selection: Selection) -> int:
        hit = self._mask_cache.get(id(selection))
        if hit is not None and hit[0] is selection:
            return hit[1]
        if selection.match("measurements")[0]:
            self.unsupported("regenerating the observed 'measurements' site")
        mask = sum(bit for bit, a in ((1, "x"), (2, "y"), (4, "theta")) if selection.match(a)[0])
        if len(self._mask_cache) < 256:
            self._mask_cache[id(selection)] = (selection, mask)      # (the entry keeps the selection alive)
        return mask

    def move_keys_block(self, spec: MoveSpec, scan_key, n_moves: int) -> np.ndarray:
        """move_keys of the moves 0 .. n_moves - 1 of a rejuvenation scan, move m on a fresh interpreter on
        fold_in(scan_key, m) (smc.py:756-765): one host call (gjx_loc_move_keys) instead of a dozen splits per move."""
        return B.loc_move_keys(scan_key, n_moves, spec.kind, self.sel_mask(spec.selection))

    def move_keys(self, spec: MoveSpec, key) -> np.ndarray:
        """Sub-keys of ONE move drawn from a fresh interpreter on `key`, laid out as the kernels
        expect.  mh: x', y', theta' (model order, selected sites only), then u.  hmc: momenta in
        sorted-key order theta, x, y (selected only), one split for the leapfrog scan, then u
        (mcmc.py:497,535,565)."""
        seq = B.KeySeq(key)
        mask = self.sel_mask(spec.selection)
        out = np.zeros((4, 2), np.uint32)
        if spec.kind == "mh":
            for slot, bit in ((0, 1), (1, 2), (2, 4)):
                if mask & bit:
                    out[slot] = seq.next()
        else:
            for slot, bit in ((0, 4), (1, 1), (2, 2)):
                if mask & bit:
                    out[slot] = seq.next()
            seq.next()  # lax.scan over leapfrog steps takes one split and draws nothing
        out[3] = seq.next()
        return out

    def fused_moves(self, trace: LocTrace, spec: MoveSpec, move_keys, offset=0, accept_counts=None):
        mask = self.sel_mask(spec.selection)
        if mask == 0:
            return trace
        w = trace.world.native(trace.n_rays)
        if spec.kind == "mh" and trace._cols is None and len(move_keys) <= B.N.MAX_MOVES:
            # freshly resampled particles: the gather rides on the move kernel's loads, and the
            # source particles' stored scores (if any) replace the lidar solve of the starting pose
            anc, base, src = trace._pending
            out = B.loc_mh_resampled(w, move_keys, trace.is_initial, trace.obs, anc, src._materialise(),
                                     sel_mask=mask, offset=offset, idx_base=base, accept_counts=accept_counts,
                                     ray_scores=src.ray_scores)
            return LocTrace(self, *out, trace.obs, trace.time_index, trace.world, trace.n_rays)
        pose, means = (trace.x, trace.y, trace.th), (trace.mx, trace.my, trace.mth)
        if spec.kind == "mh":
            out = B.loc_mh(w, move_keys, trace.is_initial, trace.obs, pose, means, sel_mask=mask, offset=offset,
                           accept_counts=accept_counts)
        elif spec.kind == "hmc":
            out = B.loc_hmc(w, move_keys, spec.step_size, spec.n_steps, trace.is_initial, trace.obs, pose, means,
                            sel_mask=mask, offset=offset, accept_counts=accept_counts)
        else:
            raise GjxError(f"unknown move kind {spec.kind}")
        return trace._like(x=out[0], y=out[1], th=out[2])

    # -- GFI face (particle-batched)
    def generate(self, x, *args, **kwargs):
        """Fused when only the measurements are constrained; the @gen twin when a pose site is (the generate call
        behind a user-written proposal, smc.py:332-350, 494-509)."""
        from ..core import batch_size

        if set(x or {}) - {"measurements"} or "measurements" not in (x or {}):
            return self.generic.generate(x, *args, **kwargs)
        n = batch_size() or 1
        tr, w = self.fused_init(args, n, x)
        return tr, w

    def assess(self, x, *args, **kwargs):
        prev_pose, t, world, R = self._unpack(args)
        obs = self._obs_of({"measurements": x["measurements"]}, R)
        cx = x["x"].reshape(-1).contiguous()
        n = cx.numel()
        cy, cth = self._col(x["y"], n), self._col(x["theta"], n)
        if t == 0:
            w = world.native(R)
            means = tuple(torch.full((n,), float(w.init_mean[i]), device=cx.device) for i in range(3))
        else:
            means = (torch.clamp(self._col(prev_pose.x, n), 0.1, world.width - 0.1),
                     torch.clamp(self._col(prev_pose.y, n), 0.1, world.height - 0.1), self._col(prev_pose.theta, n))
        logp, _ = B.loc_logp_grad(world.native(R), t == 0, obs, cx, cy, cth, *means)
        tr = LocTrace(self, cx, cy, cth, *means, obs, t, world, R)
        return logp, tr.get_retval()

    # simulate / update / regenerate / merge / filter: the @gen twin below (FusedFamily delegates to `generic`)


def _is_initial(time_index) -> bool:
    if isinstance(time_index, torch.Tensor):
        return int(time_index.reshape(-1)[0].item()) == 0
    return int(time_index) == 0


def _sensor_model_single_ray(true_distance, ray_idx):
    """One LIDAR ray: normal(true_distance, 0.3) @ "distance", clipped at 0 on the way out (core.py:386-395)."""
    from ..distributions import normal

    obs_dist = normal(true_distance, 0.3) @ "distance"
    return torch.clamp(obs_dist, min=0.0)


def _localization_model(prev_pose, time_index, world, n_rays=Const(8)):
    """The model as the reference writes it (core.py:312-383), over columns: pose sites x, y, theta around the
    initial position or the previous pose, clipped into the world; the lidar kernel gives the (n, R) true
    distances; a Vmap over the ray axis observes them.  Site order, and so the key schedule, is the fused
    kernels'."""
    from ..core import Vmap
    from ..core import gen as gen_
    from ..distributions import normal

    R = n_rays.value if isinstance(n_rays, Const) else int(n_rays)
    first = _is_initial(time_index)
    x = normal(1.5 if first else prev_pose.x, 0.5 if first else 0.25) @ "x"
    y = normal(1.5 if first else prev_pose.y, 0.5 if first else 0.25) @ "y"
    theta = normal(0.0 if first else prev_pose.theta, 0.3 if first else 0.1) @ "theta"
    x = torch.clamp(x, 0.1, world.width - 0.1)
    y = torch.clamp(y, 0.1, world.height - 0.1)
    pose = Pose(x, y, theta)
    true_distances = distance_to_wall_lidar(pose, world, n_angles=R)
    sensor = Vmap(gen_(_sensor_model_single_ray), in_axes=Const((0, 0)), axis_size=Const(R), axis_name=Const(None),
                  spmd_axis_name=Const(None))
    sensor(true_distances, torch.arange(R, device=true_distances.device)) @ "measurements"
    return pose, time_index + 1, world, n_rays


localization_model = LocalizationModel()
localization_model.generic = _gen(_localization_model)
sensor_model_single_ray = _gen(_sensor_model_single_ray)


class LocallyOptimalProposal:
    """create_locally_optimal_proposal(world, grid_size, noise_std, grid_radius)
    (examples/localization/core.py:730-809) as an `extension_proposal` of extend(): the grid
    evaluation, argmax, noisy draw and the importance weight run in ONE fused launch
    (gjx_loc_extend_locopt, grid_size^3 + 1 lidar evaluations per particle)."""

    def __init__(self, world, grid_size=15, noise_std=0.1, grid_radius=1.5):
        self.world, self.grid_size = world, int(grid_size)
        self.noise_std, self.grid_radius = float(noise_std), float(grid_radius)

    def fused_extend(self, family, trace, log_weights, args, constraints, offset=0):
        if not isinstance(args, LocRetval):
            raise GjxError("locally-optimal proposal: extend() must be fed the previous trace's return value")
        src = args.trace
        _, t, world, R = family._unpack(tuple(args))
        obs = family._obs_of(constraints, R)
        keys = family._keys(3)                            # proposal sites x, y, theta (core.py:804-806)
        out = B.loc_extend_locopt(world.native(R), keys, self.grid_size, self.noise_std, self.grid_radius, obs,
                                  (src.x, src.y, src.th), log_weights, offset=offset)
        tr = LocTrace(family, out["x"], out["y"], out["th"], out["mx"], out["my"], out["mth"], obs, t, world, R)
        return tr, out["logw"]


def create_locally_optimal_proposal(world, grid_size=15, noise_std=0.1, grid_radius=1.5):
    return LocallyOptimalProposal(world, grid_size, noise_std, grid_radius)


# --------------------------------------------------------------------------- #
# the case study's drivers (examples/localization/core.py:560-1002): one rejuvenation_smc call each, every
# timestep kept, returned as (particle_history, weight_history, diagnostic_weights)
# --------------------------------------------------------------------------- #
CATEGORICAL_RESAMPLING_MAX = 1 << 16       # the reference's resample() is an n x n Gumbel matrix


def _run_smc(n_particles, observations, world, key, n_rays, kernel=None, proposal=None, K=None,
             resample_method=None):
    """particle_history[t] is ONE Pose whose x / y / theta are the (n_particles,) columns of step t (the
    reference builds n_particles Pose objects per step in Python, core.py:592-600; at 2^24 particles that list
    would not fit the host); weight_history[t] = exp(logw) / sum(exp(logw)) (core.py:602-603).
    resample_method: the reference's categorical resampling up to 65536 particles, systematic beyond."""
    from ..core import const, seed
    from ..inference.smc import rejuvenation_smc
    from .localization_export import collection_history

    n = int(n_particles)
    method = resample_method or ("categorical" if n <= CATEGORICAL_RESAMPLING_MAX else "systematic")
    obs = observations if isinstance(observations, torch.Tensor) else np.asarray(
        [o.detach().cpu().numpy() if isinstance(o, torch.Tensor) else o for o in observations], np.float32)
    particles = seed(rejuvenation_smc)(
        key, localization_model, transition_proposal=proposal,
        mcmc_kernel=None if kernel is None else const(kernel),
        observations={"measurements": {"distance": obs}},
        initial_model_args=(Pose(0.0, 0.0, 0.0), 0, world, const(int(n_rays))),
        n_particles=const(n), return_all_particles=const(True),
        n_rejuvenation_moves=const(1) if K is None else (K if isinstance(K, Const) else const(int(K))),
        resample_method=const(method))
    return collection_history(particles)


def run_smc_basic(n_particles, observations, world, key, n_rays=8, resample_method=None):
    """Bootstrap filter: extend, resample when ESS < n / 2, no rejuvenation (core.py:560-607)."""
    return _run_smc(n_particles, observations, world, key, n_rays, resample_method=resample_method)


def _pose_selection():
    from ..core import sel

    return sel("x") | sel("y") | sel("theta")


def run_smc_with_mh(n_particles, observations, world, key, n_rays=8, K=None, resample_method=None):
    """K Metropolis-Hastings moves over (x, y, theta) per step, K = const(10) by default (core.py:610-667)."""
    from ..core import const
    from ..inference.mcmc import mh

    return _run_smc(n_particles, observations, world, key, n_rays, kernel=lambda tr: mh(tr, _pose_selection()),
                    K=const(10) if K is None else K, resample_method=resample_method)


def run_smc_with_hmc(n_particles, observations, world, key, n_rays=8, K=None, resample_method=None):
    """K HMC moves (step 0.01, 10 leapfrog steps) over (x, y, theta) per step (core.py:670-727)."""
    from ..core import const
    from ..inference.mcmc import hmc

    return _run_smc(n_particles, observations, world, key, n_rays,
                    kernel=lambda tr: hmc(tr, _pose_selection(), step_size=0.01, n_steps=10),
                    K=const(10) if K is None else K, resample_method=resample_method)


def run_smc_with_locally_optimal(n_particles, observations, world, key, n_rays=8, K=None, resample_method=None):
    """Locally optimal transition proposal: 25^3 grid, radius 2.0, noise 0.15 (core.py:812-868)."""
    from ..core import const

    return _run_smc(n_particles, observations, world, key, n_rays,
                    proposal=create_locally_optimal_proposal(world, grid_size=25, noise_std=0.15, grid_radius=2.0),
                    K=const(10) if K is None else K, resample_method=resample_method)


def run_smc_with_locally_optimal_big_grid(n_particles, observations, world, key, n_rays=8, K=None,
                                          resample_method=None):
    """The wider search: 25^3 grid, radius 3.0, noise 0.1 (core.py:871-927)."""
    from ..core import const

    return _run_smc(n_particles, observations, world, key, n_rays,
                    proposal=create_locally_optimal_proposal(world, grid_size=25, noise_std=0.1, grid_radius=3.0),
                    K=const(10) if K is None else K, resample_method=resample_method)


def _wall_clock(run, warmup_runs, repeats, inner_repeats, trim=0.2):
    """(mean, std) in seconds over `repeats` measurements, each the trimmed mean of `inner_repeats` device-
    synchronised calls, after `warmup_runs` untimed calls -- the protocol the case study times its methods with
    (methodology of src/genjax/timing.py:37-137; the code is not mirrored)."""
    import time

    def one_call():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    for _ in range(warmup_runs):
        one_call()
    table = np.sort(np.array([[one_call() for _ in range(inner_repeats)] for _ in range(repeats)], np.float64), axis=1)
    cut = int(inner_repeats * trim) if inner_repeats > 2 else 0
    per_repeat = table[:, cut:inner_repeats - cut].mean(axis=1)
    return float(per_repeat.mean()), float(per_repeat.std())


def benchmark_smc_methods(n_particles, observations, world, key, n_rays=8, repeats=5, K=20, K_hmc=25,
                          n_particles_big_grid=5):
    """Time the four methods with the case studies' protocol and keep one run of each for the quality plots
    (core.py:930-1002): {method: {timing_stats, particle_history, weight_history, diagnostic_weights,
    n_particles}}."""
    from ..core import const

    methods = {
        "smc_basic": (run_smc_basic, n_particles, None),
        "smc_hmc": (run_smc_with_hmc, n_particles, K_hmc),
        "smc_locally_optimal": (run_smc_with_locally_optimal, n_particles, K),
        "smc_locally_optimal_big_grid": (run_smc_with_locally_optimal_big_grid, n_particles_big_grid, K),
    }
    results = {}
    for name, (fn, n, moves) in methods.items():
        print(f"Benchmarking {name}...")

        def run_method(fn=fn, n=n, moves=moves):
            if moves is None:
                return fn(n, observations, world, key, n_rays)
            return fn(n, observations, world, key, n_rays, const(moves))

        mean_time, std_time = _wall_clock(run_method, warmup_runs=3, repeats=repeats, inner_repeats=3)
        history, weights, diag = run_method()
        results[name] = {"timing_stats": (mean_time, std_time), "particle_history": history,
                         "weight_history": weights, "diagnostic_weights": diag, "n_particles": n}
        print(f"  {name}: {mean_time:.3f} ± {std_time:.3f} seconds")
    return results
