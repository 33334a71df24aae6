"""Device-resident rollout: host-side mirror of the reference's rollout loop
(``compute_rollout`` gnn_manip/utils/rollout_utils.py:14-67 == ``cma_objective``'s loop
gnn_manip/utils/traj_utils.py:119-152) and of ``get_position_from_prediction``
(rollout_utils.py:145-158).

One step = state_pre -> node features -> radius graph -> destination sort -> edge features ->
encode/process/decode -> Euler integration -> state_post, all enqueued on the current HIP stream
by ``gm_rollout_step`` with no host synchronisation and no PCIe traffic.
"""
import ctypes as C
import os

import torch

from ._lib import ModelDesc, check, current_stream, lib, ptr
from .graph import _need_cuda, _ws, make_feature_desc


def get_position_from_prediction(stats, cartesian_idx, pred_acc, obs_seq, _desc=None):
    """Reference ``get_position_from_prediction`` (rollout_utils.py:145-158) on the device."""
    _need_cuda(pred_acc, "pred_acc")
    obs = obs_seq.contiguous().float()
    pred = pred_acc.contiguous().float()
    k, n, dd = obs.shape
    if _desc is None:
        one = [1.0, 1.0, 1.0]
        full = dict(velocity_mean=[0.0] * 3, velocity_std=one)
        full.update(stats)
        _desc = make_feature_desc(1.0, full, dict(lower_bounds=[0.0] * 3, upper_bounds=one), cartesian_idx,
                                  [0], None, k, dd)
    out = torch.empty((n, 3), dtype=torch.float32, device=obs.device)
    check(lib().gm_integrate(ptr(pred), ptr(obs), n, C.byref(_desc), ptr(out), current_stream()))
    return out


class RolloutEngine:
    """Runs rollouts of an ``EncProcDecGNN`` for scenes of ``n_nodes`` particles.

    graph_attr: a ``GraphBoundedMultimaterial(Control)`` (gnn_manip_amd.graph) carrying conn_r,
    stats, bounds and the column indices, exactly like ``dataset.graph_attr`` in the reference.
    """

    # run() renumbers the particles of scenes at least this large in grid-cell order, again every RENUMBER_EVERY steps
    RENUMBER_MIN_NODES = 20000
    RENUMBER_EVERY = 64

    def __init__(self, model, graph_attr, n_nodes, k_steps=6, data_dim=None, max_neighbours=20, device="cuda:0",
                 candidates=1, renumber="auto"):
        """candidates > 1: the engine steps that many equal-sized scenes at once, stored back to back along the
        node axis ([k, candidates*n_nodes, D]); the radius graph never links two scenes (block-diagonal batch,
        the offset rule of collate_utils.py:76), everything else is per node / per edge.

        renumber: True / False / "auto" (scenes of RENUMBER_MIN_NODES particles or more).  ``run`` then asks the library
        (gm_rollout, renumber_every = RENUMBER_EVERY) to work on a copy of the state whose rows are in grid-cell order (the
        radius graph's own grid, x fastest; particles of a cell in index order: the same every time), so that the per-edge
        gathers of neighbouring rows find each other in cache, re-ordered every RENUMBER_EVERY steps (particles move a
        fraction of a cell per step), and to write the result back in the caller's numbering.  A radius graph does not depend on the numbering
        (neighbours are ranked by distance; only an exact tie in distance falls back on the index) and every per-node /
        per-edge function is numbering-free, so what changes is the order in which a node's incoming messages are
        summed: float32 rounding, far inside the 1e-5 parity bound."""
        self.model = model
        self.graph_attr = graph_attr
        self.candidates = int(candidates)
        self.n_per = int(n_nodes)
        self.n = int(n_nodes) * self.candidates
        self.k = int(k_steps)
        self.device = torch.device(device)
        self.max_neighbours = int(max_neighbours)
        if data_dim is None:
            data_dim = (graph_attr.control_idx[-1] + 1) if graph_attr.control_idx is not None else graph_attr.cartesian_idx[-1] + 1
        self.data_dim = int(data_dim)
        self.fdesc = make_feature_desc(graph_attr.conn_r, graph_attr.stats, graph_attr.bounds, graph_attr.cartesian_idx,
                                       graph_attr.material_idx, graph_attr.control_idx, self.k, self.data_dim)
        self.fdesc.nodes_per_graph = self.n_per if self.candidates > 1 else 0
        self.mdesc = ModelDesc(*model.model_desc())
        L = lib()
        self.ws = _ws(L.gm_rollout_workspace_bytes(C.byref(self.mdesc), self.n, self.max_neighbours), self.device)
        self.rigid_rank = None
        self.n_rigid = 0
        if renumber == "auto" and os.environ.get("GM_RENUMBER") in ("0", "1"):   # A/B runs of the benchmark
            renumber = os.environ["GM_RENUMBER"] == "1"
        self.renumber = (self.n_per >= self.RENUMBER_MIN_NODES) if renumber == "auto" else bool(renumber)
        self._renumber_ws = None

    def _rank_rigid(self, obs):
        rank = torch.empty(self.n, dtype=torch.int32, device=self.device)
        cnt = torch.zeros(1, dtype=torch.int32, device=self.device)
        check(lib().gm_rigid_rank(ptr(obs), self.n, C.byref(self.fdesc), ptr(rank), ptr(cnt), current_stream()))
        return rank, cnt

    def set_scene(self, obs):
        """Classify rigid rows (material == 1, rollout_utils.py:20) once per scene."""
        _need_cuda(obs, "obs")
        assert obs.shape == (self.k, self.n, self.data_dim) and obs.dtype == torch.float32 and obs.is_contiguous()
        self.rigid_rank, cnt = self._rank_rigid(obs)
        self.n_rigid = int(cnt.item())
        return self.n_rigid

    def _check_state(self, obs, rigid_target, pred_out, use_rigid):
        """The C entry takes raw pointers: shapes, dtypes and devices are checked here."""
        if not (isinstance(obs, torch.Tensor) and obs.is_cuda and obs.device == self.device):
            raise ValueError(f"obs must be a tensor on {self.device}")
        if tuple(obs.shape) != (self.k, self.n, self.data_dim) or obs.dtype != torch.float32 or not obs.is_contiguous():
            raise ValueError(f"obs must be contiguous float32 [{self.k}, {self.n}, {self.data_dim}], got {tuple(obs.shape)} {obs.dtype}")
        if use_rigid and self.rigid_rank is None:
            raise RuntimeError("RolloutEngine.set_scene(obs) must be called before step()")
        if rigid_target is not None:
            if not use_rigid:
                raise ValueError("rigid_target needs use_rigid=True")
            if (rigid_target.device != self.device or rigid_target.dtype != torch.float32 or not rigid_target.is_contiguous()
                    or tuple(rigid_target.shape) != (self.n_rigid, 3)):
                raise ValueError(f"rigid_target must be contiguous float32 [{self.n_rigid}, 3] on {self.device}, got {tuple(rigid_target.shape)}")
        if pred_out is not None:
            if (pred_out.device != self.device or pred_out.dtype != torch.float32 or not pred_out.is_contiguous()
                    or pred_out.numel() != self.n * 3):
                raise ValueError(f"pred_out must be contiguous float32 with {self.n * 3} elements on {self.device}")

    def step(self, obs, rigid_target=None, pred_out=None, use_rigid=True):
        """One rollout step in place on ``obs`` [k, N, D]; rigid_target: [N_rigid, 3] scripted pose or None."""
        self._check_state(obs, rigid_target, pred_out, use_rigid)
        handle = self.model.device_handle(self.device)
        rr = self.rigid_rank if use_rigid else None
        check(lib().gm_rollout_step(handle, ptr(obs), self.n, C.byref(self.fdesc), self.max_neighbours, ptr(rr),
                                    ptr(rigid_target), ptr(pred_out), ptr(self.ws), self.ws.numel(), current_stream()))

    def status(self):
        """Synchronises; raises on a device-side data error; returns the last step's edge count."""
        e = C.c_int64(0)
        check(lib().gm_rollout_status(ptr(self.ws), C.byref(self.mdesc), self.n, self.max_neighbours, C.byref(e), current_stream()))
        return int(e.value)

    def run(self, obs, trajectory=None, steps=None, record=False):
        """`steps` rollout steps in place on ``obs`` inside ONE library call (gm_rollout): no Python between steps.
        trajectory: [T, N_rigid, 3] contiguous device tensor of scripted poses or None.  Returns the recorded last
        frames [steps, N, D] when asked (the reference's per-step record), else None."""
        T = 0 if trajectory is None else int(trajectory.shape[0])
        steps = T if steps is None else int(steps)
        self._check_state(obs, None, None, True)
        if trajectory is not None:
            if (trajectory.device != self.device or trajectory.dtype != torch.float32 or not trajectory.is_contiguous()
                    or tuple(trajectory.shape[1:]) != (self.n_rigid, 3)):
                raise ValueError(f"trajectory must be contiguous float32 [T, {self.n_rigid}, 3] on {self.device}, got {tuple(trajectory.shape)}")
        recs = torch.empty((steps, self.n, self.data_dim), dtype=torch.float32, device=self.device) if record else None
        handle = self.model.device_handle(self.device)  # resolved once per rollout
        L = lib()
        every, rws = 0, None
        if self.renumber and self.n > 0 and steps > 0:
            # rows in grid-cell order inside the library (gm_rollout, renumber_every): order, row maps, gathers and the write-back
            # are device kernels of the same call -- nothing is sorted or synchronised here
            every = self.RENUMBER_EVERY
            if self._renumber_ws is None:
                self._renumber_ws = _ws(L.gm_rollout_renumber_workspace_bytes(C.byref(self.fdesc), self.n), self.device)
            rws = self._renumber_ws
        check(L.gm_rollout(handle, ptr(obs), self.n, C.byref(self.fdesc), self.max_neighbours, ptr(self.rigid_rank),
                           ptr(trajectory), T, self.n_rigid, steps, ptr(recs), every, ptr(rws), 0 if rws is None else rws.numel(),
                           ptr(self.ws), self.ws.numel(), current_stream()))
        return recs

    def rollout_candidates(self, obs0, trajectories, horizon=None):
        """Roll `candidates` copies of one initial state [k, N, D] under per-candidate scripted rigid poses
        `trajectories` [B, T, N_rigid, 3] (the CMA-ES population of traj_utils.py:247-259, evaluated together
        instead of serially).  Returns the final states [B, k, N, D]."""
        b, k, n = self.candidates, self.k, self.n_per
        assert trajectories.shape[0] == b
        obs = obs0.unsqueeze(1).repeat(1, b, 1, 1).reshape(k, b * n, self.data_dim).contiguous()
        self.set_scene(obs)
        steps = horizon if horizon is not None else trajectories.shape[1]
        # [T, B * Nr, 3]: one step's poses of all candidates are contiguous, candidate-major like the rigid rows
        traj_t = trajectories.permute(1, 0, 2, 3).reshape(trajectories.shape[1], -1, 3).contiguous().float()
        self.run(obs, traj_t, steps)
        self.status()
        return obs.reshape(k, b, n, self.data_dim).permute(1, 0, 2, 3).contiguous()

    def rollout(self, obs0, trajectory=None, horizon=None, record=False):
        """cma_objective's loop (traj_utils.py:119-152).  trajectory: [T, N_rigid, 3] device tensor of scripted
        rigid poses (or None: no rigid overwrite).  Returns the final state (and the recorded last frames)."""
        obs = obs0.clone().contiguous()
        self.set_scene(obs)  # per call: the engine may be given another scene
        steps = horizon if horizon is not None else (trajectory.shape[0] if trajectory is not None else 0)
        traj = None if trajectory is None else trajectory.contiguous().float()
        recs = self.run(obs, traj, steps, record=record)
        self.status()
        if record:
            return obs, recs
        return obs


def get_rigid_body_trajectory_from_diff(trajectory, horizon, ty_init, rigid_particles):
    """rollout_utils.py:160-174: absolute (rotation [rad], translation) pairs -> [horizon, N_rigid, 3] poses on the device."""
    from .planner import get_rigid_body_trajectory
    return get_rigid_body_trajectory(trajectory[:, 0], trajectory[:, 1], horizon, ty_init, rigid_particles)


def extract_groundtruth(dataset, nof_steps, nof_particles=None, data_dim=None):
    """rollout_utils.py:84-93: last frame of every window, [nof_steps, N, D] (device)."""
    return torch.stack([dataset[i][0][-1] for i in range(nof_steps)]).float()


def compute_rollout(dataset, model, args):
    """Mirror of ``compute_rollout`` (rollout_utils.py:12-67): roll the model out from the first window of a
    ``CoffeeTestDataset``, driving the rigid body either with a planned trajectory (``args.cma_traj``: .npy of
    absolute [rotation, translation] per step) or with the recorded ground truth.  Returns the reference's
    ``prediction`` array [nof_steps, N, D] (numpy): the last frame of every step after the control overwrite.
    The step itself (graph, features, model, integration, window shift) is gm_rollout_step on the device; only the
    ground-truth mode's control / pose overwrite -- taken verbatim from recorded frames -- is done with tensor
    indexing around it."""
    import numpy as np
    obs0, _ = dataset[0]
    k, n, dd = obs0.shape
    dev = obs0.device
    nof_steps = dataset.time_steps if args.cma_traj is not None else dataset.time_steps - args.k_steps
    eng = RolloutEngine(model, dataset.graph_attr, n, k_steps=k, data_dim=dd, max_neighbours=20, device=dev)
    obs = obs0.clone().contiguous()
    eng.set_scene(obs)
    rigid = obs[0, :, dataset.material_id] == 1
    c0, u0 = dataset.cartesian_idx[0], dataset.control_idx[0]
    with torch.no_grad():
        if args.cma_traj is not None:
            npy_trajectory = np.load(args.cma_traj) if isinstance(args.cma_traj, str) else np.asarray(args.cma_traj)
            rigid_pos = obs[-1, rigid, c0:c0 + 3].contiguous()
            trajectory = get_rigid_body_trajectory_from_diff(npy_trajectory, nof_steps, [0.5, 0.5, 0.4], rigid_pos)
            _, recs = eng.rollout(obs, trajectory, horizon=nof_steps, record=True)
            return recs.cpu().numpy().astype(np.float64)
        groundtruth = extract_groundtruth(dataset, nof_steps)
        prediction = []
        for i in range(nof_steps):
            obs[-1, rigid, u0:u0 + 3] = groundtruth[i, rigid, u0:u0 + 3]          # control from the recording (:43)
            prediction.append(obs[-1].clone())
            eng.step(obs, None, use_rigid=False)                                  # predict, integrate, shift
            new_rigid = prediction[-1][rigid].clone()                             # rigid rows keep the frame's attributes ...
            new_rigid[:, c0:c0 + 3] = groundtruth[i, rigid, c0:c0 + 3]            # ... with the recorded pose (:57)
            obs[-1, rigid] = new_rigid
        eng.status()
        return torch.stack(prediction).cpu().numpy().astype(np.float64)
