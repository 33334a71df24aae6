"""Planner-side pieces of the hot path: the scripted cup pose (``compute_particles_tmatrix``,
``interpolate_trajectory``, ``get_rigid_body_trajectory_from_diff``; gnn_manip/utils/traj_utils.py:87-103,
167-228) and candidate-parallel evaluation of CMA-ES populations across the GPUs of a node
(SURVEY.md section 8e): candidates are independent rollouts (traj_utils.py:114-159), so each rank
evaluates a contiguous block; one broadcast of the candidate matrix and one all-gather of the
per-candidate results per generation (RCCL over xGMI when the backend is ``nccl``).
"""
import ctypes as C

import numpy as np
import torch

from ._lib import check, current_stream, lib, ptr


def interpolate_trajectory(x, n_points, rx_init, scale_rot, scale_ty, max_rot, max_ty):
    """TrajectoryCMAsolver.interpolate_trajectory (traj_utils.py:206-228); host float64 like the reference.
    rx_init / max_rot in radians."""
    prev_r, prev_t = rx_init, 0.0
    rot, ty = [rx_init], [0.0]
    for i in range(n_points):
        inc_r = np.clip(np.deg2rad(scale_rot * np.rad2deg(x[i])), -max_rot, max_rot)
        inc_t = np.clip(scale_ty * x[i + n_points], -max_ty, max_ty)
        prev_r = prev_r + inc_r
        prev_t = prev_t + inc_t
        rot.append(prev_r)
        ty.append(prev_t)
    return rot, ty


def compute_particles_tmatrix(rotation, translation, ty_init, rigid_particles):
    """traj_utils.py:167-194 for ONE pose; rigid_particles: [Nr, 3] CUDA tensor."""
    return get_rigid_body_trajectory([rotation], [translation], 1, ty_init, rigid_particles)[0]


def get_rigid_body_trajectory(traj_rot, traj_ty, horizon, ty_init, rigid_particles):
    """Body of get_rigid_body_trajectory_from_diff (traj_utils.py:97-99): [horizon, Nr, 3] on the device.
    cos / sin are evaluated on the host in float64 and rounded to float32, as the reference does when it
    builds the 4x4 float32 matrix (traj_utils.py:171-172)."""
    rp = rigid_particles.contiguous().float()
    assert rp.is_cuda
    if len(traj_rot) < horizon or len(traj_ty) < horizon:
        # the reference indexes traj[i] for i < horizon (traj_utils.py:97-99) and raises IndexError on a short trajectory
        raise IndexError(f"trajectory of {min(len(traj_rot), len(traj_ty))} poses is shorter than the horizon {horizon}")
    rot = np.asarray(traj_rot[:horizon], dtype=np.float64)
    ty = np.asarray(traj_ty[:horizon], dtype=np.float64)
    cst = np.stack((np.cos(rot), np.sin(rot), float(ty_init[1]) + ty), axis=1).astype(np.float32)
    cst_d = torch.from_numpy(cst).to(rp.device)
    out = torch.empty((horizon, rp.shape[0], 3), dtype=torch.float32, device=rp.device)
    t3 = (C.c_float * 3)(*[float(v) for v in ty_init])
    check(lib().gm_rigid_transform(ptr(rp), rp.shape[0], ptr(cst_d), horizon, C.byref(t3), ptr(out), current_stream()))
    return out


def shard_range(n_items, world_size, rank):
    """Contiguous block [lo, hi) of `n_items` candidates owned by `rank` (blocks differ by at most one)."""
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class CandidateEvaluator:
    """Evaluates a population of candidates in parallel over the ranks of a torch.distributed group.

    evaluate_fn(candidate: np.ndarray) -> 1-D float tensor/array of fixed length `result_dim`
    (e.g. a loss, or final particle statistics).  Rank 0 supplies the population; every rank
    returns the full [popsize, result_dim] result matrix.
    """

    def __init__(self, evaluate_fn, result_dim=1, group=None, device=None):
        self.fn = evaluate_fn
        self.result_dim = int(result_dim)
        self.group = group
        self.device = device
        self.collective_s = 0.0   # wall time spent in the broadcast / all-gather calls so far (host clock around each call)

    def _dist(self):
        import torch.distributed as dist
        return dist if (dist.is_available() and dist.is_initialized()) else None

    def evaluate_blocks(self, population, block_fn):
        """Like evaluate(), but the rank's whole block goes to block_fn(list of candidates) -> list of results at once
        (batched rollouts on the device)."""
        self._block_fn = block_fn
        try:
            return self.evaluate(population)
        finally:
            self._block_fn = None

    _block_fn = None

    def evaluate(self, population):
        dist = self._dist()
        world = dist.get_world_size(self.group) if dist else 1
        rank = dist.get_rank(self.group) if dist else 0
        dev = self.device if self.device is not None else torch.device("cpu")
        import time
        if dist:
            t_c = time.perf_counter()
            shape = torch.zeros(2, dtype=torch.int64, device=dev)
            if rank == 0:
                pop = torch.as_tensor(np.asarray(population, dtype=np.float64), device=dev)
                shape[0], shape[1] = pop.shape
            dist.broadcast(shape, src=0, group=self.group)
            if rank != 0:
                pop = torch.empty((int(shape[0]), int(shape[1])), dtype=torch.float64, device=dev)
            dist.broadcast(pop, src=0, group=self.group)
            pop_np = pop.cpu().numpy()
            self.collective_s += time.perf_counter() - t_c
        else:
            pop_np = np.asarray(population, dtype=np.float64)
        n = pop_np.shape[0]
        per = -(-n // world)  # padded block so that all_gather sees equal shapes
        lo, hi = shard_range(n, world, rank)
        local = torch.zeros((per, self.result_dim), dtype=torch.float64, device=dev)
        if self._block_fn is not None:
            res = self._block_fn([pop_np[c] for c in range(lo, hi)]) if hi > lo else []
            for j, r in enumerate(res):
                local[j] = torch.as_tensor(r, dtype=torch.float64).reshape(-1).to(dev)
        else:
            for j, c in enumerate(range(lo, hi)):
                r = self.fn(pop_np[c])
                local[j] = torch.as_tensor(r, dtype=torch.float64).reshape(-1).to(dev)
        if not dist:
            return local[:n].cpu().numpy()
        t_c = time.perf_counter()   # includes the wait for the slowest rank: load imbalance shows up here
        gathered = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(gathered, local, group=self.group)
        out = np.zeros((n, self.result_dim))
        for r in range(world):
            a, b = shard_range(n, world, r)
            out[a:b] = gathered[r][:b - a].cpu().numpy()
        self.collective_s += time.perf_counter() - t_c
        return out


class TrajectoryCMAsolver:
    """Mirror of the reference's ``TrajectoryCMAsolver`` (traj_utils.py:14-76,197-285) with the population of a
    CMA-ES generation evaluated TOGETHER on the device(s) instead of one ``cma_objective`` call per candidate:

    * ``cma_objective(x)``: one candidate, the reference's loop (rollout on the device, no PCIe per step);
    * ``population_losses(X)``: the generation is cut into blocks of ``candidates_per_gpu`` block-diagonal
      batched rollouts (``RolloutEngine(candidates=B)``) and, under ``torch.distributed``, sharded over the ranks
      (``CandidateEvaluator``: one broadcast of X, one all-gather of the losses per generation);
    * ``optimize_trajectory(desired_position)``: ``cmaes.fmin2`` over that (the reference: ``cma.fmin2``, :257).

    Same constructor arguments as the reference (``state_init`` = (obs [k, N, D], next positions) on the device).
    """

    def __init__(self, model, graph_attr, state_init, rx_init, ty_init, scale_rot, scale_ty, alpha, beta, gamma, penalty,
                 rho, device, cma_iter=10, cma_var=0.5, cma_popsize=21, cma_rand=1234, max_rot=1.9337, max_ty=6.67e-4,
                 total_steps=300, traj_points=10, candidates_per_gpu=8):
        from . import cmaes
        from .losses import SamplesLoss
        from .rollout import RolloutEngine
        self.initial_state = state_init
        self.rx_init = np.deg2rad(rx_init)
        self.ty_init = ty_init
        self.sample_traj = None
        self.desired_pos = None
        self.model, self.graph_attr = model, graph_attr
        self.horizon = total_steps
        self.device = torch.device(device)
        self.collective_device = None   # where the broadcast / all-gather payloads live; None: self.device (RCCL).  "cpu" for gloo
        self.scale_rot, self.scale_ty = scale_rot, scale_ty
        self.nr_traj_points = traj_points
        self.traj_points = int(self.horizon / self.nr_traj_points)
        self.alpha, self.beta, self.gamma, self.penalty, self.rho = alpha, beta, gamma, penalty, rho
        self.max_rot = np.deg2rad(max_rot)
        self.max_ty = max_ty
        self.total_steps = total_steps
        # limits of the boundary penalties (traj_utils.py:57-61)
        self.left_limit, self.right_limit = 0.3, 0.7
        self.scale_ty = (self.ty_init[0] - self.left_limit) / self.scale_rot
        self.rotation_limit = 2.8973
        obs = self.initial_state[0]
        mat = graph_attr.material_idx[0]
        c0 = graph_attr.cartesian_idx[0]
        self.rigid_particles_idx = obs[-1, :, mat] == 1
        self.coffee_particles_idx = obs[-1, :, mat] == 0
        self._coffee_rows = torch.nonzero(self.coffee_particles_idx).reshape(-1)   # taken once: a boolean-mask index synchronises per use
        self.rigid_particles = obs[-1, self.rigid_particles_idx, c0:c0 + 3].contiguous()
        self.loss = SamplesLoss(loss="sinkhorn", p=2, blur=.05)
        self.cma_options = cmaes.CMAOptions()
        self.cma_options['seed'] = cma_rand
        self.cma_options['maxiter'] = cma_iter
        self.cma_options['popsize'] = cma_popsize
        self.cma_initial_var = cma_var
        self.candidates_per_gpu = int(candidates_per_gpu)
        k, n, dd = obs.shape
        self._mk_engine = lambda b: RolloutEngine(model, graph_attr, n, k_steps=k, data_dim=dd, device=self.device, candidates=b)
        self._engines = {}

    # ---- trajectory parametrisation (traj_utils.py:199-228)
    def set_sample_traj(self, sample_traj):
        sample_traj = np.asarray(sample_traj)
        d = sample_traj[2:] - sample_traj[1:-1]
        self.sample_traj = np.stack((np.deg2rad(d[:, 0] / self.scale_rot), d[:, 1] / self.scale_ty)).T

    def interpolate_trajectory(self, x):
        return interpolate_trajectory(x, self.sample_traj.shape[0], self.rx_init, self.scale_rot, self.scale_ty, self.max_rot,
                                      self.max_ty)

    def get_rigid_body_trajectory_from_diff(self, x, demo=False):
        if demo:
            traj_rot, traj_ty = x[:, 0], x[:, 1]
        else:
            traj_rot, traj_ty = self.interpolate_trajectory(x)
        traj = get_rigid_body_trajectory(traj_rot, traj_ty, self.horizon, self.ty_init, self.rigid_particles)
        actions = np.zeros((self.horizon, 2))
        actions[:, 0] = traj_rot[:self.horizon]
        actions[:, 1] = traj_ty[:self.horizon]
        return traj, actions

    # ---- loss terms (traj_utils.py:161-165,230-285)
    @staticmethod
    def compute_vel_acc(actions):
        return actions[1:, :] - actions[:-1, :], actions[2:, :] - 2 * actions[1:-1, :] + actions[:-2, :]

    def compute_vel_loss(self, vel):
        return float(np.linalg.norm(vel / np.array([self.max_rot, self.max_ty])[None, :]))

    def compute_acc_loss(self, acc):
        return float(np.linalg.norm(acc / np.array([self.max_rot, self.max_ty])[None, :]))

    def compute_boundaries_penalty(self, actions):
        rot = actions[:, 0]
        if rot.max() > self.rx_init + self.rotation_limit or rot.min() < self.rx_init - self.rotation_limit:
            return 20.0
        return 0.0

    def compute_loss(self, end_position, actions, cup_states=None, coffee_states=None, x=None):
        """traj_utils.py:275-285 for one candidate (one device loss, one transfer)."""
        return self._assemble_loss(float(self.loss(end_position, self.desired_pos).item()), actions, x)

    def _assemble_loss(self, wasserstein_loss, actions, x=None):
        """Everything of compute_loss but the Wasserstein term itself: host float64 like the reference."""
        vel, acc = self.compute_vel_acc(actions)
        vel_loss, acc_loss = self.compute_vel_loss(vel), self.compute_acc_loss(acc)
        bound_penalty = self.compute_boundaries_penalty(actions)
        loss = self.beta * wasserstein_loss + self.penalty * bound_penalty + self.alpha * vel_loss + self.gamma * acc_loss
        return loss, wasserstein_loss, vel_loss, acc_loss, bound_penalty, 0.0

    # ---- objective
    def _engine(self, b):
        if b not in self._engines:
            self._engines[b] = self._mk_engine(b)
        return self._engines[b]

    def _end_positions(self, final_states):
        """final_states [B, k, N, D] -> the coffee particles' end positions [B, Nc, 3] (traj_utils.py:154-155)."""
        c0 = self.graph_attr.cartesian_idx[0]
        return final_states[:, -1].index_select(1, self._coffee_rows)[:, :, c0:c0 + 3].contiguous()

    def cma_objective(self, x):
        """traj_utils.py:114-159 for one candidate."""
        return self._local_losses([np.asarray(x, dtype=np.float64)])[0]

    def _block_ends(self, xs):
        """Block-diagonal batched rollout of the candidates xs: their end clouds [B, Nc, 3] (device) and action arrays."""
        trajs, acts = zip(*[self.get_rigid_body_trajectory_from_diff(x) for x in xs])
        eng = self._engine(len(xs))
        with torch.no_grad():
            finals = eng.rollout_candidates(self.initial_state[0].contiguous(), torch.stack(trajs), horizon=self.horizon)
        return self._end_positions(finals), list(acts)

    def population_losses(self, X):
        """Losses of a whole generation.  Under torch.distributed rank 0's X is broadcast and every rank evaluates
        a contiguous block; returns the full list on every rank."""
        import torch.distributed as dist
        X = [np.asarray(x, dtype=np.float64) for x in X]
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        if world > 1:
            ev = CandidateEvaluator(None, result_dim=1, device=self.collective_device or self.device)
            return ev.evaluate_blocks(X, self._local_losses).reshape(-1).tolist()
        return self._local_losses(X)

    _loss_takes_x = False   # InterpolatedCMAsolver.compute_loss reads the search variables too

    def _local_losses(self, X):
        """This rank's candidates: rollouts in blocks of `candidates_per_gpu`, then the Wasserstein terms of ALL of them in one
        batched launch sequence (SamplesLoss.batched) and ONE transfer of the values; the penalty terms on the host."""
        if not len(X):
            return []
        ends, acts = [], []
        for b in range(0, len(X), self.candidates_per_gpu):
            e, a = self._block_ends(X[b:b + self.candidates_per_gpu])
            ends.append(e)
            acts += a
        w = self.loss.batched(torch.cat(ends), self.desired_pos).double().cpu().numpy()
        return [self._assemble_loss(float(w[i]), acts[i], X[i] if self._loss_takes_x else None)[0] for i in range(len(X))]

    def _start_point(self):
        """The search starts at the sample trajectory: all rotation variables, then all translation variables."""
        return np.ascontiguousarray(self.sample_traj.T, dtype=np.float64).reshape(-1)

    def optimize_trajectory(self, desired_position):
        """traj_utils.py:247-259."""
        from . import cmaes
        initial_traj = self._start_point()
        self.desired_pos = desired_position.clone()
        return cmaes.fmin2(self.cma_objective, initial_traj.tolist(), self.cma_initial_var, options=self.cma_options,
                           parallel_objective=self.population_losses)


class InterpolatedCMAsolver(TrajectoryCMAsolver):
    """Mirror of the reference's ``InterpolatedCMAsolver`` (traj_utils.py:288-452): the search variables are key
    points (every ``traj_points``-th step) of the rotation / translation, PCHIP-interpolated to one pose per step;
    increments between key points are constrained (``ineq_constraint``, solved with ``cmaes.fmin_con``) and the
    rotation is box-bounded.  Objective evaluation (batched rollouts, device loss) is inherited."""

    def set_sample_traj(self, sample_traj):
        """traj_utils.py:296-304: the key points are every ``traj_points``-th pose after the first, as offsets from the
        initial pose in search-variable units (column 0 rotation [deg in the file], column 1 translation)."""
        keys = np.asarray(sample_traj, dtype=np.float64)[self.nr_traj_points::self.nr_traj_points]
        offset = np.array([self.rx_init, self.ty_init[0]])
        scale = np.array([self.scale_rot, self.scale_ty])
        self.sample_traj = (np.column_stack((np.deg2rad(keys[:, 0]), keys[:, 1])) - offset) / scale

    def interpolate_trajectory(self, x, type_interp='pchip'):
        from scipy.interpolate import interp1d, pchip_interpolate
        x = np.asarray(x, dtype=np.float64)
        k = self.sample_traj.shape[0]
        rot_points = [self.rx_init] + (self.rx_init + x[:k] * self.scale_rot).tolist()
        ty_points = [0.0] + (x[k:] * self.scale_ty).tolist()
        traj_idx = np.arange(0, self.horizon + 1, self.nr_traj_points)
        idx_interp = np.arange(self.horizon)
        if type_interp == 'cubic':
            return (interp1d(x=traj_idx, y=rot_points, kind='cubic')(idx_interp),
                    interp1d(x=traj_idx, y=ty_points, kind='cubic')(idx_interp))
        return pchip_interpolate(traj_idx, rot_points, idx_interp), pchip_interpolate(traj_idx, ty_points, idx_interp)

    def compute_acc_loss(self, acc):
        return float(np.linalg.norm(acc / np.array([2.2e-4, 1.45e-4])[None, :]))  # fixed means, traj_utils.py:343-353

    def compute_vel_loss(self, vel):
        return float(np.linalg.norm(vel / np.array([1e-2, 4e-4])[None, :]))        # traj_utils.py:355-364

    def compute_vel_noninterp(self, x):
        x = np.asarray(x, dtype=np.float64)
        rot, ty = x[:self.traj_points] * self.scale_rot, x[self.traj_points:] * self.scale_ty
        ineq_rot = np.abs(rot[1:] - rot[:-1]) - self.max_rot * self.nr_traj_points
        ineq_ty = np.abs(ty[1:] - ty[:-1]) - self.max_ty * self.nr_traj_points
        return float(np.exp(max(ineq_rot.max(), ineq_ty.max())))

    def ineq_constraint(self, x):
        x = np.asarray(x, dtype=np.float64)
        actions = np.zeros((self.traj_points + 1, 2))
        actions[1:, 0] = x[:self.traj_points] * self.scale_rot
        actions[1:, 1] = x[self.traj_points:] * self.scale_ty
        vel, _ = self.compute_vel_acc(actions)
        upper = np.abs(vel) - np.array([self.max_rot * self.nr_traj_points, self.max_ty * self.nr_traj_points])
        return np.concatenate((upper[:, 0] / self.scale_rot, upper[:, 1] / self.scale_ty))

    _loss_takes_x = True

    def _assemble_loss(self, wasserstein_loss, actions, x=None):
        vel, acc = self.compute_vel_acc(actions)
        vel_loss, acc_loss = self.compute_vel_loss(vel), self.compute_acc_loss(acc)
        interp_loss = self.compute_vel_noninterp(x) if x is not None else 0.0
        loss = self.beta * wasserstein_loss + self.alpha * vel_loss + self.gamma * acc_loss + self.rho * interp_loss
        return loss, wasserstein_loss, vel_loss, acc_loss, interp_loss, 0.0

    def optimize_trajectory(self, desired_position):
        """traj_utils.py:324-337."""
        from . import cmaes
        self.cma_options['bounds'] = [-self.rotation_limit / self.scale_rot, self.rotation_limit / self.scale_rot]
        initial_traj = self._start_point()
        self.desired_pos = desired_position.clone()
        return cmaes.fmin_con(self.cma_objective, initial_traj.tolist(), self.cma_initial_var, g=self.ineq_constraint,
                              options=self.cma_options, parallel_objective=self.population_losses)
