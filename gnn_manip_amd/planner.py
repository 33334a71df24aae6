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

    def _dist(self):
        import torch.distributed as dist
        return dist if (dist.is_available() and dist.is_initialized()) else None

    def evaluate(self, population):
        dist = self._dist()
        world = dist.get_world_size(self.group) if dist else 1
        rank = dist.get_rank(self.group) if dist else 0
        dev = self.device if self.device is not None else torch.device("cpu")
        if dist:
            shape = torch.zeros(2, dtype=torch.int64, device=dev)
            if rank == 0:
                pop = torch.as_tensor(np.asarray(population, dtype=np.float64), device=dev)
                shape[0], shape[1] = pop.shape
            dist.broadcast(shape, src=0, group=self.group)
            if rank != 0:
                pop = torch.empty((int(shape[0]), int(shape[1])), dtype=torch.float64, device=dev)
            dist.broadcast(pop, src=0, group=self.group)
            pop_np = pop.cpu().numpy()
        else:
            pop_np = np.asarray(population, dtype=np.float64)
        n = pop_np.shape[0]
        per = -(-n // world)  # padded block so that all_gather sees equal shapes
        lo, hi = shard_range(n, world, rank)
        local = torch.zeros((per, self.result_dim), dtype=torch.float64, device=dev)
        for j, c in enumerate(range(lo, hi)):
            r = self.fn(pop_np[c])
            local[j] = torch.as_tensor(r, dtype=torch.float64).reshape(-1).to(dev)
        if not dist:
            return local[:n].cpu().numpy()
        gathered = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(gathered, local, group=self.group)
        out = np.zeros((n, self.result_dim))
        for r in range(world):
            a, b = shard_range(n, world, r)
            out[a:b] = gathered[r][:b - a].cpu().numpy()
        return out
