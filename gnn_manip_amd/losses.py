"""Planner loss on the device: the ``geomloss.SamplesLoss(loss="sinkhorn", p=2, blur=.05)`` the reference builds at
gnn_manip/utils/traj_utils.py:69 and calls at :279.  geomloss is an un-vendored pip dependency of the reference
(environment.yml:25, version not pinned): csrc/sinkhorn.hip restates its published algorithm; see the header there.
"""
import ctypes as C

import torch

from ._lib import check, current_stream, lib, ptr
from .graph import _need_cuda, _ws


class SamplesLoss:
    """Callable with geomloss' constructor keywords; only what the reference uses is served (loudly otherwise).

    ``loss(x, y)`` is the reference's call (traj_utils.py:279): one pair, a 0-dim tensor.  ``loss.batched(X, y)`` takes the
    final clouds of a whole block of candidates, X [B, N, 3], against the desired cloud y [M, 3] (or one per candidate,
    [B, M, 3]) in ONE launch sequence and returns the B losses as a device tensor -- element b bit-equal to ``loss(X[b], y)``.
    ``diameter`` (geomloss keyword, default None = bounding box of each pair): given, no host synchronisation happens at all;
    otherwise one per call (the launch count is the longest epsilon schedule of the batch)."""

    def __init__(self, loss="sinkhorn", p=2, blur=0.05, scaling=0.5, debias=True, diameter=None, **unsupported):
        if loss != "sinkhorn" or p != 2 or not debias or unsupported:
            raise NotImplementedError("SamplesLoss: only loss='sinkhorn', p=2, debias=True (the reference's configuration, "
                                      "traj_utils.py:69) runs on the HIP device")
        self.blur, self.scaling = float(blur), float(scaling)
        self.diameter = 0.0 if diameter is None else float(diameter)
        self._ws = None

    def batched(self, x, y):
        """x [B, N, 3], y [M, 3] or [B, M, 3] float32 CUDA tensors with uniform weights -> [B] float32 tensor on the device."""
        _need_cuda(x, "x")
        _need_cuda(y, "y")
        x = x.contiguous().float()
        y = y.contiguous().float()
        if x.dim() != 3 or x.shape[2] != 3 or y.shape[-1] != 3 or y.dim() not in (2, 3) or (y.dim() == 3 and y.shape[0] != x.shape[0]):
            raise ValueError("SamplesLoss.batched: point clouds must be [B, N, 3] and [M, 3] or [B, M, 3]")
        bsz, n, m = x.shape[0], x.shape[1], y.shape[-2]
        L = lib()
        need = L.gm_sinkhorn_batched_workspace_bytes(bsz, n, m)
        if self._ws is None or self._ws.numel() < need or self._ws.device != x.device:
            self._ws = _ws(need, x.device)
        out = torch.empty((bsz,), dtype=torch.float32, device=x.device)
        check(L.gm_sinkhorn_divergence_batched(ptr(x), bsz, n, ptr(y), m, 1 if y.dim() == 2 else 0, self.blur, self.scaling,
                                               self.diameter, ptr(out), ptr(self._ws), self._ws.numel(), current_stream()))
        return out

    def __call__(self, x, y):
        """x [N, 3], y [M, 3] float32 CUDA tensors with uniform weights -> 0-dim float32 tensor on the device."""
        _need_cuda(x, "x")
        _need_cuda(y, "y")
        if x.dim() != 2 or y.dim() != 2 or x.shape[1] != 3 or y.shape[1] != 3:
            raise ValueError("SamplesLoss: point clouds must be [N, 3] and [M, 3]")
        return self.batched(x.unsqueeze(0), y)[0]
