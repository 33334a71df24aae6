"""Planner loss on the device: the ``geomloss.SamplesLoss(loss="sinkhorn", p=2, blur=.05)`` the reference builds at
gnn_manip/utils/traj_utils.py:69 and calls at :279.  geomloss is an un-vendored pip dependency of the reference
(environment.yml:25, version not pinned): csrc/sinkhorn.hip restates its published algorithm; see the header there.
"""
import ctypes as C

import torch

from ._lib import check, current_stream, lib, ptr
from .graph import _need_cuda, _ws


class SamplesLoss:
    """Callable with geomloss' constructor keywords; only what the reference uses is served (loudly otherwise)."""

    def __init__(self, loss="sinkhorn", p=2, blur=0.05, scaling=0.5, debias=True, **unsupported):
        if loss != "sinkhorn" or p != 2 or not debias or unsupported:
            raise NotImplementedError("SamplesLoss: only loss='sinkhorn', p=2, debias=True (the reference's configuration, "
                                      "traj_utils.py:69) runs on the HIP device")
        self.blur, self.scaling = float(blur), float(scaling)
        self._ws = None

    def __call__(self, x, y):
        """x [N, 3], y [M, 3] float32 CUDA tensors with uniform weights -> 0-dim float32 tensor on the device."""
        _need_cuda(x, "x")
        _need_cuda(y, "y")
        x = x.contiguous().float()
        y = y.contiguous().float()
        if x.dim() != 2 or y.dim() != 2 or x.shape[1] != 3 or y.shape[1] != 3:
            raise ValueError("SamplesLoss: point clouds must be [N, 3] and [M, 3]")
        L = lib()
        need = L.gm_sinkhorn_workspace_bytes(x.shape[0], y.shape[0])
        if self._ws is None or self._ws.numel() < need or self._ws.device != x.device:
            self._ws = _ws(need, x.device)
        out = torch.empty((), dtype=torch.float32, device=x.device)
        check(L.gm_sinkhorn_divergence(ptr(x), x.shape[0], ptr(y), y.shape[0], self.blur, self.scaling, ptr(out), ptr(self._ws),
                                       self._ws.numel(), current_stream()))
        return out
