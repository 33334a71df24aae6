"""Graph construction and featurisation on the MI355X -- host-side mirror of the reference's
``gnn_manip/utils/utils.py`` and ``gnn_manip/utils/collate_utils.py`` call surface.

Same function / class names, argument meaning and return values as the reference; tensors
live on the GPU (``cuda``) and every function calls into libgnnmanip_hip.so.  CPU tensors are
rejected: there is no CPU fallback in the product path.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import FeatureDesc, check, current_stream, lib, ptr


def _need_cuda(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise RuntimeError(f"{name} must be a CUDA tensor: gnn_manip_amd runs on the HIP device only")


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def _pos_view(pos):
    """(pointer tensor, stride in floats) for an [N, 3] float32 view whose rows may be strided."""
    if pos.dtype != torch.float32 or pos.dim() != 2 or pos.shape[1] != 3:
        raise ValueError("positions must be float32 [N, 3] (3-D scenes)")
    if pos.shape[0] > 1 and (pos.stride(1) != 1):
        pos = pos.contiguous()
    stride = pos.stride(0) if pos.shape[0] > 1 else 3
    return pos, max(int(stride), 3)


class RadiusGraph:
    """Device-resident result of one radius-graph build (neighbour lists in a workspace)."""

    def __init__(self, pos_nodes, conn_r, max_neighbours=20, nodes_per_graph=None):
        _need_cuda(pos_nodes, "pos_nodes")
        pos, stride = _pos_view(pos_nodes)
        self._keep = pos
        self.n = int(pos.shape[0])
        self.max_neighbours = int(max_neighbours)
        self.device = pos.device
        L = lib()
        self.ws = _ws(L.gm_graph_workspace_bytes(self.n, self.max_neighbours), self.device)
        per = self.n if nodes_per_graph is None else int(nodes_per_graph)
        check(L.gm_radius_graph_build_batched(C.c_void_p(pos.data_ptr()), stride, self.n, per, float(conn_r),
                                              self.max_neighbours, ptr(self.ws), self.ws.numel(), current_stream()))

    def num_edges(self):
        e = C.c_int64(0)
        check(lib().gm_radius_graph_num_edges(ptr(self.ws), C.byref(e), current_stream()))
        return int(e.value)

    def edges(self):
        e = self.num_edges()
        senders = torch.empty(e, dtype=torch.int64, device=self.device)
        receivers = torch.empty(e, dtype=torch.int64, device=self.device)
        check(lib().gm_radius_graph_edges(ptr(self.ws), self.n, self.max_neighbours, ptr(senders),
                                          ptr(receivers), e, current_stream()))
        return senders, receivers


def get_connectivity(pos_nodes, conn_r, max_neighbours=20, nodes_per_graph=None):
    """Reference ``get_connectivity`` (gnn_manip/utils/utils.py:64-93).

    Returns (senders, receivers) int64: senders = query node repeated, receivers = its in-radius
    neighbours by ascending distance, at most ``max_neighbours`` (self edge first).
    ``nodes_per_graph``: the positions are a batch of equal-sized graphs stored back to back; edges never
    cross graphs and indices carry the batch offset (collate_utils.py:76)."""
    return RadiusGraph(pos_nodes, conn_r, max_neighbours, nodes_per_graph).edges()


def get_edges_displacement(last_pos, senders, receivers, conn_r):
    """Reference ``get_edges_displacement`` (utils.py:43-61): [(p_s - p_r)/conn_r, ||.||]."""
    _need_cuda(last_pos, "last_pos")
    pos, stride = _pos_view(last_pos)
    senders = senders.contiguous().long()
    receivers = receivers.contiguous().long()
    e = int(senders.numel())
    out = torch.empty((e, 4), dtype=torch.float32, device=pos.device)
    check(lib().gm_edge_features(C.c_void_p(pos.data_ptr()), stride, ptr(senders), ptr(receivers), e,
                                 float(conn_r), ptr(out), current_stream()))
    return out


def compute_acceleration(next_pos, pos_seq):
    """Reference ``compute_acceleration`` (utils.py:10-24).  Target-side helper (training data),
    not on the rollout hot path: plain tensor arithmetic."""
    return next_pos - 2 * pos_seq[-1, :, :] + pos_seq[-2, :, :]


def random_walk_noise(pos_seq, noise_std, noise_sample=None, generator=None):
    """Reference ``random_walk_noise`` (utils.py:96-115) on the device: per-step velocity noise
    N(0, noise_std / sqrt(k-1)) accumulated twice over time (velocity, then position), zero for the first frame.
    ``noise_sample`` ([k-1, N, 3]) replaces the draw (tests pin the arithmetic with the reference's own draw)."""
    _need_cuda(pos_seq, "pos_seq")
    k, n, d = pos_seq.shape
    if noise_sample is None:
        noise_sample = torch.randn((k - 1, n, d), dtype=torch.float32, device=pos_seq.device, generator=generator)
        noise_sample = noise_sample * (float(noise_std) / (k - 1) ** 0.5)
    noisy_pos = torch.cumsum(torch.cumsum(noise_sample.float(), dim=0), dim=0)
    return torch.cat((torch.zeros((1, n, d), dtype=torch.float32, device=pos_seq.device), noisy_pos), dim=0)


def _contiguous_cols(idx, name):
    idx = list(idx)
    if len(idx) != 3 or idx[1] != idx[0] + 1 or idx[2] != idx[0] + 2:
        raise ValueError(f"{name} must be three consecutive columns, got {idx}")
    return int(idx[0])


def make_feature_desc(conn_r, stats, bounds, cartesian_idx, material_idx, control_idx, k_steps, data_dim):
    d = FeatureDesc()
    d.conn_r = float(conn_r)
    d.k_steps = int(k_steps)
    d.data_dim = int(data_dim)
    d.cart_col = _contiguous_cols(cartesian_idx, "cartesian_idx")
    d.material_col = int(material_idx[0] if isinstance(material_idx, (list, tuple)) else material_idx)
    d.control_col = -1 if control_idx is None else _contiguous_cols(control_idx, "control_idx")
    d.nodes_per_graph = 0

    def put(dst, src):
        vals = [float(v) for v in (src.tolist() if hasattr(src, "tolist") else src)]
        if len(vals) != 3:
            raise ValueError("statistics / bounds must have 3 components (3-D scenes)")
        for i in range(3):
            dst[i] = vals[i]

    put(d.vel_mean, stats["velocity_mean"])
    put(d.vel_std, stats["velocity_std"])
    put(d.acc_mean, stats["acceleration_mean"])
    put(d.acc_std, stats["acceleration_std"])
    put(d.lower_bounds, bounds["lower_bounds"])
    put(d.upper_bounds, bounds["upper_bounds"])
    return d


class GraphBoundedMultimaterial:
    """Mirror of the reference class of the same name (collate_utils.py:162-209)."""

    def __init__(self, conn_r, stats, cartesian_idx, material_idx, bounds, noise=None, max_neighbours=20):
        self.conn_r = conn_r
        self.stats = stats
        self.cartesian_idx = list(cartesian_idx)
        self.material_idx = list(material_idx)
        self.control_idx = None
        self.action_idx = None
        self.bounds = bounds
        self.noise_std = noise
        self.max_neighbours = max_neighbours
        self.generator = None  # optional torch.Generator (device) for the noise draw

    # -- helpers
    def feature_desc(self, obs):
        return make_feature_desc(self.conn_r, self.stats, self.bounds, self.cartesian_idx, self.material_idx,
                                 self.control_idx, obs.shape[0], obs.shape[2])

    @property
    def node_dim(self):
        raise AttributeError("node_dim depends on k; use compute_nodes(obs).shape[1]")

    # -- reference surface
    def compute_nodes(self, obs):
        _need_cuda(obs, "obs")
        obs = obs.contiguous().float()
        k, n, _ = obs.shape
        d = self.feature_desc(obs)
        f = 3 * (k - 1) + 7 + (3 if d.control_col >= 0 else 0)
        out = torch.empty((n, f), dtype=torch.float32, device=obs.device)
        check(lib().gm_node_features(ptr(obs), n, C.byref(d), ptr(out), current_stream()))
        return out

    def compute_edges(self, obs, senders, receivers):
        last_pos = obs[-1][:, self.cartesian_idx[0]:self.cartesian_idx[0] + 3]
        return get_edges_displacement(last_pos, senders, receivers, self.conn_r)

    def compute_target(self, obs, tgt):
        pos_seq = obs[:, :, self.cartesian_idx[0]:self.cartesian_idx[0] + 3]
        acc = compute_acceleration(tgt, pos_seq)
        dev = acc.device
        mean = torch.as_tensor(self.stats["acceleration_mean"], dtype=torch.float32, device=dev)
        std = torch.as_tensor(self.stats["acceleration_std"], dtype=torch.float32, device=dev)
        return (acc - mean) / std

    def process(self, obs, tgt, noise_sample=None):
        """GraphAttributes.process (collate_utils.py:23-27): the noisy path when a noise std was given."""
        if self.noise_std is None:
            return self._process_simple(obs, tgt)
        return self._process_noisy(obs, tgt, noise_sample)

    def _process_noisy(self, obs, tgt, noise_sample=None):
        """collate_utils.py:169-193: random-walk noise on the position columns of the window and on the target;
        features, graph (with self.max_neighbours, :187) and target acceleration come from the noisy state."""
        _need_cuda(obs, "obs")
        obs = obs.contiguous().float()
        c0 = self.cartesian_idx[0]
        seq = random_walk_noise(obs[:, :, c0:c0 + 3], self.noise_std, noise_sample, self.generator)
        noisy_obs = obs.clone()
        noisy_obs[:, :, c0:c0 + 3] += seq
        last_pos = noisy_obs[-1][:, c0:c0 + 3]
        nodes = self.compute_nodes(noisy_obs)
        senders, receivers = get_connectivity(last_pos, self.conn_r, self.max_neighbours)
        edge_attr = self.compute_edges(noisy_obs, senders, receivers)
        return nodes, edge_attr, senders, receivers, self.compute_target(noisy_obs, tgt + seq[-1])

    def _process_simple(self, obs, tgt):
        """_process_simple (collate_utils.py:29-40).  NB: like the reference, the graph is built with
        the default max_neighbours=20 here (collate_utils.py:34 does not forward the attribute)."""
        obs = obs.contiguous().float()
        last_pos = obs[-1][:, self.cartesian_idx[0]:self.cartesian_idx[0] + 3]
        nodes = self.compute_nodes(obs)
        senders, receivers = get_connectivity(last_pos, self.conn_r)
        edge_attr = self.compute_edges(obs, senders, receivers)
        nodes_tgt = self.compute_target(obs, tgt) if tgt is not None else None
        return nodes, edge_attr, senders, receivers, nodes_tgt

    def process_collate(self, batch):
        """collate_utils.py:68-87: concatenate graphs, edge indices offset by N*i."""
        nl, el, il, tl = [], [], [], []
        for i, item in enumerate(batch):
            obs, tgt = item[0], item[1]
            nodes, edge_attr, s, r, t = self.process(obs, tgt)
            ei = torch.stack((s, r)).long() + nodes.shape[0] * i
            nl.append(nodes)
            el.append(edge_attr)
            il.append(ei)
            tl.append(t)
        tgt = torch.cat(tl) if tl[0] is not None else None
        return torch.cat(nl), torch.cat(el), torch.cat(il, dim=1), tgt


class GraphBoundedMultimaterialControl(GraphBoundedMultimaterial):
    """Mirror of collate_utils.py:211-232 (node features with the 3 control columns)."""

    def __init__(self, conn_r, stats, cartesian_idx, material_idx, control_idx, bounds, noise=None,
                 max_neighbours=20):
        super().__init__(conn_r, stats, cartesian_idx, material_idx, bounds, noise, max_neighbours)
        self.control_idx = list(control_idx)
