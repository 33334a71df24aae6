"""Encode-process-decode GNN on the MI355X: drop-in for the reference's
``gnn_manip/models/epd_gnn.py`` (``EncProcDecGNN``) and for the two ``torch_graphnet`` blocks it
is built from (``GraphIndependent``, ``InteractionNetwork``; call sites epd_gnn.py:30-33,42-45,88,101).

The modules own ordinary ``nn.Sequential`` MLPs, so ``state_dict()`` / ``load_state_dict()``
round-trip vanilla checkpoints (keys ``encoder.phi_edge.0.weight`` ...).  ``forward`` never runs
those Sequentials: it hands the parameters to libgnnmanip_hip.so, which keeps a packed MFMA
operand image of them (re-packed when a parameter changes) and runs the fused HIP kernels.

Block semantics (the torch_graphnet source is absent from the reference tree; fixed by
BASELINE.json north_star): j = edge_index[0] (source), i = edge_index[1] (target);
e' = phi_e(cat[h_i, h_j, e]); agg_i = sum_{e -> i} e'; h' = phi_v(cat[h, agg]); no residual inside
the block.  ``EncProcDecGNN.forward`` is differentiable w.r.t. every parameter (``train_dyn.py``:
forward with an activation tape + hand-written HIP backward, see csrc/train.hip), and so are the two
standalone blocks (parameters and inputs), so the reference's own
``EncProcDecGNN`` wiring trains with them as well.
"""
import ctypes as C

import torch
import torch.nn as nn

from ._lib import ModelDesc, check, current_stream, lib, ptr
from .graph import _need_cuda, _ws


FLOWS = {"source_to_target": 0, "target_to_source": 1}


def _convention(flow, concat, node_concat):
    """(flow, col_i, col_j, col_e, node_agg_first) of gm_model_desc from the block's keyword arguments."""
    if flow not in FLOWS:
        raise ValueError("flow must be 'source_to_target' (aggregate at edge_index[1]) or 'target_to_source'")
    concat = tuple(concat)
    if sorted(concat) != ["e", "i", "j"]:
        raise ValueError("concat must be a permutation of ('i', 'j', 'e')")
    node_concat = tuple(node_concat)
    if sorted(node_concat) != ["agg", "h"]:
        raise ValueError("node_concat must be ('h', 'agg') or ('agg', 'h')")
    return (FLOWS[flow], concat.index("i"), concat.index("j"), concat.index("e"), 1 if node_concat[0] == "agg" else 0)


def _mlp_params(seq):
    """Parameters of a reference-style MLP in state_dict order."""
    return [p for _, p in seq.named_parameters()]


def _mlp_dims(seq):
    lin = [m for m in seq if isinstance(m, nn.Linear)]
    norm = [m for m in seq if isinstance(m, nn.LayerNorm)]
    return lin, norm


class _Model:
    """A gm_model* that lives as long as anything refers to it (the owning _Handle, or the ctx of a pending backward):
    ctypes passes `_as_parameter_`; the library object is destroyed when the last reference goes."""

    def __init__(self, pointer):
        self._as_parameter_ = pointer

    def __del__(self):
        try:
            lib().gm_model_destroy(self._as_parameter_)
        except Exception:
            pass


class _Handle:
    """Owns a gm_model built from a list of parameter tensors; re-packs when they change."""

    def __init__(self):
        self.h = None
        self.key = None
        self.built_for = None
        self.desc = None
        self.edge_kernel = 0
        self.prof_mask = 0

    def get(self, desc_tuple, params, device, key_params=None):
        """key_params: the tensors whose (address, version) decide whether `params` changed -- `params` themselves, unless those are
        DERIVED tensors (EncProcDecGNN._padded_training builds fresh padded copies every step: version 0 always, and the caching
        allocator hands out the same addresses step after step), in which case the caller names the parameters they derive from."""
        key = (desc_tuple, str(device), tuple((p.data_ptr(), p._version) for p in (params if key_params is None else key_params)))
        if self.h is not None and key == self.key:
            return self.h
        L = lib()
        tensors = [p.detach().to(device=device, dtype=torch.float32).contiguous() for p in params]
        arr = (C.c_void_p * len(tensors))(*[ptr(t).value for t in tensors])
        d = ModelDesc(*desc_tuple)
        if self.h is not None and self.built_for == (desc_tuple, str(device)):   # (not self.key: invalidate() clears that)
            check(L.gm_model_update(self.h, arr, len(tensors), 1, current_stream()))
        else:
            self.close()
            out = C.c_void_p()
            check(L.gm_model_create(C.byref(d), arr, len(tensors), 1, current_stream(), C.byref(out)))
            self.h = _Model(out)
            if self.edge_kernel:
                check(L.gm_model_set_edge_kernel(self.h, self.edge_kernel))
            if self.prof_mask:
                check(L.gm_model_profile(self.h, self.prof_mask))
        # no synchronisation: the pack kernels are queued on torch's current stream, and the caching allocator hands the
        # temporaries' memory out again only in that stream's order
        self.key = key
        self.built_for = (desc_tuple, str(device))
        self.desc = d
        return self.h

    def set_edge_kernel(self, choice):
        self.edge_kernel = int(choice)
        if self.h is not None:
            check(lib().gm_model_set_edge_kernel(self.h, self.edge_kernel))

    def profile(self, kind_mask):
        self.prof_mask = int(kind_mask)
        if self.h is not None:
            check(lib().gm_model_profile(self.h, self.prof_mask))

    def profile_query(self, kind):
        launches, ms = C.c_int64(0), C.c_double(0.0)
        if self.h is not None:
            check(lib().gm_model_profile_query(self.h, int(kind), C.byref(launches), C.byref(ms)))
        return int(launches.value), float(ms.value)

    def invalidate(self):
        """Forget the packed state: the next use re-packs.  Needed only after writes that bypass autograd's version
        counter (``p.data.copy_(..)``, raw pointers); in-place ops on the parameters themselves are noticed."""
        self.key = None

    def close(self):
        self.h = None   # destroyed once no autograd ctx holds it either

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _grad_arrays(params, device):
    """(tensors, flat gradient buffer, per-tensor views, ctypes arrays) for a backward call."""
    tensors = [p.detach().to(device=device, dtype=torch.float32).contiguous() for p in params]
    flat = torch.zeros(sum(t.numel() for t in tensors), dtype=torch.float32, device=device)
    views, off = [], 0
    for t in tensors:
        views.append(flat[off:off + t.numel()].view_as(t))
        off += t.numel()
    t_arr = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    g_arr = (C.c_void_p * len(views))(*[v.data_ptr() for v in views])
    return tensors, views, t_arr, g_arr


def _wants_grad(params, *inputs):
    return torch.is_grad_enabled() and (any(p.requires_grad for p in params) or any(t.requires_grad for t in inputs))


def _check_edge_index(edge_index, n, e, ranges=True):
    _need_cuda(edge_index, "edge_index")
    if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.shape[0] != 2 or edge_index.shape[1] != e:
        raise ValueError("edge_index must be int64 [2, E]")
    if ranges and e and (int(edge_index.min()) < 0 or int(edge_index.max()) >= n):   # two blocking reductions
        raise ValueError("edge_index entry out of range [0, n_nodes)")


class _GraphIndependentFunction(torch.autograd.Function):
    """GraphIndependent under autograd: gradients for the block's parameters (its inputs are data)."""

    @staticmethod
    def forward(ctx, block, n_own, x, edge_attr, *params):
        L = lib()
        desc, _ = block._standalone(x.device)
        h = block._handle.get(desc, list(params), x.device)
        d = ModelDesc(*desc)
        n, e = int(x.shape[0]), int(edge_attr.shape[0])
        tape = _ws(L.gm_block_tape_bytes(C.byref(d), 0, n, e), x.device)
        h_out = torch.empty((n, desc[3]), dtype=torch.float32, device=x.device)
        e_out = torch.empty((e, desc[3]), dtype=torch.float32, device=x.device)
        check(L.gm_graph_independent_forward_train(h, ptr(x), n, ptr(edge_attr), e, ptr(h_out), ptr(e_out), ptr(tape),
                                                   tape.numel(), current_stream()))
        ctx.handle, ctx.desc, ctx.tape, ctx.n_own = h, d, tape, n_own
        ctx.save_for_backward(x, edge_attr, *params)
        return h_out, e_out

    @staticmethod
    def backward(ctx, dh, de):
        L = lib()
        x, edge_attr, *params = ctx.saved_tensors
        n, e = int(x.shape[0]), int(edge_attr.shape[0])
        dh = torch.zeros((n, ctx.desc.hidden_size), device=x.device) if dh is None else dh.contiguous().float()
        de = torch.zeros((e, ctx.desc.hidden_size), device=x.device) if de is None else de.contiguous().float()
        tensors, views, t_arr, g_arr = _grad_arrays(params, x.device)
        ws = _ws(L.gm_block_backward_workspace_bytes(C.byref(ctx.desc), n, e), x.device)
        dx = torch.empty_like(x) if ctx.needs_input_grad[2] else None
        dea = torch.empty_like(edge_attr) if ctx.needs_input_grad[3] else None
        check(L.gm_graph_independent_backward(ctx.handle, t_arr, len(tensors), ptr(x), ptr(edge_attr), n, e, ptr(dh), ptr(de),
                                              ptr(dx), ptr(dea), g_arr, ptr(ctx.tape), ctx.tape.numel(), ptr(ws), ws.numel(),
                                              current_stream()))
        ctx.tape = None
        return (None, None, dx, dea) + tuple(v if i < ctx.n_own else None for i, v in enumerate(views))


class _InteractionNetworkFunction(torch.autograd.Function):
    """InteractionNetwork under autograd: gradients for its parameters and for both inputs (h, e)."""

    @staticmethod
    def forward(ctx, block, own, x, edge_attr, edge_index, *params):
        L = lib()
        desc, _ = block._standalone(x.device)
        h = block._handle.get(desc, list(params), x.device)
        d = ModelDesc(*desc)
        n, e = int(x.shape[0]), int(edge_attr.shape[0])
        tape = _ws(L.gm_block_tape_bytes(C.byref(d), 1, n, e), x.device)
        h_out = torch.empty_like(x)
        e_out = torch.empty_like(edge_attr)
        ei = edge_index.contiguous()
        check(L.gm_interaction_network_forward_train(h, 0, ptr(x), n, ptr(edge_attr), ptr(ei), e, ptr(h_out), ptr(e_out),
                                                     ptr(tape), tape.numel(), current_stream()))
        ctx.handle, ctx.desc, ctx.tape, ctx.own = h, d, tape, own
        ctx.save_for_backward(x, edge_attr, *params)
        return h_out, e_out

    @staticmethod
    def backward(ctx, dh, de):
        L = lib()
        x, edge_attr, *params = ctx.saved_tensors
        n, e = int(x.shape[0]), int(edge_attr.shape[0])
        dh = torch.zeros_like(x) if dh is None else dh.contiguous().float()
        de = torch.zeros_like(edge_attr) if de is None else de.contiguous().float()
        tensors, views, t_arr, g_arr = _grad_arrays(params, x.device)
        dh_in = torch.empty_like(x)
        de_in = torch.empty_like(edge_attr)
        ws = _ws(L.gm_block_backward_workspace_bytes(C.byref(ctx.desc), n, e), x.device)
        check(L.gm_interaction_network_backward(ctx.handle, 0, t_arr, len(tensors), ptr(x), ptr(edge_attr), n, e, ptr(dh), ptr(de),
                                                ptr(dh_in), ptr(de_in), g_arr, ptr(ctx.tape), ctx.tape.numel(), ptr(ws), ws.numel(),
                                                current_stream()))
        ctx.tape = None
        lo, hi = ctx.own
        return (None, None, dh_in, de_in, None) + tuple(v if lo <= i < hi else None for i, v in enumerate(views))


class DstCsr:
    """Destination-sorted edge structure of an edge_index [2, E] (int64) on the device."""

    def __init__(self, edge_index, n_nodes, flow=0):
        _need_cuda(edge_index, "edge_index")
        if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.shape[0] != 2:
            raise ValueError("edge_index must be int64 [2, E]")
        ei = edge_index.contiguous()
        self.n = int(n_nodes)
        self.e = int(ei.shape[1])
        L = lib()
        self.ws = _ws(L.gm_csr_workspace_bytes(self.n, self.e), ei.device)
        check(L.gm_csr_from_edge_index_flow(ptr(ei), self.n, self.e, int(flow), ptr(self.ws), self.ws.numel(), current_stream()))

    def validate(self):
        e = C.c_int64(0)
        check(lib().gm_csr_num_edges(ptr(self.ws), C.byref(e), current_stream()))
        return int(e.value)



class _HeaderWatch:
    """Asynchronous look at the 16-byte header (n_edges, error flags, flow, left-out edges) at the start of a csr workspace -- or of a
    training tape, which begins with one: the copy into a pinned int32 row is queued behind the work enqueued so far, ``poll``
    tells without blocking whether it has arrived and whether a flag is set, ``check`` raises the library's message for it
    (gm_csr_header_status: no device access).  The workspace itself is not kept alive."""

    def __init__(self, ws, pinned_row):
        self._host = pinned_row
        self._host.zero_()     # a recycled row must not show an earlier forward's verdict while this copy is in flight
        # the copy and the event go on the stream the library's work for `ws` runs on: the current stream of ws's device, which
        # need not be the current device (_lib.current_stream / DevGuard support tensors on another device)
        with torch.cuda.device(ws.device):
            self._host.copy_(ws[:16].view(torch.int32), non_blocking=True)
            self._event = torch.cuda.Event()
            self._event.record(torch.cuda.current_stream(ws.device))

    def poll(self):
        """None: still in flight; False: finished clean; True: finished with an error flag."""
        if not self._event.query():
            return None
        return int(self._host[1]) != 0

    def wait(self):
        self._event.synchronize()

    def check(self):
        check(lib().gm_csr_header_status(C.c_void_p(self._host.data_ptr()), None))


def padded_hidden(hidden):
    """Width the kernels run a model of this hidden size at (gm_padded_hidden_size): sizes up to 256 are zero-padded to
    64 / 128 / 256; latent tensors handed to / returned by the standalone blocks have that row stride inside the library."""
    w = int(lib().gm_padded_hidden_size(int(hidden)))
    if w <= 0:
        raise ValueError(f"hidden size {hidden}: supported are 1 .. 256")
    return w


def _pad_cols(t, width):
    return t if t.shape[1] == width else torch.nn.functional.pad(t, (0, width - t.shape[1]))


def _zeros_mlp(fin, hidden, fout, num_layers, norm, device):
    t = [torch.zeros(hidden, fin, device=device), torch.zeros(hidden, device=device)]
    for _ in range(num_layers - 1):
        t += [torch.zeros(hidden, hidden, device=device), torch.zeros(hidden, device=device)]
    t += [torch.zeros(fout, hidden, device=device), torch.zeros(fout, device=device)]
    if norm:
        t += [torch.ones(fout, device=device), torch.zeros(fout, device=device)]
    return t


class GraphIndependent(nn.Module):
    """``torch_graphnet.GraphIndependent(phi_edge=, phi_node=)``: (x, e, idx) -> (phi_node(x), phi_edge(e), None)."""

    def __init__(self, phi_edge, phi_node):
        super().__init__()
        self.phi_edge = phi_edge
        self.phi_node = phi_node
        self._handle = _Handle()
        self._pad = {}

    def _standalone(self, device):
        # a standalone block runs through a one-step model handle whose other MLPs are zero placeholders
        le, ne = _mlp_dims(self.phi_edge)
        ln, nn_ = _mlp_dims(self.phi_node)
        hidden, nl = le[-1].out_features, len(le) - 1
        eps = ne[0].eps if ne else 1e-5
        desc = (ln[0].in_features, le[0].in_features, 1, hidden, nl, 1, float(eps))
        if str(device) not in self._pad:
            self._pad[str(device)] = (_zeros_mlp(3 * hidden, hidden, hidden, nl, True, device)
                                      + _zeros_mlp(2 * hidden, hidden, hidden, nl, True, device)
                                      + _zeros_mlp(hidden, hidden, 1, nl, False, device))
        params = _mlp_params(self.phi_edge) + _mlp_params(self.phi_node) + self._pad[str(device)]
        return desc, params

    def forward(self, x, edge_attr, edge_index=None):
        _need_cuda(x, "x")
        desc, params = self._standalone(x.device)
        x = x.contiguous().float()
        edge_attr = edge_attr.contiguous().float()
        own = list(self.parameters())
        if _wants_grad(own, x, edge_attr):
            h_out, e_out = _GraphIndependentFunction.apply(self, len(own), x, edge_attr, *params)
            return h_out, e_out, None
        h = self._handle.get(desc, params, x.device)
        hidden = desc[3]
        hp = padded_hidden(hidden)   # the library's row stride of latents; the padded columns come back as zeros
        h_out = torch.empty((x.shape[0], hp), dtype=torch.float32, device=x.device)
        e_out = torch.empty((edge_attr.shape[0], hp), dtype=torch.float32, device=x.device)
        check(lib().gm_graph_independent_forward(h, ptr(x), x.shape[0], ptr(edge_attr), edge_attr.shape[0],
                                                 ptr(h_out), ptr(e_out), current_stream()))
        if hp != hidden:
            h_out, e_out = h_out[:, :hidden].contiguous(), e_out[:, :hidden].contiguous()
        return h_out, e_out, None


class InteractionNetwork(nn.Module):
    """``torch_graphnet.InteractionNetwork(phi_edge=, phi_node=)``: (h, e, idx) -> (h', e', None).

    The block's source is absent from the reference tree (.gitmodules:1-3); the defaults are the convention of DESIGN.md
    section 2.  ``flow`` / ``concat`` / ``node_concat`` select another one so that a checkpoint trained with the real block can
    be matched: flow='target_to_source' aggregates at edge_index[0]; concat is the order in which (h_i, h_j, e) enter phi_e;
    node_concat the order of (h, agg) in phi_v.  They act when the weights are packed and the edges are sorted, not at run time."""

    def __init__(self, phi_edge, phi_node, flow="source_to_target", concat=("i", "j", "e"), node_concat=("h", "agg")):
        super().__init__()
        self.phi_edge = phi_edge
        self.phi_node = phi_node
        self.convention = _convention(flow, concat, node_concat)
        self._handle = _Handle()
        self._pad = {}

    def _standalone(self, device):
        le, ne = _mlp_dims(self.phi_edge)
        hidden, nl = le[-1].out_features, len(le) - 1
        if le[0].in_features != 3 * hidden:
            raise ValueError("InteractionNetwork.phi_edge must take 3*hidden inputs ([h_i, h_j, e])")
        eps = ne[0].eps if ne else 1e-5
        desc = (1, 1, 1, hidden, nl, 1, float(eps)) + self.convention
        if str(device) not in self._pad:
            self._pad[str(device)] = (_zeros_mlp(1, hidden, hidden, nl, True, device) + _zeros_mlp(1, hidden, hidden, nl, True, device),
                                      _zeros_mlp(hidden, hidden, 1, nl, False, device))
        pre, post = self._pad[str(device)]
        params = pre + _mlp_params(self.phi_edge) + _mlp_params(self.phi_node) + post
        return desc, params

    def forward(self, x, edge_attr, edge_index):
        _need_cuda(x, "x")
        desc, params = self._standalone(x.device)
        own = list(self.parameters())
        if _wants_grad(own, x, edge_attr):
            x = x.contiguous().float()
            edge_attr = edge_attr.contiguous().float()
            _check_edge_index(edge_index, x.shape[0], edge_attr.shape[0])
            lo = len(self._pad[str(x.device)][0])  # placeholder encoder tensors come first
            h_out, e_out = _InteractionNetworkFunction.apply(self, (lo, lo + len(own)), x, edge_attr, edge_index, *params)
            return h_out, e_out, None
        h = self._handle.get(desc, params, x.device)
        return _run_block(h, desc, 0, x, edge_attr, edge_index)


def _run_block(handle, desc, k, x, edge_attr, edge_index, csr=None):
    hidden, hp = desc[3], padded_hidden(desc[3])
    x = _pad_cols(x.float(), hp).contiguous()            # library row stride (zero padding stays zero through the block)
    edge_attr = _pad_cols(edge_attr.float(), hp).contiguous()
    n, e = x.shape[0], edge_attr.shape[0]
    if csr is None:
        csr = DstCsr(edge_index, n, flow=desc[7] if len(desc) > 7 else 0)
    L = lib()
    d = ModelDesc(*desc)
    fwd = _ws(L.gm_block_workspace_bytes(C.byref(d), n, e), x.device)
    h_out = torch.empty_like(x)
    e_out = torch.empty_like(edge_attr)
    check(L.gm_interaction_network_forward(handle, k, ptr(x), n, ptr(edge_attr), ptr(csr.ws), e, ptr(h_out),
                                           ptr(e_out), ptr(fwd), fwd.numel(), current_stream()))
    if hp != hidden:
        h_out, e_out = h_out[:, :hidden].contiguous(), e_out[:, :hidden].contiguous()
    return h_out, e_out, None


class _EpdTrainFunction(torch.autograd.Function):
    """``EncProcDecGNN.forward`` under autograd (examples/train_dyn.py:45-72): the forward records the
    activation tape in one device buffer, the backward is gm_epd_backward.  Gradients are produced for the
    parameters only; nodes / edge_attr / edge_index are data.  `spec` = (model descriptor tuple, _Handle): the module's own, or
    -- for a hidden size between the training kernels' widths -- the zero-padded model's (EncProcDecGNN._padded_training), with
    the module's own parameters as the third element: the tensors whose versions say when the padded copies are stale."""

    @staticmethod
    def forward(ctx, module, spec, nodes, edge_attr, edge_index, *params):
        L = lib()
        n, e = int(nodes.shape[0]), int(edge_attr.shape[0])
        desc_tuple, handle, key_params = spec
        h = handle.get(desc_tuple, list(params), nodes.device, key_params)
        d = ModelDesc(*desc_tuple)
        tape = _ws(L.gm_train_tape_bytes(C.byref(d), n, e), nodes.device)
        out = torch.empty((n, module.dims[2]), dtype=torch.float32, device=nodes.device)
        ei = edge_index.contiguous()
        check(L.gm_epd_forward_train(h, ptr(nodes), n, ptr(edge_attr), ptr(ei), e, ptr(out), ptr(tape), tape.numel(),
                                     current_stream()))
        ctx.module, ctx.handle, ctx.desc, ctx.tape = module, h, d, tape
        ctx.sizes = (n, e)
        # the tape begins with the forward's csr workspace: its header carries the edge_index verdict.  Watched whatever
        # auto_status says (that switch only decides whether a later forward looks at it unasked): status() must see it.
        module._watch(tape)
        ctx.save_for_backward(nodes, edge_attr, *params)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        L = lib()
        nodes, edge_attr, *params = ctx.saved_tensors
        n, e = ctx.sizes
        dev = nodes.device
        grad_out = grad_out.contiguous().float()
        tensors, views, t_arr, g_arr = _grad_arrays(params, dev)
        ws = _ws(L.gm_train_backward_workspace_bytes(C.byref(ctx.desc), n, e), dev)
        check(L.gm_epd_backward(ctx.handle, t_arr, len(tensors), ptr(nodes), ptr(edge_attr), n, e, ptr(grad_out), g_arr,
                                ptr(ctx.tape), ctx.tape.numel(), ptr(ws), ws.numel(), current_stream()))
        ctx.tape = None
        return (None, None, None, None, None) + tuple(views)


TRAIN_WIDTHS = (64, 128, 256)   # hidden sizes the training kernels are instantiated for (csrc/train.hip)


class EncProcDecGNN(nn.Module):
    """Drop-in for the reference ``EncProcDecGNN`` (gnn_manip/models/epd_gnn.py:11-105)."""

    def __init__(self, node_dim, edge_dim, out_dim, hidden_size, num_layers, m_steps, norm_type='LayerNorm', *,
                 flow="source_to_target", concat=("i", "j", "e"), node_concat=("h", "agg")):
        """Positional arguments as the reference (epd_gnn.py:13-14).  The keyword-only flow / concat / node_concat choose the
        InteractionNetwork convention (see ``InteractionNetwork``); the defaults are the documented one."""
        super().__init__()
        assert (num_layers >= 2), "The number of layers num_layers must be at least 2"
        assert (m_steps >= 1), "The number of m_steps message pasting steps must be at least 1"
        if norm_type != 'LayerNorm':
            # the reference's BatchNorm2d / InstanceNorm2d branches (epd_gnn.py:53-58) cannot run on its 2-D inputs
            raise NotImplementedError("only norm_type='LayerNorm' is supported")
        self.dims = (node_dim, edge_dim, out_dim, hidden_size, num_layers, m_steps)
        self.convention = _convention(flow, concat, node_concat)
        self.encoder = GraphIndependent(phi_edge=self._build_mlp(edge_dim, hidden_size, hidden_size, num_layers, norm=True),
                                        phi_node=self._build_mlp(node_dim, hidden_size, hidden_size, num_layers, norm=True))
        self.processor = nn.ModuleList([
            InteractionNetwork(phi_edge=self._build_mlp(3 * hidden_size, hidden_size, hidden_size, num_layers, norm=True),
                               phi_node=self._build_mlp(2 * hidden_size, hidden_size, hidden_size, num_layers, norm=True),
                               flow=flow, concat=concat, node_concat=node_concat)
            for _ in range(m_steps)])
        self.decoder = self._build_mlp(hidden_size, hidden_size, out_dim, num_layers, norm=False)
        self._handle = _Handle()

    @staticmethod
    def _build_mlp(input_dim, hidden_size, output_dim, num_layers, norm=False):
        modules = [nn.Linear(input_dim, hidden_size), nn.ReLU()]
        for _ in range(num_layers - 1):
            modules.append(nn.Linear(hidden_size, hidden_size))
            modules.append(nn.ReLU())
        modules.append(nn.Linear(hidden_size, output_dim))
        if norm:
            modules.append(nn.LayerNorm(output_dim))
        return nn.Sequential(*modules)

    def print_structure(self):
        print("Encoder ")
        print(self.encoder)
        print("Processor")
        for module in self.processor:
            print(module)
        print("Decoder ")
        print(self.decoder)

    # -- device handle
    def model_desc(self):
        eps = self.encoder.phi_edge[-1].eps
        return tuple(int(v) for v in self.dims) + (float(eps),) + self.convention

    def device_handle(self, device):
        """gm_model* for the current parameters on `device` (packed once, re-packed on change)."""
        return self._handle.get(self.model_desc(), list(self.parameters()), device)

    def _padded_training(self, params):
        """Training at a hidden size between the kernels' widths (train_dyn.py:237-238 takes any int): the model runs zero-padded
        at the next width, as the inference kernels do.  The padded parameters are built from the real ones by differentiable torch
        operations, so autograd carries the gradients back through them:
          * every hidden dimension is zero-padded (input blocks of the processors' first Linears each to the padded width);
          * the Linear in front of a LayerNorm is centred over its outputs (W - mean_rows(W), b - mean(b)): its outputs have zero
            mean over the features that exist, the padded ones are exactly zero, and a LayerNorm over the padded width Hp with
            eps' = eps Hv / Hp and gamma' = gamma sqrt(Hv / Hp) is the LayerNorm over the Hv real features:
            x / sqrt(Q / Hp + eps') * gamma' = x / sqrt(Q / Hv + eps) * gamma  (Q = sum of squares); beta is padded with zeros, so
            padded features leave every LayerNorm as zeros again.
        Returns ((descriptor tuple of the padded model, its handle), padded parameter tensors)."""
        node_dim, edge_dim, out_dim, hv, nl, m_steps = self.dims
        hp = padded_hidden(hv)
        if hp not in TRAIN_WIDTHS:
            raise NotImplementedError(f"training at hidden_size={hv}: supported are 1 .. {TRAIN_WIDTHS[-1]}")
        pad = torch.nn.functional.pad
        scale = (hv / hp) ** 0.5
        it = iter(params)

        def mlp(in_blocks, normed, out_rows):
            # in_blocks: widths of the column blocks of the first Linear that are hidden-sized (padded each) or raw (kept)
            out = []
            for l in range(nl + 1):
                w, b = next(it), next(it)
                last = l == nl
                if last and normed:
                    w, b = w - w.mean(dim=0, keepdim=True), b - b.mean()
                if l == 0:
                    cols, c0 = [], 0
                    for width, hidden_block in in_blocks:
                        blk = w[:, c0:c0 + width]
                        cols.append(pad(blk, (0, hp - width)) if hidden_block else blk)
                        c0 += width
                    w = torch.cat(cols, dim=1)
                else:
                    w = pad(w, (0, hp - hv))
                if not last or normed:   # a hidden-sized output (all but the decoder's last Linear): padded rows / bias entries are zero
                    w, b = pad(w, (0, 0, 0, hp - hv)), pad(b, (0, hp - hv))
                out += [w.contiguous(), b.contiguous()]
            if normed:
                g, bt = next(it), next(it)
                out += [pad(g * scale, (0, hp - hv)), pad(bt, (0, hp - hv))]
            return out

        padded = mlp([(edge_dim, False)], True, hv) + mlp([(node_dim, False)], True, hv)
        for _ in range(m_steps):
            padded += mlp([(hv, True)] * 3, True, hv) + mlp([(hv, True)] * 2, True, hv)
        padded += mlp([(hv, True)], False, out_dim)
        eps = float(self.encoder.phi_edge[-1].eps) * hv / hp
        desc = (int(node_dim), int(edge_dim), int(out_dim), hp, int(nl), int(m_steps), eps) + self.convention
        if "_pad_handle" not in self.__dict__:
            self.__dict__["_pad_handle"] = _Handle()
        return (desc, self._pad_handle), padded

    auto_status = True   # check the previous inference forward's device-side error flags at the start of the next one

    EDGE_KERNELS = {"auto": 0, "sys": 5, "hm": 6, "sys_all": 7}   # 1 .. 4 were the round-1 fp32 / bf16 x 6 kernels (removed in round 5)

    def profile(self, kind_mask):
        """HIP-event timing of this model's launches (gm_model_profile; bit 0 processor edge kernel, 1 processor node
        kernel, 2 radius-graph build of the rollout step, 3 encoders).  0 switches it off."""
        self._handle.profile(kind_mask)

    def profile_query(self, kind):
        """(launches, total milliseconds) recorded for `kind` since it was enabled; synchronises on the events."""
        return self._handle.profile_query(kind)

    def invalidate_packed_weights(self):
        """Re-pack the device weight images on next use (after writes through ``.data`` / raw pointers, which the
        version counters do not see)."""
        self._handle.invalidate()
        if "_pad_handle" in self.__dict__:   # the zero-padded training model of a hidden size between the kernels' widths
            self._pad_handle.invalidate()

    def set_edge_kernel(self, choice):
        """Processor edge kernel of this model (diagnostics / A-B measurements; no reference counterpart): 'auto', 'sys'
        (systolic fp16 x 3, hidden 128 / num_layers 2), 'hm' (streamed fp16 x 3), 'sys_all' (as 'sys', and the systolic node /
        projection kernels whatever the graph's size: 'auto' takes them for graphs of 49152 nodes or more).  See include/gnn_manip_hip.h."""
        self._handle.set_edge_kernel(self.EDGE_KERNELS.get(choice, choice))

    def forward(self, nodes, edge_attr, edge_index):
        """epd_gnn.py:86-98: encoder -> m_steps x (InteractionNetwork + residuals) -> decoder, fused."""
        _need_cuda(nodes, "nodes")
        nodes = nodes.contiguous().float()
        edge_attr = edge_attr.contiguous().float()
        n, e = int(nodes.shape[0]), int(edge_attr.shape[0])
        if nodes.shape[1] != self.dims[0] or edge_attr.shape[1] != self.dims[1]:
            raise ValueError("nodes / edge_attr feature widths do not match the model")
        if edge_index.shape[1] != e:
            raise ValueError("edge_index and edge_attr disagree on the number of edges")
        params = list(self.parameters())
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            # training (examples/train_dyn.py:45-72): forward with tape, HIP backward
            if nodes.requires_grad or edge_attr.requires_grad:
                raise NotImplementedError("EncProcDecGNN: gradients w.r.t. nodes / edge_attr are not produced "
                                          "(they are data in train_dyn.py); detach them")
            # edge_index entries outside [0, n) are flagged on the device by the forward's destination sort (and left out; the
            # kernels stay inside their arrays): no blocking range check here -- a training loop queues its steps ahead of the GPU.
            # The flag surfaces as GMError at a later forward (auto_status) or at status(), like the inference path's.  Until it
            # does, the flagged step cannot inject garbage: its prediction is NaN (so its loss is, visibly) and its backward returns
            # exactly zero gradients (csrc/train_model.hip: poison_if_flagged_kernel / gate_grad_out_kernel).  That is NOT the same
            # as the reference's raise at the forward for every optimiser: plain SGD leaves the weights alone on a zero gradient,
            # but Adam (train_dyn.py:58) still advances its step count, decays its moment estimates and moves the weights along its
            # first-moment history, and weight decay applies as always -- for the one to eight steps until the error is raised.
            _check_edge_index(edge_index, n, e, ranges=False)
            if self.auto_status:
                self._reap_watched(block=False)
            hidden = self.dims[3]
            if hidden in TRAIN_WIDTHS:
                return _EpdTrainFunction.apply(self, (self.model_desc(), self._handle, None), nodes, edge_attr, edge_index, *params)
            spec, padded = self._padded_training(params)
            # the padded tensors are rebuilt every step (version 0, recycled addresses): the handle is keyed on the parameters they
            # come from, so an optimiser step on ANY of them -- whatever is frozen -- re-packs the padded model's weight streams
            return _EpdTrainFunction.apply(self, spec + (tuple(params),), nodes, edge_attr, edge_index, *padded)
        if self.auto_status:
            # EARLIER inference forwards of this model: a device-side error (edge_index entry out of range, fp16 split range
            # exceeded) of one that has FINISHED surfaces here -- a reference-style caller never calls status() itself.
            # Nothing blocks: every forward leaves an asynchronous copy of its CSR header in pinned memory behind it, and
            # this only looks at the copies that have arrived; a forward still in flight is looked at by a later call, or by
            # status().  (A loop over forward() keeps queueing work ahead of the GPU.  model.auto_status = False opts out.)
            self._reap_watched(block=False)
        h = self.device_handle(nodes.device)
        csr = DstCsr(edge_index, n, flow=self.convention[0])
        L = lib()
        d = ModelDesc(*self.model_desc())
        fwd = _ws(L.gm_forward_workspace_bytes(C.byref(d), n, e), nodes.device)
        out = torch.empty((n, self.dims[2]), dtype=torch.float32, device=nodes.device)
        check(L.gm_epd_forward(h, ptr(nodes), n, ptr(edge_attr), 0, ptr(csr.ws), e, ptr(out), ptr(fwd),
                               fwd.numel(), current_stream()))
        # no synchronisation here: an out-of-range edge_index entry is dropped by the destination sort and flagged in the
        # CSR header, like a value outside the fp16 split range; status() -- or a later forward -- reports it
        self._last_csr = csr
        if self.auto_status:
            self._watch(csr)
        return out

    _WATCH_SLOTS = 8

    def _watch(self, csr):
        w = self.__dict__.setdefault("_watched", [])
        if "_watch_pin" not in self.__dict__ or self._watch_pin.device != torch.device("cpu"):
            self._watch_pin, self._watch_next = torch.zeros((self._WATCH_SLOTS, 4), dtype=torch.int32).pin_memory(), 0
        if len(w) >= self._WATCH_SLOTS:   # every pinned row in use (a loop 8 forwards ahead of the GPU)
            if self.auto_status:
                self._reap_watched(block=True, at_most=1)   # the oldest forward is waited for, its error raised here
            else:
                w.pop(0)   # the caller opted out of unasked checks: the oldest watch is dropped unread, nothing blocks or raises (its
                           # row is reused: the new copy is queued behind the old one on the same stream)
        is_csr = isinstance(csr, DstCsr)   # else: a training tape (it begins with the forward's csr workspace); not kept alive here
        w.append((_HeaderWatch(csr.ws if is_csr else csr, self._watch_pin[self._watch_next]), csr if is_csr else None))
        self._watch_next = (self._watch_next + 1) % self._WATCH_SLOTS

    def _reap_watched(self, block, at_most=None):
        w = self.__dict__.get("_watched", [])
        done = 0
        while w and (at_most is None or done < at_most):
            r = w[0][0].poll()
            if r is None:
                if not block:
                    break
                w[0][0].wait()
                r = w[0][0].poll()
            watch, csr = w.pop(0)
            done += 1
            if r:
                if csr is getattr(self, "_last_csr", None):
                    self._last_csr = None    # reported once
                watch.check()                # raises the library's message for the flag

    def status(self):
        """Checks the last inference forward (synchronises): raises GMError if its edge_index held an entry outside
        [0, n_nodes) -- such edges were left out --, else returns its edge count."""
        self._reap_watched(block=True)       # earlier forwards first: the oldest error is the one reported
        csr = getattr(self, "_last_csr", None)
        if csr is None:
            return 0
        try:
            return csr.validate()
        except Exception:
            self._last_csr = None   # reported once
            raise

    # the reference's per-step helper, kept for API parity (epd_gnn.py:100-105)
    def _process(self, in_module, prev_latent_node, prev_latent_edge, edge_index):
        latent_node_k, latent_edge_k, _ = in_module(prev_latent_node, prev_latent_edge, edge_index)
        return latent_node_k + prev_latent_node, latent_edge_k + prev_latent_edge
