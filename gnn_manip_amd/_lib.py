"""ctypes binding of libgnnmanip_hip.so (include/gnn_manip_hip.h).

The HIP library is the product: there is no CPU fallback.  Loading fails with a clear
error if the shared object has not been built (``python -m gnn_manip_amd.build`` or
``__graft_entry__.build()``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GM_LIB_PATH") or os.path.join(_HERE, "libgnnmanip_hip.so")   # GM_LIB_PATH: A/B builds of the same library

GM_OK = 0


class GMError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libgnnmanip_hip: {msg} (status {code})")
        self.code = code


class FeatureDesc(C.Structure):
    _fields_ = [("conn_r", C.c_double), ("k_steps", C.c_int32), ("data_dim", C.c_int32),
                ("cart_col", C.c_int32), ("material_col", C.c_int32), ("control_col", C.c_int32),
                ("nodes_per_graph", C.c_int32),
                ("vel_mean", C.c_float * 3), ("vel_std", C.c_float * 3),
                ("acc_mean", C.c_float * 3), ("acc_std", C.c_float * 3),
                ("lower_bounds", C.c_float * 3), ("upper_bounds", C.c_float * 3)]


class ModelDesc(C.Structure):
    _fields_ = [("node_dim", C.c_int32), ("edge_dim", C.c_int32), ("out_dim", C.c_int32),
                ("hidden_size", C.c_int32), ("num_layers", C.c_int32), ("m_steps", C.c_int32),
                ("ln_eps", C.c_float),
                # InteractionNetwork convention (all zero = default)
                ("flow", C.c_int32), ("col_i", C.c_int32), ("col_j", C.c_int32), ("col_e", C.c_int32),
                ("node_agg_first", C.c_int32)]


_vp, _i64, _i32, _sz, _f32, _f64 = C.c_void_p, C.c_int64, C.c_int, C.c_size_t, C.c_float, C.c_double
_FD, _MD = C.POINTER(FeatureDesc), C.POINTER(ModelDesc)

# name -> (restype, argtypes); every symbol declared in include/gnn_manip_hip.h
PROTOTYPES = {
    "gm_last_error": (C.c_char_p, []),
    "gm_abi_version": (_i32, []),
    "gm_padded_hidden_size": (_i32, [_i32]),
    "gm_graph_workspace_bytes": (_sz, [_i64, _i32]),
    "gm_radius_graph_build": (_i32, [_vp, _i64, _i64, _f64, _i32, _vp, _sz, _vp]),
    "gm_radius_graph_build_batched": (_i32, [_vp, _i64, _i64, _i64, _f64, _i32, _vp, _sz, _vp]),
    "gm_radius_graph_num_edges": (_i32, [_vp, C.POINTER(_i64), _vp]),
    "gm_radius_graph_edges": (_i32, [_vp, _i64, _i32, _vp, _vp, _i64, _vp]),
    "gm_csr_workspace_bytes": (_sz, [_i64, _i64]),
    "gm_csr_from_graph": (_i32, [_vp, _i64, _i32, _vp, _sz, _vp]),
    "gm_csr_from_edge_index": (_i32, [_vp, _i64, _i64, _vp, _sz, _vp]),
    "gm_csr_from_graph_flow": (_i32, [_vp, _i64, _i32, _i32, _vp, _sz, _vp]),
    "gm_csr_from_edge_index_flow": (_i32, [_vp, _i64, _i64, _i32, _vp, _sz, _vp]),
    "gm_csr_num_edges": (_i32, [_vp, C.POINTER(_i64), _vp]),
    "gm_csr_header_status": (_i32, [_vp, C.POINTER(_i64)]),
    "gm_edge_features": (_i32, [_vp, _i64, _vp, _vp, _i64, _f32, _vp, _vp]),
    "gm_edge_features_csr": (_i32, [_vp, _i64, _vp, _i64, _i64, _f32, _vp, _vp]),
    "gm_node_features": (_i32, [_vp, _i64, _FD, _vp, _vp]),
    "gm_integrate": (_i32, [_vp, _vp, _i64, _FD, _vp, _vp]),
    "gm_rigid_rank": (_i32, [_vp, _i64, _FD, _vp, _vp, _vp]),
    "gm_state_pre": (_i32, [_vp, _i64, _FD, _vp, _vp, _vp]),
    "gm_state_post": (_i32, [_vp, _i64, _FD, _vp, _vp, _vp, _vp]),
    "gm_rigid_transform": (_i32, [_vp, _i64, _vp, _i64, C.POINTER(_f32 * 3), _vp, _vp]),
    "gm_model_num_tensors": (_i32, [_MD]),
    "gm_model_create": (_i32, [_MD, C.POINTER(_vp), _i32, _i32, _vp, C.POINTER(_vp)]),
    "gm_model_update": (_i32, [_vp, C.POINTER(_vp), _i32, _i32, _vp]),
    "gm_model_destroy": (None, [_vp]),
    "gm_forward_workspace_bytes": (_sz, [_MD, _i64, _i64]),
    "gm_block_workspace_bytes": (_sz, [_MD, _i64, _i64]),
    "gm_epd_forward": (_i32, [_vp, _vp, _i64, _vp, _i32, _vp, _i64, _vp, _vp, _sz, _vp]),
    "gm_graph_independent_forward": (_i32, [_vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp]),
    "gm_interaction_network_forward": (_i32, [_vp, _i32, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _sz, _vp]),
    "gm_train_tape_bytes": (_sz, [_MD, _i64, _i64]),
    "gm_train_backward_workspace_bytes": (_sz, [_MD, _i64, _i64]),
    "gm_epd_forward_train": (_i32, [_vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _sz, _vp]),
    "gm_epd_backward": (_i32, [_vp, _vp, _i32, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _sz, _vp, _sz, _vp]),
    "gm_block_tape_bytes": (_sz, [_MD, _i32, _i64, _i64]),
    "gm_block_backward_workspace_bytes": (_sz, [_MD, _i64, _i64]),
    "gm_graph_independent_forward_train": (_i32, [_vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _sz, _vp]),
    "gm_graph_independent_backward": (_i32, [_vp, _vp, _i32, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _sz, _vp]),
    "gm_interaction_network_forward_train": (_i32, [_vp, _i32, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _sz, _vp]),
    "gm_interaction_network_backward": (_i32, [_vp, _i32, _vp, _i32, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _sz, _vp]),
    "gm_sinkhorn_workspace_bytes": (_sz, [_i64, _i64]),
    "gm_sinkhorn_divergence": (_i32, [_vp, _i64, _vp, _i64, _f32, _f32, _vp, _vp, _sz, _vp]),
    "gm_sinkhorn_batched_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "gm_sinkhorn_divergence_batched": (_i32, [_vp, _i64, _i64, _vp, _i64, _i32, _f32, _f32, _f32, _vp, _vp, _sz, _vp]),
    "gm_rollout_workspace_bytes": (_sz, [_MD, _i64, _i32]),
    "gm_rollout_step": (_i32, [_vp, _vp, _i64, _FD, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "gm_rollout_renumber_workspace_bytes": (_sz, [_FD, _i64]),
    "gm_rollout": (_i32, [_vp, _vp, _i64, _FD, _i32, _vp, _vp, _i64, _i64, _i64, _vp, _i64, _vp, _sz, _vp, _sz, _vp]),
    "gm_rollout_status": (_i32, [_vp, _MD, _i64, _i32, C.POINTER(_i64), _vp]),
    "gm_model_profile": (_i32, [_vp, _i32]),
    "gm_model_set_edge_kernel": (_i32, [_vp, _i32]),
    "gm_model_profile_query": (_i32, [_vp, _i32, C.POINTER(_i64), C.POINTER(_f64)]),
}

_lib = None


def lib():
    """The loaded library (loads on first use; raises if it was never built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension has not been built "
                "(run `python -m gnn_manip_amd.build`). gnn_manip_amd has no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)  # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(status):
    if status != GM_OK:
        msg = lib().gm_last_error()
        raise GMError(status, msg.decode() if msg else "unknown error")


import threading

_tls = threading.local()   # per host thread: two Python threads may drive two GPUs


def ptr(t):
    """Device (or host) address of a contiguous torch tensor, or None."""
    if t is None:
        return None
    assert t.is_contiguous(), "libgnnmanip_hip needs contiguous tensors"
    if t.is_cuda:
        _tls.device = t.device
    return C.c_void_p(t.data_ptr())


def current_stream(device=None):
    """hipStream_t of torch's current stream on `device` (a torch.device or a CUDA tensor).  Without an argument: the device of
    the last CUDA tensor THIS THREAD handed to ptr() -- the tensors of the call being assembled (the stream argument comes
    last in every entry point); call sites that may pass only host / null pointers give the device explicitly.  The
    library itself switches to the device that owns its pointer arguments (DevGuard in csrc/common.h)."""
    import torch
    if device is not None and not isinstance(device, torch.device):
        device = device.device
    dev = device if device is not None else getattr(_tls, "device", None)
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
