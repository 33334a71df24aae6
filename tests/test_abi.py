"""CPU: the C-ABI library loads and exports every symbol include/gnn_manip_hip.h declares."""
import os
import re

from conftest import ROOT


def test_library_exports_every_declared_symbol():
    from gnn_manip_amd import _lib
    from gnn_manip_amd.build import build
    build()
    header = open(os.path.join(ROOT, "include", "gnn_manip_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(gm_[a-z_0-9]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    handle = _lib.lib()
    for name in declared:
        assert hasattr(handle, name), name
    assert handle.gm_abi_version() == 7


def test_workspace_queries_and_host_side_errors():
    import ctypes as C
    from gnn_manip_amd import _lib
    L = _lib.lib()
    assert L.gm_graph_workspace_bytes(1000, 20) > 1000 * 20 * 4
    assert L.gm_csr_workspace_bytes(1000, 20000) > 3 * 20000 * 4
    d = _lib.ModelDesc(25, 4, 3, 128, 2, 10, 1e-5)
    assert L.gm_model_num_tensors(C.byref(d)) == 182  # 182 tensors
    assert L.gm_forward_workspace_bytes(C.byref(d), 1000, 20000) >= (1000 * 4 + 20000) * 128 * 4
    # argument validation happens before any device call
    bad = _lib.ModelDesc(25, 4, 3, 128, 1, 10, 1e-5)
    out = C.c_void_p()
    arr = (C.c_void_p * 1)()
    rc = L.gm_model_create(C.byref(bad), arr, 1, 0, None, C.byref(out))
    assert rc == -1 and b"num_layers must be at least 2" in L.gm_last_error()
    rc = L.gm_radius_graph_build(None, 3, 10, -1.0, 20, None, 0, None)
    assert rc == -1


def test_product_modules_reject_cpu_tensors():
    import pytest
    import torch
    from gnn_manip_amd import EncProcDecGNN, get_connectivity
    with pytest.raises(RuntimeError):
        get_connectivity(torch.zeros(4, 3), 0.015)
    m = EncProcDecGNN(25, 4, 3, 128, 2, 2)
    with pytest.raises(RuntimeError), torch.no_grad():
        m.forward(torch.zeros(4, 25), torch.zeros(4, 4), torch.zeros(2, 4, dtype=torch.long))


def test_state_dict_layout_matches_reference_fixture(golden):
    import torch
    from gnn_manip_amd import EncProcDecGNN
    g7 = golden("g7_epd_wiring.npz")
    m = EncProcDecGNN(25, 4, 3, 128, 2, 10)
    assert sorted(m.state_dict().keys()) == list(g7["h128.keys"])
    assert sum(p.numel() for p in m.parameters()) == 1591299
    m3 = EncProcDecGNN(25, 4, 3, 64, 3, 2)
    assert sorted(m3.state_dict().keys()) == list(g7["h64_l3_m2.keys"])
