"""GPU parity: the HIP path (through the C ABI) against the golden fixtures and the CPU oracle.

Integer / index results are compared bit-exactly.  Float32 tolerances are written at each check:
the north_star bar is 1e-5 relative on the predicted accelerations.
"""
import numpy as np
import pytest
import torch

from conftest import BOUNDS, CART, CTRL, G1_CASES, MAT, STATS, assert_forward_close
from oracle import epd_oracle as orc

pytestmark = pytest.mark.gpu

KW = dict(stats=STATS, bounds=BOUNDS, conn_r=0.015, cartesian_idx=CART, material_idx=MAT)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _model(params, dims, dev):
    from gnn_manip_amd import EncProcDecGNN
    m = EncProcDecGNN(*dims)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    return m.to(dev)


def _ga():
    from gnn_manip_amd import GraphBoundedMultimaterialControl
    return GraphBoundedMultimaterialControl(0.015, STATS, CART, MAT, CTRL, BOUNDS)


# ------------------------------------------------------------------ K1 radius graph
@pytest.mark.parametrize("case", G1_CASES)
def test_radius_graph_golden_bit_exact(golden, dev, case):
    from gnn_manip_amd import get_connectivity
    g = golden("g1_connectivity.npz")
    r, cap = g[f"{case}.r_cap"]
    s, rcv = get_connectivity(_t(g[f"{case}.pos"], dev), float(r), int(cap))
    assert s.dtype == torch.int64 and rcv.dtype == torch.int64
    assert np.array_equal(s.cpu().numpy(), g[f"{case}.senders"])
    assert np.array_equal(rcv.cpu().numpy(), g[f"{case}.receivers"])


@pytest.mark.parametrize("n,side,seed", [(20000, 0.22, 5), (7000, 0.5, 6), (1, 0.1, 7), (3, 0.001, 8)])
def test_radius_graph_vs_oracle(dev, n, side, seed):
    from gnn_manip_amd import get_connectivity
    rng = np.random.Generator(np.random.PCG64(seed))
    pos = (0.2 + side * rng.random((n, 3))).astype(np.float32)
    s, r = get_connectivity(_t(pos, dev), 0.015, 20)
    so, ro = orc.get_connectivity(pos, 0.015, 20)
    assert np.array_equal(s.cpu().numpy(), so)
    assert np.array_equal(r.cpu().numpy(), ro)


def test_radius_graph_strided_view_and_idempotence(dev, golden):
    from gnn_manip_amd import get_connectivity
    g = golden("g4_features.npz")
    obs = _t(g["obs_a"], dev)
    view = obs[-1][:, 2:5]  # stride 8
    s1, r1 = get_connectivity(view, 0.015)
    s2, r2 = get_connectivity(view.contiguous(), 0.015)
    assert torch.equal(s1, s2) and torch.equal(r1, r2)
    assert np.array_equal(s1.cpu().numpy(), g["proc_senders"])
    assert np.array_equal(r1.cpu().numpy(), g["proc_receivers"])


def test_radius_graph_rejects_nonfinite(dev):
    from gnn_manip_amd import get_connectivity
    from gnn_manip_amd._lib import GMError
    pos = torch.rand(100, 3, device=dev)
    pos[5, 1] = float("nan")
    with pytest.raises(GMError):
        get_connectivity(pos, 0.015)


# ------------------------------------------------------------------ K2 / K3 / K10
@pytest.mark.parametrize("case", G1_CASES)
def test_edge_features_golden(golden, dev, case):
    from gnn_manip_amd import get_edges_displacement
    g = golden("g1_connectivity.npz")
    r = float(g[f"{case}.r_cap"][0])
    ea = get_edges_displacement(_t(g[f"{case}.pos"], dev), _t(g[f"{case}.senders"].astype(np.int64), dev),
                                _t(g[f"{case}.receivers"].astype(np.int64), dev), r)
    np.testing.assert_allclose(ea.cpu().numpy(), g[f"{case}.edge_attr"], rtol=2e-7, atol=1e-7)


def test_node_features_process_collate_golden(golden, dev):
    from gnn_manip_amd import GraphBoundedMultimaterial
    g = golden("g4_features.npz")
    ga = _ga()
    n1 = ga.compute_nodes(_t(g["obs_a"], dev))
    np.testing.assert_allclose(n1.cpu().numpy(), g["nodes_ctrl_a"], rtol=2e-7, atol=1e-7)
    gn = GraphBoundedMultimaterial(0.015, STATS, CART, MAT, BOUNDS)
    n0 = gn.compute_nodes(_t(g["obs_a"][:, :, :5], dev))
    np.testing.assert_allclose(n0.cpu().numpy(), g["nodes_noctrl_a"], rtol=2e-7, atol=1e-7)
    nodes, ea, s, r, tgt = ga.process(_t(g["obs_a"], dev), _t(g["tgt_a"], dev))
    assert np.array_equal(s.cpu().numpy(), g["proc_senders"]) and np.array_equal(r.cpu().numpy(), g["proc_receivers"])
    np.testing.assert_allclose(nodes.cpu().numpy(), g["proc_nodes"], rtol=2e-7, atol=1e-7)
    np.testing.assert_allclose(ea.cpu().numpy(), g["proc_edge_attr"], rtol=2e-7, atol=1e-7)
    np.testing.assert_allclose(tgt.cpu().numpy(), g["proc_tgt"], rtol=1e-5, atol=1e-5)
    batch = [(_t(g["obs_a"], dev), _t(g["tgt_a"], dev)), (_t(g["obs_b"], dev), _t(g["tgt_b"], dev))]
    nodes, ea, ei, tgt = ga.process_collate(batch)
    assert np.array_equal(ei.cpu().numpy(), g["coll_edge_index"])
    np.testing.assert_allclose(nodes.cpu().numpy(), g["coll_nodes"], rtol=2e-7, atol=1e-7)
    np.testing.assert_allclose(ea.cpu().numpy(), g["coll_edge_attr"], rtol=2e-7, atol=1e-7)


def test_integrator_golden_bit_exact(golden, dev):
    from gnn_manip_amd import get_position_from_prediction
    g = golden("g4_features.npz")
    nxt = get_position_from_prediction(STATS, CART, _t(g["pred_acc"], dev), _t(g["obs_a"], dev))
    np.testing.assert_array_equal(nxt.cpu().numpy(), g["next_pos"])


def test_rigid_transform_golden(golden, dev):
    from gnn_manip_amd.planner import get_rigid_body_trajectory
    g = golden("g6_trajectory.npz")
    rp = _t(g["obs_c"][-1, 36:, 2:5], dev)
    out = get_rigid_body_trajectory(g["traj_rot"], g["traj_ty"], 300, [0.5, 0.5, 0.4], rp).cpu().numpy()
    np.testing.assert_allclose(out[g["rigid_traj_steps"]], g["rigid_traj"], rtol=0, atol=2e-7)


# ------------------------------------------------------------------ destination-sorted structure
def test_dst_csr_is_stable_sort_by_destination(dev):
    import ctypes as C
    from gnn_manip_amd.epd_gnn import DstCsr
    rng = np.random.Generator(np.random.PCG64(3))
    n, e = 500, 7000
    ei = rng.integers(0, n, size=(2, e)).astype(np.int64)
    ei[1, :50] = 7  # a long segment
    csr = DstCsr(_t(ei, dev), n)
    assert csr.validate() == e
    raw = csr.ws.cpu().numpy()
    # parse the workspace the same way the library carves it (256-byte aligned arrays)
    def take(off, count):
        off = (off + 255) // 256 * 256
        return np.frombuffer(raw[off:off + 4 * count].tobytes(), dtype=np.int32), off + 4 * count
    off = 16
    in_ptr, off = take(off, n + 1)
    _, off = take(off, n + 1)
    dst, off = take(off, e)
    src, off = take(off, e)
    eid, off = take(off, e)
    order = np.argsort(ei[1], kind="stable")
    assert np.array_equal(eid, order)
    assert np.array_equal(dst, ei[1][order]) and np.array_equal(src, ei[0][order])
    assert np.array_equal(in_ptr, np.r_[0, np.cumsum(np.bincount(ei[1], minlength=n))])


def test_bad_edge_index_raises(dev):
    from gnn_manip_amd._lib import GMError
    from gnn_manip_amd.epd_gnn import DstCsr
    ei = torch.tensor([[0, 1, 2], [1, 2, 9]], dtype=torch.int64, device=dev)
    with pytest.raises(GMError):
        DstCsr(ei, 3).validate()
    # the fused forward does not synchronise to check; the bad edge is left out and reported by status()
    from gnn_manip_amd import EncProcDecGNN
    m = EncProcDecGNN(25, 4, 3, 128, 2, 2).to(dev)
    with torch.no_grad():
        out = m.forward(torch.randn(3, 25, device=dev), torch.randn(3, 4, device=dev), ei)
        assert torch.isfinite(out).all()
        # a caller that never asks: the error of the previous forward surfaces at the start of the next one
        with pytest.raises(GMError, match="out of range"):
            m.forward(torch.zeros(3, 25, device=dev), torch.zeros(2, 4, device=dev), ei[:, :2].contiguous())
        good = m.forward(torch.zeros(3, 25, device=dev), torch.zeros(2, 4, device=dev), ei[:, :2].contiguous())
    assert torch.isfinite(good).all()
    assert m.status() == 2          # the valid forward
    with torch.no_grad():
        m.forward(torch.zeros(3, 25, device=dev), torch.zeros(3, 4, device=dev), ei)
    with pytest.raises(GMError, match="out of range"):
        m.status()
    # the check never blocks: a loop may run ahead of the GPU (more forwards than the engine has pinned header rows), and an
    # error inside the loop is reported by a later forward or by status() at the latest
    with torch.no_grad():
        for _ in range(11):
            m.forward(torch.zeros(3, 25, device=dev), torch.zeros(2, 4, device=dev), ei[:, :2].contiguous())
    assert m.status() == 2 and not m._watched
    with torch.no_grad(), pytest.raises(GMError, match="out of range"):
        m.forward(torch.zeros(3, 25, device=dev), torch.zeros(3, 4, device=dev), ei)
        for _ in range(11):
            m.forward(torch.zeros(3, 25, device=dev), torch.zeros(2, 4, device=dev), ei[:, :2].contiguous())
        m.status()
    assert m.status() in (0, 2)     # reported once
    # opting out restores the fire-and-forget behaviour: nothing is checked until status()
    m.auto_status = False
    with torch.no_grad():
        m.forward(torch.zeros(3, 25, device=dev), torch.zeros(3, 4, device=dev), ei)
        good = m.forward(torch.zeros(3, 25, device=dev), torch.zeros(2, 4, device=dev), ei[:, :2].contiguous())
    assert m.status() == 2


# ------------------------------------------------------------------ K4-K9 model
def _scene_graph(golden):
    g4 = golden("g4_features.npz")
    nodes, ea, s, r, _ = orc.process(g4["obs_a"], None, control_idx=CTRL, **KW)
    return nodes, ea, np.stack((s, r))


def test_graph_independent_block_vs_golden(golden, dev):
    g7 = golden("g7_epd_wiring.npz")
    params = orc.init_params(25, 4, 3, 128, 2, 10, 41)
    m = _model(params, (25, 4, 3, 128, 2, 10), dev)
    nodes, ea, ei = _scene_graph(golden)
    with torch.no_grad():
        h0, e0, _ = m.encoder(_t(nodes, dev), _t(ea, dev), _t(ei, dev))
    np.testing.assert_allclose(h0.cpu().numpy(), g7["h128.h0"], rtol=1e-5, atol=3e-6)
    np.testing.assert_allclose(e0.cpu().numpy()[:64], g7["h128.e0_head"], rtol=1e-5, atol=3e-6)
    ho, eo = orc.graph_independent(params, "encoder", nodes, ea, 2)
    np.testing.assert_allclose(e0.cpu().numpy(), eo, rtol=1e-5, atol=3e-6)


def test_interaction_network_block_vs_golden(golden, dev):
    g7 = golden("g7_epd_wiring.npz")
    params = orc.init_params(25, 4, 3, 128, 2, 10, 41)
    m = _model(params, (25, 4, 3, 128, 2, 10), dev)
    nodes, ea, ei = _scene_graph(golden)
    h0, e0 = orc.graph_independent(params, "encoder", nodes, ea, 2)
    with torch.no_grad():
        h1, e1, _ = m.processor[0](_t(h0, dev), _t(e0, dev), _t(ei, dev))
    h1o, e1o = orc.interaction_network(params, "processor.0", h0, e0, ei, 2)
    np.testing.assert_allclose(e1.cpu().numpy(), e1o, rtol=1e-5, atol=5e-6)
    np.testing.assert_allclose(h1.cpu().numpy(), h1o, rtol=1e-5, atol=5e-6)
    np.testing.assert_allclose(h1.cpu().numpy(), g7["h128.h1"], rtol=1e-5, atol=5e-6)
    np.testing.assert_allclose(e1.cpu().numpy()[:64], g7["h128.e1_head"], rtol=1e-5, atol=5e-6)


def test_epd_forward_golden(golden, dev):
    g7 = golden("g7_epd_wiring.npz")
    params = orc.init_params(25, 4, 3, 128, 2, 10, 41)
    m = _model(params, (25, 4, 3, 128, 2, 10), dev)
    nodes, ea, ei = _scene_graph(golden)
    with torch.no_grad():
        out = m.forward(_t(nodes, dev), _t(ea, dev), _t(ei, dev)).cpu().numpy()
    ref = g7["h128.out"]
    assert_forward_close(out, ref)  # north_star: 1e-5 relative fp32, per element


@pytest.mark.parametrize("n,side,seed", [(3000, 0.11, 61), (130, 0.3, 62), (1, 0.1, 63)])
def test_epd_forward_vs_oracle(dev, n, side, seed):
    """Bigger / ragged graphs: several tiles, segments crossing tile and wave boundaries, isolated nodes."""
    from gnn_manip_amd import scene
    obs = scene.make_scene(n, seed=seed, side=side)
    params = orc.init_params(25, 4, 3, 128, 2, 10, seed)
    m = _model(params, (25, 4, 3, 128, 2, 10), dev)
    nodes, ea, s, r, _ = orc.process(obs, None, control_idx=CTRL, **KW)
    ei = np.stack((s, r))
    with torch.no_grad():
        out = m.forward(_t(nodes, dev), _t(ea, dev), _t(ei, dev)).cpu().numpy()
    ref = orc.epd_forward(params, nodes, ea, ei, 2, 10)
    assert_forward_close(out, ref, floor=1e-3)


@pytest.mark.parametrize("n,side,seed", [(700, 0.075, 65), (129, 0.2, 66)])
def test_epd_forward_hidden_256_vs_oracle(dev, n, side, seed):
    """BASELINE config C4 geometry (hidden=256): one-wave-per-SIMD instantiation of the same kernels."""
    from gnn_manip_amd import scene
    obs = scene.make_scene(n, seed=seed, side=side)
    params = orc.init_params(25, 4, 3, 256, 2, 10, seed)
    m = _model(params, (25, 4, 3, 256, 2, 10), dev)
    nodes, ea, s, r, _ = orc.process(obs, None, control_idx=CTRL, **KW)
    ei = np.stack((s, r))
    with torch.no_grad():
        out = m.forward(_t(nodes, dev), _t(ea, dev), _t(ei, dev)).cpu().numpy()
        h0, e0, _ = m.encoder(_t(nodes, dev), _t(ea, dev), _t(ei, dev))
        h1, e1, _ = m.processor[0](h0, e0, _t(ei, dev))
    ho, eo = orc.graph_independent(params, "encoder", nodes, ea, 2)
    np.testing.assert_allclose(h0.cpu().numpy(), ho, rtol=1e-5, atol=3e-6)
    np.testing.assert_allclose(e0.cpu().numpy(), eo, rtol=1e-5, atol=3e-6)
    h1o, e1o = orc.interaction_network(params, "processor.0", ho, eo, ei, 2)
    np.testing.assert_allclose(e1.cpu().numpy(), e1o, rtol=1e-5, atol=5e-6)
    np.testing.assert_allclose(h1.cpu().numpy(), h1o, rtol=1e-5, atol=5e-6)
    ref = orc.epd_forward(params, nodes, ea, ei, 2, 10)
    assert_forward_close(out, ref, floor=1e-3)


def test_unsupported_sizes_fail_loudly(dev):
    from gnn_manip_amd import EncProcDecGNN
    from gnn_manip_amd._lib import GMError
    m = EncProcDecGNN(25, 4, 3, 320, 2, 2).to(dev)   # hidden sizes up to 256 run (zero-padded to 64 / 128 / 256); larger ones do not
    with pytest.raises(GMError, match="hidden_size=320"), torch.no_grad():
        m.forward(torch.zeros(4, 25, device=dev), torch.zeros(4, 4, device=dev),
                  torch.zeros(2, 4, dtype=torch.long, device=dev))
    m = EncProcDecGNN(25, 4, 3, 64, 3, 2).to(dev)   # runs on the streamed kernels; the systolic one is for hidden 128 / num_layers 2
    with pytest.raises(GMError, match="systolic kernel is for hidden_size 128"), torch.no_grad():
        m.set_edge_kernel("sys")
        m.forward(torch.zeros(4, 25, device=dev), torch.zeros(4, 4, device=dev),
                  torch.zeros(2, 4, dtype=torch.long, device=dev))
    # the round-1 fp32 / bf16 x 6 kernels (choices 1 .. 4) are gone from the library
    m = EncProcDecGNN(25, 4, 3, 128, 2, 2).to(dev)
    with pytest.raises(GMError, match="removed from the library"), torch.no_grad():
        m.set_edge_kernel(2)        # takes effect when the handle is built: at the first forward
        m.forward(torch.zeros(4, 25, device=dev), torch.zeros(4, 4, device=dev), torch.zeros(2, 4, dtype=torch.long, device=dev))


def test_epd_forward_permutation_of_edges_is_immaterial(dev):
    """Property: the result does not depend on the caller's edge order (destination sort + eid indirection),
    and a hub node with in-degree > 128 (a segment spanning whole tiles) aggregates correctly."""
    rng = np.random.Generator(np.random.PCG64(71))
    n, e = 400, 6000
    ei = rng.integers(0, n, size=(2, e)).astype(np.int64)
    ei[1, :700] = 11
    nodes = rng.standard_normal((n, 25)).astype(np.float32)
    ea = rng.standard_normal((e, 4)).astype(np.float32)
    params = orc.init_params(25, 4, 3, 128, 2, 3, 72)
    m = _model(params, (25, 4, 3, 128, 2, 3), dev)
    with torch.no_grad():
        out1 = m.forward(_t(nodes, dev), _t(ea, dev), _t(ei, dev)).cpu().numpy()
        perm = rng.permutation(e)
        out2 = m.forward(_t(nodes, dev), _t(ea[perm], dev), _t(ei[:, perm], dev)).cpu().numpy()
    ref = orc.epd_forward(params, nodes, ea, ei, 2, 3)
    scale = np.abs(ref).max()
    assert np.abs(out1 - ref).max() <= 2e-5 * scale
    assert np.abs(out1 - out2).max() <= 2e-5 * scale


def test_weight_update_repacks(dev, golden):
    params = orc.init_params(25, 4, 3, 128, 2, 2, 81)
    m = _model(params, (25, 4, 3, 128, 2, 2), dev)
    nodes, ea, ei = _scene_graph(golden)
    with torch.no_grad():
        a = m.forward(_t(nodes, dev), _t(ea, dev), _t(ei, dev)).cpu().numpy()
        m.decoder[4].bias.add_(1.0)
        b = m.forward(_t(nodes, dev), _t(ea, dev), _t(ei, dev)).cpu().numpy()
    np.testing.assert_allclose(b - a, 1.0, atol=1e-5)


# ------------------------------------------------------------------ rollout
def test_rollout_golden_g8(golden, dev):
    from gnn_manip_amd import RolloutEngine
    from gnn_manip_amd.planner import get_rigid_body_trajectory
    g = golden("g8_rollout.npz")
    nd, ed, od, hid, nl, ms, seed, horizon = [int(v) for v in g["cfg"]]
    params = orc.init_params(nd, ed, od, hid, nl, ms, seed)
    m = _model(params, (nd, ed, od, hid, nl, ms), dev)
    obs0 = g["obs0"]
    rigid = obs0[-1, :, 1] == 1
    traj = get_rigid_body_trajectory(g["traj_rot"], g["traj_ty"], horizon, [0.5, 0.5, 0.4], _t(obs0[-1][rigid][:, 2:5], dev))
    eng = RolloutEngine(m, _ga(), obs0.shape[1], device=dev)
    with torch.no_grad():
        final, recs = eng.rollout(_t(obs0, dev), traj, horizon=horizon, record=True)
    final, recs = final.cpu().numpy(), recs.cpu().numpy()
    np.testing.assert_allclose(final[-1][~rigid][:, 2:5], g["end_coffee"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(recs[:, ~rigid][:, :, 2:5], g["coffee_states"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(recs[:, rigid][:, :, 2:5], g["cup_states"], rtol=0, atol=5e-6)


def test_rollout_vs_oracle_whole_state(dev):
    from gnn_manip_amd import RolloutEngine, scene
    n, steps = 2500, 3
    obs = scene.make_scene(n, seed=91, side=0.1)
    traj = scene.rigid_drift_trajectory(obs, steps - 1)  # last step has no scripted pose (traj_utils.py:130-131)
    params = orc.init_params(25, 4, 3, 128, 2, 10, 92)
    m = _model(params, (25, 4, 3, 128, 2, 10), dev)
    eng = RolloutEngine(m, _ga(), n, device=dev)
    with torch.no_grad():
        final = eng.rollout(_t(obs, dev), _t(traj, dev), horizon=steps).cpu().numpy()
    ref = orc.rollout(params, obs, traj, steps, STATS, BOUNDS, 0.015, CART, MAT, CTRL)
    np.testing.assert_allclose(final[:, :, 2:5], ref[:, :, 2:5], rtol=0, atol=5e-6)
    np.testing.assert_allclose(final[:, :, 5:8], ref[:, :, 5:8], rtol=0, atol=5e-6)
    np.testing.assert_array_equal(final[:, :, :2], ref[:, :, :2])
    assert eng.status() > 0


@pytest.mark.parametrize("n,side,seed", [(3, 0.01, 301), (40, 0.03, 302), (33, 0.4, 303), (1000, 0.35, 304), (777, 0.06, 305)])
def test_rollout_small_and_ragged_scenes(dev, n, side, seed):
    """The systolic kernels (edge encoder + processor edge MLP of the rollout path) on edge lists that are not whole blocks:
    a handful of particles (one partial block, most workgroups idle), isolated particles (self edges only, every segment one
    row long), a sparse scene, and a dense one whose last block is ragged -- two steps against the oracle's rollout."""
    from gnn_manip_amd import RolloutEngine, scene
    obs = scene.make_scene(n, seed=seed, side=side)
    traj = scene.rigid_drift_trajectory(obs, 2)
    params = orc.init_params(25, 4, 3, 128, 2, 3, seed)
    m = _model(params, (25, 4, 3, 128, 2, 3), dev)
    eng = RolloutEngine(m, _ga(), n, device=dev)
    with torch.no_grad():
        final = eng.rollout(_t(obs, dev), _t(traj, dev), horizon=2).cpu().numpy()
    ref = orc.rollout(params, obs, traj, 2, STATS, BOUNDS, 0.015, CART, MAT, CTRL, 2, 3)
    assert np.isfinite(final).all() and eng.status() >= n   # at least the self edges
    np.testing.assert_allclose(final[:, :, 2:8], ref[:, :, 2:8], rtol=0, atol=5e-6)


def test_renumbered_rollout_matches_the_plain_one_and_the_oracle(dev):
    """RolloutEngine(renumber=True) runs the rollout on a copy of the state in grid-cell order (default for large scenes), re-sorted
    every RENUMBER_EVERY steps, and returns it in the caller's numbering: same graph every step (edge count), the rigid rows follow THEIR scripted poses, the
    per-step record comes back row for row, and the result is the plain engine's up to the summation order of a node's
    messages -- both within the oracle bound.  Scrambled particle ids, so the renumbering really moves every row; also as a
    batch of candidates (each scene keeps its own block of rows) and bit-stable from run to run."""
    from gnn_manip_amd import RolloutEngine, scene
    n, steps, b = 1500, 3, 2
    obs = scene.make_scene(n, seed=311, side=0.09)
    order = np.random.Generator(np.random.PCG64(312)).permutation(n)
    obs = np.ascontiguousarray(obs[:, order])                 # rigid rows scattered over the ids
    rigid = obs[-1, :, 1] == 1
    traj = scene.rigid_drift_trajectory(obs, steps, seed=313, step_size=3e-4)
    params = orc.init_params(25, 4, 3, 128, 2, 10, 314)
    m = _model(params, (25, 4, 3, 128, 2, 10), dev)
    with torch.no_grad():
        plain = RolloutEngine(m, _ga(), n, device=dev, renumber=False)
        f0, r0 = plain.rollout(_t(obs, dev), _t(traj, dev), horizon=steps, record=True)
        e0 = plain.status()
        ren = RolloutEngine(m, _ga(), n, device=dev, renumber=True)
        ren.RENUMBER_EVERY = 2                                # three steps = a chunk of two and one of one: the re-sort path too
        f1, r1 = ren.rollout(_t(obs, dev), _t(traj, dev), horizon=steps, record=True)
        e1 = ren.status()
        f2, r2 = ren.rollout(_t(obs, dev), _t(traj, dev), horizon=steps, record=True)
    assert ren.renumber and not plain.renumber
    assert not torch.equal(f1, f0)       # the rows really moved: another summation order somewhere, not the plain loop again
    assert e0 == e1 and ren.n_rigid == plain.n_rigid == int(rigid.sum())
    assert torch.equal(f1, f2) and torch.equal(r1, r2)
    f0, r0, f1, r1 = (x.cpu().numpy() for x in (f0, r0, f1, r1))
    np.testing.assert_array_equal(f1[:, :, :2], obs[:, :, :2])                       # ids and materials stay on their rows
    np.testing.assert_array_equal(f1[-1, rigid, 2:5], traj[steps - 1])               # every rigid row got its own pose
    np.testing.assert_allclose(f1[:, :, 2:8], f0[:, :, 2:8], rtol=0, atol=2e-6)
    np.testing.assert_allclose(r1[:, :, 2:8], r0[:, :, 2:8], rtol=0, atol=2e-6)
    ref = orc.rollout(params, obs, traj, steps, STATS, BOUNDS, 0.015, CART, MAT, CTRL)
    np.testing.assert_allclose(f1[:, :, 2:8], ref[:, :, 2:8], rtol=0, atol=5e-6)
    trajs = np.stack([traj, scene.rigid_drift_trajectory(obs, steps, seed=315, step_size=3e-4)])
    with torch.no_grad():
        eng_b = RolloutEngine(m, _ga(), n, device=dev, candidates=b, renumber=True)
        eng_b.RENUMBER_EVERY = 2
        out = eng_b.rollout_candidates(_t(obs, dev), _t(trajs, dev)).cpu().numpy()
        one = ren.rollout(_t(obs, dev), _t(trajs[1], dev), horizon=steps).cpu().numpy()
        # a horizon beyond the scripted poses (the rigid body then stays where it is: traj_utils.py:126-134), cut by a chunk boundary
        long_p = plain.rollout(_t(obs, dev), _t(traj[:1], dev), horizon=steps).cpu().numpy()
        long_r = ren.rollout(_t(obs, dev), _t(traj[:1], dev), horizon=steps).cpu().numpy()
    np.testing.assert_allclose(long_r[:, :, 2:8], long_p[:, :, 2:8], rtol=0, atol=2e-6)
    assert np.array_equal(out[0], f1) and np.array_equal(out[1], one)
    assert RolloutEngine(m, _ga(), RolloutEngine.RENUMBER_MIN_NODES, device=dev).renumber and not RolloutEngine(m, _ga(), 5000, device=dev).renumber


def test_gm_rollout_renumbers_behind_the_c_abi(dev):
    """gm_rollout(..., renumber_every, renumber_ws, ...) driven raw through ctypes -- pointers, sizes and a stream, no RolloutEngine --
    gives the bits of RolloutEngine(renumber=True): the cell order, the row maps, the gathers and the write-back in the caller's
    numbering are all behind the C ABI (include/gnn_manip_hip.h; the loop it replaces: rollout_utils.py:38-61).  With
    renumber_every = 0 and no renumber workspace the same entry point is the plain loop."""
    import ctypes as C
    from gnn_manip_amd import RolloutEngine, scene
    from gnn_manip_amd._lib import ModelDesc, check, lib
    n, steps, every = 1500, 5, 2
    obs = scene.make_scene(n, seed=321, side=0.09)
    obs = np.ascontiguousarray(obs[:, np.random.Generator(np.random.PCG64(322)).permutation(n)])
    traj = scene.rigid_drift_trajectory(obs, steps, seed=323, step_size=3e-4)
    params = orc.init_params(25, 4, 3, 128, 2, 3, 324)
    m = _model(params, (25, 4, 3, 128, 2, 3), dev)
    with torch.no_grad():
        ren = RolloutEngine(m, _ga(), n, device=dev, renumber=True)
        ren.RENUMBER_EVERY = every
        want, want_rec = ren.rollout(_t(obs, dev), _t(traj, dev), horizon=steps, record=True)
        plain = RolloutEngine(m, _ga(), n, device=dev, renumber=False)
        want_plain = plain.rollout(_t(obs, dev), _t(traj, dev), horizon=steps)
    L = lib()
    fdesc, mdesc = ren.fdesc, ModelDesc(*m.model_desc())
    handle = m.device_handle(dev)
    u8 = lambda nbytes: torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
    ws = u8(L.gm_rollout_workspace_bytes(C.byref(mdesc), n, 20))
    rws = u8(L.gm_rollout_renumber_workspace_bytes(C.byref(fdesc), n))
    rws.fill_(0xff)                       # whatever the scratch held
    state, t_dev = _t(obs, dev).clone(), _t(traj, dev)
    rank = torch.empty(n, dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    check(L.gm_rigid_rank(p(state), n, C.byref(fdesc), p(rank), p(cnt), stream))
    n_rigid = int(cnt.item())
    rec = torch.empty((steps, n, obs.shape[2]), dtype=torch.float32, device=dev)
    check(L.gm_rollout(handle, p(state), n, C.byref(fdesc), 20, p(rank), p(t_dev), steps, n_rigid, steps, p(rec), every, p(rws), rws.numel(),
                       p(ws), ws.numel(), stream))
    torch.cuda.synchronize()
    assert torch.equal(state, want) and torch.equal(rec, want_rec)
    state2 = _t(obs, dev).clone()
    check(L.gm_rollout(handle, p(state2), n, C.byref(fdesc), 20, p(rank), p(t_dev), steps, n_rigid, steps, None, 0, None, 0, p(ws), ws.numel(), stream))
    torch.cuda.synchronize()
    assert torch.equal(state2, want_plain) and not torch.equal(state2, state)
    # argument errors come back as status codes, before any launch
    assert L.gm_rollout(handle, p(state2), n, C.byref(fdesc), 20, p(rank), p(t_dev), steps, n_rigid, steps, None, every, None, 0, p(ws), ws.numel(), stream) != 0
    assert L.gm_rollout(handle, p(state2), n, C.byref(fdesc), 20, p(rank), p(t_dev), steps, n_rigid, steps, None, every, p(rws), 16, p(ws), ws.numel(), stream) != 0


# ------------------------------------------------------------------ batches of scenes (candidates)
def test_batched_radius_graph_is_block_diagonal(dev, golden):
    """collate_utils.py:68-87: a batch is the graphs side by side, indices offset by N*i -- no cross edges."""
    from gnn_manip_amd import get_connectivity
    g = golden("g4_features.npz")
    pa = g["obs_a"][-1, :, 2:5]
    rng = np.random.Generator(np.random.PCG64(5))
    pb = (pa + 1e-3 * rng.standard_normal(pa.shape)).astype(np.float32)   # overlapping in space, different graph
    pc = (pa[::-1] + np.float32(0.2)).copy()
    n = pa.shape[0]
    s, r = get_connectivity(_t(np.concatenate((pa, pb, pc)), dev), 0.015, 20, nodes_per_graph=n)
    ss, rr = [], []
    for i, p in enumerate((pa, pb, pc)):
        so, ro = orc.get_connectivity(p, 0.015, 20)
        ss.append(so + n * i)
        rr.append(ro + n * i)
    assert np.array_equal(s.cpu().numpy(), np.concatenate(ss))
    assert np.array_equal(r.cpu().numpy(), np.concatenate(rr))


def test_candidate_batched_rollout_matches_independent_rollouts(dev):
    """Candidate c of a batch == candidate c rolled out alone, bit for bit, with the raw (undamped) random decoder: every
    per-row result is independent of its tile, and the scatter-add's blocks / chunks are laid out per graph
    (build_edge_blocks), so the partial sums of a graph do not depend on what else shares the launch."""
    from gnn_manip_amd import RolloutEngine, scene
    n, steps, b = 700, 4, 3
    obs = scene.make_scene(n, seed=95, side=0.075)
    trajs = np.stack([scene.rigid_drift_trajectory(obs, steps, seed=100 + c, step_size=3e-4) for c in range(b)])
    params = orc.init_params(25, 4, 3, 128, 2, 10, 96)
    m = _model(params, (25, 4, 3, 128, 2, 10), dev)
    for kernel in ("auto", "hm", "sys_all"):
        m.set_edge_kernel(kernel)
        with torch.no_grad():
            eng_b = RolloutEngine(m, _ga(), n, device=dev, candidates=b)
            out = eng_b.rollout_candidates(_t(obs, dev), _t(trajs, dev)).cpu().numpy()
            eng_1 = RolloutEngine(m, _ga(), n, device=dev)
            for c in range(b):
                one = eng_1.rollout(_t(obs, dev), _t(trajs[c], dev), horizon=steps).cpu().numpy()
                assert np.array_equal(out[c], one), (kernel, c, np.abs(out[c] - one).max())
    ref = orc.rollout(params, obs, trajs[1], 2, STATS, BOUNDS, 0.015, CART, MAT, CTRL)
    with torch.no_grad():
        two = eng_1.rollout(_t(obs, dev), _t(trajs[1], dev), horizon=2).cpu().numpy()
    np.testing.assert_allclose(two[:, :, 2:5], ref[:, :, 2:5], rtol=0, atol=5e-6)


@pytest.mark.parametrize("choice,name", [(5, "systolic fp16 x 3"), (6, "streamed fp16 x 3"), (7, "systolic edge, node and projection kernels")])
def test_every_processor_edge_kernel_form_vs_oracle(dev, choice, name):
    """Each selectable form of the processor edge kernel (EncProcDecGNN.set_edge_kernel, a per-model option) on a
    multi-tile graph, a ragged small one and a single node: same 1e-5 bar against the oracle."""
    from gnn_manip_amd import scene
    for n, side, seed in ((3000, 0.11, 61), (130, 0.3, 62), (1, 0.1, 63), (333, 0.05, 64)):
        obs = scene.make_scene(n, seed=seed, side=side)
        params = orc.init_params(25, 4, 3, 128, 2, 10, seed)
        m = _model(params, (25, 4, 3, 128, 2, 10), dev)
        m.set_edge_kernel(choice)
        nodes, ea, s, r, _ = orc.process(obs, None, control_idx=CTRL, **KW)
        ei = np.stack((s, r))
        with torch.no_grad():
            out = m.forward(_t(nodes, dev), _t(ea, dev), _t(ei, dev)).cpu().numpy()
        ref = orc.epd_forward(params, nodes, ea, ei, 2, 10)
        assert_forward_close(out, ref, floor=1e-3, what=(name, n))


@pytest.mark.parametrize("seed", [71, 72, 73, 74, 75])
@pytest.mark.parametrize("choice", [5, 6, 7])
def test_split_operand_kernels_are_as_accurate_as_float32(dev, choice, seed):
    """The matrix-pipe forms compute fp32 results (three exact partial products of two-way fp16 operand splits with pre-scaled
    weights, fp32 accumulation): against a float64 evaluation of the same model their error must be of the order of a plain
    float32 evaluation's, not of a reduced-precision one -- over a run of seeds (scene and weights), none hand-picked; the same
    at the target size: tests/test_gpu_fullsize.py."""
    from gnn_manip_amd import scene
    from oracle import torch_epd
    obs = scene.make_scene(2500, seed=seed, side=0.1)
    params = orc.init_params(25, 4, 3, 128, 2, 10, seed)
    m = _model(params, (25, 4, 3, 128, 2, 10), dev)
    nodes, ea, s, r, _ = orc.process(obs, None, control_idx=CTRL, **KW)
    ei = np.stack((s, r))
    p64 = {k: torch.tensor(v, dtype=torch.float64) for k, v in params.items()}
    ref = torch_epd.epd_forward(p64, torch.tensor(nodes, dtype=torch.float64), torch.tensor(ea, dtype=torch.float64),
                                torch.tensor(ei), 2, 10).numpy()
    f32 = orc.epd_forward(params, nodes, ea, ei, 2, 10)
    err32 = np.abs(f32 - ref).max() / np.abs(ref).max()
    m.set_edge_kernel(choice)
    with torch.no_grad():
        out = m.forward(_t(nodes, dev), _t(ea, dev), _t(ei, dev)).cpu().numpy()
    err = np.abs(out - ref).max() / np.abs(ref).max()
    assert err <= max(2.5 * err32, 2.5e-6), (err, err32)


# ------------------------------------------------------------------ a9: the block-convention switch
CONVENTIONS = [
    dict(flow="target_to_source"),
    dict(concat=("e", "j", "i"), node_concat=("agg", "h")),
    dict(flow="target_to_source", concat=("j", "e", "i")),
]


@pytest.mark.parametrize("conv", CONVENTIONS)
def test_block_convention_switch_forward_block_and_rollout(dev, conv):
    """torch_graphnet's source is absent from the reference: the aggregation row (flow) and the concat orders of the
    InteractionNetwork are selectable.  Each choice must equal the oracle restated with the same choice -- in the fused
    forward (systolic edge kernel), in the standalone block and in the device-resident rollout (graph -> sort path)."""
    from gnn_manip_amd import EncProcDecGNN, RolloutEngine, scene
    n = 900
    obs = scene.make_scene(n, seed=95, side=0.075)
    params = orc.init_params(25, 4, 3, 128, 2, 3, 96)
    m = EncProcDecGNN(25, 4, 3, 128, 2, 3, **conv)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    m = m.to(dev)
    nodes, ea, s, r, _ = orc.process(obs, None, control_idx=CTRL, **KW)
    ei = np.stack((s, r))
    with torch.no_grad():
        out = m.forward(_t(nodes, dev), _t(ea, dev), _t(ei, dev)).cpu().numpy()
    ref = orc.epd_forward(params, nodes, ea, ei, 2, 3, **conv)
    dflt = orc.epd_forward(params, nodes, ea, ei, 2, 3)
    assert_forward_close(out, ref)
    assert np.abs(ref - dflt).max() > 1e-3 * np.abs(ref).max()   # the conventions really differ on this (asymmetric) graph
    # standalone block
    h0, e0 = orc.graph_independent(params, "encoder", nodes, ea, 2)
    with torch.no_grad():
        h1, e1, _ = m.processor[0](_t(h0, dev), _t(e0, dev), _t(ei, dev))
    h1o, e1o = orc.interaction_network(params, "processor.0", h0, e0, ei, 2, **conv)
    np.testing.assert_allclose(e1.cpu().numpy(), e1o, rtol=1e-5, atol=5e-6)
    np.testing.assert_allclose(h1.cpu().numpy(), h1o, rtol=1e-5, atol=5e-6)
    # rollout: the destination sort of the radius graph follows the flow
    traj = scene.rigid_drift_trajectory(obs, 2)
    eng = RolloutEngine(m, _ga(), n, device=dev)
    with torch.no_grad():
        final = eng.rollout(_t(obs, dev), _t(traj, dev), horizon=2).cpu().numpy()
    fwd = lambda nn, ee, ii: orc.epd_forward(params, nn, ee, ii, 2, 3, **conv)
    refs = orc.rollout(params, obs, traj, 2, STATS, BOUNDS, 0.015, CART, MAT, CTRL, forward_fn=fwd)
    np.testing.assert_allclose(final[:, :, 2:5], refs[:, :, 2:5], rtol=0, atol=5e-6)


def test_block_convention_switch_trains(dev):
    """The training path packs the same column blocks: loss gradients of a non-default convention against autograd through
    a plain-torch restatement of that convention."""
    import torch.nn.functional as F
    from gnn_manip_amd import EncProcDecGNN, scene
    conv = dict(flow="target_to_source", concat=("e", "j", "i"), node_concat=("agg", "h"))
    n = 300
    obs = scene.make_scene(n, seed=97, side=0.06)
    params = orc.init_params(25, 4, 3, 128, 2, 2, 98)
    nodes, ea, s, r, _ = orc.process(obs, None, control_idx=CTRL, **KW)
    ei = np.stack((s, r))
    m = EncProcDecGNN(25, 4, 3, 128, 2, 2, **conv)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    m = m.to(dev)
    out = m.forward(_t(nodes, dev), _t(ea, dev), _t(ei, dev))
    out.abs().sum().backward()
    # float64 torch restatement of the same convention
    p = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in params.items()}
    from oracle import torch_epd
    x, e_t, idx = torch.tensor(nodes, dtype=torch.float64), torch.tensor(ea, dtype=torch.float64), torch.tensor(ei)
    i, j = idx[0], idx[1]
    h = torch_epd.mlp(p, "encoder.phi_node", x, 2, True)
    e = torch_epd.mlp(p, "encoder.phi_edge", e_t, 2, True)
    for k in range(2):
        en = torch_epd.mlp(p, f"processor.{k}.phi_edge", torch.cat((e, h[j], h[i]), dim=1), 2, True)
        agg = torch.zeros_like(h).index_add_(0, i, en)
        hn = torch_epd.mlp(p, f"processor.{k}.phi_node", torch.cat((agg, h), dim=1), 2, True)
        h, e = h + hn, e + en
    ref = torch_epd.mlp(p, "decoder", h, 2, False)
    ref.abs().sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-4, atol=1e-5)
    for name, prm in m.named_parameters():
        g, gr = prm.grad.cpu().numpy(), p[name].grad.numpy()
        assert np.abs(g - gr).max() <= 2e-3 * max(np.abs(gr).max(), 1e-6), name


# ------------------------------------------------------------------ a6: any hidden_size / num_layers the reference accepts
def test_epd_h64_l3_golden(golden, dev):
    """The reference's own wiring at hidden 64, num_layers 3, m_steps 2 (train_dyn.py:237-238 exposes both options;
    epd_gnn.py:72-84 builds num_layers - 1 hidden Linears for any value): encoder, first processor block and the full
    forward against the fixture the reference produced."""
    g7 = golden("g7_epd_wiring.npz")
    nd, ed, od, hid, nl, ms, seed = [int(v) for v in g7["h64_l3_m2.cfg"]]
    assert (hid, nl, ms) == (64, 3, 2)
    params = orc.init_params(nd, ed, od, hid, nl, ms, seed)
    m = _model(params, (nd, ed, od, hid, nl, ms), dev)
    nodes, ea, ei = _scene_graph(golden)
    with torch.no_grad():
        h0, e0, _ = m.encoder(_t(nodes, dev), _t(ea, dev), _t(ei, dev))
        h1, e1, _ = m.processor[0](h0, e0, _t(ei, dev))
        out = m.forward(_t(nodes, dev), _t(ea, dev), _t(ei, dev)).cpu().numpy()
    np.testing.assert_allclose(h0.cpu().numpy(), g7["h64_l3_m2.h0"], rtol=1e-5, atol=3e-6)
    np.testing.assert_allclose(e0.cpu().numpy()[:64], g7["h64_l3_m2.e0_head"], rtol=1e-5, atol=3e-6)
    np.testing.assert_allclose(h1.cpu().numpy(), g7["h64_l3_m2.h1"], rtol=1e-5, atol=5e-6)
    np.testing.assert_allclose(e1.cpu().numpy()[:64], g7["h64_l3_m2.e1_head"], rtol=1e-5, atol=5e-6)
    ref = g7["h64_l3_m2.out"]
    assert_forward_close(out, ref)


@pytest.mark.parametrize("hid,nl,ms", [(64, 2, 3), (64, 5, 2), (128, 3, 3), (128, 2, 3), (256, 2, 3), (256, 4, 2)])
def test_epd_any_size_vs_oracle(dev, hid, nl, ms):
    """Streamed fp16-split kernels (hmlp.hip) at every hidden size they are instantiated for and several depths: forward on a
    multi-tile graph and on ragged small ones, the standalone blocks, and a rollout step -- all against the oracle."""
    from gnn_manip_amd import RolloutEngine, scene
    params = orc.init_params(25, 4, 3, hid, nl, ms, 7 * hid + nl)
    m = _model(params, (25, 4, 3, hid, nl, ms), dev)
    m.set_edge_kernel("hm")
    for n, side, seed in ((2500, 0.1, 81), (130, 0.3, 82), (1, 0.1, 83), (517, 0.06, 84)):
        obs = scene.make_scene(n, seed=seed, side=side)
        nodes, ea, s, r, _ = orc.process(obs, None, control_idx=CTRL, **KW)
        ei = np.stack((s, r))
        with torch.no_grad():
            out = m.forward(_t(nodes, dev), _t(ea, dev), _t(ei, dev)).cpu().numpy()
        ref = orc.epd_forward(params, nodes, ea, ei, nl, ms)
        assert_forward_close(out, ref, floor=1e-3, what=n)
    # blocks on the last (517-node) graph
    with torch.no_grad():
        h0, e0, _ = m.encoder(_t(nodes, dev), _t(ea, dev), _t(ei, dev))
        h1, e1, _ = m.processor[0](h0, e0, _t(ei, dev))
    ho, eo = orc.graph_independent(params, "encoder", nodes, ea, nl)
    np.testing.assert_allclose(h0.cpu().numpy(), ho, rtol=1e-5, atol=3e-6)
    np.testing.assert_allclose(e0.cpu().numpy(), eo, rtol=1e-5, atol=3e-6)
    h1o, e1o = orc.interaction_network(params, "processor.0", ho, eo, ei, nl)
    np.testing.assert_allclose(h1.cpu().numpy(), h1o, rtol=1e-5, atol=5e-6)
    np.testing.assert_allclose(e1.cpu().numpy(), e1o, rtol=1e-5, atol=5e-6)
    # one device-resident rollout step
    n = 600
    obs = scene.make_scene(n, seed=85, side=0.07)
    traj = scene.rigid_drift_trajectory(obs, 1)
    eng = RolloutEngine(m, _ga(), n, device=dev)
    with torch.no_grad():
        final = eng.rollout(_t(obs, dev), _t(traj, dev), horizon=1).cpu().numpy()
    fwd = lambda nn, ee, ii: orc.epd_forward(params, nn, ee, ii, nl, ms)
    refs = orc.rollout(params, obs, traj, 1, STATS, BOUNDS, 0.015, CART, MAT, CTRL, forward_fn=fwd)
    np.testing.assert_allclose(final[:, :, 2:5], refs[:, :, 2:5], rtol=0, atol=5e-6)


def test_host_resident_weights_use_the_same_kernels(dev):
    """A model whose parameters live in host memory is staged to the device whole, so it gets every operand image: same
    kernels, bit-identical output to the model created from device tensors."""
    from gnn_manip_amd import _lib, scene
    n = 700
    obs = scene.make_scene(n, seed=91, side=0.07)
    params = orc.init_params(25, 4, 3, 128, 2, 4, 92)
    m = _model(params, (25, 4, 3, 128, 2, 4), dev)
    nodes, ea, s, r, _ = orc.process(obs, None, control_idx=CTRL, **KW)
    ei = np.stack((s, r))
    with torch.no_grad():
        out_dev = m.forward(_t(nodes, dev), _t(ea, dev), _t(ei, dev)).cpu().numpy()
    import ctypes as C
    from gnn_manip_amd.epd_gnn import DstCsr, _ws
    L = _lib.lib()
    host = [np.ascontiguousarray(p.detach().cpu().numpy()) for p in m.parameters()]   # state_dict order
    arr = (C.c_void_p * len(host))(*[h.ctypes.data for h in host])
    desc = _lib.ModelDesc(*m.model_desc())
    handle = C.c_void_p()
    xn, xe, xi = _t(nodes, dev), _t(ea, dev), _t(ei, dev)
    csr = DstCsr(xi, n)
    _lib.check(L.gm_model_create(C.byref(desc), arr, len(host), 0, _lib.current_stream(), C.byref(handle)))
    try:
        e = int(ei.shape[1])
        out = torch.empty((n, 3), device=dev)
        fwd = _ws(L.gm_forward_workspace_bytes(C.byref(desc), n, e), xn.device)
        _lib.check(L.gm_epd_forward(handle, _lib.ptr(xn), n, _lib.ptr(xe), 0, _lib.ptr(csr.ws), e, _lib.ptr(out), _lib.ptr(fwd), fwd.numel(),
                                    _lib.current_stream()))
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), out_dev)
    finally:
        L.gm_model_destroy(handle)


def test_profile_is_per_model(dev):
    """gm_model_profile: HIP-event timing belongs to one model handle; a second model's launches are not recorded."""
    from gnn_manip_amd import EncProcDecGNN
    a = EncProcDecGNN(25, 4, 3, 128, 2, 2).to(dev)
    b = EncProcDecGNN(25, 4, 3, 128, 2, 3).to(dev)
    x, ea = torch.randn(50, 25, device=dev), torch.randn(200, 4, device=dev)
    ei = torch.randint(0, 50, (2, 200), device=dev)
    a.profile(0b11)
    with torch.no_grad():
        a.forward(x, ea, ei)
        b.forward(x, ea, ei)
    assert a.profile_query(0)[0] == 2 and a.profile_query(1)[0] == 2    # a's two processor steps
    assert a.profile_query(0)[1] > 0.0
    assert b.profile_query(0) == (0, 0.0)
    a.profile(0)
    with torch.no_grad():
        a.forward(x, ea, ei)
    assert a.profile_query(0)[0] == 2


def test_profile_kinds_cover_a_rollout_step_and_report_when_they_run_out(dev):
    """The five profiling kinds of gm_model_profile bracket every launch of a rollout step (0 processor edge kernels, 1 node side,
    2 radius graph, 3 encoders, 4 the rest: state update + features + resets, destination sort + edge features + block tables,
    integration) -- bench.py's breakdown adds them up against the step.  A kind records 4096 scopes; once more were opened its
    count comes back NEGATIVE (minus the number opened) instead of a silently truncated total."""
    from gnn_manip_amd import RolloutEngine, scene
    n = 400
    obs = scene.make_scene(n, seed=71, side=0.06)
    m = _model(orc.init_params(25, 4, 3, 128, 2, 10, 72), (25, 4, 3, 128, 2, 10), dev)
    eng = RolloutEngine(m, _ga(), n, device=dev)
    state = _t(obs, dev)
    eng.set_scene(state)
    traj = _t(scene.rigid_drift_trajectory(obs, 8, seed=73, step_size=1e-5), dev)
    with torch.no_grad():
        eng.run(state, traj[:2].contiguous(), 2)     # packs the weight images
        m.profile(31)
        eng.run(state, traj[2:5].contiguous(), 3)
    counts = [m.profile_query(k) for k in range(5)]
    assert [c[0] for c in counts] == [30, 30, 3, 6, 9], counts     # per step: 10 edge, 10 node (streamed path), 1 graph scope, 2 encoders, 3 rest scopes
    assert all(c[1] > 0.0 for c in counts)
    with torch.no_grad():
        eng.run(state, traj[5:].contiguous().repeat(140, 1, 1), 420)        # 4200 more edge scopes: beyond the 4096 of a kind
    full, ms_full = m.profile_query(0)
    assert full == -(30 + 4200) and ms_full > counts[0][1]
    assert m.profile_query(2)[0] == 3 + 420           # the kinds with room left keep counting
    m.profile(0)


@pytest.mark.parametrize("kernel", ["auto", "hm"])
def test_scatter_add_is_deterministic_with_hub_nodes(dev, kernel):
    """Destinations whose segments span many 4-block groups (in-degree 700 and 300, i.e. > 5 and > 2 groups of 128 edges):
    their head partials go to the side buffer and are added in group order by the node kernel -- no atomics -- so two
    runs agree bit for bit, and the result is the oracle's."""
    rng = np.random.Generator(np.random.PCG64(73))
    n, e = 500, 9000
    ei = rng.integers(0, n, size=(2, e)).astype(np.int64)
    ei[1, :700] = 17
    ei[1, 700:1000] = 401
    nodes = rng.standard_normal((n, 25)).astype(np.float32)
    ea = rng.standard_normal((e, 4)).astype(np.float32)
    params = orc.init_params(25, 4, 3, 128, 2, 3, 74)
    m = _model(params, (25, 4, 3, 128, 2, 3), dev)
    m.set_edge_kernel(kernel)
    with torch.no_grad():
        outs = [m.forward(_t(nodes, dev), _t(ea, dev), _t(ei, dev)).cpu().numpy() for _ in range(4)]
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])
    ref = orc.epd_forward(params, nodes, ea, ei, 2, 3)
    assert_forward_close(outs[0], ref, floor=1e-3)
    # the standalone block (edges in the caller's order, eid indirection) goes through the same lists
    h0, e0 = orc.graph_independent(params, "encoder", nodes, ea, 2)
    with torch.no_grad():
        h1, e1, _ = m.processor[0](_t(h0, dev), _t(e0, dev), _t(ei, dev))
    h1o, e1o = orc.interaction_network(params, "processor.0", h0, e0, ei, 2)
    np.testing.assert_allclose(h1.cpu().numpy(), h1o, rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("flow", [0, 1])
def test_edge_features_csr_is_flow_aware(dev, flow):
    """gm_edge_features_csr on a structure built with either aggregation row returns the reference feature
    (p_sender - p_receiver) / r, sender = edge_index[0] (utils.py:43-61) -- the same rows as gm_edge_features, in sorted order."""
    import ctypes as C
    from gnn_manip_amd._lib import check, current_stream, lib, ptr
    rng = np.random.Generator(np.random.PCG64(17))
    n, e = 300, 2500
    pos = rng.random((n, 3)).astype(np.float32)
    ei = rng.integers(0, n, size=(2, e)).astype(np.int64)
    L = lib()
    tp, tei = _t(pos, dev), _t(ei, dev)
    ws = torch.empty(L.gm_csr_workspace_bytes(n, e), dtype=torch.uint8, device=dev)
    check(L.gm_csr_from_edge_index_flow(ptr(tei), n, e, flow, ptr(ws), ws.numel(), current_stream(dev)))
    out_csr = torch.empty((e, 4), dtype=torch.float32, device=dev)
    check(L.gm_edge_features_csr(ptr(tp), 3, ptr(ws), n, e, 0.015, ptr(out_csr), current_stream(dev)))
    out_ref = torch.empty((e, 4), dtype=torch.float32, device=dev)
    check(L.gm_edge_features(ptr(tp), 3, ptr(tei[0].contiguous()), ptr(tei[1].contiguous()), e, 0.015, ptr(out_ref), current_stream(dev)))
    a, b = out_csr.cpu().numpy(), out_ref.cpu().numpy()
    np.testing.assert_array_equal(a[np.lexsort(a.T[::-1])], b[np.lexsort(b.T[::-1])])   # same rows, another order
    np.testing.assert_allclose(b, orc.get_edges_displacement(pos, ei[0], ei[1], 0.015), rtol=2e-7, atol=0)


@pytest.mark.parametrize("hidden,nl,ms", [(96, 2, 3), (32, 2, 2), (192, 3, 2), (224, 2, 1), (100, 2, 3), (7, 2, 2), (150, 3, 2), (255, 2, 1)])
def test_hidden_sizes_between_the_instantiated_widths(dev, hidden, nl, ms):
    """epd_gnn.py:13-14,72-84 take any hidden_size: any size up to 256 runs zero-padded at the next instantiated width with the
    LayerNorm statistics over the features that exist -- forward, standalone blocks and a short rollout against the oracle."""
    from gnn_manip_amd import RolloutEngine, scene
    obs = scene.make_scene(700, seed=140 + hidden, side=0.075)
    params = orc.init_params(25, 4, 3, hidden, nl, ms, 900 + hidden)
    m = _model(params, (25, 4, 3, hidden, nl, ms), dev)
    nodes, ea, s, r, _ = orc.process(obs, None, control_idx=CTRL, **KW)
    ei = np.stack((s, r))
    with torch.no_grad():
        out = m.forward(_t(nodes, dev), _t(ea, dev), _t(ei, dev)).cpu().numpy()
        h0, e0, _ = m.encoder(_t(nodes, dev), _t(ea, dev), _t(ei, dev))
        h1, e1, _ = m.processor[0](h0, e0, _t(ei, dev))
    assert m.status() == ei.shape[1]
    ref = orc.epd_forward(params, nodes, ea, ei, nl, ms)
    assert_forward_close(out, ref)
    ho, eo = orc.graph_independent(params, "encoder", nodes, ea, nl)
    assert tuple(h0.shape) == (nodes.shape[0], hidden) and tuple(e0.shape) == (ea.shape[0], hidden)
    np.testing.assert_allclose(h0.cpu().numpy(), ho, rtol=1e-5, atol=5e-6)
    np.testing.assert_allclose(e0.cpu().numpy(), eo, rtol=1e-5, atol=5e-6)
    h1o, e1o = orc.interaction_network(params, "processor.0", ho, eo, ei, nl)
    np.testing.assert_allclose(h1.cpu().numpy(), h1o, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(e1.cpu().numpy(), e1o, rtol=1e-5, atol=1e-5)
    if hidden in (96, 100):
        traj = scene.rigid_drift_trajectory(obs, 2)
        eng = RolloutEngine(m, _ga(), obs.shape[1], device=dev)
        with torch.no_grad():
            final = eng.rollout(_t(obs, dev), _t(traj, dev), horizon=2).cpu().numpy()
        refr = orc.rollout(params, obs, traj, 2, STATS, BOUNDS, 0.015, CART, MAT, CTRL, nl, ms)
        np.testing.assert_allclose(final[:, :, 2:5], refr[:, :, 2:5], rtol=0, atol=5e-6)
