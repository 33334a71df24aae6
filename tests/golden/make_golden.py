#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the REFERENCE's own Python.

Runs only in the build container (needs /root/reference, read-only).  The reference
cannot be imported the normal way (its package __init__s pull in cma / geomloss /
torch_geometric / torch_graphnet / a missing action_networks.py), so its files are
loaded BY PATH into synthetic packages, with inert stand-ins for the absent third-party
modules.  Nothing from the reference is copied: the fixtures hold inputs and the
outputs the reference code produced for them.

    python -B tests/golden/make_golden.py

The two torch_graphnet blocks are absent from the reference tree.
The stand-ins used for fixtures G7/G8 are OUR plain-torch blocks implementing the
north_star semantics; those two fixtures therefore pin the reference's *wiring*
(epd_gnn.py MLP structure, LayerNorm/residual placement, state_dict naming, the
cma_objective state-update loop), not the block arithmetic.
"""
import importlib.util
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import epd_oracle as orc  # noqa: E402  (weights generator + tie check only)


# ------------------------------------------------------------------ stand-in blocks
class StubGraphIndependent(nn.Module):
    def __init__(self, phi_edge, phi_node):
        super().__init__()
        self.phi_edge = phi_edge
        self.phi_node = phi_node

    def forward(self, x, edge_attr, edge_index):
        return self.phi_node(x), self.phi_edge(edge_attr), None


class StubInteractionNetwork(nn.Module):
    def __init__(self, phi_edge, phi_node):
        super().__init__()
        self.phi_edge = phi_edge
        self.phi_node = phi_node

    def forward(self, x, edge_attr, edge_index):
        j, i = edge_index[0], edge_index[1]
        e = self.phi_edge(torch.cat((x[i], x[j], edge_attr), dim=-1))
        agg = torch.zeros_like(x).index_add_(0, i, e)
        h = self.phi_node(torch.cat((x, agg), dim=-1))
        return h, e, None


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def load_reference():
    class _Opts(dict):
        pass
    cma = _mod("cma")
    cma.evolution_strategy = _mod("cma.evolution_strategy", CMAOptions=_Opts)
    _mod("geomloss", SamplesLoss=lambda **kw: None)
    tg = _mod("torch_geometric")
    tg.data = _mod("torch_geometric.data", Data=object, Dataset=object)
    _mod("torch_graphnet", InteractionNetwork=StubInteractionNetwork, GraphIndependent=StubGraphIndependent)
    for pkg in ("gnn_manip", "gnn_manip.utils", "gnn_manip.models"):
        p = _mod(pkg)
        p.__path__ = []
    _mod("gnn_manip.utils.plot_utils", plot_multiple_nodes=lambda *a, **k: None)
    _mod("gnn_manip.models.action_networks", ActionInteractionNetwork=None, ActionGraphIndependent=None)
    ref = types.SimpleNamespace()
    ref.utils = _load("gnn_manip.utils.utils", "gnn_manip/utils/utils.py")
    ref.collate = _load("gnn_manip.utils.collate_utils", "gnn_manip/utils/collate_utils.py")
    ref.epd = _load("gnn_manip.models.epd_gnn", "gnn_manip/models/epd_gnn.py")
    sys.modules["gnn_manip.models"].EncProcDecGNN = ref.epd.EncProcDecGNN
    ref.dataset = _load("gnn_manip.utils.coffee_dataset", "gnn_manip/utils/coffee_dataset.py")
    ref.rollout = _load("gnn_manip.utils.rollout_utils", "gnn_manip/utils/rollout_utils.py")
    ref.traj = _load("gnn_manip.utils.traj_utils", "gnn_manip/utils/traj_utils.py")
    return ref


# ------------------------------------------------------------------ synthetic inputs
def scene_positions(n, side, seed, lo=0.3):
    rng = np.random.Generator(np.random.PCG64(seed))
    return (lo + side * rng.random((n, 3))).astype(np.float32)


STATS = dict(velocity_mean=[1.5e-4, -2.5e-4, 0.5e-4], velocity_std=[2.1e-3, 3.2e-3, 1.9e-3],
             acceleration_mean=[1.0e-6, -8.0e-6, 2.0e-6], acceleration_std=[2.4e-4, 3.1e-4, 2.2e-4])
BOUNDS = dict(lower_bounds=[0.1, 0.1, 0.1], upper_bounds=[0.9, 0.9, 0.9])
CART, MAT, CTRL = [2, 3, 4], [1], [5, 6, 7]


def scene_obs(n, n_rigid, side, seed, k=6, lo=0.3):
    """[k, N, 8] float32: cols [id, material, x, y, z, cx, cy, cz]; last n_rigid rows rigid."""
    rng = np.random.Generator(np.random.PCG64(seed))
    p0 = lo + side * rng.random((n, 3))
    v = 1e-3 * rng.standard_normal((n, 3))
    obs = np.zeros((k, n, 8), dtype=np.float32)
    for t in range(k):
        obs[t, :, 2:5] = (p0 + t * v + 1e-5 * rng.standard_normal((n, 3))).astype(np.float32)
    obs[:, :, 0] = np.arange(n)
    obs[:, n - n_rigid:, 1] = 1.0
    obs[-1, n - n_rigid:, 5:8] = (1e-3 * rng.standard_normal((n_rigid, 3))).astype(np.float32)
    return obs


def tstats():
    return {k: torch.tensor(v, dtype=torch.float32) for k, v in STATS.items()}


def tbounds():
    return {k: torch.tensor(v, dtype=torch.float32) for k, v in BOUNDS.items()}


def main():
    ref = load_reference()
    torch.set_num_threads(4)
    out = {}

    # ---------------- G1 connectivity (utils.py:64-93) + G2 edge features (utils.py:43-61)
    g1 = {}
    cases = [("dense200", 200, 0.06, 11, 0.015, 20), ("mixed500", 500, 0.12, 12, 0.015, 20),
             ("sparse64", 64, 0.20, 13, 0.015, 20), ("mean20_3000", 3000, 0.128, 14, 0.015, 20),
             ("cap5", 300, 0.08, 15, 0.015, 5), ("cap40_r03", 400, 0.14, 16, 0.03, 40)]
    for name, n, side, seed, r, cap in cases:
        pos = scene_positions(n, side, seed)
        assert orc.connectivity_is_tie_free(pos, r, cap), name
        s, rcv = ref.utils.get_connectivity(torch.from_numpy(pos), r, cap)
        ea = ref.utils.get_edges_displacement(torch.from_numpy(pos), s, rcv, r)
        g1[f"{name}.pos"] = pos
        g1[f"{name}.r_cap"] = np.array([r, cap], dtype=np.float64)
        g1[f"{name}.senders"] = s.numpy().astype(np.int32)
        g1[f"{name}.receivers"] = rcv.numpy().astype(np.int32)
        g1[f"{name}.edge_attr"] = ea.numpy()
        print("G1", name, "N", n, "E", s.numel())
    np.savez_compressed(os.path.join(HERE, "g1_connectivity.npz"), **g1)

    # ---------------- G3/G4 node features, process, process_collate (collate_utils.py)
    g4 = {}
    obs_a = scene_obs(180, 40, 0.07, 21)
    obs_b = scene_obs(150, 30, 0.09, 22)
    tgt_a = obs_a[-1, :, 2:5] + np.float32(1e-3)
    tgt_b = obs_b[-1, :, 2:5] - np.float32(5e-4)
    ga = ref.collate.GraphBoundedMultimaterialControl(conn_r=0.015, stats=tstats(), cartesian_idx=CART,
                                                     material_idx=MAT, control_idx=CTRL, bounds=tbounds(),
                                                     noise=None)
    gn = ref.collate.GraphBoundedMultimaterial(conn_r=0.015, stats=tstats(), cartesian_idx=CART,
                                               material_idx=MAT, bounds=tbounds(), noise=None)
    g4["obs_a"], g4["obs_b"], g4["tgt_a"], g4["tgt_b"] = obs_a, obs_b, tgt_a, tgt_b
    g4["vel_a"] = ref.utils.get_nodes_vel(torch.from_numpy(obs_a[:, :, 2:5]), tstats()["velocity_mean"],
                                          tstats()["velocity_std"]).numpy()
    g4["nodes_ctrl_a"] = ga.compute_nodes(torch.from_numpy(obs_a)).numpy()
    g4["nodes_noctrl_a"] = gn.compute_nodes(torch.from_numpy(obs_a[:, :, :5])).numpy()
    nodes, ea, s, r, tgt = ga.process(torch.from_numpy(obs_a), torch.from_numpy(tgt_a))
    g4["proc_nodes"], g4["proc_edge_attr"], g4["proc_tgt"] = nodes.numpy(), ea.numpy(), tgt.numpy()
    g4["proc_senders"], g4["proc_receivers"] = s.numpy().astype(np.int32), r.numpy().astype(np.int32)
    batch = [(torch.from_numpy(obs_a), torch.from_numpy(tgt_a)), (torch.from_numpy(obs_b), torch.from_numpy(tgt_b))]
    nodes, ea, ei, tgt = ga.process_collate(batch)
    g4["coll_nodes"], g4["coll_edge_attr"], g4["coll_tgt"] = nodes.numpy(), ea.numpy(), tgt.numpy()
    g4["coll_edge_index"] = ei.numpy().astype(np.int32)
    for nm, o in (("obs_a", obs_a), ("obs_b", obs_b)):
        assert orc.connectivity_is_tie_free(o[-1, :, 2:5], 0.015, 20), nm
    # ---------------- G5 target acceleration + integrator (utils.py:10-24, rollout_utils.py:145-158)
    g4["acc_a"] = ref.utils.compute_acceleration(torch.from_numpy(tgt_a), torch.from_numpy(obs_a[:, :, 2:5])).numpy()
    rng = np.random.Generator(np.random.PCG64(23))
    pred = rng.standard_normal((180, 3)).astype(np.float32)
    g4["pred_acc"] = pred
    g4["next_pos"] = ref.rollout.get_position_from_prediction(tstats(), CART, torch.from_numpy(pred),
                                                              torch.from_numpy(obs_a)).numpy()
    np.savez_compressed(os.path.join(HERE, "g4_features.npz"), **g4)
    print("G3-G5 done; collate E", ei.shape[1])

    # ---------------- G6 trajectory functions fed with dataset/sample_traj.npy
    g6 = {}
    sample = np.load(os.path.join(REF, "dataset", "sample_traj.npy"))
    obs_c = scene_obs(60, 24, 0.05, 31)
    obs_c[:, 36:, 2:5] = (np.array([0.5, 0.4, 0.5], dtype=np.float32)
                          + 0.04 * (np.random.Generator(np.random.PCG64(32)).random((24, 3)).astype(np.float32) - 0.5))
    gc = ref.collate.GraphBoundedMultimaterialControl(conn_r=0.015, stats=tstats(), cartesian_idx=CART,
                                                     material_idx=MAT, control_idx=CTRL, bounds=tbounds(),
                                                     noise=None)
    state = (torch.from_numpy(obs_c.copy()), torch.from_numpy(obs_c[-1, :, 2:5].copy()))
    kw = dict(alpha=0.0, beta=1000.0, gamma=0.05, penalty=0.0, rho=0.0, device="cpu")
    solver = ref.traj.TrajectoryCMAsolver(None, gc, state, 180, [0.5, 0.5, 0.4], scale_rot=1.0, scale_ty=1.0,
                                          total_steps=300, **kw)
    solver.set_sample_traj(sample)
    n_inc = solver.sample_traj.shape[0]
    x0 = np.concatenate((solver.sample_traj[:, 0], solver.sample_traj[:, 1]))
    rot, ty = solver.interpolate_trajectory(x0)
    rb, actions = solver.get_rigid_body_trajectory_from_diff(x0)
    g6["obs_c"] = obs_c
    g6["scale_ty_eff"] = np.array([solver.scale_ty, solver.scale_rot, solver.rx_init, solver.max_rot, solver.max_ty])
    g6["sample_scaled"] = solver.sample_traj
    g6["x0"] = x0
    g6["traj_rot"], g6["traj_ty"] = np.array(rot), np.array(ty)
    g6["rigid_traj_steps"] = np.array([0, 1, 2, 50, 150, 299])
    g6["rigid_traj"] = rb.numpy()[g6["rigid_traj_steps"]]
    g6["actions"] = actions
    # perturbed candidate that exercises the clipping branches
    xr = x0 + 0.3 * np.random.Generator(np.random.PCG64(33)).standard_normal(x0.shape) * np.abs(x0).max()
    rot2, ty2 = solver.interpolate_trajectory(xr)
    g6["x_pert"], g6["traj_rot_pert"], g6["traj_ty_pert"] = xr, np.array(rot2), np.array(ty2)
    # rollout_utils duplicate (rollout_utils.py:161-205)
    rp = torch.from_numpy(obs_c[-1, 36:, 2:5].copy())
    demo = np.stack((np.array(rot), np.array(ty))).T
    rb2 = ref.rollout.get_rigid_body_trajectory_from_diff(demo, 4, [0.5, 0.5, 0.4], rp)
    g6["rigid_traj_rollout_utils"] = rb2.numpy()
    np.savez_compressed(os.path.join(HERE, "g6_trajectory.npz"), **g6)
    print("G6 done; increments", n_inc)

    # ---------------- G7 EncProcDecGNN wiring (epd_gnn.py) with seeded oracle weights
    g7 = {}
    for tag, hid, nl, ms, seed in (("h128", 128, 2, 10, 41), ("h64_l3_m2", 64, 3, 2, 42)):
        params = orc.init_params(25, 4, 3, hid, nl, ms, seed)
        model = ref.epd.EncProcDecGNN(node_dim=25, edge_dim=4, out_dim=3, hidden_size=hid, num_layers=nl,
                                      m_steps=ms)
        sd = {k: torch.from_numpy(v) for k, v in params.items()}
        model.load_state_dict(sd, strict=True)  # also pins state_dict key names / shapes
        nodes, ea, s, r, _ = ga.process(torch.from_numpy(obs_a), torch.from_numpy(tgt_a))
        ei = torch.stack((s, r))
        with torch.no_grad():
            y = model.forward(nodes, ea, ei)
            # intermediate pins: encoder output and first processor step (pre-residual block output)
            h0, e0, _ = model.encoder(nodes, ea, ei)
            h1, e1, _ = model.processor[0](h0, e0, ei)
        g7[f"{tag}.cfg"] = np.array([25, 4, 3, hid, nl, ms, seed])
        g7[f"{tag}.out"] = y.numpy()
        g7[f"{tag}.h0"], g7[f"{tag}.e0_head"] = h0.numpy(), e0.numpy()[:64]
        g7[f"{tag}.h1"], g7[f"{tag}.e1_head"] = h1.numpy(), e1.numpy()[:64]
        g7[f"{tag}.keys"] = np.array(sorted(model.state_dict().keys()))
        print("G7", tag, "params", sum(v.size for v in params.values()), "out absmax", float(y.abs().max()))
    np.savez_compressed(os.path.join(HERE, "g7_epd_wiring.npz"), **g7)

    # ---------------- G8 cma_objective rollout loop (traj_utils.py:114-159)
    g8 = {}
    captured = {}

    class Capture(ref.traj.TrajectoryCMAsolver):
        def compute_loss(self, end_position, actions, cup_states=None, coffee_states=None, x=None):
            captured["end"] = end_position.clone()
            captured["cup"] = torch.stack(cup_states)
            captured["coffee"] = torch.stack(coffee_states)
            return 0.0, 0, 0, 0, 0, 0

    hid, nl, ms, seed = 128, 2, 10, 51
    params = orc.init_params(25, 4, 3, hid, nl, ms, seed)
    model = ref.epd.EncProcDecGNN(25, 4, 3, hid, nl, ms)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    horizon = 7  # interpolate_trajectory yields n_increments+1 = 7 poses (sample[:8] -> 6 increments)
    state = (torch.from_numpy(obs_c.copy()), torch.from_numpy(obs_c[-1, :, 2:5].copy()))
    cap = Capture(model, gc, state, 180, [0.5, 0.5, 0.4], scale_rot=1.0, scale_ty=1.0, total_steps=horizon, **kw)
    cap.set_sample_traj(sample[:8])
    x = np.concatenate((cap.sample_traj[:, 0], cap.sample_traj[:, 1]))
    cap.cma_objective(x)
    rot, ty = cap.interpolate_trajectory(x)
    g8["cfg"] = np.array([25, 4, 3, hid, nl, ms, seed, horizon])
    g8["obs0"] = obs_c
    g8["x"] = x
    g8["traj_rot"], g8["traj_ty"] = np.array(rot), np.array(ty)
    g8["end_coffee"] = captured["end"].numpy()
    g8["cup_states"] = captured["cup"].numpy()
    g8["coffee_states"] = captured["coffee"].numpy()
    np.savez_compressed(os.path.join(HERE, "g8_rollout.npz"), **g8)
    print("G8 done; end absmax", float(np.abs(g8["end_coffee"]).max()))

    # ---------------- G10 planner loss terms (traj_utils.py:161-165,230-285) with the Wasserstein term stubbed
    # (geomloss is absent): pins compute_vel_acc, compute_vel_loss, compute_acc_loss, compute_boundaries_penalty and
    # the weighting in compute_loss.
    g10 = {}
    kw10 = dict(alpha=0.3, beta=1000.0, gamma=0.05, penalty=2.0, rho=0.0, device="cpu")
    solver = ref.traj.TrajectoryCMAsolver(model, gc, state, 180, [0.5, 0.5, 0.4], scale_rot=1.0, scale_ty=1.0,
                                          total_steps=horizon, **kw10)
    solver.loss = lambda a, b: torch.tensor(0.125)
    solver.desired_pos = state[1]
    rng = np.random.Generator(np.random.PCG64(77))
    for tag, spread in (("inside", 0.3), ("outside", 3.5)):
        actions = np.zeros((40, 2))
        actions[:, 0] = np.pi + spread * np.sin(np.linspace(0, 3, 40)) + 1e-3 * rng.standard_normal(40)
        actions[:, 1] = 1e-3 * np.cumsum(rng.standard_normal(40))
        out = solver.compute_loss(state[1], actions)
        g10[f"{tag}.actions"] = actions
        g10[f"{tag}.out"] = np.array([float(v) for v in out])
    g10["cfg"] = np.array([solver.alpha, solver.beta, solver.gamma, solver.penalty, solver.rho, solver.max_rot, solver.max_ty,
                           solver.rx_init, solver.rotation_limit])
    # InterpolatedCMAsolver (traj_utils.py:288-452): key-point parametrisation, pchip interpolation, its loss terms and
    # inequality constraints (Wasserstein term stubbed as above)
    hz, npts = 40, 10
    kw11 = dict(alpha=0.3, beta=1000.0, gamma=0.05, penalty=2.0, rho=0.7, device="cpu")
    isol = ref.traj.InterpolatedCMAsolver(model, gc, state, 180, [0.5, 0.5, 0.4], scale_rot=np.pi, scale_ty=1.0,
                                          total_steps=hz, traj_points=npts, **kw11)
    isol.loss = lambda a, b: torch.tensor(0.125)
    isol.desired_pos = state[1]
    isol.set_sample_traj(sample[:hz + 1])
    xi = np.concatenate((isol.sample_traj[:, 0], isol.sample_traj[:, 1]))
    xi = xi + 0.02 * rng.standard_normal(xi.shape)
    rot_i, ty_i = isol.interpolate_trajectory(xi)
    act_i = np.stack((rot_i, ty_i), axis=1)
    g10["interp.sample_in"] = sample[:hz + 1]
    g10["interp.sample_traj"] = isol.sample_traj
    g10["interp.x"] = xi
    g10["interp.rot"], g10["interp.ty"] = np.asarray(rot_i), np.asarray(ty_i)
    g10["interp.out"] = np.array([float(v) for v in isol.compute_loss(state[1], act_i, x=xi)])
    g10["interp.ineq"] = isol.ineq_constraint(xi)
    g10["interp.cfg"] = np.array([isol.alpha, isol.beta, isol.gamma, isol.penalty, isol.rho, isol.max_rot, isol.max_ty, isol.rx_init,
                                  isol.rotation_limit, isol.scale_rot, isol.scale_ty, hz, npts])
    np.savez_compressed(os.path.join(HERE, "g10_planner_loss.npz"), **g10)
    print("G10 done", g10["inside.out"], g10["outside.out"])

    # ---------------- G9: on-disk dataset (CSV + metadata.json) and the training-time noise path
    # coffee_dataset.py:18-113 (read_metadata, CoffeeDataset._load_data / graph_attr), utils.py:96-115 (random_walk_noise),
    # collate_utils.py:169-193 (_process_noisy).  The dataset is synthetic and tiny; it is stored in the fixture as
    # arrays and re-written to disk by the tests.
    import json, tempfile
    g9 = {}
    T, N9, NR = 9, 60, 10
    sims = []
    for sid in (1, 2):
        o = scene_obs(N9, NR, 0.05, 90 + sid, k=T)
        sims.append(o[:, :, :5].astype(np.float32))  # on disk: [id, material, x, y, z]
    vel = np.concatenate([np.diff(d[:, :, 2:5].astype(np.float64), axis=0).reshape(-1, 3) for d in sims])
    acc = np.concatenate([np.diff(np.diff(d[:, :, 2:5].astype(np.float64), axis=0), axis=0).reshape(-1, 3) for d in sims])
    meta = {"cartesian_idx": [2, 3, 4], "control_idx": [5, 6, 7], "material_id": 1, "bounds": [[0.1, 0.9]] * 3,
            "sequence_length": T, "dim": 3, "data_dim": 5,
            "vel_mean": list(vel.mean(0)), "vel_std": list(vel.std(0)), "acc_mean": list(acc.mean(0)), "acc_std": list(acc.std(0))}
    with tempfile.TemporaryDirectory() as td:
        root = td + "/"
        os.makedirs(root + "train")
        with open(root + "metadata.json", "w") as fp:
            json.dump(meta, fp)
        with open(root + "train/sim_data.csv", "w") as fp:
            for sid in (1, 2):
                fp.write(f"{sid},0\n")
        for sid, d in zip((1, 2), sims):
            np.savetxt(root + f"train/particles_{sid:06d}.csv", d.reshape(-1, 5).astype(np.float64), delimiter=",", fmt="%.9g")
        for tag, use_control in (("ctl", True), ("noctl", False)):
            ds = ref.dataset.CoffeeDataset.__new__(ref.dataset.CoffeeDataset)
            ref.dataset.CoffeeDataset.__init__(ds, root, 6, 0.015, split="train", use_control=use_control)
            g9[f"{tag}.len"] = np.array(len(ds))
            for idx in (0, 4):
                obs = ds.raw_samples["observations"][idx]
                nxt = ds.raw_samples["next_positions"][idx]
                nodes, ea, s_, r_, tgt = ds.graph_attr.process(obs, nxt)
                assert orc.connectivity_is_tie_free(obs[-1][:, 2:5].numpy(), 0.015, 20)
                g9[f"{tag}.{idx}.obs"], g9[f"{tag}.{idx}.next"] = obs.numpy(), nxt.numpy()
                g9[f"{tag}.{idx}.nodes"], g9[f"{tag}.{idx}.edge_attr"] = nodes.numpy(), ea.numpy()
                g9[f"{tag}.{idx}.senders"], g9[f"{tag}.{idx}.receivers"], g9[f"{tag}.{idx}.tgt"] = s_.numpy(), r_.numpy(), tgt.numpy()
    g9["sims"] = np.stack(sims)
    g9["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    # noise: the reference draws Normal(0, std / sqrt(k-1)) first thing after the seed; the same draw is stored as `sample`
    std = 3e-4
    obs = torch.from_numpy(scene_obs(120, 20, 0.06, 95))
    pos_seq = obs[:, :, CART]
    torch.manual_seed(11)
    vel_seq = torch.diff(pos_seq, n=1, dim=0).float()
    sample = torch.distributions.Normal(loc=torch.zeros_like(vel_seq), scale=std / vel_seq.shape[0] ** 0.5).sample().float()
    torch.manual_seed(11)
    ns = ref.utils.random_walk_noise(pos_seq, std)
    g9["noise.obs"], g9["noise.sample"], g9["noise.sequence"], g9["noise.std"] = obs.numpy(), sample.numpy(), ns.numpy(), np.array(std)
    tgt = obs[-1][:, 2:5] + 1e-3
    gan = ref.collate.GraphBoundedMultimaterialControl(0.015, tstats(), CART, MAT, CTRL, tbounds(), noise=std)
    torch.manual_seed(11)
    nodes, ea, s_, r_, nacc = gan.process(obs, tgt)
    g9["noise.tgt"] = tgt.numpy()
    g9["noise.nodes"], g9["noise.edge_attr"], g9["noise.senders"], g9["noise.receivers"], g9["noise.acc"] = (
        nodes.numpy(), ea.numpy(), s_.numpy(), r_.numpy(), nacc.numpy())
    np.savez_compressed(os.path.join(HERE, "g9_dataset.npz"), **g9)
    print("G9 done; samples", int(g9["ctl.len"]), "noise absmax", float(np.abs(g9["noise.sequence"]).max()))


if __name__ == "__main__":
    main()
