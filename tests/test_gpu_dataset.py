"""Training-time data path on the device: dataset files -> resident tensors -> graphs,
against fixture G9 (reference CoffeeDataset / random_walk_noise / _process_noisy outputs)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import BOUNDS, CART, CTRL, MAT, STATS
from oracle import epd_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _write_dataset(g, root):
    """Re-create the synthetic dataset the fixture was generated from (same bytes as make_golden.py wrote)."""
    os.makedirs(root + "train")
    with open(root + "metadata.json", "w") as fp:
        fp.write(bytes(g["meta_json"]).decode())
    with open(root + "train/sim_data.csv", "w") as fp:
        for sid in (1, 2):
            fp.write(f"{sid},0\n")
    for sid, d in zip((1, 2), g["sims"]):
        np.savetxt(root + f"train/particles_{sid:06d}.csv", d.reshape(-1, 5).astype(np.float64), delimiter=",", fmt="%.9g")


@pytest.mark.parametrize("tag,use_control", [("ctl", True), ("noctl", False)])
def test_dataset_golden(golden, dev, tmp_path, tag, use_control):
    from gnn_manip_amd import CoffeeDataset
    g = golden("g9_dataset.npz")
    root = str(tmp_path) + "/"
    _write_dataset(g, root)
    ds = CoffeeDataset(root, 6, 0.015, split="train", device=dev, use_control=use_control)
    assert len(ds) == int(g[f"{tag}.len"])
    for idx in (0, 4):
        obs, nxt = ds.sample(idx)
        np.testing.assert_array_equal(obs.cpu().numpy(), g[f"{tag}.{idx}.obs"])
        np.testing.assert_array_equal(nxt.cpu().numpy(), g[f"{tag}.{idx}.next"])
        d = ds[idx]
        np.testing.assert_array_equal(d.edge_index[0].cpu().numpy(), g[f"{tag}.{idx}.senders"])  # bit-exact edge list
        np.testing.assert_array_equal(d.edge_index[1].cpu().numpy(), g[f"{tag}.{idx}.receivers"])
        np.testing.assert_allclose(d.x.cpu().numpy(), g[f"{tag}.{idx}.nodes"], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(d.edge_attr.cpu().numpy(), g[f"{tag}.{idx}.edge_attr"], rtol=2e-6, atol=1e-6)
        np.testing.assert_allclose(d.y.cpu().numpy(), g[f"{tag}.{idx}.tgt"], rtol=1e-5, atol=2e-4)


def test_loader_batches_like_the_reference_collate(golden, dev, tmp_path):
    """batch_size=2 (train_dyn.py default): concatenated graphs, second graph's indices offset by N."""
    from gnn_manip_amd import CoffeeDataset, GraphLoader
    g = golden("g9_dataset.npz")
    root = str(tmp_path) + "/"
    _write_dataset(g, root)
    ds = CoffeeDataset(root, 6, 0.015, split="train", device=dev, use_control=True)
    batches = list(GraphLoader(ds, batch_size=2, shuffle=False))
    assert len(batches) == 3
    meta = json.loads(bytes(g["meta_json"]).decode())
    data_dim, T, cart, ctrl, mat, bounds, stats = orc.read_metadata(meta)
    obs_list, next_list = orc.dataset_samples([s.reshape(-1, data_dim) for s in g["sims"]], T, data_dim, 6, cart, mat, True)
    nodes, ea, ei, tgt = orc.process_collate([(obs_list[0], next_list[0]), (obs_list[1], next_list[1])], stats=stats, bounds=bounds,
                                             conn_r=0.015, cartesian_idx=cart, material_idx=[mat], control_idx=ctrl)
    b = batches[0]
    np.testing.assert_array_equal(b.edge_index.cpu().numpy(), ei)
    np.testing.assert_allclose(b.x.cpu().numpy(), nodes, rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(b.edge_attr.cpu().numpy(), ea, rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(b.y.cpu().numpy(), tgt, rtol=1e-5, atol=2e-4)


def test_noise_path_golden(golden, dev):
    """random_walk_noise / _process_noisy with the reference's own Normal draw injected."""
    from gnn_manip_amd import GraphBoundedMultimaterialControl, random_walk_noise
    g = golden("g9_dataset.npz")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    obs, sample, std = t(g["noise.obs"]), t(g["noise.sample"]), float(g["noise.std"])
    seq = random_walk_noise(obs[:, :, 2:5], std, sample)
    np.testing.assert_allclose(seq.cpu().numpy(), g["noise.sequence"], rtol=1e-6, atol=1e-9)
    ga = GraphBoundedMultimaterialControl(0.015, STATS, CART, MAT, CTRL, BOUNDS, noise=std)
    nodes, ea, s, r, acc = ga.process(obs, t(g["noise.tgt"]), noise_sample=sample)
    np.testing.assert_array_equal(s.cpu().numpy(), g["noise.senders"])
    np.testing.assert_array_equal(r.cpu().numpy(), g["noise.receivers"])
    # one ulp of a noisy position (3e-8) is 1.5e-5 in a velocity feature (std 2e-3) and 2e-4 in the target (std 2e-4)
    np.testing.assert_allclose(nodes.cpu().numpy(), g["noise.nodes"], rtol=2e-6, atol=3e-5)
    np.testing.assert_allclose(ea.cpu().numpy(), g["noise.edge_attr"], rtol=2e-6, atol=4e-6)
    np.testing.assert_allclose(acc.cpu().numpy(), g["noise.acc"], rtol=1e-5, atol=6e-4)


def test_noise_draw_statistics(dev):
    """The device draw has the reference's distribution: velocity noise of the last step has std noise_std
    (utils.py:97-101), position noise of the first frame is zero."""
    from gnn_manip_amd import random_walk_noise
    gen = torch.Generator(device=dev).manual_seed(3)
    pos = torch.zeros((6, 20000, 3), device=dev)
    seq = random_walk_noise(pos, 3e-4, generator=gen)
    assert not seq[0].any()
    last_vel_noise = (seq[-1] - seq[-2]).cpu().numpy()
    assert abs(last_vel_noise.std() / 3e-4 - 1.0) < 0.02
    assert abs(last_vel_noise.mean()) < 1e-5


def test_test_dataset_items_are_raw_windows(golden, dev, tmp_path):
    """CoffeeTestDataset (coffee_dataset.py:136-215): one simulation, items = (obs_seq, next_pos) as the rollout and
    the planner consume them; its graph_attr builds the same graph as the training dataset's."""
    from gnn_manip_amd import CoffeeTestDataset
    g = golden("g9_dataset.npz")
    root = str(tmp_path) + "/"
    _write_dataset(g, root)
    ds = CoffeeTestDataset(root, 6, 0.015, split="train", device=dev, use_control=True, sim_id=1)
    assert len(ds) == 3
    obs, nxt = ds[0]
    np.testing.assert_array_equal(obs.cpu().numpy(), g["ctl.0.obs"])
    np.testing.assert_array_equal(nxt.cpu().numpy(), g["ctl.0.next"])
    nodes, ea, s, r, acc = ds.graph_attr.process(obs, nxt)
    np.testing.assert_array_equal(s.cpu().numpy(), g["ctl.0.senders"])
    assert nodes.shape[1] == 25 and ea.shape[1] == 4 and acc.shape[1] == 3  # what get_model reads (rollout_utils.py:123-130)


def test_compute_rollout_both_modes(golden, dev, tmp_path):
    """compute_rollout (rollout_utils.py:12-67) over the mirrored CoffeeTestDataset: planned-trajectory mode against the
    oracle's rollout loop (pinned by G8), ground-truth mode against a restatement of the reference loop built from oracle
    pieces."""
    import types
    from gnn_manip_amd import CoffeeTestDataset, EncProcDecGNN
    from gnn_manip_amd.rollout import compute_rollout
    g = golden("g9_dataset.npz")
    root = str(tmp_path) + "/"
    _write_dataset(g, root)
    ds = CoffeeTestDataset(root, 6, 0.015, split="train", device=dev, use_control=True, sim_id=1)
    meta = json.loads(bytes(g["meta_json"]).decode())
    data_dim, T, cart, ctrl, mat, bounds, stats = orc.read_metadata(meta)
    params = orc.init_params(25, 4, 3, 128, 2, 2, 44)
    params["decoder.4.weight"] = params["decoder.4.weight"] * 1e-2
    m = EncProcDecGNN(25, 4, 3, 128, 2, 2)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    m = m.to(dev)
    obs_list, next_list = orc.dataset_samples([g["sims"][0].reshape(-1, data_dim)], T, data_dim, 6, cart, mat, True)
    # --- planned trajectory
    traj_np = np.stack((np.pi - 0.01 * np.arange(T), 1e-4 * np.arange(T)), axis=1)
    pred = compute_rollout(ds, m, types.SimpleNamespace(cma_traj=traj_np, k_steps=6, device=dev, plot=False))
    rigid = obs_list[0][-1][:, mat] == 1
    poses = orc.rigid_body_trajectory(traj_np[:, 0], traj_np[:, 1], T, [0.5, 0.5, 0.4], obs_list[0][-1][rigid][:, cart])
    _, recs = orc.rollout(params, obs_list[0], poses, T, stats, bounds, 0.015, cart, [mat], ctrl, 2, 2, record=True)
    assert pred.shape == (T, obs_list[0].shape[1], obs_list[0].shape[2])
    np.testing.assert_allclose(pred, recs, rtol=0, atol=2e-5)
    # --- ground truth mode: the reference loop restated with oracle pieces
    steps = T - 6
    pred_gt = compute_rollout(ds, m, types.SimpleNamespace(cma_traj=None, k_steps=6, device=dev, plot=False))
    cur = obs_list[0].copy()
    ci, ui = list(cart), list(ctrl)
    for i in range(steps):
        gt = obs_list[i][-1]
        new_rigid = cur[-1][rigid].copy()
        new_rigid[:, ui] = gt[rigid][:, ui]
        cur[-1][rigid] = new_rigid
        np.testing.assert_allclose(pred_gt[i], cur[-1], rtol=0, atol=2e-5)
        nodes, ea, s, r, _ = orc.process(cur, None, stats, bounds, 0.015, cart, [mat], ctrl)
        acc = orc.epd_forward(params, nodes, ea, np.stack((s, r)), 2, 2)
        nxt = orc.get_position_from_prediction(stats, cart, acc, cur)
        cur[:-1] = cur[1:].copy()
        cur[-1][:, ci] = nxt
        new_rigid[:, ci] = gt[rigid][:, ci]
        cur[-1][rigid] = new_rigid


def test_training_loop_over_noisy_dataset(golden, dev, tmp_path):
    """train_dyn.py's loop end to end on the device: noisy CoffeeDataset -> GraphLoader (batch of 2) -> model forward
    with tape -> L1 loss -> HIP backward -> Adam; state_dict round trip through torch.save / load."""
    from gnn_manip_amd import CoffeeDataset, EncProcDecGNN, GraphLoader
    g = golden("g9_dataset.npz")
    root = str(tmp_path) + "/"
    _write_dataset(g, root)
    # noise well below the synthetic data's acceleration scale, so that the target stays learnable
    ds = CoffeeDataset(root, 6, 0.015, split="train", noise=1e-7, device=dev, use_control=True)
    ds.graph_attr.generator = torch.Generator(device=dev).manual_seed(5)
    loader = GraphLoader(ds, batch_size=2, shuffle=True, seed=1)
    torch.manual_seed(0)
    model = EncProcDecGNN(25, 4, 3, 128, 2, 2).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    crit = torch.nn.L1Loss(reduction="sum")
    first = last = None
    for epoch in range(6):
        tot = 0.0
        for batch in loader:
            pred = model.forward(batch.x, batch.edge_attr, batch.edge_index)
            loss = crit(pred, batch.y) / pred.shape[0]
            opt.zero_grad()
            loss.backward()
            opt.step()
            tot += float(loss.detach())
        first = tot if first is None else first
        last = tot
    assert np.isfinite(last) and last < first, (first, last)
    path = str(tmp_path) + "/gns_model.pth"
    torch.save(model.state_dict(), path)
    clone = EncProcDecGNN(25, 4, 3, 128, 2, 2).to(dev)
    clone.load_state_dict(torch.load(path, map_location=dev))
    b = next(iter(GraphLoader(ds, batch_size=1)))
    with torch.no_grad():
        np.testing.assert_array_equal(model.forward(b.x, b.edge_attr, b.edge_index).cpu().numpy(),
                                      clone.forward(b.x, b.edge_attr, b.edge_index).cpu().numpy())
