"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A plain numpy restatement of the gnn-manip rollout hot path:
radius graph -> node/edge featurisation -> encode-process-decode GNN -> semi-implicit
Euler integration -> rollout state update -> rigid-body (cup) trajectory.

This module is the *checker*.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  Nothing under ``gnn_manip_amd/``
imports it: the product path is the HIP library and fails loudly without it.

Pinning (see DESIGN.md "Oracle"):
  * every function except the two message-passing blocks is pinned against outputs of
    the reference's own Python, imported by file path in the build container by
    ``tests/golden/make_golden.py`` and committed as ``tests/golden/*.npz``;
  * the arithmetic inside ``InteractionNetwork`` / ``GraphIndependent`` lives in the
    third-party ``dblanm/torch-graphnet`` submodule, which is ABSENT from the reference
    tree (empty ``deps/torch-graphnet``; version not recoverable).  For those two blocks
    parity is UNPINNED: semantics follow BASELINE.json's north_star
    (``e' = phi_e([h_i, h_j, e_ij])``, scatter-add at i, ``h' = phi_v([h_i, agg_i])``,
    PyG source_to_target: j = edge_index[0], i = edge_index[1]).  The reference's
    *wiring* around the blocks (epd_gnn.py) is pinned by fixture G7.

All reference citations are relative to /root/reference.
"""
import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------------------
# K1  radius graph  (gnn_manip/utils/utils.py:64-93)
# --------------------------------------------------------------------------------------
def _sq_dists_f64(q, p):
    """d2[a,b] = sum_j ((double)q[a,j]-(double)p[b,j])^2 accumulated j = 0,1,2 in order.

    Follows sklearn's EuclideanDistance64.rdist (metrics/_dist_metrics.pxd: sequential
    ``d += tmp*tmp`` loop) on the float64-upcast data KDTree stores (utils.py:76).
    """
    q = np.asarray(q, dtype=np.float64)
    p = np.asarray(p, dtype=np.float64)
    d2 = np.zeros((q.shape[0], p.shape[0]), dtype=np.float64)
    for j in range(q.shape[1]):
        t = q[:, j:j + 1] - p[None, :, j]
        d2 = d2 + t * t  # separate multiply and add: no FMA in numpy
    return d2


def get_connectivity(pos_nodes, conn_r, max_neighbours=20, chunk=1024):
    """Reference ``get_connectivity`` (utils.py:64-93).

    For every query node i (the reference calls it the *sender*): all j with
    ``d2(i,j) <= conn_r*conn_r`` (float64; KDTree.query_radius leaf test), ordered by
    ascending distance (sort_results=True, utils.py:78), first ``max_neighbours`` kept
    (utils.py:80-85).  senders = repeat(i, len_i) (utils.py:87-88), receivers = concat of
    the per-query lists (utils.py:90-91).

    Ties in distance have no defined order in the reference (unstable quicksort over
    tree traversal order); the oracle, like the HIP kernel, breaks ties on the smaller
    index.  Fixtures assert tie-freeness.
    Returns int64 arrays (senders, receivers).
    """
    pos = np.asarray(pos_nodes)
    n = pos.shape[0]
    r = float(conn_r)
    r2 = r * r
    use_cells = n > 4096
    if use_cells:
        return _get_connectivity_cells(pos, r, max_neighbours)
    send, recv = [], []
    for a in range(0, n, chunk):
        d2 = _sq_dists_f64(pos[a:a + chunk], pos)
        for row in range(d2.shape[0]):
            idx = np.nonzero(d2[row] <= r2)[0]
            order = np.lexsort((idx, d2[row, idx]))  # primary d2, secondary index
            idx = idx[order][:max_neighbours]
            send.append(np.full(idx.shape[0], a + row, dtype=np.int64))
            recv.append(idx.astype(np.int64))
    if not send:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    return np.concatenate(send), np.concatenate(recv)


def _get_connectivity_cells(pos, r, max_neighbours):
    """Same result as the brute-force path, candidates from a uniform cell list."""
    n = pos.shape[0]
    p64 = pos.astype(np.float64)
    r2 = r * r
    h = r * 1.001
    lo = p64.min(axis=0)
    cell = np.floor((p64 - lo) / h).astype(np.int64)
    dims = cell.max(axis=0) + 1
    key = (cell[:, 2] * dims[1] + cell[:, 1]) * dims[0] + cell[:, 0]
    order = np.argsort(key, kind="stable")
    skey = key[order]
    ncell = int(dims[0] * dims[1] * dims[2])
    start = np.searchsorted(skey, np.arange(ncell + 1))
    send, recv = [], []
    # group queries by cell so candidate sets are shared
    ucells = np.unique(key)
    res = [None] * n
    for c in ucells:
        cx = c % dims[0]
        cy = (c // dims[0]) % dims[1]
        cz = c // (dims[0] * dims[1])
        cand = []
        for dz in (-1, 0, 1):
            z = cz + dz
            if z < 0 or z >= dims[2]:
                continue
            for dy in (-1, 0, 1):
                y = cy + dy
                if y < 0 or y >= dims[1]:
                    continue
                x0 = max(cx - 1, 0)
                x1 = min(cx + 1, dims[0] - 1)
                c0 = (z * dims[1] + y) * dims[0] + x0
                c1 = (z * dims[1] + y) * dims[0] + x1
                cand.append(order[start[c0]:start[c1 + 1]])
        cand = np.sort(np.concatenate(cand))
        q = order[start[c]:start[c + 1]]
        d2 = _sq_dists_f64(pos[q], pos[cand])
        for row, qi in enumerate(q):
            m = np.nonzero(d2[row] <= r2)[0]
            o = np.lexsort((cand[m], d2[row, m]))
            res[qi] = cand[m][o][:max_neighbours].astype(np.int64)
    lens = np.array([len(x) for x in res], dtype=np.int64)
    senders = np.repeat(np.arange(n, dtype=np.int64), lens)
    receivers = np.concatenate(res) if n else np.zeros(0, np.int64)
    return senders, receivers


def connectivity_is_tie_free(pos_nodes, conn_r, max_neighbours=20):
    """True when no query has two in-radius neighbours at equal sqrt-distance among its
    first max_neighbours+1 (the reference sorts on sqrt(d2): _binary_tree.pxi query_radius)."""
    pos = np.asarray(pos_nodes)
    r2 = float(conn_r) * float(conn_r)
    for a in range(0, pos.shape[0], 1024):
        d2 = _sq_dists_f64(pos[a:a + 1024], pos)
        for row in range(d2.shape[0]):
            d = np.sort(np.sqrt(d2[row][d2[row] <= r2]))[:max_neighbours + 1]
            if d.shape[0] > 1 and np.any(d[1:] == d[:-1]):
                return False
    return True


# --------------------------------------------------------------------------------------
# K2  edge features  (utils.py:43-61)
# --------------------------------------------------------------------------------------
def get_edges_displacement(last_pos, senders, receivers, conn_r):
    """``[(p_s - p_r)/conn_r, ||.||_2]`` in float32 (utils.py:52-59)."""
    last_pos = np.asarray(last_pos, dtype=F32)
    ps = last_pos[np.asarray(senders)]
    pr = last_pos[np.asarray(receivers)]
    disp = (ps - pr) / F32(conn_r)
    sq = disp * disp
    acc = sq[:, 0]
    for j in range(1, sq.shape[1]):
        acc = acc + sq[:, j]
    dist = np.sqrt(acc)[:, None]
    return np.concatenate((disp, dist), axis=-1).astype(F32)


# --------------------------------------------------------------------------------------
# K3  node features  (utils.py:27-40, collate_utils.py:195-232)
# --------------------------------------------------------------------------------------
def get_nodes_vel(pos_seq, velocity_mean, velocity_std):
    """diff over time, normalise, [N, (k-1)*dim] time-major within node (utils.py:27-40)."""
    pos_seq = np.asarray(pos_seq, dtype=F32)
    vel = pos_seq[1:] - pos_seq[:-1]
    vel = (vel - np.asarray(velocity_mean, F32)) / np.asarray(velocity_std, F32)
    vel = np.transpose(vel, (1, 0, 2))
    return np.ascontiguousarray(vel).reshape(vel.shape[0], -1)


def compute_nodes(obs, stats, bounds, conn_r, cartesian_idx, material_idx, control_idx=None):
    """GraphBoundedMultimaterial(Control).compute_nodes (collate_utils.py:195-208, 217-232)."""
    obs = np.asarray(obs, dtype=F32)
    pos_seq = obs[:, :, cartesian_idx]
    last_pos = pos_seq[-1]
    vel_attr = get_nodes_vel(pos_seq, stats["velocity_mean"], stats["velocity_std"])
    lower = last_pos - np.asarray(bounds["lower_bounds"], F32)
    upper = np.asarray(bounds["upper_bounds"], F32) - last_pos
    b = np.concatenate((lower, upper), axis=1) / F32(conn_r)
    b = np.clip(b, F32(-1), F32(1))
    mat = obs[-1][:, material_idx]
    parts = [vel_attr, b, mat]
    if control_idx is not None:
        ctrl = (obs[-1][:, control_idx] - np.asarray(stats["velocity_mean"], F32)) \
            / np.asarray(stats["velocity_std"], F32)
        parts.append(ctrl)
    return np.concatenate(parts, axis=-1).astype(F32)


def compute_acceleration(next_pos, pos_seq):
    """utils.py:10-24."""
    next_pos = np.asarray(next_pos, F32)
    pos_seq = np.asarray(pos_seq, F32)
    return next_pos - F32(2) * pos_seq[-1] + pos_seq[-2]


def compute_target(obs, tgt, stats, cartesian_idx):
    """GraphSimple.compute_target (collate_utils.py:150-159)."""
    pos_seq = np.asarray(obs, F32)[:, :, cartesian_idx]
    acc = compute_acceleration(tgt, pos_seq)
    return (acc - np.asarray(stats["acceleration_mean"], F32)) / np.asarray(stats["acceleration_std"], F32)


def process(obs, tgt, stats, bounds, conn_r, cartesian_idx, material_idx, control_idx=None,
            max_neighbours=20):
    """GraphAttributes._process_simple (collate_utils.py:29-40).

    NB the reference does not forward max_neighbours here (always the default 20,
    collate_utils.py:34); callers wanting reference behaviour pass 20.
    """
    obs = np.asarray(obs, F32)
    last_pos = obs[-1][:, cartesian_idx]
    nodes = compute_nodes(obs, stats, bounds, conn_r, cartesian_idx, material_idx, control_idx)
    senders, receivers = get_connectivity(last_pos, conn_r, max_neighbours)
    edge_attr = get_edges_displacement(last_pos, senders, receivers, conn_r)
    nodes_tgt = compute_target(obs, tgt, stats, cartesian_idx) if tgt is not None else None
    return nodes, edge_attr, senders, receivers, nodes_tgt


def process_collate(batch, **kw):
    """GraphAttributes.process_collate (collate_utils.py:68-87): per-graph offset N*i."""
    nl, el, il, tl = [], [], [], []
    for i, (obs, tgt) in enumerate(batch):
        nodes, edge_attr, s, r, t = process(obs, tgt, **kw)
        ei = np.stack((s, r)) + nodes.shape[0] * i
        nl.append(nodes)
        el.append(edge_attr)
        il.append(ei)
        tl.append(t)
    tgt = np.concatenate(tl) if tl[0] is not None else None
    return np.concatenate(nl), np.concatenate(el), np.concatenate(il, axis=1), tgt


def random_walk_noise(pos_seq, noise_std, noise_sample):
    """random_walk_noise (utils.py:96-115) with the Normal(0, noise_std / sqrt(k-1)) draw passed in
    (`noise_sample`, shape [k-1, N, 3]): velocity noise = cumsum over time, position noise = cumsum of
    that, zero for the first frame."""
    noise_sample = np.asarray(noise_sample, F32)
    noisy_vel = np.cumsum(noise_sample, axis=0, dtype=F32)
    noisy_pos = np.cumsum(noisy_vel, axis=0, dtype=F32)
    return np.concatenate((np.zeros((1,) + noisy_pos.shape[1:], F32), noisy_pos), axis=0)


def process_noisy(obs, tgt, noise_sample, stats, bounds, conn_r, cartesian_idx, material_idx, control_idx=None,
                  max_neighbours=20):
    """GraphBoundedMultimaterial._process_noisy (collate_utils.py:169-193): the position columns of the whole
    window and the target get the random-walk noise; features, graph and target come from the noisy state."""
    obs = np.asarray(obs, F32)
    seq = random_walk_noise(obs[:, :, cartesian_idx], None, noise_sample)
    noise = np.zeros_like(obs)
    noise[:, :, cartesian_idx] = seq
    noisy_obs = obs + noise
    last_pos = noisy_obs[-1][:, cartesian_idx]
    noisy_tgt = np.asarray(tgt, F32) + seq[-1]
    nodes = compute_nodes(noisy_obs, stats, bounds, conn_r, cartesian_idx, material_idx, control_idx)
    senders, receivers = get_connectivity(last_pos, conn_r, max_neighbours)
    edge_attr = get_edges_displacement(last_pos, senders, receivers, conn_r)
    return nodes, edge_attr, senders, receivers, compute_target(noisy_obs, noisy_tgt, stats, cartesian_idx)


# --------------------------------------------------------------------------------------
# on-disk formats (gnn_manip/utils/coffee_dataset.py)
# --------------------------------------------------------------------------------------
def read_metadata(metadata):
    """read_metadata (coffee_dataset.py:18-43) on the parsed metadata.json dict."""
    b = np.asarray(metadata["bounds"], F32)
    bounds = {"upper_bounds": b[:, 1], "lower_bounds": b[:, 0]}
    stats = {"velocity_mean": np.asarray(metadata["vel_mean"], F32), "velocity_std": np.asarray(metadata["vel_std"], F32),
             "acceleration_mean": np.asarray(metadata["acc_mean"], F32), "acceleration_std": np.asarray(metadata["acc_std"], F32)}
    return (metadata["data_dim"], metadata["sequence_length"], metadata["cartesian_idx"], metadata["control_idx"],
            metadata["material_id"], bounds, stats)


def dataset_samples(sim_tables, time_steps, data_dim, k, cartesian_idx, material_id, use_control):
    """CoffeeDataset._load_data (coffee_dataset.py:73-102): every window of k frames of every simulation and the
    position that follows it; with use_control the 3 control columns (next position - position for rigid rows
    (material == 1), zero elsewhere) are appended to every frame of the window."""
    obs_list, next_list = [], []
    for table in sim_tables:
        data = np.asarray(table, np.float64).reshape(time_steps, -1, data_dim)
        pos = data[:, :, cartesian_idx]
        for t in range(time_steps - k):
            obs = data[t:t + k].astype(F32)
            nxt = pos[t + k].astype(F32)
            if use_control:
                ctr = nxt[None] - obs[:, :, cartesian_idx]
                ctr[obs[:, :, material_id] != 1] = 0
                obs = np.concatenate((obs, ctr.astype(F32)), axis=-1)
            obs_list.append(obs)
            next_list.append(nxt)
    return obs_list, next_list


# --------------------------------------------------------------------------------------
# K4-K9  encode-process-decode  (gnn_manip/models/epd_gnn.py)
# --------------------------------------------------------------------------------------
def layer_norm(x, weight, bias, eps=1e-5):
    """torch.nn.LayerNorm over the last dim (epd_gnn.py:60-61,80-81), float32."""
    x = np.asarray(x, F32)
    mean = x.mean(axis=-1, keepdims=True, dtype=F32)
    xc = x - mean
    var = (xc * xc).mean(axis=-1, keepdims=True, dtype=F32)
    return (xc / np.sqrt(var + F32(eps))) * weight + bias


def mlp(params, prefix, x, num_layers, norm):
    """``_build_mlp`` (epd_gnn.py:72-84): Linear ReLU [Linear ReLU]*(L-1) Linear [LayerNorm].

    Sequential indices: linears at 0,2,...,2L; LayerNorm at 2L+1.
    """
    h = np.asarray(x, F32)
    for li in range(num_layers + 1):
        w = params[f"{prefix}.{2 * li}.weight"]
        b = params[f"{prefix}.{2 * li}.bias"]
        h = h @ w.T + b
        if li < num_layers:
            h = np.maximum(h, F32(0))
    if norm:
        h = layer_norm(h, params[f"{prefix}.{2 * num_layers + 1}.weight"],
                       params[f"{prefix}.{2 * num_layers + 1}.bias"])
    return h.astype(F32)


def graph_independent(params, prefix, x, edge_attr, num_layers):
    """GraphIndependent block (call site epd_gnn.py:88): independent node / edge MLPs."""
    return (mlp(params, f"{prefix}.phi_node", x, num_layers, True),
            mlp(params, f"{prefix}.phi_edge", edge_attr, num_layers, True))


def interaction_network(params, prefix, h, e, edge_index, num_layers, flow="source_to_target", concat=("i", "j", "e"),
                        node_concat=("h", "agg")):
    """InteractionNetwork block (call site epd_gnn.py:101); semantics per north_star.

    Default convention: j = edge_index[0] (source), i = edge_index[1] (target / aggregation index),
    e' = phi_e(cat[h_i, h_j, e]); agg_i = sum_{e->i} e'; h' = phi_v(cat[h, agg]).
    No residual inside the block (added by the caller, epd_gnn.py:103-104).
    The block's source is absent from the reference tree: flow / concat / node_concat restate the other conventions a
    PyG-style block could have (flow 'target_to_source': i = edge_index[0]; concat: order of (h_i, h_j, e); node_concat:
    order of (h, agg)), so that the product's switch for them can be checked.
    """
    if flow == "source_to_target":
        j, i = np.asarray(edge_index[0]), np.asarray(edge_index[1])
    else:
        i, j = np.asarray(edge_index[0]), np.asarray(edge_index[1])
    parts = {"i": h[i], "j": h[j], "e": e}
    e_in = np.concatenate([parts[c] for c in concat], axis=-1)
    e_new = mlp(params, f"{prefix}.phi_edge", e_in, num_layers, True)
    agg = np.zeros_like(h)
    np.add.at(agg, i, e_new)
    nparts = {"h": h, "agg": agg}
    h_in = np.concatenate([nparts[c] for c in node_concat], axis=-1)
    h_new = mlp(params, f"{prefix}.phi_node", h_in, num_layers, True)
    return h_new, e_new


def epd_forward(params, nodes, edge_attr, edge_index, num_layers=2, m_steps=10, **convention):
    """EncProcDecGNN.forward (epd_gnn.py:86-105)."""
    h, e = graph_independent(params, "encoder", nodes, edge_attr, num_layers)
    for k in range(m_steps):
        hn, en = interaction_network(params, f"processor.{k}", h, e, edge_index, num_layers, **convention)
        h = hn + h  # epd_gnn.py:103
        e = en + e  # epd_gnn.py:104
    return mlp(params, "decoder", h, num_layers, False)


def init_params(node_dim, edge_dim, out_dim, hidden, num_layers, m_steps, seed):
    """Deterministic, platform-independent weights (numpy PCG64) in state_dict naming.

    Distribution mimics torch's default Linear init (uniform +-1/sqrt(fan_in)); LayerNorm
    weights are perturbed from (1, 0) so that affine parity is actually exercised.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    p = {}

    def lin(name, fin, fout):
        bound = 1.0 / np.sqrt(fin)
        p[f"{name}.weight"] = rng.uniform(-bound, bound, (fout, fin)).astype(F32)
        p[f"{name}.bias"] = rng.uniform(-bound, bound, (fout,)).astype(F32)

    def build(prefix, fin, fout, norm):
        lin(f"{prefix}.0", fin, hidden)
        for li in range(1, num_layers):
            lin(f"{prefix}.{2 * li}", hidden, hidden)
        lin(f"{prefix}.{2 * num_layers}", hidden, fout)
        if norm:
            p[f"{prefix}.{2 * num_layers + 1}.weight"] = (1.0 + 0.1 * rng.standard_normal(fout)).astype(F32)
            p[f"{prefix}.{2 * num_layers + 1}.bias"] = (0.1 * rng.standard_normal(fout)).astype(F32)

    build("encoder.phi_edge", edge_dim, hidden, True)
    build("encoder.phi_node", node_dim, hidden, True)
    for k in range(m_steps):
        build(f"processor.{k}.phi_edge", 3 * hidden, hidden, True)
        build(f"processor.{k}.phi_node", 2 * hidden, hidden, True)
    build("decoder", hidden, out_dim, False)
    return p


# --------------------------------------------------------------------------------------
# K10  integrator  (gnn_manip/utils/rollout_utils.py:145-158)
# --------------------------------------------------------------------------------------
def get_position_from_prediction(stats, cartesian_idx, pred_acc, obs_seq):
    pred_acc = np.asarray(pred_acc, F32)
    obs_seq = np.asarray(obs_seq, F32)
    acc = pred_acc * np.asarray(stats["acceleration_std"], F32) + np.asarray(stats["acceleration_mean"], F32)
    last_pos = obs_seq[-1][:, cartesian_idx]
    last_vel = last_pos - obs_seq[-2][:, cartesian_idx]
    vel = last_vel + acc
    return (last_pos + vel).astype(F32)


# --------------------------------------------------------------------------------------
# a12  rigid-body trajectory  (traj_utils.py:87-103,167-228; rollout_utils.py:161-205)
# --------------------------------------------------------------------------------------
def compute_particles_tmatrix(rotation, translation, ty_init, rigid_particles):
    """traj_utils.py:167-194 == rollout_utils.py:178-205 (rotation about X, y/z swap)."""
    rp = np.asarray(rigid_particles, F32)
    c = F32(np.cos(rotation))
    s = F32(np.sin(rotation))
    T = np.array([[1, 0, 0, ty_init[0]],
                  [0, c, -s, ty_init[1] + translation],
                  [0, s, c, ty_init[2]],
                  [0, 0, 0, 1]], dtype=F32)
    init = np.ones((4, rp.shape[0]), dtype=F32)
    init[0] = F32(ty_init[0]) - rp[:, 0]
    init[1] = F32(ty_init[1]) - rp[:, 2]
    init[2] = F32(ty_init[2]) - rp[:, 1]
    tp = T @ init
    out = np.zeros_like(rp)
    out[:, 0] = tp[0]
    out[:, 2] = tp[1]
    out[:, 1] = tp[2]
    return out


def set_sample_traj(sample_traj, scale_rot, scale_ty):
    """TrajectoryCMAsolver.set_sample_traj (traj_utils.py:199-204)."""
    d = sample_traj[2:] - sample_traj[1:-1]
    return np.stack((np.deg2rad(d[:, 0] / scale_rot), d[:, 1] / scale_ty)).T


def interpolate_trajectory(x, n_points, rx_init, scale_rot, scale_ty, max_rot, max_ty):
    """TrajectoryCMAsolver.interpolate_trajectory (traj_utils.py:206-228).

    rx_init, max_rot in radians (already deg2rad'ed by the solver ctor, traj_utils.py:31,54).
    """
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


def rigid_body_trajectory(traj_rot, traj_ty, horizon, ty_init, rigid_particles):
    """get_rigid_body_trajectory_from_diff body (traj_utils.py:97-99): [horizon, Nr, 3]."""
    return np.stack([compute_particles_tmatrix(traj_rot[i], traj_ty[i], ty_init, rigid_particles)
                     for i in range(horizon)])


# --------------------------------------------------------------------------------------
# a11  rollout loop  (rollout_utils.py:38-61 == traj_utils.py:123-152)
# --------------------------------------------------------------------------------------
def rollout(params, obs0, trajectory, horizon, stats, bounds, conn_r, cartesian_idx, material_idx,
            control_idx, num_layers=2, m_steps=10, max_neighbours=20, record=False,
            forward_fn=None):
    """cma_objective's loop (traj_utils.py:119-152).  obs0: [k, N, D] float32.

    Returns the final state [k, N, D] (and per-step last-frame records if asked).
    ``forward_fn(nodes, edge_attr, edge_index)`` defaults to the oracle's epd_forward.
    """
    obs = np.array(obs0, dtype=F32, copy=True)
    mat_col = material_idx[0]
    rigid = obs[-1][:, mat_col] == 1
    ci = list(cartesian_idx)
    ui = list(control_idx)
    recs = []
    if forward_fn is None:
        def forward_fn(n, ea, ei):
            return epd_forward(params, n, ea, ei, num_layers, m_steps)
    for i in range(horizon):
        new_rigid = obs[-1][rigid].copy()
        cur = obs[-1][rigid][:, ci]
        if i >= trajectory.shape[0]:
            new_rigid[:, ui] = cur  # traj_utils.py:130-131
        else:
            new_rigid[:, ui] = trajectory[i] - cur  # traj_utils.py:133
        obs[-1][rigid] = new_rigid
        if record:
            recs.append(obs[-1].copy())
        nodes, edge_attr, s, r, _ = process(obs, None, stats, bounds, conn_r, cartesian_idx,
                                            material_idx, control_idx, max_neighbours)
        pred = forward_fn(nodes, edge_attr, np.stack((s, r)))
        next_pos = get_position_from_prediction(stats, cartesian_idx, pred, obs)
        obs[:-1] = obs[1:].copy()
        last = obs[-1]
        last[:, ci] = next_pos
        if i < trajectory.shape[0]:
            new_rigid[:, ci] = trajectory[i]
        last[rigid] = new_rigid
    if record:
        return obs, np.stack(recs)
    return obs


# --------------------------------------------------------------------------------------
# planner loss (gnn_manip/utils/traj_utils.py:69,161-165,230-285)
# --------------------------------------------------------------------------------------
def sinkhorn_divergence(x, y, blur=0.05, scaling=0.5, dtype=np.float64, diameter=None):
    """geomloss.SamplesLoss(loss="sinkhorn", p=2, blur=blur) between two uniform clouds (call sites
    traj_utils.py:69,279).  geomloss is an un-vendored, un-pinned pip dependency (environment.yml:25): this is a
    restatement of its published algorithm (Feydy et al. 2019; geomloss sinkhorn_divergence.py `sinkhorn_loop` /
    `sinkhorn_cost`, tensorized backend): C = |x-y|^2 / 2, eps-scaling from diameter^2 to blur^2, symmetrised
    log-domain updates with debiasing, one final extrapolation, S = <a, b_x - a_x> + <b, a_y - b_y>.
    PARITY UNPINNED against geomloss itself (absent); pinned by known answers (tests)."""
    x = np.asarray(x, dtype)
    y = np.asarray(y, dtype)
    n, m = x.shape[0], y.shape[0]
    both = np.concatenate((x, y)).astype(np.float32)
    if diameter is None:   # geomloss: max_diameter of the two clouds unless the caller names one (`diameter=` keyword)
        diameter = float(np.sqrt(((both.max(0) - both.min(0)).astype(np.float32) ** 2).sum(dtype=np.float32)))
    diameter = float(np.float32(diameter))
    if diameter == 0.0:
        return 0.0
    eps_s = [diameter ** 2] + [float(np.exp(e)) for e in np.arange(2 * np.log(diameter), 2 * np.log(blur), 2 * np.log(scaling))] + [blur ** 2]

    def cost(a, b):
        return ((a[:, None, :] - b[None, :, :]) ** 2).sum(-1) / 2

    def softmin(eps, C, h):  # -eps * logsumexp_j (h_j - C_ij / eps)
        v = h[None, :] - C / eps
        mx = v.max(axis=1)
        return -eps * (mx + np.log(np.exp(v - mx[:, None]).sum(axis=1)))

    C_xx, C_yy, C_xy, C_yx = cost(x, x), cost(y, y), cost(x, y), cost(y, x)
    la, lb = np.full(n, -np.log(n), dtype), np.full(m, -np.log(m), dtype)
    eps = eps_s[0]
    a_x, b_y = softmin(eps, C_xx, la), softmin(eps, C_yy, lb)
    a_y, b_x = softmin(eps, C_yx, la), softmin(eps, C_xy, lb)
    for eps in eps_s:
        at_x, bt_y = softmin(eps, C_xx, la + a_x / eps), softmin(eps, C_yy, lb + b_y / eps)
        at_y, bt_x = softmin(eps, C_yx, la + b_x / eps), softmin(eps, C_xy, lb + a_y / eps)
        a_x, b_y = 0.5 * (a_x + at_x), 0.5 * (b_y + bt_y)
        a_y, b_x = 0.5 * (a_y + at_y), 0.5 * (b_x + bt_x)
    a_x, b_y = softmin(eps, C_xx, la + a_x / eps), softmin(eps, C_yy, lb + b_y / eps)
    a_y, b_x = softmin(eps, C_yx, la + b_x / eps), softmin(eps, C_xy, lb + a_y / eps)
    return float((b_x - a_x).mean() + (a_y - b_y).mean())


def compute_vel_acc(actions):
    """traj_utils.py:161-165."""
    actions = np.asarray(actions, np.float64)
    return actions[1:] - actions[:-1], actions[2:] - 2 * actions[1:-1] + actions[:-2]


def planner_penalties(actions, rx_init, rotation_limit, vel_scale, acc_scale):
    """(vel_loss, acc_loss, bound_penalty) of compute_loss (traj_utils.py:230-285): Frobenius norms of the
    finite differences normalised per column (CMAESolver: max_rot / max_ty; TrajectoryCMAsolver: fixed means,
    traj_utils.py:343-365), and the 20.0 rotation-range penalty."""
    vel, acc = compute_vel_acc(actions)
    vel_loss = float(np.linalg.norm(vel / np.asarray(vel_scale, np.float64)[None, :]))
    acc_loss = float(np.linalg.norm(acc / np.asarray(acc_scale, np.float64)[None, :]))
    rot = np.asarray(actions, np.float64)[:, 0]
    penalty = 20.0 if (rot.max() > rx_init + rotation_limit or rot.min() < rx_init - rotation_limit) else 0.0
    return vel_loss, acc_loss, penalty


# ---- InterpolatedCMAsolver (traj_utils.py:288-452): host-side float64 pieces
def interp_set_sample_traj(sample_traj, nr_traj_points, rx_init, ty_init0, scale_rot, scale_ty):
    """traj_utils.py:296-304: every nr_traj_points-th pose of the sample becomes a key point, scaled."""
    sample_traj = np.asarray(sample_traj, np.float64)
    pts = sample_traj[list(range(nr_traj_points, sample_traj.shape[0], nr_traj_points)), :]
    return np.stack(((np.deg2rad(pts[:, 0]) - rx_init) / scale_rot, (pts[:, 1] - ty_init0) / scale_ty)).T


def interp_interpolate_trajectory(x, n_key, horizon, nr_traj_points, rx_init, scale_rot, scale_ty):
    """traj_utils.py:393-416 (pchip branch): key points -> one pose per step by PCHIP interpolation."""
    from scipy.interpolate import pchip_interpolate
    x = np.asarray(x, np.float64)
    rot_points = [rx_init] + (rx_init + x[:n_key] * scale_rot).tolist()
    ty_points = [0.0] + (x[n_key:] * scale_ty).tolist()
    traj_idx = np.arange(0, horizon + 1, nr_traj_points)
    idx = np.arange(horizon)
    return pchip_interpolate(traj_idx, rot_points, idx), pchip_interpolate(traj_idx, ty_points, idx)


def interp_vel_noninterp(x, traj_points, nr_traj_points, scale_rot, scale_ty, max_rot, max_ty):
    """compute_vel_noninterp (traj_utils.py:418-436): exp of the largest key-point increment excess."""
    x = np.asarray(x, np.float64)
    rot, ty = x[:traj_points] * scale_rot, x[traj_points:] * scale_ty
    ineq_rot = np.abs(rot[1:] - rot[:-1]) - max_rot * nr_traj_points
    ineq_ty = np.abs(ty[1:] - ty[:-1]) - max_ty * nr_traj_points
    return float(np.exp(max(ineq_rot.max(), ineq_ty.max())))


def interp_ineq_constraint(x, traj_points, nr_traj_points, scale_rot, scale_ty, max_rot, max_ty):
    """ineq_constraint (traj_utils.py:366-391): |key-point increments| - limits, de-normalised, rotation then translation."""
    x = np.asarray(x, np.float64)
    actions = np.zeros((traj_points + 1, 2))
    actions[1:, 0] = x[:traj_points] * scale_rot
    actions[1:, 1] = x[traj_points:] * scale_ty
    vel = actions[1:] - actions[:-1]
    upper = np.abs(vel) - np.array([max_rot * nr_traj_points, max_ty * nr_traj_points])
    return np.concatenate((upper[:, 0] / scale_rot, upper[:, 1] / scale_ty))
