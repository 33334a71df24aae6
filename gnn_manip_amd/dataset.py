"""On-disk formats of the reference and the training loader, device-resident -- mirror of
``gnn_manip/utils/coffee_dataset.py`` (``read_metadata``, ``CoffeeDataset``) plus the batching the reference
gets from ``torch_geometric.data.DataLoader`` (examples/train_dyn.py:8,49-53).

MI355X layout: every simulation file is parsed once and kept as one ``[T, N, D]`` float32 tensor in HBM
(288 GB per GPU hold whole datasets); a sample is a window view of it, turned into a graph by the HIP
kernels of graph.py when it is requested.  The reference materialises every window on the host instead
(k-fold duplication, coffee_dataset.py:82-102).
"""
import json

import numpy as np
import torch

from .graph import GraphBoundedMultimaterial, GraphBoundedMultimaterialControl


def read_metadata(metadata_file):
    """coffee_dataset.py:18-43: (data_dim, time_steps, cartesian_idx, control_idx, material_id, bounds, stats)."""
    with open(metadata_file) as fp:
        metadata = json.load(fp)
    b = torch.tensor(metadata["bounds"])
    bounds = {"upper_bounds": b[:, 1], "lower_bounds": b[:, 0]}
    stats = {"velocity_mean": torch.tensor(metadata["vel_mean"]), "velocity_std": torch.tensor(metadata["vel_std"]),
             "acceleration_mean": torch.tensor(metadata["acc_mean"]), "acceleration_std": torch.tensor(metadata["acc_std"])}
    return (metadata["data_dim"], metadata["sequence_length"], metadata["cartesian_idx"], metadata["control_idx"],
            metadata["material_id"], bounds, stats)


def read_simulation(file, time_steps, data_dim):
    """One ``particles_%06d.csv`` (rows = time-major particles, no header) -> float64 [T, N, data_dim]."""
    import pandas as pd
    return np.array(pd.read_csv(file, header=None)).reshape(time_steps, -1, data_dim)


class GraphData:
    """The fields of ``torch_geometric.data.Data`` that train_dyn.py:49-53 reads (x, edge_attr, edge_index, y)."""

    def __init__(self, x, edge_attr, edge_index, y=None):
        self.x, self.edge_attr, self.edge_index, self.y = x, edge_attr, edge_index, y

    def to(self, device):
        mv = lambda t: None if t is None else t.to(device)
        return GraphData(mv(self.x), mv(self.edge_attr), mv(self.edge_index), mv(self.y))

    @property
    def num_nodes(self):
        return int(self.x.shape[0])


def collate_graphs(items):
    """Batch of GraphData -> one GraphData, edge indices offset by the node counts before them (the
    torch_geometric batching rule; for equal-sized graphs the N*i rule of collate_utils.py:75-76)."""
    xs, es, idx, ys, off = [], [], [], [], 0
    for g in items:
        xs.append(g.x)
        es.append(g.edge_attr)
        idx.append(g.edge_index + off)
        ys.append(g.y)
        off += g.x.shape[0]
    y = torch.cat(ys) if ys and ys[0] is not None else None
    return GraphData(torch.cat(xs), torch.cat(es), torch.cat(idx, dim=1), y)


class CoffeeDataset(torch.utils.data.Dataset):
    """Mirror of coffee_dataset.py:46-133 (same constructor arguments); ``device`` must be a HIP device."""

    def __init__(self, root, k, conn_r, split='train', noise=None, device=torch.device('cuda:0'), transform=None,
                 use_control=False, max_neighbours=20):
        assert split in ['train', 'test']
        import pandas as pd
        self.dir, self.split = root, split
        simulation_data = np.array(pd.read_csv(f'{self.dir}{self.split}/sim_data.csv', header=None))
        self.files = [f'{self.dir}{self.split}/particles_{int(sim_id):06d}.csv' for (sim_id, *_) in simulation_data]
        self.metadata_file = f'{self.dir}metadata.json'
        self.k, self.conn_r, self.max_neighbours = k, conn_r, max_neighbours
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("CoffeeDataset: gnn_manip_amd builds graphs on the HIP device only")
        self.noise, self.use_control = noise, use_control
        (self.data_dim, self.time_steps, self.cartesian_idx, self.control_idx, self.material_id, self.bounds,
         self.stats) = read_metadata(self.metadata_file)
        self._load_data(self.files)
        self.graph_attr = self._get_graph_attr()

    def _load_data(self, files):
        # one resident [T, N, D] tensor per simulation; samples index (simulation, first frame)
        self.sims, self.index = [], []
        for s, file in enumerate(files):
            data = read_simulation(file, self.time_steps, self.data_dim)
            self.sims.append(torch.from_numpy(data).float().to(self.device))
            self.index += [(s, t) for t in range(self.time_steps - self.k)]

    def __len__(self):
        return len(self.index)

    def sample(self, idx):
        """(obs_seq [k, N, D(+3)], next_pos [N, 3]) of sample idx, as coffee_dataset.py:82-102 stores them."""
        s, t = self.index[idx]
        data = self.sims[s]
        c0 = self.cartesian_idx[0]
        obs_seq = data[t:t + self.k]
        next_pos = data[t + self.k][:, c0:c0 + 3]
        if self.use_control:
            # control input = next position - position for rigid-body particles (material == 1), else 0
            ctr = next_pos.unsqueeze(0) - obs_seq[:, :, c0:c0 + 3]
            ctr = torch.where((obs_seq[:, :, self.material_id] != 1).unsqueeze(-1), torch.zeros_like(ctr), ctr)
            obs_seq = torch.cat((obs_seq, ctr), dim=-1)
        # independent tensors, like the reference's stored samples: an in-place rollout step on a sample must not touch
        # the resident simulation (the windows overlap)
        return obs_seq.clone().contiguous(), next_pos.clone().contiguous()

    def __getitem__(self, idx):
        obs_seq, next_pos = self.sample(idx)
        nodes, edge_attr, senders, receivers, tgt = self.graph_attr.process(obs_seq, next_pos)
        return GraphData(nodes, edge_attr, torch.stack((senders, receivers)).long(), tgt)

    def _get_graph_attr(self):
        if self.use_control:
            return GraphBoundedMultimaterialControl(conn_r=self.conn_r, stats=self.stats, cartesian_idx=self.cartesian_idx,
                                                    material_idx=[self.material_id], control_idx=self.control_idx,
                                                    bounds=self.bounds, noise=self.noise, max_neighbours=self.max_neighbours)
        return GraphBoundedMultimaterial(conn_r=self.conn_r, stats=self.stats, cartesian_idx=self.cartesian_idx,
                                         material_idx=[self.material_id], bounds=self.bounds, noise=self.noise,
                                         max_neighbours=self.max_neighbours)


class CoffeeTestDataset(CoffeeDataset):
    """Mirror of coffee_dataset.py:136-215: ONE simulation (``sim_id``) of a split; items are the raw
    ``(obs_seq [k, N, D(+3)], next_pos [N, 3])`` pairs the rollout / planner start from (rollout_utils.py:110-130,
    optimise_traj.py:253-259)."""

    def __init__(self, directory, k, conn_r, split='train', noise=None, max_neighbours=20, device=torch.device('cuda:0'),
                 use_control=False, sim_id=1):
        self.dir, self.split = directory, split
        self.files = [f'{self.dir}{self.split}/particles_{sim_id:06d}.csv']
        self.metadata_file = f'{self.dir}metadata.json'
        self.k, self.conn_r, self.max_neighbours = k, conn_r, max_neighbours
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("CoffeeTestDataset: gnn_manip_amd builds graphs on the HIP device only")
        self.noise, self.use_control = noise, use_control
        (self.data_dim, self.time_steps, self.cartesian_idx, self.control_idx, self.material_id, self.bounds,
         self.stats) = read_metadata(self.metadata_file)
        self._load_data(self.files)
        self.graph_attr = self._get_graph_attr()

    def __getitem__(self, index):
        return self.sample(index)


class GraphLoader:
    """What train_dyn.py gets from ``torch_geometric.data.DataLoader(dataset, batch_size, shuffle)``: an iterable of
    collated batches.  Everything stays on the dataset's device."""

    def __init__(self, dataset, batch_size=2, shuffle=False, seed=None):
        self.dataset, self.batch_size, self.shuffle = dataset, batch_size, shuffle
        self.rng = np.random.default_rng(seed)

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        order = self.rng.permutation(len(self.dataset)) if self.shuffle else np.arange(len(self.dataset))
        for b in range(0, len(order), self.batch_size):
            yield collate_graphs([self.dataset[int(i)] for i in order[b:b + self.batch_size]])
