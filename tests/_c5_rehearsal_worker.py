"""Rank program of tests/test_gpu_planner.py::test_c5_two_ranks_on_one_gpu: the real TrajectoryCMAsolver.population_losses
(block-diagonal batched rollouts + one device Sinkhorn loss per candidate, traj_utils.py:114-159,257) under torch.distributed
with the gloo backend, every rank on cuda:0 (RCCL refuses two ranks per device; the sharding / broadcast / all-gather code is
the same).  Rank 0 writes the losses of the whole population."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    out_path = sys.argv[1]
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    import test_gpu_planner as tp
    s, _, _ = tp._solver(dev, cands=2)
    s.collective_device = torch.device("cpu")   # collective payloads of the gloo rehearsal live on the host
    x0 = np.concatenate((s.sample_traj[:, 0], s.sample_traj[:, 1]))
    rng = np.random.default_rng(6)
    X = [x0 + 0.05 * rng.standard_normal(x0.shape) for _ in range(7)]   # 7 candidates: ranks get 4 + 3 (ragged blocks)
    losses = s.population_losses(X if rank == 0 else [np.zeros_like(x0)] * 7)   # only rank 0's population counts
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump([float(v) for v in losses], f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
