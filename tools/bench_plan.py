"""One CMA-ES generation of the planner on one GPU (BASELINE config C5's per-GPU share: 64 candidates over 8 GPUs =
8 candidates x 200 rollout steps per GPU, then the Sinkhorn loss of every candidate).  Prints one JSON line.

    python tools/bench_plan.py [--n 5000] [--candidates 8] [--horizon 200] [--generations 2]
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=5000)
    ap.add_argument("--candidates", type=int, default=8)
    ap.add_argument("--horizon", type=int, default=200)
    ap.add_argument("--generations", type=int, default=2)
    args = ap.parse_args()
    from gnn_manip_amd import EncProcDecGNN, GraphBoundedMultimaterialControl, scene
    from gnn_manip_amd.planner import TrajectoryCMAsolver
    dev = torch.device("cuda:0")
    side = 0.152 * (args.n / 5000.0) ** (1.0 / 3.0) * 0.8
    obs = torch.from_numpy(scene.make_scene(args.n, seed=7, side=side, vel_scale=1e-6)).to(dev)
    torch.manual_seed(0)
    model = EncProcDecGNN(25, 4, 3, 128, 2, 10).to(dev)
    with torch.no_grad():
        model.decoder[4].weight.mul_(1e-5)  # stationary scene with random weights (see bench.py)
        model.decoder[4].bias.zero_()
    stats = dict(scene.STATS)
    stats["acceleration_mean"] = [0.0, 0.0, 0.0]
    ga = GraphBoundedMultimaterialControl(0.015, stats, scene.CART, scene.MAT, scene.CTRL, scene.BOUNDS)
    state = (obs, obs[-1][:, 2:5].clone())
    s = TrajectoryCMAsolver(model, ga, state, 180, [0.5, 0.5, 0.4], scale_rot=1.0, scale_ty=1.0, alpha=0.1, beta=1000.0, gamma=0.05,
                            penalty=1.0, rho=0.0, device=dev, cma_iter=args.generations, cma_popsize=args.candidates,
                            total_steps=args.horizon, candidates_per_gpu=args.candidates)
    sample = np.stack((180.0 - 0.01 * np.arange(args.horizon + 1), 1e-6 * np.arange(args.horizon + 1)), axis=1)
    s.set_sample_traj(sample)
    coffee = obs[-1][obs[-1][:, 1] == 0][:, 2:5]
    s.desired_pos = (coffee + 0.002).contiguous()
    x0 = np.concatenate((s.sample_traj[:, 0], s.sample_traj[:, 1]))
    X = [x0 * (1 + 0.01 * i) for i in range(args.candidates)]
    s.population_losses(X)  # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.generations):
        losses = s.population_losses(X)
    torch.cuda.synchronize()
    gen = (time.perf_counter() - t0) / args.generations
    # loss share
    ends = [coffee + 0.001 * i for i in range(args.candidates)]
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for e in ends:
        s.loss(e.contiguous(), s.desired_pos).item()
    torch.cuda.synchronize()
    loss_t = time.perf_counter() - t1
    print(json.dumps({"metric": "CMA-ES generation time per GPU", "value": gen, "unit": "s", "higher_is_better": False,
                      "config": {"workload": f"{args.candidates} candidates x {args.horizon} steps, N={args.n}, hidden=128, 10 MP",
                                 "coffee_particles": int(coffee.shape[0])},
                      "rollout_steps_per_s": args.candidates * args.horizon / gen, "sinkhorn_losses_s": loss_t,
                      "loss_example": float(losses[0])}))


if __name__ == "__main__":
    main()
