"""Training-step benchmark: forward with tape + HIP backward + Adam on the
reference's training configuration (examples/train_dyn.py defaults: batch of 2 graphs, hidden 128,
10 message-passing steps), synthetic scenes.  Prints one JSON line.

    python tools/bench_train.py [--n 5000] [--batch 2] [--steps 20] [--warmup 3] [--no-cpu-baseline]
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=5000)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--adam", choices=["foreach", "fused"], default="foreach", help="torch.optim.Adam implementation (the reference's train_dyn.py takes the default: foreach)")
    args = ap.parse_args()
    from gnn_manip_amd import EncProcDecGNN, GraphBoundedMultimaterialControl, scene
    dev = torch.device("cuda:0")
    side = 0.152 * (args.n / 5000.0) ** (1.0 / 3.0) * 0.8
    ga = GraphBoundedMultimaterialControl(0.015, scene.STATS, scene.CART, scene.MAT, scene.CTRL, scene.BOUNDS)
    batch = []
    for b in range(args.batch):
        obs = torch.from_numpy(scene.make_scene(args.n, seed=100 + b, side=side)).to(dev)
        batch.append((obs, obs[-1][:, 2:5] + 1e-4))
    with torch.no_grad():
        nodes, edge_attr, edge_index, tgt = ga.process_collate(batch)  # collate_utils.py:68-87 on the device
    n, e = int(nodes.shape[0]), int(edge_attr.shape[0])
    H, M = args.hidden, 10
    torch.manual_seed(0)
    model = EncProcDecGNN(25, 4, 3, H, 2, M).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, **({"fused": True} if args.adam == "fused" else {}))
    crit = torch.nn.L1Loss(reduction="sum")

    def step():
        pred = model.forward(nodes, edge_attr, edge_index)
        loss = crit(pred, tgt) / pred.shape[0]
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    # forward-only share
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        model.forward(nodes, edge_attr, edge_index)
    torch.cuda.synchronize()
    fwd = (time.perf_counter() - t1) / args.steps
    f_fwd = (10 * H * H * e + 8 * H * H * n) * M + 2 * (4 * H + 2 * H * H) * e + 2 * (25 * H + 2 * H * H) * n + 2 * (2 * H * H + 3 * H) * n
    out = {"metric": "training steps/sec (forward + backward + Adam)", "value": 1.0 / dt, "unit": "steps/s", "ms_per_step": dt * 1e3,
           "forward_ms": fwd * 1e3, "backward_plus_optimizer_ms": (dt - fwd) * 1e3, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"batch of {args.batch} scenes x N={args.n} (collated), hidden={H}, 10 MP steps", "nodes": n, "edges": e},
           "alg_tflops": 3 * f_fwd / dt / 1e12, "loss": float(loss.detach())}
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT))
        from oracle import torch_epd
        p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
        xn, xe, xi, xt = nodes.cpu(), edge_attr.cpu(), edge_index.cpu(), tgt.cpu()
        t2 = time.perf_counter()
        o = torch_epd.epd_forward(p, xn, xe, xi, 2, M)
        l = torch.nn.functional.l1_loss(o, xt, reduction="sum") / o.shape[0]
        l.backward()
        cpu = time.perf_counter() - t2
        out["cpu_baseline"] = {"value": 1.0 / cpu, "unit": "steps/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": "1 forward+backward of oracle/torch_epd.py (PyTorch float32 CPU autograd), same batch and weights"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
