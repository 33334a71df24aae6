#!/usr/bin/env python3
"""Rollout-throughput bench for the MI355X rollout engine (contract: see DESIGN.md "Measurement").

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload target|c2|c3|c4|c5]

A "step" is one full rollout step of the hot path (state update -> node features -> radius graph -> destination sort
-> edge features -> encode / 10x process / decode -> Euler integration), the loop body of compute_rollout
(gnn_manip/utils/rollout_utils.py:38-61), on a seeded synthetic dense granular scene resident in HBM.  The K timed steps
run inside ONE library call (gm_rollout).  Default workload = the north_star target line (N = 100k, hidden 128, 10
message-passing steps); at N = 1 the other single-GPU configurations of BASELINE.json (C2, C3, C4) are measured in the
same process and reported as sub-records under "extra", each with its own roofline.

N > 1 = candidate-parallel (weak scaling): every rank rolls out its own candidates of the same scene; the scripted
trajectories are broadcast before and the per-candidate results all-gathered after the loop, over RCCL.
--workload c5 is BASELINE config C5's shape: 64 CMA-ES candidates x 200-step rollouts at N = 5k, 64 / N candidates per
rank in block-diagonal batches, one device Sinkhorn loss per candidate, broadcast / all-gather per generation.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Test hooks of the launcher's failure handling (tests/test_planner_dist.py, tests/test_gpu_planner.py): a rank that dies / never
# arrives.  Handled before the heavy imports, so that what the launcher sees does not depend on how long `import torch` takes.
if int(os.environ.get("WORLD_SIZE", "1")) > 1 and "RANK" in os.environ:
    if os.environ.get("GM_BENCH_FAIL_RANK") == os.environ["RANK"]:     # a rank that dies before the rendezvous
        raise SystemExit(3)
    if os.environ.get("GM_BENCH_STUCK_RANK") in (os.environ["RANK"], "all"):   # a rank that never reaches the rendezvous
        sys.stderr.write(f"rank {os.environ['RANK']}: GM_BENCH_STUCK_RANK set, sleeping\n")
        sys.stderr.flush()
        while True:
            time.sleep(1.0)

import numpy as np
import torch

WORKLOADS = {
    "target": dict(name="north_star target: N=100k dense synthetic scene, conn_r=0.015, max_neighbours=20, hidden=128, 10 MP steps",
                   n=100000, hidden=128, steps=20, warmup=5),
    # BASELINE.json configs[1..3]
    "c2": dict(name="C2: N=5k dense synthetic granular scene, conn_r=0.015, max_neighbours=20, hidden=128, 10 MP steps, 100-step rollout",
               n=5000, hidden=128, steps=100, warmup=10),
    "c3": dict(name="C3: N=50k dense synthetic scene, conn_r=0.015, max_neighbours=20, hidden=128, 10 MP steps", n=50000, hidden=128,
               steps=20, warmup=5),
    "c4": dict(name="C4: N=100k dense synthetic scene, hidden=256, 10 MP steps (MFMA-bound MLP run)", n=100000, hidden=256,
               steps=8, warmup=2),
    "c5": dict(name="C5: CMA-ES generation, 64 candidate rollouts x 200 steps at N=5k (hidden=128, 10 MP steps) + one Sinkhorn loss per "
                    "candidate, candidates sharded over the ranks", n=5000, hidden=128, steps=200, warmup=0),
}
MFMA_F32_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, exact fp32
MFMA_16BIT_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense bf16 / fp16 MFMA
HBM_PEAK_GBS = 8000.0


def edge_kernel_alg_flops(E, H=128, num_layers=2):
    """ALGORITHMIC flops of one processor edge-kernel launch: phi_e on cat[h_i, h_j, e] for E edges
    = 2*(3H*H + (num_layers-1)*H*H + H*H) per edge (10 H^2 at num_layers=2).
    The kernel ISSUES 2*3*H*H per edge (layer-1 node terms are factorised into the node kernel)."""
    return E * 2.0 * (3 * H * H + (num_layers - 1) * H * H + H * H)


def edge_kernel_issued_flops(E, H=128, num_layers=2):
    return E * 2.0 * (H * H * (num_layers + 1))


def edge_kernel_alg_bytes(E, N, H=128, m_steps=10):
    """Compulsory HBM bytes of one launch: read e, write e', P (2H per node), agg (H per node), indices -- averaged
    over the m_steps launches of a forward, the last of which writes no e' (nobody reads it: the decoder takes h, epd_gnn.py:96)."""
    return E * H * 4 * (2 - 1.0 / m_steps) + N * H * 4 * 3 + E * 12


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(obs, model, stats, scene, hidden):
    """The CPU restatement of the reference step: oracle graph build + featurisation (numpy, the
    KD-tree query of the reference is single-threaded too) and the plain-torch forward of oracle/torch_epd.py with all host
    threads, on the same scene and weights; bounded sample."""
    from oracle import epd_oracle as orc
    from oracle import torch_epd
    cores = os.cpu_count() or 1
    params = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    t0 = time.perf_counter()
    last = np.asarray(obs[-1][:, scene.CART], np.float32)
    s, r = orc.get_connectivity(last, scene.CONN_R, 20)
    t_graph = time.perf_counter() - t0
    t0 = time.perf_counter()
    nodes = orc.compute_nodes(obs, stats, scene.BOUNDS, scene.CONN_R, scene.CART, scene.MAT, scene.CTRL)
    ea = orc.get_edges_displacement(last, s, r, scene.CONN_R)
    t_feat = time.perf_counter() - t0
    tn, te, ti = torch.from_numpy(nodes), torch.from_numpy(ea), torch.from_numpy(np.stack((s, r)))
    # bounded sample of the forward: encoder + decoder alone (m_steps = 0) and with ONE of the ten identical message-passing
    # steps; forward = t0 + 10 (t1 - t0).  PyTorch's CPU ops stop scaling (and then slow down) well before a big host's
    # core count, so two thread counts are tried and the faster one is reported.
    def timed_forward(ms):
        t0 = time.perf_counter()
        o = torch_epd.epd_forward(params, tn, te, ti, 2, ms)
        return time.perf_counter() - t0, o
    best = None
    with torch.no_grad():
        for th in sorted({min(32, cores), cores}):
            torch.set_num_threads(th)
            if best is None:
                timed_forward(0)   # first touch of the weights / allocator warm-up
            t_encdec, out = timed_forward(0)
            t_one, _ = timed_forward(1)
            fwd = t_encdec + 10.0 * max(t_one - t_encdec, 0.0)
            if best is None or fwd < best[0]:
                best = (fwd, t_encdec, t_one, th, out)
    _, t_encdec, t_one, threads, _ = best
    # ... and the whole forward (all ten message-passing steps) TIMED once with that thread count: the baseline is a measured full
    # step, the two short passes above only choose the thread count
    with torch.no_grad():
        torch.set_num_threads(threads)
        t_forward, out = timed_forward(10)
    t0 = time.perf_counter()
    orc.get_position_from_prediction(stats, scene.CART, out.numpy(), obs)
    t_int = time.perf_counter() - t0
    step = t_graph + t_feat + t_forward + t_int
    return dict(value=1.0 / step, unit="rollout steps/s", cores=threads, host_logical_cpus=cores, kind="port", cpu=cpu_model_name(),
                graph_build_ms_single_thread=t_graph * 1e3, features_ms=t_feat * 1e3, forward_ms=t_forward * 1e3,
                integrate_ms=t_int * 1e3,
                forward_encoder_decoder_ms=t_encdec * 1e3, forward_one_mp_step_ms=max(t_one - t_encdec, 0.0) * 1e3,
                sample=f"ONE whole rollout step of the same scene and weights, timed end to end (N={obs.shape[1]}, E={len(s)}, hidden={hidden}): "
                       f"oracle graph build + features (numpy, single thread), the full oracle/torch_epd.py forward -- encoder, all 10 "
                       f"message-passing steps, decoder -- with torch.set_num_threads({threads}) (the faster of 32 and {cores} threads, chosen on a "
                       f"short pass: encoder + decoder + one step), integration")


def build_engine(wl, dev, rank, candidates, edge_kernel, total_steps):
    from gnn_manip_amd import EncProcDecGNN, GraphBoundedMultimaterialControl, RolloutEngine, scene
    n, hidden = wl["n"], wl["hidden"]
    # Stationary workload: a random-weight model would blow the pile apart within ~30 steps (edge count halves, kernels run
    # on a shrinking graph).  The decoder's output layer is scaled by 1e-5, the acceleration mean is zero and the initial
    # velocities are tiny, so the scene stays dense (E ~ 20 N) for the whole run; every kernel still runs the full
    # architecture on random weights.
    stats = dict(scene.STATS, acceleration_mean=[0.0, 0.0, 0.0])
    obs_np = scene.make_scene(n, seed=1000 + rank, vel_scale=1e-6)
    traj_np = scene.rigid_drift_trajectory(obs_np, total_steps, seed=2000 + rank, step_size=1e-6)
    torch.manual_seed(1234)
    model = EncProcDecGNN(25, 4, 3, hidden, 2, 10)
    with torch.no_grad():
        model.decoder[-1].weight.mul_(1e-5)
        model.decoder[-1].bias.mul_(1e-5)
    model = model.to(dev)
    model.set_edge_kernel(edge_kernel)
    ga = GraphBoundedMultimaterialControl(scene.CONN_R, stats, scene.CART, scene.MAT, scene.CTRL, scene.BOUNDS)
    eng = RolloutEngine(model, ga, n, device=dev, candidates=candidates)
    obs = torch.from_numpy(obs_np).to(dev)
    traj = torch.from_numpy(traj_np).to(dev)
    if candidates > 1:  # the same scene under `candidates` scripted trajectories, stored back to back
        nb = candidates
        obs = obs.unsqueeze(1).repeat(1, nb, 1, 1).reshape(obs.shape[0], nb * n, obs.shape[2]).contiguous()
        traj = traj.unsqueeze(1).repeat(1, nb, 1, 1)
        traj = (traj + 1e-7 * torch.arange(nb, device=dev).view(1, nb, 1, 1)).reshape(total_steps, -1, 3).contiguous()
    eng.set_scene(obs)
    return model, eng, obs, traj, obs_np, stats, scene


def resolve_kernel(edge_kernel, hidden):
    ek = edge_kernel
    if ek == "auto":
        ek = "sys" if hidden == 128 else "hm"
    return "sys" if ek == "sys_all" else ek   # sys_all: the systolic edge kernel, and the systolic node path at every size


def roofline_record(model, ek, hidden, edges, n_nodes, workload_key):
    n_launch, total_ms = model.profile_query(0)
    if n_launch < 0:   # more launches than the kind records (4096): the total covers the recorded ones
        n_launch = 4096
    k_ms = total_ms / max(n_launch, 1)
    alg = edge_kernel_alg_flops(edges, hidden)
    issued = edge_kernel_issued_flops(edges, hidden)
    # MFMA utilisation is priced on the flops the kernel ISSUES (3 HxH products per edge; SURVEY.md 8d "utilisation uses
    # F_issued"); the split-operand kernels issue 3 matrix-pipe product blocks (fp16 two-way split) per fp32-equivalent block
    # and are priced against the 16-bit pipe.
    mult = 3
    peak = MFMA_16BIT_PEAK_TFLOPS
    achieved = issued * mult / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
    kname = {"sys": "sys_edge_kernel", "hm": f"hm_edge_kernel<{hidden},false>"}[ek]
    alg_bytes = edge_kernel_alg_bytes(edges, n_nodes, hidden)
    hbm = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    pipe = "fp16 MFMA (2.5 PF dense)"
    mfma = {"achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "pipe": pipe,
            "floor_ms": issued * mult / (peak * 1e12) * 1e3}
    hbmr = {"achieved": hbm, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm / HBM_PEAK_GBS, "floor_ms": alg_bytes / (HBM_PEAK_GBS * 1e9) * 1e3}
    # the bound is the roof with the longer floor for this kernel and size: the fp16 x 3 scheme put the matrix-pipe floor of
    # the hidden-128 kernel below its HBM floor
    bound = "hbm" if hbmr["floor_ms"] >= mfma["floor_ms"] else "mfma"
    top = hbmr if bound == "hbm" else mfma
    rec = {"bound": bound, "kernel": kname + " (processor phi_e + scatter-add)",
           "achieved": top["achieved"], "peak": top["peak"], "unit": top["unit"], "frac": top["frac"], "traffic": None,
           "avg_launch_ms": k_ms, "launches_timed": n_launch,
           "issued_flops_per_launch": issued * mult, "fp32_equivalent_flops_per_launch": issued,
           "fp32_equivalent_tflops": issued / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0,
           "alg_flops_per_launch": alg, "alg_tflops": alg / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0,
           "alg_bytes_per_launch": alg_bytes, "mfma": mfma, "hbm": hbmr}
    # Two figures this run does not measure itself come from committed profile files and only while those files were collected from
    # THIS tree's kernel sources (each records their digest): the kernel's HBM traffic (rocprofv3 PMC passes, tools/profile_round.sh)
    # and the board power / shader clock under the workload (rocm-smi, tools/micro/power_watch.sh).  Otherwise: null / absent, with
    # the reason.  No literal of another run is printed.
    try:
        import glob
        from gnn_manip_amd.build import source_digest
        pfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_power.json")))
        if pfiles:
            pw = json.load(open(pfiles[-1]))
            pname = "profiles/" + os.path.basename(pfiles[-1])
            if pw.get("source_digest") != source_digest():
                rec["power_source"] = f"absent: {pname} was collected from other kernel sources than this tree's (digest mismatch)"
            elif pw.get(workload_key):
                rec["power"] = dict(pw[workload_key], source=f"{pname} (rocm-smi once a second during a long timed region of this "
                                                              "build, tools/micro/power_watch.sh; not re-measured by this run)")
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_traffic.json")))
        tr = json.load(open(files[-1]))
        name = "profiles/" + os.path.basename(files[-1])
        ent = tr.get(workload_key)
        if tr.get("source_digest") != source_digest():
            rec["traffic_source"] = f"null: {name} was collected from other kernel sources than this tree's (digest mismatch)"
        elif ent and ent.get("kernel") == kname:
            rec["traffic"] = ent["traffic_bytes_per_launch"]
            rec["traffic_source"] = (f"{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of this build, FETCH doubled "
                                     "per the guide)")
    except (OSError, ValueError, KeyError, IndexError):
        pass
    return rec, k_ms


def measure(wl_key, dev, rank, world, dist, cdev, args, steps, warmup, candidates=1):
    from gnn_manip_amd import _lib
    wl = WORKLOADS[wl_key]
    total = steps + warmup
    model, eng, obs, traj, obs_np, stats, scene = build_engine(wl, dev, rank, candidates, args.edge_kernel, total)
    ek = resolve_kernel(args.edge_kernel, wl["hidden"])
    L = _lib.lib()

    def barrier():
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        if warmup:
            eng.run(obs, traj[:warmup].contiguous(), warmup)
        eng.status()
        timed_traj = traj[warmup:].contiguous()
        if dist:
            # untimed warm-up of the exchange itself, same shapes as the timed one: the first broadcast / all-gather of a process
            # group sets up its channels (tens of milliseconds with RCCL -- a tenth of this timed region)
            w0 = timed_traj[0].to(cdev).clone()
            dist.broadcast(w0, src=0)
            w1 = obs[-1, :, 2:5].mean(dim=0).to(cdev)
            dist.all_gather([torch.empty_like(w1) for _ in range(world)], w1)
        barrier()
        coll = 0.0   # the timed region below carries no instrumentation: kernel timings come from a separate pass after it
        t0 = time.perf_counter()
        if dist:  # per-generation exchange of the candidate-parallel planner: scripted poses out ...
            first = timed_traj[0].to(cdev)
            dist.broadcast(first, src=0)
            timed_traj[0].copy_(first)
            torch.cuda.synchronize()
            coll += time.perf_counter() - t0
        eng.run(obs, timed_traj, steps)  # K steps, one library call
        result = obs[-1, :, 2:5].mean(dim=0)
        if dist:  # ... per-candidate results back
            result = result.to(cdev)
            torch.cuda.synchronize()   # this rank's rollout is done: what follows is exchange + waiting for the other ranks
            t1 = time.perf_counter()
            gathered = [torch.empty_like(result) for _ in range(world)]
            dist.all_gather(gathered, result)
            torch.cuda.synchronize()
            coll += time.perf_counter() - t1
        barrier()
        el = time.perf_counter() - t0
    edges = eng.status()  # edge count of the last timed step
    if dist:
        t = torch.tensor([el], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    rec = None
    if rank == 0:
        # Kernel timings: the same K steps once more, untimed, with HIP events (on the launch stream) around every model kernel
        # of this handle -- the dominant kernel's average launch duration for the roofline, the others for the breakdown.
        # A kind records at most 4096 scopes (gm_model_profile): the pass is bounded so that none fills up (a step opens ~31 node-side
        # scopes), and a kind that did fill up is reported as null instead of a figure that is too low.
        psteps = min(steps, 64)
        model.profile(31)
        with torch.no_grad():
            eng.run(obs, timed_traj[:psteps].contiguous(), psteps)
        torch.cuda.synchronize()
        model.profile(0)
        roof, k_ms = roofline_record(model, ek, wl["hidden"], edges, wl["n"] * candidates, wl_key)
        br = {"edge_kernel_ms_per_step": k_ms * 10, "profiled_steps": psteps}
        # node side; radius graph; encoders; and the rest of the step: state update + node features, destination sort + block
        # tables + edge features, clears, integration + window shift
        parts = [k_ms * 10]
        for name, kind in {"node_kernel": 1, "graph_build": 2, "encoder_kernels": 3, "csr_and_features": 4}.items():
            n_k, ms_k = model.profile_query(kind)   # totals over the profiled steps (a step's node side is several launches)
            full = n_k < 0   # more scopes than the kind records
            br[name + "_ms_per_step"] = None if full else ms_k / psteps
            br[name + "_launches_per_step"] = abs(n_k) / psteps
            parts.append(None if full else ms_k / psteps)
        # the parts are kernel time (HIP events of a separate, instrumented pass); ms_per_step is the wall time of the uninstrumented
        # timed region: their ratio says how much of the step is launch gaps / unattributed work
        br["sum_of_parts_ms_per_step"] = None if any(p is None for p in parts) else sum(parts)
        rec = {"value": world * candidates * steps / el, "unit": "rollout steps/s", "steps": steps, "warmup": warmup,
               "ms_per_step": el / steps * 1e3,
               "config": {"workload": wl["name"], "n_particles": wl["n"], "hidden": wl["hidden"], "edges_last_step": edges, "k_steps": 6,
                          "candidates_per_gpu": candidates, "parallelism": f"candidate-parallel x{world}",
                          "particle_ids": "random (scene.make_scene); the engine renumbers its working copy in grid-cell order per run() call, "
                                          "inside the timed region" if eng.renumber else "random (scene.make_scene), used as given"},
               "roofline": roof, "breakdown": br, "collective_ms": coll * 1e3}
        if br["sum_of_parts_ms_per_step"] is not None:
            br["sum_of_parts_over_step"] = br["sum_of_parts_ms_per_step"] / rec["ms_per_step"]
    return rec, (model, obs_np, stats, scene, ek)


def run_c5(dev, rank, world, dist, cdev, args):
    """One CMA-ES generation of BASELINE config C5: 64 candidates x 200 rollout steps at N = 5k sharded over the ranks in
    block-diagonal batches, one device Sinkhorn loss per candidate, candidates broadcast and losses all-gathered inside the
    timed region (planner.CandidateEvaluator)."""
    from gnn_manip_amd import GraphBoundedMultimaterialControl, RolloutEngine, planner
    from gnn_manip_amd.losses import SamplesLoss
    wl = WORKLOADS["c5"]
    popsize, horizon = args.candidates_total, args.steps if args.steps > 0 else wl["steps"]
    model, eng0, obs, traj, obs_np, stats, scn = build_engine(wl, dev, 0, 1, args.edge_kernel, horizon)
    del eng0
    per_rank = -(-popsize // world)
    batch = max(1, min(args.batch, per_rank))
    ga = GraphBoundedMultimaterialControl(scn.CONN_R, stats, scn.CART, scn.MAT, scn.CTRL, scn.BOUNDS)
    eng = RolloutEngine(model, ga, wl["n"], device=dev, candidates=batch)
    coffee_rows = torch.nonzero(obs[-1, :, 1] != 1).reshape(-1)   # once: a boolean-mask index synchronises per use
    target_cloud = (obs[-1].index_select(0, coffee_rows)[:, 2:5] + 0.01).contiguous()
    loss_fn = SamplesLoss("sinkhorn", p=2, blur=0.05)

    def objective(block):  # block: [b, dim] candidate parameters (here: an offset of the scripted cup drift per candidate)
        block = np.asarray(block, np.float64)
        ends = []
        for lo in range(0, block.shape[0], batch):
            cand = block[lo:lo + batch]
            b = cand.shape[0]
            offs = torch.as_tensor(np.asarray(cand[:, :3], np.float32), device=dev)
            if b < batch:
                offs = torch.cat((offs, offs[-1:].repeat(batch - b, 1)))
            trajs = (traj.unsqueeze(0) + 1e-6 * offs.view(batch, 1, 1, 3)).contiguous()
            with torch.no_grad():
                final = eng.rollout_candidates(obs, trajs, horizon)
            ends.append(final[:b, -1].index_select(1, coffee_rows)[:, :, 2:5])
        # the Sinkhorn losses of the rank's whole block in one batched launch sequence, one transfer of the values
        return loss_fn.batched(torch.cat(ends).contiguous(), target_cloud).double().cpu().numpy()

    ev = planner.CandidateEvaluator(None, result_dim=1, group=None, device=cdev)
    rng = np.random.Generator(np.random.PCG64(7))
    X = rng.standard_normal((popsize, 8))

    def generation(pop):  # broadcast of the population, contiguous block per rank, all-gather of the losses
        return ev.evaluate_blocks(pop, lambda lst: list(objective(np.stack(lst))) if lst else [])[:, 0]

    generation(X[:max(world, 1) * 1])  # warm-up: one candidate per rank
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    ev.collective_s = 0.0
    t0 = time.perf_counter()
    losses = generation(X)
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if dist:
        t = torch.tensor([el], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    if rank != 0:
        return None
    return {"metric": "rollout steps/sec (N particles, 10 MP steps, hidden=128)", "value": popsize * horizon / el, "unit": "rollout steps/s",
            "n_gpus": world, "steps": horizon, "warmup": 0, "ms_per_step": el / (per_rank * horizon) * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (seeded dense scene; random-init weights; a candidate = an offset of the scripted cup drift)",
            "config": {"workload": wl["name"], "n_particles": wl["n"], "candidates": popsize, "candidates_per_rank": per_rank,
                       "block_diagonal_batch": batch, "horizon": horizon, "generation_s": el, "loss_mean": float(np.mean(losses)),
                       "losses_finite": bool(np.isfinite(losses).all()), "parallelism": f"candidate-parallel x{world}"},
            # rank 0's wall time inside the generation's broadcast + all-gather calls (the all-gather includes the wait for the
            # slowest rank): what is not rollout or loss work when the 1 -> N curve falls short
            "collective_ms": ev.collective_s * 1e3, "collective_backend": dist.get_backend() if dist else None}


def extra_c5_1gpu(dev, args):
    """BASELINE config C5 at its own size on this ONE GPU: a CMA-ES generation of 64 candidates x 200 rollout steps at N = 5k in
    block-diagonal batches of 8 + the 64 device Sinkhorn losses in one batched launch sequence (run_c5 with one rank; what the
    8-GPU run shards 8 ways)."""
    import copy
    a = copy.copy(args)
    a.candidates_total, a.batch, a.steps = 64, 8, 200
    r = run_c5(dev, 0, 1, None, dev, a)
    return {"value": r["value"], "unit": r["unit"], "ms": r["config"]["generation_s"] * 1e3, "config": r["config"],
            "note": "the same record is `--workload c5` (add `--collectives always` for the RCCL exchange at one rank)"}


def extra_train(dev, steps=5, warmup=2):
    """A few training steps of the reference's configuration (examples/train_dyn.py:49-72 defaults: batch of 2 graphs, hidden
    128, 10 message-passing steps, L1 loss, Adam) on synthetic N = 5k scenes: forward with tape + HIP backward + optimiser."""
    from gnn_manip_amd import EncProcDecGNN, GraphBoundedMultimaterialControl, scene
    n, bsz, H, M = 5000, 2, 128, 10
    ga = GraphBoundedMultimaterialControl(0.015, scene.STATS, scene.CART, scene.MAT, scene.CTRL, scene.BOUNDS)
    batch = []
    for b in range(bsz):
        o = torch.from_numpy(scene.make_scene(n, seed=100 + b, side=0.152 * 0.8)).to(dev)
        batch.append((o, o[-1][:, 2:5] + 1e-4))
    with torch.no_grad():
        nodes, edge_attr, edge_index, tgt = ga.process_collate(batch)
    torch.manual_seed(0)
    model = EncProcDecGNN(25, 4, 3, H, 2, M).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)   # train_dyn.py:58 builds Adam(lr); fused = the same update in one launch
    crit = torch.nn.L1Loss(reduction="sum")

    def step():
        pred = model.forward(nodes, edge_attr, edge_index)
        loss = crit(pred, tgt) / pred.shape[0]
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    # Algorithmic HBM bytes of one step with a float32 tape: per processor block and edge 4H(5L + 12) bytes --
    # forward rows + tape 4H(L + 3), aggregation 4H, backward chain 4H(2L + 4), weight gradients 8H(L + 1), the two
    # segment sums of dz1 8H -- and per node 4H(5L + 16) on the node side; encoders / decoder the same way without the aggregation.
    L, E, N = 2, int(edge_attr.shape[0]), int(nodes.shape[0])
    per_block = 4 * H * ((5 * L + 12) * E + (5 * L + 16) * N)
    enc_dec = 4 * H * ((5 * L + 9) * (E + N) + (4 * L + 6) * N)
    alg_bytes = M * per_block + enc_dec
    flop = 3 * ((10 * H * H * E + 8 * H * H * N) * M + 2 * (4 * H + 2 * H * H) * E + 2 * (25 * H + 2 * H * H) * N + 2 * (2 * H * H + 3 * H) * N)
    return {"value": 1.0 / dt, "unit": "training steps/s", "ms": dt * 1e3, "loss": float(loss.detach()),
            "roofline": {"bound": "hbm", "achieved": alg_bytes / dt / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": alg_bytes / dt / 8e12,
                         "traffic": None, "alg_bytes_per_step": alg_bytes,
                         "mfma": {"alg_tflops": flop / dt / 1e12, "note": "forward + 2 x forward flop of the backward; run as six bf16 partial products per multiply"}},
            "arithmetic": "float32 results: operands split into three bf16 parts, six partial products on v_mfma_f32_32x32x16_bf16, fp32 accumulation",
            "config": {"workload": f"batch of {bsz} synthetic scenes x N={n} (collated), hidden={H}, {M} MP steps, L1 loss, Adam (fused=True)",
                       "nodes": int(nodes.shape[0]), "edges": int(edge_attr.shape[0]), "steps": steps, "warmup": warmup}}


def spawn_ranks(n, rank_timeout):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes of this script (one per GPU, the env
    rendezvous torch.distributed.run would have set up, 127.0.0.1), relay rank 0's JSON line, return non-zero if any
    rank failed.  The parent makes no GPU call and replaces no process (gpurun rules): children are ordinary
    subprocesses.  Nothing hangs silently: every rank's stderr (and the stdout of ranks > 0) goes to a file of its own, a rank
    that fails takes the others down, and after `rank_timeout` seconds the children still running -- exactly the PIDs started
    here -- are terminated (killed 10 s later if they ignore it), the exit status is 124 and the tails of all the rank files
    are printed with the failure."""
    import socket
    import subprocess
    import tempfile
    import threading
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    logdir = tempfile.mkdtemp(prefix="gm_bench_ranks_")
    procs, files = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        ferr = open(os.path.join(logdir, f"rank{r}.err"), "wb")
        fout = subprocess.PIPE if r == 0 else open(os.path.join(logdir, f"rank{r}.out"), "wb")
        files += [ferr] + ([] if r == 0 else [fout])
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=fout, stderr=ferr))
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc, why = 0, ""
    live = list(procs)
    deadline = time.monotonic() + rank_timeout

    def stop(ps):   # exactly the PIDs started above
        for q in ps:
            q.terminate()
        t_kill = time.monotonic() + 10.0
        for q in ps:
            try:
                q.wait(timeout=max(0.1, t_kill - time.monotonic()))
            except subprocess.TimeoutExpired:
                q.kill()
                q.wait()

    while live:   # a rank that fails takes the others down with it (they would wait in the rendezvous / a collective)
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc, why = code, f"rank {procs.index(p)} exited with status {code}"
                stop(live)
        if live and rc == 0 and time.monotonic() > deadline:
            stuck = [procs.index(p) for p in live]
            rc, why = 124, f"rank(s) {stuck} still running after --rank-timeout {rank_timeout:g} s: terminated"
            stop(live)
            live = []
    reader.join(timeout=10.0)
    for f in files:
        f.close()
    out0 = b"".join(chunks).decode(errors="replace")
    lines = [ln for ln in out0.splitlines() if ln.startswith("{")]
    if lines and rc == 0:
        print(lines[-1], flush=True)
    elif rc == 0:
        rc, why = 1, "rank 0 printed no JSON line"
    if rc:
        sys.stderr.write(f"bench.py: {why}\n")
        for r in range(n):   # what every rank said last
            for kind in ("out", "err"):
                path = os.path.join(logdir, f"rank{r}.{kind}")
                text = out0 if (r == 0 and kind == "out") else (open(path, errors="replace").read() if os.path.exists(path) else "")
                tail = text.strip().splitlines()[-15:]
                if tail:
                    sys.stderr.write(f"---- rank {r} std{kind} (last lines; files under {logdir})\n" + "\n".join(tail) + "\n")
        return 124 if rc == 124 else 1
    import shutil
    shutil.rmtree(logdir, ignore_errors=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (default: per workload)")
    ap.add_argument("--warmup", type=int, default=-1, help="untimed warm-up steps (default: per workload)")
    ap.add_argument("--workload", default="target", choices=sorted(WORKLOADS))
    ap.add_argument("--candidates", type=int, default=1,
                    help="candidate rollouts batched per GPU (block-diagonal); value counts candidates x steps")
    ap.add_argument("--candidates-total", type=int, default=64, help="c5: CMA-ES population size")
    ap.add_argument("--batch", type=int, default=8, help="c5: candidates per block-diagonal batch")
    ap.add_argument("--collectives", default="auto", choices=["auto", "always"],
                    help="always: initialise torch.distributed (nccl = RCCL) and run the per-generation broadcast / all-gather even "
                         "with ONE rank -- the multi-GPU code path, device-resident payloads included, on a one-GPU box")
    ap.add_argument("--rank-timeout", type=float, default=540.0,
                    help="seconds after which a multi-rank run that has not finished is ended with a non-zero exit (the launcher "
                         "terminates the ranks it started; a rank started by torch.distributed.run gives up its rendezvous / "
                         "collectives after the same time)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the C2 / C3 / C4 sub-records")
    ap.add_argument("--edge-kernel", default="auto", choices=["auto", "sys", "sys_all", "hm"],
                    help="processor edge kernel (per-model option): auto = systolic fp16 x 3 kernel for hidden 128 (DESIGN.md 5.1)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher (it never touches the GPU)
        raise SystemExit(spawn_ranks(args.gpus, args.rank_timeout))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher's environment says WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU: the product path has no CPU fallback"
    # GM_BENCH_REHEARSE=1: all ranks on cuda:0 with gloo (multi-rank rehearsal on a one-GPU box; RCCL refuses two ranks per
    # device).  Never set by the driver.
    rehearse = os.environ.get("GM_BENCH_REHEARSE") == "1"
    dev = torch.device("cuda:0" if rehearse else f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    dist = None
    if world > 1 or args.collectives == "always":
        import torch.distributed as dist
        if world == 1:   # a single rank has no launcher around it: rendezvous with itself on the loopback interface
            import socket
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        import datetime
        # a rendezvous / collective that never completes raises -- a minute after the launcher's own deadline, so that a run started
        # by spawn_ranks ends with the launcher's report (which rank was stuck) and one started by torch.distributed.run still ends
        tmo = datetime.timedelta(seconds=args.rank_timeout + 60.0)
        if rehearse:
            dist.init_process_group("gloo", timeout=tmo)
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)
    cdev = torch.device("cpu") if rehearse else dev  # where collective payloads live

    if args.workload == "c5":
        out = run_c5(dev, rank, world, dist, cdev, args)
        if rank == 0:
            print(json.dumps(out))
    else:
        wl = WORKLOADS[args.workload]
        steps = args.steps if args.steps > 0 else wl["steps"]
        warmup = args.warmup if args.warmup >= 0 else wl["warmup"]
        rec, (model, obs_np, stats, scene, ek) = measure(args.workload, dev, rank, world, dist, cdev, args, steps, warmup, args.candidates)
        if rank == 0:
            hidden = wl["hidden"]
            out = {"metric": f"rollout steps/sec (N particles, 10 MP steps, hidden={hidden})",
                   "value": rec["value"], "unit": "rollout steps/s", "n_gpus": world, "steps": steps, "warmup": warmup,
                   "ms_per_step": rec["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                   "dtype": "f32",
                   "arithmetic": "fp32 operands as two-way fp16 splits, three exact fp16 x fp16 partial products per multiply on the fp16 "
                                 "MFMA pipe, fp32 accumulation (1e-6 vs float64 through the model, like plain fp32)",
                   "data": "synthetic (seeded dense scene; random-init weights, decoder output layer scaled 1e-5 so the pile stays dense "
                           "over the rollout)",
                   "config": rec["config"], "roofline": rec["roofline"], "breakdown": rec["breakdown"]}
            if dist:
                out["collective_ms"] = rec["collective_ms"]   # rank 0: broadcast + all-gather (incl. waiting for the slowest rank)
                out["collective_backend"] = dist.get_backend()
            if world == 1 and not args.no_extra and args.workload == "target" and args.candidates == 1:
                extra = {}

                def sub(key, fn):   # a sub-record that fails is reported as such; the headline above stands on its own
                    try:
                        extra[key] = fn()
                    except Exception as e:   # noqa: BLE001
                        extra[key] = {"error": f"{type(e).__name__}: {e}"}
                    torch.cuda.empty_cache()

                for key in ("c2", "c3", "c4"):
                    w2 = WORKLOADS[key]
                    sub(key, lambda: measure(key, dev, rank, world, None, cdev, args, w2["steps"], w2["warmup"], 1)[0])
                sub("c5_1gpu", lambda: extra_c5_1gpu(dev, args))
                sub("train", lambda: extra_train(dev))
                out["extra"] = extra
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(obs_np, model, stats, scene, hidden)
            print(json.dumps(out))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
