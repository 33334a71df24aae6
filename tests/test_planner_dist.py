"""CPU: candidate-parallel evaluation over world_size-2 gloo (the N>1 path of bench.py / planner.py),
the host-side trajectory parameterisation, and the sharding arithmetic."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def test_shard_range_partitions_exactly():
    from gnn_manip_amd.planner import shard_range
    for n in (0, 1, 7, 64, 65):
        for w in (1, 2, 3, 8):
            blocks = [shard_range(n, w, r) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


def test_interpolate_trajectory_matches_reference_fixture(golden):
    from gnn_manip_amd.planner import interpolate_trajectory
    g = golden("g6_trajectory.npz")
    scale_ty, scale_rot, rx_init, max_rot, max_ty = g["scale_ty_eff"]
    for xk, rk, tk in (("x0", "traj_rot", "traj_ty"), ("x_pert", "traj_rot_pert", "traj_ty_pert")):
        x = g[xk]
        rot, ty = interpolate_trajectory(x, x.shape[0] // 2, rx_init, scale_rot, scale_ty, max_rot, max_ty)
        np.testing.assert_allclose(rot, g[rk], rtol=0, atol=1e-13)
        np.testing.assert_allclose(ty, g[tk], rtol=0, atol=1e-15)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, popsize, q):
    import torch.distributed as dist
    from gnn_manip_amd.planner import CandidateEvaluator, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []

    def objective(x):  # stands in for one candidate rollout: deterministic function of the candidate
        calls.append(float(x[0]))
        return [float(np.sum(x * x)), float(x[0])]

    ev = CandidateEvaluator(objective, result_dim=2)
    rng = np.random.Generator(np.random.PCG64(5))
    pop = rng.standard_normal((popsize, 6)) if rank == 0 else None  # only rank 0 owns the population
    res = ev.evaluate(pop)
    lo, hi = shard_range(popsize, world, rank)
    q.put((rank, res, len(calls), hi - lo))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("popsize", [7, 8])
def test_candidate_evaluator_gloo_world2(popsize):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, popsize, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.Generator(np.random.PCG64(5))
    pop = rng.standard_normal((popsize, 6))
    expect = np.stack((np.sum(pop * pop, axis=1), pop[:, 0]), axis=1)
    for rank, res, ncalls, nlocal in out:
        np.testing.assert_allclose(res, expect, rtol=0, atol=0)  # every rank gets the full result matrix
        assert ncalls == nlocal                                    # and evaluated only its own block


def test_candidate_evaluator_single_process():
    from gnn_manip_amd.planner import CandidateEvaluator
    ev = CandidateEvaluator(lambda x: [float(x.sum())], result_dim=1)
    pop = np.arange(12.0).reshape(4, 3)
    np.testing.assert_array_equal(ev.evaluate(pop)[:, 0], pop.sum(axis=1))


def _worker_cma(rank, world, port, q):
    """One CMA-ES generation loop whose populations are evaluated block-wise over the ranks (the shape of
    TrajectoryCMAsolver.population_losses: rank 0's candidates are broadcast, each rank evaluates its contiguous
    block in one batched call, every rank receives all losses)."""
    import torch.distributed as dist
    from gnn_manip_amd import cmaes
    from gnn_manip_amd.planner import CandidateEvaluator
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    blocks = []

    def block_fn(xs):  # stands in for one block-diagonal batched rollout + losses
        blocks.append(len(xs))
        return [float(np.sum((np.asarray(x) - 0.3) ** 2)) for x in xs]

    ev = CandidateEvaluator(None, result_dim=1)
    # every rank runs the same optimiser state machine; only rank 0's candidates count (they are broadcast)
    es = cmaes.CMAEvolutionStrategy([0.0] * 5, 0.3, {"seed": 11 + rank, "popsize": 9, "maxiter": 25})
    while not es.stop():
        X = es.ask()
        F = ev.evaluate_blocks(X if rank == 0 else None, block_fn).reshape(-1)
        Xs = [X]  # non-root ranks adopt rank 0's population so that the optimiser states stay identical
        dist.broadcast_object_list(Xs, src=0)
        es.tell(Xs[0], F.tolist())
    q.put((rank, es.result.fbest, es.result.xbest, blocks))
    dist.barrier()
    dist.destroy_process_group()


def test_cma_generation_loop_sharded_over_gloo_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_cma, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, f0, x0, b0), (r1, f1, x1, b1) = out
    assert f0 == f1 and np.array_equal(x0, x1)          # identical optimiser state on both ranks
    assert f0 < 1e-3 and np.abs(x0 - 0.3).max() < 0.05  # and it optimises
    assert set(b0) == {5} and set(b1) == {4}            # 9 candidates -> blocks of 5 and 4, one batched call each


def _bench(extra, env_extra, timeout=120):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"] + extra, env=env, capture_output=True, timeout=timeout)


def test_bench_launcher_deadline_without_a_gpu():
    """The launcher of `bench.py --gpus N` (spawn_ranks) makes no GPU call itself, so its failure handling runs here: ranks that
    never finish (stuck before anything touches the GPU) are terminated after --rank-timeout, the exit status is 124, no JSON line is
    printed, and the report names the ranks and quotes the tails of their output files."""
    import time
    t0 = time.monotonic()
    r = _bench(["--workload", "c2", "--rank-timeout", "8"], {"GM_BENCH_STUCK_RANK": "all"}, timeout=300)
    err = r.stderr.decode(errors="replace")
    assert r.returncode == 124, (r.returncode, err)
    assert time.monotonic() - t0 < 240   # (the launcher itself imports torch before it starts the ranks: slow on a cold box)
    assert not [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert "rank(s) [0, 1] still running after --rank-timeout 8 s" in err, err
    assert err.count("GM_BENCH_STUCK_RANK set, sleeping") == 2   # both ranks' stderr tails


@pytest.mark.skipif(torch.cuda.is_available(), reason="needs a box WITHOUT a GPU: the ranks must fail on their own")
def test_bench_launcher_reports_failing_ranks_without_a_gpu():
    """Without a GPU every rank stops at bench.py's own check (the product path has no CPU fallback): the launcher turns that into a
    non-zero exit with the rank's message, not a hang and not a JSON line."""
    r = _bench(["--workload", "c2", "--rank-timeout", "60"], {})
    err = r.stderr.decode(errors="replace")
    assert r.returncode == 1, (r.returncode, err)
    assert "exited with status" in err and "bench.py needs a GPU" in err, err
    assert not [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
