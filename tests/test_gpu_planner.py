"""Planner on the device: Sinkhorn loss kernel against the oracle restatement and known
answers; TrajectoryCMAsolver's batched objective against one-by-one evaluation; a short optimisation run."""
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


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("n,m,seed", [(150, 130, 1), (1000, 777, 2), (3, 5, 3), (2500, 2500, 4)])
def test_sinkhorn_kernel_vs_oracle(dev, n, m, seed):
    """float32 HIP kernels vs the float64 numpy restatement: 1e-4 relative (log-sum-exp over up to 2500 terms in
    float32 with __expf, ~20 Sinkhorn iterations)."""
    from gnn_manip_amd.losses import SamplesLoss
    rng = np.random.default_rng(seed)
    x = (0.5 + 0.05 * rng.standard_normal((n, 3))).astype(np.float32)
    y = (0.53 + 0.07 * rng.standard_normal((m, 3))).astype(np.float32)
    got = float(SamplesLoss(loss="sinkhorn", p=2, blur=.05)(_t(x, dev), _t(y, dev)).item())
    ref = orc.sinkhorn_divergence(x, y, blur=0.05)
    assert abs(got - ref) <= 1e-4 * abs(ref) + 1e-9


@pytest.mark.parametrize("n,m,seed,blur", [(400, 350, 11, 0.05), (1500, 1500, 12, 0.05), (900, 1100, 13, 0.02)])
def test_sinkhorn_kernel_vs_float64_dense_log_domain(dev, n, m, seed, blur):
    """The float32 device kernels (distances recomputed per pair, two-pass log-sum-exp per row) against a dense float64 log-domain
    evaluation of the same schedule (oracle.sinkhorn_divergence: full N x M cost matrices in float64): 1e-5 relative on the
    divergence, which is itself a difference of potentials two orders of magnitude larger."""
    from gnn_manip_amd.losses import SamplesLoss
    rng = np.random.default_rng(seed)
    x = (0.5 + 0.05 * rng.standard_normal((n, 3))).astype(np.float32)
    y = (0.52 + 0.06 * rng.standard_normal((m, 3))).astype(np.float32)
    got = float(SamplesLoss(loss="sinkhorn", p=2, blur=blur)(_t(x, dev), _t(y, dev)).item())
    ref = orc.sinkhorn_divergence(x, y, blur=blur)
    assert abs(got - ref) <= 1e-5 * abs(ref), (got, ref, abs(got - ref) / abs(ref))


def test_sinkhorn_kernel_known_answers(dev):
    from gnn_manip_amd.losses import SamplesLoss
    rng = np.random.default_rng(5)
    x = (0.5 + 0.05 * rng.standard_normal((600, 3))).astype(np.float32)
    loss = SamplesLoss(loss="sinkhorn", p=2, blur=.05)
    assert abs(float(loss(_t(x, dev), _t(x, dev)).item())) < 1e-7
    t = np.array([0.03, -0.02, 0.05], np.float32)
    np.testing.assert_allclose(float(loss(_t(x, dev), _t(x + t, dev)).item()), 0.5 * float((t ** 2).sum()), rtol=3e-3)
    with pytest.raises(NotImplementedError):
        SamplesLoss(loss="energy")


def test_sinkhorn_batched_equals_one_call_per_pair(dev):
    """gm_sinkhorn_divergence_batched (the losses of a block of candidates in one launch sequence): pair b gets the epsilon
    schedule of a call on it alone -- the clouds below have diameters from 0.1 to 3, i.e. schedules of different lengths, one
    pair is a single repeated point (diameter 0: loss 0) -- and its loss equals that call's bit for bit, with the desired cloud
    shared by the batch (the planner's case) and with one per pair; every value also against the float64 oracle."""
    from gnn_manip_amd.losses import SamplesLoss
    rng = np.random.default_rng(21)
    n, m, B = 700, 620, 6
    spreads = [0.02, 0.3, 0.05, 1.0, 0.1, 0.0]
    X = np.stack([(0.5 + sp * rng.standard_normal((n, 3))).astype(np.float32) for sp in spreads])
    y = (0.52 + 0.04 * rng.standard_normal((m, 3))).astype(np.float32)
    Y = np.stack([(0.5 + (sp + 0.01) * rng.standard_normal((m, 3))).astype(np.float32) for sp in spreads])
    Y[5] = X[5][:m]   # pair 5 (per-pair case): every point of both clouds is the same point
    loss = SamplesLoss(loss="sinkhorn", p=2, blur=.05)
    got_shared = loss.batched(_t(X, dev), _t(y, dev)).cpu().numpy()
    got_own = loss.batched(_t(X, dev), _t(Y, dev)).cpu().numpy()
    assert got_shared.shape == (B,) and got_shared.dtype == np.float32
    for b in range(B):
        one_s = loss(_t(X[b], dev), _t(y, dev)).cpu().numpy()
        one_o = loss(_t(X[b], dev), _t(Y[b], dev)).cpu().numpy()
        assert got_shared[b].tobytes() == one_s.tobytes(), (b, got_shared[b], one_s)
        assert got_own[b].tobytes() == one_o.tobytes(), (b, got_own[b], one_o)
        ref = orc.sinkhorn_divergence(X[b], y, blur=0.05)
        assert abs(got_shared[b] - ref) <= 2e-5 * abs(ref) + 1e-9, (b, got_shared[b], ref)
    assert got_own[5] == 0.0
    # a different batch around the same pair changes nothing for that pair
    sub = loss.batched(_t(X[[3, 0]], dev), _t(y, dev)).cpu().numpy()
    assert sub[0].tobytes() == got_shared[3].tobytes() and sub[1].tobytes() == got_shared[0].tobytes()


def test_sinkhorn_diameter_keyword_and_bad_input(dev):
    """geomloss's `diameter=` keyword: the schedule starts from the caller's diameter (no host synchronisation in the library);
    against the oracle restated with the same keyword.  A non-finite coordinate is reported, not propagated."""
    from gnn_manip_amd._lib import GMError
    from gnn_manip_amd.losses import SamplesLoss
    rng = np.random.default_rng(22)
    x = (0.5 + 0.05 * rng.standard_normal((500, 3))).astype(np.float32)
    y = (0.53 + 0.06 * rng.standard_normal((450, 3))).astype(np.float32)
    for d in (0.8, 0.2):
        got = float(SamplesLoss(loss="sinkhorn", p=2, blur=.05, diameter=d)(_t(x, dev), _t(y, dev)).item())
        ref = orc.sinkhorn_divergence(x, y, blur=0.05, diameter=d)
        assert abs(got - ref) <= 2e-5 * abs(ref), (d, got, ref)
    xb = x.copy()
    xb[17, 1] = np.nan
    with pytest.raises(GMError):
        SamplesLoss(loss="sinkhorn", p=2, blur=.05).batched(_t(np.stack((x, xb)), dev), _t(y, dev))


def _solver(dev, n=400, horizon=5, cands=4):
    from gnn_manip_amd import EncProcDecGNN, GraphBoundedMultimaterialControl, scene
    from gnn_manip_amd.planner import TrajectoryCMAsolver
    obs = scene.make_scene(n, seed=31, side=0.06)
    params = orc.init_params(25, 4, 3, 128, 2, 2, 31)
    params["decoder.4.weight"] = params["decoder.4.weight"] * 1e-3  # keep the random-weight scene from exploding
    m = EncProcDecGNN(25, 4, 3, 128, 2, 2)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    m = m.to(dev)
    ga = GraphBoundedMultimaterialControl(0.015, STATS, CART, MAT, CTRL, BOUNDS)
    state = (_t(obs, dev), _t(obs[-1, :, 2:5], dev))
    s = TrajectoryCMAsolver(m, ga, state, 180, [0.5, 0.5, 0.4], scale_rot=1.0, scale_ty=1.0, alpha=0.1, beta=1000.0, gamma=0.05,
                            penalty=1.0, rho=0.0, device=dev, cma_iter=3, cma_popsize=6, total_steps=horizon,
                            candidates_per_gpu=cands)
    sample = np.stack((180.0 - 0.4 * np.arange(horizon + 1), 1e-4 * np.arange(horizon + 1)), axis=1)
    s.set_sample_traj(sample)
    coffee = obs[-1, obs[-1, :, 1] == 0][:, 2:5]
    s.desired_pos = _t(coffee + np.float32(0.002), dev)
    return s, params, obs


def test_batched_population_equals_one_by_one(dev):
    """cma_objective per candidate (the reference's way) == the block-diagonal batched evaluation of the population."""
    s, _, _ = _solver(dev)
    rng = np.random.default_rng(6)
    x0 = np.concatenate((s.sample_traj[:, 0], s.sample_traj[:, 1]))
    X = [x0 + 0.05 * rng.standard_normal(x0.shape) for _ in range(6)]
    one = [s.cma_objective(x) for x in X]
    batch = s.population_losses(X)
    np.testing.assert_allclose(batch, one, rtol=2e-5)


def test_objective_matches_oracle_rollout_and_loss(dev):
    """The whole objective of one candidate against the CPU restatement: oracle rollout (pinned by G8) -> oracle
    Sinkhorn -> oracle penalties (pinned by G10)."""
    s, params, obs = _solver(dev)
    x = np.concatenate((s.sample_traj[:, 0], s.sample_traj[:, 1])) * 1.1
    got = s.cma_objective(x)
    rot, ty = orc.interpolate_trajectory(x, s.sample_traj.shape[0], s.rx_init, s.scale_rot, s.scale_ty, s.max_rot, s.max_ty)
    rigid = obs[-1, obs[-1, :, 1] == 1][:, 2:5]
    traj = orc.rigid_body_trajectory(rot, ty, s.horizon, s.ty_init, rigid)
    final = orc.rollout(params, obs, traj, s.horizon, STATS, BOUNDS, 0.015, CART, MAT, CTRL, 2, 2)
    end = final[-1, final[-1, :, 1] == 0][:, 2:5]
    w = orc.sinkhorn_divergence(end, s.desired_pos.cpu().numpy(), blur=0.05)
    actions = np.stack((rot[:s.horizon], ty[:s.horizon]), axis=1)
    v, a, b = orc.planner_penalties(actions, s.rx_init, s.rotation_limit, [s.max_rot, s.max_ty], [s.max_rot, s.max_ty])
    ref = s.beta * w + s.penalty * b + s.alpha * v + s.gamma * a
    # the Wasserstein term alone: device loss of the DEVICE rollout's end cloud against the float64 oracle loss of the ORACLE rollout's
    # (the two rollouts agree to a few 1e-6 in position; the loss is quadratic in displacements of ~2e-3)
    print(f"objective: device {got:.9g}, oracle {ref:.9g}, relative difference {abs(got - ref) / abs(ref):.2e}")
    assert abs(got - ref) <= 3e-4 * abs(ref)


def test_optimize_trajectory_improves_the_objective(dev):
    """A short CMA-ES run from the sample trajectory (step size of the order of the trajectory increments): the best
    candidate beats the starting point, is reproducible, and the bookkeeping matches popsize x iterations."""
    s, _, _ = _solver(dev)
    s.cma_options["maxiter"], s.cma_options["popsize"], s.cma_initial_var = 12, 8, 0.003
    # a jerky starting trajectory (alternating increments): smoother candidates have a lower velocity / acceleration cost
    jerky = np.stack((180.0 - np.cumsum([0, 0.8, 0.0, 0.8, 0.0, 0.8]), 1e-4 * np.cumsum([0, 2, 0, 2, 0, 2])), axis=1)
    s.set_sample_traj(jerky)
    x0 = np.concatenate((s.sample_traj[:, 0], s.sample_traj[:, 1]))
    f0 = s.cma_objective(x0)
    xbest, es = s.optimize_trajectory(s.desired_pos)
    assert es.countiter == 12 and es.countevals == 96
    assert es.result.fbest < f0, (es.result.fbest, f0)
    assert abs(s.cma_objective(xbest) - es.result.fbest) <= 1e-5 * abs(f0)


def test_interpolated_solver_runs_constrained_optimisation(dev):
    """InterpolatedCMAsolver end to end on the device: key-point trajectory, batched objective incl. the key-point
    penalty, fmin_con with the increment constraints; feasible candidates are tracked."""
    from gnn_manip_amd import EncProcDecGNN, GraphBoundedMultimaterialControl, scene
    from gnn_manip_amd.planner import InterpolatedCMAsolver
    obs = scene.make_scene(300, seed=33, side=0.055)
    params = orc.init_params(25, 4, 3, 128, 2, 2, 33)
    params["decoder.4.weight"] = params["decoder.4.weight"] * 1e-3
    m = EncProcDecGNN(25, 4, 3, 128, 2, 2)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    m = m.to(dev)
    ga = GraphBoundedMultimaterialControl(0.015, STATS, CART, MAT, CTRL, BOUNDS)
    state = (_t(obs, dev), _t(obs[-1, :, 2:5], dev))
    horizon, npts = 8, 2
    s = InterpolatedCMAsolver(m, ga, state, 180, [0.5, 0.5, 0.4], scale_rot=np.pi, scale_ty=1.0, alpha=0.1, beta=1000.0, gamma=0.05,
                              penalty=1.0, rho=0.5, device=dev, cma_iter=4, cma_popsize=6, cma_var=1e-3, total_steps=horizon,
                              traj_points=npts, candidates_per_gpu=3)
    sample = np.stack((180.0 - 0.2 * np.arange(horizon + 1), 0.5 + 1e-4 * np.arange(horizon + 1)), axis=1)
    s.set_sample_traj(sample)
    assert s.sample_traj.shape == (4, 2)
    x0 = np.concatenate((s.sample_traj[:, 0], s.sample_traj[:, 1]))
    rot, ty = s.interpolate_trajectory(x0)
    assert len(rot) == horizon and abs(rot[0] - np.pi) < 1e-12
    coffee = obs[-1, obs[-1, :, 1] == 0][:, 2:5]
    xbest, es = s.optimize_trajectory(_t(coffee + np.float32(0.002), dev))
    assert es.countiter == 4 and es.countevals == 24 and np.isfinite(es.result.fbest)
    assert es.best_feasible.info is None or (es.best_feasible.info["g"] <= 0).all()
    # the batched objective equals the one-by-one objective here as well
    X = [x0, x0 * 1.01, x0 * 0.99]
    np.testing.assert_allclose(s.population_losses(X), [s.cma_objective(x) for x in X], rtol=2e-5)


def test_c5_two_ranks_on_one_gpu(dev, tmp_path):
    """BASELINE C5's shape at rehearsal size: a CMA-ES generation sharded over two ranks (gloo, both on this GPU), each rank
    rolling its candidates out in block-diagonal batches and taking one Sinkhorn loss per candidate -- the real
    TrajectoryCMAsolver.population_losses -- must equal the single-process evaluation of the same population."""
    import json
    import os
    import socket
    import subprocess
    import sys
    s, _, _ = _solver(dev, cands=2)
    x0 = np.concatenate((s.sample_traj[:, 0], s.sample_traj[:, 1]))
    rng = np.random.default_rng(6)
    X = [x0 + 0.05 * rng.standard_normal(x0.shape) for _ in range(7)]
    single = s.population_losses(X)
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    out = tmp_path / "losses.json"
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_c5_rehearsal_worker.py")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, worker, str(out)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        logs.append(o.decode(errors="replace"))
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    sharded = json.load(open(out))
    assert len(sharded) == 7
    np.testing.assert_allclose(sharded, single, rtol=0, atol=0)   # same kernels, same per-graph tables: identical


def test_bench_c5_two_rank_rehearsal(dev):
    """The driver's multi-GPU launch of bench.py, rehearsed on one card: `bench.py --gpus 2 --workload c5` as two fresh rank
    processes (env rendezvous on 127.0.0.1, GM_BENCH_REHEARSE=1: gloo, both ranks on cuda:0, reduced sizes).  Checks the JSON
    line rank 0 prints: rank count, contiguous candidate blocks (SURVEY 8e / traj_utils.py:247-259), finite losses, and the
    separately timed broadcast + all-gather share."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "c5", "--candidates-total", "6", "--batch", "2",
           "--steps", "4"]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", GM_BENCH_REHEARSE="1")
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=400)
        except subprocess.TimeoutExpired:
            p.kill()
            o, e = p.communicate()
        outs.append((o.decode(errors="replace"), e.decode(errors="replace")))
    assert all(p.returncode == 0 for p in procs), "\n".join(o + e for o, e in outs)
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not [l for l in outs[1][0].splitlines() if l.startswith("{")]   # ONE line, from rank 0
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "strong" and rec["unit"] == "rollout steps/s"
    cfg = rec["config"]
    assert cfg["candidates"] == 6 and cfg["candidates_per_rank"] == 3 and cfg["block_diagonal_batch"] == 2 and cfg["horizon"] == 4
    assert np.isfinite(cfg["loss_mean"]) and rec["value"] > 0
    assert rec["collective_ms"] >= 0.0 and rec["collective_ms"] < cfg["generation_s"] * 1e3


def test_bench_launches_its_own_ranks(dev):
    """`python bench.py --gpus 2 ...` as ONE command with no launcher around it (how the driver may start the scaling runs):
    the parent starts two fresh rank processes itself and relays rank 0's single JSON line.  Rehearsed on one card
    (GM_BENCH_REHEARSE=1: gloo, both ranks on cuda:0), default workload at a reduced step count and the C5 shape."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GM_BENCH_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    for extra, scaling in ((["--workload", "c5", "--candidates-total", "4", "--batch", "2", "--steps", "3"], "strong"),
                           (["--workload", "c2", "--steps", "3", "--warmup", "1"], "weak")):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"] + extra, env=env, capture_output=True, timeout=600)
        assert r.returncode == 0, r.stdout.decode(errors="replace") + r.stderr.decode(errors="replace")
        lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
        assert len(lines) == 1
        rec = json.loads(lines[0])
        assert rec["n_gpus"] == 2 and rec["scaling"] == scaling and rec["value"] > 0 and rec["unit"] == "rollout steps/s"


def test_bench_collectives_run_through_rccl_on_one_rank(dev):
    """The multi-GPU code path of bench.py on the one card the test box has: `--collectives always` initialises the `nccl` backend
    (= RCCL on ROCm) with WORLD_SIZE = 1 in a fresh child process and runs the per-generation exchange of SURVEY 8e /
    traj_utils.py:247-259 -- broadcast of the candidate matrix / scripted poses, all-gather of the results, barrier, MAX all-reduce
    of the time -- on DEVICE-resident payloads, for the C5 shape (through planner.CandidateEvaluator) and for the default
    workload.  (Two ranks on one device is what RCCL refuses; the gloo rehearsals above cover rank > 0.)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GM_BENCH_REHEARSE"):
        env.pop(k, None)
    for extra, scaling in ((["--workload", "c5", "--candidates-total", "4", "--batch", "2", "--steps", "3"], "strong"),
                           (["--workload", "c2", "--steps", "3", "--warmup", "1", "--no-extra", "--no-cpu-baseline"], "weak")):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--collectives", "always"] + extra, env=env, capture_output=True, timeout=600)
        assert r.returncode == 0, r.stdout.decode(errors="replace") + r.stderr.decode(errors="replace")
        lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
        assert len(lines) == 1
        rec = json.loads(lines[0])
        assert rec["collective_backend"] == "nccl" and rec["n_gpus"] == 1 and rec["scaling"] == scaling
        assert rec["value"] > 0 and rec["collective_ms"] >= 0.0


def test_bench_launcher_reports_a_failed_rank(dev):
    """A rank that dies must turn the whole `bench.py --gpus N` command into a non-zero exit, not a hang."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GM_BENCH_REHEARSE="1", GM_BENCH_FAIL_RANK="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "c2", "--steps", "2", "--warmup", "0"],
                       env=env, capture_output=True, timeout=300)
    assert r.returncode != 0
    assert not [l for l in r.stdout.decode().splitlines() if l.startswith("{")]


def test_c5_full_size_generation_through_rccl(dev):
    """BASELINE config C5 at its OWN size on the one card: `bench.py --workload c5 --collectives always`, unreduced -- a CMA-ES
    generation of 64 candidates x 200 rollout steps at N = 5k (hidden 128, 10 message-passing steps) in block-diagonal batches of
    8, the 64 Sinkhorn losses in one batched launch sequence, the population broadcast and the losses all-gathered through the
    `nccl` backend (RCCL; one rank) -- as a fresh child process.  What the 8-GPU run shards eight ways
    (traj_utils.py:114-159,247-259)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GM_BENCH_REHEARSE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "c5", "--collectives", "always"], env=env,
                       capture_output=True, timeout=900)
    assert r.returncode == 0, r.stdout.decode(errors="replace") + r.stderr.decode(errors="replace")
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    cfg = rec["config"]
    assert cfg["candidates"] == 64 and cfg["horizon"] == 200 and cfg["n_particles"] == 5000
    assert cfg["candidates_per_rank"] == 64 and cfg["block_diagonal_batch"] == 8
    assert rec["collective_backend"] == "nccl" and rec["n_gpus"] == 1 and rec["scaling"] == "strong"
    assert cfg["losses_finite"] and np.isfinite(cfg["loss_mean"]) and cfg["loss_mean"] > 0
    assert rec["value"] > 0 and rec["unit"] == "rollout steps/s" and rec["steps"] == 200
    assert 0.0 <= rec["collective_ms"] < cfg["generation_s"] * 1e3
    print(f"C5 64 x 200 on one GPU: {rec['value']:.0f} rollout steps/s, generation {cfg['generation_s']:.2f} s, "
          f"collectives {rec['collective_ms']:.1f} ms")


def _c5_scene(dev, m_steps=10, seed=83):
    from gnn_manip_amd import EncProcDecGNN, GraphBoundedMultimaterialControl, scene
    n = 5000
    obs = scene.make_scene(n, seed=41, vel_scale=1e-6)
    params = orc.init_params(25, 4, 3, 128, 2, m_steps, seed)
    params["decoder.4.weight"] = params["decoder.4.weight"] * 1e-3   # the pile stays dense over 200 steps
    params["decoder.4.bias"] = params["decoder.4.bias"] * 1e-3
    model = EncProcDecGNN(25, 4, 3, 128, 2, m_steps)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    model = model.to(dev)
    stats = dict(STATS, acceleration_mean=[0.0, 0.0, 0.0])
    ga = GraphBoundedMultimaterialControl(0.015, stats, CART, MAT, CTRL, BOUNDS)
    return n, obs, params, model, ga, stats


def test_c5_candidates_of_a_full_batch_equal_their_standalone_rollouts(dev):
    """C5's unit of work at its own size: 8 candidates x 200 steps at N = 5k as ONE block-diagonal batch.  Candidates 2 and 5 of
    the batch equal their standalone 200-step rollouts bit for bit (per-graph block tables: what shares a launch never changes a
    sum), and so do their Sinkhorn losses taken in the batch of 8 against the losses taken one by one."""
    from gnn_manip_amd import RolloutEngine, scene
    from gnn_manip_amd.losses import SamplesLoss
    n, obs, params, model, ga, _ = _c5_scene(dev)
    horizon, B = 200, 8
    base = scene.rigid_drift_trajectory(obs, horizon, seed=42, step_size=2e-5)
    offs = np.linspace(-1.0, 1.0, B).astype(np.float32)
    trajs = np.stack([base + np.float32(3e-5) * o * np.arange(1, horizon + 1, dtype=np.float32)[:, None, None] for o in offs])
    obs_d = _t(obs, dev)
    eng8 = RolloutEngine(model, ga, n, device=dev, candidates=B)
    with torch.no_grad():
        finals = eng8.rollout_candidates(obs_d, _t(trajs, dev), horizon)
    assert finals.shape == (B, 6, n, 8) and bool(torch.isfinite(finals).all())
    assert eng8.status() > 8 * 85000   # the 8 piles are still dense after 200 steps
    rows = torch.nonzero(obs_d[-1, :, 1] != 1).reshape(-1)
    clouds = finals[:, -1].index_select(1, rows)[:, :, 2:5].contiguous()
    target = (obs_d[-1].index_select(0, rows)[:, 2:5] + 0.01).contiguous()
    loss = SamplesLoss(loss="sinkhorn", p=2, blur=.05)
    w8 = loss.batched(clouds, target).cpu().numpy()
    assert np.isfinite(w8).all() and (w8 > 0).all() and len(set(w8.tolist())) == B   # eight different candidates
    eng1 = RolloutEngine(model, ga, n, device=dev)
    for c in (2, 5):
        with torch.no_grad():
            alone = eng1.rollout(obs_d, _t(trajs[c], dev), horizon=horizon)
        assert torch.equal(alone, finals[c]), (c, float((alone - finals[c]).abs().max()))
        w1 = loss(alone[-1].index_select(0, rows)[:, 2:5].contiguous(), target).cpu().numpy()
        assert w1.tobytes() == w8[c].tobytes()


def test_c5_short_horizon_candidate_against_the_oracle(dev):
    """One candidate of the C5 scene over a short horizon against the CPU restatement end to end: oracle.rollout (the reference's
    cma_objective loop, pinned by fixture G8) and oracle.sinkhorn_divergence (float64, dense) on its end cloud."""
    from gnn_manip_amd import RolloutEngine, scene
    from gnn_manip_amd.losses import SamplesLoss
    n, obs, params, model, ga, stats = _c5_scene(dev)
    horizon = 3
    traj = scene.rigid_drift_trajectory(obs, horizon, seed=43, step_size=2e-5)
    B = 8
    trajs = np.stack([traj + np.float32(1e-5 * c) for c in range(B)])
    eng = RolloutEngine(model, ga, n, device=dev, candidates=B)
    with torch.no_grad():
        finals = eng.rollout_candidates(_t(obs, dev), _t(trajs, dev), horizon)
    c = 3
    ref = orc.rollout(params, obs, trajs[c], horizon, stats, BOUNDS, 0.015, CART, MAT, CTRL, 2, 10)
    got = finals[c].cpu().numpy()
    d = np.abs(got[:, :, 2:8] - ref[:, :, 2:8])
    assert d.max() <= 5e-5 and (d > 5e-6).mean() <= 2e-4, (d.max(), (d > 5e-6).sum())
    coffee = obs[-1, :, 1] != 1
    target = obs[-1, coffee, 2:5] + np.float32(0.01)
    rows = torch.nonzero(_t(obs, dev)[-1, :, 1] != 1).reshape(-1)
    w = SamplesLoss(loss="sinkhorn", p=2, blur=.05).batched(finals[:, -1].index_select(1, rows)[:, :, 2:5].contiguous(), _t(target, dev))
    w_ref = orc.sinkhorn_divergence(ref[-1, coffee, 2:5], target, blur=0.05)
    assert abs(float(w[c]) - w_ref) <= 1e-4 * abs(w_ref), (float(w[c]), w_ref)


def test_bench_launcher_ends_a_stuck_rank(dev):
    """A rank that never arrives (stuck before the rendezvous -- what a wedged RCCL init looks like from outside) must not hang
    `bench.py --gpus N` until somebody else's timeout: after --rank-timeout the launcher terminates the ranks it started, exits
    non-zero, prints no JSON line, and says which rank was stuck together with the tails of the per-rank output files."""
    import os
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GM_BENCH_REHEARSE="1", GM_BENCH_STUCK_RANK="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "c2", "--steps", "2", "--warmup", "0",
                        "--rank-timeout", "45"], env=env, capture_output=True, timeout=300)
    took = time.monotonic() - t0
    err = r.stderr.decode(errors="replace")
    assert r.returncode == 124, (r.returncode, err)
    assert took < 120, took
    assert not [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert "rank(s) [0, 1] still running after --rank-timeout 45 s" in err, err   # rank 0 waits in the rendezvous, rank 1 never came
    assert "GM_BENCH_STUCK_RANK set" in err   # the stuck rank's own stderr tail is part of the message
