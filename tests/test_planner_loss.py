"""Planner loss + optimiser pieces on the CPU: oracle restatements against the reference
fixture G10 and against known answers; the own CMA-ES on standard test functions."""
import numpy as np
import pytest

from oracle import epd_oracle as orc


def test_penalty_terms_golden(golden):
    """compute_vel_acc / compute_vel_loss / compute_acc_loss / compute_boundaries_penalty and the weighting of
    compute_loss (traj_utils.py:161-165,230-285), Wasserstein term stubbed to 0.125 when the fixture was made."""
    g = golden("g10_planner_loss.npz")
    alpha, beta, gamma, penalty, rho, max_rot, max_ty, rx_init, rot_limit = g["cfg"]
    for tag in ("inside", "outside"):
        ref = g[f"{tag}.out"]
        v, a, b = orc.planner_penalties(g[f"{tag}.actions"], rx_init, rot_limit, [max_rot, max_ty], [max_rot, max_ty])
        np.testing.assert_allclose([v, a, b], ref[2:5], rtol=1e-12)
        np.testing.assert_allclose(beta * 0.125 + penalty * b + alpha * v + gamma * a, ref[0], rtol=1e-12)
    assert g["outside.out"][4] == 20.0 and g["inside.out"][4] == 0.0


def test_sinkhorn_known_answers():
    """Properties any correct debiased Sinkhorn divergence with cost |x-y|^2/2 has (the restatement is unpinned
    against geomloss itself, which is absent): S(a, a) = 0; symmetry; a rigid translation t of the same cloud
    costs exactly |t|^2 / 2 (the translation splits off the quadratic cost for every epsilon); >= 0."""
    rng = np.random.default_rng(3)
    x = 0.5 + 0.05 * rng.standard_normal((150, 3))
    y = 0.55 + 0.08 * rng.standard_normal((130, 3))
    assert abs(orc.sinkhorn_divergence(x, x)) < 1e-12
    sxy, syx = orc.sinkhorn_divergence(x, y), orc.sinkhorn_divergence(y, x)
    assert sxy > 0 and abs(sxy - syx) <= 1e-9 * sxy
    t = np.array([0.03, -0.02, 0.05])
    np.testing.assert_allclose(orc.sinkhorn_divergence(x, x + t), 0.5 * (t ** 2).sum(), rtol=2e-3)
    # blur -> 0 on two tiny clouds: the divergence approaches the exact optimal-transport cost (brute force)
    a = np.array([[0.0, 0, 0], [1.0, 0, 0], [0, 1.0, 0]])
    b = np.array([[0.1, 0, 0], [1.0, 0.2, 0], [0, 1.0, 0.3]])
    import itertools
    best = min(sum(0.5 * ((a[i] - b[p[i]]) ** 2).sum() for i in range(3)) / 3 for p in itertools.permutations(range(3)))
    np.testing.assert_allclose(orc.sinkhorn_divergence(a, b, blur=0.01), best, rtol=1e-3)


def test_sinkhorn_float32_restatement_agrees_with_float64():
    rng = np.random.default_rng(4)
    x = (0.5 + 0.05 * rng.standard_normal((200, 3))).astype(np.float32)
    y = (0.52 + 0.06 * rng.standard_normal((180, 3))).astype(np.float32)
    s64, s32 = orc.sinkhorn_divergence(x, y), orc.sinkhorn_divergence(x, y, dtype=np.float32)
    assert abs(s64 - s32) <= 2e-4 * abs(s64)


def test_cmaes_minimises_standard_functions():
    from gnn_manip_amd import cmaes
    x, es = cmaes.fmin2(lambda v: float(np.sum(np.asarray(v) ** 2)), [1.0] * 8, 0.5, {"seed": 3, "maxiter": 300})
    assert es.result.fbest < 1e-10 and np.abs(x).max() < 1e-4
    rosen = lambda v: float(sum(100 * (v[i + 1] - v[i] ** 2) ** 2 + (1 - v[i]) ** 2 for i in range(len(v) - 1)))
    x, es = cmaes.fmin2(rosen, [0.0] * 6, 0.5, {"seed": 4, "maxiter": 1500})
    np.testing.assert_allclose(x, 1.0, atol=1e-4)
    # reproducible for a seed; population interface; bounds respected
    a = cmaes.CMAEvolutionStrategy([0.5] * 4, 0.3, {"seed": 9, "popsize": 12, "bounds": [-0.6, 0.6]})
    b = cmaes.CMAEvolutionStrategy([0.5] * 4, 0.3, {"seed": 9, "popsize": 12, "bounds": [-0.6, 0.6]})
    Xa, Xb = a.ask(), b.ask()
    assert len(Xa) == 12 and all(np.array_equal(p, q) for p, q in zip(Xa, Xb))
    assert max(np.abs(p).max() for p in Xa) <= 0.6
    calls = []
    x, es = cmaes.fmin2(None, [1.0] * 5, 0.4, {"seed": 1, "maxiter": 40, "popsize": 10},
                        parallel_objective=lambda X: calls.append(len(X)) or [float(np.sum(np.asarray(v) ** 2)) for v in X])
    assert calls == [10] * 40 and es.result.fbest < 1e-2


def test_interpolated_solver_pieces_golden(golden):
    """InterpolatedCMAsolver (traj_utils.py:288-452): key points from the sample, PCHIP interpolation, its loss terms
    (fixed normalisation means, exp penalty of the key-point increments) and inequality constraints -- oracle and the
    package's host-side mirror against the reference's outputs."""
    import torch
    from gnn_manip_amd.planner import InterpolatedCMAsolver
    g = golden("g10_planner_loss.npz")
    alpha, beta, gamma, penalty, rho, max_rot, max_ty, rx_init, rot_limit, scale_rot, scale_ty, hz, npts = g["interp.cfg"]
    hz, npts = int(hz), int(npts)
    st = orc.interp_set_sample_traj(g["interp.sample_in"], npts, rx_init, 0.5, scale_rot, scale_ty)
    np.testing.assert_allclose(st, g["interp.sample_traj"], rtol=1e-13, atol=1e-15)
    x = g["interp.x"]
    rot, ty = orc.interp_interpolate_trajectory(x, st.shape[0], hz, npts, rx_init, scale_rot, scale_ty)
    np.testing.assert_allclose(rot, g["interp.rot"], rtol=1e-13)
    np.testing.assert_allclose(ty, g["interp.ty"], rtol=1e-12, atol=1e-16)
    np.testing.assert_allclose(orc.interp_ineq_constraint(x, hz // npts, npts, scale_rot, scale_ty, max_rot, max_ty), g["interp.ineq"],
                               rtol=1e-12, atol=1e-15)
    ref = g["interp.out"]  # (loss, wasserstein, vel_loss, acc_loss, interp_loss, 0)
    np.testing.assert_allclose(orc.interp_vel_noninterp(x, hz // npts, npts, scale_rot, scale_ty, max_rot, max_ty), ref[4], rtol=1e-12)
    # host-side mirror (no GPU needed for these members)
    obs = torch.zeros((6, 10, 8))
    obs[-1, 5:, 1] = 1.0
    s = InterpolatedCMAsolver(None, type("GA", (), {"material_idx": [1], "cartesian_idx": [2, 3, 4]})(), (obs, obs[-1, :, 2:5]), 180,
                              [0.5, 0.5, 0.4], scale_rot=np.pi, scale_ty=1.0, alpha=alpha, beta=beta, gamma=gamma, penalty=penalty,
                              rho=rho, device="cpu", total_steps=hz, traj_points=npts)
    s.set_sample_traj(g["interp.sample_in"])
    np.testing.assert_allclose(s.sample_traj, g["interp.sample_traj"], rtol=1e-13, atol=1e-15)
    r2, t2 = s.interpolate_trajectory(x)
    np.testing.assert_allclose(r2, g["interp.rot"], rtol=1e-13)
    np.testing.assert_allclose(s.ineq_constraint(x), g["interp.ineq"], rtol=1e-12, atol=1e-15)
    s.loss = lambda a, b: torch.tensor(0.125)
    s.desired_pos = obs[-1, :, 2:5]
    out = s.compute_loss(obs[-1, :, 2:5], np.stack((g["interp.rot"], g["interp.ty"]), axis=1), x=x)
    np.testing.assert_allclose(out[:5], ref[:5], rtol=1e-10)


def test_fmin_con_respects_constraints():
    from gnn_manip_amd import cmaes
    # minimise |x|^2 subject to x0 >= 1 (g = 1 - x0 <= 0): optimum (1, 0, 0)
    x, es = cmaes.fmin_con(lambda v: float(np.sum(np.asarray(v) ** 2)), [2.0, 1.0, -1.0], 0.5, g=lambda v: [1.0 - v[0]],
                           options={"seed": 5, "maxiter": 250})
    assert es.best_feasible.info is not None and (es.best_feasible.info["g"] <= 0).all()
    np.testing.assert_allclose(es.best_feasible.info["x"], [1.0, 0.0, 0.0], atol=2e-2)
    assert abs(es.best_feasible.f - 1.0) < 3e-2
