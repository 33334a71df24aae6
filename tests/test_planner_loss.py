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
