"""Seeded synthetic granular scenes shared by bench.py and the tests.

Pure numpy (PCG64), so a seed gives the same scene on every machine.  State layout follows the
reference's test dataset with control (coffee_dataset.py:170-181): [k, N, 8] float32 rows
``[id, material, x, y, z, cx, cy, cz]``; the last ``rigid_frac`` of the rows are rigid (material 1).
"""
import numpy as np

STATS = dict(velocity_mean=[1.5e-4, -2.5e-4, 0.5e-4], velocity_std=[2.1e-3, 3.2e-3, 1.9e-3],
             acceleration_mean=[1.0e-6, -8.0e-6, 2.0e-6], acceleration_std=[2.4e-4, 3.1e-4, 2.2e-4])
BOUNDS = dict(lower_bounds=[0.1, 0.1, 0.1], upper_bounds=[0.9, 0.9, 0.9])
CART, MAT, CTRL = [2, 3, 4], [1], [5, 6, 7]
CONN_R = 0.015


def dense_side(n, conn_r=CONN_R, mean_in_radius=28.0):
    """Box side giving about `mean_in_radius` particles inside a conn_r ball (cap of 20 binds almost everywhere)."""
    vol_ball = 4.0 / 3.0 * np.pi * conn_r ** 3
    return float((n * vol_ball / mean_in_radius) ** (1.0 / 3.0))


def make_scene(n, seed=0, k=6, rigid_frac=0.1, side=None, lo=0.3, vel_scale=5e-4):
    rng = np.random.Generator(np.random.PCG64(seed))
    side = dense_side(n) if side is None else side
    lo = min(lo, 0.85 - side)
    p0 = lo + side * rng.random((n, 3))
    v = vel_scale * rng.standard_normal((n, 3))
    obs = np.zeros((k, n, 8), dtype=np.float32)
    for t in range(k):
        obs[t, :, 2:5] = (p0 + t * v + 0.02 * vel_scale * rng.standard_normal((n, 3))).astype(np.float32)
    obs[:, :, 0] = np.arange(n, dtype=np.float32)
    n_rigid = int(round(n * rigid_frac))
    if n_rigid:
        obs[:, n - n_rigid:, 1] = 1.0
    return obs


def rigid_drift_trajectory(obs, steps, seed=1, step_size=2e-4):
    """[steps, N_rigid, 3] scripted poses: the rigid rows translate rigidly along a fixed direction."""
    rng = np.random.Generator(np.random.PCG64(seed))
    rigid = obs[-1, :, 1] == 1
    base = obs[-1, rigid, 2:5].astype(np.float32)
    d = rng.standard_normal(3)
    d = (d / np.linalg.norm(d) * step_size).astype(np.float32)
    return np.stack([base + (i + 1) * d for i in range(steps)]).astype(np.float32)
