"""(mu/mu_w, lambda)-CMA-ES with the small part of the ``cma`` package's interface that the reference's planner
uses (gnn_manip/utils/traj_utils.py:71-76,257: ``CMAOptions`` keys 'seed' / 'maxiter' / 'popsize' / 'bounds',
``cma.fmin2(objective, x0, sigma0, options)`` -> ``(xbest, es)``, ``es.ask()`` / ``es.tell()`` / ``es.stop()``).

``cma`` is an un-vendored pip dependency of the reference (environment.yml:9, version not pinned) and absent here:
this is the textbook algorithm (N. Hansen, "The CMA Evolution Strategy: A Tutorial", 2016: default strategy
parameters, rank-one + rank-mu covariance update, cumulative step-size adaptation), NOT a bit-for-bit clone of the
package's sampling sequence or termination heuristics.  Host-side numpy: the search dimension is 2 x (trajectory
points) <= 598 and a generation costs one eigendecomposition against popsize 200-step rollouts on the GPUs.

``fmin2`` additionally takes ``parallel_objective(list_of_candidates) -> list_of_fitnesses``: the whole population
of a generation is evaluated in one call (``planner.TrajectoryCMAsolver`` batches it on the device and shards it
over the ranks of a node).
"""
import numpy as np


class CMAOptions(dict):
    """Option dictionary with the package's defaults for the keys the reference touches."""

    def __init__(self, *a, **kw):
        super().__init__(seed=None, maxiter=None, popsize=None, bounds=None, tolfun=1e-11, tolx=1e-11, verbose=0)
        self.update(*a, **kw)


class _Result:
    def __init__(self, es):
        self.xbest, self.fbest, self.evals_best = es.best_x, es.best_f, es.best_evals
        self.evaluations, self.iterations = es.countevals, es.countiter
        self.xfavorite, self.stds = es.mean.copy(), es.sigma * np.sqrt(np.diag(es.C))


class CMAEvolutionStrategy:
    def __init__(self, x0, sigma0, inopts=None):
        opts = CMAOptions(inopts or {})
        self.opts = opts
        self.N = n = len(x0)
        self.mean = np.asarray(x0, dtype=np.float64).copy()
        self.sigma = float(sigma0)
        self.rng = np.random.Generator(np.random.PCG64(opts["seed"]))
        lam = opts["popsize"] or 4 + int(3 * np.log(n))
        self.popsize = lam
        self.maxiter = opts["maxiter"] or 100 + 150 * (n + 3) ** 2 // lam ** 0.5
        b = opts["bounds"]
        self.lower, self.upper = (None, None) if b is None else (b[0], b[1])
        # strategy parameters (tutorial, table 1)
        mu = lam // 2
        w = np.log(mu + 0.5) - np.log(np.arange(1, mu + 1))
        self.weights = w / w.sum()
        self.mu = mu
        self.mueff = 1.0 / (self.weights ** 2).sum()
        self.cc = (4 + self.mueff / n) / (n + 4 + 2 * self.mueff / n)
        self.cs = (self.mueff + 2) / (n + self.mueff + 5)
        self.c1 = 2 / ((n + 1.3) ** 2 + self.mueff)
        self.cmu = min(1 - self.c1, 2 * (self.mueff - 2 + 1 / self.mueff) / ((n + 2) ** 2 + self.mueff))
        self.damps = 1 + 2 * max(0.0, np.sqrt((self.mueff - 1) / (n + 1)) - 1) + self.cs
        self.chiN = n ** 0.5 * (1 - 1 / (4 * n) + 1 / (21 * n * n))
        self.pc, self.ps = np.zeros(n), np.zeros(n)
        self.C = np.eye(n)
        self.B, self.D = np.eye(n), np.ones(n)
        self.invsqrtC = np.eye(n)
        self.eigeneval = 0
        self.countevals = self.countiter = 0
        self.best_x, self.best_f, self.best_evals = self.mean.copy(), np.inf, 0
        self._last_f = None
        self._fit_hist = []

    def _update_eigensystem(self):
        if self.countevals - self.eigeneval > self.popsize / (self.c1 + self.cmu) / self.N / 10:
            self.eigeneval = self.countevals
            self.C = np.triu(self.C) + np.triu(self.C, 1).T
            d, self.B = np.linalg.eigh(self.C)
            self.D = np.sqrt(np.maximum(d, 1e-300))
            self.invsqrtC = self.B @ np.diag(1 / self.D) @ self.B.T

    def _repair(self, x):
        if self.lower is None and self.upper is None:
            return x
        return np.clip(x, self.lower, self.upper)

    def ask(self):
        """popsize candidates ~ mean + sigma * N(0, C) (clipped into 'bounds' when given)."""
        self._update_eigensystem()
        z = self.rng.standard_normal((self.popsize, self.N))
        self._y = (z * self.D) @ self.B.T
        return [self._repair(self.mean + self.sigma * y) for y in self._y]

    def tell(self, solutions, fitnesses):
        f = np.asarray(fitnesses, dtype=np.float64)
        X = np.asarray(solutions, dtype=np.float64)
        self.countevals += len(f)
        self.countiter += 1
        order = np.argsort(f, kind="stable")
        if f[order[0]] < self.best_f:
            self.best_f, self.best_x, self.best_evals = float(f[order[0]]), X[order[0]].copy(), self.countevals
        n = self.N
        old = self.mean
        ysel = (X[order[:self.mu]] - old) / self.sigma
        yw = self.weights @ ysel
        self.mean = old + self.sigma * yw
        self.ps = (1 - self.cs) * self.ps + np.sqrt(self.cs * (2 - self.cs) * self.mueff) * (self.invsqrtC @ yw)
        hsig = (np.linalg.norm(self.ps) / np.sqrt(1 - (1 - self.cs) ** (2 * self.countiter)) / self.chiN) < 1.4 + 2 / (n + 1)
        self.pc = (1 - self.cc) * self.pc + hsig * np.sqrt(self.cc * (2 - self.cc) * self.mueff) * yw
        rank_mu = (ysel * self.weights[:, None]).T @ ysel
        self.C = ((1 - self.c1 - self.cmu) * self.C + self.c1 * (np.outer(self.pc, self.pc) + (1 - hsig) * self.cc * (2 - self.cc) * self.C)
                  + self.cmu * rank_mu)
        self.sigma *= np.exp((self.cs / self.damps) * (np.linalg.norm(self.ps) / self.chiN - 1))
        self._fit_hist.append(float(f[order[0]]))
        self._last_f = f[order]

    def stop(self):
        """Non-empty dict of the conditions that hold (empty = keep going), like the package."""
        out = {}
        if self.countiter >= self.maxiter:
            out["maxiter"] = self.maxiter
        if self._last_f is not None:
            h = self._fit_hist[-(10 + int(30 * self.N / self.popsize)):]
            if len(h) > 10 and max(h) - min(h) < self.opts["tolfun"] and self._last_f[-1] - self._last_f[0] < self.opts["tolfun"]:
                out["tolfun"] = self.opts["tolfun"]
            if self.sigma * np.sqrt(np.diag(self.C)).max() < self.opts["tolx"]:
                out["tolx"] = self.opts["tolx"]
        return out

    @property
    def result(self):
        return _Result(self)


def fmin2(objective_function, x0, sigma0, options=None, parallel_objective=None):
    """``cma.fmin2``: minimise; returns (xbest, es).  parallel_objective, when given, receives the whole
    population of a generation at once."""
    es = CMAEvolutionStrategy(x0, sigma0, options)
    while not es.stop():
        X = es.ask()
        F = parallel_objective(X) if parallel_objective is not None else [objective_function(x) for x in X]
        es.tell(X, F)
    return es.result.xbest, es


class _BestFeasible:
    def __init__(self):
        self.f, self.info = np.inf, None


def fmin_con(objective_function, x0, sigma0, g=lambda x: [], options=None, parallel_objective=None):
    """``cma.fmin_con``'s role (traj_utils.py:336): minimise f subject to g(x) <= 0 (vector valued).  Augmented
    Lagrangian on top of the same CMA-ES: L = f + sum_i (lam_i g_i + mu/2 g_i^2 where the constraint is active,
    -lam_i^2 / (2 mu) elsewhere), multipliers updated at the distribution mean once per generation.  Returns
    (xbest of L, es) with ``es.best_feasible.f / .info`` (= {'x', 'f', 'g'}) tracking the best sampled point that
    satisfies every constraint, as optimise_traj.py:196-200 reads it.  Like fmin2, not a clone of the package."""
    es = CMAEvolutionStrategy(x0, sigma0, options)
    es.best_feasible = _BestFeasible()
    lam, mu = None, 1.0
    while not es.stop():
        X = es.ask()
        F = np.asarray(parallel_objective(X) if parallel_objective is not None else [objective_function(x) for x in X], dtype=np.float64)
        G = np.asarray([np.asarray(g(x), dtype=np.float64).reshape(-1) for x in X])
        if lam is None:
            lam = np.zeros(G.shape[1])
            spread = np.ptp(F) if np.ptp(F) > 0 else 1.0
            gs = np.abs(G).mean() if G.size and np.abs(G).mean() > 0 else 1.0
            mu = spread / gs ** 2  # penalty on the scale of the objective's spread per unit of squared violation
        L = F.copy()
        if G.shape[1]:
            active = G > -lam[None, :] / mu
            L += np.where(active, lam[None, :] * G + 0.5 * mu * G * G, -lam[None, :] ** 2 / (2 * mu)).sum(axis=1)
            for x, f, gv in zip(X, F, G):
                if (gv <= 0).all() and f < es.best_feasible.f:
                    es.best_feasible.f, es.best_feasible.info = float(f), {"x": np.array(x), "f": float(f), "g": gv.copy()}
        es.tell(X, L.tolist())
        if G.shape[1]:
            lam = np.maximum(0.0, lam + mu * np.asarray(g(es.mean), dtype=np.float64).reshape(-1))
    return es.result.xbest, es
