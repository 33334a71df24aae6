"""TEST INFRASTRUCTURE -- plain-PyTorch (CPU, differentiable) restatement of the encode-process-decode
model, used only by tests/ and bench.py's cpu_baseline as the gradient oracle of the HIP backward.

Follows gnn_manip/models/epd_gnn.py:72-105 (MLP structure, LayerNorm / residual placement) with the
block semantics of DESIGN.md section 2 (e' = phi_e(cat[h_i, h_j, e]), agg_i = sum e', h' =
phi_v(cat[h, agg]); j = edge_index[0], i = edge_index[1]).  Parameters are a state_dict-style mapping
(keys ``encoder.phi_edge.0.weight`` ...), the same one oracle/epd_oracle.py:init_params produces.
The forward of this file is itself checked against the numpy oracle (tests/test_oracle_golden.py).
"""
import torch
import torch.nn.functional as F


def mlp(p, prefix, x, num_layers, norm):
    """Linear ReLU [Linear ReLU]x(L-1) Linear [LayerNorm]  (epd_gnn.py:72-84)."""
    for l in range(num_layers):
        x = F.relu(F.linear(x, p[f"{prefix}.{2 * l}.weight"], p[f"{prefix}.{2 * l}.bias"]))
    k = 2 * num_layers
    x = F.linear(x, p[f"{prefix}.{k}.weight"], p[f"{prefix}.{k}.bias"])
    if norm:
        x = F.layer_norm(x, (x.shape[1],), p[f"{prefix}.{k + 1}.weight"], p[f"{prefix}.{k + 1}.bias"], 1e-5)
    return x


def epd_forward(p, nodes, edge_attr, edge_index, num_layers, m_steps):
    """epd_gnn.py:86-105."""
    j, i = edge_index[0], edge_index[1]
    h = mlp(p, "encoder.phi_node", nodes, num_layers, True)
    e = mlp(p, "encoder.phi_edge", edge_attr, num_layers, True)
    for k in range(m_steps):
        e_new = mlp(p, f"processor.{k}.phi_edge", torch.cat((h[i], h[j], e), dim=1), num_layers, True)
        agg = torch.zeros_like(h).index_add_(0, i, e_new)
        h_new = mlp(p, f"processor.{k}.phi_node", torch.cat((h, agg), dim=1), num_layers, True)
        h, e = h + h_new, e + e_new
    return mlp(p, "decoder", h, num_layers, False)


def loss_and_grads(params_np, nodes, edge_attr, edge_index, target, num_layers, m_steps, dtype=torch.float64):
    """L1 loss of train_dyn.py:65 (sum / N) and its gradient w.r.t. every parameter, on the CPU."""
    p = {k: torch.tensor(v, dtype=dtype, requires_grad=True) for k, v in params_np.items()}
    out = epd_forward(p, torch.tensor(nodes, dtype=dtype), torch.tensor(edge_attr, dtype=dtype),
                      torch.tensor(edge_index, dtype=torch.int64), num_layers, m_steps)
    loss = F.l1_loss(out, torch.tensor(target, dtype=dtype), reduction="sum") / out.shape[0]
    loss.backward()
    return out.detach().numpy(), float(loss.detach()), {k: v.grad.numpy() for k, v in p.items()}


def _mlp_taped(p, prefix, x, num_layers, norm, tape):
    """mlp() that records every hidden pre-activation z and activation a = relu(z) (both keep their gradients)."""
    for l in range(num_layers):
        z = F.linear(x, p[f"{prefix}.{2 * l}.weight"], p[f"{prefix}.{2 * l}.bias"])
        x = F.relu(z)
        x.retain_grad()
        tape.append((z, x))
    k = 2 * num_layers
    x = F.linear(x, p[f"{prefix}.{k}.weight"], p[f"{prefix}.{k}.bias"])
    if norm:
        x = F.layer_norm(x, (x.shape[1],), p[f"{prefix}.{k + 1}.weight"], p[f"{prefix}.{k + 1}.bias"], 1e-5)
    return x


def relu_flip_allowance(params_np, nodes, edge_attr, edge_index, target, num_layers, m_steps, tau=1e-5, max_units=256):
    """The gradient of a ReLU network is discontinuous where a pre-activation crosses zero: two evaluations that are both accurate to
    float32 rounding can disagree on the sign of a pre-activation that lies within rounding distance of zero, and then their
    gradients differ by what toggling that unit's ReLU derivative changes.  This returns, per parameter, an upper bound of that
    effect for THIS input (float64, first order): for every hidden unit with |z| < tau * rms(z of its Linear) -- at most max_units,
    the closest to zero first -- the gradient that flows back from toggling it alone, absolute values summed over the units.
    Returns ({name: max |allowed change|}, number of such units)."""
    dtype = torch.float64
    p = {k: torch.tensor(v, dtype=dtype, requires_grad=True) for k, v in params_np.items()}
    tape = []
    idx = torch.tensor(edge_index, dtype=torch.int64)
    j, i = idx[0], idx[1]
    h = _mlp_taped(p, "encoder.phi_node", torch.tensor(nodes, dtype=dtype), num_layers, True, tape)
    e = _mlp_taped(p, "encoder.phi_edge", torch.tensor(edge_attr, dtype=dtype), num_layers, True, tape)
    for k in range(m_steps):
        e_new = _mlp_taped(p, f"processor.{k}.phi_edge", torch.cat((h[i], h[j], e), dim=1), num_layers, True, tape)
        agg = torch.zeros_like(h).index_add_(0, i, e_new)
        h_new = _mlp_taped(p, f"processor.{k}.phi_node", torch.cat((h, agg), dim=1), num_layers, True, tape)
        h, e = h + h_new, e + e_new
    out = _mlp_taped(p, "decoder", h, num_layers, False, tape)
    loss = F.l1_loss(out, torch.tensor(target, dtype=dtype), reduction="sum") / out.shape[0]
    loss.backward(retain_graph=True)
    units = []   # (|z| / rms, tape entry, flat index)
    for t, (z, a) in enumerate(tape):
        if z.numel() == 0:
            continue
        zz = z.detach()
        rms = float(zz.pow(2).mean().sqrt())
        if rms <= 0.0:
            continue
        near = torch.nonzero(zz.abs().flatten() < tau * rms).flatten()
        for q in near.tolist():
            units.append((float(zz.flatten()[q].abs()) / rms, t, q))
    units.sort()
    units = units[:max_units]
    names = list(p.keys())
    allow = {k: torch.zeros_like(v) for k, v in p.items()}
    for _, t, q in units:
        z, a = tape[t]
        ga = a.grad.flatten()[q]   # d loss / d a_u: what reaches the unit from above whether or not its derivative lets it through
        if float(ga) == 0.0:
            continue
        seed = torch.zeros_like(z).flatten()
        seed[q] = ga
        g = torch.autograd.grad(z, [p[k] for k in names], grad_outputs=seed.view_as(z), retain_graph=True, allow_unused=True)
        for k, gk in zip(names, g):
            if gk is not None:
                allow[k] += gk.abs()
    return {k: float(v.max()) for k, v in allow.items()}, len(units)
