"""Training path on the MI355X: forward with tape + HIP backward (csrc/train.hip) against the plain-PyTorch
float64 CPU restatement oracle/torch_epd.py (reference call site examples/train_dyn.py:45-72).

Tolerances (floating point, stated here): forward 1e-5 relative (north_star).  Gradients: max |g - g64| <=
max(2e-4, 4 x the error of the SAME plain-PyTorch model run in float32) x max |g64| per tensor, g64 = float64
oracle.  The second term exists because the gradient of a ReLU network with an L1 loss is discontinuous:
one pre-activation whose sign differs between float32 and float64 arithmetic moves a whole row's contribution
(~1e-3 of a tensor at these sizes) -- plain PyTorch float32 shows exactly the same deviations on the same tensors."""
import numpy as np
import pytest
import torch

from conftest import BOUNDS, CART, CTRL, MAT, STATS
from oracle import epd_oracle as orc
from oracle import torch_epd

pytestmark = pytest.mark.gpu

KW = dict(stats=STATS, bounds=BOUNDS, conn_r=0.015, cartesian_idx=CART, material_idx=MAT)
GRAD_TOL = 2e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _model(params, dims, dev):
    from gnn_manip_amd import EncProcDecGNN
    m = EncProcDecGNN(*dims)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    return m.to(dev)


def _graph(n, side, seed):
    from gnn_manip_amd import scene
    obs = scene.make_scene(n, seed=seed, side=side)
    nodes, ea, s, r, _ = orc.process(obs, None, control_idx=CTRL, **KW)
    return nodes, ea, np.stack((s, r))


def _check(m, params, nodes, ea, ei, dims, dev, seed):
    rng = np.random.default_rng(seed)
    target = rng.standard_normal((nodes.shape[0], dims[2])).astype(np.float32)
    out = m.forward(_t(nodes, dev), _t(ea, dev), _t(ei, dev))
    loss = torch.nn.functional.l1_loss(out, _t(target, dev), reduction="sum") / out.shape[0]  # train_dyn.py:65
    loss.backward()
    ref_out, ref_loss, ref_g = torch_epd.loss_and_grads(params, nodes, ea, ei, target, dims[4], dims[5])
    _, _, g32 = torch_epd.loss_and_grads(params, nodes, ea, ei, target, dims[4], dims[5], torch.float32)
    assert np.abs(out.detach().cpu().numpy() - ref_out).max() <= 1e-5 * max(np.abs(ref_out).max(), 1e-3)
    assert abs(float(loss.detach()) - ref_loss) <= 1e-5 * abs(ref_loss)
    _compare_gradients(m, params, nodes, ea, ei, target, dims[4], dims[5], ref_g, g32)


def _compare_gradients(m, params, nodes, ea, ei, target, num_layers, m_steps, ref_g, g32):
    """Every parameter gradient against float64: within max(GRAD_TOL, 4 x PyTorch float32's own error on this tensor) of the
    tensor's maximum.  A tensor beyond that is allowed only what the ReLU units of THIS input whose float64 pre-activation lies
    within 1e-5 of its Linear's rms of zero can explain (oracle/torch_epd.py: relu_flip_allowance -- two float32-accurate
    evaluations may disagree on the sign of exactly those units): the bound is computed, not assumed, and no seed is exempt."""
    def worst_of(allow):
        worst = ("", 0.0)
        for name, p in m.named_parameters():
            assert p.grad is not None, name
            g, r = p.grad.cpu().numpy(), ref_g[name]
            assert g.shape == r.shape, name
            scale = max(np.abs(r).max(), 1e-12)
            err = np.abs(g - r).max() / scale
            tol = max(GRAD_TOL, 4.0 * np.abs(g32[name] - r).max() / scale) + (allow[name] / scale if allow else 0.0)
            if err / tol > worst[1]:
                worst = (name, err / tol, err, tol)
        return worst
    worst = worst_of(None)
    if worst[1] > 1.0:
        allow, n_units = torch_epd.relu_flip_allowance(params, nodes, ea, ei, target, num_layers, m_steps)
        worst2 = worst_of(allow)
        assert worst2[1] <= 1.0, (worst, worst2, n_units)


@pytest.mark.parametrize("n,side,seed,m_steps", [(900, 0.075, 91, 3), (130, 0.3, 92, 2), (2500, 0.1, 93, 10)])
def test_backward_matches_torch_autograd(dev, n, side, seed, m_steps):
    """All 22 + 8*m parameter gradients of the L1 training loss; dense, sparse (isolated nodes) and full-depth graphs."""
    dims = (25, 4, 3, 128, 2, m_steps)
    params = orc.init_params(*dims, seed)
    m = _model(params, dims, dev)
    nodes, ea, ei = _graph(n, side, seed)
    _check(m, params, nodes, ea, ei, dims, dev, seed)


def test_backward_hidden_256(dev):
    dims = (25, 4, 3, 256, 2, 2)
    params = orc.init_params(*dims, 94)
    m = _model(params, dims, dev)
    nodes, ea, ei = _graph(400, 0.07, 94)
    _check(m, params, nodes, ea, ei, dims, dev, 94)


@pytest.mark.parametrize("hidden,num_layers,m_steps,seed", [(64, 2, 3, 120), (64, 3, 2, 121), (128, 3, 2, 122), (128, 4, 1, 126), (256, 3, 1, 128),
                                                             (128, 4, 2, 129), (64, 5, 2, 130), (128, 6, 1, 131)])
def test_backward_other_widths_and_depths(dev, hidden, num_layers, m_steps, seed):
    """build_mlp takes any num_layers >= 2 (epd_gnn.py:72-84); the training kernels loop over the hidden Linears at run time
    and are instantiated for hidden 64 / 128 / 256.  Depth 4 with two steps and depths 5 / 6 put more weight-gradient jobs between
    the model's flush points than one batch holds (kWgJobsMax): the batch flushes itself when full."""
    dims = (25, 4, 3, hidden, num_layers, m_steps)
    params = orc.init_params(*dims, seed)
    m = _model(params, dims, dev)
    nodes, ea, ei = _graph(700, 0.07, seed)
    _check(m, params, nodes, ea, ei, dims, dev, seed)


@pytest.mark.parametrize("hidden,num_layers,m_steps,seed", [(100, 2, 2, 140), (40, 3, 2, 141), (200, 2, 1, 142)])
def test_backward_at_hidden_sizes_between_the_kernel_widths(dev, hidden, num_layers, m_steps, seed):
    """train_dyn.py:237-238 takes any hidden size.  Sizes between the training kernels' widths (64 / 128 / 256) train zero-padded
    at the next width: the padded parameters are differentiable functions of the real ones (EncProcDecGNN._padded_training: zero
    padding; the Linear in front of every LayerNorm centred over its outputs; eps and gamma rescaled so that the LayerNorm over the
    padded width is the one over the features that exist), so every parameter gradient comes back in the caller's shapes -- held to
    the same float64 yardstick as the native widths, forward 1e-5."""
    dims = (25, 4, 3, hidden, num_layers, m_steps)
    params = orc.init_params(*dims, seed)
    m = _model(params, dims, dev)
    nodes, ea, ei = _graph(600, 0.07, seed)
    _check(m, params, nodes, ea, ei, dims, dev, seed)
    # the fused inference path of the same module (native zero-padded kernels) agrees with the training forward
    with torch.no_grad():
        out_inf = m.forward(_t(nodes, dev), _t(ea, dev), _t(ei, dev))
    out_train = m.forward(_t(nodes, dev), _t(ea, dev), _t(ei, dev)).detach()
    assert (out_train - out_inf).abs().max() <= 1e-5 * max(float(out_inf.abs().max()), 0.1)


def test_optimizer_steps_at_a_padded_hidden_size_with_a_frozen_decoder(dev):
    """Several Adam steps at hidden 100 (run zero-padded at 128) with the decoder frozen, against the same loop of the float64
    PyTorch restatement.  The padded parameter tensors are rebuilt by every forward (version 0, recycled addresses): the packed
    weight streams must follow the module's REAL parameters -- keyed on the padded copies, nothing in the key changes once the
    decoder (whose last bias passes through unpadded) is left out of the optimiser, and every step after the first would run on
    step 1's weights (round-5 advisor finding).  The per-step losses are the witness: they track the oracle's, step by step."""
    dims = (25, 4, 3, 100, 2, 2)
    params = orc.init_params(*dims, 143)
    m = _model(params, dims, dev)
    m.decoder.requires_grad_(False)
    nodes, ea, ei = _graph(500, 0.07, 143)
    target = np.random.default_rng(143).standard_normal((nodes.shape[0], 3)).astype(np.float32)
    x, a, idx, tgt = _t(nodes, dev), _t(ea, dev), _t(ei, dev), _t(target, dev)
    lr, steps = 2e-3, 5
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=lr)
    got = []
    for _ in range(steps):
        out = m.forward(x, a, idx)
        loss = torch.nn.functional.l1_loss(out, tgt, reduction="sum") / out.shape[0]
        opt.zero_grad()
        loss.backward()
        opt.step()
        got.append(float(loss.detach()))
    p64 = {k: torch.tensor(v, dtype=torch.float64, requires_grad=not k.startswith("decoder.")) for k, v in params.items()}
    opt64 = torch.optim.Adam([v for v in p64.values() if v.requires_grad], lr=lr)
    n64, e64, i64, t64 = (torch.tensor(nodes, dtype=torch.float64), torch.tensor(ea, dtype=torch.float64),
                          torch.tensor(ei, dtype=torch.int64), torch.tensor(target, dtype=torch.float64))
    ref = []
    for _ in range(steps):
        o = torch_epd.epd_forward(p64, n64, e64, i64, dims[4], dims[5])
        l = torch.nn.functional.l1_loss(o, t64, reduction="sum") / o.shape[0]
        opt64.zero_grad()
        l.backward()
        opt64.step()
        ref.append(float(l.detach()))
    assert ref[-1] < ref[0] - 5e-3 * ref[0], ref            # the steps do move the loss: stale weights would show
    np.testing.assert_allclose(got, ref, rtol=1e-3)    # float32 against float64 through five Adam steps (sign-like first updates)
    assert all(p.grad is None for p in m.decoder.parameters())
    # ... and the inference path sees the trained weights; invalidate_packed_weights() reaches the padded model too
    with torch.no_grad():
        out_inf = m.forward(x, a, idx)
        for k, v in p64.items():
            if not k.startswith("decoder."):
                dict(m.named_parameters())[k].data.copy_(v.detach().float())   # a write behind autograd's back ...
        m.invalidate_packed_weights()                                           # ... announced
    o_ref = torch_epd.epd_forward(p64, n64, e64, i64, dims[4], dims[5]).detach().numpy()
    out_t = m.forward(x, a, idx).detach().cpu().numpy()                         # training forward on the copied weights
    assert np.abs(out_t - o_ref).max() <= 1e-5 * max(np.abs(o_ref).max(), 1e-3)
    with torch.no_grad():
        out_i2 = m.forward(x, a, idx).cpu().numpy()                             # ... and the fused inference path (the module's own handle)
    assert np.abs(out_i2 - o_ref).max() <= 1e-5 * max(np.abs(o_ref).max(), 1e-3)
    assert np.isfinite(out_inf.cpu().numpy()).all() and np.abs(out_inf.cpu().numpy() - out_i2).max() > 0   # the copy did change the weights


def test_backward_over_many_seeds(dev):
    """The single-seed tests above use seeds on which no pre-activation sits within rounding distance of zero.  Over a run of
    seeds that cannot hold: a ReLU whose sign differs between two float32-accurate evaluations moves the gradients by ~1e-4 ..
    1e-3 of a tensor -- and plain PyTorch float32 shows the same deviation from float64 on the same seed.  So every seed is held
    to the yardstick of _check(): each gradient within max(GRAD_TOL, 4 x the error of PyTorch's own float32 evaluation of that
    tensor on that seed) of the float64 gradient -- plus, only where that fails, what toggling the units of that seed whose
    float64 pre-activation is within 1e-5 rms of zero explains (_compare_gradients) -- and forward 1e-5.  No seed is exempt and
    none is hand-picked."""
    dims = (25, 4, 3, 128, 2, 2)
    for seed in range(200, 210):
        params = orc.init_params(*dims, seed)
        m = _model(params, dims, dev)
        nodes, ea, ei = _graph(300, 0.06, seed)
        _check(m, params, nodes, ea, ei, dims, dev, seed)


def test_backward_collated_batch_of_two(dev, golden):
    """The training loader's input: two graphs collated with the index offset (collate_utils.py:68-87)."""
    g4 = golden("g4_features.npz")
    nodes, ea, ei, _ = orc.process_collate([(g4["obs_a"], g4["tgt_a"]), (g4["obs_b"], g4["tgt_b"])], control_idx=CTRL, **KW)
    dims = (25, 4, 3, 128, 2, 2)
    params = orc.init_params(*dims, 95)
    m = _model(params, dims, dev)
    _check(m, params, nodes, ea, ei, dims, dev, 95)


def test_training_step_reduces_loss_and_inference_sees_new_weights(dev):
    """train_dyn.py:49-72: zero_grad / backward / Adam step, repeated; the fused inference path then runs on the
    updated parameters (re-pack on change) and agrees with the training forward."""
    dims = (25, 4, 3, 128, 2, 2)
    params = orc.init_params(*dims, 96)
    m = _model(params, dims, dev)
    nodes, ea, ei = _graph(500, 0.06, 96)
    x, a, idx = _t(nodes, dev), _t(ea, dev), _t(ei, dev)
    tgt = torch.from_numpy(np.random.default_rng(96).standard_normal((nodes.shape[0], 3)).astype(np.float32)).to(dev)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    losses = []
    for _ in range(5):
        out = m.forward(x, a, idx)
        loss = torch.nn.functional.l1_loss(out, tgt, reduction="sum") / out.shape[0]
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0]
    out_train = m.forward(x, a, idx).detach()
    with torch.no_grad():
        out_inf = m.forward(x, a, idx)
    # two different kernel sets (tape-recording bf16 x 3 chains vs fused fp16 x 3 inference kernels) on the updated weights
    assert (out_train - out_inf).abs().max() <= 1e-5 * max(float(out_inf.abs().max()), 0.1)


@pytest.mark.parametrize("n,side,seed,m_steps,hidden,num_layers", [(800, 0.07, 105, 3, 128, 2), (150, 0.3, 98, 2, 128, 2),
                                                                    (500, 0.07, 125, 2, 64, 3)])
def test_reference_wiring_over_standalone_blocks_trains(dev, n, side, seed, m_steps, hidden, num_layers):
    """The reference's own forward (epd_gnn.py:86-105: encoder block, m x (block + residuals), torch decoder) run over
    the standalone GraphIndependent / InteractionNetwork modules under autograd: every parameter gradient against the
    float64 oracle.  Exercises the block-level backward incl. the InteractionNetwork's input gradients.  (Without a ReLU
    sign flip between two float32-accurate evaluations the observed error is ~1e-6 on every tensor; what a flip may add is
    bounded per seed by _compare_gradients.)"""
    dims = (25, 4, 3, hidden, num_layers, m_steps)
    params = orc.init_params(*dims, seed)
    m = _model(params, dims, dev)
    nodes, ea, ei = _graph(n, side, seed)
    rng = np.random.default_rng(seed)
    target = rng.standard_normal((nodes.shape[0], 3)).astype(np.float32)
    x, a, idx = _t(nodes, dev), _t(ea, dev), _t(ei, dev)
    h, e, _ = m.encoder(x, a, idx)
    for blk in m.processor:
        h, e = m._process(blk, h, e, idx)
    out = m.decoder(h)  # plain torch Sequential, as in the reference
    loss = torch.nn.functional.l1_loss(out, _t(target, dev), reduction="sum") / out.shape[0]
    loss.backward()
    ref_out, ref_loss, ref_g = torch_epd.loss_and_grads(params, nodes, ea, ei, target, num_layers, m_steps)
    _, _, g32 = torch_epd.loss_and_grads(params, nodes, ea, ei, target, num_layers, m_steps, torch.float32)
    assert np.abs(out.detach().cpu().numpy() - ref_out).max() <= 1e-5 * max(np.abs(ref_out).max(), 1e-3)
    _compare_gradients(m, params, nodes, ea, ei, target, num_layers, m_steps, ref_g, g32)


@pytest.mark.parametrize("n,with_edges", [(5, False), (1, True), (130, False)])
def test_backward_degenerate_graphs(dev, n, with_edges):
    """No edges at all (every node isolated) and a single node with its self edge: the edge-side kernels get zero /
    one row, the edge MLP gradients are exactly what autograd gives (zeros without edges)."""
    dims = (25, 4, 3, 128, 2, 2)
    params = orc.init_params(*dims, 99)
    m = _model(params, dims, dev)
    rng = np.random.default_rng(n)
    nodes = rng.standard_normal((n, 25)).astype(np.float32)
    if with_edges:
        ei = np.stack((np.arange(n), np.arange(n))).astype(np.int64)
        ea = np.zeros((n, 4), np.float32)
    else:
        ei = np.zeros((2, 0), np.int64)
        ea = np.zeros((0, 4), np.float32)
    _check(m, params, nodes, ea, ei, dims, dev, 99)


def test_graph_independent_input_gradients(dev):
    """Encoder block alone with inputs that require grad (not the reference's use, where they are data, but part of the
    module contract): d loss / d x and d loss / d edge_attr against float64 autograd."""
    dims = (25, 4, 3, 128, 2, 2)
    params = orc.init_params(*dims, 101)
    m = _model(params, dims, dev)
    nodes, ea, ei = _graph(300, 0.06, 101)
    x = _t(nodes, dev).requires_grad_(True)
    a = _t(ea, dev).requires_grad_(True)
    h, e, _ = m.encoder(x, a, _t(ei, dev))
    rng = np.random.default_rng(101)
    wh = rng.standard_normal(h.shape).astype(np.float32)
    we = rng.standard_normal(e.shape).astype(np.float32)
    ((h * _t(wh, dev)).sum() + (e * _t(we, dev)).sum()).backward()
    p64 = {k: torch.tensor(v, dtype=torch.float64) for k, v in params.items()}
    x64 = torch.tensor(nodes, dtype=torch.float64, requires_grad=True)
    a64 = torch.tensor(ea, dtype=torch.float64, requires_grad=True)
    h64 = torch_epd.mlp(p64, "encoder.phi_node", x64, 2, True)
    e64 = torch_epd.mlp(p64, "encoder.phi_edge", a64, 2, True)
    ((h64 * torch.tensor(wh, dtype=torch.float64)).sum() + (e64 * torch.tensor(we, dtype=torch.float64)).sum()).backward()
    for got, ref in ((x.grad, x64.grad), (a.grad, a64.grad)):
        ref = ref.numpy()
        assert np.abs(got.cpu().numpy() - ref).max() <= 2e-4 * np.abs(ref).max()


def test_training_forward_flags_a_bad_edge_index_without_blocking(dev):
    """The training forward makes no blocking range check of edge_index (a training loop queues its steps ahead of the GPU): an
    entry outside [0, n_nodes) is flagged and left out by the destination sort on the device, the slots past the valid edges get
    valid indices (the kernels walk all E slots with the host-side count and must stay inside their arrays), and the flag
    surfaces as GMError at a later forward or at status() -- the step itself completes (its numbers are not used)."""
    from gnn_manip_amd import EncProcDecGNN
    from gnn_manip_amd._lib import GMError
    torch.manual_seed(3)
    n, e = 300, 4000
    m = EncProcDecGNN(25, 4, 3, 128, 2, 2).to(dev)
    x, ea = torch.randn(n, 25, device=dev), torch.randn(e, 4, device=dev)
    ei = torch.randint(0, n, (2, e), device=dev)
    bad = ei.clone()
    bad[1, 17] = n + 5
    bad[0, 3000] = -2
    out = m.forward(x, ea, bad)
    out.abs().sum().backward()                       # the backward walks the same slots
    torch.cuda.synchronize()
    # the flagged step is visibly unusable and harmless: NaN prediction, exactly zero gradients (an optimiser step taken before
    # the error surfaces does not move the weights by numbers computed on a different graph)
    assert torch.isnan(out).all()
    assert all(p.grad is not None and float(p.grad.abs().max()) == 0.0 for p in m.parameters())
    with pytest.raises(GMError, match="out of range"):
        m.status()
    m.zero_grad()
    out = m.forward(x, ea, ei)                       # a clean step afterwards: nothing pending, gradients finite
    out.abs().sum().backward()
    assert m.status() == 0 and all(torch.isfinite(p.grad).all() for p in m.parameters())
    m.forward(x, ea, bad).sum().backward()
    torch.cuda.synchronize()
    with pytest.raises(GMError, match="out of range"):
        m.forward(x, ea, ei)                         # ... or at the next forward, for a caller that never asks
    m.auto_status = False                            # opting out of the unasked look does not lose the verdict: status() has it
    m.zero_grad()
    m.forward(x, ea, bad).sum().backward()
    m.forward(x, ea, ei)
    with pytest.raises(GMError, match="out of range"):
        m.status()
