"""The PINN step as layer-block launches (openpystruct_amd/pinn_fused.py, csrc/mlp_block.hip) against the module itself:
FNNWithResidual + CompositeLoss (/root/reference/OpenPyStruct_PINN_MultiCase.py:395-653) evaluated in float64 on the CPU by
autograd, with the dropout masks the launches drew.  The launches compute in bfloat16 with float32 accumulation (what bf16
autocast does), so the bounds are bf16 bounds: 2e-2 on values, 4e-2 relative L2 on gradients -- a wrong or missing term
(stencil path, residual, a normalisation's backward) shows up as O(1)."""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


class _FixedMask(nn.Module):
    def __init__(self):
        super().__init__()
        self.mask, self.scale = None, 1.0

    def forward(self, x):
        return x if self.mask is None else x * self.mask * self.scale


def _make(seed, p_drop, nblk=2, F=684, H=350, C=302, nel=100, smooth=False):
    from openpystruct_amd.surrogates import CompositeLoss, FNNWithResidual
    torch.manual_seed(seed)
    model = FNNWithResidual(F, H, nblk, C, p_drop)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() == 2:                                  # Linear weights: bf16-representable, so both sides use the same numbers
                p.copy_(p.to(torch.bfloat16).float())
            elif "conv1.weight" in name:
                p.copy_(torch.tensor([0.6, -0.9, 0.5]).reshape(p.shape) * (1 + 0.1 * torch.randn(p.shape, generator=g)))
            elif "bn1.weight" in name:
                p.fill_(1.3)
            elif "norm" in name and "weight" in name:
                p.copy_(1.0 + 0.3 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.2 * torch.randn(p.shape, generator=g))
        for name, b in model.named_buffers():
            if "running_mean" in name:
                b.copy_(0.1 * torch.randn(b.shape, generator=g))
            elif "running_var" in name:
                b.copy_(1.0 + 0.2 * torch.rand(b.shape, generator=g))
    if smooth:      # pure mean-squared error on the inertia block: no |.|, no box penalty, no relative-L1 term -- a loss whose
        # gradient does not flip sign with the bf16 rounding of a prediction
        crit = CompositeLoss(nel, nel + 1, C - 2 * nel - 1, 0.0, 0.0, -1.0e6, 1.0e6, 0.0)
    else:
        crit = CompositeLoss(nel, nel + 1, C - 2 * nel - 1, 0.5, 1e-1, -1.0, 1.2, 1.5e-2)
    return model, crit


def _attach_flat(model):
    params = list(model.parameters())
    flat = torch.zeros(sum(q.numel() for q in params), device=params[0].device)
    off = 0
    for q in params:
        q.grad = flat[off:off + q.numel()].view_as(q)
        off += q.numel()
    return flat


def _reference(model, crit, x, y, masks, p_drop):
    """float64 CPU autograd over the module (its plain tensor-op path), dropout replaced by the given masks."""
    ref = copy.deepcopy(model).cpu().double()
    for q in ref.parameters():
        q.grad = None
    rc = copy.deepcopy(crit).cpu().double()
    ref.train()
    drops = [_FixedMask() for _ in range(1 + len(ref.residual_blocks))]
    ref.dropout = drops[0]
    for k, blk in enumerate(ref.residual_blocks):
        blk[0].dropout = drops[1 + k]
    if masks is not None:
        for d, m in zip(drops, masks):
            d.mask, d.scale = m.double().cpu(), 1.0 / (1.0 - p_drop)
    preds = ref(x.double().cpu())
    loss = rc(preds, y.double().cpu())
    loss.backward()
    return ref, preds.detach(), float(loss.detach())


def _autocast_reference(before, crit, x, y):
    """The autograd path under bf16 autocast (library GEMMs + csrc/fused_bn.hip tails): the same arithmetic contract."""
    from openpystruct_amd.surrogates import fused_loss
    m2 = copy.deepcopy(before)
    _attach_flat(m2)
    m2.train()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = fused_loss(crit, m2(x), y)
    loss.backward()
    torch.cuda.synchronize()
    return m2, float(loss.detach())


def _floor(name, grads):
    """A bias in front of a normalisation has a mathematically zero gradient (so have conv1.bias and bn1.bias in front of the
    block norm's mean subtraction... only approximately): measure those against their layer's weight-gradient norm."""
    for suffix, other in (("input_fc.bias", "input_fc.weight"), ("fc2.bias", "fc2.weight"), ("conv1.bias", "conv1.weight"),
                          ("bn1.bias", "bn1.weight")):
        if name.endswith(suffix):
            return float(grads[name[:-len(suffix)] + other].norm())
    return 0.0


@pytest.mark.parametrize("B,p_drop,seed", [(128, 0.0, 0), (77, 0.0, 1), (128, 0.0, 10), (9, 0.0, 4), (128, 0.5, 2), (33, 0.3, 3)])
def test_step_matches_module_autograd(B, p_drop, seed):
    from openpystruct_amd.pinn_fused import PinnFusedStep, eligible
    dev = torch.device("cuda:0")
    model, crit = _make(seed, p_drop)
    model, crit = model.to(dev), crit.to(dev)
    _attach_flat(model)
    assert eligible(model, crit, 128)
    before = copy.deepcopy(model)                # parameters and buffers before the step (the launches update the buffers)
    g = torch.Generator().manual_seed(100 + seed)
    x = torch.randn(B, 684, generator=g).to(torch.bfloat16).float().to(dev)
    y = (0.8 * torch.randn(B, 302, generator=g)).to(dev)
    eng = PinnFusedStep(model, crit, seed=1234 + seed)
    model.train()
    eng.set_batch(x, y)
    loss = eng.fwd_bwd(B)
    torch.cuda.synchronize()
    masks = None
    if p_drop > 0:
        masks = [eng.read(eng.o[0][0], B, 350) != 0] + [eng.read(eng.h[k][0], B, 175) != 0 for k in range(2)]
        for m in masks:
            assert abs(float(m.float().mean()) - (1 - p_drop)) < 0.03
    ref, preds_ref, loss_ref = _reference(before, crit, x, y, masks, p_drop)
    preds = eng.predictions(B).float().cpu().double()
    assert float((preds - preds_ref).abs().max()) <= 2e-2 * float(preds_ref.abs().max())
    assert abs(float(loss) - loss_ref) <= 1e-2 * abs(loss_ref)
    gref = {n: q.grad for n, q in ref.named_parameters()}
    got = {n: q.grad.detach().cpu().double() for n, q in model.named_parameters()}
    rel = lambda a, b, n: float((a - b).norm()) / (float(b.norm()) + _floor(n, gref) + 1e-30)      # noqa: E731
    if p_drop == 0.0:
        # (a) against the autograd path with the same arithmetic contract: the two bf16 paths agree far better than either
        # agrees with float64 (they share the L1 term's sign flips and the roundings of every layer output)
        m2, loss2 = _autocast_reference(before, crit, x, y)
        assert abs(float(loss) - loss2) <= 1e-3 * abs(loss2)
        for n, q in m2.named_parameters():
            ga = q.grad.detach().cpu().double()
            e_blocks, e_auto = rel(got[n], gref[n], n), rel(ga, gref[n], n)
            assert e_blocks <= 1.5 * e_auto + 2e-2, (n, e_blocks, e_auto)
    # (b) against float64 autograd with the masks the launches drew: bf16 bounds (sums with cancellation -- the norms' biases,
    # bn1.weight -- carry 10-30 % at these sizes on BOTH bf16 paths); a wrong term is O(1)
    for n in got:
        e = rel(got[n], gref[n], n)
        bound = 0.35 if (n.endswith(".bias") or "bn1" in n or "conv1" in n or B < 16) else (0.12 if p_drop == 0.0 else 0.2)
        assert np.isfinite(e) and e <= bound, (n, e, float(gref[n].norm()))
    # BatchNorm buffers after one training step
    for (name, b), (_, br) in zip(model.named_buffers(), ref.named_buffers()):
        if "num_batches" in name:
            assert int(b) == int(br), name
        else:
            assert float((b.cpu().double() - br).abs().max()) <= 1e-2 * max(1e-3, float(br.abs().max())), name
    # dead rows / columns of every buffer stay zero (the layout contract the next product relies on)
    from openpystruct_amd.pinn_fused import from_tiled
    if B < 128:
        for t in [eng.o[0][0], eng.o[1][0], eng.o[2][0], eng.h[0][0], eng.dz[0][0], eng.dh[1][0], eng.gp]:
            assert float(from_tiled(t)[B:].float().abs().max()) == 0.0
        for t in [eng.o[0][1], eng.h[1][1], eng.dz[2][1], eng.gpt]:
            assert float(from_tiled(t)[:, B:].float().abs().max()) == 0.0
    assert float(from_tiled(eng.h[0][0])[:, 175:].float().abs().max()) == 0.0
    assert float(from_tiled(eng.o[1][1])[350:].float().abs().max()) == 0.0
    # both copies of a matrix hold the same numbers
    for a_, at_ in (eng.o[1], eng.h[0], eng.dz[1], eng.dh[0]):
        assert torch.equal(from_tiled(a_), from_tiled(at_).t())
    assert float(eng.loss_sum) == pytest.approx(float(loss), rel=1e-6)


def _floor2(name, grads):
    """Gradient scale of the layer a parameter belongs to.  Several parameters of this network have a mathematically zero or
    nearly cancelling gradient (a bias in front of a mean-subtracting normalisation; the block norms' biases; with identity
    activations every Linear bias; the stencil path's scalar BatchNorm pair and convolution bias): any bf16 evaluation returns
    rounding noise there, so their error is measured against their layer's principal gradient -- the same module's weight, for the
    stencil path's scalars its three convolution taps."""
    if "bn1." in name or name.endswith("conv1.bias"):
        return float(grads[name.rsplit(".", 2)[0] + ".conv1.weight"].norm())
    if name.endswith(".bias"):
        return float(grads[name[:-len("bias")] + "weight"].norm())
    return 0.0


def _run_engine(B, p_drop, seed, slope=None):
    from openpystruct_amd.pinn_fused import PinnFusedStep, eligible
    dev = torch.device("cuda:0")
    model, crit = _make(seed, p_drop, smooth=True)
    if slope is not None:
        for m in model.modules():
            if isinstance(m, nn.LeakyReLU):
                m.negative_slope = slope
    model, crit = model.to(dev), crit.to(dev)
    _attach_flat(model)
    assert eligible(model, crit, 128)
    before = copy.deepcopy(model)
    g = torch.Generator().manual_seed(200 + seed)
    x = torch.randn(B, 684, generator=g).to(torch.bfloat16).float().to(dev)
    y = (0.8 * torch.randn(B, 302, generator=g)).to(dev)
    eng = PinnFusedStep(model, crit, seed=4321 + seed)
    model.train()
    eng.set_batch(x, y)
    loss = eng.fwd_bwd(B)
    torch.cuda.synchronize()
    masks = None
    if p_drop > 0:
        masks = [eng.read(eng.o[0][0], B, 350) != 0] + [eng.read(eng.h[k][0], B, 175) != 0 for k in range(2)]
    got = {n: q.grad.detach().cpu().double() for n, q in model.named_parameters()}
    return before, crit, x, y, masks, float(loss), got


@pytest.mark.parametrize("B,seed", [(128, 20), (77, 22), (16, 23)])
def test_step_gradients_against_the_autocast_path_with_a_smooth_objective(B, seed):
    """The tight guard on the MFMA backward launches.  With the module's own loss bf16 gradients cannot be matched to float64
    tighter than 12-35 % (test above: sign flips of the L1 terms at bf16-rounded predictions).  A smooth objective -- the same
    CompositeLoss configured as a pure mean-squared error -- removes those, and what remains is the arithmetic contract's own
    error (LeakyReLU branches taken on bf16-rounded pre-activations: 2-8 % on the fc1 layers, measured equal on both bf16
    paths).  So the launches are held to the OTHER implementation of that contract -- autograd under bf16 autocast over the
    library products and csrc/fused_bn.hip / stencil_bn.hip tails -- parameter by parameter at 2 %: two independent
    implementations agree to 0.1-0.8 % (profiles/r03_notes.md), a wrong factor in a backward tail is 10 % or more."""
    before, crit, x, y, _, loss, got = _run_engine(B, 0.0, seed)
    ref, _, loss_ref = _reference(before, crit, x, y, None, 0.0)
    m2, loss2 = _autocast_reference(before, crit, x, y)
    assert abs(loss - loss2) <= 2e-4 * abs(loss2) and abs(loss - loss_ref) <= 1e-2 * abs(loss_ref)
    gref = {n: q.grad for n, q in ref.named_parameters()}
    for n, q in m2.named_parameters():
        ga = q.grad.detach().cpu().double()
        fl = _floor2(n, gref)
        same = float((got[n] - ga).norm()) / (max(float(ga.norm()), fl) + 1e-30)
        e_blocks = float((got[n] - gref[n]).norm()) / (max(float(gref[n].norm()), fl) + 1e-30)
        e_auto = float((ga - gref[n]).norm()) / (max(float(gref[n].norm()), fl) + 1e-30)
        assert same <= 2e-2, (n, same, e_blocks, e_auto)
        assert e_blocks <= 1.25 * e_auto + 1e-2, (n, e_blocks, e_auto)


@pytest.mark.parametrize("B,p_drop,seed", [(128, 0.0, 30), (128, 0.5, 31), (77, 0.3, 32), (16, 0.2, 33)])
def test_step_gradients_of_a_smooth_network_match_float64(B, p_drop, seed):
    """Smooth objective AND smooth network (LeakyReLU slope 1 = identity: no branch to flip), dropout masks replayed: nothing is left
    but bf16 rounding, and every parameter gradient of the launches must match float64 autograd to 3e-2 relative L2 -- biases,
    stencil path and normalisations included (zero-gradient parameters against their layer's scale, _floor2)."""
    before, crit, x, y, masks, loss, got = _run_engine(B, p_drop, seed, slope=1.0)
    ref, _, loss_ref = _reference(before, crit, x, y, masks, p_drop)
    assert abs(loss - loss_ref) <= 1e-2 * abs(loss_ref)
    gref = {n: q.grad for n, q in ref.named_parameters()}
    worst = {n: float((got[n] - gref[n]).norm()) / (max(float(gref[n].norm()), _floor2(n, gref)) + 1e-30) for n in got}
    # Bounds from a sweep over 12 mask draws per size (scripts/pinn_smooth_seed_sweep.py, profiles/r03_notes.md 6): every matrix / vector
    # parameter <= 3e-2 (1.8e-2 worst seen); the stencil path's SCALARS (3 taps + a 1-channel norm: sums of B x 175 products of mixed
    # sign) 0.6-2.6e-2 at 128 rows, up to 4.2e-2 at 33 rows and 6.6e-2 at 16 rows.
    def bound(n):
        if "conv1" in n or "bn1" in n:
            return 3e-2 if B >= 64 else 8e-2
        return 3e-2 if B >= 32 else 5e-2
    bad = {n: e for n, e in worst.items() if not (np.isfinite(e) and e <= bound(n))}
    assert not bad, (bad, worst)


def test_dropout_masks_change_between_calls_and_under_graph_replay():
    from openpystruct_amd.pinn_fused import PinnFusedStep
    dev = torch.device("cuda:0")
    model, crit = _make(5, 0.5)
    model, crit = model.to(dev), crit.to(dev)
    _attach_flat(model)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(128, 684, generator=g).to(dev)
    y = torch.randn(128, 302, generator=g).to(dev)
    eng = PinnFusedStep(model, crit, seed=99)
    model.train()
    eng.set_batch(x, y)
    live = lambda: (eng.read(eng.o[0][0], 128, 350) != 0).clone()      # noqa: E731
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        eng.fwd_bwd(128)
        m1 = live()
        eng.fwd_bwd(128)
        m2 = live()
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            eng.fwd_bwd(128)
    torch.cuda.current_stream().wait_stream(side)
    graph.replay(); torch.cuda.synchronize()
    m3 = live()
    graph.replay(); torch.cuda.synchronize()
    m4 = live()
    for a, b in ((m1, m2), (m2, m3), (m3, m4)):
        assert 0.4 < float((a != b).float().mean()) < 0.6
    assert torch.isfinite(eng.loss)


@pytest.mark.parametrize("nblk,H,B", [(2, 350, 128), (3, 350, 77), (1, 64, 16)])
def test_call_counters_advance_once_per_drawing_launch(nblk, H, B):
    """csrc/call_counter.hpp: [calls, tally].  The fc1 launches carry side-job workgroups and the gather launch carries target workgroups
    that never read the counter; only the drawing workgroups report, so after n steps the dropout counter stands at n * (1 + blocks)
    with an empty tally, the gather's at n -- and every step draws new masks at every site (r03: the tally was compared with the
    whole grid, the stream froze after a few steps and every later step repeated one mask)."""
    from openpystruct_amd.pinn_fused import PinnFusedStep
    dev = torch.device("cuda:0")
    model, crit = _make(40 + nblk, 0.5, nblk=nblk, H=H)
    model, crit = model.to(dev), crit.to(dev)
    _attach_flat(model)
    g = torch.Generator().manual_seed(9)
    X = torch.randn(400, 684, generator=g).to(dev)
    Y = torch.randn(400, 302, generator=g).to(dev)
    eng = PinnFusedStep(model, crit, seed=5)
    model.train()
    sig = torch.tensor(0.05, device=dev)
    idx = torch.arange(B, device=dev)
    Hh = eng.Hh
    sites = lambda: [(eng.read(eng.o[0][0], B, H) != 0).clone()] + [(eng.read(eng.h[k][0], B, Hh) != 0).clone() for k in range(nblk)]  # noqa: E731
    prev, prev_x = None, None
    n = 12
    for step in range(1, n + 1):
        eng.gather(X, Y, idx, sig, 17)
        eng.fwd_bwd(B)
        torch.cuda.synchronize()
        assert eng.drop_counter.tolist() == [step * (1 + nblk), 0], (step, eng.drop_counter.tolist())
        assert eng.prep_counter.tolist() == [step, 0], (step, eng.prep_counter.tolist())
        cur, cur_x = sites(), eng.read(eng.x, B, 684).float().clone()
        for m in cur:
            assert 0.42 < float(m.float().mean()) < 0.58           # keep rate 0.5
        if prev is not None:
            for a, b in zip(prev, cur):
                assert 0.4 < float((a != b).float().mean()) < 0.6    # independent masks: half the entries differ
            assert float((cur_x - prev_x).abs().mean()) > 0.02      # the same rows, another noise field
        prev, prev_x = cur, cur_x


def test_invalid_layouts_are_refused():
    from openpystruct_amd import _cabi
    lib = _cabi.load()
    a = _cabi.MlpStripArgs()
    assert lib.ops_mlp_strip_launch(a, None) == _cabi.ERR_INVALID_ARG
    t = torch.zeros(128, 64, dtype=torch.bfloat16, device="cuda:0")
    a.B, a.N, a.K, a.A, a.lda, a.W, a.ldw, a.Y, a.ldy = 200, 16, 32, t.data_ptr(), 64, t.data_ptr(), 64, t.data_ptr(), 64
    assert lib.ops_mlp_strip_launch(a, None) == _cabi.ERR_INVALID_ARG          # more rows than a workgroup owns
    a.B, a.lda = 64, 20
    assert lib.ops_mlp_strip_launch(a, None) == _cabi.ERR_INVALID_ARG          # leading dimension not a multiple of 8
    a.lda, a.tail = 64, _cabi.MLP_TAIL_BN
    assert lib.ops_mlp_strip_launch(a, None) == _cabi.ERR_INVALID_ARG          # normalisation without its parameters


def test_gather_writes_both_layouts_and_the_targets():
    from openpystruct_amd.pinn_fused import PinnFusedStep, from_tiled
    dev = torch.device("cuda:0")
    model, crit = _make(6, 0.0)
    model, crit = model.to(dev), crit.to(dev)
    _attach_flat(model)
    eng = PinnFusedStep(model, crit, seed=3)
    g = torch.Generator().manual_seed(8)
    X = torch.randn(500, 684, generator=g).to(dev)
    Y = torch.randn(500, 302, generator=g).to(dev)
    idx = torch.randperm(500, generator=g)[:93].to(dev)
    B = eng.gather(X, Y, idx, torch.zeros((), device=dev), 17)
    torch.cuda.synchronize()
    xb = X[idx].to(torch.bfloat16)
    assert torch.equal(from_tiled(eng.x)[:B, :684], xb) and torch.equal(from_tiled(eng.xt)[:684, :B], xb.t())
    assert float(from_tiled(eng.x)[B:].float().abs().max()) == 0.0 and float(from_tiled(eng.x)[:, 684:].float().abs().max()) == 0.0
    assert torch.equal(eng.targets_t[:, :B], Y[idx].t()) and float(eng.targets_t[:, B:].abs().max()) == 0.0
    # noise: N(0, sigma^2) on top, a fresh stream per call
    sig = torch.tensor(0.05, device=dev)
    eng.gather(X, Y, idx, sig, 17)
    n1 = (from_tiled(eng.x)[:B, :684].float() - X[idx]).clone()
    eng.gather(X, Y, idx, sig, 17)
    n2 = from_tiled(eng.x)[:B, :684].float() - X[idx]
    assert abs(float(n1.std()) - 0.05) < 0.005 and abs(float(n1.mean())) < 0.002 and float((n1 - n2).abs().mean()) > 0.02


def test_tiled_layout_helpers_roundtrip():
    from openpystruct_amd.pinn_fused import from_tiled, to_tiled
    x = torch.arange(48 * 96, dtype=torch.float32).reshape(48, 96)
    t = to_tiled(x)
    assert torch.equal(from_tiled(t), x)
    # element (row, k) sits at tile * 512 + ((k >> 3) & 3) * 128 + (row & 15) * 8 + (k & 7)
    for row, k in ((0, 0), (17, 40), (47, 95), (5, 33)):
        off = ((row >> 4) * 3 + (k >> 5)) * 512 + ((k >> 3) & 3) * 128 + (row & 15) * 8 + (k & 7)
        assert float(t.reshape(-1)[off]) == float(x[row, k])


@pytest.mark.parametrize("B", [128, 37])
def test_evaluation_pass_of_the_engine_equals_the_module_in_eval_mode(B):
    """PinnFusedStep.evaluate: the forward stages with running statistics (BatchNorm1d layers and the stencil's BatchNorm1d(1)), no
    dropout, loss on the tile -- against model.eval() under bf16 autocast + the criterion, after two training steps have moved the
    running statistics; nothing but the evaluation's own buffers changes."""
    from openpystruct_amd.pinn_fused import PinnFusedStep
    dev = torch.device("cuda:0")
    model, crit = _make(7, 0.3)
    model, crit = model.to(dev), crit.to(dev)
    _attach_flat(model)
    g = torch.Generator().manual_seed(3)
    eng = PinnFusedStep(model, crit, seed=99)
    model.train()
    for _ in range(2):                                       # running statistics away from their initial values
        xt = torch.randn(128, 684, generator=g).to(dev)
        yt = (0.8 * torch.randn(128, 302, generator=g)).to(dev)
        eng.set_batch(xt, yt)
        eng.fwd_bwd(128)
    torch.cuda.synchronize()
    x = torch.randn(B, 684, generator=g).to(torch.bfloat16).float().to(dev)
    y = (0.8 * torch.randn(B, 302, generator=g)).to(dev)
    bufs = {n: b.clone() for n, b in model.named_buffers()}
    grads = {n: q.grad.clone() for n, q in model.named_parameters()}
    eng.eval_loss_sum.zero_()
    eng.set_batch(x, y)
    loss = float(eng.evaluate(B))
    preds = eng.predictions(B).float()
    loss2 = float(eng.evaluate(B))
    torch.cuda.synchronize()
    model.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        pr = model(x).float()
        lr = float(crit(pr, y))
    assert float((preds - pr).norm() / pr.norm()) < 2e-2
    assert abs(loss - lr) <= 1e-2 * abs(lr) and loss2 == loss
    assert float(eng.eval_loss_sum) == pytest.approx(2 * loss, rel=1e-6)
    for n, b in model.named_buffers():
        assert torch.equal(b, bufs[n]), n                    # running statistics / counters untouched
    for n, q in model.named_parameters():
        assert torch.equal(q.grad, grads[n]), n


def test_evaluation_slots_equal_the_batch_by_batch_evaluation():
    """make_eval_slots / evaluate_slots (one launch per stage for a whole validation set, grid.y = batches, the last one ragged)
    against gather + evaluate of each batch on the engine's own buffers: the same arithmetic per batch, bit for bit -- losses and
    predictions -- and neither the model's buffers nor the training batch in the engine's buffers are touched."""
    from openpystruct_amd.pinn_fused import PinnFusedStep
    dev = torch.device("cuda:0")
    model, crit = _make(11, 0.3)
    model, crit = model.to(dev), crit.to(dev)
    _attach_flat(model)
    g = torch.Generator().manual_seed(5)
    eng = PinnFusedStep(model, crit, seed=1)
    model.train()
    for _ in range(2):
        eng.set_batch(torch.randn(128, 684, generator=g).to(dev), (0.8 * torch.randn(128, 302, generator=g)).to(dev))
        eng.fwd_bwd(128)
    n, bs = 5 * 96 + 37, 96
    X = torch.randn(n, 684, generator=g).to(dev).contiguous()
    Y = (0.8 * torch.randn(n, 302, generator=g)).to(dev).contiguous()
    bufs = {k: b.clone() for k, b in model.named_buffers()}
    S = eng.make_eval_slots(X, Y, bs)
    assert S == 6
    x_before = eng.x.clone()
    eng.evaluate_slots()
    eng.evaluate_slots()                                     # idempotent: nothing accumulates into what the losses are read from
    torch.cuda.synchronize()
    slot_losses = eng.eval_slot_losses().clone()
    slot_preds = [eng._slot_preds[i].clone() for i in range(S)]
    assert torch.equal(eng.x, x_before)
    rows = torch.arange(n, device=dev)
    for i in range(S):
        idx = rows[i * bs:(i + 1) * bs]
        B = eng.gather(X, Y, idx, None, 0)
        li = eng.evaluate(B).clone()
        assert float(li) == float(slot_losses[i]), i
        assert torch.equal(eng.predictions(B), slot_preds[i][:B, :eng.C]), i
    for k, b in model.named_buffers():
        assert torch.equal(b, bufs[k]), k
    lib, a = eng.lib, eng._slot_stages[0]
    a.slot_total_rows = n + bs                               # more rows than the slots hold: refused
    assert lib.ops_mlp_strip_launch(__import__("ctypes").byref(a), None) == 1
    a.slot_total_rows = n


@pytest.mark.parametrize("B,p_drop", [(128, 0.5), (48, 0.0)])
def test_gradient_norm_left_by_the_weight_gradient_launch(B, p_drop):
    """r05: `PinnFusedStep.enable_norm` -- the grouped weight-gradient launch leaves the partial sums of ||g||^2 (its tiles' squares +
    extra workgroups over the gradients no matrix covers), advances the optimiser's step and tabulates its bias corrections, and
    `FlatClipAdam` skips its norm launch (OPS_ADAM_NORM_READY).  The partial sums add up to the flat buffer's squared norm; two
    optimiser steps taken that way equal the two-launch form (norm launch + update) to float32 rounding of the clip factor."""
    from openpystruct_amd import _cabi, train
    from openpystruct_amd.pinn_fused import PinnFusedStep
    dev = torch.device("cuda:0")

    def run(fold):
        model, crit = _make(3, p_drop)
        model, crit = model.to(dev), crit.to(dev)
        flat = _attach_flat(model)
        params = list(model.parameters())
        opt = train.FlatClipAdam(params, flat, 5e-4, weight_decay=1e-3, max_norm=0.25)     # (below the gradient norm: the clip factor is in play)
        eng = PinnFusedStep(model, crit, seed=77)
        opt.repack = eng._repack
        if fold:
            opt.norm_ready_parts = eng.enable_norm(flat, opt.ws, opt.step_count, opt.betas)
            assert 0 < opt.norm_ready_parts <= _cabi.FLAT_ADAM_MAX_PARTS
        g = torch.Generator().manual_seed(5)
        x = torch.randn(B, 684, generator=g).to(torch.bfloat16).float().to(dev)
        y = (0.8 * torch.randn(B, 302, generator=g)).to(dev)
        model.train()
        norms = []
        for step in range(2):
            eng.set_batch(x, y)
            eng.fwd_bwd(B)
            torch.cuda.synchronize()
            if fold:
                part = opt.ws.view(torch.float64)
                tot = float(part[:opt.norm_ready_parts].sum())
                want = float((flat.double() ** 2).sum())
                assert abs(tot - want) <= 1e-6 * want, (tot, want)
                assert int(opt.step_count) == step + 1
                assert float(part[_cabi.FLAT_ADAM_MAX_PARTS]) == pytest.approx(1.0 - 0.9 ** (step + 1), rel=1e-6)      # (beta1 travels as float32)
            norms.append(float(flat.double().norm()))
            opt.step()
            torch.cuda.synchronize()
            assert int(opt.step_count) == step + 1
        return opt.p.clone(), norms

    p0, n0 = run(False)
    p1, n1 = run(True)
    assert n0[0] == n1[0] and n0[0] > 0.25             # the first step's gradients are the same launches' output (and the clip is active)
    assert float((p0 - p1).abs().max()) <= 2e-6 * float(p0.abs().max())


def test_weight_copies_rebuilt_by_the_next_gather_equal_the_optimisers_repack_launch():
    """r05: `PinnFusedStep.repack_in_gather` -- the tiled bf16 weight copies of optimiser step n are rebuilt by extra workgroups of the
    batch-assembly launch of step n + 1 (ops_mlp_gather_noise_repack) instead of a launch of their own behind the update.  Three training
    steps either way: the same float32 parameters bit for bit, the same copies after the closing `repack_now()`, and the copies a step
    reads are the ones of the update before it."""
    from openpystruct_amd import train
    from openpystruct_amd.pinn_fused import PinnFusedStep
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(9)
    X = torch.randn(300, 684, generator=g).to(dev)
    Y = (0.8 * torch.randn(300, 302, generator=g)).to(dev)
    sig = torch.tensor(0.01, device=dev)

    def run(merged):
        model, crit = _make(4, 0.5)
        model, crit = model.to(dev), crit.to(dev)
        flat = _attach_flat(model)
        opt = train.FlatClipAdam(list(model.parameters()), flat, 5e-4, weight_decay=1e-3, max_norm=1.0)
        eng = PinnFusedStep(model, crit, seed=21)
        if merged:
            eng.repack_in_gather(opt.p)
        else:
            opt.repack = eng._repack
        model.train()
        losses = []
        for step in range(3):
            idx = torch.arange(100 * step, 100 * step + 100, device=dev)
            eng.gather(X, Y, idx, sig, 5)
            if merged and step > 0:
                # the copies this step reads are those of the update before it: equal to a fresh repack of the current parameters
                torch.cuda.synchronize()
                seen = [t.clone() for pair in eng.wp.values() for t in pair]
                eng.repack_now()
                torch.cuda.synchronize()
                assert all(torch.equal(a, b) for a, b in zip(seen, [t for pair in eng.wp.values() for t in pair]))
            losses.append(float(eng.fwd_bwd(100)))
            opt.step()
        eng.repack_now()
        torch.cuda.synchronize()
        return opt.p.clone(), [t.clone() for pair in eng.wp.values() for t in pair], losses

    p0, w0, l0 = run(False)
    p1, w1, l1 = run(True)
    assert l0 == l1 and torch.equal(p0, p1)
    assert all(torch.equal(a, b) for a, b in zip(w0, w1))
