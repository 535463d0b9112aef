"""Row-wise encoder blocks of the Transformer-Diffusion surrogate (csrc/seq_block.hip, openpystruct_amd/tfd_fused.py) against the
framework: scaled-dot-product attention, dropout + add + LayerNorm, ReLU + dropout, and the patched nn.TransformerEncoder
(/root/reference/OpenPyStruct_TransformerDiffusionModule_MultiCase.py:539-575).  bf16 activations: bounds are bf16 bounds."""
import copy
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).norm()) / (float(b.norm()) + 1e-30)


def _attention_reference(qkv, B, S, H):
    d = qkv.shape[1] // 3
    q, k, v = (t.reshape(B, S, H, d // H).permute(0, 2, 1, 3) for t in qkv.float().split(d, dim=1))
    o = torch.softmax(q @ k.transpose(-1, -2) / (d // H) ** 0.5, dim=-1) @ v
    return o.permute(0, 2, 1, 3).reshape(B * S, d)


@pytest.mark.parametrize("B,S,H,dh", [(37, 7, 8, 15), (512, 7, 8, 15), (5, 8, 4, 32), (3, 1, 2, 8)])
def test_attention_matches_softmax_attention(B, S, H, dh):
    from openpystruct_amd import tfd_fused as TF
    g = torch.Generator().manual_seed(B + S)
    d = H * dh
    qkv = (1.5 * torch.randn(B * S, 3 * d, generator=g)).to(torch.bfloat16).to(DEV).requires_grad_()
    go = torch.randn(B * S, d, generator=g).to(torch.bfloat16).to(DEV)
    st = TF._State(torch.device(DEV), 1)
    out = TF.SeqAttention.apply(qkv, B, S, H, 0.0, st, 0)
    out.backward(go)
    ref_in = qkv.detach().clone().requires_grad_()
    ref = _attention_reference(ref_in, B, S, H)
    ref.backward(go.float())
    assert _rel(out, ref) < 6e-3
    assert _rel(qkv.grad, ref_in.grad) < 1.5e-2


def test_attention_dropout_is_consistent_between_the_passes():
    """With dropout the map v -> ctx is linear for a fixed mask: <g, ctx(v)> = <dv(g), v> (the backward launch regenerates the mask of the
    forward launch), the kept fraction is 1 - p, and a replay draws another mask."""
    from openpystruct_amd import tfd_fused as TF
    B, S, H, dh, p = 64, 7, 8, 15, 0.3
    d = H * dh
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn(B * S, 3 * d, generator=g).to(torch.bfloat16).to(DEV)
    qkv[:, :2 * d] = 0                                                  # uniform attention: every probability 1 / S
    qkv[:, 2 * d:] = 1                                                  # v = 1: ctx = sum of the kept, rescaled probabilities
    st = TF._State(torch.device(DEV), 5)
    o1 = TF.SeqAttention.apply(qkv, B, S, H, p, st, 0).float()
    st.advance()
    o2 = TF.SeqAttention.apply(qkv, B, S, H, p, st, 0).float()
    kept = o1[:, ::dh] * (1 - p) * S                                    # number of kept keys per (row, head)
    assert abs(float(kept.mean()) / S - (1 - p)) < 0.02 and float((o1 != o2).float().mean()) > 0.3
    x = torch.randn(B * S, 3 * d, generator=g).to(torch.bfloat16).to(DEV).requires_grad_()
    go = torch.randn(B * S, d, generator=g).to(torch.bfloat16).to(DEV)
    out = TF.SeqAttention.apply(x, B, S, H, p, st, 0)
    out.backward(go)
    lhs = float((go.double() * out.detach().double()).sum())
    rhs = float((x.grad[:, 2 * d:].double() * x.detach()[:, 2 * d:].double()).sum())
    assert abs(lhs - rhs) < 2e-2 * (abs(lhs) + float(go.double().norm() * out.detach().double().norm()) * 1e-2)


@pytest.mark.parametrize("T,d,res_bf16", [(3584, 120, False), (1027, 256, False), (77, 8, True)])
def test_dropout_add_layernorm_matches_the_modules(T, d, res_bf16):
    from openpystruct_amd import tfd_fused as TF
    g = torch.Generator().manual_seed(T)
    x = torch.randn(T, d, generator=g).to(torch.bfloat16).to(DEV).requires_grad_()
    res = (2 * torch.randn(T, d, generator=g) + 0.5).to(DEV)
    if res_bf16:
        res = res.to(torch.bfloat16)
    res.requires_grad_()
    ln = nn.LayerNorm(d).to(DEV)
    with torch.no_grad():
        ln.weight.copy_(1 + 0.3 * torch.randn(d, generator=g)); ln.bias.copy_(0.2 * torch.randn(d, generator=g))
    g32 = torch.randn(T, d, generator=g).to(DEV)
    g16 = torch.randn(T, d, generator=g).to(torch.bfloat16).to(DEV)
    st = TF._State(torch.device(DEV), 2)
    y32, y16 = TF.DropoutAddLayerNorm.apply(x, res, ln.weight, ln.bias, ln.eps, 0.0, st, 1)
    torch.autograd.backward([y32, y16], [g32, g16])
    xr, rr = x.detach().float().requires_grad_(), res.detach().float().requires_grad_()
    lr = copy.deepcopy(ln)
    lr.weight.grad = lr.bias.grad = None
    yr = lr(rr + xr)
    yr.backward(g32 + g16.float())
    assert _rel(y32, yr) < 1e-5 and _rel(y16.float(), yr) < 4e-3
    assert _rel(x.grad.float(), xr.grad) < 4e-3 and _rel(res.grad.float(), rr.grad) < (4e-3 if res_bf16 else 1e-5)
    assert _rel(ln.weight.grad, lr.weight.grad) < 1e-4 and _rel(ln.bias.grad, lr.bias.grad) < 1e-4
    # only one of the two gradients present
    ln.weight.grad = ln.bias.grad = None
    x.grad = res.grad = None
    y32, y16 = TF.DropoutAddLayerNorm.apply(x, res, ln.weight, ln.bias, ln.eps, 0.0, st, 1)
    y16.backward(g16)
    xr.grad = rr.grad = lr.weight.grad = lr.bias.grad = None
    lr(rr + xr).backward(g16.float())
    assert _rel(x.grad.float(), xr.grad) < 4e-3 and _rel(ln.weight.grad, lr.weight.grad) < 1e-4


def test_dropout_add_layernorm_mask_is_shared_by_both_passes_and_redrawn_per_call():
    from openpystruct_amd import tfd_fused as TF
    T, d, p = 512, 120, 0.25
    g = torch.Generator().manual_seed(9)
    x = (torch.randn(T, d, generator=g).abs() + 0.5).to(torch.bfloat16).to(DEV).requires_grad_()      # > 0: a dropped entry is its row's minimum
    res = torch.zeros(T, d, device=DEV)
    ln = nn.LayerNorm(d).to(DEV)
    st = TF._State(torch.device(DEV), 4)
    masks = []
    for _ in range(2):
        x.grad = None
        st.advance()
        y32, _ = TF.DropoutAddLayerNorm.apply(x, res, ln.weight, ln.bias, ln.eps, p, st, 3)
        y32.backward(torch.randn(T, d, generator=g).to(DEV))
        z = (y32.detach() - ln.bias) / ln.weight                       # = (z - mean) rstd: dropped entries are the row's minimum (x > 0)
        keep_fwd = z > z.min(dim=1, keepdim=True).values + 1e-6
        keep_bwd = x.grad != 0
        assert float((keep_fwd != keep_bwd).float().mean()) < 2e-3     # (a kept gradient can round to exactly zero)
        assert abs(float(keep_bwd.float().mean()) - (1 - p)) < 0.02
        masks.append(keep_bwd)
    assert 0.2 < float((masks[0] != masks[1]).float().mean()) < 0.6


def test_act_dropout():
    from openpystruct_amd import tfd_fused as TF
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3584, 256, generator=g).to(torch.bfloat16).to(DEV).requires_grad_()
    go = torch.randn(3584, 256, generator=g).to(torch.bfloat16).to(DEV)
    st = TF._State(torch.device(DEV), 6)
    y = TF.ActDropout.apply(x, 0.0, 0.0, st, 2)
    y.backward(go)
    assert torch.equal(y, torch.relu(x.detach())) and torch.equal(x.grad, torch.where(x.detach() > 0, go, torch.zeros_like(go)))
    x.grad = None
    y = TF.ActDropout.apply(x, 0.0, 0.1, st, 2)
    y.backward(go)
    pos = x.detach() > 0
    kept = (y != 0)[pos]
    assert abs(float(kept.float().mean()) - 0.9) < 0.01
    assert torch.equal((x.grad != 0)[pos & (go != 0)], (y != 0)[pos & (go != 0)])
    assert _rel(y[y != 0].float(), (x.detach()[y != 0].float() / 0.9)) < 4e-3


def _encoder(d=120, H=8, ff=256, layers=2, p=0.0, seed=0):
    torch.manual_seed(seed)
    layer = nn.TransformerEncoderLayer(d_model=d, nhead=H, dim_feedforward=ff, dropout=p, activation="relu", batch_first=True)
    enc = nn.TransformerEncoder(layer, num_layers=layers)
    with torch.no_grad():
        for n, q in enc.named_parameters():
            if "norm" in n and "weight" in n:
                q.add_(0.2 * torch.randn_like(q))
            elif "bias" in n:
                q.add_(0.1 * torch.randn_like(q))
    return enc.to(DEV)


def test_patched_encoder_matches_the_framework_encoder_under_autocast(monkeypatch):
    from openpystruct_amd import tfd_fused as TF, train
    monkeypatch.setattr(train, "_SPLIT_WGRAD_ROWS", 512)      # 448 rows through the LIBRARY products (r04's default sends them to the split-row kernel: next test)
    enc = _encoder()
    ref = copy.deepcopy(enc)
    params = list(enc.parameters())
    flat = torch.zeros(sum(q.numel() for q in params), device=DEV)
    off = 0
    for q in params:
        q.grad = flat[off:off + q.numel()].view_as(q)
        off += q.numel()
    opt = train.FlatClipAdam(params, flat, 1e-3)
    stash, dst, patched = train.enable_shadow_linears(enc, opt, params, flat)
    assert TF.patch_encoder(enc, seed=11, direct_param_grads=True)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(64, 7, 120, generator=g).to(DEV).requires_grad_()
    w = torch.randn(64, 7, 120, generator=g).to(DEV)
    enc.train(); ref.train()
    flat.zero_()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = enc(x)
        (out * w).sum().backward()
    live = [(dd, ss) for dd, ss in zip(dst, stash) if ss is not None]
    assert len(live) == len(dst)                                        # every registered product ran (448 rows: the library path)
    torch._foreach_copy_([a for a, _ in live], [b for _, b in live])
    xr = x.detach().clone().requires_grad_()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        outr = ref(xr)
        (outr * w).sum().backward()
    assert out.dtype == torch.float32 and _rel(out, outr) < 1e-2
    assert _rel(x.grad, xr.grad) < 3e-2
    for (n, q), (_, qr) in zip(enc.named_parameters(), ref.named_parameters()):
        assert _rel(q.grad, qr.grad) < 6e-2, n              # two bf16 evaluations of the same network
    # evaluation keeps the framework's forward
    enc.eval(); ref.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        assert _rel(enc(x), ref(x)) < 1e-2
    train.disable_shadow_linears(patched)
    TF.unpatch_encoder(enc)
    assert "forward" not in enc.__dict__ and not hasattr(enc.layers[0].self_attn, "_ops_in_proj")


def test_patched_encoder_with_thousands_of_rows_takes_the_split_row_gradient_path():
    """Same comparison at the training batch (512 x 7 tokens): weight and bias gradients through ops_linear_wgrad_accumulate."""
    from openpystruct_amd import tfd_fused as TF, train
    enc = _encoder(seed=3)
    ref = copy.deepcopy(enc)
    params = list(enc.parameters())
    flat = torch.zeros(sum(q.numel() for q in params), device=DEV)
    off = 0
    for q in params:
        q.grad = flat[off:off + q.numel()].view_as(q)
        off += q.numel()
    opt = train.FlatClipAdam(params, flat, 1e-3)
    stash, dst, patched = train.enable_shadow_linears(enc, opt, params, flat)
    assert TF.patch_encoder(enc, seed=12, direct_param_grads=True)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(512, 7, 120, generator=g).to(DEV).requires_grad_()
    w = torch.randn(512, 7, 120, generator=g).to(DEV) / 512
    enc.train(); ref.train()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        (enc(x) * w).sum().backward()
    assert all(ss is None for ss in stash)                              # nothing went through the stash
    xr = x.detach().clone().requires_grad_()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        (ref(xr) * w).sum().backward()
    assert _rel(x.grad, xr.grad) < 3e-2
    for (n, q), (_, qr) in zip(enc.named_parameters(), ref.named_parameters()):
        assert _rel(q.grad, qr.grad) < 6e-2, n              # two bf16 evaluations of the same network
    train.disable_shadow_linears(patched)
    TF.unpatch_encoder(enc)


def test_diffusion_noise_drawn_inside_the_launch():
    """ops_diffusion_noise_draw: step indices uniform over [0, T), noise standard normal (moments, neighbour correlation), the arithmetic
    of ops_diffusion_noise on the draws it reports, the same draws for the same counter value and fresh ones for the next."""
    from openpystruct_amd import _cabi
    lib = _cabi.load()
    rows, d, T = 3584, 120, 1000
    g = torch.Generator().manual_seed(0)
    x = torch.randn(rows, d, generator=g).to(DEV)
    acp = torch.cumprod(1.0 - torch.linspace(1e-4, 0.02, T), 0).to(DEV)
    counter = torch.zeros(1, dtype=torch.int64, device=DEV)

    def draw():
        xn32, xn16 = torch.empty(rows, d, device=DEV), torch.empty(rows, d, dtype=torch.bfloat16, device=DEV)
        sa, sb = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
        t, eps = torch.empty(rows, dtype=torch.int64, device=DEV), torch.empty(rows, d, device=DEV)
        rc = lib.ops_diffusion_noise_draw(rows, d, T, x.data_ptr(), acp.data_ptr(), 12345, counter.data_ptr(), xn32.data_ptr(), xn16.data_ptr(),
                                          sa.data_ptr(), sb.data_ptr(), t.data_ptr(), eps.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        torch.cuda.synchronize()
        return xn32, xn16, sa, sb, t, eps

    xn32, xn16, sa, sb, t, eps = draw()
    assert int(t.min()) >= 0 and int(t.max()) < T
    hist = torch.bincount(t // 100, minlength=10).double().cpu().numpy()
    assert np.all(np.abs(hist - rows / 10) < 5 * np.sqrt(rows / 10 * 0.9)), hist
    e = eps.double().flatten()
    n = e.numel()
    assert abs(float(e.mean())) < 5 / np.sqrt(n) and abs(float(e.var()) - 1.0) < 5 * np.sqrt(2.0 / n)
    assert abs(float((e ** 3).mean())) < 5 * np.sqrt(15.0 / n) and abs(float((e ** 4).mean()) - 3.0) < 5 * np.sqrt(96.0 / n)
    assert abs(float((e[1:] * e[:-1]).mean())) < 5 / np.sqrt(n) and abs(float((eps[1:] * eps[:-1]).double().mean())) < 5 / np.sqrt(n)
    torch.testing.assert_close(sa, acp[t].sqrt(), rtol=1e-6, atol=0)
    torch.testing.assert_close(sb, (1 - acp[t]).sqrt(), rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(xn32, sa[:, None] * x + sb[:, None] * eps, rtol=1e-5, atol=1e-6)
    assert torch.equal(xn16, xn32.to(torch.bfloat16))
    again = draw()
    assert torch.equal(again[4], t) and torch.equal(again[5], eps)                  # same (seed, counter): same draws
    counter += 1
    nxt = draw()
    assert float((nxt[4] == t).float().mean()) < 0.01 and abs(float((nxt[5] * eps).mean())) < 5 / np.sqrt(n)


@pytest.mark.parametrize("draw", [True, False])
def test_patched_model_matches_the_module_with_the_same_noise(monkeypatch, draw):
    """ModelOnePassTransformerWithDiffusion through patch_model (fused diffusion front end, encoder blocks, fused head) vs its own
    forward, both under bf16 autocast, dropout 0, and with torch.randint / torch.randn_like replaced by a replayable stream:
    draw = True: the step indices / noise the front-end launch drew itself (kept and replayed to the module); False: framework draws."""
    from openpystruct_amd import tfd_fused as TF, train
    from openpystruct_amd.surrogates import ModelOnePassTransformerWithDiffusion
    monkeypatch.setattr(TF, "DRAW", draw)
    monkeypatch.setattr(TF, "KEEP_DRAWS", True)
    torch.manual_seed(5)
    model = ModelOnePassTransformerWithDiffusion(6, 120, 100, dropout=0.0).to(DEV)
    ref = copy.deepcopy(model)
    params = list(model.parameters())
    flat = torch.zeros(sum(q.numel() for q in params), device=DEV)
    off = 0
    for q in params:
        q.grad = flat[off:off + q.numel()].view_as(q)
        off += q.numel()
    opt = train.FlatClipAdam(params, flat, 1e-3)
    stash, dst, patched = train.enable_shadow_linears(model, opt, params, flat)
    assert TF.patch_model(model, seed=3, direct_param_grads=True) and "forward" in model.__dict__
    B = 512
    g = torch.Generator().manual_seed(6)
    x = torch.randn(B, 6, 120, generator=g).to(DEV)
    w = torch.randn(B, 100, generator=g).to(DEV) / B
    draws = {}

    def fake_randn_like(t, **kw):
        if "e" not in draws:
            draws["e"] = torch.randn(t.shape, generator=torch.Generator().manual_seed(8)).to(t.device)
        return draws["e"]

    orig_randint = torch.randint

    def fake_randint(lo, hi, size, device=None, **kw):
        if "t" not in draws:
            draws["t"] = orig_randint(lo, hi, size, generator=torch.Generator().manual_seed(7)).to(device)
        return draws["t"]

    monkeypatch.setattr(torch, "randint", fake_randint)
    monkeypatch.setattr(torch, "randn_like", fake_randn_like)
    model.train(); ref.train()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(x)
        (out.float() * w).sum().backward()
    live = [(dd, ss) for dd, ss in zip(dst, stash) if ss is not None]
    if live:
        torch._foreach_copy_([a for a, _ in live], [b for _, b in live])
    if draw:
        t_k, e_k = model.transformer_encoder._ops_dropout_state[x.device].draws
        assert int(t_k.min()) >= 0 and int(t_k.max()) < model.diffusion.T and abs(float(e_k.mean())) < 0.01 and abs(float(e_k.std()) - 1.0) < 0.01
        draws["t"], draws["e"] = t_k.clone(), e_k.clone()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        outr = ref(x)
        (outr.float() * w).sum().backward()
    assert _rel(out.float(), outr.float()) < 1.5e-2
    for (n, q), (_, qr) in zip(model.named_parameters(), ref.named_parameters()):
        assert _rel(q.grad, qr.grad) < 8e-2, n                          # two bf16 evaluations of the same network
    train.disable_shadow_linears(patched)
    TF.unpatch_model(model)
    assert "forward" not in model.__dict__ and "forward" not in model.transformer_encoder.__dict__


def test_head_dropout_mask_is_redrawn_every_step_and_every_graph_replay():
    """The head's dropout (TFD:573, site 103 of the fast path) must draw a fresh mask per step.  With the encoder's dropout off, the
    diffusion noise negligible (beta <= 1e-5) and fc2 replaced by the identity the output IS the dropped hidden vector: its zero
    pattern is the mask.  Two eager passes and two replays of one captured pass must each show a different pattern."""
    from openpystruct_amd import tfd_fused as TF, train
    from openpystruct_amd.surrogates import ModelOnePassTransformerWithDiffusion
    torch.manual_seed(5)
    model = ModelOnePassTransformerWithDiffusion(6, 120, 100, dropout=0.0).to(DEV)
    model.dropout.p = 0.5
    params = list(model.parameters())
    flat = torch.zeros(sum(q.numel() for q in params), device=DEV)
    off = 0
    for q in params:
        q.grad = flat[off:off + q.numel()].view_as(q)
        off += q.numel()
    opt = train.FlatClipAdam(params, flat, 1e-3)
    stash, dst, patched = train.enable_shadow_linears(model, opt, params, flat)
    assert TF.patch_model(model, seed=3, direct_param_grads=True)
    model.fc2 = nn.Identity()
    model.train()
    x = torch.randn(64, 6, 120, generator=torch.Generator().manual_seed(6)).to(DEV)

    def zeros_of(t):
        return (t.float() == 0)

    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        m1 = zeros_of(model(x))
        m2 = zeros_of(model(x))
    for m in (m1, m2):
        assert 0.35 < float(m.float().mean()) < 0.85            # ReLU zeros + p = 0.5 dropout zeros
    assert float((m1 != m2).float().mean()) > 0.1                # another mask in the second pass
    static = x.clone()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side), torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        model(static)                                            # warm-up on the capture stream
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            out = model(static)
    graph.replay(); torch.cuda.synchronize()
    r1 = zeros_of(out).clone()
    graph.replay(); torch.cuda.synchronize()
    r2 = zeros_of(out).clone()
    assert float((r1 != r2).float().mean()) > 0.1                # ... and in every replay of a captured step
    train.disable_shadow_linears(patched)
    TF.unpatch_model(model)


@pytest.mark.parametrize("B,p", [(512, 0.1), (37, 0.1), (3, 0.0)])
def test_one_launch_layer_forward_equals_the_eight_launch_form(monkeypatch, B, p):
    """csrc/seq_layer.hip (the whole encoder layer's forward as one MFMA launch) against the same layer through the separate launches +
    library products: SAME dropout streams (seeds, element indices, call counter), so with dropout on the two forms drop the same
    elements and differ only by the accumulation order of the bf16 products -- outputs and every gradient to 1e-2 relative L2."""
    from openpystruct_amd import tfd_fused as TF, train
    from openpystruct_amd.surrogates import ModelOnePassTransformerWithDiffusion

    def run(layer_fwd, layer_bwd=False):
        monkeypatch.setattr(TF, "LAYER_FWD", layer_fwd)
        monkeypatch.setattr(TF, "LAYER_BWD", layer_bwd)
        torch.manual_seed(5)
        model = ModelOnePassTransformerWithDiffusion(6, 120, 100, dropout=p).to(DEV)
        params = list(model.parameters())
        flat = torch.zeros(sum(q.numel() for q in params), device=DEV)
        off = 0
        for q in params:
            q.grad = flat[off:off + q.numel()].view_as(q)
            off += q.numel()
        opt = train.FlatClipAdam(params, flat, 1e-3)
        stash, dst, patched = train.enable_shadow_linears(model, opt, params, flat)
        assert TF.patch_model(model, seed=3, direct_param_grads=True)
        model.train()
        g = torch.Generator().manual_seed(6)
        x = torch.randn(B, 6, 120, generator=g).to(DEV)
        w = torch.randn(B, 100, generator=g).to(DEV) / B
        torch.manual_seed(11)                                   # the same diffusion steps / noise in both runs
        train._WGRAD_QUEUE = []
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(x)
            (out.float() * w).sum().backward()
        train.flush_wgrad_queue(torch.device(DEV))
        train._WGRAD_QUEUE = None
        live = [(dd, ss) for dd, ss in zip(dst, stash) if ss is not None]
        if live:
            torch._foreach_copy_([a for a, _ in live], [b for _, b in live])
        torch.cuda.synchronize()
        res = out.float().clone(), {n: q.grad.clone() for n, q in model.named_parameters()}
        train.disable_shadow_linears(patched)
        TF.unpatch_model(model)
        return res

    out0, g0 = run(False)
    for fwd_bwd in ((True, False), (True, True)):          # one-launch forward with the eight backward launches / with the one-launch backward
        out1, g1 = run(*fwd_bwd)
        assert _rel(out1, out0) < 1e-2
        for n in g0:
            assert _rel(g1[n], g0[n]) < 2e-2 or float(g0[n].norm()) < 1e-6, (fwd_bwd, n, _rel(g1[n], g0[n]))


@pytest.mark.parametrize("B,p,layers", [(512, 0.1, 2), (37, 0.1, 3), (3, 0.0, 2), (2, 0.0, 2), (1, 0.1, 3)])
def test_two_layers_in_one_launch_equal_the_two_launches(monkeypatch, B, p, layers):
    """r04: two consecutive encoder layers' forward passes as ONE launch (ops_tfd_encoder_layer_pair_fwd: a workgroup runs the second layer
    on the rows it has just written as the first one's output) against one launch per layer: the same code on the same values -- outputs
    bit-equal, and so is everything saved for the backward pass (the gradients differ only by their float atomics); likewise the two
    backward passes (ops_tfd_encoder_layer_pair_bwd: the later layer's launch waits for its predecessor's call).  Three layers: a pair
    and a single launch.  B = 2 / B = 1 (T = 14 / 7 rows: fewer than the split-row threshold of 16): the weight-gradient products are the
    library's and run at once, so the later layer's backward must NOT wait (it read buffers nobody had written yet)."""
    from openpystruct_amd import tfd_fused as TF, train
    from openpystruct_amd.surrogates import ModelOnePassTransformerWithDiffusion

    def run(pair, pair_bwd=False):
        monkeypatch.setattr(TF, "LAYER_PAIR_FWD", pair)
        monkeypatch.setattr(TF, "LAYER_PAIR_BWD", pair_bwd)
        torch.manual_seed(5)
        model = ModelOnePassTransformerWithDiffusion(6, 120, 100, num_transformer_layers=layers, dropout=p).to(DEV)
        params = list(model.parameters())
        flat = torch.zeros(sum(q.numel() for q in params), device=DEV)
        off = 0
        for q in params:
            q.grad = flat[off:off + q.numel()].view_as(q)
            off += q.numel()
        opt = train.FlatClipAdam(params, flat, 1e-3)
        stash, dst, patched = train.enable_shadow_linears(model, opt, params, flat)
        assert TF.patch_model(model, seed=3, direct_param_grads=True)
        model.train()
        g = torch.Generator().manual_seed(6)
        x = torch.randn(B, 6, 120, generator=g).to(DEV)
        w = torch.randn(B, 100, generator=g).to(DEV) / B
        torch.manual_seed(11)
        train._WGRAD_QUEUE = []
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(x)
            (out.float() * w).sum().backward()
        train.flush_wgrad_queue(torch.device(DEV))
        train._WGRAD_QUEUE = None
        live = [(dd, ss) for dd, ss in zip(dst, stash) if ss is not None]
        if live:
            torch._foreach_copy_([a for a, _ in live], [b for _, b in live])
        torch.cuda.synchronize()
        assert TF._PENDING_LAYER is None and TF._PENDING_BWD is None
        res = out.float().clone(), flat.clone()
        train.disable_shadow_linears(patched)
        TF.unpatch_model(model)
        return res

    o0, g0 = run(False)
    for pair_bwd in (False, True):                 # ... and the two backward passes as one launch too
        o1, g1 = run(True, pair_bwd)
        assert torch.equal(o0, o1) and float(o0.abs().max()) > 0
        assert float((g0 - g1).norm() / g0.norm()) < 1e-5


def test_pair_launches_with_several_workgroups_per_cu(monkeypatch):
    """The pair launches re-read rows they wrote in the same launch (device-scope loads: rows are not multiples of the 128-byte cache line, so
    a workgroup's first / last line also holds a neighbour's bytes, and a neighbour on the same CU could leave a stale copy in the vector L1),
    and the first layer's launch is DEFERRED to its successor's call: in a no-grad pass nothing but the pending record keeps the first
    layer's buffers alive until then (without it the caching allocator handed their memory to the second layer's buffers: wrong evaluation
    outputs, caught by test_evaluation_pass_through_the_one_launch_kernels_equals_the_module in full-suite order).  Evaluation-mode
    forwards at several batch sizes with allocator churn in between: pair launch bit-equal to one launch per layer, every time."""
    from openpystruct_amd import tfd_fused as TF, train
    from openpystruct_amd.surrogates import ModelOnePassTransformerWithDiffusion
    torch.manual_seed(5)
    model = ModelOnePassTransformerWithDiffusion(6, 120, 100, dropout=0.1).to(DEV)
    params = list(model.parameters())
    flat = torch.zeros(sum(q.numel() for q in params), device=DEV)
    opt = train.FlatClipAdam(params, flat, 1e-3)
    stash, dst, patched = train.enable_shadow_linears(model, opt, params, flat)
    assert TF.patch_model(model, seed=3, direct_param_grads=True)
    model.diffusion._acp.fill_(1.0)                       # (x_noisy = x: the two passes see the same input whatever is drawn)
    model.eval()
    g = torch.Generator().manual_seed(6)
    keep = []
    bad = 0
    for rep in range(6):
        for B in (40, 100, 200, 300, 77):
            x = torch.randn(B, 6, 120, generator=g).to(DEV)
            outs = []
            for pair in (False, True):
                monkeypatch.setattr(TF, "LAYER_PAIR_FWD", pair)
                with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                    outs.append(model(x).float().clone())
            bad += int(not torch.equal(outs[0], outs[1]))
            keep.append(torch.empty(1000 * (rep + 1) + 8 * B, device=DEV))      # allocator churn: the next pass finds other free blocks
            if rep % 2:
                keep.pop(0)
    train.disable_shadow_linears(patched)
    TF.unpatch_model(model)
    assert bad == 0, bad


@pytest.mark.parametrize("B,p,alpha0,second", [(512, 0.1, 0.5, False), (288, 0.0, None, False), (37, 0.3, 0.5, False), (512, 0.1, 0.5, True), (37, 0.0, None, True)])
def test_loss_on_the_head_tile_equals_the_loss_launch(monkeypatch, B, p, alpha0, second):
    """r04: the training loss of the fast path computed by the head's forward launch on its output tile and finished by the head's backward
    launch (tfd_fused.arm_head_loss) against the same step with the loss as its own two launches behind the head (surrogates.fused_loss):
    the same arithmetic on the same bf16 predictions -- value to float32 round-off (per-workgroup partial sums in another order), the
    parameter gradients to the order of the float atomics that accumulate them (two runs of ONE path differ as much), the running sum
    advanced.  `second`: another term hangs on the predictions (as the FE-residual one does): its bfloat16 gradient reaches the head's
    backward launch as `g2`, summed there with the launch's own (no addition node), and the output layer's weight gradient sees the sum."""
    from openpystruct_amd import tfd_fused as TF, train
    from openpystruct_amd.surrogates import ModelOnePassTransformerWithDiffusion, TrainableL1L2Loss, fused_loss

    def run(on_tile):
        torch.manual_seed(5)
        model = ModelOnePassTransformerWithDiffusion(6, 120, 100, dropout=p).to(DEV)
        crit = TrainableL1L2Loss(0.5, torch.tensor(-1.0, device=DEV), torch.tensor(0.4, device=DEV), 0.5).to(DEV)
        params = list(model.parameters())
        flat = torch.zeros(sum(q.numel() for q in params), device=DEV)
        off = 0
        for q in params:
            q.grad = flat[off:off + q.numel()].view_as(q)
            off += q.numel()
        opt = train.FlatClipAdam(params, flat, 1e-3)
        stash, dst, patched = train.enable_shadow_linears(model, opt, params, flat)
        assert TF.patch_model(model, seed=3, direct_param_grads=True)
        model.train()
        g = torch.Generator().manual_seed(6)
        x = torch.randn(B, 6, 120, generator=g).to(DEV)
        y = torch.randn(B, 100, generator=g).to(DEV)
        acc = torch.full((), 2.0, device=DEV)
        train._WGRAD_QUEUE = []
        with torch.autocast("cuda", dtype=torch.bfloat16):
            if on_tile:
                assert TF.arm_head_loss(y, crit, alpha0, acc)
                out = model(x)
                loss = TF.take_head_loss()
                assert loss is not None and TF.take_head_loss() is None
            else:
                out = model(x)
                loss = fused_loss(crit, out, y, alpha0=alpha0, unit_grad=True, acc=acc)
            term = None
            if second:
                term = 0.03 * (out * torch.linspace(-1.0, 2.0, 100, device=DEV).to(out.dtype)).sum().float()
                # (the tile's loss value exists once the head's BACKWARD launch has run: the terms are tied, not added, as train.py ties them)
                loss = train._TieTerms.apply(loss, term) if on_tile else loss + term
        loss.backward(gradient=torch.ones((), device=DEV))
        train.flush_wgrad_queue(torch.device(DEV))
        train._WGRAD_QUEUE = None
        live = [(dd, ss) for dd, ss in zip(dst, stash) if ss is not None]
        if live:
            torch._foreach_copy_([a for a, _ in live], [b for _, b in live])
        torch.cuda.synchronize()
        res = float(loss) + (float(term) if (second and on_tile) else 0.0), float(acc), out.float().clone(), flat.clone()
        train.disable_shadow_linears(patched)
        TF.unpatch_model(model)
        return res

    l0, a0, o0, g0 = run(False)
    l1, a1, o1, g1 = run(True)
    assert torch.equal(o0, o1)
    assert abs(l1 - l0) <= 2e-6 * abs(l0) and abs(a1 - a0) <= 4e-6 * abs(a0)
    if not second:
        assert abs((a1 - 2.0) - l1) <= 1e-6 * abs(l1)
    assert float((g0 - g1).norm() / g0.norm()) < 1e-5 and float(g0.abs().max()) > 0


def test_four_layer_encoder_keeps_training_on_fresh_weights(monkeypatch):
    """num_transformer_layers = 4: one optimiser launch refreshes 16 tiled weight copies, i.e. three layers + the head's four.  The
    fourth layer must then NOT take the one-launch form (r03: it kept the tiles of patch_encoder's first enable_layer_tiles call, which
    nothing refreshed after patch_model's second call replaced the entry list, and trained on its initial weights).  Three Adam steps at
    lr 1e-2 through the one-launch kernels against the same steps through the separate launches + library products (which read the
    optimiser's bf16 shadow): same dropout streams, so the fourth step's outputs agree to bf16 accumulation order."""
    from openpystruct_amd import tfd_fused as TF, train
    from openpystruct_amd.surrogates import ModelOnePassTransformerWithDiffusion
    B = 64

    def run(layer_fwd):
        monkeypatch.setattr(TF, "LAYER_FWD", layer_fwd)
        monkeypatch.setattr(TF, "LAYER_BWD", layer_fwd)
        torch.manual_seed(5)
        model = ModelOnePassTransformerWithDiffusion(6, 120, 100, num_transformer_layers=4, dropout=0.1).to(DEV)
        params = list(model.parameters())
        flat = torch.zeros(sum(q.numel() for q in params), device=DEV)
        off = 0
        for q in params:
            q.grad = flat[off:off + q.numel()].view_as(q)
            off += q.numel()
        opt = train.FlatClipAdam(params, flat, 1e-2)
        stash, dst, patched = train.enable_shadow_linears(model, opt, params, flat)
        assert TF.patch_model(model, seed=3, direct_param_grads=True)
        opt.repack = getattr(model.transformer_encoder, "_ops_tile_entries", None)
        if layer_fwd:
            tiled = [hasattr(l, "_ops_tiles") for l in model.transformer_encoder.layers]
            assert opt.repack is not None and tiled == [True, True, True, False], tiled
        model.train()
        g = torch.Generator().manual_seed(6)
        x = torch.randn(B, 6, 120, generator=g).to(DEV)
        w = torch.randn(B, 100, generator=g).to(DEV) / B
        outs = []
        for step in range(4):
            flat.zero_()
            torch.manual_seed(11 + step)
            train._WGRAD_QUEUE = []
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = model(x)
                (out.float() * w).sum().backward()
            train.flush_wgrad_queue(torch.device(DEV))
            train._WGRAD_QUEUE = None
            live = [(dd, ss) for dd, ss in zip(dst, stash) if ss is not None]
            if live:
                torch._foreach_copy_([a for a, _ in live], [b for _, b in live])
            outs.append(out.float().clone())
            opt.step()
        torch.cuda.synchronize()
        train.disable_shadow_linears(patched)
        TF.unpatch_model(model)
        return outs

    o0 = run(False)
    o1 = run(True)
    assert _rel(o0[3], o0[0]) > 0.2                 # three steps at lr 1e-2 moved the outputs: stale weights would show
    for a, b in zip(o1, o0):
        assert _rel(a, b) < 3e-2, [_rel(a, b) for a, b in zip(o1, o0)]


def _tfd_both_ways(monkeypatch, d, cfg, seed, epochs, **kw):
    """The TFD loop with the one-launch encoder / head / front end and through the framework's modules: same initial weights, same
    batch order per epoch."""
    from openpystruct_amd import tfd_fused, train
    n_tr, order = int(d.X_train.shape[0]), {}

    def batch_order(epoch):
        if epoch not in order:
            order[epoch] = torch.randperm(n_tr, generator=torch.Generator().manual_seed(1000 * seed + epoch))
        return order[epoch]

    hist = {}
    for fast in (True, False):
        monkeypatch.setattr(tfd_fused, "ENABLED", fast)
        out = train.train_surrogate("tfd", d, cfg, device="cuda", max_epochs=epochs, seed=seed, batch_order=batch_order, **kw)
        hist[fast] = out["history"]
        assert all(np.isfinite(hist[fast]["train"])) and all(np.isfinite(hist[fast]["val"]))
        assert hist[fast]["train"][-1] < 0.95 * hist[fast]["train"][0]
    return hist


@pytest.fixture(scope="module")
def tfd_data():
    from openpystruct_amd import dataprep, sizing
    rec = sizing.generate_dataset(6000, sizing.SizingConfig(max_e=60), "cuda")
    return dataprep.prepare(rec, kind="tfd", device="cuda")


@pytest.mark.parametrize("seed", [1, 2])
def test_tfd_training_paths_agree_epoch_for_epoch_without_randomness(monkeypatch, tfd_data, seed):
    """Dropout 0, input noise 0, the diffusion schedule's alpha_cumprod set to 1 (x_noisy = x whatever is drawn), one batch order: the
    fast path and the framework path are two bf16 evaluations of the same six epochs (two steps each: a full batch and a 288-row
    tail).  Measured over 10 seeds (profiles/r04_tfd_follow_spread.log): worst epoch-wise deviation 0.04 % (training loss) / 0.08 %
    (validation loss); bound 0.5 %."""
    from openpystruct_amd import train
    cfg = train.TfdConfig()
    cfg.dropout_rate, cfg.sigma_0 = 0.0, 0.0
    hist = _tfd_both_ways(monkeypatch, tfd_data, cfg, seed, 6, init_fn=lambda m: m.diffusion._acp.fill_(1.0))
    for key in ("train", "val"):
        dev = np.abs(np.array(hist[True][key]) / np.array(hist[False][key]) - 1.0)
        assert dev.max() < 5e-3, (key, dev)


def test_front_end_assembling_its_own_batch_reproduces_the_assembly_launch(monkeypatch, tfd_data):
    """r04: the front-end launch gathers rows order[cursor ..] of the training set, adds the assembly's input noise (same stream: seed, step
    counter, element index) and advances counter and cursor itself; the head's loss reads its targets through the gathered rows -- against
    the same run with the batch assembly as a launch of its own per step (switch tfd_front_gather = 0's path).  Same draws everywhere, so
    the loss histories agree to the order of the float atomics that accumulate the gradients (dropout and noise ON: any slip in a counter
    or an index would change every mask)."""
    from openpystruct_amd import tfd_fused, train
    n_tr, order = int(tfd_data.X_train.shape[0]), {}

    def batch_order(epoch):
        if epoch not in order:
            order[epoch] = torch.randperm(n_tr, generator=torch.Generator().manual_seed(77 + epoch))
        return order[epoch]

    hist = {}
    for fused in (True, False):
        monkeypatch.setattr(tfd_fused, "FRONT_GATHER", fused)
        out = train.train_surrogate("tfd", tfd_data, device="cuda", max_epochs=4, seed=3, batch_order=batch_order)
        hist[fused] = out["history"]
        assert tfd_fused._GATHER is None                     # disarmed at the end of the run
    for key in ("train", "val"):
        a, b = np.array(hist[True][key]), np.array(hist[False][key])
        assert np.all(np.isfinite(a)) and np.abs(a / b - 1.0).max() < 2e-3, (key, a, b)
    assert hist[True]["train"][-1] < 0.97 * hist[True]["train"][0]


def test_validation_as_one_forward_gives_the_per_batch_validation_loss(monkeypatch, tfd_data):
    """r04: the fast path's validation pass = one forward over all validation rows + the loss per reference batch on row slices, against
    the same run evaluating batch by batch (switch val_whole = 0).  No randomness (dropout 0, noise 0, alpha_cumprod 1), batch 32 so that
    the validation set is several batches and a ragged last one, and learning rate 0 so that both runs evaluate the SAME weights (two
    trainings drift apart by the order of their float atomics: up to 1 % after three epochs): rows are independent in evaluation mode and
    each batch's loss is the same arithmetic on the same bf16 predictions -- the validation losses agree to float32 round-off."""
    from openpystruct_amd import train
    cfg = train.TfdConfig()
    cfg.dropout_rate, cfg.sigma_0, cfg.batch_size, cfg.learning_rate = 0.0, 0.0, 32, 0.0
    order = {}

    def batch_order(epoch):
        if epoch not in order:
            order[epoch] = torch.randperm(int(tfd_data.X_train.shape[0]), generator=torch.Generator().manual_seed(5 + epoch))
        return order[epoch]

    assert int(tfd_data.X_val.shape[0]) > 2 * 32 and int(tfd_data.X_val.shape[0]) % 32 != 0
    hist = {}
    for whole in ("1", "0"):
        from openpystruct_amd import switches
        monkeypatch.setitem(switches._values, "val_whole", whole)
        out = train.train_surrogate("tfd", tfd_data, cfg, device="cuda", max_epochs=2, seed=4, batch_order=batch_order,
                                    init_fn=lambda m: m.diffusion._acp.fill_(1.0))
        hist[whole] = out["history"]
    a, b = np.array(hist["1"]["val"]), np.array(hist["0"]["val"])
    assert np.all(np.isfinite(a)) and np.all(a > 0) and np.abs(a / b - 1.0).max() < 2e-6, (a, b)
    assert abs(a[1] / a[0] - 1.0) < 2e-6                        # (nothing was learned: the two epochs evaluate the same weights)


@pytest.mark.stochastic
def test_tfd_training_paths_draw_from_the_same_process(monkeypatch, tfd_data):
    """The reference's configuration (dropout 0.1, diffusion + input noise on), different random streams on the two paths: three seeds,
    initial weights and batch order shared per seed.  Measured over 10 seeds (profiles/r04_tfd_follow_spread.log): ratio of the final
    training losses 1 - 0.0001 on average, at most 0.27 % off; bounds 1 % on the mean, 2 % on every seed."""
    from openpystruct_amd import train
    r = []
    for seed in (1, 2, 3):
        hist = _tfd_both_ways(monkeypatch, tfd_data, train.TfdConfig(), seed, 6)
        r.append(hist[True]["train"][-1] / hist[False]["train"][-1])
    r = np.array(r)
    assert abs(r.mean() - 1.0) < 0.01 and np.abs(r - 1.0).max() < 0.02, r


@pytest.mark.parametrize("T,N,K", [(3584, 360, 120), (3584, 120, 256), (1000, 302, 175), (257, 7, 33)])
def test_split_row_weight_gradient(T, N, K):
    from openpystruct_amd import _cabi
    lib = _cabi.load()
    g = torch.Generator().manual_seed(T + N)
    dY = torch.randn(T, N, generator=g).to(torch.bfloat16).to(DEV)
    X = torch.randn(T, K, generator=g).to(torch.bfloat16).to(DEV)
    dW = torch.full((N, K), 0.5, device=DEV)                            # accumulates on top of what is there
    db = torch.full((N,), -0.25, device=DEV)
    assert lib.ops_linear_wgrad_accumulate(T, N, K, dY.data_ptr(), X.data_ptr(), dW.data_ptr(), db.data_ptr(), None) == _cabi.OK
    torch.cuda.synchronize()
    assert _rel(dW, dY.double().t() @ X.double() + 0.5) < 1e-5
    assert _rel(db, dY.double().sum(0) - 0.25) < 1e-5
    dW.zero_()
    assert lib.ops_linear_wgrad_accumulate(T, N, K, dY.data_ptr(), X.data_ptr(), dW.data_ptr(), None, None) == _cabi.OK
    assert _rel(dW, dY.double().t() @ X.double()) < 1e-5


def test_grouped_split_row_weight_gradients():
    from openpystruct_amd import _cabi
    lib = _cabi.load()
    g = torch.Generator().manual_seed(11)
    shapes = [(3584, 360, 120), (3584, 120, 256), (512, 100, 256), (1000, 302, 175), (700, 7, 33)]
    ops, arr = [], (_cabi.WgradProblem * len(shapes))()
    for e, (T, N, K) in zip(arr, shapes):
        dY = torch.randn(T, N, generator=g).to(torch.bfloat16).to(DEV)
        X = torch.randn(T, K, generator=g).to(torch.bfloat16).to(DEV)
        dW, db = torch.zeros(N, K, device=DEV), torch.zeros(N, device=DEV)
        ops.append((dY, X, dW, db))
        e.T, e.N, e.K, e.dY, e.X, e.dW, e.dbias = T, N, K, dY.data_ptr(), X.data_ptr(), dW.data_ptr(), db.data_ptr()
    arr[3].dbias = None                                                 # one product without a bias
    assert lib.ops_linear_wgrad_accumulate_group(len(shapes), arr, None) == _cabi.OK
    torch.cuda.synchronize()
    for i, (dY, X, dW, db) in enumerate(ops):
        assert _rel(dW, dY.double().t() @ X.double()) < 1e-5
        if i != 3:
            assert _rel(db, dY.double().sum(0)) < 1e-5
        else:
            assert float(db.abs().max()) == 0.0
    assert lib.ops_linear_wgrad_accumulate_group(17, arr, None) == _cabi.ERR_INVALID_ARG


@pytest.mark.parametrize("B,cfg", [(512, None), (37, None),
                                   (50, dict(n_cases=3, feat_dim=64, n_elem=12, hidden_units=72, num_heads=4, dim_feedforward=136, diffusion_hidden_dim=80)),
                                   (21, dict(n_cases=7, feat_dim=96, n_elem=40, hidden_units=128, num_heads=8, dim_feedforward=256, diffusion_hidden_dim=64,
                                             num_transformer_layers=3))])
def test_fast_path_gradients_against_float64_autograd(monkeypatch, B, cfg):
    """The TFD fast path (fused front end, one-launch encoder layers forward AND backward, fused head, grouped weight gradients) against
    float64 autograd of the same module: dropout 0, the step indices / noise the front-end launch drew replayed to the module, a smooth
    (linear) objective -- nothing is left but bf16 rounding and ReLU branches taken on bf16-rounded pre-activations.  Bounds from the
    measured worst cases (profiles/r03_notes.md 6): every parameter <= 3e-2 relative L2 (zero-gradient parameters against their layer's
    scale)."""
    from openpystruct_amd import tfd_fused as TF, train
    from openpystruct_amd.surrogates import ModelOnePassTransformerWithDiffusion
    monkeypatch.setattr(TF, "KEEP_DRAWS", True)
    torch.manual_seed(5)
    cfg = cfg or dict(n_cases=6, feat_dim=120, n_elem=100)         # default: the reference's sizes; the others: other tile counts / head widths
    model = ModelOnePassTransformerWithDiffusion(dropout=0.0, **cfg).to(DEV)
    ref = copy.deepcopy(model).double()
    params = list(model.parameters())
    flat = torch.zeros(sum(q.numel() for q in params), device=DEV)
    off = 0
    for q in params:
        q.grad = flat[off:off + q.numel()].view_as(q)
        off += q.numel()
    opt = train.FlatClipAdam(params, flat, 1e-3)
    stash, dst, patched = train.enable_shadow_linears(model, opt, params, flat)
    assert TF.patch_model(model, seed=3, direct_param_grads=True)
    st_probe = TF._State(torch.device(DEV), 1, True)
    assert TF._front_fused_ok(model, st_probe, cfg["feat_dim"]) and TF._head_fused_ok(model, st_probe, cfg["feat_dim"])     # the one-launch blocks are what runs
    assert all(TF._layer_fused_ok(l, st_probe) for l in model.transformer_encoder.layers[:2])
    g = torch.Generator().manual_seed(6)
    x = torch.randn(B, cfg["n_cases"], cfg["feat_dim"], generator=g).to(DEV)
    w = torch.randn(B, cfg["n_elem"], generator=g).to(DEV) / B
    model.train(); ref.train()
    train._WGRAD_QUEUE = []
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(x)
        (out.float() * w).sum().backward()
    train.flush_wgrad_queue(torch.device(DEV))
    train._WGRAD_QUEUE = None
    live = [(dd, ss) for dd, ss in zip(dst, stash) if ss is not None]
    if live:
        torch._foreach_copy_([a for a, _ in live], [b for _, b in live])
    t_k, e_k = model.transformer_encoder._ops_dropout_state[x.device].draws
    monkeypatch.setattr(torch, "randint", lambda lo, hi, size, device=None, **kw: t_k)
    monkeypatch.setattr(torch, "randn_like", lambda t, **kw: e_k.to(t.dtype))
    outr = ref(x.double())
    (outr * w.double()).sum().backward()
    assert _rel(out.float(), outr) < 1e-2
    gref = {n: q.grad for n, q in ref.named_parameters()}
    worst = {}
    for n, q in model.named_parameters():
        scale = float(gref[n].norm())
        if n.endswith("bias"):                     # a bias in front of a mean-subtracting LayerNorm has (almost) no gradient: its weight's scale
            wn = n[:-4] + "weight"
            if wn in gref:
                scale = max(scale, float(gref[wn].norm()) / np.sqrt(gref[wn].shape[-1]))
        worst[n] = float((q.grad.double() - gref[n]).norm()) / (scale + 1e-30)
    # the module itself under bf16 autocast (framework encoder, same draws): how far two bf16 evaluations of this network are from float64
    train.disable_shadow_linears(patched)
    TF.unpatch_model(model)
    auto = copy.deepcopy(ref).float()
    for q in auto.parameters():
        q.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        outa = auto(x)
        (outa.float() * w).sum().backward()
    e_auto = {}
    for n, q in auto.named_parameters():
        scale = float(gref[n].norm())
        if n.endswith("bias"):
            wn = n[:-4] + "weight"
            if wn in gref:
                scale = max(scale, float(gref[wn].norm()) / np.sqrt(gref[wn].shape[-1]))
        e_auto[n] = float((q.grad.double() - gref[n]).norm()) / (scale + 1e-30)
    if os.environ.get("OPS_AMD_PRINT_GRAD_TABLE"):
        for n in worst:
            print("%-55s fast %.4f   autocast module %.4f" % (n, worst[n], e_auto[n]))
    # bound: every parameter within 3e-2 of float64 OR no worse than 1.25 x the framework's own bf16 evaluation + 1e-2
    bad = {n: (e, e_auto[n]) for n, e in worst.items() if not (np.isfinite(e) and (e <= 3e-2 or e <= 1.25 * e_auto[n] + 1e-2))}
    assert not bad, bad


@pytest.mark.parametrize("B", [512, 37])
def test_fast_path_gradients_of_a_smooth_network_match_float64(monkeypatch, B):
    """The absolute guard of the fast path's backward launches: with every ReLU replaced by the identity
    (`identity_act`, verification only), dropout 0 and a linear objective the network is smooth -- no branch can be taken on the other
    side of a bf16 rounding -- and every parameter gradient of the one-launch kernels must agree with float64 autograd of the same
    module to bf16 rounding: <= 3 % relative L2 (zero-gradient biases against their weight's scale), no reference to how well the
    framework's own bf16 evaluation does."""
    from openpystruct_amd import tfd_fused as TF, train
    from openpystruct_amd.surrogates import ModelOnePassTransformerWithDiffusion
    monkeypatch.setattr(TF, "KEEP_DRAWS", True)
    monkeypatch.setattr(TF, "IDENTITY_ACT", True)
    torch.manual_seed(5)
    cfg = dict(n_cases=6, feat_dim=120, n_elem=100)
    model = ModelOnePassTransformerWithDiffusion(dropout=0.0, **cfg).to(DEV)
    ref = copy.deepcopy(model).double()
    for layer in ref.transformer_encoder.layers:             # the same smooth network in float64
        layer.activation = lambda t: t
        layer.activation_relu_or_gelu = 0
    ref.diffusion.mlp[1] = torch.nn.Identity()
    params = list(model.parameters())
    flat = torch.zeros(sum(q.numel() for q in params), device=DEV)
    off = 0
    for q in params:
        q.grad = flat[off:off + q.numel()].view_as(q)
        off += q.numel()
    opt = train.FlatClipAdam(params, flat, 1e-3)
    stash, dst, patched = train.enable_shadow_linears(model, opt, params, flat)
    assert TF.patch_model(model, seed=3, direct_param_grads=True)
    st_probe = TF._State(torch.device(DEV), 1, True)
    assert TF._front_fused_ok(model, st_probe, 120) and TF._head_fused_ok(model, st_probe, 120)
    assert all(TF._layer_fused_ok(l, st_probe) for l in model.transformer_encoder.layers)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(B, 6, 120, generator=g).to(DEV)
    w = torch.randn(B, 100, generator=g).to(DEV) / B
    model.train(); ref.train()
    train._WGRAD_QUEUE = []
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(x)
        (out.float() * w).sum().backward()
    train.flush_wgrad_queue(torch.device(DEV))
    train._WGRAD_QUEUE = None
    live = [(dd, ss) for dd, ss in zip(dst, stash) if ss is not None]
    if live:
        torch._foreach_copy_([a for a, _ in live], [b for _, b in live])
    t_k, e_k = model.transformer_encoder._ops_dropout_state[x.device].draws
    monkeypatch.setattr(torch, "randint", lambda lo, hi, size, device=None, **kw: t_k)
    monkeypatch.setattr(torch, "randn_like", lambda t, **kw: e_k.to(t.dtype))
    monkeypatch.setattr(torch, "relu", lambda t: t)          # the head's ReLU (TFD:573) of the reference module
    outr = ref(x.double())
    (outr * w.double()).sum().backward()
    monkeypatch.undo()
    assert _rel(out.float(), outr) < 1e-2
    gref = {n: q.grad for n, q in ref.named_parameters()}
    worst = {}
    for n, q in model.named_parameters():
        scale = float(gref[n].norm())
        if n.endswith("bias"):
            wn = n[:-4] + "weight"
            if wn in gref:
                scale = max(scale, float(gref[wn].norm()) / np.sqrt(gref[wn].shape[-1]))
        worst[n] = float((q.grad.double() - gref[n]).norm()) / (scale + 1e-30)
    train.disable_shadow_linears(patched)
    TF.unpatch_model(model)
    bad = {n: e for n, e in worst.items() if not (np.isfinite(e) and e <= 3e-2)}
    assert not bad, (bad, max(worst.values()))


def _tiled_pair(lib, W):
    from openpystruct_amd import _cabi
    N, K = W.shape
    ru = lambda v, m: (v + m - 1) // m * m      # noqa: E731
    wp = torch.zeros(ru(N, 16), ru(K, 32), dtype=torch.bfloat16, device=W.device)
    wtp = torch.zeros(ru(K, 16), ru(N, 32), dtype=torch.bfloat16, device=W.device)
    ent = (_cabi.MlpRepackEntry * 1)()
    ent[0].W, ent[0].N, ent[0].K, ent[0].Wp, ent[0].ldw, ent[0].Wtp, ent[0].ldwt = W.data_ptr(), N, K, wp.data_ptr(), wp.shape[1], wtp.data_ptr(), wtp.shape[1]
    assert lib.ops_mlp_repack_weights(1, ent, torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    return wp, wtp


@pytest.mark.parametrize("B,S,d,hid,C,p", [(512, 7, 120, 256, 100, 0.0), (37, 7, 120, 256, 100, 0.3), (5, 3, 64, 72, 12, 0.0)])
def test_one_launch_head_against_the_framework_ops(B, S, d, hid, C, p):
    """csrc/seq_layer.hip tfd_head_fwd / _bwd (fc1 -> LayerNorm -> ReLU -> dropout -> fc2 on the [CLS] rows, TFD:568-575) against the
    same chain of framework ops with the launch's own dropout mask (read off h): outputs, every saved tensor, the gradient of the [CLS]
    rows (and zeros everywhere else), d_a, gamma / beta gradients; with p > 0: keep fraction, another mask for another counter value."""
    import ctypes
    from openpystruct_amd import _cabi
    lib = _cabi.load()
    dev = torch.device(DEV)
    g = torch.Generator().manual_seed(B + hid)
    y16 = torch.randn(B * S, d, generator=g).to(torch.bfloat16).to(dev)
    W1, W2 = (torch.randn(hid, d, generator=g) * 0.1).to(dev), (torch.randn(C, hid, generator=g) * 0.1).to(dev)
    b1, b2 = (torch.randn(hid, generator=g) * 0.1).to(torch.bfloat16).to(dev), (torch.randn(C, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    gamma, beta = (1 + 0.1 * torch.randn(hid, generator=g)).to(dev), (0.1 * torch.randn(hid, generator=g)).to(dev)
    (W1p, W1t), (W2p, W2t) = _tiled_pair(lib, W1), _tiled_pair(lib, W2)
    cnt, used = torch.zeros(2, dtype=torch.int64, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)
    bf = dict(dtype=torch.bfloat16, device=dev)
    s = torch.cuda.current_stream().cuda_stream

    def fwd():
        a16, h, out = torch.empty(B, hid, **bf), torch.empty(B, hid, **bf), torch.empty(B, C, **bf)
        mean, rstd = torch.empty(B, device=dev), torch.empty(B, device=dev)
        a = _cabi.TfdHeadArgs(B=B, S=S, d=d, hid=hid, C=C, y16=y16.data_ptr(), W1=W1p.data_ptr(), b1=b1.data_ptr(), gamma=gamma.data_ptr(),
                              beta=beta.data_ptr(), eps=1e-5, W2=W2p.data_ptr(), b2=b2.data_ptr(), p_drop=p, seed=5, counter=cnt.data_ptr(),
                              used_call=used.data_ptr(), a16=a16.data_ptr(), mean=mean.data_ptr(), rstd=rstd.data_ptr(), h=h.data_ptr(), out=out.data_ptr())
        assert lib.ops_tfd_head_fwd(ctypes.byref(a), s) == 0
        torch.cuda.synchronize()
        return a16, mean, rstd, h, out

    a16, mean, rstd, h, out = fwd()
    x = y16.view(B, S, d)[:, 0, :].float().requires_grad_()
    W1r, W2r = W1.to(torch.bfloat16).float(), W2.to(torch.bfloat16).float()
    gr, br = gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    ar = (x @ W1r.t() + b1.float())
    a_b = ar + (ar.to(torch.bfloat16).float() - ar).detach()                 # bf16 rounding, straight-through
    yr = torch.nn.functional.layer_norm(a_b, (hid,), gr, br, 1e-5)
    y_b = yr + (yr.to(torch.bfloat16).float() - yr).detach()
    keep = (h.float() != 0) | (y_b.detach() <= 0)                              # where ReLU passed, h's zeros are the dropout's
    hr = torch.relu(y_b) * keep / (1.0 - p)
    h_b = hr + (hr.to(torch.bfloat16).float() - hr).detach()
    outr = h_b @ W2r.t() + b2.float()
    assert _rel(a16.float(), a_b) < 1e-4 and _rel(mean, a_b.mean(1)) < 1e-4 and _rel(rstd, (a_b.var(1, unbiased=False) + 1e-5).rsqrt()) < 1e-4
    assert _rel(h.float(), h_b) < 3e-3 and _rel(out.float(), outr) < 4e-3
    if p > 0:
        pos = y_b.detach() > 0
        assert abs(float((h.float() != 0)[pos].float().mean()) - (1 - p)) < 0.02
        cnt[0] += 1
        h2 = fwd()[3]
        assert float(((h2.float() != 0) != (h.float() != 0))[pos].float().mean()) > 0.2
        cnt[0] -= 1
    # backward
    go = (torch.randn(B, C, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    d_a, full = torch.empty(B, hid, **bf), torch.zeros(B * S, d, **bf)
    dg, db = torch.zeros(hid, device=dev), torch.zeros(hid, device=dev)
    ab = _cabi.TfdHeadBwdArgs(B=B, S=S, d=d, hid=hid, C=C, g=go.data_ptr(), Wt2=W2t.data_ptr(), Wt1=W1t.data_ptr(), gamma=gamma.data_ptr(), p_drop=p,
                              a16=a16.data_ptr(), mean=mean.data_ptr(), rstd=rstd.data_ptr(), h=h.data_ptr(), d_a=d_a.data_ptr(), dcls_rows=full.data_ptr(),
                              dgamma=dg.data_ptr(), dbeta=db.data_ptr())
    assert lib.ops_tfd_head_bwd(ctypes.byref(ab), s) == 0
    torch.cuda.synchronize()
    a_b.retain_grad()
    (outr * go.float()).sum().backward()
    assert _rel(d_a.float(), a_b.grad) < 1.5e-2
    assert _rel(full.view(B, S, d)[:, 0, :].float(), x.grad) < 1.5e-2
    assert float(full.view(B, S, d)[:, 1:, :].float().abs().max()) == 0.0 if S > 1 else True
    assert _rel(dg, gr.grad) < 1.5e-2 and _rel(db, br.grad) < 1.5e-2


@pytest.mark.parametrize("B,Nc,d,hid", [(512, 6, 120, 256), (37, 6, 120, 256), (5, 3, 64, 72)])
def test_one_launch_front_end_against_the_framework_ops(B, Nc, d, hid):
    """csrc/seq_layer.hip tfd_front_fwd / _bwd (draws, x_noisy, the diffusion MLP, the combine with [CLS] token and positional encoding:
    TFD:443-478, :563-567) against the same chain of framework ops on the draws the launch reports; backward against autograd."""
    import ctypes
    from openpystruct_amd import _cabi
    lib = _cabi.load()
    dev = torch.device(DEV)
    S, T, rows = Nc + 1, 300, B * Nc
    g = torch.Generator().manual_seed(B + d)
    x = torch.randn(B, Nc, d, generator=g).to(dev)
    acp = torch.cumprod(1.0 - torch.linspace(1e-4, 0.02, T), 0).to(dev)
    W0, W2 = (torch.randn(hid, d, generator=g) * 0.1).to(dev), (torch.randn(d, hid, generator=g) * 0.1).to(dev)
    b0, b2 = (torch.randn(hid, generator=g) * 0.1).to(torch.bfloat16).to(dev), (torch.randn(d, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    cls, pe = torch.randn(d, generator=g).to(dev), torch.randn(16, d, generator=g).to(dev)
    (W0p, _), (W2p, W2t) = _tiled_pair(lib, W0), _tiled_pair(lib, W2)
    cnt = torch.zeros(2, dtype=torch.int64, device=dev)
    bf = dict(dtype=torch.bfloat16, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    xn16, h = torch.empty(rows, d, **bf), torch.empty(rows, hid, **bf)
    sa, sb = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    z, z16 = torch.empty(B, S, d, device=dev), torch.empty(B * S, d, **bf)
    t, eps = torch.empty(rows, dtype=torch.int64, device=dev), torch.empty(rows, d, device=dev)
    a = _cabi.TfdFrontArgs(B=B, Nc=Nc, d=d, hid=hid, T=T, x=x.data_ptr(), alpha_cumprod=acp.data_ptr(), seed=77, counter=cnt.data_ptr(), W0=W0p.data_ptr(),
                           b0=b0.data_ptr(), W2=W2p.data_ptr(), b2=b2.data_ptr(), cls=cls.data_ptr(), pe=pe.data_ptr(), xn16=xn16.data_ptr(), h=h.data_ptr(),
                           sa=sa.data_ptr(), sb=sb.data_ptr(), z=z.data_ptr(), z16=z16.data_ptr(), t_out=t.data_ptr(), eps_out=eps.data_ptr())
    assert lib.ops_tfd_front_fwd(ctypes.byref(a), s) == 0
    torch.cuda.synchronize()
    assert int(t.min()) >= 0 and int(t.max()) < T
    if rows * d > 100000:
        assert abs(float(eps.mean())) < 0.01 and abs(float(eps.std()) - 1.0) < 0.01
    torch.testing.assert_close(sa, acp[t].sqrt(), rtol=1e-6, atol=0)
    xn = sa[:, None] * x.reshape(rows, d) + sb[:, None] * eps
    assert torch.equal(xn16, xn.to(torch.bfloat16)) or _rel(xn16.float(), xn) < 3e-3
    W0r, W2r = W0.to(torch.bfloat16).float(), W2.to(torch.bfloat16).float()
    hr = torch.relu((xn16.float() @ W0r.t() + b0.float()).to(torch.bfloat16).float())
    assert _rel(h.float(), hr) < 2e-3
    mr = (h.float() @ W2r.t() + b2.float()).to(torch.bfloat16).float()
    zr = torch.empty(B, S, d, device=dev)
    zr[:, 0, :] = cls + pe[0]
    zr[:, 1:, :] = ((xn - sb[:, None] * mr) / sa[:, None]).view(B, Nc, d) + pe[1:S]
    assert _rel(z, zr) < 2e-3 and float((z[:, 0, :] - zr[:, 0, :]).abs().max()) == 0.0
    assert torch.equal(z16, z.reshape(B * S, d).to(torch.bfloat16))
    # backward: g32 and g16 together
    g32 = torch.randn(B, S, d, generator=g).to(dev)
    g16 = (torch.randn(B * S, d, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    dm, d_h, dcls = torch.empty(rows, d, **bf), torch.empty(rows, hid, **bf), torch.zeros(d, device=dev)
    ab = _cabi.TfdFrontBwdArgs(B=B, Nc=Nc, d=d, hid=hid, g32=g32.data_ptr(), g16=g16.data_ptr(), sa=sa.data_ptr(), sb=sb.data_ptr(), h=h.data_ptr(),
                               Wt2=W2t.data_ptr(), dm=dm.data_ptr(), d_h=d_h.data_ptr(), dcls=dcls.data_ptr())
    assert lib.ops_tfd_front_bwd(ctypes.byref(ab), s) == 0
    torch.cuda.synchronize()
    gt = g32 + g16.float().view(B, S, d)
    dmr = (-(sb / sa)[:, None] * gt[:, 1:, :].reshape(rows, d)).to(torch.bfloat16)
    assert torch.equal(dm, dmr) or _rel(dm.float(), dmr.float()) < 3e-3
    dhr = ((dm.float() @ W2r).to(torch.bfloat16).float() * (h.float() > 0))
    assert _rel(d_h.float(), dhr) < 3e-3
    assert _rel(dcls, gt[:, 0, :].sum(0)) < 1e-5


def test_evaluation_pass_through_the_one_launch_kernels_equals_the_module(monkeypatch):
    """model.eval() + no_grad under bf16 autocast: the patched forward runs the same launches with every dropout probability 0 and the
    draws of the front-end launch; against the module's own evaluation forward on those draws."""
    from openpystruct_amd import tfd_fused as TF, train
    from openpystruct_amd.surrogates import ModelOnePassTransformerWithDiffusion
    monkeypatch.setattr(TF, "KEEP_DRAWS", True)
    torch.manual_seed(5)
    model = ModelOnePassTransformerWithDiffusion(6, 120, 100, dropout=0.1).to(DEV)
    ref = copy.deepcopy(model)
    params = list(model.parameters())
    flat = torch.zeros(sum(q.numel() for q in params), device=DEV)
    off = 0
    for q in params:
        q.grad = flat[off:off + q.numel()].view_as(q)
        off += q.numel()
    opt = train.FlatClipAdam(params, flat, 1e-3)
    stash, dst, patched = train.enable_shadow_linears(model, opt, params, flat)
    assert TF.patch_model(model, seed=3, direct_param_grads=True)
    x = torch.randn(200, 6, 120, generator=torch.Generator().manual_seed(6)).to(DEV)
    model.eval(); ref.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(x)
        st = model.transformer_encoder._ops_dropout_state[x.device]
        t_k, e_k = st.draws
        assert t_k.shape == (200, 6) and st.train_mode
        out2 = model(x)
        assert not torch.equal(st.draws[1], e_k)                     # fresh noise for the next evaluation batch
        monkeypatch.setattr(torch, "randint", lambda lo, hi, size, device=None, **kw: t_k)
        monkeypatch.setattr(torch, "randn_like", lambda t, **kw: e_k.to(t.dtype))
        outr = ref(x)
    assert out.dtype == torch.bfloat16 and _rel(out.float(), outr.float()) < 1.5e-2 and _rel(out2.float(), out.float()) > 1e-4
    with torch.autocast("cuda", dtype=torch.bfloat16):               # with gradients enabled the evaluation pass is the module's own
        assert model(x).requires_grad
    train.disable_shadow_linears(patched)
    TF.unpatch_model(model)


def test_grouped_launch_column_sum_jobs():
    """K = 0 problems of ops_linear_wgrad_accumulate_group: out [N] += column sums of a float32 [T, N] matrix with a row stride (the
    per-workgroup LayerNorm gamma / beta partial sums of the one-launch layer backward), next to an ordinary product in the same launch."""
    from openpystruct_amd import _cabi
    lib = _cabi.load()
    g = torch.Generator().manual_seed(4)
    part = torch.randn(224, 4, 128, generator=g).to(DEV)
    dY = torch.randn(700, 72, generator=g).to(torch.bfloat16).to(DEV)
    X = torch.randn(700, 40, generator=g).to(torch.bfloat16).to(DEV)
    outs = [torch.full((120,), 0.5, device=DEV) for _ in range(4)]
    dW = torch.zeros(72, 40, device=DEV)
    arr = (_cabi.WgradProblem * 5)()
    flatp = part.view(224, 512)
    for k in range(4):
        v = flatp[:, 128 * k:128 * k + 120]
        arr[k].T, arr[k].N, arr[k].K, arr[k].dY, arr[k].dW, arr[k].ldy = 224, 120, 0, v.data_ptr(), outs[k].data_ptr(), v.stride(0)
    arr[4].T, arr[4].N, arr[4].K, arr[4].dY, arr[4].X, arr[4].dW = 700, 72, 40, dY.data_ptr(), X.data_ptr(), dW.data_ptr()
    assert lib.ops_linear_wgrad_accumulate_group(5, arr, torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    for k in range(4):
        torch.testing.assert_close(outs[k], 0.5 + part[:, k, :120].double().sum(0).float(), rtol=1e-5, atol=1e-5)
    assert _rel(dW, dY.float().t() @ X.float()) < 1e-5
    bad = (_cabi.WgradProblem * 1)()
    bad[0].T, bad[0].N, bad[0].K, bad[0].dY, bad[0].dW, bad[0].ldy = 8, 120, 0, part.data_ptr(), outs[0].data_ptr(), 64      # stride < N
    assert lib.ops_linear_wgrad_accumulate_group(1, bad, torch.cuda.current_stream().cuda_stream) == _cabi.ERR_INVALID_ARG


def test_deterministic_mode_makes_the_tfd_step_bit_reproducible():
    """Library option "deterministic" (r06): one row split per weight-gradient product, one workgroup per column-sum strip and for the [CLS] sums,
    the head's LayerNorm sums in workgroup order -- two runs of one seed then give the SAME BITS (default mode: float atomics land in arrival
    order and two runs sit on nearby trajectories, ~1e-4 apart), and the same trajectory as the default mode to that spread."""
    from openpystruct_amd import _cabi, dataprep, sizing, train
    rec = sizing.generate_dataset(1800, sizing.SizingConfig(max_e=30), "cuda", seed=11)
    d = dataprep.prepare(rec, kind="tfd", seed=0, device="cuda")
    cfg = train.TfdConfig(batch_size=64)

    def run():
        r = train.train_surrogate("tfd", d, cfg, device="cuda", max_epochs=3, seed=5)
        return np.array([float(x) for x in r["history"]["train"]] + [float(x) for x in r["history"]["val"]])
    base = run()
    try:
        _cabi.set_option("deterministic", 1)
        assert _cabi.get_option("deterministic") == 1
        a, b = run(), run()
    finally:
        _cabi.set_option("deterministic", 0)
    assert np.isfinite(a).all() and np.array_equal(a, b), np.abs(a - b).max()
    assert float(np.abs(a - base).max() / np.abs(base).max()) < 2e-3
