"""Surrogate models, losses, data prep and the data-parallel loop on CPU (gloo for the 2-rank cases)."""
import os

import numpy as np
import pytest
import torch

from openpystruct_amd import dataprep, sizing, surrogates, train


def _fake_records(S=120, seed=0):
    """A dataset with the reference's 13-field shape (fixed bridge) without running the FE generator."""
    cfg = sizing.SizingConfig()
    cases = sizing.make_cases(S, cfg, seed=seed)
    g = torch.Generator().manual_seed(seed)
    xs = cases.node_positions.numpy()
    return {
        "roller_x_locations": [[float(xs[b, n - 1]) for n in cases.roller_nodes[b]] for b in range(S)],
        "force_x_locations": [[float(xs[b, n - 1]) for n in cases.force_nodes[b]] for b in range(S)],
        "force_values": cases.force_values,
        "node_positions": cases.node_positions,
        "I_values": torch.rand(S, 100, generator=g) * 0.5 + 0.01,
        "deflections": torch.randn(S, 101, generator=g, dtype=torch.float64) * 1e-2,
        "rotations": torch.randn(S, 101, generator=g, dtype=torch.float64) * 1e-3,
    }


def test_parameter_counts_and_state_dict_names():
    pinn = surrogates.FNNWithResidual(684, 350, 2, 302)
    assert surrogates.count_parameters(pinn) == 593914
    keys = set(pinn.state_dict())
    for k in ("input_fc.weight", "input_norm.running_mean", "residual_blocks.0.0.fc1.weight", "residual_blocks.0.0.conv1.weight",
              "residual_blocks.0.0.bn1.weight", "residual_blocks.1.1.weight", "output_fc.bias"):
        assert k in keys
    tfd = surrogates.ModelOnePassTransformerWithDiffusion(6, 120, 100)
    assert surrogates.count_parameters(tfd) == 359876
    keys = set(tfd.state_dict())
    for k in ("diffusion.mlp.0.weight", "diffusion.mlp.2.bias", "pos_encoder.pe", "cls_token", "fc1.weight", "norm1.weight", "fc2.bias",
              "transformer_encoder.layers.1.self_attn.in_proj_weight", "transformer_encoder.layers.0.linear2.weight"):
        assert k in keys
    assert tuple(tfd(torch.randn(5, 6, 120)).shape) == (5, 100)
    assert tuple(pinn(torch.randn(5, 684)).shape) == (5, 302)
    with pytest.raises(AssertionError):
        tfd(torch.randn(5, 7, 120))                                        # TFD:550-551
    fnn = surrogates.FNNPlain(684, 128, 4, 100, 0.5)                       # FNN:472-478
    # 684*128+128 input, 4 x (128*128+128 fc1 + 2*128 LayerNorm), 128*100+100 output
    assert surrogates.count_parameters(fnn) == 684 * 128 + 128 + 4 * (128 * 128 + 128 + 256) + 128 * 100 + 100
    assert {"input_fc.weight", "residual_blocks.3.fc1.bias", "residual_blocks.0.norm.weight", "output_fc.weight"} <= set(fnn.state_dict())
    assert tuple(fnn(torch.randn(5, 684)).shape) == (5, 100)


def test_losses_against_hand_computed_values():
    p = torch.tensor([[0.0, 2.0, -3.0]]); t = torch.tensor([[1.0, 1.0, 1.0]])
    crit = surrogates.TrainableL1L2Loss(0.5, torch.tensor(-1.0), torch.tensor(1.5), 0.1)
    l1 = (1 + 1 + 4) / 3; l2 = (1 + 1 + 16) / 3; pen = (2.0 - 1.5) + (-1.0 + 3.0)
    assert float(crit(p, t).detach()) == pytest.approx(0.5 * l1 + 0.5 * l2 + 0.1 * pen, rel=1e-6)
    assert [n for n, _ in crit.named_parameters()] == ["alpha"]           # a Parameter the optimiser never sees (PINN:696)
    comp = surrogates.CompositeLoss(1, 1, 1, 0.5, 0.0, None, None, penalty_pinn=2.0)
    val = comp(p, t)   # I: |0-1|; deflection rel |2-1|/1; rotation rel |-3-1|/1
    assert float(val) == pytest.approx(0.5 * 1 + 0.5 * 1 + 2.0 * (1.0 + 4.0), rel=1e-6)


def test_positional_encoding_and_schedule():
    pe = surrogates.PositionalEncoding(7, max_len=16)                      # odd d_model: last column stays 0
    assert float(pe.pe[0, :, 6].abs().max()) == 0.0 and float(pe.pe[0, 0, 1]) == 1.0
    s = surrogates.DiffusionSchedule(512)
    assert float(s.beta[0]) == pytest.approx(1e-12) and float(s.beta[-1]) == pytest.approx(1e-5)
    assert 0.997 < float(s.alpha_cumprod[-1]) < 1.0


def test_scaler_matches_sklearn_and_numpy():
    from sklearn.preprocessing import StandardScaler
    rng = np.random.default_rng(0)
    a = rng.normal(size=(50, 7)).astype(np.float32); a[:, 3] = 2.5     # zero-variance column
    sk = StandardScaler().fit(a)
    mine = dataprep.StandardScalerT().fit(torch.as_tensor(a))
    np.testing.assert_allclose(mine.transform(torch.as_tensor(a)).numpy(), sk.transform(a), rtol=1e-5, atol=1e-6)
    y3 = torch.as_tensor(rng.normal(size=(9, 6, 4)).astype(np.float32))
    ref = y3.numpy().mean(axis=1) + 0.5 * y3.numpy().std(axis=1)          # PINN:79-92
    np.testing.assert_allclose(dataprep.unify_label_with_c(y3, 0.5).numpy(), ref, rtol=1e-5, atol=1e-6)
    assert tuple(dataprep.pad_sequences([[1, 2], [3]], 4).shape) == (2, 4)


def test_prepare_shapes_pinn_and_tfd():
    rec = _fake_records(120)
    d = dataprep.prepare(rec, kind="pinn", seed=1)
    assert d.X_train.shape == (16, 684) and d.Y_train.shape == (16, 302) and d.X_val.shape == (4, 684)   # 5+4+4+101 = 114; x6
    assert d.max_lengths == {"roller_x": 5, "force_x": 4, "force_values": 4, "node_positions": 101}
    assert float(d.min_constraint) == float(d.Y_train[:, :100].min())
    t = dataprep.prepare(rec, kind="tfd", seed=1)
    assert t.X_train.shape == (16, 6, 120) and t.Y_train.shape == (16, 100) and t.feat_dim == 120         # padded to 8 heads
    assert float(t.X_train[:, :, 114:].abs().max()) == 0.0
    f = dataprep.prepare(rec, kind="fnn", seed=1, c=1.0)
    assert f.X_train.shape == (16, 684) and f.Y_train.shape == (16, 100) and torch.equal(f.X_train, d.X_train)
    with pytest.raises(ValueError):
        dataprep.prepare(_fake_records(4), kind="pinn")


@pytest.mark.parametrize("kind", ["pinn", "tfd", "fnn", "gnn", "fno"])
def test_training_loop_runs_and_early_stops(kind):
    rec = _fake_records(240, seed=3)
    d = dataprep.prepare(rec, kind=kind, seed=2)
    cfg = {"pinn": train.PinnConfig, "tfd": train.TfdConfig, "fnn": train.FnnConfig, "gnn": train.GnnConfig,
           "fno": train.FnoConfig}[kind](batch_size=16, patience=2)
    out = train.train_surrogate(kind, d, cfg, device="cpu", autocast_dtype=None, max_epochs=4)
    assert 1 <= out["epochs"] <= 4 and len(out["history"]["val"]) == out["epochs"]
    assert np.isfinite(out["history"]["train"]).all() and np.isfinite(out["r2_val_I"])
    assert out["best_state"] is not None and set(out["best_state"]) == set(out["model"].state_dict())


def _ddp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=120))
    torch.manual_seed(0)
    model = surrogates.FNNWithResidual(24, 16, 1, 10, dropout_rate=0.0, use_conv=False, norm_type="layer")   # no BatchNorm: batch statistics are per rank
    crit = surrogates.CompositeLoss(4, 3, 3, 0.5, 0.0, None, None)         # mean-type terms only
    X = torch.randn(8, 24, generator=torch.Generator().manual_seed(1)); Y = torch.randn(8, 10, generator=torch.Generator().manual_seed(2))
    ddp = torch.nn.parallel.DistributedDataParallel(model, bucket_cap_mb=8)
    sl = slice(rank * 4, rank * 4 + 4)
    crit(ddp(X[sl]), Y[sl]).backward()
    g = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    # scaler moments all-reduced over the two shards == moments of the whole array
    sc = dataprep.StandardScalerT().fit(X[sl], distributed=True)
    q.put((rank, g.numpy(), sc.mean_.numpy(), sc.scale_.numpy()))
    dist.destroy_process_group()


def test_ddp_gradients_equal_single_process_on_concatenated_batch():
    from tests.helpers import run_ranks
    outs = run_ranks(_ddp_worker, 2, timeout=240)
    torch.manual_seed(0)
    model = surrogates.FNNWithResidual(24, 16, 1, 10, dropout_rate=0.0, use_conv=False, norm_type="layer")   # no BatchNorm: batch statistics are per rank
    crit = surrogates.CompositeLoss(4, 3, 3, 0.5, 0.0, None, None)
    X = torch.randn(8, 24, generator=torch.Generator().manual_seed(1)); Y = torch.randn(8, 10, generator=torch.Generator().manual_seed(2))
    crit(model(X), Y).backward()
    g = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).numpy()
    from sklearn.preprocessing import StandardScaler
    sk = StandardScaler().fit(X.numpy())
    for _, gr, mean, scale in outs:
        np.testing.assert_allclose(gr, g, rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(mean, sk.mean_, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(scale, sk.scale_, rtol=1e-5, atol=1e-6)


def _ddp_train_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=120))
    rec = _fake_records(240, seed=3)
    lo, hi = sizing.shard_range(240, rank, world)
    shard = {k: (v[lo:hi] if torch.is_tensor(v) else v[lo:hi]) for k, v in rec.items()}
    d = dataprep.prepare(shard, kind="tfd", seed=2, distributed=True)
    out = train.train_surrogate("tfd", d, train.TfdConfig(batch_size=8, patience=3), device="cpu", autocast_dtype=None, max_epochs=2)
    w = torch.cat([p.detach().reshape(-1) for p in out["model"].parameters()])
    q.put((rank, out["epochs"], out["history"]["val"], float(w.sum()), d.scalers_Y["I"].mean_.numpy()))
    dist.destroy_process_group()


def test_two_rank_training_stays_in_sync():
    from tests.helpers import run_ranks
    outs = sorted(run_ranks(_ddp_train_worker, 2, timeout=400))
    (_, e0, v0, w0, m0), (_, e1, v1, w1, m1) = outs
    assert e0 == e1 and v0 == pytest.approx(v1)              # all-reduced losses: identical early-stop decisions
    assert w0 == pytest.approx(w1, rel=1e-6)                 # replicas hold the same weights after training
    np.testing.assert_allclose(m0, m1)                       # global scaler statistics on every rank


def _ddp_train_worker_uneven(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=180))
    n = 311                                                # shards of 77 / 78 / 78 / 78 samples -> 12 / 13 / 13 / 13 groups -> 9 / 10 / 10 / 10 training groups
    rec = _fake_records(n, seed=3)
    lo, hi = sizing.shard_range(n, rank, world)
    shard = {k: (v[lo:hi] if torch.is_tensor(v) else v[lo:hi]) for k, v in rec.items()}
    d = dataprep.prepare(shard, kind="pinn", seed=2 + rank, distributed=True)
    out = train.train_surrogate("pinn", d, train.PinnConfig(batch_size=4, patience=3), device="cpu", autocast_dtype=None, max_epochs=2)
    w = torch.cat([p.detach().reshape(-1) for p in out["model"].parameters()])
    bn = torch.cat([b.detach().double().reshape(-1) for b in out["model"].buffers() if b.is_floating_point()])
    q.put((rank, out["epochs"], out["steps_per_epoch"], out["history"]["val"], float(w.double().sum()), float(bn.sum()),
           int(d.X_train.shape[0]), float(d.min_constraint)))
    dist.destroy_process_group()


def test_four_rank_training_with_uneven_shards_stays_in_sync():
    """world_size 4, shards that do not divide evenly (6 666 groups / (4 x 128) in the real run): every rank runs the same
    number of steps (the minimum over ranks), stops at the same epoch, ends with the same weights and BatchNorm buffers."""
    from tests.helpers import run_ranks
    outs = sorted(run_ranks(_ddp_train_worker_uneven, 4, timeout=600))
    sizes = [o[6] for o in outs]
    assert len(set(sizes)) > 1                              # the shards really are uneven
    assert len({o[1] for o in outs}) == 1 and len({o[2] for o in outs}) == 1
    assert outs[0][2] == 2                                  # 9 groups / batch 4: the one-row tail is dropped (BatchNorm), 10 -> 3: min = 2
    for o in outs[1:]:
        assert o[3] == pytest.approx(outs[0][3])            # all-reduced validation losses
        assert o[4] == pytest.approx(outs[0][4], rel=1e-6) and o[5] == pytest.approx(outs[0][5], rel=1e-6)
        assert o[7] == pytest.approx(outs[0][7])            # box constraint: global minimum on every rank


def test_stencil_batchnorm_path_equals_the_library_modules():
    """ResidualBlock's Conv1d(1,1,3)+BatchNorm1d(1) fast path == nn.Conv1d / nn.BatchNorm1d: outputs, gradients, buffers."""
    torch.manual_seed(0)
    conv, bn = torch.nn.Conv1d(1, 1, 3, padding=1), torch.nn.BatchNorm1d(1)
    with torch.no_grad():
        bn.weight.fill_(1.3); bn.bias.fill_(-0.2)
    conv2, bn2 = (type(m)(*a) for m, a in ((conv, (1, 1, 3)), (bn, (1,))))
    conv2 = torch.nn.Conv1d(1, 1, 3, padding=1); conv2.load_state_dict(conv.state_dict()); bn2.load_state_dict(bn.state_dict())
    for step in range(3):
        x = torch.randn(16, 35, requires_grad=True); x2 = x.detach().clone().requires_grad_(True)
        a = surrogates.conv3_bn_single_channel(x, conv, bn, True)
        b = bn2(conv2(x2.unsqueeze(1))).squeeze(1)
        assert torch.allclose(a, b, atol=1e-5)
        g = torch.randn_like(a)
        a.backward(g); b.backward(g)
        assert torch.allclose(x.grad, x2.grad, atol=1e-5)
        for p, q in zip(list(conv.parameters()) + list(bn.parameters()), list(conv2.parameters()) + list(bn2.parameters())):
            assert torch.allclose(p.grad, q.grad, atol=1e-4, rtol=1e-4)
            p.grad = None; q.grad = None
    assert torch.allclose(bn.running_mean, bn2.running_mean, atol=1e-6) and torch.allclose(bn.running_var, bn2.running_var, atol=1e-6)
    assert int(bn.num_batches_tracked) == int(bn2.num_batches_tracked) == 3
    x = torch.randn(5, 35)
    assert torch.allclose(surrogates.conv3_bn_single_channel(x, conv, bn, False), bn2.eval()(conv2(x.unsqueeze(1))).squeeze(1), atol=1e-5)


def test_chain_gnn_stencil_equals_the_dense_adjacency_product():
    """GCNLayer's two shifted copies == einsum('ij,bjd->bid', A_hat, xW) of GNN:264-286; names and sizes as the reference."""
    torch.manual_seed(0)
    g = surrogates.ChainGNN(684, 100, 128, 128, 2, 0.0).eval()
    A = g.A_hat
    assert float(A.diagonal().abs().max()) == 0.0 and float(A[0, 1]) == pytest.approx(1 / np.sqrt(2), rel=1e-6) and float(A[5, 6]) == pytest.approx(0.5, rel=1e-6)
    x = torch.randn(3, 684)
    h = g.encoder(x).view(3, 100, 128)
    for gcn, norm in zip(g.gcn_layers, g.norms):
        h = h + torch.einsum("ij,bjd->bid", A, gcn.linear(norm(h)))
    assert torch.allclose(g(x), g.out_layer(h).squeeze(-1), atol=1e-5)
    assert surrogates.count_parameters(g) == 684 * 128 + 128 + 128 * 12800 + 12800 + 2 * (128 * 128 + 256) + 129
    assert {"A_hat", "encoder.0.weight", "encoder.2.bias", "gcn_layers.1.linear.weight", "norms.0.weight", "out_layer.bias"} <= set(g.state_dict())


@pytest.mark.parametrize("n,modes", [(6, 4), (6, 3), (7, 4), (5, 2), (8, 5)])
def test_spectral_conv_closed_form_equals_the_literal_einsum_over_rfft(n, modes):
    """SpectralConv1d == rfft -> the reference's broadcasting einsum + sum(dim=2) -> zero-pad -> irfft (FNO:356-403)."""
    torch.manual_seed(n * 10 + modes)
    sc = surrogates.SpectralConv1d(8, 8, modes)
    x = torch.randn(5, 8, n)
    x_ft = torch.fft.rfft(x, n=n)[:, :, :modes]
    w_r, w_i = sc.weights_real.unsqueeze(0), sc.weights_imag.unsqueeze(0)
    o_r = (torch.einsum("bim, iojm -> bojm", x_ft.real, w_r) - torch.einsum("bim, iojm -> bojm", x_ft.imag, w_i)).sum(dim=2)
    o_i = (torch.einsum("bim, iojm -> bojm", x_ft.real, w_i) + torch.einsum("bim, iojm -> bojm", x_ft.imag, w_r)).sum(dim=2)
    lit = torch.fft.irfft(torch.nn.functional.pad(torch.complex(o_r, o_i), (0, n // 2 + 1 - modes)), n=n)
    assert torch.allclose(sc(x), lit, atol=1e-5)


def test_fno_model_shapes_names_and_prep():
    f = surrogates.FNO1dModel(6, 114, 100, 4, 128, 4, 512, 0.1).eval()
    assert tuple(f(torch.randn(4, 6, 114)).shape) == (4, 100)
    keys = set(f.state_dict())
    assert {"fc0.weight", "fno_blocks.0.conv.weights_real", "fno_blocks.3.w.weight", "fno_blocks.1.bn.running_var", "fc_out.0.weight", "fc_out.3.bias"} <= keys
    assert tuple(f.state_dict()["fno_blocks.0.w.weight"].shape) == (128, 128, 1)
    assert surrogates.count_parameters(f) == 114 * 128 + 128 + 4 * (2 * 128 * 128 * 4 + 128 * 128 + 128 + 256) + 768 * 512 + 512 + 512 * 100 + 100
    rec = _fake_records(120)
    d = dataprep.prepare(rec, kind="fno", seed=1)
    assert d.X_train.shape == (16, 6, 114) and d.feat_dim == 114 and d.Y_train.shape == (16, 100)     # no head padding (nheads 1)
    g = dataprep.prepare(rec, kind="gnn", seed=1)
    assert g.X_train.shape == (16, 684) and g.Y_train.shape == (16, 100)


def test_user_input_front_end_scales_like_training_rows():
    """dataprep.user_inputs == the training-time scaling applied to the same raw lists; predicted_inertia inverts scaler_Y."""
    rec = _fake_records(120, seed=5)
    for kind in ("pinn", "tfd", "fno"):
        d = dataprep.prepare(rec, kind=kind, seed=1)
        g0 = list(range(6))                                      # raw rows of some group
        args = [[rec["roller_x_locations"][i] for i in g0], [rec["force_x_locations"][i] for i in g0],
                [rec["force_values"][i] for i in g0], [rec["node_positions"][i].tolist() for i in g0]]
        X = dataprep.user_inputs(d, kind, *args)
        assert X.shape[0] == 1 and X.shape[1:] == d.X_train.shape[1:]
        # the same numbers by hand for the first feature of the first case
        sc = d.scalers_inputs["roller_x"]
        first = (args[0][0][0] - float(sc.mean_[0])) / float(sc.scale_[0])
        assert float(X.reshape(-1)[0]) == pytest.approx(first, rel=1e-5)
    y = d.Y_train[:3]
    back = dataprep.predicted_inertia(d, y)
    assert torch.allclose(d.scalers_Y["I"].transform(back), y[:, :100], atol=1e-5)


def _ddp_async_worker(rank, world, port, q):
    import importlib
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=180))
    n = 60 * world + 7
    rec = _fake_records(n, seed=5)
    lo, hi = sizing.shard_range(n, rank, world)
    shard = {k: (v[lo:hi] if torch.is_tensor(v) else v[lo:hi]) for k, v in rec.items()}
    res = []
    for async_op in ("1", "0"):
        os.environ["OPS_AMD_DP_ASYNC"] = async_op
        importlib.reload(train)
        d = dataprep.prepare(shard, kind="pinn", seed=2, distributed=True)
        out = train.train_surrogate("pinn", d, train.PinnConfig(batch_size=4, patience=3), device="cpu", autocast_dtype=None, max_epochs=2, seed=4)
        res.append(torch.cat([p.detach().reshape(-1) for p in out["model"].parameters()]).numpy())
    q.put((rank, res[0], res[1]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_async_gradient_allreduce_equals_the_blocking_form_bit_for_bit(world):
    """The data-parallel step enqueues its one collective asynchronously (train.allreduce_grads; OPS_AMD_DP_ASYNC=0 is the blocking
    call): same arithmetic, so two epochs end with bit-identical weights, on every rank, at world 2 and 4."""
    from tests.helpers import run_ranks
    outs = sorted(run_ranks(_ddp_async_worker, world, timeout=600), key=lambda o: o[0])
    for _, wa, wb in outs:
        assert np.array_equal(wa, wb)
        assert np.array_equal(wa, outs[0][1])


def test_stall_guard_ends_the_process_with_its_own_exit_code():
    """r06: the data-parallel step's captured collective replays under `train._StallGuard` the first time -- a replay that hangs (possible only on
    N > 1 ranks, past every in-process fallback) must turn into a non-zero exit of the rank, not a silent stall."""
    import subprocess
    import sys
    code = ("import time\nfrom openpystruct_amd.train import _StallGuard\n"
            "with _StallGuard(0.3, 'a test block'):\n    time.sleep(20)\n")
    p = subprocess.run([sys.executable, "-c", code], cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), capture_output=True, text=True, timeout=120)
    assert p.returncode == 17 and "a test block did not finish within" in p.stderr and "OPS_AMD_DP_ONE_GRAPH=0" in p.stderr
    # the caller's hook speaks first (bench.py prints the FE record it has already measured), then the exit
    code_hook = ("import time\nfrom openpystruct_amd import train\ntrain.stall_hook = lambda what: print('last words: ' + what, flush=True)\n"
                 "with train._StallGuard(0.3, 'a test block'):\n    time.sleep(20)\n")
    p = subprocess.run([sys.executable, "-c", code_hook], cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), capture_output=True, text=True, timeout=120)
    assert p.returncode == 17 and "last words: a test block" in p.stdout
    code_ok = "from openpystruct_amd.train import _StallGuard\nwith _StallGuard(5.0, 'x'):\n    pass\nprint('fine')\n"
    p = subprocess.run([sys.executable, "-c", code_ok], cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "fine" in p.stdout
