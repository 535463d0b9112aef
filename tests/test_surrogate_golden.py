"""Surrogate rows against fixtures produced by the REFERENCE'S OWN code.

tests/golden/make_surrogate_golden.py executed the five reference model scripts (their data-prep block, their classes,
their optimiser / criterion construction and their training loop) in the build container and stored what they computed:
    OpenPyStruct_PINN_MultiCase.py:66-120, :190-388, :395-653, :741-808
    OpenPyStruct_TransformerDiffusionModule_MultiCase.py:72-236, :239-371, :383-633, :722-791
    OpenPyStruct_FNN_MultiCase.py:61-183, :330-438, :525-594
    OpenPyStruct_GNN_MultiCase_Beta.py:64-114, :249-379, :429-485
    OpenPyStruct_FNO_MultiCase_Beta.py:69-195, :340-537, :600-664
The CPU tests compare the build's host logic and module arithmetic (fp32, framework kernels) with those numbers; the
`-m gpu` tests do the same on the MI355X with the hand-written pieces ON (fused stencil+BN, fused loss, flat clip+Adam,
bf16 shadow linears, HIP-graph step).  Tolerances: fp32 1e-5 relative to the tensor's scale (1e-4 on gradients and on
3-epoch loss histories, where fp32 summation order differs); bf16 2e-2.
"""
import importlib.util
import os

import numpy as np
import pytest
import torch

from openpystruct_amd import dataprep, train
from openpystruct_amd import surrogates as S

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
_spec = importlib.util.spec_from_file_location("make_surrogate_golden", os.path.join(GOLD, "make_surrogate_golden.py"))
msg = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(msg)          # fill_state / projections / DeterministicNoise / unpack_records: no reference access

KINDS = ("pinn", "tfd", "fnn", "gnn", "fno")
CFG = {"pinn": train.PinnConfig, "tfd": train.TfdConfig, "fnn": train.FnnConfig, "gnn": train.GnnConfig, "fno": train.FnoConfig}


def gold(kind):
    return np.load(os.path.join(GOLD, f"surrogate_{kind}.npz"), allow_pickle=False)


@pytest.fixture(scope="module")
def records():
    return msg.unpack_records(np.load(os.path.join(GOLD, "surrogate_records.npz")))


def cfg_for(kind):
    c = CFG[kind]()
    c.dropout_rate, c.sigma_0, c.batch_size = 0.0, 0.0, 8        # the generator's overrides
    return c


def prep(kind, records, g, device=None):
    c = cfg_for(kind)
    return dataprep.prepare(records, kind=kind, n_cases=c.n_cases, c=c.c, train_split=c.train_split, perm=torch.as_tensor(g["perm"]),
                            device=device)


_RECORD = None      # when a dict: close() records errors instead of asserting (bf16 A/B against the framework's own bf16 path)


def close(got, want, tol, what=""):
    got = got.detach().double().cpu().numpy() if torch.is_tensor(got) else np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    if tol >= 1e-2:      # bf16 runs: relative L2 error (element-wise maxima of a bf16 result scatter by several ulps of 2^-8)
        err = float(np.linalg.norm(got - want)) / max(float(np.linalg.norm(want)), 1e-30)
    else:
        err = float(np.abs(got - want).max()) / max(float(np.abs(want).max()), 1e-30)
    if _RECORD is not None:
        _RECORD[what] = max(err, _RECORD.get(what, 0.0))
        return
    assert err <= tol, f"{what}: rel err {err:.3e} > {tol:.1e}"


# ----------------------------------------------------------------------------------------------------------------
# a12: data prep
# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", KINDS)
def test_dataprep_matches_reference(kind, records):
    g = gold(kind)
    d = prep(kind, records, g)
    for name, got in (("X_train_tensor", d.X_train), ("Y_train_tensor", d.Y_train), ("X_val_tensor", d.X_val), ("Y_val_tensor", d.Y_val)):
        want = g[name]
        assert tuple(got.shape) == want.shape, name
        assert float(np.abs(got.numpy() - want).max()) <= 2e-6 * max(1.0, float(np.abs(want).max())), name
    assert float(d.min_constraint) == pytest.approx(float(g["min_constraint"]), rel=1e-6)
    assert float(d.max_constraint) == pytest.approx(float(g["max_constraint"]), rel=1e-6)
    for name, sc in d.scalers_Y.items():
        close(sc.mean_, g[f"scaler_Y/{name}/mean"], 1e-6, name)
        close(sc.scale_, g[f"scaler_Y/{name}/scale"], 1e-6, name)


@pytest.mark.parametrize("kind", KINDS)
def test_user_inputs_match_reference(kind, records):
    """The scripts' inference front end (`scale_user_inputs`, PINN:143-188) on one group's raw inputs.  For "tfd" / "gnn" the
    reference has re-fitted the SAME scaler objects on the validation split (TFD:325-328), so its front end scales with the
    validation statistics; the fixture holds what it did."""
    g = gold(kind)
    d = prep(kind, records, g)
    c = cfg_for(kind)
    grp, nc = int(g["user_group"]), c.n_cases
    raw = [[records[k][grp * nc + i] for i in range(nc)] for k in dataprep.INPUT_KEYS]
    X = dataprep.user_inputs(d, kind, *raw)
    want = g["user_feat3"]                                   # [1, n_cases, feat] before flattening / head padding
    got = X.reshape(1, nc, -1)[:, :, : want.shape[2]]
    assert float(np.abs(got.numpy() - want).max()) <= 2e-6 * max(1.0, float(np.abs(want).max()))


# ----------------------------------------------------------------------------------------------------------------
# a10 / a11 / f3: modules, losses, one training forward/backward, the loop
# ----------------------------------------------------------------------------------------------------------------
def build(kind, records, g, device):
    d = prep(kind, records, g, device=device)
    model, crit = train.build_model_and_loss(kind, cfg_for(kind), d, torch.device(device))
    return d, model, crit


@pytest.mark.parametrize("kind", KINDS)
def test_state_dict_layout_matches_reference(kind, records):
    g = gold(kind)
    _, model, crit = build(kind, records, g, "cpu")
    sd = model.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["sd_keys"]]
    assert [",".join(str(v) for v in t.shape) for t in sd.values()] == [str(s) for s in g["sd_shapes"]]
    assert S.count_parameters(model) == int(g["n_params"])
    assert list(crit.state_dict().keys()) == [str(k) for k in g["crit_sd_keys"]]


def _forward_checks(kind, records, device, tol, tol_grad, autocast=None, plain=False):
    """plain=True: the framework's modules only (no fused stencil+BN, no fused loss) -- the A/B baseline of the bf16 test."""
    g = gold(kind)
    d, model, crit = build(kind, records, g, device)
    msg.fill_state(model)
    dev = torch.device(device)
    noise = msg.DeterministicNoise() if kind == "tfd" else None
    ac = torch.autocast(device_type=dev.type, dtype=autocast, enabled=autocast is not None)
    Xe, Ye = d.X_val[:6], d.Y_val[:6]
    Xt, Yt = d.X_train[:8].clone().requires_grad_(True), d.Y_train[:8]

    def run():
        if noise is not None:
            noise.calls = 1000
        model.eval()
        with torch.no_grad(), ac:
            pe = model(Xe)
            le = crit(pe.float(), Ye)
        close(pe, g["eval_preds"], tol, "eval preds")
        close(le, g["eval_loss"], tol, "eval loss")
        if dev.type == "cuda" and not plain:
            with torch.no_grad(), ac:
                close(S.fused_loss(crit, model(Xe) if noise is None else pe, Ye), g["eval_loss"], tol, "fused eval loss")
        if noise is not None:
            noise.calls = 2000
        model.train()
        with ac:
            pt = model(Xt)
            loss = S.fused_loss(crit, pt, Yt) if dev.type == "cuda" and not plain else crit(pt.float(), Yt)
        close(pt, g["train_preds"], tol, "train preds")
        close(loss, g["train_loss"], tol, "train loss")
        loss.backward()
        close(Xt.grad, g["train_input_grad"], tol_grad, "input grad")
        got = msg.projections((n, p.grad) for n, p in model.named_parameters())
        numel = {n: p.numel() for n, p in model.named_parameters()}
        # gradients that are mathematically zero (a bias in front of a BatchNorm) hold rounding noise only: every parameter
        # is compared relative to max(its own L1 norm, numel x 1e-2 x the largest mean |gradient| of the model)
        gmean = max(float(g["grad/" + n][1]) / numel[n] for n in numel)
        for n in numel:
            want, v = g["grad/" + n], got[n]
            ref = max(float(want[1]), numel[n] * gmean * 1e-2)
            if _RECORD is not None:
                _RECORD["grad L1 " + n] = abs(v[1] - want[1]) / ref
                continue
            assert abs(v[1] - want[1]) <= tol_grad * ref, n
            assert abs(v[0] - want[0]) <= tol_grad * ref * 10, n
            if n + "/full" in got:
                assert float(np.abs(got[n + "/full"] - g["grad/" + n + "/full"]).max()) <= tol_grad * ref, n
        for k, t in model.state_dict().items():
            if "running_" in k:
                close(t, g["bn_after/" + k], max(tol, 1e-5), k)
            elif "num_batches" in k:
                assert int(t) == int(g["bn_after/" + k])

    if noise is not None:
        with noise:
            run()
    else:
        run()


@pytest.mark.parametrize("kind", KINDS)
def test_modules_match_reference_cpu(kind, records):
    _forward_checks(kind, records, "cpu", 2e-5, 2e-4)


def _loop_check(kind, records, device, tol, autocast_dtype, use_graph=None):
    g = gold(kind)
    d = prep(kind, records, g, device=device)
    cfg = cfg_for(kind)
    batches = g["loop_batches"]                              # [epochs, steps, batch]
    noise = msg.DeterministicNoise() if kind == "tfd" else None

    def go():
        return train.train_surrogate(kind, d, cfg, device=device, autocast_dtype=autocast_dtype, max_epochs=batches.shape[0],
                                     init_fn=msg.fill_state, batch_order=lambda ep: batches[ep - 1], use_graph=use_graph)

    if noise is not None:
        with noise:
            noise.calls = 0
            res = go()
    else:
        res = go()
    if kind == "fno":
        # The FNO run with these weights is chaotic: perturbing the initial weights of EITHER implementation by 1e-7 relative
        # moves its own per-step losses by 1.5e-6, 1.1e-4, 4.8e-4, 6.2e-3, ... 1.1e-2 (measured, CPU fp32) -- the same profile
        # as the difference to the reference.  So: first epoch tight, the rest to that amplification; the eval-mode losses
        # (BatchNorm on 3-step running statistics: 1e2..1e4) are not compared.
        close(np.array(res["history"]["train"][:1]), g["loop_train_losses"][:1], max(tol, 5e-4), "first-epoch train loss")
        close(np.array(res["history"]["train"]), g["loop_train_losses"], max(tol, 5e-2), "train-loss history")
        return res
    close(np.array(res["history"]["train"]), g["loop_train_losses"], tol, "train-loss history")
    close(np.array(res["history"]["val"]), g["loop_val_losses"], tol, "val-loss history")
    # the reference's evaluation block (PINN:815-852, TFD:800-829, executed by make_surrogate_golden.py past the loop): best checkpoint
    # reloaded, evaluation pass, un-standardised clipped inertias, R^2
    assert int(np.argmin(res["history"]["val"])) + 1 == int(g["eval_best_epoch"])
    close(res["val_true_I"].cpu().numpy(), g["eval_labels_unstd"], 1e-5, "un-standardised validation labels")
    close(res["val_pred_I"].cpu().numpy(), g["eval_preds_unstd"], max(10 * tol, 2e-3), "un-standardised validation predictions")
    ss_tot = float(((g["eval_labels_unstd"] - g["eval_labels_unstd"].mean()) ** 2).sum())
    # R^2 = 1 - ss_res / ss_tot: its error is the error of ss_res / ss_tot, bounded through the prediction error just checked
    assert abs(res["r2_val_I"] - float(g["eval_r2_val"])) <= max(10 * tol, 2e-3) * 4 * (1.0 - float(g["eval_r2_val"])) + 1e-4, (res["r2_val_I"], float(g["eval_r2_val"]), ss_tot)
    return res


@pytest.mark.parametrize("kind", KINDS)
def test_loop_matches_reference_cpu(kind, records):
    """Three epochs of the reference's own loop (its DataLoader order replayed) vs train_surrogate on the CPU."""
    _loop_check(kind, records, "cpu", 5e-4, None)


# ----------------------------------------------------------------------------------------------------------------
# the same on the MI355X, hand-written kernels ON
# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("kind", KINDS)
def test_dataprep_matches_reference_on_the_gpu(kind, records):
    """`dataprep.prepare(device="cuda")`: the tensor-native prep runs on the device the records live on (a12)."""
    g = gold(kind)
    d = prep(kind, records, g, device="cuda")
    for name, got in (("X_train_tensor", d.X_train), ("Y_train_tensor", d.Y_train), ("X_val_tensor", d.X_val), ("Y_val_tensor", d.Y_val)):
        want = g[name]
        assert got.is_cuda and tuple(got.shape) == want.shape, name
        assert float(np.abs(got.cpu().numpy() - want).max()) <= 2e-6 * max(1.0, float(np.abs(want).max())), name


@pytest.mark.gpu
@pytest.mark.parametrize("kind", KINDS)
def test_modules_match_reference_gpu_fp32(kind, records):
    _forward_checks(kind, records, "cuda", 2e-5, 3e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ("pinn", "tfd", "fnn", "gnn"))
def test_modules_match_reference_gpu_bf16(kind, records, monkeypatch):
    """bf16 autocast with the hand-written pieces ON: predictions and loss values follow the reference's fp32 numbers to
    bf16 accuracy (2e-2 relative L2).  Gradients of an L1-type loss flip sign wherever a bf16 prediction lands on the other side
    of its target, so they are compared A/B instead: the fused path must be as close to the reference as the framework's own
    bf16 autocast path (plain modules, same weights) is -- within 1.5x + 1e-2."""
    global _RECORD
    errs = {}
    for plain in (True, False):
        monkeypatch.setattr(S, "_FUSED_STENCIL", not plain)
        _RECORD = errs[plain] = {}
        try:
            _forward_checks(kind, records, "cuda", 2e-2, 6e-2, autocast=torch.bfloat16, plain=plain)
        finally:
            _RECORD = None
    for what, e in errs[False].items():
        if what in ("eval preds", "eval loss", "fused eval loss", "train preds", "train loss"):
            assert e <= 2e-2, (what, e)
        elif what in errs[True]:
            # PINN: CompositeLoss weights sign(p - t) by 1.5e-6 / (|t| + 1e-8) (PINN:646-652); the standardised deflection
            # targets at the supports are exactly 0, so a handful of entries carry weights of 150 and the gradient is
            # decided by the SIGN of bf16 predictions that are ~0: any two bf16 evaluations (the framework's with and
            # without its own fused conv, too) differ by O(1) there.  Bounded, not matched.
            bound = 0.5 if kind == "pinn" else 1.5 * errs[True][what] + 1e-2
            assert e <= bound, (what, e, errs[True][what])


@pytest.mark.gpu
def test_pinn_bf16_gradients_without_the_zero_target_entries(records, monkeypatch):
    """The absolute form of the PINN bf16 gradient check.  CompositeLoss weights sign(p - t) by 1.5e-6 / (|t| + 1e-8)
    (PINN:646-652); the standardised deflection / rotation targets at the supports are exactly 0, so a handful of entries carry
    weights of 150 and their gradient is the SIGN of a bf16 prediction that is ~0 -- no two bf16 evaluations agree there.  Take
    exactly those terms out of the loss (same divisors, the loss minus the zero-target terms) and every parameter gradient of the
    hand-written bf16 path must agree with the float32 framework path (itself pinned to the reference's fixture at 3e-4 by
    test_modules_match_reference_gpu_fp32) to a plain bound -- no reference to how well the framework's bf16 path does.  Measured
    (8 rows, one box): matrices and vectors 0.6-8.0 % relative L2; the stencil's five scalars (conv1.weight [3], bn1.weight, bn1.bias:
    sums of 2 800 sign-carrying terms of the remaining L1 parts that cancel to ~1 % of their size) 0.01-36 %.  Bounds 12 % / 50 %; the
    scalars' own absolute guard is the smooth-objective test of tests/test_gpu_pinn_fused.py."""
    g = gold("pinn")
    dev = torch.device("cuda")

    def grads(autocast, plain):
        monkeypatch.setattr(S, "_FUSED_STENCIL", not plain)
        d, model, crit = build("pinn", records, g, "cuda")
        msg.fill_state(model)
        Xt, Yt = d.X_train[:8].clone(), d.Y_train[:8]
        n, nd = crit.nelem, crit.deflection_dim
        zero = (Yt.abs() < 1e-12)
        zero[:, :n] = False
        assert int(zero.sum()) > 0
        model.train()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            pt = model(Xt)
            loss = crit(pt.float(), Yt) if plain else S.fused_loss(crit, pt, Yt)
        pf = pt.float()
        rel = (pf - Yt).abs() / (Yt.abs() + 1e-8) * zero
        B = Yt.shape[0]
        corr = crit.penalty_pinn * (rel[:, n:n + nd].sum() / (B * nd) + rel[:, n + nd:].sum() / (B * (Yt.shape[1] - n - nd)))
        (loss - corr).backward()
        return {k: q.grad.detach().double().clone() for k, q in model.named_parameters()}

    ref = grads(False, True)
    got = grads(True, False)
    gmean = max(float(v.abs().mean()) for v in ref.values())
    worst = {}
    for k, v in ref.items():
        scale = max(float(v.norm()), float(np.sqrt(v.numel())) * gmean * 1e-2)      # (mathematically zero gradients: against the model's scale)
        worst[k] = float((got[k] - v).norm()) / scale
    if os.environ.get("OPS_AMD_PRINT_GRAD_TABLE"):
        for k, e in worst.items():
            print("%-45s %.4f" % (k, e))
    bad = {k: e for k, e in worst.items() if not e <= (0.12 if ref[k].numel() > 3 else 0.5)}
    assert not bad, (bad, max(worst.values()))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", KINDS)
def test_loop_matches_reference_gpu_fp32(kind, records):
    """fused loss + flat clip/Adam (+ fused stencil for the PINN), HIP-graph step where the noise allows it."""
    _loop_check(kind, records, "cuda", 1e-3, None, use_graph=(kind != "tfd"))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ("pinn", "tfd", "fnn", "gnn"))
def test_loop_matches_reference_gpu_bf16(kind, records):
    """bf16 autocast + shadow linears: the loss history follows the reference's fp32 run to bf16 accuracy."""
    _loop_check(kind, records, "cuda", 5e-2, torch.bfloat16, use_graph=(kind != "tfd"))
