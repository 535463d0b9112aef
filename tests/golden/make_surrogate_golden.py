"""Generates tests/golden/surrogate_*.npz by EXECUTING the reference's own model scripts (build container only).

Each of the five model scripts under /root/reference is run top to bottom -- its own configuration block, data-prep
block, class definitions, model / optimiser / criterion construction and its own training loop -- statement by statement
(ast), in a scratch directory that holds a small synthetic `StructDataLite.json`.  Nothing of the reference's text is
stored: the fixtures hold inputs and the numbers the reference's code produced from them.

What is overridden (the scripts have no CLI; every override replaces the VALUE of one top-level constant right after the
script assigned it): num_epochs 3, batch_size 8 (the synthetic dataset has 30 groups), dropout_rate 0 and sigma_0 0 (random
streams of a CPU run cannot be reproduced on the GPU).  The diffusion module of the Transformer-Diffusion script draws
random numbers in eval mode too; for that script `torch.randint` / `torch.randn_like` are replaced, while the script runs,
by counter-based deterministic versions (`DeterministicNoise`) that the parity test installs around the build's model as well.

Weights: the script's freshly constructed model is filled, in `state_dict()` order, from `numpy.random.default_rng(FILL_SEED)`
(`fill_state`) -- the fixture stores the reference's state-dict key names and shapes, the test asserts the build's modules
have exactly those and fills them the same way, so no multi-megabyte weight file is needed.  Gradients are stored as one
random projection + L1 norm per parameter (full tensors for parameters below 4096 elements).

Run (in the build container, where /root/reference exists):  python tests/golden/make_surrogate_golden.py
The GPU box never sees /root/reference; tests read only the .npz files written here.
"""
from __future__ import annotations

import ast
import json
import os
import sys
import tempfile
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
FILL_SEED = 20250307
SCRIPTS = {
    "pinn": "OpenPyStruct_PINN_MultiCase.py",
    "tfd": "OpenPyStruct_TransformerDiffusionModule_MultiCase.py",
    "fnn": "OpenPyStruct_FNN_MultiCase.py",
    "gnn": "OpenPyStruct_GNN_MultiCase_Beta.py",
    "fno": "OpenPyStruct_FNO_MultiCase_Beta.py",
}
OVERRIDES = dict(num_epochs=3, batch_size=8, dropout_rate=0.0, sigma_0=0.0)


# ------------------------------------------------------------------------------------------------------------------
# helpers shared with tests/test_surrogate_golden.py (imported from there; they never touch /root/reference)
# ------------------------------------------------------------------------------------------------------------------
def fill_state(module, seed=FILL_SEED):
    """Deterministic weights in state_dict() order: U(-a, a) with a = 1/sqrt(fan_in) for >= 2-D tensors, U(-0.1, 0.1)
    (+1 for norm scales and running variances) for 1-D ones; integer and constant buffers untouched."""
    import torch
    rng = np.random.default_rng(seed)
    sd = module.state_dict()
    pnames = {n for n, _ in module.named_parameters()}
    with torch.no_grad():
        for k, t in sd.items():
            leaf = k.rsplit(".", 1)[-1]
            if not t.is_floating_point() or (k not in pnames and leaf not in ("running_mean", "running_var")):
                continue            # constant buffers (adjacency, positional table) keep what the constructor computed
            shape = tuple(t.shape)
            if t.dim() >= 2:
                fan_in = int(np.prod(shape[1:]))
                a = 1.0 / np.sqrt(max(fan_in, 1))
                val = rng.uniform(-a, a, size=shape)
            else:
                val = rng.uniform(-0.1, 0.1, size=shape)
                leaf = k.rsplit(".", 1)[-1]
                if leaf == "running_var" or (leaf == "weight" and t.dim() == 1):
                    val = val + 1.0
            t.copy_(torch.as_tensor(val, dtype=t.dtype))
    return module


def projections(named_tensors, seed=FILL_SEED + 1):
    """name -> (random projection, L1 norm) of a tensor; full values for small tensors."""
    rng = np.random.default_rng(seed)
    out = {}
    for k, t in named_tensors:
        a = t.detach().double().cpu().numpy()
        r = rng.standard_normal(a.shape)
        out[k] = np.array([float((a * r).sum()), float(np.abs(a).sum())])
        if a.size < 4096:
            out[k + "/full"] = a
    return out


class DeterministicNoise:
    """Context manager: torch.randint / torch.randn_like return counter-based numpy streams (call index -> seed), so the
    reference's CPU run and the build's GPU run see the same noise in the same call order."""

    def __init__(self, seed=77):
        self.seed, self.calls = seed, 0

    def __enter__(self):
        import torch
        self._ri, self._rn = torch.randint, torch.randn_like
        me = self

        def randint(low, high=None, size=None, **kw):
            if size is None:            # torch.randint(high, size)
                low, high, size = 0, low, high
            rng = np.random.default_rng([me.seed, me.calls]); me.calls += 1
            return torch.as_tensor(rng.integers(low, high, size=tuple(size)), dtype=torch.int64).to(kw.get("device", "cpu"))

        def randn_like(x, **kw):
            rng = np.random.default_rng([me.seed, me.calls]); me.calls += 1
            return torch.as_tensor(rng.standard_normal(tuple(x.shape)), dtype=torch.float32).to(device=x.device, dtype=x.dtype)

        torch.randint, torch.randn_like = randint, randn_like
        return self

    def __exit__(self, *a):
        import torch
        torch.randint, torch.randn_like = self._ri, self._rn


# ------------------------------------------------------------------------------------------------------------------
# synthetic dataset (from the build's CPU oracle; float32-representable because the scripts cast to float32 on load)
# ------------------------------------------------------------------------------------------------------------------
def make_records(n_samples=180, seed=4242):
    sys.path.insert(0, ROOT)
    from oracle import beam_oracle as bo
    rng = np.random.default_rng(seed)
    N = bo.N_NODES_REF
    x = np.linspace(0.0, bo.L_REF, N)
    fix = bo.reference_fix_mask()
    rollers = [int(i) for i in np.nonzero(fix)[0][1:] + 1]
    rec = {k: [] for k in ("roller_x_locations", "force_x_locations", "force_values", "I_values", "shear_forces", "bending_moments",
                           "node_positions", "roller_nodes", "force_nodes", "num_nodes", "L", "rotations", "deflections")}
    avail = np.array([n for n in range(2, N) if n not in rollers])
    f32 = lambda a: [float(np.float32(v)) for v in a]  # noqa: E731
    for _ in range(n_samples):
        k = int(rng.integers(1, 5))
        nodes = sorted(int(n) for n in rng.choice(avail, size=k, replace=False))
        vals = rng.uniform(bo.MAX_FORCE, bo.MIN_FORCE, size=k)
        I = np.exp(rng.uniform(np.log(3e-3), np.log(0.75), size=N - 1)).astype(np.float32).astype(np.float64)
        Fy = np.zeros(N)
        Fy[np.array(nodes) - 1] = vals
        v, th, V, M, st = bo.solve_beam_batched(x, bo.E_REF, I[None], fix, Fy[None], bo.UDL_REF)
        rec["roller_x_locations"].append(f32(x[np.array(rollers) - 1]))
        rec["force_x_locations"].append(f32(x[np.array(nodes) - 1]))
        rec["force_values"].append(f32(vals))
        rec["I_values"].append(f32(I))
        rec["shear_forces"].append(f32(V[0])); rec["bending_moments"].append(f32(M[0]))
        rec["node_positions"].append(f32(x))
        rec["roller_nodes"].append(rollers); rec["force_nodes"].append(nodes)
        rec["num_nodes"].append(N); rec["L"].append(float(bo.L_REF))
        rec["rotations"].append(f32(th[0])); rec["deflections"].append(f32(v[0]))
    return rec


def pack_records(rec):
    """13-key record lists -> compact float32 arrays (ragged force lists zero-padded, with counts)."""
    n = len(rec["I_values"])
    cnt = np.array([len(r) for r in rec["force_nodes"]], dtype=np.int32)
    fn = np.zeros((n, 4), dtype=np.int32); fv = np.zeros((n, 4), dtype=np.float32); fx = np.zeros((n, 4), dtype=np.float32)
    for i in range(n):
        fn[i, :cnt[i]] = rec["force_nodes"][i]; fv[i, :cnt[i]] = rec["force_values"][i]; fx[i, :cnt[i]] = rec["force_x_locations"][i]
    f = lambda k: np.asarray(rec[k], dtype=np.float32)  # noqa: E731
    return dict(force_count=cnt, force_nodes=fn, force_values=fv, force_x_locations=fx, I_values=f("I_values"),
                rotations=f("rotations"), deflections=f("deflections"), node_positions=f("node_positions")[0],
                roller_nodes=np.asarray(rec["roller_nodes"][0], dtype=np.int32), roller_x_locations=f("roller_x_locations")[0],
                L=np.array(rec["L"][0]))


def unpack_records(z):
    """Inverse of pack_records: the reference's JSON layout (shear / moment lists, which no model script reads, are zeros)."""
    n = z["I_values"].shape[0]
    cnt = z["force_count"]
    fl = lambda a: [float(v) for v in a]  # noqa: E731
    return {
        "roller_x_locations": [fl(z["roller_x_locations"])] * n,
        "force_x_locations": [fl(z["force_x_locations"][i, :cnt[i]]) for i in range(n)],
        "force_values": [fl(z["force_values"][i, :cnt[i]]) for i in range(n)],
        "I_values": [fl(r) for r in z["I_values"]],
        "shear_forces": [[0.0] * z["I_values"].shape[1]] * n, "bending_moments": [[0.0] * z["I_values"].shape[1]] * n,
        "node_positions": [fl(z["node_positions"])] * n,
        "roller_nodes": [[int(v) for v in z["roller_nodes"]]] * n,
        "force_nodes": [[int(v) for v in z["force_nodes"][i, :cnt[i]]] for i in range(n)],
        "num_nodes": [int(z["node_positions"].shape[0])] * n, "L": [float(z["L"])] * n,
        "rotations": [fl(r) for r in z["rotations"]], "deflections": [fl(r) for r in z["deflections"]],
    }


# ------------------------------------------------------------------------------------------------------------------
# the runner: executes a reference script's top-level statements one by one
# ------------------------------------------------------------------------------------------------------------------
def run_script(kind, workdir, before_loop, after_loop, after_eval=None):
    import matplotlib
    matplotlib.use("Agg")
    if "seaborn" not in sys.modules:        # styling only; absent from this image
        sys.modules["seaborn"] = types.ModuleType("seaborn")
    path = os.path.join(REF, SCRIPTS[kind])
    tree = ast.parse(open(path).read(), filename=path)
    ns = {"__name__": "__reference__", "__file__": path}
    cwd = os.getcwd()
    os.chdir(workdir)
    done_loop = False
    try:
        for node in tree.body:
            is_loop = isinstance(node, ast.For) and isinstance(node.target, ast.Name) and node.target.id == "epoch"
            is_loop = is_loop and not done_loop
            if is_loop:
                before_loop(ns)
            code = compile(ast.Module(body=[node], type_ignores=[]), path, "exec")
            exec(code, ns)
            if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
                name = node.targets[0].id
                if name in OVERRIDES:
                    ns[name] = OVERRIDES[name]
            if is_loop:
                after_loop(ns)
                done_loop = True
                if after_eval is None:
                    break
            if after_eval is not None and "r2_val" in ns:       # the script's evaluation block has run (PINN:815-852, TFD:800-829)
                after_eval(ns)
                break
    finally:
        os.chdir(cwd)
    return ns


def generate(kind, records, out_path):
    import torch
    torch.manual_seed(1234)
    np.random.seed(1234)
    out = {}
    noise = DeterministicNoise() if kind == "tfd" else None
    state = {}

    def before_loop(ns):
        model, crit = ns["model"], ns["criterion"]
        fill_state(model)
        sd = model.state_dict()
        out["sd_keys"] = np.array(list(sd.keys()))
        out["sd_shapes"] = np.array([",".join(str(d) for d in t.shape) for t in sd.values()])
        out["n_params"] = np.array(sum(p.numel() for p in model.parameters()))
        out["crit_sd_keys"] = np.array(list(crit.state_dict().keys()))
        # ---- data prep: what the script's own module-level code produced
        out["perm"] = np.asarray(ns["indices"], dtype=np.int64)
        for k in ("X_train_tensor", "Y_train_tensor", "X_val_tensor", "Y_val_tensor"):
            out[k] = ns[k].numpy()
        out["min_constraint"] = np.array(float(ns["min_constraint"])); out["max_constraint"] = np.array(float(ns["max_constraint"]))
        for name, sc in ns["scalers_inputs"].items():
            out[f"scaler_in/{name}/mean"] = sc.mean_; out[f"scaler_in/{name}/scale"] = sc.scale_
        scY = ns["scalers_Y"] if "scalers_Y" in ns else {"I": ns["scaler_Y"]}
        for name, sc in scY.items():
            out[f"scaler_Y/{name}/mean"] = sc.mean_; out[f"scaler_Y/{name}/scale"] = sc.scale_
        # ---- inference front end (the scripts' own scale_user_inputs on the first validation group's raw inputs)
        g = int(ns["val_idx"][0])
        nc = ns["n_cases"]
        raw = [[records[k][g * nc + i] for i in range(nc)] for k in ("roller_x_locations", "force_x_locations", "force_values", "node_positions")]
        feat3 = ns["scale_user_inputs"](*raw, ns["scalers_inputs"], nc, ns["max_lengths"])
        out["user_group"] = np.array(g)
        out["user_feat3"] = np.asarray(feat3, dtype=np.float64)
        # ---- model: eval forward, train-mode forward + loss + gradients (dropout 0), on the first 8 training rows
        snap = {k: v.clone() for k, v in sd.items()}
        Xe, Ye = ns["X_val_tensor"][:6], ns["Y_val_tensor"][:6]
        Xt, Yt = ns["X_train_tensor"][:8].clone().requires_grad_(True), ns["Y_train_tensor"][:8]
        if noise is not None:
            noise.calls = 1000
        model.eval()
        with torch.no_grad():
            pe = model(Xe)
            out["eval_preds"] = pe.numpy()
            out["eval_loss"] = np.array(float(crit(pe, Ye)))
        if noise is not None:
            noise.calls = 2000
        model.train()
        pt = model(Xt)
        loss = crit(pt, Yt)
        out["train_preds"] = pt.detach().numpy()
        out["train_loss"] = np.array(float(loss))
        model.zero_grad()
        loss.backward()
        out["train_input_grad"] = Xt.grad.numpy()
        for k, v in projections((n, p.grad) for n, p in model.named_parameters()).items():
            out["grad/" + k] = v
        out["sd_after_train_fwd"] = np.array(0)
        for k, v in model.state_dict().items():     # BatchNorm running statistics after ONE training forward
            if "running_" in k or "num_batches" in k:
                out["bn_after/" + k] = v.numpy().copy()
        model.load_state_dict(snap)
        model.zero_grad()
        # ---- hooks for the loop: batches the DataLoader hands out, per-step losses
        state["steps_X"], state["step_loss"] = [], []
        model.register_forward_pre_hook(lambda m, a: state["steps_X"].append(a[0].detach().clone()) if m.training else None)
        crit.register_forward_hook(lambda m, a, o: state["step_loss"].append(float(o)) if model.training else None)
        if noise is not None:
            noise.calls = 0

    def after_loop(ns):
        model = ns["model"]
        Xtr = ns["X_train_tensor"]
        flat = Xtr.reshape(Xtr.shape[0], -1)
        sched = []
        for xb in state["steps_X"]:                 # sigma_0 = 0: every batch row IS a training row -> recover its index
            xb = xb.reshape(xb.shape[0], -1)
            d = (xb[:, None, :] - flat[None, :, :]).abs().amax(dim=2)
            idx = d.argmin(dim=1)
            assert float(d.min(dim=1).values.max()) == 0.0
            sched.append(idx.numpy())
        nb = len(sched) // OVERRIDES["num_epochs"]
        out["loop_batches"] = np.stack(sched).reshape(OVERRIDES["num_epochs"], nb, -1)
        out["loop_step_loss"] = np.array(state["step_loss"])
        out["loop_train_losses"] = np.array(ns["train_losses"]); out["loop_val_losses"] = np.array(ns["val_losses"])
        for k, v in projections(model.state_dict().items()).items():
            out["final/" + k] = v
        out["loop_final_lr"] = np.array(ns["optimizer"].param_groups[0]["lr"])

    def after_eval(ns):
        """The script's own evaluation block: best checkpoint reloaded from the file its loop wrote, validation pass in evaluation mode,
        un-standardised and clipped inertias, sklearn's r2_score on the raveled arrays."""
        out["eval_r2_val"] = np.array(float(ns["r2_val"]))
        pu = ns["all_preds_I_unstd"] if "all_preds_I_unstd" in ns else ns["all_preds_unstd"]
        lu = ns["all_labels_I_unstd"] if "all_labels_I_unstd" in ns else ns["all_labels_unstd"]
        out["eval_preds_unstd"], out["eval_labels_unstd"] = np.asarray(pu, dtype=np.float64), np.asarray(lu, dtype=np.float64)
        out["eval_best_epoch"] = np.array(int(np.argmin(ns["val_losses"])) + 1)
        for k, v in projections(ns["model"].state_dict().items()).items():      # the reloaded best state
            out["best/" + k] = v

    with tempfile.TemporaryDirectory() as tmp:
        for name in ("StructDataLite.json", "StructDataMedium.json", "training_data_PINN_Case_two.json"):   # PINN:192, FNO:199, GNN:118
            with open(os.path.join(tmp, name), "w") as f:
                json.dump(records, f)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if noise is not None:
                with noise:
                    run_script(kind, tmp, before_loop, after_loop, after_eval)
            else:
                run_script(kind, tmp, before_loop, after_loop, after_eval)
    out["overrides"] = np.array(json.dumps(OVERRIDES))
    np.savez_compressed(out_path, **out)
    return out


def main():
    assert os.path.isdir(REF), "the reference is only present in the build container"
    packed = pack_records(make_records())
    np.savez_compressed(os.path.join(HERE, "surrogate_records.npz"), **packed)
    rec = unpack_records(packed)
    for kind in (sys.argv[1:] or SCRIPTS):
        o = generate(kind, rec, os.path.join(HERE, f"surrogate_{kind}.npz"))
        print(kind, "params", int(o["n_params"]), "train_loss", float(o["train_loss"]), "loop", o["loop_train_losses"], o["loop_val_losses"])


if __name__ == "__main__":
    main()
