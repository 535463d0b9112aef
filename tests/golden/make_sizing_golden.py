"""Generates tests/golden/sizing_reference_*.npz by EXECUTING the reference's own generator / optimiser scripts (build container only).

The five scripts that drive the FE solve -- SingleCore (SC), MultiCore (MC), GPU, BeamOpt (BO), FrameOpt_Discrete_Beta (FR) -- are run
statement by statement (ast) with `openseespy.opensees` replaced by tests/golden/opensees_stub.py, a recorder of the command API whose
`analyze` is the oracle's 3-DOF banded solve.  Everything around the solve is therefore the reference's own code, executed: the case
draws (`random`), `setup_model`'s command sequence, the float32 tensors, the loss, torch's Adam / ExponentialLR, the clamp, the
early-stop bookkeeping, the record assembly and `main()`'s loop (for MC: its batches of 500 with the failed-sample filter; joblib's
process pool is replaced by an in-process stand-in because worker processes could not see the stub).

Nothing of the reference's text is stored.  The fixtures hold: the drawn inputs, the 13-field records `main()` collected, the number of
epochs and the loss of every epoch (captured by observing `Tensor.backward`), and -- as data -- the command log of one model build with
the answers the stub gave for it.

Overrides (the scripts have no CLI; each replaces the VALUE of one top-level constant right after the script assigned it):
`num_samples` (100000 -> the number of cases below), `random_bridge` for the random-bridge fixtures, `num_workers` irrelevant.  FR: none
except the seed of `random`.  `random.seed(...)` is set before each run (the reference never seeds).

Run (in the build container, where /root/reference exists):  python tests/golden/make_sizing_golden.py
The GPU box never sees /root/reference; tests read only the .npz files written here.
"""
from __future__ import annotations

import ast
import json
import os
import random
import sys
import tempfile
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import opensees_stub as stub  # noqa: E402

SCRIPTS = {
    "sc": "OpenPyStruct_BeamOpt_training_SingleCore.py",
    "mc": "OpenPyStruct_BeamOpt_training_MultiCore.py",
    "gpu": "OpenPyStruct_BeamOpt_training_GPU.py",
    "bo": "OpenPyStruct_BeamOpt.py",
    "fr": "OpenPyStruct_FrameOpt_Discrete_Beta.py",
}
RECORD_KEYS = ("roller_x_locations", "force_x_locations", "force_values", "I_values", "shear_forces", "bending_moments",
               "node_positions", "roller_nodes", "force_nodes", "num_nodes", "L", "rotations", "deflections")


class BackwardTap:
    """Observes `Tensor.backward`: the value of every tensor the script back-propagates from (its `total_loss`), in call order."""

    def __init__(self):
        self.values = []

    def __enter__(self):
        import torch
        self._orig = torch.Tensor.backward
        me = self

        def backward(t, *a, **k):
            me.values.append(float(t.detach()))
            return me._orig(t, *a, **k)

        torch.Tensor.backward = backward
        return self

    def __exit__(self, *a):
        import torch
        torch.Tensor.backward = self._orig


def _is_main_guard(node):
    return isinstance(node, ast.If) and isinstance(node.test, ast.Compare) and isinstance(node.test.left, ast.Name) and node.test.left.id == "__name__"


def _is_epoch_loop(node):
    return isinstance(node, ast.For) and isinstance(node.target, ast.Name) and node.target.id == "epoch"


def _calls_plt(node):
    return any(isinstance(n, ast.Attribute) and isinstance(n.value, ast.Name) and n.value.id == "plt" for n in ast.walk(node))


def exec_script(kind, overrides, stop, after_stmt=None):
    """Executes the script's top-level statements in order until `stop(node, state)` says so; returns its namespace."""
    import matplotlib
    matplotlib.use("Agg")
    path = os.path.join(REF, SCRIPTS[kind])
    tree = ast.parse(open(path).read(), filename=path)
    ns = {"__name__": "__reference__", "__file__": path}
    state = {"loop_done": False}
    for node in tree.body:
        if stop(node, state):
            break
        exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
        if _is_epoch_loop(node):
            state["loop_done"] = True
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
            if node.targets[0].id in overrides:
                ns[node.targets[0].id] = overrides[node.targets[0].id]
        if after_stmt is not None:
            after_stmt(node, ns)
    return ns


def _pad(rows, width, dtype):
    out = np.zeros((len(rows), width), dtype=dtype)
    for i, r in enumerate(rows):
        out[i, :len(r)] = r
    return out


def pack_training_data(td):
    n = len(td["I_values"])
    out = {"n": np.array(n)}
    out["n_rollers"] = np.array([len(r) for r in td["roller_nodes"]], dtype=np.int32)
    out["n_forces"] = np.array([len(r) for r in td["force_nodes"]], dtype=np.int32)
    wr, wf = int(out["n_rollers"].max()), int(out["n_forces"].max())
    out["roller_nodes"] = _pad(td["roller_nodes"], wr, np.int32)
    out["roller_x_locations"] = _pad(td["roller_x_locations"], wr, np.float64)
    out["force_nodes"] = _pad(td["force_nodes"], wf, np.int32)
    out["force_x_locations"] = _pad(td["force_x_locations"], wf, np.float64)
    out["force_values"] = _pad(td["force_values"], wf, np.float64)
    for k in ("I_values", "shear_forces", "bending_moments"):
        a = np.asarray(td[k], dtype=np.float64)
        assert np.array_equal(a, a.astype(np.float32).astype(np.float64)), k       # float32-valued (SC:163, :189-190)
        out[k] = a.astype(np.float32)
    for k in ("node_positions", "rotations", "deflections", "L"):
        out[k] = np.asarray(td[k], dtype=np.float64)
    out["num_nodes"] = np.asarray(td["num_nodes"], dtype=np.int32)
    return out


def run_generator(kind, n_cases, seed, random_bridge=0):
    """SC / MC / GPU: executes the module level, then the script's own `main()` with num_samples = n_cases."""
    stub.install()
    per_case = []          # (epochs, [loss per epoch], first-epoch forces / final result captured below)
    first = {}

    def after_stmt(node, ns):
        if isinstance(node, ast.ImportFrom) and node.module == "joblib":
            # in-process stand-in for the loky pool (MC:258): worker processes would import the real openseespy
            ns["delayed"] = lambda f: (lambda *a, **k: (f, a, k))
            ns["Parallel"] = lambda **kw: (lambda jobs: [f(*a, **k) for f, a, k in jobs])

    ns = exec_script(kind, dict(num_samples=n_cases, random_bridge=random_bridge), lambda node, st: _is_main_guard(node), after_stmt)
    assert ns["num_samples"] == n_cases and ns["flag"] == random_bridge
    inner = ns["generate_sample"]
    tap = BackwardTap()

    def generate_sample(*a, **k):
        stub.reset_counters()
        mark = len(tap.values)
        if not first:
            stub.start_log()
        res = inner(*a, **k)
        losses = tap.values[mark:]
        per_case.append((stub.n_analyze, losses, res is not None))
        return res

    def hook(result):
        if not first:                                   # the very first analyze of the run: log + answers
            first["log"] = stub.stop_log()
            first["forces"] = result["forces"].copy()
            first["disp"] = result["disp"].copy()

    ns["generate_sample"] = generate_sample
    stub.analyze_hook = hook
    random.seed(seed)
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp, tap, warnings.catch_warnings():
        warnings.simplefilter("ignore")
        os.chdir(tmp)
        try:
            ns["main"]()
            td_file = json.load(open("training_data_PINN_mini.json"))       # what the script wrote (SC:263)
        finally:
            os.chdir(cwd)
            stub.analyze_hook = None
    td = ns["training_data"]
    assert list(td_file) == list(RECORD_KEYS)
    out = pack_training_data(td)
    kept = [c for c in per_case if c[2]]
    assert len(kept) == int(out["n"])
    out["epochs_run"] = np.array([c[0] for c in kept], dtype=np.int32)
    assert all(c[0] == len(c[1]) for c in kept)
    out["loss_history"] = _pad([c[1] for c in kept], max(c[0] for c in kept), np.float64)
    out["n_drawn"] = np.array(len(per_case))
    out["command_log"] = np.array(json.dumps(first["log"]))
    out["log_forces"], out["log_disp"] = first["forces"], first["disp"]
    out["constants"] = np.array(json.dumps({k: ns[k] for k in ("E", "nu", "A", "L_max", "num_nodes", "max_force", "min_force", "uniform_udl", "I_0",
                                                               "max_e", "lr", "gamma", "alpha_moment", "alpha_shear", "tolerance", "patience")}))
    out["seed"] = np.array(seed)
    out["random_bridge"] = np.array(random_bridge)
    return out


def run_beam_opt(seeds):
    """BO: the whole script is one case; one run per seed, stopped after the block that reads the final responses (BO:262-270)."""
    stub.install()
    cols = {k: [] for k in ("roller_nodes", "force_nodes", "force_values", "I_values", "shear_forces", "bending_moments", "deflections",
                            "rotations", "epochs_run", "loss_total", "loss_primary", "loss_bending", "loss_shear")}
    consts = None
    for seed in seeds:
        random.seed(seed)
        stub.reset_counters()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ns = exec_script("bo", {}, lambda node, st: st["loop_done"] and _calls_plt(node) and not isinstance(node, ast.For))
        # the script stops here with the LAST epoch's model still in the domain (BO:262-268 reads it)
        r = stub.current_result()
        cols["roller_nodes"].append(ns["roller_nodes"]); cols["force_nodes"].append(ns["force_nodes"])
        cols["force_values"].append(ns["force_values"])
        cols["I_values"].append(ns["I_tensor"].detach().numpy().copy())
        cols["shear_forces"].append(r["forces"][:, 1].copy()); cols["bending_moments"].append(r["forces"][:, 2].copy())
        cols["deflections"].append(r["disp"][:, 1].copy()); cols["rotations"].append(r["disp"][:, 2].copy())
        cols["epochs_run"].append(stub.n_analyze)
        h = ns["loss_history"]
        assert len(h["total"]) == stub.n_analyze
        cols["loss_total"].append(h["total"]); cols["loss_primary"].append(h["primary"])
        cols["loss_bending"].append(h["bending_energy"]); cols["loss_shear"].append(h["shear_energy"])
        consts = {k: ns[k] for k in ("E", "nu", "A", "L", "num_nodes", "N_rollers", "M_forces", "L_min", "max_force", "uniform_udl", "I_0",
                                     "num_epochs", "lr", "gamma", "alpha_moment", "alpha_shear", "tolerance", "patience")}
    w = max(cols["epochs_run"])
    out = dict(n=np.array(len(seeds)), seeds=np.array(seeds), roller_nodes=np.array(cols["roller_nodes"], dtype=np.int32),
               force_nodes=np.array(cols["force_nodes"], dtype=np.int32), force_values=np.array(cols["force_values"]),
               I_values=np.array(cols["I_values"], dtype=np.float32), shear_forces=np.array(cols["shear_forces"]),
               bending_moments=np.array(cols["bending_moments"]), deflections=np.array(cols["deflections"]),
               rotations=np.array(cols["rotations"]), epochs_run=np.array(cols["epochs_run"], dtype=np.int32),
               constants=np.array(json.dumps(consts)))
    for k in ("loss_total", "loss_primary", "loss_bending", "loss_shear"):
        out[k] = _pad(cols[k], w, np.float64)
    return out


def run_frame_opt(seeds, max_epochs=None):
    """FR: one run per seed (the seed decides bays x stories), stopped after the re-run of the analysis with the optimised
    inertias (FR:213-219)."""
    stub.install()
    runs = {}
    for i, seed in enumerate(seeds):
        random.seed(seed)
        stub.reset_counters()
        first = {}

        def hook(result, first=first):
            if "log" not in first:
                first["log"] = stub.stop_log()
                first["forces"], first["disp"] = result["forces"].copy(), result["disp"].copy()

        stub.analyze_hook = hook
        stub.start_log()
        ov = {} if max_epochs is None else dict(num_epochs=max_epochs)
        import contextlib, io
        with warnings.catch_warnings(), contextlib.redirect_stdout(io.StringIO()):
            warnings.simplefilter("ignore")
            ns = exec_script("fr", ov, lambda node, st: st["loop_done"] and _calls_plt(node))
        stub.analyze_hook = None
        r = stub.current_result()                      # the re-run with the optimised inertias
        p = f"run{i}/"
        runs[p + "num_bays"], runs[p + "num_stories"] = np.array(ns["num_bays"]), np.array(ns["num_stories"])
        runs[p + "epochs_run"] = np.array(stub.n_analyze - 1)
        runs[p + "loss_history"] = np.array(ns["loss_history"])
        assert len(ns["loss_history"]) == stub.n_analyze - 1
        runs[p + "I_values"] = ns["opt_I"].astype(np.float32)
        runs[p + "forces"], runs[p + "disp"] = r["forces"].copy(), r["disp"].copy()
        runs[p + "n_eq"], runs[p + "kd"] = np.array(r["n_eq"]), np.array(r["kd"])
        runs[p + "best_loss"] = np.array(ns["best_loss"])
        if i == 0:
            runs["command_log"] = np.array(json.dumps(first["log"]))
            runs["log_forces"], runs["log_disp"] = first["forces"], first["disp"]
            runs["constants"] = np.array(json.dumps({k: ns[k] for k in ("max_bays", "max_stories", "bay_width", "story_height", "E", "nu", "A", "I0",
                                                                         "alpha_moment", "alpha_shear", "k", "lateral_load", "vertical_load",
                                                                         "num_epochs", "lr", "tolerance", "patience")}))
    runs["n"] = np.array(len(seeds))
    runs["seeds"] = np.array(seeds)
    return runs


def main():
    assert os.path.isdir(REF), "the reference is only present in the build container"
    what = sys.argv[1:] or ["sc", "mc", "gpu", "sc_rb", "mc_rb", "bo", "fr"]
    for w in what:
        if w in ("sc", "mc", "gpu"):
            o = run_generator(w, 16 if w != "gpu" else 6, seed={"sc": 101, "mc": 202, "gpu": 303}[w])
        elif w in ("sc_rb", "mc_rb"):
            o = run_generator(w[:2], 8, seed={"sc_rb": 404, "mc_rb": 505}[w], random_bridge=1)
        elif w == "bo":
            o = run_beam_opt([11, 12, 13, 14, 15, 16])
        elif w == "fr":
            o = run_frame_opt([1, 2, 3, 4, 5, 6])
        else:
            raise SystemExit(f"unknown fixture {w}")
        path = os.path.join(HERE, f"sizing_reference_{w}.npz")
        np.savez_compressed(path, **o)
        ep = o["epochs_run"] if "epochs_run" in o else [int(o[f"run{i}/epochs_run"]) for i in range(int(o["n"]))]
        print(w, "cases", int(o["n"]), "epochs", list(map(int, ep)), "bytes", os.path.getsize(path), flush=True)


if __name__ == "__main__":
    main()
