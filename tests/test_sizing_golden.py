"""The generator rows (b, f1's loop, f4) against fixtures the REFERENCE'S OWN CODE produced.

tests/golden/sizing_reference_*.npz come from tests/golden/make_sizing_golden.py: the reference's SingleCore / MultiCore / GPU
generator scripts (their `main()` and `generate_sample`), its single-case BeamOpt script and its FrameOpt script executed statement by
statement in the build container, with `openseespy.opensees` replaced by a recorder whose `analyze` is the oracle's 3-DOF band solve
(tests/golden/opensees_stub.py).  Everything around the FE solve -- case draws, the command sequence, float32 tensors, loss, torch's Adam
/ ExponentialLR, clamp, early stop, record assembly -- is therefore the reference's, executed, not re-typed.

CPU tests (no marker): the fixtures' own consistency, oracle/sizing_oracle.py and oracle/frame_sizing_oracle.py (the re-typed loops the
other GPU tests use as their checker) reproduce them, and the recorded command logs replay into the product's OpenSees-command shim.
`-m gpu` tests: the HIP sizing loop (`sizing.optimize_cases`), the frame loop (`frames.optimize_frames`) and the shim's solve
reproduce them.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import beam_oracle as bo
from oracle import frame_sizing_oracle as fso
from oracle import sizing_oracle as so
from tests.helpers import assert_stop_epochs_agree, relerr

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GENERATORS = ("sc", "mc", "gpu", "sc_rb", "mc_rb")
# what each script's main() hands generate_sample as `patience` (SC passes its module-level 5, MC passes nothing and
# gets the signature's default 10, GPU passes its module-level 100)
PATIENCE = {"sc": 5, "mc": 10, "gpu": 100, "sc_rb": 5, "mc_rb": 10}


def load(kind):
    z = np.load(os.path.join(GOLD, f"sizing_reference_{kind}.npz"))
    return z, json.loads(str(z["constants"]))


def case_lists(z, i):
    nr, nf = int(z["n_rollers"][i]), int(z["n_forces"][i])
    return (z["node_positions"][i], z["roller_nodes"][i, :nr].tolist(), z["force_nodes"][i, :nf].tolist(), z["force_values"][i, :nf].tolist())


def replay(ops, log, upto_analyze=True):
    """Feeds a recorded command log (list of [name, *args]) to a module with the OpenSees command names.  Returns analyze's code."""
    rc = None
    for name, *args in log:
        if name == "analyze":
            if not upto_analyze:
                break
            rc = ops.analyze(*args)
        else:
            getattr(ops, name)(*args)
    return rc


# ---------------------------------------------------------------------------------------------------------------------------------
# CPU: the fixtures themselves, the oracles against them, the command log into the shim's recorder
# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", GENERATORS)
def test_fixture_is_a_set_of_reference_records(kind):
    """Shapes and invariants of what the script's main() collected (SingleCore.py:73-87, :235-249)."""
    z, c = load(kind)
    n = int(z["n"])
    assert n == int(z["n_drawn"]) >= 6                              # no sample failed, none was filtered (MC:265)
    assert z["I_values"].shape == (n, 100) and z["I_values"].dtype == np.float32
    assert z["deflections"].shape == z["rotations"].shape == z["node_positions"].shape == (n, 101)
    assert (z["num_nodes"] == 101).all() and (z["epochs_run"] >= 100).all() and (z["epochs_run"] <= c["max_e"]).all()
    assert (z["I_values"] >= np.float32(1e-8)).all()                # the clamp (SC:208)
    fv = z["force_values"][z["force_nodes"] > 0]
    assert fv.min() >= c["max_force"] and fv.max() <= c["min_force"]            # SC:160
    if kind.endswith("_rb"):
        assert int(z["random_bridge"]) == 1 and (z["L"] >= 15).all() and (z["L"] <= 215).all() and len(set(z["L"].tolist())) == n
        assert np.allclose(z["node_positions"][:, -1], z["L"])
    else:
        assert (z["roller_nodes"] == np.array([10, 30, 70, 85, 100])).all() and (z["L"] == 200.0).all()
        assert np.array_equal(z["roller_x_locations"][0], [18.0, 58.0, 138.0, 168.0, 198.0])
    if kind.startswith("mc"):                                       # MC:222-223: the last node's rotation / deflection forced to 0.0
        assert (z["deflections"][:, -1] == 0.0).all() and (z["rotations"][:, -1] == 0.0).all()
    else:
        assert (z["deflections"][:, -1] != 0.0).all()
    # the recorded responses belong to the solve BEFORE the last Adam step: every supported node sits still in them
    for i in range(n):
        _, rollers, _, _ = case_lists(z, i)
        assert abs(z["deflections"][i, 0]) == 0.0 and all(z["deflections"][i, r - 1] == 0.0 for r in rollers if r < 101 or not kind.startswith("mc"))


@pytest.mark.parametrize("kind", GENERATORS)
def test_sizing_oracle_reproduces_the_reference_runs(kind):
    """oracle/sizing_oracle.py (re-typed from SC:163-249, C band solve) against what the reference's code did: epoch counts equal,
    every epoch's loss and the final float32 inertias equal to float32 round-off (the two FE formulations differ by ~1e-12, which the
    float32 rounding of M / V (SC:189-190) almost always absorbs), recorded responses to 1e-9."""
    z, c = load(kind)
    n = int(z["n"])
    same = 0
    for i in range(n):
        x, rollers, fn, fv = case_lists(z, i)
        ref = so.generate_sample(x, rollers, fn, fv, E=c["E"], udl=c["uniform_udl"], I_0=c["I_0"], max_e=c["max_e"], lr=c["lr"],
                                 gamma=c["gamma"], alpha_moment=c["alpha_moment"], alpha_shear=c["alpha_shear"], tolerance=c["tolerance"],
                                 patience=PATIENCE[kind], zero_last_node=kind.startswith("mc"))
        ep = int(z["epochs_run"][i])
        assert ref["epochs_run"] == ep, (i, ref["epochs_run"], ep)
        np.testing.assert_allclose(ref["loss_history"], z["loss_history"][i, :ep], rtol=1e-6)
        I = np.array(ref["I_values"], dtype=np.float32)
        np.testing.assert_allclose(I, z["I_values"][i], rtol=2e-5, atol=1e-9)
        same += int(np.array_equal(I, z["I_values"][i]))
        assert relerr(np.array(ref["bending_moments"]), z["bending_moments"][i]) < 1e-6
        assert relerr(np.array(ref["shear_forces"]), z["shear_forces"][i]) < 1e-6
        assert relerr(np.array(ref["deflections"]), z["deflections"][i]) < 1e-8
        assert relerr(np.array(ref["rotations"]), z["rotations"][i]) < 1e-8
        assert ref["roller_x_locations"] == z["roller_x_locations"][i, :len(rollers)].tolist()
        assert ref["force_x_locations"] == z["force_x_locations"][i, :len(fn)].tolist()
    assert same >= 0.8 * n, f"only {same} of {n} cases bit-equal in I_values"


def test_beam_opt_script_against_the_oracle():
    """OpenPyStruct_BeamOpt.py (one case per run): its roller draw honours the 15-node spacing, 5 loads in [0.5, 1] x max_force, and
    the oracle loop with ITS constants (UDL -5000, tolerance 1e-2, patience 10, 1000 epochs) reproduces its loss history and inertias.
    The script reads its final responses AFTER the loop from the last epoch's model (BO:262-268): same one-step lag."""
    z, c = load("bo")
    assert c["uniform_udl"] == -5000 and c["num_epochs"] == 1000 and c["patience"] == 10 and c["tolerance"] == 1e-2
    x = np.linspace(0, c["L"], c["num_nodes"])
    for i in range(int(z["n"])):
        r = sorted(z["roller_nodes"][i].tolist())
        assert len(r) == 5 and min(np.diff(r)) >= c["L_min"] and len(set(z["force_nodes"][i].tolist())) == 5          # BO:57-79
        assert (z["force_values"][i] <= 0.5 * c["max_force"]).all() and (z["force_values"][i] >= c["max_force"]).all()
        ref = so.generate_sample(x, z["roller_nodes"][i].tolist(), z["force_nodes"][i].tolist(), z["force_values"][i].tolist(), E=c["E"],
                                 udl=c["uniform_udl"], I_0=c["I_0"], max_e=c["num_epochs"], lr=c["lr"], gamma=c["gamma"],
                                 alpha_moment=c["alpha_moment"], alpha_shear=c["alpha_shear"], tolerance=c["tolerance"], patience=c["patience"])
        ep = int(z["epochs_run"][i])
        assert ref["epochs_run"] == ep
        np.testing.assert_allclose(ref["loss_history"], z["loss_total"][i, :ep], rtol=1e-6)
        np.testing.assert_allclose(np.array(ref["I_values"], dtype=np.float32), z["I_values"][i], rtol=2e-5, atol=1e-9)
        assert relerr(np.array(ref["deflections"]), z["deflections"][i]) < 1e-8
        np.testing.assert_allclose(z["loss_total"][i, :ep], (z["loss_primary"] + z["loss_bending"] + z["loss_shear"])[i, :ep], rtol=1e-6)


def _frame_model(nb, ns, c):
    nb1 = nb + 1
    coords = np.array([(j * c["bay_width"], i * c["story_height"]) for i in range(ns + 1) for j in range(nb1)])
    cols = [(i * nb1 + j, (i + 1) * nb1 + j) for i in range(ns) for j in range(nb1)]
    beams = [(i * nb1 + j, i * nb1 + j + 1) for i in range(1, ns + 1) for j in range(nb)]
    conn = np.array(cols + beams)
    fix3 = np.zeros((len(coords), 3), dtype=np.int64)
    fix3[coords[:, 1] == 0.0] = 1
    loads = np.zeros((len(coords), 3))
    loads[(coords[:, 0] == 0.0) & (coords[:, 1] != 0.0), 0] = c["lateral_load"]
    w = np.zeros(len(conn)); w[len(cols):] = c["vertical_load"]
    return coords, conn, fix3, loads, w


def test_frame_sizing_oracle_reproduces_the_reference_runs():
    """oracle/frame_sizing_oracle.py (re-typed from FR:141-206) against the FrameOpt script's own runs: epoch count, loss history,
    final inertias, and the re-run analysis with the optimised inertias (FR:213-219)."""
    z, c = load("fr")
    sizes = set()
    for i in range(int(z["n"])):
        p = f"run{i}/"
        nb, ns = int(z[p + "num_bays"]), int(z[p + "num_stories"])
        sizes.add((nb, ns))
        coords, conn, fix3, loads, w = _frame_model(nb, ns, c)
        ep = int(z[p + "epochs_run"])
        if ep > 400:          # keep the CPU suite short: runs that go on for thousands of epochs (most never meet the 1e-3 / 10-epoch stop before
            # num_epochs = 5000) are checked on their first 150 epochs here; the -m gpu test follows them to the end
            ref = fso.optimize_frame(coords, conn, fix3, loads, w, w, A=c["A"], E=c["E"], nu=c["nu"], I0=c["I0"], alpha_moment=c["alpha_moment"],
                                     alpha_shear=c["alpha_shear"], k=c["k"], num_epochs=150, lr=c["lr"], tolerance=c["tolerance"], patience=c["patience"])
            np.testing.assert_allclose(ref["loss_history"], z[p + "loss_history"][:150], rtol=2e-6)
            continue
        ref = fso.optimize_frame(coords, conn, fix3, loads, w, w, A=c["A"], E=c["E"], nu=c["nu"], I0=c["I0"], alpha_moment=c["alpha_moment"],
                                 alpha_shear=c["alpha_shear"], k=c["k"], num_epochs=c["num_epochs"], lr=c["lr"], tolerance=c["tolerance"],
                                 patience=c["patience"])
        assert ref["epochs_run"] == ep, (i, nb, ns, ref["epochs_run"], ep)
        np.testing.assert_allclose(ref["loss_history"], z[p + "loss_history"], rtol=2e-6)
        np.testing.assert_allclose(ref["I"], z[p + "I_values"], rtol=5e-5, atol=1e-9)
        d, f, st, n_eq, kd = bo.solve_model_3dof(coords, conn, c["A"], c["E"], z[p + "I_values"].astype(np.float64), fix3, loads, wy=w, wx=w)
        assert st == 0 and n_eq == int(z[p + "n_eq"]) and kd == int(z[p + "kd"])
        assert relerr(f, z[p + "forces"]) < 1e-12 and relerr(d, z[p + "disp"]) < 1e-12
    assert len(sizes) >= 4


@pytest.mark.parametrize("kind", ["sc", "gpu", "sc_rb", "fr"])
def test_command_log_replays_into_the_shim_recorder(kind):
    """The command sequence the reference's setup_model / setup_frame_model ISSUED (recorded as data) is accepted by the product's
    `ops` module call for call, and leaves the model the reference built: geometry, supports, inertias, loads."""
    from openpystruct_amd import ops
    z, c = load(kind)
    log = json.loads(str(z["command_log"]))
    names = [e[0] for e in log]
    assert names[-1] == "analyze" and names.count("analyze") == 1 and "wipe" in names and names[names.index("analysis") + 1 if kind != "fr" else -1] == "analyze"
    replay(ops, log, upto_analyze=False)
    d = ops._dom
    assert d.analysis == "Static"
    if kind == "fr":
        nb, ns = int(z["run0/num_bays"]), int(z["run0/num_stories"])
        coords, conn, fix3, loads, w = _frame_model(nb, ns, c)
        assert not ops._is_straight_beam(d) or ns == 0
        a = ops._frame_arrays(d)
        np.testing.assert_array_equal(a["coords"], coords); np.testing.assert_array_equal(a["conn"], conn)
        np.testing.assert_array_equal(a["fix3"].astype(np.int64), fix3); np.testing.assert_array_equal(a["loads"], loads)
        np.testing.assert_array_equal(a["wy"], w); np.testing.assert_array_equal(a["wx"], w)       # beamUniform(w, w): FR:131
        assert (a["I"] == np.float64(np.float32(c["I0"]))).all() and (a["A"] == c["A"]).all()
        return
    assert ops._is_straight_beam(d)
    a = ops._arrays(d)
    x, rollers, fn, fv = case_lists(z, 0)
    np.testing.assert_array_equal(a["x"], x)
    fix = np.zeros(101, dtype=np.uint8); fix[0] = 1; fix[np.array(rollers) - 1] = 1
    np.testing.assert_array_equal(a["fix"], fix)
    Fy = np.zeros(101); Fy[np.array(fn) - 1] = fv
    np.testing.assert_array_equal(a["Fy"], Fy)
    assert (a["I"] == 0.5).all() and (a["E"] == c["E"]).all() and (a["wy"] == c["uniform_udl"]).all() and (a["wx"] == c["uniform_udl"]).all()
    assert len(log) == 1 + 1 + 101 + 1 + len(rollers) + 1 + 100 + 2 + len(fn) + 100 + 5 + 1 + 1 + (1 if kind == "gpu" else 0)   # wipe, model, nodes, fixes, transf, elements, ...


def test_cases_from_lists_builds_the_recorded_cases():
    from openpystruct_amd import sizing
    z, c = load("sc_rb")
    n = int(z["n"])
    lists = [case_lists(z, i) for i in range(n)]
    cases = sizing.cases_from_lists([l[0] for l in lists], [l[1] for l in lists], [l[2] for l in lists], [l[3] for l in lists])
    assert len(cases) == n and cases.roller_nodes == [l[1] for l in lists] and cases.force_nodes == [l[2] for l in lists]
    assert cases.force_values == [l[3] for l in lists] and torch.equal(cases.L, torch.as_tensor(z["L"]))
    for i in range(n):
        fix = np.zeros(101, dtype=np.uint8); fix[0] = 1; fix[np.array(lists[i][1]) - 1] = 1
        assert (cases.fix[i].numpy() == fix).all() and float(cases.Fy[i].sum()) == pytest.approx(sum(lists[i][3]))
    with pytest.raises(ValueError):
        sizing.cases_from_lists(z["node_positions"][0], [[10]], [[5, 6]], [[-1.0]])
    with pytest.raises(ValueError):
        sizing.cases_from_lists(z["node_positions"][0], [[10]], [[102]], [[-1.0]])


# ---------------------------------------------------------------------------------------------------------------------------------
# GPU: the HIP path against the reference's runs
# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def oa():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    import openpystruct_amd as oa_
    from openpystruct_amd import _cabi
    _cabi.load()
    return oa_


def _config(sizing, kind, c):
    return sizing.SizingConfig(E=c["E"], nu=c["nu"], A=c["A"], uniform_udl=float(c["uniform_udl"]), I_0=c["I_0"], max_e=c["max_e"], lr=c["lr"],
                               gamma=c["gamma"], alpha_moment=c["alpha_moment"], alpha_shear=c["alpha_shear"], tolerance=c["tolerance"],
                               patience=PATIENCE[kind], random_bridge=int(kind.endswith("_rb")), zero_last_node=kind.startswith("mc"))


# what the HIP loop is held to against the reference's own runs (r06: set from the RECORDED deviations, profiles/r06_sizing_deviation.json --
# 5 x the worst case of the 60 runs, rounded up; r05 held I to 2e-3 of its maximum and v / theta to 5e-3 without having measured them)
TOL_I, TOL_MV, TOL_VT, TOL_LOSS, TOL_LOSS_20 = 2e-4, 2e-4, 5e-3, 1.5e-4, 5e-6      # recorded worst: 4.2e-5, 4.3e-5, 1.5e-3 (theta, random bridges), 2.9e-5, 3.1e-6


def _compare_with_reference_runs(st, z, n, lag_tol, tolerance, patience, record=None):
    """Stopping epochs equal -- or within `patience` where the reference's own history has a stop decision inside float32 round-off of
    its threshold (sums are ordered differently on the GPU: such a decision can fall either way, tests/helpers.py); for EVERY case the
    loss history over the common prefix; records where the counts coincide.  Returns that number.  `record`: a dict that receives the ACHIEVED
    maximum deviations (what the tolerances above are set from)."""
    ep = st.epochs_run.cpu().numpy()
    I, hist = st.I.cpu().numpy(), st.loss_history.cpu().numpy()
    V32, M32 = st.V32.cpu().numpy(), st.M32.cpu().numpy()
    v, th = st.sol.v.cpu().numpy(), st.sol.theta.cpu().numpy()
    assert int(st.active.sum()) == 0 and int(st.sol.status.abs().sum()) == 0
    matched = 0
    dev = {"I_rel_to_max": 0.0, "I_ulps_max": 0.0, "I_bit_equal_cases": 0, "M32": 0.0, "V32": 0.0, "v": 0.0, "theta": 0.0, "loss_prefix": 0.0, "loss_first20": 0.0}
    for i in range(n):
        ref_ep = int(z["epochs_run"][i])
        assert_stop_epochs_agree(ep[i], ref_ep, z["loss_history"][i, :ref_ep], tolerance, patience, what=i)
        m = min(int(ep[i]), ref_ep)
        ref_h = z["loss_history"][i, :m]
        dev["loss_prefix"] = max(dev["loss_prefix"], float((np.abs(hist[:m, i] - ref_h) / np.abs(ref_h)).max()))
        dev["loss_first20"] = max(dev["loss_first20"], float((np.abs(hist[:20, i] - z["loss_history"][i, :20]) / np.abs(z["loss_history"][i, :20])).max()))
        np.testing.assert_allclose(hist[:m, i], ref_h, rtol=TOL_LOSS)                                  # all of the common prefix, every case
        np.testing.assert_allclose(hist[:20, i], z["loss_history"][i, :20], rtol=TOL_LOSS_20)          # the first epochs: float32 round-off only
        if int(ep[i]) != ref_ep:
            continue
        matched += 1
        Iref = z["I_values"][i].astype(np.float64)
        Iref32 = z["I_values"][i].astype(np.float32)
        dev["I_rel_to_max"] = max(dev["I_rel_to_max"], float(np.abs(I[i] - Iref).max() / Iref.max()))
        dev["I_ulps_max"] = max(dev["I_ulps_max"], float((np.abs(I[i].astype(np.float32) - Iref32) / np.spacing(Iref32)).max()))
        dev["I_bit_equal_cases"] += int(np.array_equal(I[i].astype(np.float32), Iref32))
        dev["M32"] = max(dev["M32"], relerr(M32[i], z["bending_moments"][i])); dev["V32"] = max(dev["V32"], relerr(V32[i], z["shear_forces"][i]))
        dev["v"] = max(dev["v"], relerr(v[i], z["deflections"][i])); dev["theta"] = max(dev["theta"], relerr(th[i], z["rotations"][i]))
        assert np.abs(I[i] - Iref).max() / Iref.max() < TOL_I
        assert relerr(M32[i], z["bending_moments"][i]) < lag_tol and relerr(V32[i], z["shear_forces"][i]) < lag_tol
        assert relerr(v[i], z["deflections"][i]) < TOL_VT and relerr(th[i], z["rotations"][i]) < TOL_VT
    if record is not None:
        record.update(dev, cases=n, stopped_at_the_reference_epoch=matched)
    return matched


@pytest.mark.gpu
@pytest.mark.parametrize("kind", GENERATORS)
def test_hip_sizing_loop_reproduces_the_reference_runs(oa, kind):
    """`sizing.optimize_cases` (fused HIP epoch kernel: solve + loss + Adam + early stop) on the cases the reference drew, against the
    records the reference's own generate_sample / main() produced for them."""
    from openpystruct_amd import sizing
    z, c = load(kind)
    n = int(z["n"])
    lists = [case_lists(z, i) for i in range(n)]
    cases = sizing.cases_from_lists([l[0] for l in lists], [l[1] for l in lists], [l[2] for l in lists], [l[3] for l in lists])
    cfg = _config(sizing, kind, c)
    st = sizing.optimize_cases(cases, cfg, "cuda", record_loss=True)
    if cfg.zero_last_node:                   # what generate_dataset does with the flag (MC:222-223)
        st.sol.v[:, -1] = 0.0; st.sol.theta[:, -1] = 0.0
    rec = {}
    matched = _compare_with_reference_runs(st, z, n, TOL_MV, cfg.tolerance, cfg.patience, record=rec)
    # (how many stop at the reference's very epoch depends on how many of its stop decisions sat within float32 round-off of the threshold:
    #  ~90 % with patience 5, 75-90 % with patience 10 on the flat loss tails of random bridges; every other case is covered above by its
    #  common prefix and by the marginal-decision check.  The counts go to gpurun_out/ for the notes.)
    os.makedirs(os.path.join(os.path.dirname(GOLD), "..", "gpurun_out"), exist_ok=True)
    with open(os.path.join(os.path.dirname(GOLD), "..", "gpurun_out", f"sizing_golden_matched_{kind}.json"), "w") as f:
        json.dump({"kind": kind, "cases": n, "stopped_at_the_reference_epoch": matched, "achieved_max_deviation": rec,
                   "epochs_gpu": st.epochs_run.cpu().tolist(), "epochs_reference": z["epochs_run"].tolist()}, f)
    assert matched >= 0.7 * n, f"only {matched} of {n} cases stopped at the reference's epoch"
    # the graph-replayed loop the generator uses gives the same records as the launch-by-launch loop just checked
    st2 = sizing.optimize_cases(cases, cfg, "cuda", poll_every=25)
    assert torch.equal(st2.epochs_run, st.epochs_run) and torch.equal(st2.I, st.I)


@pytest.mark.gpu
def test_hip_sizing_loop_reproduces_the_beam_opt_script(oa):
    from openpystruct_amd import sizing
    z, c = load("bo")
    n = int(z["n"])
    x = np.linspace(0, c["L"], c["num_nodes"])
    cases = sizing.cases_from_lists(x, [z["roller_nodes"][i].tolist() for i in range(n)], [z["force_nodes"][i].tolist() for i in range(n)],
                                    [z["force_values"][i].tolist() for i in range(n)])
    cfg = sizing.SizingConfig.beam_opt()
    assert (cfg.uniform_udl, cfg.max_e, cfg.tolerance, cfg.patience, cfg.lr, cfg.gamma) == (c["uniform_udl"], c["num_epochs"], c["tolerance"], c["patience"], c["lr"], c["gamma"])
    st = sizing.optimize_cases(cases, cfg, "cuda", record_loss=True)
    zz = dict(epochs_run=z["epochs_run"], loss_history=z["loss_total"], I_values=z["I_values"], bending_moments=z["bending_moments"],
              shear_forces=z["shear_forces"], deflections=z["deflections"], rotations=z["rotations"])
    rec = {}
    matched = _compare_with_reference_runs(st, zz, n, TOL_MV, cfg.tolerance, cfg.patience, record=rec)
    with open(os.path.join(os.path.dirname(GOLD), "..", "gpurun_out", "sizing_golden_matched_bo.json"), "w") as f:
        json.dump({"kind": "bo", "cases": n, "stopped_at_the_reference_epoch": matched, "achieved_max_deviation": rec}, f)
    assert matched >= n - 2


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["sc", "gpu", "sc_rb", "mc_rb", "bo_like_fr"])
def test_command_log_through_the_shim_gives_the_stubs_answers(oa, kind):
    """Row b: the call sequence the reference's code issued, replayed through `openpystruct_amd.ops` (one HIP launch behind analyze),
    answers what the recorder answered the reference: all 6 end forces of every element, all 3 displacements of every node."""
    from openpystruct_amd import ops
    z, c = load("fr" if kind == "bo_like_fr" else kind)
    log = json.loads(str(z["command_log"]))
    assert replay(ops, log) == 0
    F, D = z["log_forces"], z["log_disp"]
    got_f = np.array([ops.eleResponse(e + 1, "forces") for e in range(F.shape[0])])
    got_d = np.array([[ops.nodeDisp(nd + 1, k) for k in (1, 2, 3)] for nd in range(D.shape[0])])
    scale = np.abs(F).max(axis=0, keepdims=True)
    assert np.abs(got_f - F).max() / np.abs(F).max() < 1e-9 and (np.abs(got_f - F) / np.maximum(scale, 1e-30)).max() < 1e-7
    assert np.abs(got_d[:, 1:] - D[:, 1:]).max() / np.abs(D[:, 1:]).max() < 1e-9          # deflections, rotations (SC:226-230)
    assert np.abs(got_d[:, 0] - D[:, 0]).max() <= 1e-9 * max(np.abs(D[:, 0]).max(), 1e-30)  # axial displacement (Wx = Wy quirk, SC:117)


@pytest.mark.gpu
def test_hip_frame_loop_reproduces_the_frame_opt_script(oa):
    """`frames.optimize_frames` against the FrameOpt script's own runs (FR:163-206): epochs, loss history, inertias, and the final
    analysis with the optimised inertias (FR:213-219) through `frame_solve`."""
    from openpystruct_amd import frames
    z, c = load("fr")
    cfg = frames.FrameConfig(bay_width=c["bay_width"], story_height=c["story_height"], E=c["E"], nu=c["nu"], A=c["A"], I0=c["I0"],
                             alpha_moment=c["alpha_moment"], alpha_shear=c["alpha_shear"], k=c["k"], lateral_load=c["lateral_load"],
                             vertical_load=c["vertical_load"], num_epochs=c["num_epochs"], lr=c["lr"], tolerance=c["tolerance"], patience=c["patience"])
    matched = 0
    n = int(z["n"])
    for i in range(n):
        p = f"run{i}/"
        topo = frames.grid_frame(int(z[p + "num_bays"]), int(z[p + "num_stories"]), cfg, "cuda")
        # (the recorder numbers node by node; the product takes reverse Cuthill-McKee where that is narrower -- run 5, 10 x 2: 35 -> 8)
        assert topo.n_eq == int(z[p + "n_eq"]) and (topo.kd == int(z[p + "kd"]) if topo.numbering == "node" else topo.kd < int(z[p + "kd"]))
        hist = []
        I, sol, ep = frames.optimize_frames(topo, 2, cfg, poll_every=10, loss_history=hist)
        ref_ep = int(z[p + "epochs_run"])
        assert int(ep[0]) == int(ep[1])
        assert_stop_epochs_agree(ep[0], ref_ep, z[p + "loss_history"], cfg.tolerance, cfg.patience, what=i)
        h = torch.stack(hist).cpu().numpy()[:, 0]
        m = min(int(ep[0]), ref_ep)
        np.testing.assert_allclose(h[:m], z[p + "loss_history"][:m], rtol=5e-4)
        np.testing.assert_allclose(h[:10], z[p + "loss_history"][:10], rtol=2e-5)
        # the script's final analysis: the optimised inertias through the solve
        Iopt = torch.as_tensor(z[p + "I_values"].astype(np.float64), device="cuda")[None, :].contiguous()
        s2 = frames.frame_solve(topo, Iopt)
        assert int(s2.status[0]) == 0
        assert relerr(s2.forces[0].cpu().numpy(), z[p + "forces"]) < 1e-7 and relerr(s2.disp[0].cpu().numpy(), z[p + "disp"]) < 1e-7
        if int(ep[0]) == ref_ep:
            matched += 1
            Iref = z[p + "I_values"].astype(np.float64)
            assert np.abs(I[0].cpu().numpy() - Iref).max() / Iref.max() < 5e-3
    assert matched >= n - 2


def test_the_live_openseespy_test_file_runs_against_the_recorder():
    """tests/test_openseespy_live.py is skipped wherever the wheel is absent -- i.e. everywhere so far.  Run here with the recorder standing
    in for the module, it cannot pin anything (the recorder's solve IS the oracle) but it proves that the file's command sequences are
    well formed, that its read-outs index what it thinks they index, and that its singular-model cases come back as codes: the day a box
    has OpenSeesPy, what fails there is a difference of arithmetic, not a typo."""
    import importlib
    import sys
    sys.path.insert(0, GOLD)
    import opensees_stub as stub
    names = ("openseespy", "openseespy.opensees", "tests.test_openseespy_live")
    saved = {k: sys.modules.get(k) for k in names}
    try:
        stub.install()
        sys.modules.pop("tests.test_openseespy_live", None)
        live = importlib.import_module("tests.test_openseespy_live")
        assert live.ops is stub
        for seed in range(6):
            live.test_oracle_equals_openseespy_on_reference_cases(seed)
        live.test_singular_model_returns_a_code()
        for bays, stories in [(1, 1), (3, 2), (10, 10)]:
            live.test_frame_oracle_equals_openseespy(bays, stories)
        live.test_frame_mechanism_returns_a_code()
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
