"""GPU tests of the callers either side of the solve: the OpenSees-command shim (per-case API) and the
batched sizing loop / dataset generator, against the per-case CPU restatement (oracle/sizing_oracle.py)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import beam_oracle as bo  # noqa: E402
from oracle import sizing_oracle as so  # noqa: E402
from tests.helpers import assert_stop_epochs_agree, relerr  # noqa: E402
from tests.test_host_logic import setup_model  # noqa: E402


@pytest.fixture(scope="module")
def oa():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    import openpystruct_amd as oa_
    from openpystruct_amd import _cabi
    _cabi.load()
    return oa_


def test_ops_shim_drop_in_for_the_reference_epoch(oa):
    """One epoch body of generate_sample (SingleCore.py:176-190, :224-232) through the shim."""
    from openpystruct_amd import ops
    x = np.linspace(0, 200, 101)
    I = np.full(100, 0.5)
    ops.wipe()
    setup_model(I, x, bo.ROLLERS_REF, [50, 20], [-355857.0, -100000.0], bo.A_REF, bo.E_REF, -1000.0)
    ops.analysis('Static')
    assert ops.analyze(1) == 0
    M = np.array([ops.eleResponse(i, 'forces')[2] for i in range(1, 101)])
    V = np.array([ops.eleResponse(i, 'forces')[1] for i in range(1, 101)])
    rot = np.array([ops.nodeDisp(i, 3) for i in range(1, 102)])
    defl = np.array([ops.nodeDisp(i, 2) for i in range(1, 102)])
    Fy = np.zeros(101); Fy[49] = -355857.0; Fy[19] = -100000.0
    v, th, Vr, Mr, st = bo.solve_beam_dense(x, bo.E_REF, I, bo.reference_fix_mask(), Fy, -1000.0)
    assert relerr(defl, v) < 1e-9 and relerr(rot, th) < 1e-9 and relerr(V, Vr) < 1e-8 and relerr(M, Mr) < 1e-8
    np.testing.assert_allclose(V[:3], [36417.18, 34417.18, 32417.18], rtol=1e-6)   # SURVEY Appendix E probe
    assert defl.min() == pytest.approx(-1.634e-2, rel=1e-3)


def test_ops_shim_failure_code_and_deferred_batch(oa):
    from openpystruct_amd import ops
    x = np.linspace(0, 10, 11)
    ops.wipe()
    setup_model(np.full(10, -0.1), x, [11], [5], [-1.0], 0.01, 2e11, 0.0)     # negative inertia: not SPD
    ops.analysis('Static')
    assert ops.analyze(1) != 0                                               # a code, not an exception (MultiCore.py:182-186)
    rng = np.random.default_rng(0)
    with ops.deferred() as batch:
        Is = []
        for _ in range(7):
            I = np.exp(rng.uniform(np.log(3e-3), np.log(0.75), size=10))
            Is.append(I)
            ops.wipe()
            setup_model(I, x, [11], [5], [-1000.0], 0.01, 2e11, -10.0)
            ops.analysis('Static')
            assert ops.analyze(1) == 0
    assert batch.codes == [0] * 7
    for d, I in zip(batch.domains, Is):
        fix = np.zeros(11, dtype=np.uint8); fix[0] = fix[10] = 1
        Fy = np.zeros(11); Fy[4] = -1000.0
        v, th, V, M, st = bo.solve_beam_dense(x, 2e11, I, fix, Fy, -10.0)
        assert relerr(d.result["v"], v) < 1e-9 and relerr(d.result["forces"][:, 2], M) < 1e-8


def test_ops_shim_results_of_one_analyze_survive_the_next(oa):
    """The batch-of-one path reuses one pinned staging block per mesh size (r05): what `analyze` hands to a domain must be a copy, so that
    results read after a later `analyze` -- of another model of the same size, or of the same model with other inertias, the reference's
    epoch loop -- are still the earlier model's; and a model of another size gets its own staging."""
    from openpystruct_amd import ops
    x = np.linspace(0, 10, 11)
    fix = np.zeros(11, dtype=np.uint8); fix[0] = fix[10] = 1
    Fy = np.zeros(11); Fy[4] = -1000.0
    rng = np.random.default_rng(5)
    kept = []
    for k in range(4):
        I = np.exp(rng.uniform(np.log(3e-3), np.log(0.75), size=10))
        ops.wipe()
        setup_model(I, x, [11], [5], [-1000.0], 0.01, 2e11, -10.0)
        ops.analysis('Static')
        assert ops.analyze(1) == 0
        kept.append((I, ops._dom.result))
        if k == 1:                                         # another mesh size in between
            ops.wipe()
            setup_model(np.full(20, 0.1), np.linspace(0, 10, 21), [21], [7], [-5.0], 0.01, 2e11, 0.0)
            ops.analysis('Static')
            assert ops.analyze(1) == 0 and len(ops._dom.result["v"]) == 21
    for I, res in kept:
        v, th, V, M, st = bo.solve_beam_dense(x, 2e11, I, fix, Fy, -10.0)
        assert relerr(res["v"], v) < 1e-9 and relerr(res["th"], th) < 1e-9 and relerr(res["forces"][:, 2], M) < 1e-8


@pytest.mark.parametrize("patience,n", [(5, 32), (10, 6)])       # 32 cases at patience 5: BASELINE config 1 (SingleCore.py:257)
def test_sizing_loop_vs_per_case_oracle(oa, patience, n):
    """Batched loop (HIP solve + HIP optimiser step) vs the reference's per-case torch-CPU loop restated in
    oracle/sizing_oracle.py.  float32 optimiser arithmetic: sums are ordered differently on the GPU, so the
    trajectories agree to float32 round-off accumulated over ~250 Adam steps, not bit for bit."""
    from openpystruct_amd import sizing
    cfg = sizing.SizingConfig(patience=patience)
    cases = sizing.make_cases(n, cfg, seed=123)
    st = sizing.optimize_cases(cases, cfg, "cuda", poll_every=20, record_loss=True)
    I = st.I.cpu().numpy(); ep = st.epochs_run.cpu().numpy()
    V32 = st.V32.cpu().numpy(); M32 = st.M32.cpu().numpy()
    v = st.sol.v.cpu().numpy(); th = st.sol.theta.cpu().numpy()
    assert int(st.active.sum()) == 0
    hist = st.loss_history.cpu().numpy()
    matched = 0
    for b in range(n):
        ref = so.generate_sample(cases.node_positions[b].numpy(), cases.roller_nodes[b], cases.force_nodes[b],
                                 cases.force_values[b], patience=patience)
        assert_stop_epochs_agree(ep[b], ref["epochs_run"], ref["loss_history"], cfg.tolerance, cfg.patience, what=b)
        # EVERY case: the loss of every epoch both sides ran (a case whose count differs by one is no longer checked for the count alone)
        m = min(int(ep[b]), ref["epochs_run"])
        np.testing.assert_allclose(hist[:m, b], np.array(ref["loss_history"])[:m], rtol=2e-4)
        if int(ep[b]) == ref["epochs_run"]:
            matched += 1
            Iref = np.array(ref["I_values"])
            assert np.abs(I[b] - Iref).max() / Iref.max() < 2e-3
            assert relerr(M32[b], np.array(ref["bending_moments"])) < 2e-3      # one-step lag state (SingleCore.py:239-241)
            assert relerr(v[b], np.array(ref["deflections"])) < 5e-3
            assert relerr(th[b], np.array(ref["rotations"])) < 5e-3
    assert matched >= 0.8 * n, f"only {matched} of {n} cases stopped at the oracle's epoch"
    assert 150 < ep.mean() < 450                                               # ~237-255 epochs per sample


def test_gpu_script_variant_vs_per_case_oracle(oa):
    """`SizingConfig.gpu_script()` = OpenPyStruct_BeamOpt_training_GPU.py:50-51 (tolerance 1e-2, patience 100): with a patience
    of 100 the loop runs to within a few epochs of max_e = 600 or stops ~100 epochs after the last 1e-2 improvement."""
    from openpystruct_amd import sizing
    cfg = sizing.SizingConfig.gpu_script()
    assert cfg.tolerance == 1e-2 and cfg.patience == 100
    n = 3
    cases = sizing.make_cases(n, cfg, seed=321)
    st = sizing.optimize_cases(cases, cfg, "cuda", poll_every=25)
    ep = st.epochs_run.cpu().numpy()
    assert int(st.active.sum()) == 0
    for b in range(n):
        ref = so.generate_sample(cases.node_positions[b].numpy(), cases.roller_nodes[b], cases.force_nodes[b],
                                 cases.force_values[b], patience=cfg.patience, tolerance=cfg.tolerance)
        assert_stop_epochs_agree(ep[b], ref["epochs_run"], ref["loss_history"], cfg.tolerance, cfg.patience, what=b)
        assert ref["epochs_run"] > 250          # patience 100: far beyond the ~250 epochs of the patience-5 script
        if int(ep[b]) == ref["epochs_run"]:
            Iref = np.array(ref["I_values"])
            assert np.abs(st.I[b].cpu().numpy() - Iref).max() / Iref.max() < 3e-3


def test_first_epochs_match_oracle_tightly(oa):
    """Three optimiser steps from the common start I = 0.5: float32-level agreement with torch's Adam."""
    from openpystruct_amd import sizing
    cfg = sizing.SizingConfig(max_e=3)
    cases = sizing.make_cases(4, cfg, seed=77)
    st = sizing.optimize_cases(cases, cfg, "cuda", poll_every=1, use_graph=False)
    assert (st.epochs_run.cpu().numpy() == 3).all() and int(st.active.sum()) == 0
    for b in range(4):
        ref = so.generate_sample(cases.node_positions[b].numpy(), cases.roller_nodes[b], cases.force_nodes[b],
                                 cases.force_values[b], max_e=3)
        np.testing.assert_allclose(st.I[b].cpu().numpy(), np.array(ref["I_values"]), rtol=2e-6)
        np.testing.assert_allclose(st.last_loss[b].item(), ref["final_loss"], rtol=2e-6)
        # recorded responses belong to the solve BEFORE the last Adam step
        assert relerr(st.M32[b].cpu().numpy(), np.array(ref["bending_moments"])) < 1e-6
        assert relerr(st.sol.v[b].cpu().numpy(), np.array(ref["deflections"])) < 1e-6


def test_generate_dataset_records_and_json(oa, tmp_path):
    from openpystruct_amd import sizing
    cfg = sizing.SizingConfig(max_e=40)
    rec = sizing.generate_dataset(32, cfg, "cuda", seed=5)                      # BASELINE config 1: 32 cases
    assert rec["I_values"].shape == (32, 100) and rec["I_values"].dtype == torch.float32
    assert rec["deflections"].shape == (32, 101) and rec["deflections"].dtype == torch.float64
    assert int(rec["status"].abs().sum()) == 0 and (rec["epochs_run"] > 0).all()
    # shards reproduce the same records at any GPU count
    a = sizing.generate_dataset(32, cfg, "cuda", seed=5, rank=1, world=4)
    lo, hi = sizing.shard_range(32, 1, 4)
    assert torch.equal(a["I_values"], rec["I_values"][lo:hi]) and torch.equal(a["deflections"], rec["deflections"][lo:hi])
    path = str(tmp_path / "training_data_PINN_mini.json")
    assert sizing.records_to_reference_json(rec, path) == 32
    d = json.load(open(path))
    assert tuple(d) == sizing.RECORD_KEYS and len(d["shear_forces"]) == 32 and len(d["rotations"][0]) == 101
    assert d["roller_x_locations"][0] == [18.0, 58.0, 138.0, 168.0, 198.0] and 1 <= len(d["force_values"][0]) <= 4
    assert len(d["force_x_locations"][3]) == len(d["force_nodes"][3]) == int(rec["n_forces"][3])
    # MultiCore quirk: last node zeroed (MultiCore.py:222-223)
    z = sizing.generate_dataset(4, sizing.SizingConfig(max_e=5, zero_last_node=True), "cuda", seed=5)
    assert float(z["deflections"][:, -1].abs().max()) == 0.0 and float(z["rotations"][:, -1].abs().max()) == 0.0


def test_second_dataset_with_another_line_load_is_not_solved_with_the_cached_one(oa):
    """generate_dataset re-arms a cached state + epoch graph when shape and hyper-parameters repeat (SizingState.reset); the line load
    (SingleCore.py:32, :117) is a device scalar of that state and not a hyper-parameter: a second run that differs only in it must
    be solved with ITS load."""
    from openpystruct_amd import sizing
    a = sizing.generate_dataset(64, sizing.SizingConfig(max_e=30), "cuda", seed=3)
    b = sizing.generate_dataset(64, sizing.SizingConfig(max_e=30, uniform_udl=-20000.0), "cuda", seed=3)
    a2 = sizing.generate_dataset(64, sizing.SizingConfig(max_e=30), "cuda", seed=3)
    assert torch.equal(a["I_values"], a2["I_values"]) and torch.equal(a["deflections"], a2["deflections"])
    assert not torch.equal(a["deflections"], b["deflections"])
    # ... and that load is the one the per-case oracle is given (max_e = 30: every case runs all 30 epochs on both sides)
    for i in (0, 17, 63):
        nr, nf = int(b["n_rollers"][i]), int(b["n_forces"][i])
        ref = so.generate_sample(b["node_positions"][i].cpu().numpy(), b["roller_nodes"][i, :nr].tolist(), b["force_nodes"][i, :nf].tolist(),
                                 b["force_values"][i, :nf].tolist(), udl=-20000.0, max_e=30)
        assert int(b["epochs_run"][i]) == ref["epochs_run"]
        assert relerr(b["deflections"][i].cpu().numpy(), np.array(ref["deflections"])) < 5e-3
        assert relerr(a["deflections"][i].cpu().numpy(), np.array(ref["deflections"])) > 2e-2

def test_beam_opt_variant_against_oracle(oa):
    """OpenPyStruct_BeamOpt.py's single-case optimiser (UDL -5000, 5 spaced rollers, 5 loads in [0.5, 1] * max,
    tolerance 1e-2, patience 10) as a batch; per-beam supports with shared geometry."""
    from openpystruct_amd import sizing
    cfg = sizing.SizingConfig.beam_opt()
    cases = sizing.make_beam_opt_cases(5, cfg, seed=11)
    for b in range(5):
        r = sorted(cases.roller_nodes[b])
        assert len(r) == 5 and min(np.diff(r)) >= 15 and len(cases.force_nodes[b]) == 5
        assert all(0.5 * cfg.max_force >= f >= cfg.max_force for f in cases.force_values[b])
    st = sizing.optimize_cases(cases, cfg, "cuda", poll_every=20)
    assert st.fix.dim() == 2 and st.x.dim() == 1           # supports per beam, geometry shared
    ep = st.epochs_run.cpu().numpy()
    for b in range(2):
        ref = so.generate_sample(cases.node_positions[b].numpy(), cases.roller_nodes[b], cases.force_nodes[b], cases.force_values[b],
                                 udl=cfg.uniform_udl, max_e=cfg.max_e, tolerance=cfg.tolerance, patience=cfg.patience)
        assert_stop_epochs_agree(ep[b], ref["epochs_run"], ref["loss_history"], cfg.tolerance, cfg.patience, what=b)
        if int(ep[b]) == ref["epochs_run"]:
            Iref = np.array(ref["I_values"])
            assert np.abs(st.I[b].cpu().numpy() - Iref).max() / Iref.max() < 3e-3


def test_forces_only_entry_points_match_the_full_solve_and_honour_the_mask(oa):
    """ops_beam_solve_forces_f64 / _f32: same V, M as the full solve (f32: rounded), inactive wavefronts untouched."""
    import ctypes
    from openpystruct_amd import _cabi
    lib = _cabi.load()
    rng = np.random.default_rng(5)
    B, Ne, N = 37, 100, 101
    I, Fy = bo.random_cases(rng, B, inertia="trajectory")
    dev = "cuda"
    x = torch.linspace(0, 200, N, dtype=torch.float64, device=dev)
    fix = torch.as_tensor(bo.reference_fix_mask(), device=dev)
    It, Ft = torch.as_tensor(I, device=dev), torch.as_tensor(Fy, device=dev)
    E = torch.tensor(bo.E_REF, dtype=torch.float64, device=dev); wy = torch.tensor(bo.UDL_REF, dtype=torch.float64, device=dev)
    ref = oa.beam_solve(x, E, It, fix, Ft, wy)
    active = torch.ones(B, dtype=torch.uint8, device=dev); active[16:32] = 0       # wavefronts 4..7 of the 16-lane tiling: all inactive
    for f32 in (False, True):
        dt = torch.float32 if f32 else torch.float64
        V = torch.full((B, Ne), -7.0, dtype=dt, device=dev); M = torch.full((B, Ne), -7.0, dtype=dt, device=dev)
        st = torch.zeros(B, dtype=torch.int32, device=dev)
        fn = lib.ops_beam_solve_forces_f32 if f32 else lib.ops_beam_solve_forces_f64
        rc = fn(B, Ne, x.data_ptr(), 0, E.data_ptr(), 0, It.data_ptr(), Ne, fix.data_ptr(), 0, Ft.data_ptr(), N, wy.data_ptr(), 0,
                V.data_ptr(), M.data_ptr(), st.data_ptr(), active.data_ptr(), 16, torch.cuda.current_stream().cuda_stream)
        assert rc == _cabi.OK
        torch.cuda.synchronize()
        live = torch.ones(B, dtype=torch.bool, device=dev); live[16:32] = False
        assert torch.equal(V[live], ref.V[live].to(dt)) and torch.equal(M[live], ref.M[live].to(dt))
        assert bool((V[~live] == -7.0).all()) and bool((M[~live] == -7.0).all())


def test_records_do_not_depend_on_the_shard_size_across_the_tiling_threshold(oa):
    """40 000 cases on one GPU (a batch the plain solve would tile with 8 lanes) vs the same cases as two shards of
    20 000: the epoch kernel and the final solve use one tiling, so the records are bitwise the same."""
    from openpystruct_amd import sizing
    cfg = sizing.SizingConfig(max_e=12)
    whole = sizing.generate_dataset(40000, cfg, "cuda", seed=9)
    for r in range(2):
        part = sizing.generate_dataset(40000, cfg, "cuda", seed=9, rank=r, world=2)
        lo, hi = sizing.shard_range(40000, r, 2)
        for k in ("I_values", "shear_forces", "bending_moments", "deflections", "rotations", "epochs_run"):
            assert torch.equal(part[k], whole[k][lo:hi]), k


def test_random_bridges_through_the_sizing_loop(oa, monkeypatch):
    """random_bridge = 1 (SingleCore.py:133-151): per-case span, node coordinates and supports -- the per-beam-geometry
    variants of the fused epoch kernel -- against the per-sample oracle, and against the two-launch epoch."""
    from openpystruct_amd import sizing
    cfg = sizing.SizingConfig(random_bridge=1, max_e=4)
    cases = sizing.make_cases(9, cfg, seed=31)
    st = sizing.optimize_cases(cases, cfg, "cuda", poll_every=1, use_graph=False)
    torch.cuda.synchronize()
    assert int(st.sol.status.abs().sum()) == 0 and (st.epochs_run.cpu().numpy() == 4).all()
    for b in range(9):
        ref = so.generate_sample(cases.node_positions[b].numpy(), cases.roller_nodes[b], cases.force_nodes[b],
                                 cases.force_values[b], max_e=4)
        np.testing.assert_allclose(st.I[b].cpu().numpy(), np.array(ref["I_values"]), rtol=5e-6)
        assert relerr(st.M32[b].cpu().numpy(), np.array(ref["bending_moments"])) < 1e-6
        assert relerr(st.sol.v[b].cpu().numpy(), np.array(ref["deflections"])) < 1e-6
    monkeypatch.setattr(sizing, "_FUSED_EPOCH", False)
    st2 = sizing.optimize_cases(cases, cfg, "cuda", poll_every=1, use_graph=False)
    assert torch.equal(st2.epochs_run, st.epochs_run) and torch.allclose(st2.I, st.I, rtol=1e-6, atol=0.0)


@pytest.mark.parametrize("num_nodes", [41, 129, 151, 301])
def test_other_mesh_sizes_through_the_sizing_loop(oa, num_nodes):
    """Ne <= 128 runs the fused epoch kernel, larger meshes the two-launch epoch: both against the per-sample oracle."""
    from openpystruct_amd import sizing
    rollers = tuple(r for r in (10, 30, 70, 85, 100) if r < num_nodes - 1) or (num_nodes // 2,)
    cfg = sizing.SizingConfig(num_nodes=num_nodes, roller_nodes=rollers, max_e=3)
    cases = sizing.make_cases(5, cfg, seed=num_nodes)
    st = sizing.optimize_cases(cases, cfg, "cuda", poll_every=1, use_graph=False)
    torch.cuda.synchronize()
    assert int(st.sol.status.abs().sum()) == 0
    for b in range(5):
        ref = so.generate_sample(cases.node_positions[b].numpy(), cases.roller_nodes[b], cases.force_nodes[b],
                                 cases.force_values[b], max_e=3)
        np.testing.assert_allclose(st.I[b].cpu().numpy(), np.array(ref["I_values"]), rtol=5e-6)
        assert relerr(st.sol.v[b].cpu().numpy(), np.array(ref["deflections"])) < 1e-6


def test_chunked_generation_resumes_and_tiles_the_same_dataset(oa, tmp_path):
    """generate_dataset_to_files: chunk files, atomic replace, resume skips finished chunks, chunks == the one-shot dataset."""
    from openpystruct_amd import sizing
    cfg = sizing.SizingConfig(max_e=8)
    whole = sizing.generate_dataset(500, cfg, "cuda", seed=12)
    d = str(tmp_path / "chunks")
    files = sizing.generate_dataset_to_files(500, d, cfg, "cuda", seed=12, chunk=200)
    assert [os.path.basename(f) for f in files] == ["records_000000000_000000200.pt", "records_000000200_000000400.pt",
                                                     "records_000000400_000000500.pt"]
    mtimes = [os.path.getmtime(f) for f in files]
    os.remove(files[1])                                    # "crash" after the first chunk of a second run
    files2 = sizing.generate_dataset_to_files(500, d, cfg, "cuda", seed=12, chunk=200)
    assert files2 == files and os.path.getmtime(files[0]) == mtimes[0] and os.path.getmtime(files[2]) == mtimes[2]
    rec = sizing.concat_records([sizing.load_records(f, device="cuda") for f in files])
    for k in ("I_values", "deflections", "bending_moments", "force_nodes", "epochs_run", "case_ids"):
        assert torch.equal(rec[k], whole[k].to(rec[k].device)), k


def test_cases_drawn_on_the_gpu_distribution_determinism_and_ranges():
    """csrc/case_draw.hip: the draws of SingleCore.py:133-160 as one launch -- same checks as the host-side list
    (tests/test_host_logic.py), plus: a range of the list does not depend on what else is drawn."""
    from openpystruct_amd import sizing
    cfg = sizing.SizingConfig()
    a = sizing.make_cases(3000, cfg, seed=1, device="cuda")
    b = sizing.make_cases(3000, cfg, seed=1, device="cuda")
    assert torch.equal(a.Fy, b.Fy) and a.force_nodes == b.force_nodes
    c = sizing.make_cases(3000, cfg, seed=2, device="cuda")
    assert not torch.equal(a.Fy, c.Fy)
    k = np.array([len(f) for f in a.force_nodes])
    assert k.min() >= 1 and k.max() <= 4 and set(k) == {1, 2, 3, 4}
    assert abs(np.bincount(k)[1:] / len(k) - 0.25).max() < 0.04                      # SC:157: randint(1, 4)
    vals = np.concatenate([np.array(f) for f in a.force_values])
    assert vals.min() >= cfg.max_force and vals.max() <= cfg.min_force
    assert abs(vals.mean() - 0.5 * (cfg.max_force + cfg.min_force)) < 0.02 * abs(cfg.min_force - cfg.max_force)
    forbidden = set(cfg.roller_nodes) | {1, cfg.num_nodes}
    assert all(not (set(f) & forbidden) and len(set(f)) == len(f) for f in a.force_nodes)
    nodes = np.concatenate([np.array(f) for f in a.force_nodes])
    cnt = np.bincount(nodes, minlength=cfg.num_nodes + 1)[2:cfg.num_nodes]
    free = np.array([n not in forbidden for n in range(2, cfg.num_nodes)])
    assert (cnt[~free] == 0).all() and cnt[free].min() > 0.5 * cnt[free].mean()      # every free candidate is drawn, about evenly
    assert (a.fix.cpu().numpy() == bo.reference_fix_mask()[None, :]).all()
    b0 = np.zeros(cfg.num_nodes); b0[np.array(a.force_nodes[0]) - 1] = a.force_values[0]
    np.testing.assert_array_equal(a.Fy[0].cpu().numpy(), b0)
    assert torch.equal(a.node_positions, sizing.make_cases(4, cfg, seed=1, device="cpu").node_positions[:1].cuda().expand(3000, -1))
    # a range of the list: the same cases whatever else is drawn, whatever the list's length
    part = sizing.make_cases(100000, cfg, seed=1, device="cuda", lo=1000, hi=1500)
    assert torch.equal(part.Fy, a.Fy[1000:1500]) and part.force_nodes == a.force_nodes[1000:1500]
    # random_bridge = 1 (SC:133-151)
    cr = sizing.SizingConfig(random_bridge=1)
    d = sizing.make_cases(2000, cr, seed=3, device="cuda")
    assert float(d.L.min()) >= cr.L_min and float(d.L.max()) <= cr.L_min + cr.L_max
    nr = np.array([len(r) for r in d.roller_nodes])
    assert nr.min() >= 1 and nr.max() <= cr.N_rollers_max and set(nr) == set(range(1, cr.N_rollers_max + 1))
    assert all(len(set(r)) == len(r) and min(r) >= 2 and max(r) <= cr.num_nodes - 1 for r in d.roller_nodes)
    fx = d.fix.cpu().numpy()
    assert all(int(fx[i].sum()) == 1 + len(d.roller_nodes[i]) and fx[i, 0] == 1 and all(fx[i, n - 1] for n in d.roller_nodes[i]) for i in range(2000))
    assert all(not (set(f) & set(r)) for f, r in zip(d.force_nodes, d.roller_nodes))
    assert torch.allclose(d.node_positions[:, -1], d.L) and float(d.node_positions[:, 0].abs().max()) == 0.0
    unused = torch.arange(cr.N_rollers_max, device="cuda")[None, :] >= d.n_rollers[:, None]
    assert bool((d.roller_nodes_t[unused] == 0).all()) and bool((d.roller_nodes_t[~unused] > 0).all())


def test_generate_dataset_at_the_size_the_bench_times_against_the_per_case_oracle(oa):
    """bench.py times `generate_dataset(50 000)` (BASELINE config 3) while the largest oracle-checked sizing batch
    was 32 cases.  The full 50 000-case shard, 64 sampled records (first, last, seeded picks) against the per-case loop of
    oracle/sizing_oracle.py -- itself bit-equal to the reference's own generate_sample on the reference-run fixtures
    (tests/test_sizing_golden.py)."""
    from openpystruct_amd import sizing
    n = 50000
    cfg = sizing.SizingConfig()
    rec = sizing.generate_dataset(n, cfg, "cuda")
    assert int(rec["status"].abs().sum()) == 0 and rec["I_values"].shape == (n, 100)
    ep = rec["epochs_run"].cpu().numpy()
    assert ep.min() >= 50 and ep.max() <= cfg.max_e and 200 < ep.mean() < 320
    pick = sorted({0, 1, n - 2, n - 1} | {int(v) for v in np.random.default_rng(5).integers(0, n, size=60)})
    matched = 0
    for i in pick:
        nr, nf = int(rec["n_rollers"][i]), int(rec["n_forces"][i])
        ref = so.generate_sample(rec["node_positions"][i].cpu().numpy(), rec["roller_nodes"][i, :nr].tolist(), rec["force_nodes"][i, :nf].tolist(),
                                 rec["force_values"][i, :nf].tolist(), patience=cfg.patience)
        assert_stop_epochs_agree(ep[i], ref["epochs_run"], ref["loss_history"], cfg.tolerance, cfg.patience, what=i)
        if int(ep[i]) != ref["epochs_run"]:
            continue
        matched += 1
        Iref = np.array(ref["I_values"])
        assert np.abs(rec["I_values"][i].cpu().numpy() - Iref).max() / Iref.max() < 2e-3
        assert relerr(rec["bending_moments"][i].cpu().numpy(), np.array(ref["bending_moments"])) < 2e-3
        assert relerr(rec["deflections"][i].cpu().numpy(), np.array(ref["deflections"])) < 5e-3
        assert relerr(rec["rotations"][i].cpu().numpy(), np.array(ref["rotations"])) < 5e-3
    assert matched >= 0.8 * len(pick), f"only {matched} of {len(pick)} sampled cases stopped at the oracle's epoch"
