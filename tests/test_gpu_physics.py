"""FE-residual operator (HIP) and its vector-Jacobian product against the dense oracle stiffness matrix."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import beam_oracle as bo  # noqa: E402


def _case(B=5, seed=0):
    rng = np.random.default_rng(seed)
    x = np.linspace(0, 200, 101)
    fix = bo.reference_fix_mask()
    I, Fy = bo.random_cases(rng, B, inertia="trajectory")
    return x, fix, I, Fy


def test_residual_vanishes_at_the_fe_solution_and_matches_dense_K():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    import openpystruct_amd as oa
    from openpystruct_amd import physics
    x, fix, I, Fy = _case()
    t = lambda a, dt=torch.float64: torch.as_tensor(a, dtype=dt, device="cuda")  # noqa: E731
    sol = oa.beam_solve(t(x), t(bo.E_REF), t(I), t(fix, torch.uint8), t(Fy), t(bo.UDL_REF))
    rv, rt = physics.fe_residual(t(I), sol.v, sol.theta, t(x), t(bo.E_REF), t(fix, torch.uint8), t(Fy), t(bo.UDL_REF))
    fmax = float(np.abs(Fy).max())
    assert float(rv.abs().max()) < 1e-6 * fmax and float(rt.abs().max()) < 1e-6 * fmax      # equilibrium
    # arbitrary displacement field: compare with the dense oracle K u - f on the free DOFs
    rng = np.random.default_rng(1)
    v = rng.normal(size=Fy.shape) * 1e-2; th = rng.normal(size=Fy.shape) * 1e-3
    rv, rt = physics.fe_residual(t(I), t(v), t(th), t(x), t(bo.E_REF), t(fix, torch.uint8), t(Fy), t(bo.UDL_REF))
    for b in range(I.shape[0]):
        K, f = bo.assemble_beam(x, bo.E_REF, I[b], Fy[b], bo.UDL_REF)
        u = np.empty(202); u[0::2] = v[b]; u[1::2] = th[b]
        r = K @ u - f
        r[0::2][fix != 0] = 0.0
        np.testing.assert_allclose(rv[b].cpu().numpy(), r[0::2], rtol=1e-10, atol=1e-6 * np.abs(r).max())
        np.testing.assert_allclose(rt[b].cpu().numpy(), r[1::2], rtol=1e-10, atol=1e-6 * np.abs(r).max())


def test_vjp_matches_autograd_of_a_dense_reference():
    from openpystruct_amd import physics
    x, fix, I, Fy = _case(B=3, seed=2)
    t = lambda a, dt=torch.float64: torch.as_tensor(a, dtype=dt, device="cuda")  # noqa: E731
    rng = np.random.default_rng(3)
    v0 = rng.normal(size=Fy.shape) * 1e-2; th0 = rng.normal(size=Fy.shape) * 1e-3
    gv = rng.normal(size=Fy.shape); gt = rng.normal(size=Fy.shape)
    Iq, vq, tq = t(I).requires_grad_(), t(v0).requires_grad_(), t(th0).requires_grad_()
    rv, rt = physics.fe_residual(Iq, vq, tq, t(x), t(bo.E_REF), t(fix, torch.uint8), t(Fy), t(bo.UDL_REF))
    ((rv * t(gv)).sum() + (rt * t(gt)).sum()).backward()
    # dense torch reference on the CPU: K(I) assembled from the oracle's element matrix, autograd through it
    for b in range(3):
        Ic = torch.tensor(I[b], dtype=torch.float64, requires_grad=True)
        uc = torch.zeros(202, dtype=torch.float64); uc[0::2] = torch.tensor(v0[b]); uc[1::2] = torch.tensor(th0[b])
        uc.requires_grad_()
        K = torch.zeros(202, 202, dtype=torch.float64)
        for e in range(100):
            ke = torch.tensor(bo.element_stiffness(bo.E_REF, 2.0)) * Ic[e]      # EI = E * I, L = 2
            K[2 * e:2 * e + 4, 2 * e:2 * e + 4] = K[2 * e:2 * e + 4, 2 * e:2 * e + 4] + ke
        _, f = bo.assemble_beam(x, bo.E_REF, I[b], Fy[b], bo.UDL_REF)
        r = K @ uc - torch.tensor(f)
        mask = torch.ones(202, dtype=torch.float64); mask[0::2][torch.tensor(fix != 0)] = 0.0
        g = torch.zeros(202, dtype=torch.float64); g[0::2] = torch.tensor(gv[b]); g[1::2] = torch.tensor(gt[b])
        ((r * mask) * g).sum().backward()
        np.testing.assert_allclose(Iq.grad[b].cpu().numpy(), Ic.grad.numpy(), rtol=1e-9, atol=1e-9 * float(Ic.grad.abs().max()))
        np.testing.assert_allclose(vq.grad[b].cpu().numpy(), uc.grad[0::2].numpy(), rtol=1e-9, atol=1e-9 * float(uc.grad.abs().max()))
        np.testing.assert_allclose(tq.grad[b].cpu().numpy(), uc.grad[1::2].numpy(), rtol=1e-9, atol=1e-9 * float(uc.grad.abs().max()))


def test_residual_loss_is_zero_at_solution_and_decreases_under_gradient_descent():
    import openpystruct_amd as oa
    from openpystruct_amd import physics
    x, fix, I, Fy = _case(B=4, seed=5)
    t = lambda a, dt=torch.float64: torch.as_tensor(a, dtype=dt, device="cuda")  # noqa: E731
    sol = oa.beam_solve(t(x), t(bo.E_REF), t(I), t(fix, torch.uint8), t(Fy), t(bo.UDL_REF))
    args = (t(x), t(bo.E_REF), t(fix, torch.uint8), t(Fy), t(bo.UDL_REF))
    assert float(physics.fe_residual_loss(t(I), sol.v, sol.theta, *args)) < 1e-16
    v = (sol.v * 1.05).clone().requires_grad_(); th = (sol.theta * 0.97).clone().requires_grad_()
    l0 = physics.fe_residual_loss(t(I), v, th, *args)
    l0.backward()
    assert float(l0) > 1e-6 and torch.isfinite(v.grad).all() and float(v.grad.abs().max()) > 0


def test_pinn_training_with_the_fe_residual_term():
    """Per-case PINN (n_cases = 1) with the HIP FE-residual physics term switched on: loss finite, gradients flow."""
    from openpystruct_amd import dataprep, sizing, train
    cfg_s = sizing.SizingConfig(max_e=20)
    rec = sizing.generate_dataset(256, cfg_s, "cuda", seed=3)
    d = dataprep.prepare(rec, kind="pinn", n_cases=1, seed=0, device="cuda")
    assert d.Fy_train is not None and d.Fy_train.shape == (204, 101)
    cfg = train.PinnConfig(n_cases=1, batch_size=64)
    phys = train.PhysicsTerm(weight=1e-3, x=torch.linspace(0, 200, 101, dtype=torch.float64), E=cfg_s.E,
                             fix=torch.as_tensor(bo.reference_fix_mask()), wy=cfg_s.uniform_udl)
    out = train.train_surrogate("pinn", d, cfg, device="cuda", max_epochs=2, physics=phys)
    assert out["epochs"] == 2 and np.isfinite(out["history"]["train"]).all()
    base = train.train_surrogate("pinn", d, cfg, device="cuda", max_epochs=2)
    assert out["history"]["train"][0] != base["history"]["train"][0]       # the term is really in the loss


@pytest.mark.parametrize("kind", ["tfd", "fnn"])
def test_i_only_models_train_with_the_fe_residual_of_the_recorded_field(kind):
    """BASELINE config 4: the I-only surrogates take their physics loss from K(I_pred) u_recorded - f (HIP residual kernels)."""
    from openpystruct_amd import dataprep, physics, sizing, train
    cfg_s = sizing.SizingConfig(max_e=20)
    rec = sizing.generate_dataset(256, cfg_s, "cuda", seed=5)
    d = dataprep.prepare(rec, kind=kind, n_cases=1, seed=0, device="cuda")
    assert d.v_train is not None and d.v_train.shape == (204, 101) and d.v_train.dtype == torch.float64
    x = torch.linspace(0, 200, 101, dtype=torch.float64)
    fix = torch.as_tensor(bo.reference_fix_mask())
    # the recorded field belongs to the inertias one Adam step BEFORE the recorded ones (the reference's lag): zero
    # residual for those, and already a visible one for the recorded `I_values`
    I_rec = rec["I_solved"]
    args = (rec["deflections"], rec["rotations"], x.cuda(), cfg_s.E, fix.cuda(), None, cfg_s.uniform_udl)
    Fy = torch.zeros((256, 102), dtype=torch.float64, device="cuda").scatter_add_(1, rec["force_nodes"].long(), rec["force_values"].double())[:, 1:]
    l_true = float(physics.fe_residual_loss(I_rec, args[0], args[1], args[2], args[3], args[4], Fy, args[6]))
    l_wrong = float(physics.fe_residual_loss(rec["I_values"].double(), args[0], args[1], args[2], args[3], args[4], Fy, args[6]))
    assert l_true < 1e-6 * l_wrong
    cfg = (train.TfdConfig if kind == "tfd" else train.FnnConfig)(n_cases=1, batch_size=64)
    phys = train.PhysicsTerm(weight=1e-3, x=x, E=cfg_s.E, fix=fix, wy=cfg_s.uniform_udl)
    out = train.train_surrogate(kind, d, cfg, device="cuda", max_epochs=2, physics=phys)
    base = train.train_surrogate(kind, d, cfg, device="cuda", max_epochs=2)
    assert out["epochs"] == 2 and np.isfinite(out["history"]["train"]).all()
    assert out["history"]["train"][0] != base["history"]["train"][0]


@pytest.mark.parametrize("mode,dtype", [("recorded", torch.float32), ("predicted", torch.float32), ("recorded", torch.bfloat16), ("predicted", torch.bfloat16)])
def test_fused_residual_term_equals_the_framework_composition(mode, dtype):
    """r04: csrc/beam_residual.hip ops_physics_loss_fwd / _bwd (three launches) against what train.py built from physics.fe_residual_loss and
    ~60 framework ops: inverse scaler -> clamp -> float64 -> [row gathers] -> residual -> Jacobi scaling -> four means -> weight, and
    autograd's backward of all that.  Standardised predictions (some clamped at 1e-8), recorded displacement fields gathered by row
    (the I-only models) or predicted ones (the PINN's columns); value to 1e-6, the gradient w.r.t. the predictions to 1e-5 relative L2
    in float32 (bfloat16 predictions: the gradient's own bf16 rounding, 4e-3)."""
    from openpystruct_amd import physics
    from openpystruct_amd.dataprep import StandardScalerT
    dev = "cuda"
    x, fix, I, Fy = _case(B=40, seed=3)
    t = lambda a, dt=torch.float64: torch.as_tensor(a, dtype=dt, device=dev)  # noqa: E731
    G, nel, N = 40, 100, 101
    g = torch.Generator().manual_seed(5)
    sol_v = (torch.randn(G, N, generator=g) * 1e-2).double().to(dev)
    sol_t = (torch.randn(G, N, generator=g) * 1e-3).double().to(dev)
    sI, sD, sR = StandardScalerT(), StandardScalerT(), StandardScalerT()
    sI.fit(t(I).float()); sD.fit(sol_v.float()); sR.fit(sol_t.float())
    B = 24
    rows = torch.randperm(G, generator=g)[:B].to(dev)
    C = nel if mode == "recorded" else nel + 2 * N
    p32 = torch.randn(B, C, generator=g).to(dev)
    p32[:, :nel] = sI.transform(t(I).float()[rows]) * (1.0 + 0.05 * torch.randn(B, nel, generator=g).to(dev))
    p32[0, :7] = -50.0                                        # far below the clamp: inertia 1e-8, no gradient
    preds = p32.to(dtype).requires_grad_(True)
    ref = preds.detach().clone().requires_grad_(True)
    weight, E, wy = 1e-3, bo.E_REF, bo.UDL_REF
    # the framework composition (train.py before r04)
    pf = ref.float()
    I_p = sI.inverse_transform(pf[:, :nel]).clamp_min(1e-8)
    if mode == "recorded":
        v_p, t_p = sol_v[rows], sol_t[rows]
    else:
        v_p, t_p = sD.inverse_transform(pf[:, nel:nel + N]), sR.inverse_transform(pf[:, nel + N:])
    want = weight * physics.fe_residual_loss(I_p, v_p, t_p, t(x), t(E), t(fix, torch.uint8), t(Fy)[rows], t(wy)).float()
    want.backward()
    acc = torch.full((), 3.0, device=dev)
    disp = (sol_v, sol_t) if mode == "recorded" else (sD, sR)
    got = physics.fused_residual_term(preds, nel, sI, disp, rows, t(Fy), t(x), E, t(fix, torch.uint8), wy, weight, acc)
    got.backward(torch.ones((), device=dev))
    torch.cuda.synchronize()
    assert abs(float(got) - float(want)) <= 2e-6 * abs(float(want)) and abs(float(acc) - 3.0 - float(got)) <= 1e-6 * abs(float(got))
    gw, gg = ref.grad.double(), preds.grad.double()
    tol = 1e-5 if dtype == torch.float32 else 4e-3
    assert float((gg - gw).norm() / gw.norm()) < tol and float(gw.norm()) > 0
    assert float(gg[0, :7].abs().max()) == 0.0                 # clamped inertias
    if mode == "recorded":
        assert preds.grad.shape == (B, nel)
