"""FE-residual physics loss for the surrogates (north_star: "physics-loss via HIP FE residual").

The reference's PINN has no physics operator: its "physics" is a relative-L1 fit of regressed deflections
and rotations (/root/reference/OpenPyStruct_PINN_MultiCase.py:646-652).  This module adds
the real thing as an optional term: for predicted inertias I and nodal displacements (v, theta) of a case,

    r = D (K(I) u - f)          (free-DOF equilibrium residual; same element / load / constraint semantics
                                 as the solver, HIP kernels in csrc/beam_residual.hip)

with a custom autograd Function whose backward is the matching HIP vector-Jacobian product
(dL/du = K D g, dL/dI_e = (D g)_e^T dk_e/dI_e u_e).  `fe_residual_loss` scales the residual row-wise by the
diagonal of K (Jacobi scaling), so that it is dimensionless and commensurate with the data terms.
"""
from __future__ import annotations

import torch

from . import _cabi


def _f64(t, dev):
    return torch.as_tensor(t, dtype=torch.float64, device=dev).contiguous()


class _FEResidual(torch.autograd.Function):
    @staticmethod
    def forward(ctx, I, v, theta, x, E, fix, Fy, wy):
        lib = _cabi.load()
        if not I.is_cuda:
            raise RuntimeError("fe_residual needs GPU tensors: openpystruct_amd has no CPU fallback")
        dev = I.device
        I, v, theta = (t.detach().to(torch.float64).contiguous() for t in (I, v, theta))
        if I.dim() != 2:
            raise ValueError("I must be [B, Ne]")
        B, Ne = I.shape
        N = Ne + 1
        for name, t, shape in (("v", v, (B, N)), ("theta", theta, (B, N)), ("Fy", Fy, (B, N))):
            if tuple(t.shape) != shape or t.device != dev or t.dtype != torch.float64 or not t.is_contiguous():
                raise ValueError(f"{name} must be a contiguous float64 [{shape[0]}, {shape[1]}] tensor on {dev}, got {tuple(t.shape)} {t.dtype} on {t.device}")
        for name, t in (("x", x), ("E", E), ("fix", fix), ("wy", wy)):
            if t.device != dev:
                raise ValueError(f"{name} is on {t.device}, expected {dev}")
        rv, rt = torch.empty_like(v), torch.empty_like(theta)
        with torch.cuda.device(dev):                  # the launch goes to I's device, whatever the current one is
            s = torch.cuda.current_stream(dev).cuda_stream
            rc = lib.ops_beam_residual_f64(B, Ne, x.data_ptr(), N if x.dim() == 2 else 0, E.data_ptr(), Ne if E.numel() != 1 else 0,
                                           I.data_ptr(), fix.data_ptr(), N if fix.dim() == 2 else 0, Fy.data_ptr(), wy.data_ptr(),
                                           Ne if wy.numel() != 1 else 0, v.data_ptr(), theta.data_ptr(), rv.data_ptr(), rt.data_ptr(), s)
        if rc != _cabi.OK:
            raise RuntimeError(f"ops_beam_residual_f64 failed with code {rc}")
        ctx.save_for_backward(I, v, theta, x, E, fix)
        return rv, rt

    @staticmethod
    def backward(ctx, gv, gt):
        lib = _cabi.load()
        I, v, theta, x, E, fix = ctx.saved_tensors
        dev = I.device
        B, Ne = I.shape
        N = Ne + 1
        gv, gt = gv.to(torch.float64).contiguous(), gt.to(torch.float64).contiguous()
        sv, st_ = torch.empty_like(v), torch.empty_like(v)
        dv, dt, dI = torch.empty_like(v), torch.empty_like(v), torch.empty_like(I)
        with torch.cuda.device(dev):
            s = torch.cuda.current_stream(dev).cuda_stream
            rc = lib.ops_beam_residual_vjp_f64(B, Ne, x.data_ptr(), N if x.dim() == 2 else 0, E.data_ptr(), Ne if E.numel() != 1 else 0,
                                               I.data_ptr(), fix.data_ptr(), N if fix.dim() == 2 else 0, v.data_ptr(), theta.data_ptr(),
                                               gv.data_ptr(), gt.data_ptr(), sv.data_ptr(), st_.data_ptr(), dv.data_ptr(), dt.data_ptr(),
                                               dI.data_ptr(), s)
        if rc != _cabi.OK:
            raise RuntimeError(f"ops_beam_residual_vjp_f64 failed with code {rc}")
        return dI, dv, dt, None, None, None, None, None


def fe_residual(I, v, theta, x, E, fix, Fy, wy):
    """r_v, r_theta [B,N] = D (K(I) [v; theta] - f); differentiable w.r.t. I, v, theta (float64 inside)."""
    dev = I.device
    return _FEResidual.apply(I, v, theta, _f64(x, dev), _f64(E, dev), torch.as_tensor(fix, dtype=torch.uint8, device=dev).contiguous(),
                             _f64(Fy, dev), _f64(wy, dev))


def stiffness_diagonal(I, x, E):
    """diag(K(I)) as (d_v, d_theta) [B,N] (torch ops; used for the Jacobi scaling of the residual)."""
    dev = I.device
    x, E = _f64(x, dev), _f64(E, dev)
    L = (x[..., 1:] - x[..., :-1])
    EI = E * I.to(torch.float64)
    k12, k4 = 12.0 * EI / L ** 3, 4.0 * EI / L
    z = torch.zeros_like(k12[:, :1])
    dv = torch.cat([k12, z], 1) + torch.cat([z, k12], 1)
    dth = torch.cat([k4, z], 1) + torch.cat([z, k4], 1)
    return dv, dth


def fe_residual_loss(I, v, theta, x, E, fix, Fy, wy):
    """mean over free DOFs of (r_i / K_ii)^2 relative to the mean squared displacement: dimensionless."""
    rv, rt = fe_residual(I, v, theta, x, E, fix, Fy, wy)
    dv, dth = stiffness_diagonal(I.detach(), x, E)
    ev, et = rv / dv, rt / dth                      # displacement-like errors
    scale = (v.detach().to(torch.float64) ** 2).mean() + 1e-30
    scale_t = (theta.detach().to(torch.float64) ** 2).mean() + 1e-30
    return (ev ** 2).mean() / scale + (et ** 2).mean() / scale_t


class _FusedResidualTerm(torch.autograd.Function):
    """csrc/beam_residual.hip ops_physics_loss_fwd / _bwd: weight * fe_residual_loss of the inertias (and, for the PINN, the displacement
    fields) the model PREDICTS in standardised form, in three launches; the value enters the differentiated total with weight one."""

    @staticmethod
    def forward(ctx, preds, spec):
        import ctypes
        lib = _cabi.load()
        (nel, sI, disp, rows, Fy, x, E, fix, wy, weight, acc) = spec
        dev = preds.device
        if preds.dtype not in (torch.float32, torch.bfloat16) or preds.dim() != 2 or preds.stride(1) != 1:
            raise ValueError("predictions must be a [B, C] float32 or bfloat16 matrix with unit column stride")
        B, N = preds.shape[0], nel + 1
        f64 = dict(dtype=torch.float64, device=dev)
        ev, et = torch.empty((B, N), **f64), torch.empty((B, N), **f64)
        part = torch.empty(int(lib.ops_physics_loss_part_doubles(B, nel)), **f64)
        value = torch.empty((), dtype=torch.float32, device=dev)
        f32 = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()      # noqa: E731
        keep = [f32(sI.scale_), f32(sI.mean_), Fy, x, fix, ev, et, part, value, preds]
        a = _cabi.PhysicsLossArgs(B=B, Ne=nel, preds=preds.data_ptr(), preds_bf16=int(preds.dtype == torch.bfloat16), ldp=preds.stride(0),
                                  I_scale=keep[0].data_ptr(), I_mean=keep[1].data_ptr(), I_min=1e-8, Fy=Fy.data_ptr(), x=x.data_ptr(),
                                  fix=fix.data_ptr(), E=float(E), wy=float(wy), weight=float(weight), ev=ev.data_ptr(), et=et.data_ptr(),
                                  part=part.data_ptr(), value=value.data_ptr(), value_sum=acc.data_ptr() if acc is not None else None)
        G = int(Fy.shape[0])
        if Fy.dim() != 2 or Fy.shape[1] != N or (rows is None and G < B):
            raise ValueError(f"Fy must be [G, {N}] float64 with G >= the batch (or `rows` given)")
        if rows is not None:
            if rows.dtype != torch.int64 or rows.device != dev or rows.dim() != 1 or rows.numel() < B or not rows.is_contiguous():
                raise ValueError("rows must be a contiguous int64 vector on the predictions' device with one entry per sample")
            keep.append(rows)
            a.rows = rows.data_ptr()
        if torch.is_tensor(disp[0]):            # recorded displacement fields (I-only models): the kernels read raw float64 rows
            for t in disp[:2]:
                if not (torch.is_tensor(t) and t.dtype == torch.float64 and t.device == dev and t.dim() == 2 and t.shape == (G, N) and t.is_contiguous()):
                    raise ValueError(f"recorded displacement fields must be contiguous float64 [{G}, {N}] tensors on the predictions' device")
            keep += [disp[0], disp[1]]
            a.v_rec, a.t_rec = disp[0].data_ptr(), disp[1].data_ptr()
        else:                                   # the PINN's own outputs: (scaler of the deflections, scaler of the rotations)
            sc = [f32(disp[0].scale_), f32(disp[0].mean_), f32(disp[1].scale_), f32(disp[1].mean_)]
            keep += sc
            a.v_scale, a.v_mean, a.t_scale, a.t_mean = (t.data_ptr() for t in sc)
        with torch.cuda.device(dev):
            rc = lib.ops_physics_loss_fwd(ctypes.byref(a), torch.cuda.current_stream(dev).cuda_stream)
        if rc != _cabi.OK:
            raise RuntimeError(f"ops_physics_loss_fwd failed with code {rc}")
        ctx.args, ctx.keep = a, keep
        ctx.ncols = nel if torch.is_tensor(disp[0]) else nel + 2 * N
        return value

    @staticmethod
    def backward(ctx, go):
        import ctypes
        lib = _cabi.load()
        a, preds = ctx.args, ctx.keep[9]
        dev = preds.device
        # (the launch reads the predictions and writes their gradient with ONE row stride: the gradient takes the predictions' strides -- a
        #  row view of a wider matrix included -- instead of empty_like's dense ones)
        dp = torch.empty_strided(preds.shape, preds.stride(), dtype=preds.dtype, device=dev)
        if ctx.ncols != preds.shape[1]:
            dp.zero_()
        a.dpreds, a.ldp = dp.data_ptr(), preds.stride(0)
        with torch.cuda.device(dev):
            rc = lib.ops_physics_loss_bwd(ctypes.byref(a), torch.cuda.current_stream(dev).cuda_stream)
        if rc != _cabi.OK:
            raise RuntimeError(f"ops_physics_loss_bwd failed with code {rc}")
        return dp, None


def fused_residual_term(preds, nel, sI, disp, rows, Fy, x, E, fix, wy, weight, acc=None):
    """weight * fe_residual_loss(I, v, theta, ...) with I = clamp(sI^-1(preds[:, :nel]), 1e-8) and the displacement fields either recorded
    (`disp` = (v_rec, t_rec) float64 [G, N], row rows[b] for sample b; rows None: row b) or predicted (`disp` = (scaler of the
    deflections, scaler of the rotations): the PINN's columns nel .. nel + 2 N - 1), Fy [G, N] gathered the same way: three launches.
    The value enters the total with weight ONE (the gradient ignores the incoming one); `acc`: float32 device scalar the value is added
    to.  Shared geometry: x [N] float64, fix [N] uint8, scalars E, wy."""
    if not preds.is_cuda:
        raise RuntimeError("fused_residual_term needs GPU tensors: openpystruct_amd has no CPU fallback")
    dev = preds.device
    spec = (int(nel), sI, disp, rows, _f64(Fy, dev), _f64(x, dev), float(E), torch.as_tensor(fix, dtype=torch.uint8, device=dev).contiguous(),
            float(wy), float(weight), acc)
    return _FusedResidualTerm.apply(preds, spec)
