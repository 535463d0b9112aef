"""Batched beam FE solve: Python host side above the C ABI.

`beam_solve` is the batched operator that replaces, for B cases at once, what the
reference does per case through OpenSeesPy: `setup_model` + `ops.analyze(1)` +
`ops.eleResponse(e,'forces')[1|2]` + `ops.nodeDisp(n, 2|3)`
(/root/reference/OpenPyStruct_BeamOpt_training_SingleCore.py:89-124, :180-190, :224-232).

PyTorch is used for device memory and streams only; the arithmetic is the HIP kernel in
csrc/beam_solve.hip.  There is no CPU path: tensors must live on the GPU.
"""
from __future__ import annotations

from typing import NamedTuple, Optional

import torch

from . import _cabi


class BeamSolution(NamedTuple):
    v: torch.Tensor        # [B, N]   u_y per node             (ops.nodeDisp(n, 2))
    theta: torch.Tensor    # [B, N]   theta_z per node         (ops.nodeDisp(n, 3))
    V: torch.Tensor        # [B, Ne]  eleResponse 'forces'[1]  ("shear_forces",   SingleCore.py:190)
    M: torch.Tensor        # [B, Ne]  eleResponse 'forces'[2]  ("bending_moments", SingleCore.py:189)
    status: torch.Tensor   # [B] int32, 0 = ok (like `ops.analyze(1)` == 0, MultiCore.py:182-186)


def _dev_f64(t, device, name):
    if not torch.is_tensor(t):
        t = torch.as_tensor(t, dtype=torch.float64, device=device)
    if t.dtype != torch.float64:
        raise TypeError(f"{name} must be float64 (OpenSees solves in double; float32 loses the 1e-6 parity bar)")
    if t.device != device:
        raise ValueError(f"{name} is on {t.device}, expected {device}")
    return t.contiguous()


TILING_STREAM_OUT = 0x100     # include/openpystruct_amd.h OPS_AMD_TILING_STREAM_OUT


def beam_solve(x, E, I, fix, Fy, wy, *, tiling: int = 0, out: Optional[BeamSolution] = None, stream_out: bool = False) -> BeamSolution:
    """Solve B straight Euler-Bernoulli beams (Ne elements, N = Ne + 1 nodes) on the GPU.

    x    [N] or [B,N]      node coordinates            I    [B,Ne]  element second moments of area
    E    scalar or [B,Ne]  Young's modulus             fix  [N] or [B,N] uint8, bit0 = u_y fixed, bit1 = theta_z fixed
    Fy   [B,N]             nodal point loads           wy   scalar or [B,Ne] transverse UDL (beamUniform Wy)

    Asynchronous on the current stream of I's device; `out` lets callers reuse result buffers; `stream_out`: non-temporal
    result stores, for callers cycling through more output than the 256 MiB Infinity Cache holds.
    """
    lib = _cabi.load()
    if not torch.is_tensor(I) or not I.is_cuda:
        raise RuntimeError("beam_solve needs GPU tensors: openpystruct_amd has no CPU fallback (I must be a CUDA/HIP tensor)")
    dev = I.device
    I = _dev_f64(I, dev, "I")
    if I.dim() != 2:
        raise ValueError("I must be [B, Ne]")
    B, Ne = I.shape
    N = Ne + 1
    x = _dev_f64(x, dev, "x")
    Fy = _dev_f64(Fy, dev, "Fy")
    E = _dev_f64(E, dev, "E")
    wy = _dev_f64(wy, dev, "wy")
    if not torch.is_tensor(fix):
        fix = torch.as_tensor(fix, dtype=torch.uint8, device=dev)
    if fix.dtype != torch.uint8 or fix.device != dev:
        raise TypeError("fix must be a uint8 tensor on the same device")
    fix = fix.contiguous()
    if x.shape not in ((N,), (B, N)):
        raise ValueError(f"x must be [{N}] or [{B},{N}], got {tuple(x.shape)}")
    if fix.shape not in ((N,), (B, N)):
        raise ValueError(f"fix must be [{N}] or [{B},{N}], got {tuple(fix.shape)}")
    if Fy.shape != (B, N):
        raise ValueError(f"Fy must be [{B},{N}], got {tuple(Fy.shape)}")
    if E.numel() != 1 and E.shape != (B, Ne):
        raise ValueError("E must be a scalar or [B, Ne]")
    if wy.numel() != 1 and wy.shape != (B, Ne):
        raise ValueError("wy must be a scalar or [B, Ne]")
    if out is None:
        out = BeamSolution(
            torch.empty((B, N), dtype=torch.float64, device=dev),
            torch.empty((B, N), dtype=torch.float64, device=dev),
            torch.empty((B, Ne), dtype=torch.float64, device=dev),
            torch.empty((B, Ne), dtype=torch.float64, device=dev),
            torch.empty((B,), dtype=torch.int32, device=dev),
        )
    else:       # caller-owned result buffers: raw pointers go to the kernel, so shape / dtype / device / layout are checked here
        for name, t, shape, dt in (("out.v", out.v, (B, N), torch.float64), ("out.theta", out.theta, (B, N), torch.float64),
                                   ("out.V", out.V, (B, Ne), torch.float64), ("out.M", out.M, (B, Ne), torch.float64),
                                   ("out.status", out.status, (B,), torch.int32)):
            if tuple(t.shape) != shape or t.dtype != dt or t.device != dev or not t.is_contiguous():
                raise ValueError(f"{name} must be a contiguous {dt} tensor of shape {shape} on {dev}, "
                                 f"got {tuple(t.shape)} {t.dtype} on {t.device}")
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = lib.ops_beam_solve_batched_f64(
            B, Ne,
            x.data_ptr(), N if x.dim() == 2 else 0,
            E.data_ptr(), Ne if E.numel() != 1 else 0,
            I.data_ptr(), Ne,
            fix.data_ptr(), N if fix.dim() == 2 else 0,
            Fy.data_ptr(), N,
            wy.data_ptr(), Ne if wy.numel() != 1 else 0,
            out.v.data_ptr(), out.theta.data_ptr(), out.V.data_ptr(), out.M.data_ptr(),
            out.status.data_ptr(), int(tiling) | (TILING_STREAM_OUT if stream_out else 0), stream,
        )
    if rc != _cabi.OK:
        raise RuntimeError(f"ops_beam_solve_batched_f64 failed with code {rc}: {lib.ops_amd_last_error().decode()}")
    return out


def kernel_name(B: int, Ne: int, tiling: int = 0) -> str:
    return _cabi.load().ops_beam_solve_kernel_name(B, Ne, tiling).decode()
