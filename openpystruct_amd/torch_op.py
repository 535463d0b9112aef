"""`torch.ops.openpystruct_amd.beam_solve`: the batched solve as a registered PyTorch operator (SURVEY.md 8b:
"called from (i) the torch.library custom op and (ii) the ops shim").

Registered for the GPU dispatch key only -- CPU tensors raise NotImplementedError, there is no CPU kernel --
plus a fake (meta) implementation, so the op can sit inside `torch.compile` / FakeTensor shape propagation and
HIP-graph capture without touching the kernel.  E and wy are tensors here (0-dim/1-element = shared scalar)."""
from __future__ import annotations

from typing import Tuple

import torch

from .beam import beam_solve

T = torch.Tensor


@torch.library.custom_op("openpystruct_amd::beam_solve", mutates_args=(), device_types="cuda")
def beam_solve_op(x: T, E: T, I: T, fix: T, Fy: T, wy: T, tiling: int = 0) -> Tuple[T, T, T, T, T]:
    s = beam_solve(x, E, I, fix, Fy, wy, tiling=tiling)
    return s.v, s.theta, s.V, s.M, s.status


@beam_solve_op.register_fake
def _(x, E, I, fix, Fy, wy, tiling=0):
    B, Ne = I.shape
    return (I.new_empty((B, Ne + 1)), I.new_empty((B, Ne + 1)), I.new_empty((B, Ne)), I.new_empty((B, Ne)),
            I.new_empty((B,), dtype=torch.int32))
