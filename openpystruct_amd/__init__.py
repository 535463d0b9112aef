"""openpystruct_amd -- MI355X-native batched beam FE solve (the OpenPyStruct data-generation hot path).

Host side in Python over a C-ABI HIP library; see DESIGN.md and INTEGRATION.md.
"""
from . import runtime  # noqa: F401  (entry points call runtime.configure() before their first HIP call; importing sets nothing)
from .beam import BeamSolution, beam_solve, kernel_name  # noqa: F401
from . import torch_op  # noqa: F401  (registers torch.ops.openpystruct_amd.beam_solve)

__all__ = ["BeamSolution", "beam_solve", "kernel_name"]
