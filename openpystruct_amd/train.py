"""Data-parallel training loops of the two surrogates (one process per GPU, RCCL all-reduce over xGMI).

Restates the reference's single-device loops
(/root/reference/OpenPyStruct_PINN_MultiCase.py:741-852,
 /root/reference/OpenPyStruct_TransformerDiffusionModule_MultiCase.py:722-829):
per epoch: decaying input noise, shuffled batches, in-batch permutation, autocast forward, loss,
backward, clip_grad_norm_(1.0), Adam step; then a validation pass, ExponentialLR step, early stopping on
the validation loss with best-state checkpointing; finally R^2 of the un-standardised inertias.

What is different, and why (DESIGN.md "Training loops"):
  * bf16 autocast without a GradScaler instead of fp16 + GradScaler (BASELINE config 4; MI355X bf16 MFMA) by default; the reference's own
    mode -- `autocast(float16)` + `GradScaler` (PINN:706, :759-768) -- is `train_surrogate(..., autocast_dtype=torch.float16)` (r06): framework
    modules under fp16 autocast, loss scaling with the GradScaler's rule (skip the step and halve the scale on a non-finite gradient, double it
    after 2 000 clean steps), eager steps (the scale and the skip decision change from step to step);
  * the dataset is resident on the GPU: no per-batch host-to-device copies (PINN:749-750) and no per-step
    `.item()` (PINN:770) -- losses are accumulated on the device and reduced once per epoch;
  * data parallelism by hand instead of the DistributedDataParallel wrapper: every parameter's .grad is a
    view into ONE flat float32 buffer (the models are 1.4 - 2.4 MB), which is all-reduced with a single
    RCCL call per step (latency-bound over xGMI; nothing to bucket or overlap at this size).  That keeps the
    step capturable: [noise, forward, loss, backward] and [scale, clip, Adam] are two HIP graphs with the one
    collective between them, so the multi-GPU step costs three launches, like the single-GPU one costs one.
    Weak scaling: every rank trains on its own shard with the reference's per-GPU batch size;
  * early-stop decisions are taken on all-reduced losses, so every rank stops at the same epoch.
BatchNorm batch statistics are per rank (as per process in the reference; pass sync_bn=True for SyncBatchNorm);
the running statistics are averaged over ranks at the end of every epoch so that the replicas stay identical.
"""
from __future__ import annotations

import copy
import os
import sys
import time
from dataclasses import dataclass
from typing import Dict, Optional

import torch

from . import switches  # noqa: E402
import torch.distributed as dist
import torch.nn as nn
from torch.optim.lr_scheduler import ExponentialLR

from .dataprep import SurrogateData
from .surrogates import fused_loss, ChainGNN, CompositeLoss, FNO1dModel, FNNPlain, FNNWithResidual, ModelOnePassTransformerWithDiffusion, TrainableL1L2Loss


@dataclass
class PinnConfig:
    """PINN:35-56."""
    n_cases: int = 6
    nelem: int = 100
    box_constraint_coeff: float = 1e-1
    hidden_units: int = 350
    dropout_rate: float = 0.5
    num_blocks: int = 2
    num_epochs: int = 500
    batch_size: int = 128
    patience: int = 10
    learning_rate: float = 5e-4
    weight_decay: float = 1e-3
    train_split: float = 0.8
    sigma_0: float = 0.01
    gamma_noise: float = 0.99
    gamma: float = 0.98
    initial_alpha: float = 0.5
    c: float = 0.5
    penalty_pinn: float = 1.5e-6


@dataclass
class TfdConfig:
    """TFD:36-60."""
    n_cases: int = 6
    nelem: int = 100
    box_constraint_coeff: float = 5e-1
    hidden_units: int = 256
    dropout_rate: float = 0.1
    num_epochs: int = 500
    batch_size: int = 512
    patience: int = 10
    learning_rate: float = 3e-3
    weight_decay: float = 1e-4
    train_split: float = 0.8
    sigma_0: float = 0.01
    gamma_noise: float = 0.90
    gamma: float = 0.95
    initial_alpha: float = 0.5
    c: float = 0.5
    num_transformer_layers: int = 2
    dim_feedforward: int = 256
    num_heads: int = 8
    max_len: int = 512
    diffusion_hidden_dim: int = 256
    diffusion_T: int = 512


@dataclass
class FnnConfig:
    """/root/reference/OpenPyStruct_FNN_MultiCase.py:35-52 (num_blocks = 3 is unused there: the model is built with 4)."""
    n_cases: int = 6
    nelem: int = 100
    box_constraint_coeff: float = 5e-1
    hidden_units: int = 128
    dropout_rate: float = 0.5
    num_residual_blocks: int = 4
    num_epochs: int = 500
    batch_size: int = 128
    patience: int = 10
    learning_rate: float = 2e-4
    weight_decay: float = 1e-2
    train_split: float = 0.8
    sigma_0: float = 0.03
    gamma_noise: float = 0.97
    gamma: float = 0.99
    initial_alpha: float = 0.5
    c: float = 1.0


_FUSED_LOSS = switches.get("fused_loss") == "1"      # A/B switch: 0 = the nn.Module losses


class FlatClipAdam:
    """`clip_grad_norm_(params, max_norm)` + `torch.optim.Adam(lr, weight_decay)` as two HIP launches over flat buffers
    (csrc/flat_adam.hip).  The parameters are re-homed into one flat float32 buffer (each `p.data` becomes a view of it),
    next to the flat gradient buffer the training loop already all-reduces; `lr` and the step count are device scalars,
    so a captured HIP graph follows the scheduler."""

    def __init__(self, params, flat_grad: torch.Tensor, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 max_norm: float = 1.0, decoupled: bool = False):
        from . import _cabi
        self._cabi, self._lib = _cabi, _cabi.load()
        dev = flat_grad.device
        self.p = torch.empty_like(flat_grad)
        off = 0
        with torch.no_grad():
            for q in params:
                n = q.numel()
                self.p[off:off + n].copy_(q.reshape(-1))
                q.data = self.p[off:off + n].view_as(q)
                off += n
        assert off == flat_grad.numel()
        self.g = flat_grad
        self.m, self.v = torch.zeros_like(flat_grad), torch.zeros_like(flat_grad)
        self.lr = torch.tensor(float(lr), dtype=torch.float32, device=dev)
        self.step_count = torch.zeros((), dtype=torch.int32, device=dev)
        self.ws = torch.empty(int(self._lib.ops_flat_adam_workspace_bytes()), dtype=torch.uint8, device=dev)
        self.betas, self.eps, self.weight_decay, self.max_norm, self.decoupled = betas, eps, weight_decay, max_norm, decoupled
        self.p_bf16: Optional[torch.Tensor] = None      # bfloat16 shadow of the parameters (enable_shadow)
        self.zero_grads = False     # zero the gradient buffer inside the update launch (the next step's zero_grad(): one fill node less)
        self.repack = None          # ctypes array of MlpRepackEntry: padded bf16 weight copies the update refreshes too (pinn_fused.py)
        self.norm_ready_parts = 0   # > 0: the gradient producers leave that many norm partial sums + the advanced step in `ws` (pinn_fused.enable_norm)

    def enable_shadow(self) -> torch.Tensor:
        """A bfloat16 copy of the flat parameter buffer that every step refreshes in the Adam kernel itself: layers that
        run their GEMMs in bfloat16 read it instead of casting their float32 weights in every forward pass."""
        if self.p_bf16 is None:
            self.p_bf16 = self.p.to(torch.bfloat16)
        return self.p_bf16

    def refresh_shadow(self) -> None:
        if self.p_bf16 is not None:
            self.p_bf16.copy_(self.p)

    def step(self, grad_scale: float = 1.0) -> None:
        dev = self.g.device
        args = (self.g.numel(), self.p.data_ptr(), self.g.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), self.lr.data_ptr(),
                self.step_count.data_ptr(), self.max_norm, grad_scale, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                int(self.decoupled) | (2 if self.zero_grads else 0) | ((self._cabi.ADAM_NORM_READY | (self.norm_ready_parts << 16)) if self.norm_ready_parts else 0),
                self.p_bf16.data_ptr() if self.p_bf16 is not None else None, self.ws.data_ptr())
        with torch.cuda.device(dev):
            s = torch.cuda.current_stream(dev).cuda_stream
            if self.repack is not None:
                rc = self._lib.ops_flat_clip_adam_step_repack_f32(*args, len(self.repack), self.repack, s)
            else:
                rc = self._lib.ops_flat_clip_adam_step_f32(*args, s)
        if rc != self._cabi.OK:
            raise RuntimeError(f"ops_flat_clip_adam_step_f32 failed with code {rc}")

    def state_dict(self):
        return {"m": self.m.clone(), "v": self.v.clone(), "step": self.step_count.clone(), "lr": self.lr.clone()}

    def load_state_dict(self, sd) -> None:
        self.m.copy_(sd["m"]); self.v.copy_(sd["v"]); self.step_count.copy_(sd["step"]); self.lr.copy_(sd["lr"])


@dataclass
class GnnConfig:
    """/root/reference/OpenPyStruct_GNN_MultiCase_Beta.py:38-55 (AdamW, no alpha term, :394, :437-446)."""
    n_cases: int = 6
    nelem: int = 100
    box_constraint_coeff: float = 5e-1
    encoder_hidden_dim: int = 128
    gnn_hidden_dim: int = 128
    num_gnn_layers: int = 2
    dropout_rate: float = 0.5
    num_epochs: int = 500
    batch_size: int = 512
    patience: int = 10
    learning_rate: float = 3e-3
    weight_decay: float = 1e-2
    train_split: float = 0.8
    sigma_0: float = 0.01
    gamma_noise: float = 0.99
    gamma: float = 0.975
    initial_alpha: float = 0.5
    c: float = 0.5


@dataclass
class FnoConfig:
    """/root/reference/OpenPyStruct_FNO_MultiCase_Beta.py:34-58 (Adam, autocast disabled, alpha term, :561, :613-617)."""
    n_cases: int = 6
    nelem: int = 100
    box_constraint_coeff: float = 5e-1
    hidden_units: int = 512
    dropout_rate: float = 0.1
    num_fno_layers: int = 4
    num_epochs: int = 500
    batch_size: int = 512
    patience: int = 10
    learning_rate: float = 3e-3
    weight_decay: float = 1e-6
    train_split: float = 0.8
    sigma_0: float = 0.01
    gamma_noise: float = 0.95
    gamma: float = 0.975
    initial_alpha: float = 0.5
    c: float = 0.5
    fno_modes: int = 4
    fno_width: int = 128


_DEBUG_NAN = os.environ.get("OPS_AMD_DEBUG_NAN", "0") == "1"
_EXPLICIT_ROOT = switches.get("explicit_root") == "1"      # A/B switch: 0 = loss.backward() with its implicit ones_like() fill node
_ROOT_ONES = {}


def _root_ones(device):
    key = (device.type, device.index)
    if key not in _ROOT_ONES:
        _ROOT_ONES[key] = torch.ones((), dtype=torch.float32, device=device)
    return _ROOT_ONES[key]


def _cabi_rows() -> int:
    from . import _cabi
    return _cabi.MLP_MAX_ROWS


def clear_blas_workspaces() -> None:
    """Drop the framework's cached BLAS workspaces at the END of a training run (the caller has synchronised).  One that was allocated
    while a HIP graph was being captured lives in that graph's private pool but stays cached per (handle, stream): a later run whose
    side stream gets the same id would hand the freed memory to its GEMMs as scratch space.  (Calling this at the START of a run as
    well made things worse -- 4 NaN runs of 12 against 0 of 12, profiles/r03_notes.md 8 -- so it is not.)"""
    clear = getattr(torch._C, "_cuda_clearCublasWorkspaces", None)
    if clear is not None:
        clear()


class _TieTerms(torch.autograd.Function):
    """Two scalar loss terms as one backward root without a launch: forward = a view of the first, backward = the root gradient to both."""

    @staticmethod
    def forward(ctx, a, b):
        return a.view_as(a)

    @staticmethod
    def backward(ctx, g):
        return g, g


def _debug_nan(model, opt, flat, s_loss, epoch, b, sX, sY):
    """Diagnostics (OPS_AMD_DEBUG_NAN=1; run with OPS_AMD_ADAM_ZERO=0 to keep the gradients): first step with a non-finite value."""
    torch.cuda.synchronize()
    bad_p = [n for n, q in model.named_parameters() if not bool(torch.isfinite(q).all())]
    bad_g = [n for n, q in model.named_parameters() if q.grad is not None and not bool(torch.isfinite(q.grad).all())]
    if bad_p or bad_g or not bool(torch.isfinite(s_loss)):
        print("DEBUG_NAN epoch", epoch, "batch", b, "loss", float(s_loss), "inputs finite", bool(torch.isfinite(sX.float()).all()), bool(torch.isfinite(sY).all()),
              "gradnorm", float(flat.norm()), "\n  bad grads", bad_g[:12], "\n  bad params", bad_p[:6], flush=True)
        for n, q in model.named_parameters():
            if n in bad_g:
                idx = (~torch.isfinite(q.grad)).nonzero().flatten()[:40].tolist()
                print("   ", n, tuple(q.shape), "non-finite at", idx, "values", q.grad.flatten()[idx[:6]].tolist(), flush=True)
        raise RuntimeError("non-finite value in the training step")


_NEXT_GX_DEST = None      # see set_next_input_grad_dest


def set_next_input_grad_dest(dest: Optional[torch.Tensor]) -> None:
    """The NEXT shadow product's backward pass writes its input gradient straight into `dest` (a [rows, K] bfloat16 tensor, rows may be
    strided) instead of a fresh tensor: the Transformer-Diffusion head hands the [CLS] rows of a persistent zero tensor, and the row
    scatter that followed the product disappears from the step."""
    global _NEXT_GX_DEST
    _NEXT_GX_DEST = dest if switches.get("gx_dest") == "1" else None


class _ShadowLinearFn(torch.autograd.Function):
    """y = x W^T + b with W, b taken from the optimiser's bfloat16 shadow; the weight / bias gradients are not returned to
    autograd but left (in bfloat16) in `stash`, from where ONE multi-tensor copy moves all of them into the flat float32
    gradient buffer.  Same arithmetic as nn.Linear under bf16 autocast (bf16 operands, fp32 accumulation, bf16 results),
    without its per-step kernels: weight cast, bias cast, gradient cast back and accumulate -- four per parameter."""

    @staticmethod
    def forward(ctx, x, w_sh, b_sh, stash, iw, ib, anchor, w_grad=None, b_grad=None):
        # `anchor` (the layer's float32 weight Parameter) is not read: it makes autograd record the node even when x
        # needs no gradient (first layer); its own gradient slot stays None -- the gradients travel through `stash`
        # (or, for products over thousands of rows, straight into `w_grad`, the weight's slice of the zeroed flat gradient buffer)
        xb = x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)
        x2 = xb.reshape(-1, xb.shape[-1])
        y = torch.addmm(b_sh, x2, w_sh.t()) if b_sh is not None else x2 @ w_sh.t()
        ctx.save_for_backward(x2, w_sh)
        ctx.stash, ctx.iw, ctx.ib, ctx.xshape, ctx.xdtype, ctx.w_grad, ctx.b_grad = stash, iw, ib, x.shape, x.dtype, w_grad, b_grad
        global _NEXT_GX_DEST
        ctx.gx_dest, _NEXT_GX_DEST = _NEXT_GX_DEST, None
        return y.reshape(*x.shape[:-1], w_sh.shape[0])

    @staticmethod
    def backward(ctx, gy):
        x2, w_sh = ctx.saved_tensors
        g2 = gy.reshape(-1, gy.shape[-1])
        if g2.dtype != torch.bfloat16:
            g2 = g2.to(torch.bfloat16)
        if ctx.w_grad is not None and x2.shape[0] >= _SPLIT_WGRAD_ROWS and _SPLIT_WGRAD_ROWS > 0:
            # many rows, small output: the library walks all rows inside a handful of tiles (25 us for 3584 x 360 x 120) and the bias
            # gradient is another reduction pass (11 us); the split-row kernel of csrc/seq_block.hip adds partial tiles -- and the
            # column sums of the slabs it reads anyway -- into the float32 gradients itself
            from . import _cabi
            g2 = g2.contiguous()
            rows_ok = switches.get("wgrad_rows") == "1" and _WGRAD_QUEUE is not None and x2.dim() == 2 and x2.stride(1) == 1 and x2.stride(0) >= x2.shape[1]
            x2c = x2 if rows_ok else x2.contiguous()        # the grouped launch takes row strides (the head reads the [CLS] rows in place)
            fused_bias = ctx.ib is not None and ctx.b_grad is not None
            if _WGRAD_QUEUE is not None:
                # deferred: the training step launches every queued product at once after backward (flush_wgrad_queue)
                _WGRAD_QUEUE.append((g2, x2c, ctx.w_grad, ctx.b_grad if fused_bias else None))
            else:
                with torch.cuda.device(g2.device):
                    rc = _cabi.load().ops_linear_wgrad_accumulate(x2c.shape[0], g2.shape[1], x2c.shape[1], g2.data_ptr(), x2c.data_ptr(),
                                                                  ctx.w_grad.data_ptr(), ctx.b_grad.data_ptr() if fused_bias else None,
                                                                  torch.cuda.current_stream(g2.device).cuda_stream)
                if rc != 0:
                    raise RuntimeError(f"ops_linear_wgrad_accumulate failed with code {rc}")
            if ctx.ib is not None and not fused_bias:
                ctx.stash[ctx.ib] = g2.sum(0)
        else:
            ctx.stash[ctx.iw] = g2.t() @ x2
            if ctx.ib is not None:
                ctx.stash[ctx.ib] = g2.sum(0)
        gx = None
        if ctx.needs_input_grad[0]:
            dest = ctx.gx_dest
            if dest is not None and dest.dtype == torch.bfloat16 and tuple(dest.shape) == (g2.shape[0], w_sh.shape[1]) and ctx.xdtype == torch.bfloat16:
                gx = torch.mm(g2, w_sh, out=dest)
            else:
                gx = (g2 @ w_sh).reshape(ctx.xshape)
                if gx.dtype != ctx.xdtype:
                    gx = gx.to(ctx.xdtype)
        return gx, None, None, None, None, None, None, None, None


class ShadowRec:
    """One shadow product's operands and gradient destinations (enable_shadow_linears)."""
    __slots__ = ("w_sh", "b_sh", "w_grad", "b_grad", "stash", "iw", "ib")

    def __init__(self, w_sh, b_sh, w_grad, b_grad, stash, iw, ib):
        self.w_sh, self.b_sh, self.w_grad, self.b_grad, self.stash, self.iw, self.ib = w_sh, b_sh, w_grad, b_grad, stash, iw, ib


def queue_column_sums(rows: torch.Tensor, dest: torch.Tensor) -> bool:
    """dest [N] (float32, a slice of the zeroed flat gradient buffer) += column sums of rows [T, N] (float32, unit column stride) as a job
    of the step's grouped weight-gradient launch.  False when no grouped launch is being collected (the caller adds them itself)."""
    if _WGRAD_QUEUE is None or rows.dtype != torch.float32 or rows.stride(1) != 1 or dest.dtype != torch.float32 or not dest.is_contiguous():
        return False
    _WGRAD_QUEUE.append((rows, None, dest, None))
    return True


def shadow_param_grads(rec: ShadowRec, g2: torch.Tensor, x2: torch.Tensor) -> None:
    """Weight / bias gradients of y = x W^T + b given dY = g2 [T, N] and X = x2 [T, K] (bfloat16): exactly what
    _ShadowLinearFn.backward does with them -- queued for the step's grouped split-row launch when the product has many rows, the
    library's product into the stash otherwise."""
    if rec.w_grad is not None and x2.shape[0] >= _SPLIT_WGRAD_ROWS and _SPLIT_WGRAD_ROWS > 0:
        from . import _cabi
        g2 = g2.contiguous()
        rows_ok = _WGRAD_QUEUE is not None and x2.dim() == 2 and x2.stride(1) == 1 and x2.stride(0) >= x2.shape[1]
        x2c = x2 if rows_ok else x2.contiguous()          # (row-strided operands: the grouped launch takes the stride)
        fused_bias = rec.ib is not None and rec.b_grad is not None
        if _WGRAD_QUEUE is not None:
            _WGRAD_QUEUE.append((g2, x2c, rec.w_grad, rec.b_grad if fused_bias else None))
        else:
            with torch.cuda.device(g2.device):
                rc = _cabi.load().ops_linear_wgrad_accumulate(x2c.shape[0], g2.shape[1], x2c.shape[1], g2.data_ptr(), x2c.data_ptr(),
                                                              rec.w_grad.data_ptr(), rec.b_grad.data_ptr() if fused_bias else None,
                                                              torch.cuda.current_stream(g2.device).cuda_stream)
            if rc != 0:
                raise RuntimeError(f"ops_linear_wgrad_accumulate failed with code {rc}")
        if rec.ib is not None and not fused_bias:
            rec.stash[rec.ib] = g2.sum(0)
    else:
        rec.stash[rec.iw] = g2.t() @ x2
        if rec.ib is not None:
            rec.stash[rec.ib] = g2.sum(0)


def shadow_grads_are_deferred(rec: ShadowRec, rows: int) -> bool:
    """True when `shadow_param_grads(rec, g2, x2)` over `rows` rows only QUEUES its work for the step's grouped launch -- it then reads
    neither operand before `flush_wgrad_queue`.  A caller whose kernel has not yet written g2 (tfd_fused: the later layer of a
    pair launch) may only wait in that case: the library products and column sums of the other branches run at once."""
    return (_WGRAD_QUEUE is not None and rec.w_grad is not None and _SPLIT_WGRAD_ROWS > 0 and rows >= _SPLIT_WGRAD_ROWS
            and (rec.ib is None or rec.b_grad is not None))


# products over at least this many rows take the split-row kernel (0: never).  r04: 16 (was 512) -- the kernel now keeps a whole tile per wave and
# reduces inside the workgroup, so a 10-sample tail batch (70 / 10 rows: twelve library GEMMs of ~16 us + twelve column sums per epoch)
# is one grouped launch too; only products over fewer rows than one MFMA tile stay with the library
_SPLIT_WGRAD_ROWS = int(switches.get("split_wgrad_rows"))
_WGRAD_QUEUE = None          # a list while a training step collects its split-row weight gradients for ONE grouped launch


def flush_wgrad_queue(device) -> None:
    """Launch everything `_ShadowLinearFn.backward` queued (operands stay alive in the queue until here), 16 products per launch."""
    from . import _cabi
    tf = sys.modules.get(__package__ + ".tfd_fused")
    if tf is not None:
        tf.flush_pending_backward()        # a layer's backward launch that waited for a predecessor which never ran
    q = _WGRAD_QUEUE
    if not q:
        return
    lib = _cabi.load()
    with torch.cuda.device(device):
        s = torch.cuda.current_stream(device).cuda_stream
        for i0 in range(0, len(q), _cabi.WGRAD_MAX_GROUP):
            part = q[i0:i0 + _cabi.WGRAD_MAX_GROUP]
            arr = (_cabi.WgradProblem * len(part))()
            for e, (g2, x2, wg, bg) in zip(arr, part):
                if x2 is None:                                           # column-sum job: wg [N] += column sums of the float32 rows g2 [T, N]
                    e.T, e.N, e.K = g2.shape[0], g2.shape[1], 0
                    e.dY, e.X, e.dW, e.dbias, e.ldy, e.ldx = g2.data_ptr(), None, wg.data_ptr(), None, g2.stride(0), 0
                    continue
                e.T, e.N, e.K = x2.shape[0], g2.shape[1], x2.shape[1]
                e.dY, e.X, e.dW, e.dbias = g2.data_ptr(), x2.data_ptr(), wg.data_ptr(), (bg.data_ptr() if bg is not None else None)
                e.ldy, e.ldx = g2.stride(0), x2.stride(0)                # (row views with unit column stride travel without a copy)
            rc = lib.ops_linear_wgrad_accumulate_group(len(part), arr, s)
            if rc != 0:
                raise RuntimeError(f"ops_linear_wgrad_accumulate_group failed with code {rc}")
    q.clear()


def enable_shadow_linears(model: nn.Module, opt: "FlatClipAdam", params, flat: torch.Tensor):
    """Routes every plain nn.Linear of `model` through _ShadowLinearFn.  Returns (stash, grad_views, patched modules)."""
    import types
    sh = opt.enable_shadow()
    offs, off = {}, 0
    for q in params:
        offs[id(q)] = off
        off += q.numel()
    stash, dst, patched = [], [], []

    def register(weight, bias):
        """y = x W^T + b through the shadow of (weight, bias); their gradients travel through `stash` (or, for products over
        thousands of rows, are added straight into their slices of the zeroed flat gradient buffer)."""
        ow = offs[id(weight)]
        w_sh = sh[ow:ow + weight.numel()].view_as(weight)
        w_grad = flat[ow:ow + weight.numel()].view_as(weight)
        iw = len(stash); stash.append(None); dst.append(w_grad)
        b_sh, ib, b_grad = None, None, None
        if bias is not None:
            ob = offs[id(bias)]
            b_sh = sh[ob:ob + bias.numel()]
            b_grad = flat[ob:ob + bias.numel()]
            ib = len(stash); stash.append(None); dst.append(b_grad)
        def prod(x):
            return _ShadowLinearFn.apply(x, w_sh, b_sh, stash, iw, ib, weight, w_grad, b_grad)

        # what a fused block that runs this product itself needs: the shadow operands, and where its gradients go (shadow_param_grads)
        prod.rec = ShadowRec(w_sh, b_sh, w_grad, b_grad, stash, iw, ib)
        return prod

    for mod in model.modules():
        if isinstance(mod, nn.MultiheadAttention):
            # the attention's projections are bare Parameters / a Linear subclass that its functional forward never calls: hand the
            # shadow products to the encoder fast path (tfd_fused.py), which is their only user
            if mod._qkv_same_embed_dim and mod.in_proj_bias is not None and id(mod.in_proj_weight) in offs and id(mod.out_proj.weight) in offs:
                mod._ops_in_proj = register(mod.in_proj_weight, mod.in_proj_bias)
                mod._ops_out_proj = register(mod.out_proj.weight, mod.out_proj.bias)
                patched.append(mod)
            continue
        if type(mod) is not nn.Linear or id(mod.weight) not in offs:
            continue
        prod = register(mod.weight, mod.bias)

        def fwd(self, x, prod=prod):
            if not x.is_cuda:
                return F_linear(x, self.weight, self.bias)
            return prod(x)

        mod.forward = types.MethodType(fwd, mod)
        mod._ops_prod = prod
        patched.append(mod)
    return stash, dst, patched


def disable_shadow_linears(patched) -> None:
    for mod in patched:
        if "forward" in mod.__dict__:
            del mod.__dict__["forward"]
        for name in ("_ops_in_proj", "_ops_out_proj", "_ops_prod"):
            if name in mod.__dict__:
                del mod.__dict__[name]


_GROUP_WGRAD = switches.get("group_wgrad") == "1"    # A/B switch: 0 = one launch per product, inside backward
# data-parallel step (world > 1).  Default: [graph A: noise, forward, loss, backward] -> gradient all-reduce -> [graph B: average, clip,
# Adam], the collective enqueued asynchronously (the host never blocks on it: `wait()` only orders the compute stream behind the
# collective's) -- OPS_AMD_DP_ASYNC=0 is the plain blocking-call form, bit for bit the same arithmetic (tests/test_surrogates.py).
# r06: under `nccl` the collective is captured too -- the whole step is ONE graph, one replay per step and no host work between its segments
# (one-rank RCCL timing, profiles/r05_notes.md 6: PINN 137.5 us against 169.1 us for two graphs around an eager collective, TFD 191.3 / 219.8;
# the same bits as the plain run).  A failure while capturing falls back to the two-graph form; a captured collective that goes wrong on
# N > 1 ranks would hang at REPLAY, past every fallback, so the first replay runs under a timer that ends the rank with exit code 17
# (OPS_AMD_DP_STALL_S seconds, default 120: the launcher then takes the job down; a fresh job may set OPS_AMD_DP_ONE_GRAPH=0 -- never a re-exec
# from a process that holds the GPU).  OPS_AMD_DP_ONE_GRAPH=0: [graph A | all-reduce | graph B].
_DP_ASYNC = os.environ.get("OPS_AMD_DP_ASYNC", "1") == "1"
_DP_ONE_GRAPH = os.environ.get("OPS_AMD_DP_ONE_GRAPH", "1") == "1"
_DP_STALL_S = float(os.environ.get("OPS_AMD_DP_STALL_S", "120"))


stall_hook = None     # callable(what) run by a firing _StallGuard just before the process ends


class _StallGuard:
    """`with _StallGuard(seconds, what):` -- if the block has not finished after `seconds`, the process ends with exit code 17 and a line on
    stderr saying what stalled (a hung graph replay of a captured collective cannot be recovered inside the process)."""

    def __init__(self, seconds: float, what: str):
        self.seconds, self.what, self.timer = seconds, what, None

    def _fire(self):
        if stall_hook is not None:       # the caller's last words (bench.py: the FE record measured before the training part)
            try:
                stall_hook(self.what)
            except Exception:
                pass
        sys.stderr.write(f"openpystruct_amd: {self.what} did not finish within {self.seconds:.0f} s -- ending this rank (exit 17); "
                         f"run again with OPS_AMD_DP_ONE_GRAPH=0 for the two-graph step\n")
        sys.stderr.flush()
        os._exit(17)

    def __enter__(self):
        import threading
        self.timer = threading.Timer(self.seconds, self._fire)
        self.timer.daemon = True
        self.timer.start()
        return self

    def __exit__(self, *exc):
        self.timer.cancel()
        return False
_FORCE_DP = os.environ.get("OPS_AMD_FORCE_DP", "0") == "1"           # run the data-parallel branch with a one-rank process group too
_DP_PROFILE = os.environ.get("OPS_AMD_DP_PROFILE", "0") == "1"       # HIP events around the step's segments, reported as out["dp_segments"]
F_linear = torch.nn.functional.linear
_SHADOW_LINEAR = switches.get("shadow_linear") == "1"   # A/B switch: 0 = nn.Linear under autocast


def build_model_and_loss(kind: str, cfg, data: SurrogateData, device):
    if kind == "pinn":
        out_dim = cfg.nelem + 2 * (cfg.nelem + 1)
        model = FNNWithResidual(data.X_train.shape[1], cfg.hidden_units, cfg.num_blocks, out_dim, cfg.dropout_rate)   # PINN:687-693
        crit = CompositeLoss(cfg.nelem, cfg.nelem + 1, cfg.nelem + 1, cfg.initial_alpha, cfg.box_constraint_coeff,
                             data.min_constraint, data.max_constraint, cfg.penalty_pinn)                              # PINN:698-699
    elif kind == "tfd":
        model = ModelOnePassTransformerWithDiffusion(cfg.n_cases, data.feat_dim, cfg.nelem, cfg.hidden_units,
                                                     cfg.num_transformer_layers, cfg.num_heads, cfg.dim_feedforward,
                                                     cfg.dropout_rate, cfg.max_len, cfg.diffusion_hidden_dim, cfg.diffusion_T)  # TFD:664-676
        crit = TrainableL1L2Loss(cfg.initial_alpha, data.min_constraint, data.max_constraint, cfg.box_constraint_coeff)       # TFD:680
    elif kind == "gnn":
        model = ChainGNN(data.X_train.shape[1], cfg.nelem, cfg.encoder_hidden_dim, cfg.gnn_hidden_dim, cfg.num_gnn_layers, cfg.dropout_rate)
        crit = TrainableL1L2Loss(cfg.initial_alpha, data.min_constraint, data.max_constraint, cfg.box_constraint_coeff)
    elif kind == "fno":
        model = FNO1dModel(cfg.n_cases, data.feat_dim, cfg.nelem, cfg.fno_modes, cfg.fno_width, cfg.num_fno_layers, cfg.hidden_units,
                           cfg.dropout_rate)
        crit = TrainableL1L2Loss(cfg.initial_alpha, data.min_constraint, data.max_constraint, cfg.box_constraint_coeff)
    elif kind == "fnn":
        model = FNNPlain(data.X_train.shape[1], cfg.hidden_units, cfg.num_residual_blocks, cfg.nelem, cfg.dropout_rate)
        crit = TrainableL1L2Loss(cfg.initial_alpha, data.min_constraint, data.max_constraint, cfg.box_constraint_coeff)
    else:
        raise ValueError(kind)
    return model.to(device), crit.to(device)


@dataclass
class PhysicsTerm:
    """Optional FE-residual term (physics.py; BASELINE configs 3-4).  Needs per-case targets (cfg.n_cases == 1, so that
    inertias and displacements belong to ONE load case) and `data.Fy_train`.
    PINN: r = K(I_pred) u_pred - f on the model's own (I, v, theta) outputs.
    TFD / FNN (I-only outputs): r = K(I_pred) u_rec - f with the displacement field recorded in the dataset
    (`data.v_train`, `data.theta_train`): zero when the predicted inertias reproduce the recorded response.  The
    records must keep the last node's values (SizingConfig.zero_last_node = False: MultiCore.py:222-223 zeroes them)."""
    weight: float
    x: torch.Tensor        # [N] node coordinates
    E: float
    fix: torch.Tensor      # [N] uint8
    wy: float


def r2_score(y_true: torch.Tensor, y_pred: torch.Tensor) -> float:
    """sklearn.metrics.r2_score on the raveled arrays (PINN:851)."""
    yt, yp = y_true.double().reshape(-1), y_pred.double().reshape(-1)
    ss_res = ((yt - yp) ** 2).sum()
    ss_tot = ((yt - yt.mean()) ** 2).sum()
    return float(1.0 - ss_res / ss_tot)


def _allreduce_mean(t: torch.Tensor, world: int, dp: Optional[bool] = None) -> torch.Tensor:
    if (world > 1) if dp is None else dp:
        dist.all_reduce(t)
        t /= world
    return t


def train_surrogate(kind: str, data: SurrogateData, cfg=None, device="cuda", *, autocast_dtype=torch.bfloat16, **kw) -> Dict[str, object]:
    """`_train_surrogate` (below: every argument is documented there).  With `autocast_dtype=torch.float16` -- the reference's own AMP mode,
    fp16 autocast + GradScaler (PINN:706, :759-768; TFD:690, :744-753) -- the run takes the framework's modules throughout (the hand-written
    launches are bf16): the module-level switches of the fused pieces are off for its duration."""
    if autocast_dtype != torch.float16:
        return _train_surrogate(kind, data, cfg, device, autocast_dtype=autocast_dtype, **kw)
    from . import surrogates as _s
    g = globals()
    saved = (_s._FUSED_TAILS, _s._FUSED_STENCIL, g["_FUSED_LOSS"], switches.get("fused_prep"), switches.get("fused_physics"))
    try:
        _s._FUSED_TAILS = _s._FUSED_STENCIL = False
        g["_FUSED_LOSS"] = False
        switches.set("fused_prep", 0); switches.set("fused_physics", 0)
        return _train_surrogate(kind, data, cfg, device, autocast_dtype=autocast_dtype, **kw)
    finally:
        _s._FUSED_TAILS, _s._FUSED_STENCIL, g["_FUSED_LOSS"] = saved[:3]
        switches.set("fused_prep", saved[3]); switches.set("fused_physics", saved[4])


def _train_surrogate(kind: str, data: SurrogateData, cfg=None, device="cuda", *, autocast_dtype=torch.bfloat16,
                     sync_bn: bool = False, max_epochs: Optional[int] = None, log=None, seed: int = 0,
                     use_graph: Optional[bool] = None, physics: Optional[PhysicsTerm] = None,
                     init_fn=None, batch_order=None) -> Dict[str, object]:
    """Trains on THIS rank's `data` shard; uses DDP when torch.distributed is initialised with world_size > 1.
    Returns history, best state dict, validation R^2 (I only) and per-epoch times.
    `init_fn(model)`: called once on the freshly built model (load a checkpoint, deterministic test weights).
    `batch_order(epoch) -> LongTensor[n_train]`: the epoch's sample order instead of a random permutation (the
    reference's DataLoader shuffle, PINN:701, is unseeded; the golden-fixture tests replay the order it drew).
    `autocast_dtype=torch.float16`: the reference's AMP mode (fp16 autocast + GradScaler, PINN:706, :759-768) on the framework modules; the
    returned dict then carries `grad_scaler` = {"scale", "skipped_steps", "steps"}."""
    cfg = cfg or {"pinn": PinnConfig, "tfd": TfdConfig, "fnn": FnnConfig, "gnn": GnnConfig, "fno": FnoConfig}[kind]()
    alpha_term = kind in ("tfd", "fnn", "fno")          # (initial_alpha - alpha)^2 in the training loss (TFD:743, FNO:615)
    if kind == "fno":
        autocast_dtype = None                           # FNO:613 runs with autocast disabled
    device = torch.device(device)
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    # the data-parallel branch: more than one rank -- or, with OPS_AMD_FORCE_DP=1, a process group of ONE rank (r05: the RCCL path -- flat
    # all-reduce between graph A and graph B, its async form, the one-graph capture of the collective, dp_segments -- executed on the
    # single GPU a test box has; a one-rank all-reduce is the identity and 1 / world = 1, so losses equal the plain run bit for bit)
    dp = world > 1 or (_FORCE_DP and dist.is_available() and dist.is_initialized())
    torch.manual_seed(seed)            # identical initial weights on every rank
    model, crit = build_model_and_loss(kind, cfg, data, device)
    if kind == "tfd" and device.type == "cuda":
        from . import tfd_fused as _tf
        _tf.disarm_gather()              # (whatever an earlier run that did not reach its end left armed)
    if init_fn is not None:
        init_fn(model)
    if sync_bn and dp:
        model = nn.SyncBatchNorm.convert_sync_batchnorm(model)
    net = model
    # one flat gradient buffer; every .grad is a view into it (autograd accumulates in place)
    params = [q for q in model.parameters()]
    flat = torch.zeros(sum(q.numel() for q in params), device=device, dtype=torch.float32)
    off = 0
    for q in params:
        q.grad = flat[off:off + q.numel()].view_as(q)
        off += q.numel()
    g_stash, g_dst, patched = [], [], []
    fast_encoder = None
    shared_counter = False
    if hasattr(model, "direct_param_grads"):   # fused tails (csrc/fused_bn.hip) write BatchNorm parameter gradients into `flat` themselves
        model.direct_param_grads = device.type == "cuda"
    torch.manual_seed(seed + 1 + rank)  # different noise / shuffles per rank
    on_gpu = device.type == "cuda"
    # the reference's AMP mode: GradScaler(init_scale 65536, growth 2, backoff 0.5, growth interval 2000) -- torch.cuda.amp.GradScaler's defaults
    fp16_scaler = {"scale": 65536.0, "good": 0, "skipped": 0, "steps": 0} if (on_gpu and autocast_dtype == torch.float16) else None
    if fp16_scaler is not None:
        use_graph = False                # the scale and the skip decision are per-step host state, like `scaler.step` / `scaler.update`
    if use_graph is None:
        use_graph = on_gpu and os.environ.get("OPS_AMD_GRAPH", "1") == "1"    # the step is launch-bound (~150 tiny kernels): replay it as HIP graphs
    # under graph replay the learning rate must live in a device tensor, or the scheduler's updates would
    # never reach the captured optimiser step
    prep_counter = torch.zeros(2, dtype=torch.int64, device=device) if on_gpu else None     # batch assembly: [calls, workgroups done]
    if on_gpu:      # clip + Adam over the flat buffers in two HIP launches; loss.alpha NOT included (PINN:696)
        opt = FlatClipAdam(params, flat, cfg.learning_rate, weight_decay=cfg.weight_decay, max_norm=1.0, decoupled=kind == "gnn")
        sched = None
        if _SHADOW_LINEAR and autocast_dtype == torch.bfloat16:
            g_stash, g_dst, patched = enable_shadow_linears(model, opt, params, flat)
            if kind == "tfd":
                from . import tfd_fused          # csrc/seq_block.hip: attention, dropout + add + LayerNorm, ReLU + dropout, diffusion front end
                if tfd_fused.patch_model(model, seed=seed * 7919 + 211 + rank, direct_param_grads=True):
                    fast_encoder = model
                    opt.repack = getattr(model.transformer_encoder, "_ops_tile_entries", None)    # the Adam launch refreshes the layer kernels' weight tiles
                    if switches.get("fused_prep") == "1":
                        # the batch-assembly launch advances its call counter once per step: the dropout / noise streams read that one
                        tfd_fused.share_step_counter(model.transformer_encoder, prep_counter)
                        shared_counter = True
    else:
        opt = (torch.optim.AdamW if kind == "gnn" else torch.optim.Adam)(model.parameters(), lr=cfg.learning_rate,   # GNN:394
                                                                         weight_decay=cfg.weight_decay)
        sched = ExponentialLR(opt, gamma=cfg.gamma)
    # PINN: forward + loss + backward as 13 hand-written launches without autograd (pinn_fused.py / csrc/mlp_block.hip);
    # the module keeps owning parameters and buffers, evaluation keeps running it
    engine = None
    if on_gpu and kind == "pinn" and autocast_dtype == torch.bfloat16 and physics is None and not (sync_bn and dp):
        from . import pinn_fused
        if pinn_fused.eligible(model, crit, cfg.batch_size):
            engine = pinn_fused.PinnFusedStep(model, crit, seed=seed * 7919 + 101 + rank)
            opt.repack = engine._repack      # the Adam launch refreshes the engine's bf16 weight copies
            if not dp and switches.get("pinn_norm_fold") == "1":
                # one rank: the gradients are final when the weight-gradient launch ends, so that launch leaves the clip norm's partial
                # sums and the optimiser skips its norm launch (a node and ~4.5 us per step; data parallel: the norm is the all-reduced one)
                opt.norm_ready_parts = engine.enable_norm(flat, opt.ws, opt.step_count, opt.betas)
            if not dp and switches.get("pinn_repack_in_gather") == "1":
                # one rank: the weight copies of step n ride on the batch-assembly launch of step n + 1 (adjacent launches that depend on
                # nothing of each other: one node and ~5 us less per step); the optimiser call no longer rebuilds them
                opt.repack = None
                engine.repack_in_gather(opt.p)
    if use_graph and on_gpu and engine is None and fast_encoder is None:
        # a step the FRAMEWORK differentiates (comparator paths, the sibling surrogates): its multi-block reductions are only right under
        # graph replay when captured memset nodes are (runtime.py item 2); the hand-written paths above contain none
        from . import runtime
        if not runtime.graph_memsets_replay_correctly(device):
            use_graph = False
            if log:
                log(f"captured hipMemsetAsync nodes replay wrong values in this process ({runtime.PACKET_CAPTURE_ENV}=0 came after its "
                    "first HIP call?): the framework-differentiated step runs eagerly")
    Xtr, Ytr, Xva, Yva = (t.to(device) for t in (data.X_train, data.Y_train, data.X_val, data.Y_val))
    nb_tr = max(1, (Xtr.shape[0] + cfg.batch_size - 1) // cfg.batch_size)
    if dp:   # every rank must run the same number of steps (collectives inside backward)
        # a one-row tail batch makes train-mode BatchNorm raise (as in the reference's single process); on one rank of a
        # data-parallel job that exception would leave the other ranks waiting in the step's all-reduce: drop such a tail
        if Xtr.shape[0] % cfg.batch_size == 1 and nb_tr > 1 and any(isinstance(m, nn.modules.batchnorm._BatchNorm) for m in model.modules()):
            nb_tr -= 1
        t = torch.tensor([nb_tr], device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        nb_tr = int(t.item())
    use_ac = device.type == "cuda" and autocast_dtype is not None
    hist = {"train": [], "val": [], "epoch_s": []}
    # Transformer-Diffusion fast path, r04: the front-end launch assembles its own batch (rows order[cursor ..] of the training set + the
    # input noise) and the head's loss reads its targets through those rows: no batch-assembly launch per step.  The epoch's permutation
    # lives in `g_order`, `g_cursor` is advanced by the launch itself (reset per epoch).
    fuse_gather = False
    g_order = g_cursor = g_noise = None
    _FUSED_PHYS = on_gpu and switches.get("fused_physics") == "1"      # A/B switch: 0 = physics.fe_residual_loss from framework ops
    # the loss launch adds every step's value to the epoch's running sum itself (zeroed per epoch): no add node per step
    # (with the fused physics term both launches add into the same running sum: the epoch's total of data loss + term)
    loss_acc = torch.zeros((), dtype=torch.float32, device=device) if (on_gpu and _FUSED_LOSS and engine is None and (physics is None or _FUSED_PHYS) and switches.get("loss_acc") == "1") else None
    # the loss on the head's tile: its value exists once the head's backward launch has run, so a second term can only be tied to it
    # (not added) -- which needs the running sum to carry the step values
    head_loss_ok = fast_encoder is not None and _FUSED_LOSS and (physics is None or (_FUSED_PHYS and loss_acc is not None))
    if (head_loss_ok and shared_counter and Xtr.dtype == torch.float32 and Xtr.dim() == 3
            and Xtr.is_contiguous() and Ytr.dtype == torch.float32 and Ytr.is_contiguous() and Ytr.dim() == 2):
        from . import tfd_fused
        if tfd_fused.gather_fusable(model, device, int(Xtr.shape[2])):
            fuse_gather = True
            g_order = torch.arange(max(int(Xtr.shape[0]), cfg.batch_size), device=device) % int(Xtr.shape[0])
            g_cursor = torch.zeros((), dtype=torch.int64, device=device)
            g_noise = torch.zeros((), device=device)
            g_idx = torch.zeros(max(cfg.batch_size, 1), dtype=torch.int64, device=device)
            tfd_fused.arm_gather(model, Xtr, Ytr, g_order, g_cursor, g_idx, g_noise, (seed * 7919 + 13 + rank) & 0x7FFFFFFFFFFFFFFF)

    if physics is not None:
        if cfg.n_cases != 1 or data.Fy_train is None:
            raise ValueError("the FE-residual term needs n_cases == 1 and data.Fy_train (per-case loads)")
        if kind != "pinn" and data.v_train is None:
            raise ValueError("the FE-residual term of an I-only model needs the recorded displacements (data.v_train)")
        from .physics import fe_residual_loss
        Fy_tr = data.Fy_train.to(device)
        sI = data.scalers_Y["I"]
        if kind == "pinn":
            sD, sR = data.scalers_Y["deflections"], data.scalers_Y["rotations"]
        else:
            v_rec = data.v_train.to(device=device, dtype=torch.float64).contiguous()        # (the fused term reads raw float64 rows)
            t_rec = data.theta_train.to(device=device, dtype=torch.float64).contiguous()
        px, pfix = physics.x.to(device=device, dtype=torch.float64), physics.fix.to(device=device, dtype=torch.uint8)
        pE = torch.tensor(float(physics.E), dtype=torch.float64, device=device)      # device scalars: nothing crosses PCIe
        pwy = torch.tensor(float(physics.wy), dtype=torch.float64, device=device)    # inside a captured step

    def physics_inputs(rows, out=None):
        """Per-batch loads (and, for the I-only models, recorded displacement fields); `out`: the graph's static buffers.
        Fused term (csrc/beam_residual.hip ops_physics_loss_*): only the ROW INDICES travel -- the launches gather by them."""
        if _FUSED_PHYS:
            if fuse_gather:              # the front-end launch leaves the step's rows in g_idx
                return (g_idx[:rows.numel()],)
            if out is None:
                return (rows,)
            out[0].copy_(rows)
            return out
        srcs = (Fy_tr,) if kind == "pinn" else (Fy_tr, v_rec, t_rec)
        if out is None:
            return tuple(t[rows] for t in srcs)
        for t, o in zip(srcs, out):
            torch.index_select(t, 0, rows, out=o)
        return out

    def physics_loss(preds, pin):
        nel = cfg.nelem
        if _FUSED_PHYS:          # three launches; the value is already weighted and enters the total with weight one
            from .physics import fused_residual_term
            disp = (sD, sR) if kind == "pinn" else (v_rec, t_rec)
            return fused_residual_term(preds, nel, sI, disp, pin[0], Fy_tr, px, float(physics.E), pfix, float(physics.wy), physics.weight, loss_acc)
        preds = preds.float()
        I_p = sI.inverse_transform(preds[:, :nel]).clamp_min(1e-8)
        if kind == "pinn":
            v_p = sD.inverse_transform(preds[:, nel:2 * nel + 1])
            t_p = sR.inverse_transform(preds[:, 2 * nel + 1:])
        else:
            v_p, t_p = pin[1], pin[2]
        return physics.weight * fe_residual_loss(I_p, v_p, t_p, px, pE, pfix, pin[0], pwy).float()

    if on_gpu and engine is None and switches.get("adam_zero") == "1":
        opt.zero_grads = True            # `flat` starts zeroed (allocation) and every update leaves it zeroed

    def fwd_bwd(Xb, Yb, noise_t, pin=None, prenoised=False):
        """Segment A: local gradients of the mean batch loss into `flat`.  `prenoised`: Xb already is the gathered, noisy
        (and, under autocast, bfloat16) batch written by `gather_noise` -- one launch outside the graph instead of six nodes."""
        if engine is not None:           # batch and targets sit in the engine's buffers (engine.gather); every gradient is assigned
            return engine.fwd_bwd(int(Yb.shape[0]))
        Xn = Xb if prenoised else Xb + torch.randn_like(Xb) * noise_t   # PINN:756
        if not (on_gpu and opt.zero_grads):
            flat.zero_()                                                 # optimizer.zero_grad() (GPU: the update launch zeroes `flat` behind itself)
        with torch.autocast(device_type=device.type, dtype=autocast_dtype, enabled=use_ac):
            head_loss = None
            if head_loss_ok:
                # Transformer-Diffusion fast path: the loss on the head's output tile, finished by the head's backward launch
                from . import tfd_fused
                tfd_fused.arm_head_loss(Yb, crit, cfg.initial_alpha if alpha_term else None, loss_acc)
                preds = net(Xn)
                head_loss = tfd_fused.take_head_loss()
                if head_loss is None and fuse_gather:
                    # (with the batch assembled by the front-end launch the static target buffer `Yb` is never filled: the framework-side
                    #  loss below would be evaluated against stale targets -- fail loudly instead)
                    raise RuntimeError("the head's loss launch did not run although the front-end launch assembles the batch itself "
                                       "(criterion / shape mismatch in tfd_fused.arm_head_loss)")
            else:
                preds = net(Xn)
            if head_loss is not None:
                loss = head_loss
            elif on_gpu and _FUSED_LOSS:       # value + d/d preds in one HIP pass instead of ~80 framework kernel nodes
                loss = fused_loss(crit, preds, Yb, alpha0=cfg.initial_alpha if alpha_term else None, unit_grad=True, acc=loss_acc)
            else:
                loss = crit(preds.float(), Yb)
                if alpha_term:
                    loss = loss + (cfg.initial_alpha - crit.alpha) ** 2   # TFD:743 / FNN (constant 0: alpha never trains)
        if physics is not None:
            term = physics_loss(preds, pin)
            # (with the running sum both launches have already added their values up: the step's scalar only roots the backward pass,
            #  so the two terms are tied without an addition node -- its value is then the data loss alone)
            loss = _TieTerms.apply(loss, term) if (loss_acc is not None and _FUSED_PHYS) else loss + term
        global _WGRAD_QUEUE
        _WGRAD_QUEUE = [] if (g_stash and _GROUP_WGRAD) else None       # split-row weight gradients: one grouped launch after backward
        try:
            # the root gradient is a persistent ones tensor: no ones_like() fill node per step.  (r03 kept the implicit form because the
            # explicit one "produced" NaNs in bias gradients of framework-path runs; the NaNs were the HIP runtime's captured-memset
            # defect under the framework's own bias-gradient reductions -- runtime.py item 2 -- and the root gradient only moved the
            # memory layout that decided which garbage their semaphores saw)
            if fp16_scaler is not None:          # scaler.scale(loss).backward() (PINN:761): the root gradient IS the scale
                loss.backward(gradient=torch.full((), fp16_scaler["scale"], dtype=loss.dtype, device=device))
            elif _EXPLICIT_ROOT:
                loss.backward(gradient=_root_ones(device))
            else:
                loss.backward()
            flush_wgrad_queue(device)
        finally:
            _WGRAD_QUEUE = None
        if g_stash:                      # the shadow-linear weight / bias gradients: one multi-tensor cast-and-copy into `flat`
            live = [(dd, ss) for dd, ss in zip(g_dst, g_stash) if ss is not None]      # (a product nobody called this step leaves None)
            if live:
                torch._foreach_copy_([dd for dd, _ in live], [ss for _, ss in live])
            for k in range(len(g_stash)):
                g_stash[k] = None
        return loss.detach()

    def apply_update():
        """Segment B: average over ranks, clip, Adam."""
        if fp16_scaler is not None:
            # scaler.unscale_ + clip + scaler.step + scaler.update (PINN:763-768): a non-finite gradient skips the step and halves the scale,
            # 2 000 clean steps in a row double it; the unscaling rides in the update launch's gradient scale
            sc = fp16_scaler
            sc["steps"] += 1
            if bool(torch.isfinite(flat).all()):
                opt.step(grad_scale=1.0 / (world * sc["scale"]))
                sc["good"] += 1
                if sc["good"] >= 2000:
                    sc["scale"] *= 2.0; sc["good"] = 0
            else:
                flat.zero_()
                sc["scale"] *= 0.5; sc["good"] = 0; sc["skipped"] += 1
            return
        if on_gpu:
            opt.step(grad_scale=1.0 / world)                             # average, clip (PINN:766), Adam
            return
        if dp:
            flat.div_(world)
        torch.nn.utils.clip_grad_norm_(params, 1.0)                      # PINN:766
        opt.step()

    def allreduce_grads():
        """ONE flat all-reduce of every parameter gradient (2.38 MB PINN / 1.44 MB TFD)."""
        if _DP_ASYNC:
            dist.all_reduce(flat, async_op=True).wait()                  # GPU: stream ordering only, the host does not block
        else:
            dist.all_reduce(flat)

    def train_step(Xb, Yb, noise_t, rows=None):
        if engine is not None:
            engine.gather(Xtr, Ytr, rows, noise_t, engine_seed)
        loss = fwd_bwd(Xb, Yb, noise_t, physics_inputs(rows) if physics is not None else None)
        if dp:
            allreduce_grads()                                            # the step's only collective (RCCL over xGMI)
        apply_update()
        return loss

    def val_batch(Xb, Yb):
        with torch.no_grad(), torch.autocast(device_type=device.type, dtype=autocast_dtype, enabled=use_ac):
            preds = model(Xb)
            if on_gpu and _FUSED_LOSS:
                return fused_loss(crit, preds, Yb)
            return crit(preds.float(), Yb)

    # Transformer-Diffusion fast path: the validation pass as ONE forward over all validation rows (chunks of <= _VAL_CHUNK rows, whole
    # reference batches each) + the loss per reference batch on row slices of its output, every value added to `acc` by the loss launch
    # itself.  The reference's value -- the mean over its batches of the batch losses (TFD:760-775) -- is kept; a step-sized forward
    # fills 1/18 of the chip, so 20 of them in a row cost ~8x what one 20-fold forward does.  (Rows are independent in evaluation mode;
    # the diffusion draws of the pass come from one call of the stream instead of one per batch.)
    val_whole = bool(on_gpu and fast_encoder is not None and _FUSED_LOSS and switches.get("val_whole") == "1")
    _VAL_CHUNK = cfg.batch_size * max(1, 4096 // max(cfg.batch_size, 1))

    def val_all(acc):
        nva = int(Xva.shape[0])
        with torch.no_grad(), torch.autocast(device_type=device.type, dtype=autocast_dtype, enabled=use_ac):
            for c0 in range(0, nva, _VAL_CHUNK):
                preds = model(Xva[c0:c0 + _VAL_CHUNK])
                for b0 in range(c0, min(c0 + _VAL_CHUNK, nva), cfg.batch_size):
                    b1 = min(b0 + cfg.batch_size, nva)
                    fused_loss(crit, preds[b0 - c0:b1 - c0], Yva[b0:b1], acc=acc)

    # batch assembly in one launch (csrc/input_prep.hip): gather + noise + cast straight into the graph's input buffer
    _FUSED_PREP = on_gpu and switches.get("fused_prep") == "1"
    # bf16 batches only where the first module is a (shadow) Linear, which casts its operand to bf16 anyway
    prep_bf16 = bool(on_gpu and use_ac and autocast_dtype == torch.bfloat16 and patched and kind in ("pinn", "fnn", "gnn"))

    engine_seed = (seed * 7919 + 13 + rank) & 0x7FFFFFFFFFFFFFFF

    y_in_prep = bool(on_gpu and Ytr.dtype == torch.float32 and Ytr.is_contiguous() and switches.get("prep_targets") == "1")      # the targets travel in the same launch

    def gather_noise(idx, out, out_y=None):
        """Returns whether the targets were gathered too (out_y given and float32)."""
        if engine is not None:           # batch AND targets straight into the engine's layouts
            engine.gather(Xtr, Ytr, idx, s_noise, engine_seed)
            return True
        lib = opt._lib
        Fdim = 1
        for d_ in Xtr.shape[1:]:
            Fdim *= int(d_)
        with_y = out_y is not None and y_in_prep and out_y.dtype == torch.float32 and out_y.is_contiguous()
        Cdim = int(Ytr[0].numel()) if with_y else 0
        with torch.cuda.device(device):
            rc = lib.ops_gather_rows_noise_targets_f32(int(idx.numel()), Fdim, Xtr.data_ptr(), idx.data_ptr(), s_noise.data_ptr(),
                                                       (seed * 7919 + 13 + rank) & 0x7FFFFFFFFFFFFFFF, prep_counter.data_ptr(), out.data_ptr(),
                                                       int(out.dtype == torch.bfloat16), Ytr.data_ptr() if with_y else None, Cdim,
                                                       out_y.data_ptr() if with_y else None, torch.cuda.current_stream(device).cuda_stream)
        if rc != 0:
            raise RuntimeError(f"ops_gather_rows_noise_targets_f32 failed with code {rc}")
        return with_y

    graph = graph_b = vgraph = graph_t = vgraph_t = slot_graph = vgraph_all = None
    ev_graphs, val_rows = {}, None
    # PINN: validation through the engine's evaluation pass (running statistics, no dropout) when the validation set has its layout
    engine_eval = bool(engine is not None and switches.get("pinn_engine_eval") == "1" and Xva.dtype == torch.float32
                       and Xva.is_contiguous() and Yva.dtype == torch.float32 and Yva.is_contiguous() and Xva.dim() == 2
                       and cfg.batch_size <= _cabi_rows())
    graph_mode_one = False
    bs = cfg.batch_size
    if use_graph and Xtr.shape[0] >= bs:
        # static buffers + a few eager warm-up steps on a side stream, then capture one full-batch step:
        # not dp -> one graph; dp -> [fwd_bwd] graph, eager all-reduce, [apply_update] graph
        sX, sY = torch.zeros_like(Xtr[:bs], dtype=torch.bfloat16 if (_FUSED_PREP and prep_bf16) else Xtr.dtype), torch.zeros_like(Ytr[:bs])
        s_noise = torch.zeros((), device=device)
        sP = None
        if physics is not None:                  # static per-batch physics inputs, gathered before every replay
            sP = physics_inputs(torch.arange(bs, device=device))
            if not fuse_gather:          # (fused batch assembly: the rows are the front-end launch's own index output)
                sP = tuple(torch.zeros_like(t[:bs]) for t in sP)
        snap = (copy.deepcopy(model.state_dict()), copy.deepcopy(opt.state_dict()))
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            sX.copy_(Xtr[:bs]); sY.copy_(Ytr[:bs])
            if sP is not None:
                physics_inputs(torch.arange(bs, device=device), out=sP)
            if engine is not None:
                engine.gather(Xtr, Ytr, torch.arange(bs, device=device), s_noise, engine_seed)
            # (fused batch assembly: every launch advances g_cursor by its batch; warm-up and capture passes each start at row 0 so that no
            #  pass reads g_order past its end -- the kernel wraps as well, ops_tfd_front_args.n_order; the epoch loop zeroes it again)
            for _ in range(3):
                if fuse_gather:
                    g_cursor.zero_()
                fwd_bwd(sX, sY, s_noise, sP, prenoised=_FUSED_PREP)
                apply_update()                   # warm-up only: no collective needed for capture-readiness
            if fuse_gather:
                g_cursor.zero_()
            side.synchronize()
            try:
                graph = torch.cuda.CUDAGraph()
                one_graph = False
                if dp and _DP_ONE_GRAPH and dist.get_backend() == "nccl":
                    try:                 # the collective inside the capture: one replay per step, no host work between the segments
                        g1 = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g1, stream=side, capture_error_mode="thread_local"):
                            s_loss = fwd_bwd(sX, sY, s_noise, sP, prenoised=_FUSED_PREP)
                            dist.all_reduce(flat)
                            apply_update()
                        graph, one_graph = g1, True
                        graph_mode_one = True
                    except Exception as e:
                        if log:
                            log(f"capturing the gradient all-reduce failed ({e!r}); two graphs around an eager collective")
                        torch.cuda.synchronize(device)
                        graph = torch.cuda.CUDAGraph()
                if not one_graph:
                    with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                        s_loss = fwd_bwd(sX, sY, s_noise, sP, prenoised=_FUSED_PREP)
                        if not dp:
                            apply_update()
                    if dp:
                        graph_b = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(graph_b, stream=side, capture_error_mode="thread_local"):
                            apply_update()
            except Exception as e:          # same arithmetic eagerly; the step is then launch-bound
                if log:
                    log(f"HIP graph capture failed ({e!r}); training eagerly")
                graph = graph_b = None
                torch.cuda.synchronize(device)
            # the epoch's LAST (partial) batch as a graph of its own: an eager step is ~25 launches' worth of host time (0.3-0.4 ms of a
            # 5 ms TFD epoch), and eager model passes between replays are what the flaky-NaN hunt of r03 kept running into
            nt = int(Xtr.shape[0]) % bs
            if (graph is not None and not dp and sP is None and nt >= 2 and nb_tr == Xtr.shape[0] // bs + 1
                    and switches.get("tail_graph") == "1"):
                try:
                    sXt, sYt = (None if engine is not None else torch.zeros_like(sX[:nt])), torch.zeros_like(sY[:nt])
                    if engine is not None:       # the engine's own buffers are the graph's inputs: Yb only carries the row count
                        engine.gather(Xtr, Ytr, torch.arange(nt, device=device), s_noise, engine_seed)
                    else:
                        sXt.copy_(sX[:nt]); sYt.copy_(sY[:nt])
                    for _ in range(2):
                        if fuse_gather:
                            g_cursor.zero_()
                        fwd_bwd(sXt, sYt, s_noise, None, prenoised=_FUSED_PREP)
                        apply_update()
                    if fuse_gather:
                        g_cursor.zero_()
                    side.synchronize()
                    graph_t = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph_t, stream=side, capture_error_mode="thread_local"):
                        s_loss_t = fwd_bwd(sXt, sYt, s_noise, None, prenoised=_FUSED_PREP)
                        apply_update()
                except Exception as e:
                    if log:
                        log(f"HIP graph capture of the tail batch failed ({e!r}); that step runs eagerly")
                    graph_t = None
                    torch.cuda.synchronize(device)
        torch.cuda.current_stream(device).wait_stream(side)
        model.load_state_dict(snap[0]); opt.load_state_dict(snap[1])     # the warm-up steps never happened
        if on_gpu:
            opt.refresh_shadow()
            if engine is not None:
                engine.repack_now()
            if fast_encoder is not None:
                from . import tfd_fused
                tfd_fused.refresh_layer_tiles(fast_encoder.transformer_encoder)
        # the validation pass as graphs too (eval mode, loss accumulated into v_acc): one for the full batches, one for the last partial one
        nvt = int(Xva.shape[0]) % bs
        if engine_eval and graph is not None and switches.get("pinn_eval_slots") == "1":
            # PINN: the whole validation set per launch sequence -- every batch in an evaluation slot (pinn_fused.make_eval_slots:
            # gathered once, the rows never change), the engine's 7 evaluation launches run all slots side by side (grid.y): 7 nodes
            # per epoch instead of 14 batches x 8 launches of ~6 us
            try:
                side.wait_stream(torch.cuda.current_stream(device))
                with torch.cuda.stream(side):
                    engine.make_eval_slots(Xva, Yva, cfg.batch_size)
                    engine.evaluate_slots()
                    side.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                        engine.evaluate_slots()
                torch.cuda.current_stream(device).wait_stream(side)
                slot_graph = g
            except Exception as e:
                if log:
                    log(f"HIP graph capture of the slotted evaluation pass failed ({e!r}); one batch after the other")
                slot_graph = None
                torch.cuda.synchronize(device)
        if engine_eval and graph is not None and slot_graph is None:
            # PINN: the engine's own evaluation pass (forward stages on the running statistics + loss, 8 launches per batch) instead of
            # the module's ~30-node forward; gather + evaluate captured per batch size, the row indices in a static buffer
            val_rows = torch.arange(Xva.shape[0], device=device)
            for n in sorted({bs if Xva.shape[0] >= bs else 0, nvt} - {0}):
                try:
                    vi = val_rows[:n].clone()
                    side.wait_stream(torch.cuda.current_stream(device))
                    with torch.cuda.stream(side):
                        engine.gather(Xva, Yva, vi, None, 0); engine.evaluate(n)
                        side.synchronize()
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                            engine.gather(Xva, Yva, vi, None, 0); engine.evaluate(n)
                    torch.cuda.current_stream(device).wait_stream(side)
                    ev_graphs[n] = (g, vi)
                except Exception as e:
                    if log:
                        log(f"HIP graph capture of the engine's evaluation pass failed ({e!r}); those batches run eagerly")
                    torch.cuda.synchronize(device)
        elif graph is not None and (Xva.shape[0] >= bs or nvt >= 1):
            v_acc = torch.zeros((), device=device)
            net.eval()

            def capture_val(rows):
                bx, by = torch.zeros_like(Xva[:rows]), torch.zeros_like(Yva[:rows])
                side.wait_stream(torch.cuda.current_stream(device))
                with torch.cuda.stream(side), torch.no_grad():
                    bx.copy_(Xva[:rows]); by.copy_(Yva[:rows])
                    for _ in range(2):
                        val_batch(bx, by)
                    side.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                        v_acc.add_(val_batch(bx, by))
                torch.cuda.current_stream(device).wait_stream(side)
                return g, bx, by

            def capture_val_all():
                side.wait_stream(torch.cuda.current_stream(device))
                with torch.cuda.stream(side):
                    for _ in range(2):
                        val_all(v_acc)
                    side.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                        val_all(v_acc)
                torch.cuda.current_stream(device).wait_stream(side)
                return g

            try:
                if val_whole:
                    vgraph_all = capture_val_all()
                elif Xva.shape[0] >= bs:
                    vgraph, vX, vY = capture_val(bs)
                if not val_whole and nvt >= 1 and not dp and switches.get("tail_graph") == "1":
                    vgraph_t, vXt, vYt = capture_val(nvt)
            except Exception as e:
                if log:
                    log(f"HIP graph capture of the validation pass failed ({e!r}); evaluating eagerly")
                vgraph = vgraph_t = vgraph_all = None
                torch.cuda.synchronize(device)
            v_acc.zero_()
            net.train()

    # "the warm-up steps never happened" for the dropout / noise streams too: their call counters go back to zero, so that a seeded run
    # draws the same masks however many warm-up and capture passes its launch mode needed (one graph, two graphs around the collective,
    # with or without a tail graph, eager) -- r05: the one-rank data-parallel run reproduces the plain run
    if on_gpu:
        prep_counter.zero_()
        if engine is not None:
            engine.drop_counter.zero_(); engine.prep_counter.zero_()
        if fuse_gather:
            g_cursor.zero_()
    best_val, best_state, no_improve = float("inf"), None, 0
    n_epochs = max_epochs if max_epochs is not None else cfg.num_epochs
    # the epochs' shuffles come from a generator of their own: the framework's global one is consumed by eager steps (input noise,
    # framework dropout) but not by graph replays, so the batch order of epoch 2 would depend on which steps happened to run eagerly
    perm_gen = torch.Generator(device=device)
    perm_gen.manual_seed(seed * 1000003 + 17 + rank)
    seg_ev = [] if (_DP_PROFILE and on_gpu and graph is not None) else None      # per-step event quadruples (first epoch excluded below)
    first_one_graph_replay = bool(graph_mode_one)        # the captured collective's first replay runs under the stall timer
    for epoch in range(1, n_epochs + 1):
        if device.type == "cuda":
            torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        net.train()
        noise = cfg.sigma_0 * (cfg.gamma_noise ** epoch)                     # PINN:743
        if batch_order is not None:
            order = torch.as_tensor(batch_order(epoch), dtype=torch.long).reshape(-1).to(device)
        else:
            order = torch.randperm(Xtr.shape[0], device=device, generator=perm_gen)      # DataLoader(shuffle=True), PINN:701
        tot = torch.zeros((), device=device)
        if engine is not None:
            engine.loss_sum.zero_()                                          # the output launch adds every step's loss to it
        if loss_acc is not None:
            loss_acc.zero_()                                                 # ... and so does the fused loss launch of the other loops
        noise_t = torch.tensor(noise, device=device)
        if graph is not None:
            s_noise.copy_(noise_t)                                           # constant within the epoch
        if fuse_gather:                                                      # the front-end launch walks this permutation with its own cursor
            g_order[:order.numel()].copy_(order)
            g_cursor.zero_()
            g_noise.copy_(noise_t)
        for b in range(nb_tr):
            # `order` is the DataLoader shuffle; permute_data (PINN:753) re-permutes inside the batch, which
            # changes neither the batch statistics nor the mean loss, so it is folded into `order`
            idx = order[b * bs:(b + 1) * bs]
            if graph is not None and idx.numel() == bs:
                got_y = False
                if fuse_gather:
                    got_y = True                                             # (the captured step's first launch assembles the batch itself)
                elif _FUSED_PREP or engine is not None:
                    got_y = gather_noise(idx, sX, sY)                        # gather + noise (+ bf16 cast) + targets in one launch
                else:
                    torch.index_select(Xtr, 0, idx, out=sX)                  # gather straight into the graph's input buffers
                if engine is None and not got_y:
                    torch.index_select(Ytr, 0, idx, out=sY)
                if sP is not None and not fuse_gather:
                    physics_inputs(idx, out=sP)
                if first_one_graph_replay:
                    first_one_graph_replay = False
                    with _StallGuard(_DP_STALL_S, "the first replay of the data-parallel step graph (gradient all-reduce captured)"):
                        graph.replay()
                        torch.cuda.synchronize(device)
                    if engine is None and loss_acc is None:
                        tot += s_loss
                    continue
                if seg_ev is not None and len(seg_ev) < 4096:
                    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                    ev[0].record()
                    graph.replay()
                    ev[1].record()
                    if graph_b is not None:
                        allreduce_grads()
                        ev[2].record()
                        graph_b.replay()
                    else:
                        ev[2].record()
                    ev[3].record()
                    seg_ev.append(ev)
                else:
                    graph.replay()
                    if graph_b is not None:
                        allreduce_grads()
                        graph_b.replay()
                if engine is None and loss_acc is None:
                    tot += s_loss
                if _DEBUG_NAN and engine is None:
                    _debug_nan(model, opt, flat, s_loss, epoch, b, sX, sY)
            elif graph_t is not None and idx.numel() == sYt.shape[0]:
                got_y = False
                if engine is not None:
                    got_y = gather_noise(idx, None)                          # into the engine's buffers
                elif fuse_gather:
                    got_y = True
                elif _FUSED_PREP:
                    got_y = gather_noise(idx, sXt, sYt)
                else:
                    torch.index_select(Xtr, 0, idx, out=sXt)
                if not got_y:
                    torch.index_select(Ytr, 0, idx, out=sYt)
                graph_t.replay()
                if loss_acc is None and engine is None:
                    tot += s_loss_t
                if _DEBUG_NAN and engine is None:
                    _debug_nan(model, opt, flat, s_loss_t, epoch, -b, sXt, sYt)
            elif engine is not None:
                train_step(None, Ytr[:idx.numel()], noise_t, idx)            # Yb only carries the row count here
            else:
                if shared_counter and not fuse_gather:
                    prep_counter[0:1].add_(1)                                # (no batch-assembly launch on this path: advance the streams here)
                step_loss = train_step(Xtr[idx], Ytr[idx], noise_t, idx)
                if loss_acc is None:
                    tot += step_loss
                if _DEBUG_NAN and engine is None:
                    _debug_nan(model, opt, flat, step_loss, epoch, -b, Xtr[idx], Ytr[idx])
        if engine is not None:
            tot = engine.loss_sum.clone()
        elif loss_acc is not None:
            tot = loss_acc.clone()
        train_loss = _allreduce_mean(tot / nb_tr, world, dp)
        if dp:   # BatchNorm running statistics are per rank during the epoch: average them before evaluating
            for buf in model.buffers():
                if buf.is_floating_point():
                    dist.all_reduce(buf)
                    buf.div_(world)
        net.eval()
        if engine is not None and engine.repack_params is not None and engine_eval:
            engine.repack_now()          # (the engine's evaluation launches read the weight copies: the last update's are not built yet)
        vt = torch.zeros((), device=device)
        nb_va = max(1, (Xva.shape[0] + cfg.batch_size - 1) // cfg.batch_size)
        nb_run = nb_va
        if engine_eval and slot_graph is not None:
            slot_graph.replay()
            nb_run = 0
        elif engine_eval:
            engine.eval_loss_sum.zero_()
            if val_rows is None:
                val_rows = torch.arange(Xva.shape[0], device=device)
        elif val_whole:
            if vgraph_all is not None:
                vgraph_all.replay()
                vt += v_acc
                v_acc.zero_()
            else:
                val_all(vt)
            nb_run = 0
        for b in range(nb_run):
            sl = slice(b * cfg.batch_size, (b + 1) * cfg.batch_size)
            if engine_eval:
                rows = val_rows[sl]
                gk = ev_graphs.get(int(rows.numel()))
                if gk is not None:
                    gk[1].copy_(rows)
                    gk[0].replay()
                else:
                    engine.gather(Xva, Yva, rows, None, 0)
                    engine.evaluate(int(rows.numel()))
                continue
            if vgraph is not None and Xva[sl].shape[0] == bs:
                vX.copy_(Xva[sl]); vY.copy_(Yva[sl])
                vgraph.replay()
            elif vgraph_t is not None and Xva[sl].shape[0] == vXt.shape[0]:
                vXt.copy_(Xva[sl]); vYt.copy_(Yva[sl])
                vgraph_t.replay()
            else:
                vt += val_batch(Xva[sl], Yva[sl])
        if engine_eval and slot_graph is not None:
            vt = engine.eval_slot_losses().sum()
        elif engine_eval:
            vt = engine.eval_loss_sum.clone()
        elif (vgraph is not None or vgraph_t is not None) and not val_whole:
            vt += v_acc
            v_acc.zero_()
        val_loss = _allreduce_mean(vt / nb_va, world, dp)
        if sched is not None:
            sched.step()                                                     # PINN:788
        else:
            opt.lr.mul_(cfg.gamma)                                           # ExponentialLR on the device scalar
        tl, vl = float(train_loss), float(val_loss)                          # the epoch's only host syncs
        if device.type == "cuda":
            torch.cuda.synchronize(device)
        hist["train"].append(tl); hist["val"].append(vl); hist["epoch_s"].append(time.perf_counter() - t0)
        if vl < best_val:                                                    # PINN:791-799
            best_val, no_improve = vl, 0
            best_state = copy.deepcopy(model.state_dict())
        else:
            no_improve += 1
        if log and rank == 0:
            log(f"Epoch {epoch}/{n_epochs} | Train Loss={tl:.6f}, Val Loss={vl:.6f}, Time={hist['epoch_s'][-1]:.2f}s")
        if no_improve >= cfg.patience:
            break
    # the captured graphs (and their private memory pools) go before anything else runs on this device: the closures above form
    # reference cycles that would otherwise keep them alive until some later garbage collection
    step_was_captured = graph is not None
    graph = graph_b = vgraph = graph_t = vgraph_t = slot_graph = vgraph_all = None
    ev_graphs.clear()
    if on_gpu:
        import gc
        gc.collect()
        torch.cuda.synchronize(device)
        # BLAS workspaces that were allocated while capturing live in the graphs' private pools but stay cached per (handle, stream):
        # a later run whose side stream gets the same id would hand that freed memory to its GEMMs as scratch space (seen as NaNs /
        # drifting losses in a framework-path TFD run after a fast-path one in the same process)
        clear_blas_workspaces()
    disable_shadow_linears(patched)      # the returned model is a plain module again
    if fast_encoder is not None:
        from . import tfd_fused
        tfd_fused.disarm_gather()
        tfd_fused.unpatch_model(fast_encoder)
    if hasattr(model, "direct_param_grads"):
        model.direct_param_grads = False
    if best_state is not None:
        model.load_state_dict(best_state)
    model.eval()
    with torch.no_grad():
        preds = torch.cat([model(Xva[i:i + cfg.batch_size]).float() for i in range(0, Xva.shape[0], cfg.batch_size)])
    nel = cfg.nelem
    sI = data.scalers_Y["I"]
    p = sI.inverse_transform(preds[:, :nel]).clamp(0.0, 1e10)                # PINN:843-848
    t = sI.inverse_transform(Yva[:, :nel]).clamp(0.0, 1e10)
    # (the reference's evaluation block, PINN:815-852 / TFD:800-829: best checkpoint reloaded, evaluation-mode pass over the validation
    #  set in loader order, inertias un-standardised and clipped to [0, 1e10], r2_score on the raveled arrays)
    out = {"model": model, "history": hist, "best_val": best_val, "best_state": best_state, "r2_val_I": r2_score(t, p),
           "val_pred_I": p, "val_true_I": t, "epochs": len(hist["train"]), "steps_per_epoch": nb_tr}
    if fp16_scaler is not None:
        out["grad_scaler"] = {"scale": fp16_scaler["scale"], "skipped_steps": fp16_scaler["skipped"], "steps": fp16_scaler["steps"]}
    if dp:
        out["dp_mode"] = {"world": world, "forced_one_rank": bool(world == 1), "backend": dist.get_backend(), "async": bool(_DP_ASYNC),
                          "step": ("one graph incl. the all-reduce" if graph_mode_one else "graph A | all-reduce | graph B" if step_was_captured
                                   else "eager")}
    if seg_ev:                           # mean device time of the step's segments over the profiled steps (first epoch's excluded)
        use = seg_ev[nb_tr:] or seg_ev
        mean = lambda i, j: 1e3 * sum(e[i].elapsed_time(e[j]) for e in use) / len(use)      # noqa: E731
        out["dp_segments"] = {"steps": len(use), "step_us": mean(0, 3), "graph_a_us": mean(0, 1), "allreduce_us": mean(1, 2),
                              "graph_b_us": mean(2, 3), "one_graph": bool(graph_mode_one), "world": world}
    return out


def save_best(state: Dict[str, torch.Tensor], path: str) -> None:
    """`torch.save(model.state_dict(), "best_model_*.pth")` (PINN:794, TFD:777)."""
    torch.save(state, path)
