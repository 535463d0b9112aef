"""The PINN's training step as 14 launches (r05; r02: 15, r04: 16 with the optimiser's three): host side of csrc/mlp_block.hip.

`FNNWithResidual` (/root/reference/OpenPyStruct_PINN_MultiCase.py:454-541) + `CompositeLoss` (:603-653), forward AND backward,
for batches of up to 128 rows, without autograd: every `Linear -> [stencil + residual] -> BatchNorm1d -> LeakyReLU -> dropout`
group is one launch, so is each backward counterpart, all weight gradients are one grouped launch, and every parameter
gradient is written straight into the training loop's flat gradient buffer.  The module itself stays the owner of the
parameters and BatchNorm buffers (the launches read and update them in place); evaluation keeps using the module.

    gather (+ the last update's weight copies, r05) | input | (fc1, fc2+stencil+norm) x blocks | output+loss || d output | (d fc2, d fc1) x blocks |
    weight gradients (+ the clip norm's partial sums, r05) | Adam

Layout contract (include/openpystruct_amd.h): every activation / gradient exists as X (rows = batch) and as Xt (rows =
columns of X) in bfloat16, both FRAGMENT-TILED (1 KB tiles in MFMA lane order: `to_tiled` / `from_tiled`), zero outside the
live corner; the buffers below are zero-initialised once and the launches only ever write zeros into dead rows / columns.
"""
from __future__ import annotations

import ctypes
import os
from typing import List

import torch
import torch.nn as nn

from . import _cabi, switches
from .surrogates import CompositeLoss, FNNWithResidual, ResidualBlock

ENABLED = switches.get("pinn_layer_blocks") == "1"      # A/B switch: 0 = autograd over the fused tails


def _ru(v: int, m: int) -> int:
    return (v + m - 1) // m * m


def to_tiled(x: torch.Tensor) -> torch.Tensor:
    """[rows, K] (rows % 16 == 0, K % 32 == 0) -> the fragment-tiled storage: tile (row >> 4, k >> 5) of 512 elements, inside it
    element (row, k) at ((k >> 3) & 3) * 128 + (row & 15) * 8 + (k & 7)."""
    R, K = x.shape
    return x.reshape(R // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().reshape(R, K)


def from_tiled(t: torch.Tensor) -> torch.Tensor:
    """Inverse of `to_tiled` (t carries the logical shape [rows, K] of the matrix it stores)."""
    R, K = t.shape
    return t.reshape(R // 16, K // 32, 4, 16, 8).permute(0, 3, 1, 2, 4).contiguous().reshape(R, K)


def eligible(model: nn.Module, crit: nn.Module, batch_size: int) -> bool:
    """The launches cover exactly the reference's configuration family: BatchNorm1d norms, blocks with the 3-tap Conv1d
    path, batches of up to 128 rows, float32 parameters."""
    if not (ENABLED and isinstance(model, FNNWithResidual) and isinstance(crit, CompositeLoss)):
        return False
    if model.norm_type != "batch" or type(model.input_norm) is not nn.BatchNorm1d or batch_size > _cabi.MLP_MAX_ROWS:
        return False
    if len(model.residual_blocks) < 1:
        return False
    for blk in model.residual_blocks:
        rb, norm = blk[0], blk[1]
        if not (isinstance(rb, ResidualBlock) and rb.use_conv and rb.conv1.kernel_size == (3,) and rb.conv1.padding == (1,)
                and rb.conv1.bias is not None and type(norm) is nn.BatchNorm1d and rb.bn1.affine and rb.bn1.track_running_stats):
            return False
    for bn in [model.input_norm] + [b[1] for b in model.residual_blocks] + [b[0].bn1 for b in model.residual_blocks]:
        if not (bn.affine and bn.track_running_stats and bn.momentum is not None):
            return False
    return all(p.dtype == torch.float32 and p.is_cuda for p in model.parameters())


class PinnFusedStep:
    """Buffers and launch descriptors of one model; `gather()` assembles a batch, `fwd_bwd(B, targets)` leaves the loss in
    `self.loss` and the gradients of the mean batch loss in the parameters' `.grad` (views of the caller's flat buffer)."""

    def __init__(self, model: FNNWithResidual, crit: CompositeLoss, seed: int = 0):
        self.lib = _cabi.load()
        self.model, self.crit = model, crit
        dev = next(model.parameters()).device
        self.dev = dev
        for p in model.parameters():
            if p.grad is None or p.grad.dtype != torch.float32 or not p.grad.is_contiguous():
                raise ValueError("every parameter needs a contiguous float32 .grad (the training loop's flat buffer)")
        R = _cabi.MLP_MAX_ROWS
        bf = torch.bfloat16
        self.F_in = model.input_fc.in_features
        self.H = model.input_fc.out_features
        self.C = model.output_fc.out_features
        self.nblk = len(model.residual_blocks)
        self.Hh = model.residual_blocks[0][0].fc1.out_features
        self._keep: List[object] = []

        def act(F):     # row-major [128, ld] + transposed [F r.u. 32, 128]
            return (torch.zeros(R, _ru(F, 32), dtype=bf, device=dev), torch.zeros(_ru(F, 32), R, dtype=bf, device=dev))

        def tr(F):
            return torch.zeros(_ru(F, 32), R, dtype=bf, device=dev)

        def wpair(lin: nn.Linear):
            N, K = lin.weight.shape
            return (torch.zeros(_ru(N, 16), _ru(K, 32), dtype=bf, device=dev), torch.zeros(_ru(K, 16), _ru(N, 32), dtype=bf, device=dev))

        H, Hh, C, Fi = self.H, self.Hh, self.C, self.F_in
        self.x, self.xt = act(Fi)
        self.o = [act(H) for _ in range(self.nblk + 1)]            # o[0] = input layer output, o[k] = block k output
        self.v0t = tr(H)                                           # input layer: pre-normalisation values
        self.h = [act(Hh) for _ in range(self.nblk)]
        self.zt = [tr(H) for _ in range(self.nblk)]                # block sums before the norm
        self.preds = torch.zeros(R, _ru(C, 32), dtype=bf, device=dev)
        self.gp, self.gpt = act(C)                                 # d loss / d preds
        self.dz = [act(H) for _ in range(self.nblk + 1)]           # dz[0] = gradient at input_fc's output, dz[k] = at block k's sum
        self.dh = [act(Hh) for _ in range(self.nblk)]
        f32 = lambda n: torch.zeros(n, dtype=torch.float32, device=dev)      # noqa: E731
        self.mean = [f32(H) for _ in range(self.nblk + 1)]
        self.rstd = [f32(H) for _ in range(self.nblk + 1)]
        self.ssave = [f32(2) for _ in range(self.nblk)]
        nsp = int(self.lib.ops_mlp_spart_doubles(H))
        self.spart_f = [torch.zeros(nsp, dtype=torch.float64, device=dev) for _ in range(self.nblk)]
        self.spart_b = [torch.zeros(nsp, dtype=torch.float64, device=dev) for _ in range(self.nblk)]
        self.loss = torch.zeros((), dtype=torch.float32, device=dev)
        self.loss_sum = torch.zeros((), dtype=torch.float32, device=dev)       # += loss per step; the caller zeroes it (per epoch)
        self.targets_t = torch.zeros(C, R, dtype=torch.float32, device=dev)    # the batch's targets, transposed
        self.loss_ws = torch.zeros(int(self.lib.ops_mlp_loss_workspace_bytes()), dtype=torch.uint8, device=dev)
        self.loss_ws_eval = torch.zeros(int(self.lib.ops_mlp_loss_workspace_bytes()), dtype=torch.uint8, device=dev)
        self.eval_loss = torch.zeros((), dtype=torch.float32, device=dev)
        self.eval_loss_sum = torch.zeros((), dtype=torch.float32, device=dev)   # += loss per evaluated batch; the caller zeroes it
        self.scratch16 = torch.zeros(R, 32, dtype=bf, device=dev)               # output of the evaluation pass's finish launch
        self.drop_counter = torch.zeros(2, dtype=torch.int64, device=dev)
        self.prep_counter = torch.zeros(2, dtype=torch.int64, device=dev)     # [calls, workgroups done] (csrc/call_counter.hpp)
        self.seed = int(seed) & 0x7FFFFFFFFFFFFFFF

        lins = [model.input_fc] + [l for b in model.residual_blocks for l in (b[0].fc1, b[0].fc2)] + [model.output_fc]
        if len(lins) > _cabi.MLP_MAX_WGRAD:
            raise ValueError("too many Linear layers for one grouped launch")
        self.wp = {id(l): wpair(l) for l in lins}
        ent = (_cabi.MlpRepackEntry * len(lins))()
        for e, l in zip(ent, lins):
            wp, wtp = self.wp[id(l)]
            e.W, e.N, e.K = l.weight.data_ptr(), l.weight.shape[0], l.weight.shape[1]
            e.Wp, e.ldw, e.Wtp, e.ldwt = wp.data_ptr(), wp.shape[1], wtp.data_ptr(), wtp.shape[1]
        self._repack = ent

        l1l2 = crit.l1l2_loss
        self._alpha = l1l2.alpha.detach()
        sc = lambda v: None if v is None else (v if torch.is_tensor(v) else torch.tensor(float(v))).to(device=dev, dtype=torch.float32).reshape(())   # noqa: E731
        self._minc, self._maxc = sc(l1l2.min_constraint), sc(l1l2.max_constraint)
        self._build()
        self.repack_now()

    # ---- launch descriptors (built once: every pointer is stable, so the sequence can be captured in a HIP graph) ----
    def _strip(self, **kw) -> _cabi.MlpStripArgs:
        a = _cabi.MlpStripArgs()
        for k, v in kw.items():
            if v is None:
                continue
            if torch.is_tensor(v):
                self._keep.append(v)
                v = v.data_ptr()
            setattr(a, k, v)
        return a

    def _bn_fwd(self, bn: nn.BatchNorm1d, i: int):
        return dict(gamma=bn.weight, beta=bn.bias, eps=float(bn.eps), momentum=float(bn.momentum), running_mean=bn.running_mean,
                    running_var=bn.running_var, num_batches_tracked=bn.num_batches_tracked, mean=self.mean[i], rstd=self.rstd[i])

    def _bn_bwd(self, bn: nn.BatchNorm1d, i: int):
        return dict(gamma=bn.weight, beta=bn.bias, eps=float(bn.eps), mean=self.mean[i], rstd=self.rstd[i], dgamma=bn.weight.grad,
                    dbeta=bn.bias.grad)

    def _stencil(self, rb: ResidualBlock, k: int):
        return dict(conv_w=rb.conv1.weight, conv_b=rb.conv1.bias, sgamma=rb.bn1.weight, sbeta=rb.bn1.bias, seps=float(rb.bn1.eps),
                    smomentum=float(rb.bn1.momentum), srunning_mean=rb.bn1.running_mean, srunning_var=rb.bn1.running_var,
                    snum_batches_tracked=rb.bn1.num_batches_tracked, ssave=self.ssave[k], No=self.H)

    def _build(self) -> None:
        m, C = self.model, _cabi
        H, Hh, Fi, Co = self.H, self.Hh, self.F_in, self.C
        slope, pd = float(m.Leaky.negative_slope), float(m.dropout.p)
        fwd, bwd = [], []
        wp, wtp = self.wp[id(m.input_fc)]
        fwd.append(self._strip(N=H, K=Fi, tail=C.MLP_TAIL_BN_ACT_DROP, A=self.x, lda=self.x.shape[1], W=wp, ldw=wp.shape[1],
                               bias=m.input_fc.bias, Y=self.o[0][0], ldy=self.o[0][0].shape[1], Yt=self.o[0][1], Zt=self.v0t,
                               slope=slope, p_drop=pd, seed=self.seed, call_counter=self.drop_counter, **self._bn_fwd(m.input_norm, 0)))
        for k, blk in enumerate(m.residual_blocks):
            rb, norm = blk[0], blk[1]
            o_in, o_in_t = self.o[k]
            w1, _ = self.wp[id(rb.fc1)]
            w2, _ = self.wp[id(rb.fc2)]
            fwd.append(self._strip(N=Hh, K=H, tail=C.MLP_TAIL_ACT_DROP, side=C.MLP_SIDE_FWD_STENCIL_STATS, A=o_in, lda=o_in.shape[1], W=w1,
                                   ldw=w1.shape[1], bias=rb.fc1.bias, Y=self.h[k][0], ldy=self.h[k][0].shape[1], Yt=self.h[k][1],
                                   slope=float(rb.Leaky.negative_slope), p_drop=float(rb.dropout.p), seed=self.seed + 2 * k + 1,
                                   call_counter=self.drop_counter, Ot=o_in_t, spart=self.spart_f[k], **self._stencil(rb, k)))
            fwd.append(self._strip(N=H, K=Hh, tail=C.MLP_TAIL_BN, add_mode=C.MLP_ADD_FWD_BLOCK, A=self.h[k][0], lda=self.h[k][0].shape[1],
                                   W=w2, ldw=w2.shape[1], bias=rb.fc2.bias, Y=self.o[k + 1][0], ldy=self.o[k + 1][0].shape[1],
                                   Yt=self.o[k + 1][1], Zt=self.zt[k], Ot=o_in_t, spart=self.spart_f[k],
                                   **self._bn_fwd(norm, k + 1), **self._stencil(rb, k)))
        wo, wot = self.wp[id(m.output_fc)]
        o_last = self.o[self.nblk]
        crit, l1l2 = self.crit, self.crit.l1l2_loss
        fwd.append(self._strip(N=Co, K=H, tail=C.MLP_TAIL_LOSS, A=o_last[0], lda=o_last[0].shape[1], W=wo, ldw=wo.shape[1],
                               bias=m.output_fc.bias, Y=self.gp, ldy=self.gp.shape[1], Yt=self.gpt, P=self.preds, ldp=self.preds.shape[1],
                               nI=crit.nelem, nD=crit.deflection_dim, alpha=self._alpha, alpha0=float("nan"), min_constraint=self._minc,
                               max_constraint=self._maxc, box_weight=float(l1l2.penalty_weight), rel_penalty=float(crit.penalty_pinn),
                               loss_ws=self.loss_ws, targets_t=self.targets_t,
                               dbias=m.output_fc.bias.grad))
        # backward: d preds -> gradient at the last block's sum (through its norm)
        last_blk = m.residual_blocks[self.nblk - 1]
        bwd.append(self._strip(N=H, K=Co, tail=C.MLP_TAIL_BWD_BN, A=self.gp, lda=self.gp.shape[1], W=wot, ldw=wot.shape[1],
                               Y=self.dz[self.nblk][0], ldy=self.dz[self.nblk][0].shape[1], Yt=self.dz[self.nblk][1], Zt=self.zt[self.nblk - 1],
                               dbias=last_blk[0].fc2.bias.grad, loss_ws=self.loss_ws, loss_finish_rows=(Co + 15) // 16, loss_C=Co,
                               nI=crit.nelem, nD=crit.deflection_dim, alpha=self._alpha, alpha0=float("nan"),
                               box_weight=float(l1l2.penalty_weight), rel_penalty=float(crit.penalty_pinn), loss=self.loss,
                               loss_sum=self.loss_sum, **self._bn_bwd(last_blk[1], self.nblk)))
        for k in range(self.nblk - 1, -1, -1):
            rb = m.residual_blocks[k][0]
            o_in, o_in_t = self.o[k]
            dzk, dzk_t = self.dz[k + 1]
            _, w1t = self.wp[id(rb.fc1)]
            _, w2t = self.wp[id(rb.fc2)]
            bwd.append(self._strip(N=Hh, K=H, tail=C.MLP_TAIL_BWD_ACT_DROP, side=C.MLP_SIDE_BWD_STENCIL_SUMS, A=dzk, lda=dzk.shape[1], W=w2t,
                                   ldw=w2t.shape[1], Y=self.dh[k][0], ldy=self.dh[k][0].shape[1], Yt=self.dh[k][1], Yref_t=self.h[k][1],
                                   slope=float(rb.Leaky.negative_slope), p_drop=float(rb.dropout.p), dbias=rb.fc1.bias.grad, Ot=o_in_t,
                                   dZt=dzk_t, spart=self.spart_b[k], **self._stencil(rb, k)))
            if k > 0:
                prev = m.residual_blocks[k - 1]
                tail = dict(tail=C.MLP_TAIL_BWD_BN, Zt=self.zt[k - 1], dbias=prev[0].fc2.bias.grad, **self._bn_bwd(prev[1], k))
            else:
                tail = dict(tail=C.MLP_TAIL_BWD_BN_ACT_DROP, Zt=self.v0t, Yref_t=self.o[0][1], slope=slope, p_drop=pd,
                            dbias=m.input_fc.bias.grad, **self._bn_bwd(m.input_norm, 0))
            if rb.conv1.bias.grad.data_ptr() != rb.conv1.weight.grad.data_ptr() + 12 or rb.bn1.weight.grad.data_ptr() != \
                    rb.conv1.weight.grad.data_ptr() + 16 or rb.bn1.bias.grad.data_ptr() != rb.conv1.weight.grad.data_ptr() + 20:
                raise ValueError("conv1.weight/.bias and bn1.weight/.bias gradients must be six consecutive floats of the flat buffer")
            bwd.append(self._strip(N=H, K=Hh, add_mode=C.MLP_ADD_BWD_BLOCK, A=self.dh[k][0], lda=self.dh[k][0].shape[1], W=w1t,
                                   ldw=w1t.shape[1], Y=self.dz[k][0], ldy=self.dz[k][0].shape[1], Yt=self.dz[k][1], Ot=o_in_t, dZt=dzk_t,
                                   spart=self.spart_b[k], sdparams=rb.conv1.weight.grad, **tail, **self._stencil(rb, k)))
        self._fwd, self._bwd = fwd, bwd
        # evaluation pass (model.eval(): running statistics, no dropout, no stencil statistics) = the forward stages with eval_stats,
        # its loss partial sums in a workspace of their own, and ONE more (dummy, one-strip) launch that adds them up
        ev = []
        for a in fwd:
            e = _cabi.MlpStripArgs()
            ctypes.memmove(ctypes.addressof(e), ctypes.addressof(a), ctypes.sizeof(_cabi.MlpStripArgs))
            e.eval_stats, e.p_drop, e.side = 1, 0.0, C.MLP_SIDE_NONE
            if e.tail == C.MLP_TAIL_LOSS:
                e.loss_ws = self.loss_ws_eval.data_ptr()
                e.dbias = None
            ev.append(e)
        ev.append(self._strip(N=16, K=32, tail=C.MLP_TAIL_NONE, A=self.x, lda=self.x.shape[1], W=wp, ldw=wp.shape[1], Y=self.scratch16,
                              ldy=self.scratch16.shape[1], loss_ws=self.loss_ws_eval, loss_finish_rows=(Co + 15) // 16, loss_C=Co,
                              nI=crit.nelem, nD=crit.deflection_dim, alpha=self._alpha, alpha0=float("nan"),
                              box_weight=float(l1l2.penalty_weight), rel_penalty=float(crit.penalty_pinn), loss=self.eval_loss,
                              loss_sum=self.eval_loss_sum))
        self._eval = ev
        # grouped weight gradients
        probs = []
        probs.append((self.gpt, o_last[1], m.output_fc.weight))
        for k in range(self.nblk):
            rb = m.residual_blocks[k][0]
            probs.append((self.dz[k + 1][1], self.h[k][1], rb.fc2.weight))
            probs.append((self.dh[k][1], self.o[k][1], rb.fc1.weight))
        probs.append((self.dz[0][1], self.xt, m.input_fc.weight))
        wg = (_cabi.MlpWgradProblem * len(probs))()
        for e, (at, bt, w) in zip(wg, probs):
            e.At, e.Bt, e.out, e.N, e.K, e.ldo = at.data_ptr(), bt.data_ptr(), w.grad.data_ptr(), w.shape[0], w.shape[1], w.shape[1]
        self._wgrad = wg
        self._norm = None        # (ranges ptr array, lengths, n, workspace, step, betas, nparts holder): enable_norm()
        self.repack_params = None   # flat float32 parameter buffer: training gathers also rebuild the bf16 weight copies (repack_in_gather)
        covered = sum(p.numel() for p in m.parameters())
        named = sum(l.weight.numel() + l.bias.numel() for l in [m.input_fc, m.output_fc] + [x for b in m.residual_blocks for x in (b[0].fc1, b[0].fc2)])
        named += sum(2 * bn.weight.numel() for bn in [m.input_norm] + [b[1] for b in m.residual_blocks]) + 6 * self.nblk
        if covered != named:
            raise ValueError("the model has parameters the layer-block launches do not cover")

    def enable_norm(self, flat_grad: torch.Tensor, workspace: torch.Tensor, step: torch.Tensor, betas, grad_scale: float = 1.0) -> int:
        """r05: let the weight-gradient launch leave the optimiser's gradient-norm partial sums in `workspace`
        (ops_flat_adam_workspace_bytes(); FlatClipAdam.ws) -- its own tiles' squares plus, in extra one-wave workgroups, the squares of
        every gradient no matrix covers (biases, normalisation and stencil parameters: written by the strip launches before it) -- and
        advance `step`: the optimiser then skips its norm launch (OPS_ADAM_NORM_READY).  `flat_grad`: the buffer every parameter's
        .grad is a view of.  Returns the number of partial sums (what FlatClipAdam.norm_ready_parts takes).  Only where the gradients
        are final when this launch ends: one rank (a data-parallel step all-reduces them afterwards)."""
        m = self.model
        base, n = flat_grad.data_ptr(), flat_grad.numel()
        mats = sorted((l.weight.grad.data_ptr(), l.weight.numel()) for l in
                      [m.input_fc, m.output_fc] + [x for b in m.residual_blocks for x in (b[0].fc1, b[0].fc2)])
        ranges, cur = [], base
        for ptr, cnt in mats:
            if not (base <= ptr and ptr + 4 * cnt <= base + 4 * n):
                raise ValueError("every weight gradient must be a view of the flat gradient buffer")
            if ptr > cur:
                ranges.append((cur, (ptr - cur) // 4))
            cur = ptr + 4 * cnt
        if cur < base + 4 * n:
            ranges.append((cur, (base + 4 * n - cur) // 4))
        if len(ranges) > _cabi.MLP_MAX_NORM_RANGES:
            raise ValueError("too many gradient ranges outside the weight matrices")
        rp = (ctypes.c_void_p * max(1, len(ranges)))(*[r[0] for r in ranges])
        rl = (ctypes.c_int32 * max(1, len(ranges)))(*[r[1] for r in ranges])
        self._norm = dict(rp=rp, rl=rl, nr=len(ranges), ws=workspace, step=step, betas=(float(betas[0]), float(betas[1])),
                          scale=float(grad_scale), nparts=ctypes.c_int32(0), covered=sum(c for _, c in mats) + sum(r[1] for r in ranges))
        assert self._norm["covered"] == n
        # (the count is a function of the problem shapes alone: a dry computation of the launch geometry)
        tiles = sum(((w.shape[0] + 31) // 32) * ((w.shape[1] + 31) // 32) for w in
                    [l.weight for l in [m.input_fc, m.output_fc] + [x for b in m.residual_blocks for x in (b[0].fc1, b[0].fc2)]])
        if tiles + len(ranges) > _cabi.FLAT_ADAM_MAX_PARTS:
            self._norm = None
            raise ValueError("more partial sums than the optimiser's workspace holds")
        return tiles + len(ranges)

    def _check(self, rc: int, what: str) -> None:
        if rc != _cabi.OK:
            raise RuntimeError(f"{what} failed with code {rc}: {self.lib.ops_amd_last_error().decode()}")

    # ---- the step ----
    def repack_in_gather(self, flat_params: torch.Tensor) -> None:
        """r05: from now on every TRAINING gather (sigma given) also rebuilds the tiled bf16 weight copies from `flat_params` (the
        optimiser's flat float32 parameter buffer, every Linear weight a view of it) in extra workgroups of its launch -- the optimiser is
        then run WITHOUT its repack launch (FlatClipAdam.repack = None).  The copies are one optimiser step old between an update and the
        next training gather: call `repack_now()` before reading them elsewhere (an evaluation pass)."""
        if flat_params.dtype != torch.float32 or not flat_params.is_contiguous():
            raise ValueError("flat_params must be the contiguous float32 parameter buffer")
        self.repack_params = flat_params

    def gather(self, X: torch.Tensor, Y: torch.Tensor, idx: torch.Tensor, sigma: torch.Tensor, seed: int) -> int:
        """x <- X[idx] + sigma * N(0, 1) in both layouts, targets_t <- Y[idx]^T (PINN:748-756), one launch.  Returns the live rows."""
        B = int(idx.numel())
        if X.dtype != torch.float32 or not X.is_contiguous() or Y.dtype != torch.float32 or not Y.is_contiguous() or Y.shape[1] != self.C:
            raise ValueError("X and Y must be contiguous float32 matrices")
        s = torch.cuda.current_stream(self.dev).cuda_stream
        if self.repack_params is not None and sigma is not None:        # a training step's batch: + the weight copies of the last update
            with torch.cuda.device(self.dev):
                self._check(self.lib.ops_mlp_gather_noise_repack(B, self.F_in, X.data_ptr(), idx.data_ptr(), sigma.data_ptr(),
                                                                 int(seed) & 0x7FFFFFFFFFFFFFFF, self.prep_counter.data_ptr(), self.x.data_ptr(),
                                                                 self.x.shape[1], self.xt.data_ptr(), Y.data_ptr(), self.C, self.targets_t.data_ptr(),
                                                                 self.repack_params.numel(), self.repack_params.data_ptr(), len(self._repack),
                                                                 self._repack, s), "ops_mlp_gather_noise_repack")
            return B
        with torch.cuda.device(self.dev):
            self._check(self.lib.ops_mlp_gather_noise(B, self.F_in, X.data_ptr(), idx.data_ptr(), sigma.data_ptr() if sigma is not None else None,
                                                      int(seed) & 0x7FFFFFFFFFFFFFFF, self.prep_counter.data_ptr(), self.x.data_ptr(),
                                                      self.x.shape[1], self.xt.data_ptr(), Y.data_ptr(), self.C, self.targets_t.data_ptr(), s),
                        "ops_mlp_gather_noise")
        return B

    def set_batch(self, Xb: torch.Tensor, Yb: torch.Tensor) -> int:
        """Tests / eager callers: a ready batch [B, F_in], [B, C] instead of gather()."""
        B = Xb.shape[0]
        R = _cabi.MLP_MAX_ROWS
        xb = torch.zeros(R, self.x.shape[1], dtype=torch.bfloat16, device=self.dev)
        xb[:B, :self.F_in] = Xb.to(torch.bfloat16)
        self.x.copy_(to_tiled(xb))
        self.xt.copy_(to_tiled(xb.t().contiguous()))
        self.targets_t.zero_()
        self.targets_t[:, :B] = Yb.to(torch.float32).t()
        return B

    def read(self, buf: torch.Tensor, rows: int, cols: int) -> torch.Tensor:
        """The live [rows, cols] corner of a tiled buffer, as a plain matrix."""
        return from_tiled(buf)[:rows, :cols]

    def repack_now(self) -> None:
        """bf16 copies of the weights in both layouts from the float32 parameters.  The optimiser refreshes them in its own
        update launch (FlatClipAdam.repack); call this after anything else changed the parameters (load_state_dict)."""
        with torch.cuda.device(self.dev):
            self._check(self.lib.ops_mlp_repack_weights(len(self._repack), self._repack, torch.cuda.current_stream(self.dev).cuda_stream),
                        "ops_mlp_repack_weights")

    def forward(self, B: int, stream=None) -> None:
        s = stream if stream is not None else torch.cuda.current_stream(self.dev).cuda_stream
        for a in self._fwd:
            a.B = B
            self._check(self.lib.ops_mlp_strip_launch(ctypes.byref(a), s), "ops_mlp_strip_launch (forward)")

    def fwd_bwd(self, B: int, repack: bool = False) -> torch.Tensor:
        """Loss of the batch in x / targets_t (mean over its B rows; also added to `loss_sum`) and all parameter gradients.
        `repack`: refresh the weight copies first (callers without an optimiser that does it)."""
        if B < 2:
            raise ValueError("Expected more than 1 value per channel when training (BatchNorm1d batch statistics)")
        s = torch.cuda.current_stream(self.dev).cuda_stream
        with torch.cuda.device(self.dev):
            if repack:
                self.repack_now()
            self.forward(B, s)
            for a in self._bwd:
                a.B = B
                self._check(self.lib.ops_mlp_strip_launch(ctypes.byref(a), s), "ops_mlp_strip_launch (backward)")
            if self._norm is not None:
                nm = self._norm
                self._check(self.lib.ops_mlp_wgrad_group_norm(len(self._wgrad), self._wgrad, nm["nr"], nm["rp"], nm["rl"], nm["scale"],
                                                              nm["ws"].data_ptr(), nm["step"].data_ptr(), nm["betas"][0], nm["betas"][1],
                                                              ctypes.byref(nm["nparts"]), s), "ops_mlp_wgrad_group_norm")
            else:
                self._check(self.lib.ops_mlp_wgrad_group(len(self._wgrad), self._wgrad, s), "ops_mlp_wgrad_group")
        return self.loss

    def evaluate(self, B: int) -> torch.Tensor:
        """model.eval() forward of the batch in x / targets_t (gather with sigma = None) and its loss (CompositeLoss mean over the B rows;
        also added to `eval_loss_sum`): BatchNorm layers normalise with their running statistics, no dropout, nothing is updated.
        The predictions are in `predictions(B)` afterwards."""
        s = torch.cuda.current_stream(self.dev).cuda_stream
        with torch.cuda.device(self.dev):
            for a in self._eval:
                a.B = B
                self._check(self.lib.ops_mlp_strip_launch(ctypes.byref(a), s), "ops_mlp_strip_launch (evaluation)")
        return self.eval_loss

    def predictions(self, B: int) -> torch.Tensor:
        return self.preds[:B, :self.C]

    # ---- evaluation slots: a whole validation set per launch sequence (grid.y = batches) ----
    def make_eval_slots(self, X: torch.Tensor, Y: torch.Tensor, batch: int) -> int:
        """The rows of (X, Y) as consecutive batches of `batch` rows, every batch in an arena ("slot") of its own that holds a copy
        of each buffer the evaluation pass writes -- same layout in every slot, so one launch serves all of them (ops_mlp_strip_args
        n_slots / slot_stride: the kernel adds blockIdx.y * slot_stride to the per-batch pointers).  The batches are gathered here, ONCE
        (a validation set never changes); `evaluate_slots()` is then len(self._eval) launches for the whole set, and slot i's batch
        loss lands in `eval_slot_losses()[i]`.  Returns the number of slots."""
        n_rows = int(X.shape[0])
        if not 1 <= batch <= _cabi.MLP_MAX_ROWS or n_rows < 1:
            raise ValueError("evaluation slots need 1 .. 128 rows per batch")
        S = (n_rows + batch - 1) // batch
        written = [self.x, self.xt, self.v0t, self.preds, self.gp, self.gpt, self.targets_t, self.loss_ws_eval, self.scratch16, self.eval_loss, self.eval_loss_sum]
        written += [t for pair in self.o for t in pair] + [t for pair in self.h for t in pair] + list(self.zt)
        offs, total = {}, 0
        for b in written:
            offs[b.data_ptr()] = (total, b)
            total += _ru(b.numel() * b.element_size(), 256)
        self._arena = torch.zeros(S, total, dtype=torch.uint8, device=self.dev)
        base = self._arena.data_ptr()
        ptr_fields = [name for name, typ in _cabi.MlpStripArgs._fields_ if typ is ctypes.c_void_p]
        slotted = {"A", "Y", "Yt", "Zt", "Ot", "P", "targets_t", "loss_ws", "loss", "loss_sum"}       # what the kernel offsets per slot
        stages = []
        for a in self._eval:
            e = _cabi.MlpStripArgs()
            ctypes.memmove(ctypes.addressof(e), ctypes.addressof(a), ctypes.sizeof(_cabi.MlpStripArgs))
            for name in ptr_fields:
                v = getattr(e, name)
                if v is not None and v in offs:
                    if name not in slotted:
                        raise ValueError(f"evaluation stage field {name} points at a per-batch buffer the kernel does not offset")
                    setattr(e, name, base + offs[v][0])
                elif name in slotted and v is not None:
                    raise ValueError(f"evaluation stage field {name} is not one of the engine's per-batch buffers")
            e.B, e.n_slots, e.slot_total_rows, e.slot_stride = batch, S, n_rows, total
            stages.append(e)
        self._slot_stages, self._slot_count, self._slot_total = stages, S, total
        view = lambda i, b: self._arena[i, offs[b.data_ptr()][0]: offs[b.data_ptr()][0] + b.numel() * b.element_size()].view(b.dtype).view(b.shape)   # noqa: E731
        self._slot_loss_off = offs[self.eval_loss.data_ptr()][0]
        rows = torch.arange(n_rows, device=self.dev)
        s = torch.cuda.current_stream(self.dev).cuda_stream
        with torch.cuda.device(self.dev):
            for i in range(S):
                idx = rows[i * batch:(i + 1) * batch]
                x, xt, tt = view(i, self.x), view(i, self.xt), view(i, self.targets_t)
                self._check(self.lib.ops_mlp_gather_noise(int(idx.numel()), self.F_in, X.data_ptr(), idx.data_ptr(), None, 0, None, x.data_ptr(),
                                                          x.shape[1], xt.data_ptr(), Y.data_ptr(), self.C, tt.data_ptr(), s),
                            "ops_mlp_gather_noise (evaluation slot)")
        self._slot_preds = [view(i, self.preds) for i in range(S)]
        return S

    def evaluate_slots(self) -> None:
        """The evaluation pass of every slot: one launch per stage for the whole set."""
        s = torch.cuda.current_stream(self.dev).cuda_stream
        with torch.cuda.device(self.dev):
            for a in self._slot_stages:
                self._check(self.lib.ops_mlp_strip_launch(ctypes.byref(a), s), "ops_mlp_strip_launch (evaluation slots)")

    def eval_slot_losses(self) -> torch.Tensor:
        """[slots] float32 view: the batch loss of each slot after `evaluate_slots()` (CompositeLoss mean over the slot's rows)."""
        return self._arena[:, self._slot_loss_off:self._slot_loss_off + 4].view(torch.float32).reshape(-1)
