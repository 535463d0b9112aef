"""Fast path of the Transformer-Diffusion surrogate (ModelOnePassTransformerWithDiffusion,
/root/reference/OpenPyStruct_TransformerDiffusionModule_MultiCase.py:443-478, :539-575): host side of csrc/seq_layer.hip and csrc/seq_block.hip.

Sequences of 7 tokens ([CLS] + 6 load cases), bf16 autocast.  Autograd still drives the step, but every BLOCK is one autograd.Function
whose forward and backward are one launch each (r03):

    FrontFn          draws -> x_noisy -> diffusion MLP -> combine with [CLS] token and positional encoding
    EncoderLayerFn   in-projection, attention, out-projection, dropout + add + LayerNorm, feed-forward, dropout + add + LayerNorm
    HeadFn           fc1 -> LayerNorm -> ReLU -> dropout -> fc2 on the [CLS] rows

on fragment-tiled bf16 weight copies (enable_layer_tiles; rebuilt behind every Adam update) -- 16 kernel nodes per captured training
step where the framework's modules need ~260.  Weight / bias gradients leave through train.shadow_param_grads (one grouped split-row
launch per step).  Evaluation passes without gradients run the same launches with every dropout probability 0.  Blocks whose sizes do
not fit the one-launch kernels fall back to the r02 form: the separate launches of csrc/seq_block.hip (attention, dropout + add +
LayerNorm, ReLU + dropout, diffusion noise / combine) between the optimiser's bf16 shadow products (train._ShadowLinearFn); masks,
other activations, CPU: the framework's own forward.
"""
from __future__ import annotations

import ctypes
import os
import types
from typing import Optional

import torch
import torch.nn as nn

from . import _cabi, switches

ENABLED = switches.get("tfd_fast_encoder") == "1"      # A/B switch: 0 = nn.TransformerEncoder's own forward
LAYER_FWD = switches.get("tfd_layer_fwd") == "1"       # A/B switch: 0 = eight launches per layer forward instead of one
DRAW = switches.get("tfd_draw") == "1"                 # A/B switch: 0 = diffusion steps / noise from the framework generators
_TRACE_BWD = [] if switches.get("tfd_trace_bwd") else None   # diagnostics: stage stamps of every backward layer launch
KEEP_DRAWS = False                                                    # tests: every state keeps its last draws (`_State.draws`)
EVAL_FAST = switches.get("tfd_eval_fast") == "1"       # A/B switch: 0 = validation passes through the module's own forward
FRONT = switches.get("tfd_front") == "1"               # A/B switch: 0 = the diffusion front end as five + three launches
LN_PARTIALS = switches.get("tfd_ln_partials") == "1"   # A/B switch: 0 = LayerNorm gamma / beta gradients by float atomics in the layer launch
HEAD = switches.get("tfd_head") == "1"                 # A/B switch: 0 = the head as four launches per direction
IDENTITY_ACT = False      # verification only (tests): every ReLU of the one-launch kernels becomes the identity -- a smooth network
FRONT_GATHER = switches.get("tfd_front_gather") == "1"   # A/B switch: 0 = the batch assembly as a launch of its own per step
HEAD_LOSS = switches.get("tfd_head_loss") == "1"       # A/B switch: 0 = the training loss as launches of its own behind the head
LAYER_BWD = switches.get("tfd_layer_bwd") == "1"       # A/B switch: 0 = eight launches per layer backward instead of one


class _State:
    """The dropout stream of one encoder (its call counter must outlive a captured HIP graph) and how parameter gradients leave."""

    def __init__(self, device, seed: int, direct_param_grads: bool = False, counter: Optional[torch.Tensor] = None):
        self.device = device
        self.direct = bool(direct_param_grads)      # LayerNorm gamma / beta gradients ASSIGNED into their .grad (a flat buffer zeroed per step)
        self.seed = int(seed) & 0x7FFFFFFFFFFFFFFF
        # advanced once per encoder pass (`advance`), read by every launch -- or the CALLER's counter, which something else advances once
        # per training step (train.py: the batch-assembly launch's call counter, csrc/call_counter.hpp: no add node in the captured step)
        self.external = counter is not None and counter.device == torch.device(device) and counter.dtype == torch.int64
        self.counter = counter if self.external else torch.zeros(1, dtype=torch.int64, device=device)

        self._zeros = {}
        self._zeros16 = {}
        self.src16 = None        # bf16 copy of the encoder's input, handed over by the model's front end (DiffusionCombine)
        self.last16 = None       # bf16 copy of the encoder's output rows [T, d], for the model's head
        self.train_mode = True   # False: an evaluation pass through the one-launch kernels (every dropout probability 0)
        self.keep_draws = False  # tests: keep the front end's draws (t [B, Nc], eps [B, Nc, d]) of the last pass in `draws`
        self.draws = None

    def advance(self) -> None:
        if not self.external or not self.train_mode:        # (an evaluation pass has no batch-assembly launch in front of it)
            self.counter[0:1].add_(1)

    def zeros(self, shape) -> torch.Tensor:
        """A persistent float32 zero tensor (the `residual` of a plain LayerNorm through the dropout + add + LayerNorm launch)."""
        key = tuple(shape)
        if key not in self._zeros:
            self._zeros[key] = torch.zeros(key, dtype=torch.float32, device=self.device)
        return self._zeros[key]

    def zeros16(self, shape) -> torch.Tensor:
        """A persistent bfloat16 zero tensor (ClsRows.backward writes the [CLS] rows into it; every other row stays zero for good)."""
        key = tuple(shape)
        if key not in self._zeros16:
            self._zeros16[key] = torch.zeros(key, dtype=torch.bfloat16, device=self.device)
        return self._zeros16[key]

    def used(self, site: int) -> torch.Tensor:
        """Where a forward launch leaves the counter value it drew its mask from (one per call: its backward reads it)."""
        return torch.empty(1, dtype=torch.int64, device=self.device)


def _check(rc: int, what: str) -> None:
    if rc != _cabi.OK:
        raise RuntimeError(f"{what} failed with code {rc}: {_cabi.load().ops_amd_last_error().decode()}")


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


class SeqAttention(torch.autograd.Function):
    """ctx = dropout(softmax(q k^T / sqrt(dh))) v per sample and head; qkv [B * S, 3 d] bf16 (q | k | v per row)."""

    @staticmethod
    def forward(ctx, qkv, Bn, S, H, p, st: _State, site: int):
        lib = _cabi.load()
        qkv = qkv.contiguous()
        d = qkv.shape[1] // 3
        out = torch.empty((Bn * S, d), dtype=torch.bfloat16, device=qkv.device)
        used = st.used(site)
        with torch.cuda.device(qkv.device):
            _check(lib.ops_seq_attention_fwd(Bn, S, H, d // H, qkv.data_ptr(), out.data_ptr(), float(p), st.seed + 7919 * site,
                                             st.counter.data_ptr(), used.data_ptr(), _stream(qkv.device)), "ops_seq_attention_fwd")
        ctx.save_for_backward(qkv)
        ctx.cfg = (Bn, S, H, d, float(p), st.seed + 7919 * site, used)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _cabi.load()
        (qkv,) = ctx.saved_tensors
        Bn, S, H, d, p, seed, used = ctx.cfg
        g = g.contiguous()
        if g.dtype != torch.bfloat16:
            g = g.to(torch.bfloat16)
        dqkv = torch.empty_like(qkv)
        with torch.cuda.device(qkv.device):
            _check(lib.ops_seq_attention_bwd(Bn, S, H, d // H, qkv.data_ptr(), g.data_ptr(), dqkv.data_ptr(), p, seed, used.data_ptr(),
                                             _stream(qkv.device)), "ops_seq_attention_bwd")
        return dqkv, None, None, None, None, None, None


class DropoutAddLayerNorm(torch.autograd.Function):
    """(y32, y16) = LayerNorm(res + dropout(x)): x [T, d] bf16, res float32 or bf16.  With `st.direct` the gradients of gamma / beta
    are added straight into their `.grad` (float32 views of a buffer the caller zeroes every step: the training loop's flat one)."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, eps, p, st: _State, site: int):
        lib = _cabi.load()
        x, res = x.contiguous(), res.contiguous()
        T, d = x.shape
        dev = x.device
        y32 = torch.empty((T, d), dtype=torch.float32, device=dev)
        y16 = torch.empty((T, d), dtype=torch.bfloat16, device=dev)
        z = torch.empty((T, d), dtype=torch.float32, device=dev)
        mean = torch.empty(T, dtype=torch.float32, device=dev)
        rstd = torch.empty(T, dtype=torch.float32, device=dev)
        used = st.used(site)
        with torch.cuda.device(dev):
            _check(lib.ops_dropout_add_layernorm_fwd(T, d, x.data_ptr(), res.data_ptr(), int(res.dtype == torch.bfloat16), gamma.data_ptr(),
                                                     beta.data_ptr(), float(eps), float(p), st.seed + 7919 * site, st.counter.data_ptr(),
                                                     used.data_ptr(), y32.data_ptr(), y16.data_ptr(), z.data_ptr(), mean.data_ptr(),
                                                     rstd.data_ptr(), _stream(dev)), "ops_dropout_add_layernorm_fwd")
        ctx.save_for_backward(z, mean, rstd, gamma, beta)
        ctx.cfg = (float(p), st.seed + 7919 * site, used, st, site, res.dtype)
        ctx.set_materialize_grads(False)        # an unused output's gradient arrives as None, not as a freshly filled zero tensor (one node each)
        return y32, y16

    @staticmethod
    def backward(ctx, g32, g16):
        lib = _cabi.load()
        z, mean, rstd, gamma, beta = ctx.saved_tensors
        p, seed, used, st, site, res_dtype = ctx.cfg
        T, d = z.shape
        dev = z.device
        if g32 is None and g16 is None:
            return None, None, None, None, None, None, None, None
        if g32 is not None:
            g32 = g32.contiguous()
        if g16 is not None:
            g16 = g16.contiguous()
            if g16.dtype != torch.bfloat16:
                g16 = g16.to(torch.bfloat16)
        dx = torch.empty((T, d), dtype=torch.bfloat16, device=dev)
        dres = torch.empty((T, d), dtype=torch.float32, device=dev)
        direct = st.direct and all(t.grad is not None and t.grad.dtype == torch.float32 and t.grad.is_contiguous() for t in (gamma, beta))
        dg = gamma.grad if direct else torch.zeros_like(gamma)        # the launch ADDS (float atomics)
        db = beta.grad if direct else torch.zeros_like(beta)
        with torch.cuda.device(dev):
            _check(lib.ops_dropout_add_layernorm_bwd(T, d, g32.data_ptr() if g32 is not None else None, g16.data_ptr() if g16 is not None else None,
                                                     z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), p, seed, used.data_ptr(),
                                                     dx.data_ptr(), dres.data_ptr(), dg.data_ptr(), db.data_ptr(), _stream(dev)),
                   "ops_dropout_add_layernorm_bwd")
        if res_dtype != torch.float32:
            dres = dres.to(res_dtype)
        return dx, dres, None if direct else dg, None if direct else db, None, None, None, None


class ActDropout(torch.autograd.Function):
    """dropout(LeakyReLU_slope(x)) on bf16 (slope 0: ReLU)."""

    @staticmethod
    def forward(ctx, x, slope, p, st: _State, site: int):
        lib = _cabi.load()
        x = x.contiguous()
        y = torch.empty_like(x)
        used = st.used(site)
        with torch.cuda.device(x.device):
            _check(lib.ops_act_dropout_fwd(x.numel(), x.data_ptr(), y.data_ptr(), float(slope), float(p), st.seed + 7919 * site,
                                           st.counter.data_ptr(), used.data_ptr(), _stream(x.device)), "ops_act_dropout_fwd")
        ctx.save_for_backward(x)
        ctx.cfg = (float(slope), float(p), st.seed + 7919 * site, used)
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _cabi.load()
        (x,) = ctx.saved_tensors
        slope, p, seed, used = ctx.cfg
        g = g.contiguous()
        if g.dtype != torch.bfloat16:
            g = g.to(torch.bfloat16)
        dx = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _check(lib.ops_act_dropout_bwd(x.numel(), x.data_ptr(), g.data_ptr(), dx.data_ptr(), slope, p, seed, used.data_ptr(),
                                           _stream(x.device)), "ops_act_dropout_bwd")
        return dx, None, None, None, None


LAYER_PAIR_FWD = switches.get("tfd_layer_pair") == "1"      # A/B switch: two consecutive layers' forward passes as one launch
_PENDING_LAYER = None       # (argument block, output, tensors to keep alive) of a layer whose forward launch waits for its successor
LAYER_PAIR_BWD = switches.get("tfd_layer_pair_bwd") == "1"  # ... and their backward passes
_PENDING_BWD = None         # (argument block, dx32, tensors to keep alive) of a later layer whose backward launch waits for its predecessor


def flush_pending_backward() -> None:
    """Launch a deferred backward pass whose predecessor never came (a first layer whose inputs need no gradient).  Called by the training
    loop's weight-gradient flush, i.e. behind every backward pass."""
    global _PENDING_BWD
    if _PENDING_BWD is not None:
        pb, pdx, keep = _PENDING_BWD
        _PENDING_BWD = None
        with torch.cuda.device(pdx.device):
            _check(_cabi.load().ops_tfd_encoder_layer_bwd(ctypes.byref(pb), _stream(pdx.device)), "ops_tfd_encoder_layer_bwd")



class EncoderLayerFn(torch.autograd.Function):
    """One whole encoder layer.  Forward: ONE launch (csrc/seq_layer.hip: in-projection, attention, out-projection, dropout + add +
    LayerNorm, feed-forward, dropout + add + LayerNorm; bf16 MFMA products on the shadow weights, everything else as the separate
    launches compute it, same dropout streams).  Backward: the launches of csrc/seq_block.hip and the library's input-gradient products,
    driven from here in the order autograd would run them; weight / bias gradients go where the shadow products send them
    (train.shadow_param_grads), LayerNorm gradients straight into their `.grad` (the caller zeroes the flat buffer every step).
    Inputs (x32, x16): the residual stream and its bf16 copy (the previous block's two outputs; x16's VALUES are bf16(x32)), so that
    the two input gradients travel separately and no add node is needed."""

    @staticmethod
    def forward(ctx, x32, x16, layer, Bn, S, st: _State, li: int, defer: bool = False):
        global _PENDING_LAYER
        lib = _cabi.load()
        mha = layer.self_attn
        H, d = mha.num_heads, mha.embed_dim
        dh, ff = d // H, layer.linear1.out_features
        T = Bn * S
        dev = x32.device
        x32 = x32.contiguous()
        rin, rout, r1, r2 = mha._ops_in_proj.rec, mha._ops_out_proj.rec, layer.linear1._ops_prod.rec, layer.linear2._ops_prod.rec
        tiles = layer._ops_tiles
        bf, f32 = dict(dtype=torch.bfloat16, device=dev), dict(dtype=torch.float32, device=dev)
        qkv, ctxa = torch.empty((T, 3 * d), **bf), torch.empty((T, d), **bf)
        z1, mean1, rstd1, y1_16 = torch.empty((T, d), **f32), torch.empty(T, **f32), torch.empty(T, **f32), torch.empty((T, d), **bf)
        u, h = torch.empty((T, ff), **bf), torch.empty((T, ff), **bf)
        z2, mean2, rstd2 = torch.empty((T, d), **f32), torch.empty(T, **f32), torch.empty(T, **f32)
        y32, y16 = torch.empty((T, d), **f32), torch.empty((T, d), **bf)
        used = st.used(4 * li)
        seeds = [st.seed + 7919 * (4 * li + k) for k in range(4)]
        ps = (float(mha.dropout), float(layer.dropout1.p), float(layer.dropout.p), float(layer.dropout2.p)) if st.train_mode else (0.0, 0.0, 0.0, 0.0)
        a = _cabi.TfdLayerArgs(identity_act=int(IDENTITY_ACT), 
            Bn=Bn, S=S, H=H, dh=dh, d=d, ff=ff, x32=x32.data_ptr(),
            W_in=tiles["in"][0].data_ptr(), b_in=rin.b_sh.data_ptr(), W_out=tiles["out"][0].data_ptr(), b_out=rout.b_sh.data_ptr(),
            W_1=tiles["l1"][0].data_ptr(), b_1=r1.b_sh.data_ptr(), W_2=tiles["l2"][0].data_ptr(), b_2=r2.b_sh.data_ptr(),
            gamma1=layer.norm1.weight.data_ptr(), beta1=layer.norm1.bias.data_ptr(), eps1=float(layer.norm1.eps),
            gamma2=layer.norm2.weight.data_ptr(), beta2=layer.norm2.bias.data_ptr(), eps2=float(layer.norm2.eps),
            p_attn=ps[0], p_1=ps[1], p_act=ps[2], p_2=ps[3], seed_attn=seeds[0], seed_1=seeds[1], seed_act=seeds[2], seed_2=seeds[3],
            counter=st.counter.data_ptr(), used_call=used.data_ptr(), qkv=qkv.data_ptr(), ctx=ctxa.data_ptr(), z1=z1.data_ptr(),
            mean1=mean1.data_ptr(), rstd1=rstd1.data_ptr(), y1_16=y1_16.data_ptr(), u=u.data_ptr(), h=h.data_ptr(), z2=z2.data_ptr(),
            mean2=mean2.data_ptr(), rstd2=rstd2.data_ptr(), y32=y32.data_ptr(), y16=y16.data_ptr())
        with torch.cuda.device(dev):
            ctx.pair_later = False               # this layer is the LATER one of a pair launch (its backward may wait for its predecessor's)
            if defer:                            # the next layer's call launches both (ops_tfd_encoder_layer_pair_fwd)
                assert _PENDING_LAYER is None
                # (everything the deferred launch writes stays alive until it is launched: in a no-grad pass nothing else holds these
                #  tensors, and the caching allocator would hand their memory to the next layer's buffers -- whose launch is this same one)
                _PENDING_LAYER = (a, y32, (x32, qkv, ctxa, z1, mean1, rstd1, y1_16, u, h, z2, mean2, rstd2, y16))
            elif _PENDING_LAYER is not None:
                pa, py, _keep = _PENDING_LAYER
                _PENDING_LAYER = None
                assert py.data_ptr() == x32.data_ptr()
                _check(lib.ops_tfd_encoder_layer_pair_fwd(ctypes.byref(pa), ctypes.byref(a), _stream(dev)), "ops_tfd_encoder_layer_pair_fwd")
                ctx.pair_later = True
            else:
                _check(lib.ops_tfd_encoder_layer_fwd(ctypes.byref(a), _stream(dev)), "ops_tfd_encoder_layer_fwd")
        ctx.save_for_backward(x16, qkv, ctxa, z1, mean1, rstd1, y1_16, u, h, z2, mean2, rstd2)
        ctx.cfg = (layer, Bn, S, H, dh, d, ff, ps, seeds, used, (rin, rout, r1, r2))
        ctx.set_materialize_grads(False)
        return y32, y16

    @staticmethod
    def backward(ctx, g32, g16):
        if g32 is None and g16 is None:
            return (None,) * 8
        from . import train
        lib = _cabi.load()
        x16, qkv, ctxa, z1, mean1, rstd1, y1_16, u, h, z2, mean2, rstd2 = ctx.saved_tensors
        layer, Bn, S, H, dh, d, ff, ps, seeds, used, (rin, rout, r1, r2) = ctx.cfg
        T = Bn * S
        dev = z1.device
        bf, f32 = dict(dtype=torch.bfloat16, device=dev), dict(dtype=torch.float32, device=dev)
        ptr = lambda t: t.data_ptr() if t is not None else None      # noqa: E731
        if g32 is not None:
            g32 = g32.contiguous()
        if g16 is not None:
            g16 = g16.contiguous()
        with torch.cuda.device(dev):
            s = _stream(dev)
            if LAYER_BWD:
                # the whole backward pass: one launch (csrc/seq_layer.hip tfd_layer_bwd_kernel) + the four weight-gradient registrations
                tiles = layer._ops_tiles
                d_f, d_u, d_a, dqkv = torch.empty((T, d), **bf), torch.empty((T, ff), **bf), torch.empty((T, d), **bf), torch.empty((T, 3 * d), **bf)
                dx32 = torch.empty((T, d), **f32)
                a = _cabi.TfdLayerBwdArgs(identity_act=int(IDENTITY_ACT), 
                    Bn=Bn, S=S, H=H, dh=dh, d=d, ff=ff, g32=ptr(g32), g16=ptr(g16),
                    Wt_in=tiles["in"][1].data_ptr(), Wt_out=tiles["out"][1].data_ptr(), Wt_1=tiles["l1"][1].data_ptr(), Wt_2=tiles["l2"][1].data_ptr(),
                    gamma1=layer.norm1.weight.data_ptr(), gamma2=layer.norm2.weight.data_ptr(),
                    p_attn=ps[0], p_1=ps[1], p_act=ps[2], p_2=ps[3], seed_attn=seeds[0], seed_1=seeds[1], seed_act=seeds[2], seed_2=seeds[3],
                    used_call=used.data_ptr(), qkv=qkv.data_ptr(), z1=z1.data_ptr(), mean1=mean1.data_ptr(), rstd1=rstd1.data_ptr(), u=u.data_ptr(),
                    z2=z2.data_ptr(), mean2=mean2.data_ptr(), rstd2=rstd2.data_ptr(), d_f=d_f.data_ptr(), d_u=d_u.data_ptr(), d_a=d_a.data_ptr(),
                    dqkv=dqkv.data_ptr(), dx32=dx32.data_ptr(), dgamma1=layer.norm1.weight.grad.data_ptr(), dbeta1=layer.norm1.bias.grad.data_ptr(),
                    dgamma2=layer.norm2.weight.grad.data_ptr(), dbeta2=layer.norm2.bias.grad.data_ptr())
                nwg = (Bn + (16 // S) - 1) // (16 // S)
                part = None
                if train._WGRAD_QUEUE is not None and LN_PARTIALS:
                    # LayerNorm gamma / beta gradients: per-workgroup column sums, reduced by four jobs of the grouped weight-gradient launch
                    part = torch.empty((nwg, 4, 128), **f32)
                    a.ln_part = part.data_ptr()
                if _TRACE_BWD is not None:
                    tr = torch.zeros(16 * ((Bn + (16 // S) - 1) // (16 // S)), dtype=torch.int64, device=dev)
                    _TRACE_BWD.append(tr)
                    a.trace = tr.data_ptr()
                global _PENDING_BWD
                if _PENDING_BWD is not None:
                    pb, pdx, keep = _PENDING_BWD
                    _PENDING_BWD = None
                    if g32 is not None and g16 is None and pdx.data_ptr() == g32.data_ptr() and _TRACE_BWD is None:
                        _check(lib.ops_tfd_encoder_layer_pair_bwd(ctypes.byref(pb), ctypes.byref(a), s), "ops_tfd_encoder_layer_pair_bwd")
                    else:                        # (not the successor's gradient after all: the two launches)
                        _check(lib.ops_tfd_encoder_layer_bwd(ctypes.byref(pb), s), "ops_tfd_encoder_layer_bwd")
                        _check(lib.ops_tfd_encoder_layer_bwd(ctypes.byref(a), s), "ops_tfd_encoder_layer_bwd")
                elif (LAYER_PAIR_BWD and ctx.pair_later and _TRACE_BWD is None and train._WGRAD_QUEUE is not None      # (queue mode: a flush follows)
                      and all(train.shadow_grads_are_deferred(r, T) for r in (r2, r1, rout, rin))):
                    # (only when the four weight-gradient registrations below merely QUEUE: a product the library runs at once -- fewer rows
                    #  than one MFMA tile, e.g. a one- or two-sample tail batch, or OPS_AMD_SPLIT_WGRAD_ROWS=0 -- would read d_f / d_u / d_a /
                    #  dqkv before the waiting launch has written them)
                    # the predecessor's backward call -- the next node autograd runs: this layer's inputs are its two outputs -- launches both
                    # (the saved tensors too: autograd releases them when this call returns, before the launch that reads them)
                    _PENDING_BWD = (a, dx32, (g32, g16, d_f, d_u, d_a, dqkv, part, x16, qkv, ctxa, z1, mean1, rstd1, y1_16, u, h, z2, mean2, rstd2))
                else:
                    _check(lib.ops_tfd_encoder_layer_bwd(ctypes.byref(a), s), "ops_tfd_encoder_layer_bwd")
                if part is not None:
                    flatp = part.view(nwg, 512)
                    for k, q in enumerate((layer.norm2.weight, layer.norm2.bias, layer.norm1.weight, layer.norm1.bias)):
                        ok = train.queue_column_sums(flatp[:, 128 * k:128 * k + d], q.grad)
                        assert ok
                train.shadow_param_grads(r2, d_f, h)
                train.shadow_param_grads(r1, d_u, y1_16)
                train.shadow_param_grads(rout, d_a, ctxa)
                train.shadow_param_grads(rin, dqkv, x16)
                return dx32, None, None, None, None, None, None, None
            # LayerNorm2 <- (g32, g16)
            d_f, dres2 = torch.empty((T, d), **bf), torch.empty((T, d), **f32)
            _check(lib.ops_dropout_add_layernorm_bwd(T, d, ptr(g32), ptr(g16), z2.data_ptr(), mean2.data_ptr(), rstd2.data_ptr(),
                                                     layer.norm2.weight.data_ptr(), ps[3], seeds[3], used.data_ptr(), d_f.data_ptr(), dres2.data_ptr(),
                                                     layer.norm2.weight.grad.data_ptr(), layer.norm2.bias.grad.data_ptr(), s), "ops_dropout_add_layernorm_bwd")
            train.shadow_param_grads(r2, d_f, h)
            d_h = d_f @ r2.w_sh
            d_u = torch.empty((T, ff), **bf)
            _check(lib.ops_act_dropout_bwd(T * ff, u.data_ptr(), d_h.data_ptr(), d_u.data_ptr(), 0.0, ps[2], seeds[2], used.data_ptr(), s), "ops_act_dropout_bwd")
            train.shadow_param_grads(r1, d_u, y1_16)
            d_y1 = d_u @ r1.w_sh
            # LayerNorm1 <- (dres2, d_y1)
            d_a, dres1 = torch.empty((T, d), **bf), torch.empty((T, d), **f32)
            _check(lib.ops_dropout_add_layernorm_bwd(T, d, dres2.data_ptr(), d_y1.data_ptr(), z1.data_ptr(), mean1.data_ptr(), rstd1.data_ptr(),
                                                     layer.norm1.weight.data_ptr(), ps[1], seeds[1], used.data_ptr(), d_a.data_ptr(), dres1.data_ptr(),
                                                     layer.norm1.weight.grad.data_ptr(), layer.norm1.bias.grad.data_ptr(), s), "ops_dropout_add_layernorm_bwd")
            train.shadow_param_grads(rout, d_a, ctxa)
            d_ctx = d_a @ rout.w_sh
            dqkv = torch.empty((T, 3 * d), **bf)
            _check(lib.ops_seq_attention_bwd(Bn, S, H, dh, qkv.data_ptr(), d_ctx.data_ptr(), dqkv.data_ptr(), ps[0], seeds[0], used.data_ptr(), s),
                   "ops_seq_attention_bwd")
            train.shadow_param_grads(rin, dqkv, x16)
            d_x16 = dqkv @ rin.w_sh
        return dres1, d_x16, None, None, None, None, None, None


def _tile_pair(w: torch.Tensor):
    ru = lambda v, m: (v + m - 1) // m * m      # noqa: E731
    N, K = w.shape
    return (torch.zeros(ru(N, 16), ru(K, 32), dtype=torch.bfloat16, device=w.device), torch.zeros(ru(K, 16), ru(N, 32), dtype=torch.bfloat16, device=w.device))


def _tile_entries(todo):
    ent = (_cabi.MlpRepackEntry * len(todo))()
    for e, (w, (wp, wtp)) in zip(ent, todo):
        e.W, e.N, e.K = w.data_ptr(), w.shape[0], w.shape[1]
        e.Wp, e.ldw, e.Wtp, e.ldwt = wp.data_ptr(), wp.shape[1], wtp.data_ptr(), wtp.shape[1]
    return ent


def enable_layer_tiles(enc: nn.TransformerEncoder, extra=None):
    """Fragment-tiled bf16 copies (plain and transposed: include/openpystruct_amd.h ops_mlp_repack_weights) of the four weight matrices of
    the encoder's layers -- and of `extra` = {name: weight} (the model's head), returned tile pairs in `enc._ops_extra_tiles` -- for the
    one-launch kernels.  Returns the ctypes array of repack entries -- hand it to the optimiser (FlatClipAdam.repack: a launch behind
    its update rebuilds the copies) and call `refresh_layer_tiles` after anything else changed the parameters -- or None when nothing
    qualifies.  One optimiser launch carries OPS_MLP_MAX_REPACK = 16 matrices: the first two or three layers + the extras."""
    todo = []
    good = lambda w: w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() and w.dim() == 2      # noqa: E731
    extra = {k: w for k, w in (extra or {}).items() if good(w)}
    # a second call (patch_model after patch_encoder: now with the head's extras) fits fewer layers into the 16 entries; a layer that
    # kept the tiles of the first call would train on copies nothing refreshes any more -- only layers in THIS entry list carry tiles
    for layer in enc.layers:
        layer.__dict__.pop("_ops_tiles", None)
    for layer in enc.layers:
        if not _layer_ok(layer) or len(todo) + 4 + len(extra) > _cabi.MLP_MAX_REPACK:
            break
        mha = layer.self_attn
        ws = {"in": mha.in_proj_weight, "out": mha.out_proj.weight, "l1": layer.linear1.weight, "l2": layer.linear2.weight}
        if not all(good(w) for w in ws.values()):
            break
        tiles = {k: _tile_pair(w) for k, w in ws.items()}
        todo += [(w, tiles[k]) for k, w in ws.items()]
        layer._ops_tiles = tiles
    xt = {k: _tile_pair(w) for k, w in extra.items()}
    todo += [(w, xt[k]) for k, w in extra.items()]
    enc._ops_extra_tiles = xt
    if not todo:
        return None
    enc._ops_tile_entries = _tile_entries(todo)
    refresh_layer_tiles(enc)
    return enc._ops_tile_entries


def share_step_counter(enc: nn.Module, counter: Optional[torch.Tensor]) -> None:
    """Use the caller's device counter (int64, element 0 advanced once per training step by something the step runs anyway) as the
    dropout / noise streams' call counter instead of an own one advanced by an add node per pass.  None: back to the own counter."""
    if counter is None:
        enc.__dict__.pop("_ops_step_counter", None)
    else:
        enc._ops_step_counter = counter
    if hasattr(enc, "_ops_dropout_state"):
        enc._ops_dropout_state.clear()       # states are rebuilt on the next pass


def refresh_layer_tiles(enc: nn.Module) -> None:
    ent = getattr(enc, "_ops_tile_entries", None)
    if ent is None:
        return
    dev = enc.layers[0].linear1.weight.device
    lib = _cabi.load()
    with torch.cuda.device(dev):
        for i0 in range(0, len(ent), _cabi.MLP_MAX_WGRAD):         # (the stand-alone repack call takes 8 matrices)
            n = min(_cabi.MLP_MAX_WGRAD, len(ent) - i0)
            part = (_cabi.MlpRepackEntry * n)(*[ent[i0 + k] for k in range(n)])
            _check(lib.ops_mlp_repack_weights(n, part, _stream(dev)), "ops_mlp_repack_weights")


def _layer_fused_ok(layer: nn.Module, st: _State) -> bool:
    """The one-launch forward applies: shadow products registered on all four Linear maps, sizes inside the kernel's limits, weights
    16-byte aligned in the shadow buffer, LayerNorm gradients going straight into zeroed `.grad` views."""
    mha = layer.self_attn
    d, H = mha.embed_dim, mha.num_heads
    recs = [getattr(getattr(mha, "_ops_in_proj", None), "rec", None), getattr(getattr(mha, "_ops_out_proj", None), "rec", None),
            getattr(getattr(layer.linear1, "_ops_prod", None), "rec", None), getattr(getattr(layer.linear2, "_ops_prod", None), "rec", None)]
    if not (LAYER_FWD and st.direct and hasattr(layer, "_ops_tiles") and all(r is not None and r.b_sh is not None for r in recs)):
        return False
    ff = layer.linear1.out_features
    if not (d <= 128 and d % 8 == 0 and d // H <= 16 and H <= 8 and 16 <= ff <= 256 and ff % 8 == 0):
        return False
    return all(t.grad is not None and t.grad.dtype == torch.float32 and t.grad.is_contiguous()
               for t in (layer.norm1.weight, layer.norm1.bias, layer.norm2.weight, layer.norm2.bias))


def _layer_ok(layer: nn.Module) -> bool:
    if type(layer) is not nn.TransformerEncoderLayer or layer.norm_first:
        return False
    mha = layer.self_attn
    relu = getattr(layer, "activation_relu_or_gelu", 0) == 1
    d = mha.embed_dim
    return (relu and mha.batch_first and mha._qkv_same_embed_dim and mha.in_proj_bias is not None and mha.bias_k is None and not mha.add_zero_attn
            and d % 8 == 0 and d <= 256 and d % mha.num_heads == 0 and d // mha.num_heads <= 32 and hasattr(mha, "_ops_in_proj") and type(layer.norm1) is nn.LayerNorm
            and type(layer.norm2) is nn.LayerNorm and layer.norm1.elementwise_affine and layer.norm2.elementwise_affine)


def encoder_forward(enc: nn.TransformerEncoder, src: torch.Tensor, st: _State) -> torch.Tensor:
    """The fast path proper: src [B, S, d] float32 -> [B, S, d] float32 (training mode, S <= 8)."""
    global _PENDING_LAYER, _PENDING_BWD
    _PENDING_LAYER = None                  # (a pass that died between a deferred launch and its successor leaves nothing behind)
    _PENDING_BWD = None
    B, S, d = src.shape
    T = B * S
    st.advance()                           # fresh dropout masks for this pass (the launches only read the counter)
    res = src.reshape(T, d)
    x16: Optional[torch.Tensor] = None
    if st.src16 is not None:               # the front end's bf16 copy of src (tfd_fused.model_forward)
        if tuple(st.src16.shape) == (T, d):
            x16 = st.src16
        st.src16 = None
    nl = len(enc.layers)
    for li, layer in enumerate(enc.layers):
        mha = layer.self_attn
        if x16 is not None and _layer_fused_ok(layer, st):
            # the whole layer forward: one launch -- or, for two consecutive layers, one launch for both (the first one's is deferred)
            defer = bool(LAYER_PAIR_FWD and _PENDING_LAYER is None and li + 1 < nl and _layer_fused_ok(enc.layers[li + 1], st) and _TRACE_BWD is None)
            res, x16 = EncoderLayerFn.apply(res, x16, layer, B, S, st, li, defer)
            continue
        qkv = mha._ops_in_proj(res if x16 is None else x16)                                         # [T, 3 d] bf16
        ctx = SeqAttention.apply(qkv, B, S, mha.num_heads, mha.dropout, st, 4 * li)
        a = mha._ops_out_proj(ctx)
        res, x16 = DropoutAddLayerNorm.apply(a, res, layer.norm1.weight, layer.norm1.bias, layer.norm1.eps, layer.dropout1.p, st, 4 * li + 1)
        h = ActDropout.apply(layer.linear1(x16), 0.0, layer.dropout.p, st, 4 * li + 2)
        f = layer.linear2(h)
        res, x16 = DropoutAddLayerNorm.apply(f, res, layer.norm2.weight, layer.norm2.bias, layer.norm2.eps, layer.dropout2.p, st, 4 * li + 3)
    out = res.reshape(B, S, d)
    st.last16 = x16
    return enc.norm(out) if enc.norm is not None else out


class DiffusionCombine(torch.autograd.Function):
    """(z, z16) with z [B, 1 + Nc, d] = [cls | (x_noisy - sb m) / sa] + pe: the tail of DiffusionModule.forward (TFD:474-478), the [CLS]
    concatenation and the positional encoding (TFD:563-567) in one launch each way; z16 is the bfloat16 copy the first in-projection
    multiplies (no cast node), and the backward launch adds the two gradients itself.  m [B * Nc, d] bf16: the MLP's output."""

    @staticmethod
    def forward(ctx, m, xn32, sa, sb, cls, pe, B, Nc, st: _State):
        lib = _cabi.load()
        m = m.contiguous()
        d = m.shape[1]
        z = torch.empty((B, Nc + 1, d), dtype=torch.float32, device=m.device)
        z16 = torch.empty((B * (Nc + 1), d), dtype=torch.bfloat16, device=m.device)
        with torch.cuda.device(m.device):
            _check(lib.ops_diffusion_combine_fwd(B, Nc, d, m.data_ptr(), xn32.data_ptr(), sa.data_ptr(), sb.data_ptr(), cls.data_ptr(),
                                                 pe.data_ptr(), z.data_ptr(), z16.data_ptr(), _stream(m.device)), "ops_diffusion_combine_fwd")
        ctx.save_for_backward(sa, sb, cls)
        ctx.cfg = (B, Nc, d, st)
        ctx.set_materialize_grads(False)
        return z, z16

    @staticmethod
    def backward(ctx, g, g16):
        lib = _cabi.load()
        sa, sb, cls = ctx.saved_tensors
        B, Nc, d, st = ctx.cfg
        if g is None and g16 is None:
            return (None,) * 9
        if g is not None:
            g = g.contiguous().float()
        if g16 is not None:
            g16 = g16.contiguous()
            if g16.dtype != torch.bfloat16:
                g16 = g16.to(torch.bfloat16)
        dev = (g if g is not None else g16).device
        dm = torch.empty((B * Nc, d), dtype=torch.bfloat16, device=dev)
        direct = st.direct and cls.grad is not None and cls.grad.dtype == torch.float32 and cls.grad.is_contiguous()
        dcls = cls.grad if direct else torch.zeros_like(cls)
        with torch.cuda.device(dev):
            _check(lib.ops_diffusion_combine_bwd(B, Nc, d, g.data_ptr() if g is not None else None, g16.data_ptr() if g16 is not None else None,
                                                 sa.data_ptr(), sb.data_ptr(), dm.data_ptr(), dcls.data_ptr(), _stream(dev)),
                   "ops_diffusion_combine_bwd")
        return dm, None, None, None, None if direct else dcls, None, None, None, None


class ClsRows(torch.autograd.Function):
    """rows 0, S, 2 S, ... of x16 [B * S, d] as a strided [B, d] view (the head's first product reads it in place: row stride S d);
    backward: the gradient rows into a persistent zero tensor -- no fill, no cast."""

    @staticmethod
    def forward(ctx, x16, B, S, st: _State):
        ctx.cfg = (B, S, st)
        return x16.view(B, S, x16.shape[1])[:, 0, :]

    @staticmethod
    def backward(ctx, g):
        B, S, st = ctx.cfg
        d = g.shape[1]
        full = st.zeros16((B * S, d))
        rows = full.view(B, S, d)[:, 0, :]
        if not (g.data_ptr() == rows.data_ptr() and g.stride() == rows.stride() and g.dtype == rows.dtype):   # (the product wrote them there itself)
            rows.copy_(g)
        return full, None, None, None


class FrontFn(torch.autograd.Function):
    """The diffusion front end (TFD:443-478, :563-567) as one launch per direction (csrc/seq_layer.hip tfd_front_*_kernel): draws,
    x_noisy, the two-layer MLP on the tiled weights, the combine with the [CLS] token and the positional encoding.  x [B, Nc, d]
    float32 -> (z [B, 1 + Nc, d] float32, z16 [B (1 + Nc), d] bfloat16).  Backward: one launch; the MLP's weight / bias gradients where
    the shadow products send them, the [CLS] token's gradient by float atomics."""

    @staticmethod
    def forward(ctx, x, anchor, model, st: _State):
        # `anchor` (the [CLS] token Parameter) is not read through autograd: it makes the engine record the node although x needs no
        # gradient; its gradient slot stays None -- dcls goes into cls_token.grad by the launch's atomics
        lib = _cabi.load()
        dm = model.diffusion
        B, Nc, d = x.shape
        hid = dm.mlp[0].out_features
        r0, r2 = dm.mlp[0]._ops_prod.rec, dm.mlp[2]._ops_prod.rec
        tiles = model.transformer_encoder._ops_extra_tiles
        dev = x.device
        x = x.contiguous()
        rows = B * Nc
        bf, f32 = dict(dtype=torch.bfloat16, device=dev), dict(dtype=torch.float32, device=dev)
        xn16, h = torch.empty((rows, d), **bf), torch.empty((rows, hid), **bf)
        sa, sb = torch.empty(rows, **f32), torch.empty(rows, **f32)
        z, z16 = torch.empty((B, Nc + 1, d), **f32), torch.empty((B * (Nc + 1), d), **bf)
        keep = st.keep_draws or KEEP_DRAWS
        t = torch.empty(rows, dtype=torch.int64, device=dev) if keep else None
        eps = torch.empty((rows, d), **f32) if keep else None
        st.gathered = False
        gs = _GATHER if (_GATHER is not None and _GATHER["owner"] is model and st.train_mode and st.external and st.counter.numel() >= 2) else None
        if gs is not None and not (gs["src"].shape[1] == Nc and gs["src"].shape[2] == d and gs["src"].device == dev):
            gs = None
        a = _cabi.TfdFrontArgs(identity_act=int(IDENTITY_ACT), B=B, Nc=Nc, d=d, hid=hid, T=int(dm.T), x=x.data_ptr(), alpha_cumprod=dm._acp.data_ptr(), seed=st.seed + 7919 * 100,
                               counter=st.counter.data_ptr(), W0=tiles["mlp0"][0].data_ptr(), b0=r0.b_sh.data_ptr(), W2=tiles["mlp2"][0].data_ptr(),
                               b2=r2.b_sh.data_ptr(), cls=model.cls_token.data_ptr(), pe=model.pos_encoder.pe.data_ptr(), xn16=xn16.data_ptr(),
                               h=h.data_ptr(), sa=sa.data_ptr(), sb=sb.data_ptr(), z=z.data_ptr(), z16=z16.data_ptr(),
                               t_out=t.data_ptr() if keep else None, eps_out=eps.data_ptr() if keep else None)
        if gs is not None:      # the launch assembles its own batch: rows order[cursor .. cursor + B) of the training set + the input noise
            a.src, a.order, a.cursor, a.idx_out = gs["src"].data_ptr(), gs["order"].data_ptr(), gs["cursor"].data_ptr(), gs["idx_out"].data_ptr()
            a.sigma, a.in_seed, a.n_order = gs["sigma"].data_ptr(), gs["seed"], int(gs["order"].numel())
            st.gathered = True
        with torch.cuda.device(dev):
            _check(lib.ops_tfd_front_fwd(ctypes.byref(a), _stream(dev)), "ops_tfd_front_fwd")
        if keep:
            st.draws = (t.view(B, Nc), eps.view(B, Nc, d))
        ctx.save_for_backward(xn16, h, sa, sb)
        ctx.cfg = (model, B, Nc, d, hid, st, (r0, r2))
        ctx.set_materialize_grads(False)
        return z, z16

    @staticmethod
    def backward(ctx, g, g16):
        if g is None and g16 is None:
            return None, None, None, None
        from . import train
        lib = _cabi.load()
        xn16, h, sa, sb = ctx.saved_tensors
        model, B, Nc, d, hid, st, (r0, r2) = ctx.cfg
        tiles = model.transformer_encoder._ops_extra_tiles
        dev = xn16.device
        if g is not None:
            g = g.contiguous().float()
        if g16 is not None:
            g16 = g16.contiguous()
            if g16.dtype != torch.bfloat16:
                g16 = g16.to(torch.bfloat16)
        rows = B * Nc
        dm_, d_h = torch.empty((rows, d), dtype=torch.bfloat16, device=dev), torch.empty((rows, hid), dtype=torch.bfloat16, device=dev)
        cls = model.cls_token
        a = _cabi.TfdFrontBwdArgs(B=B, Nc=Nc, d=d, hid=hid, g32=g.data_ptr() if g is not None else None, g16=g16.data_ptr() if g16 is not None else None,
                                  sa=sa.data_ptr(), sb=sb.data_ptr(), h=h.data_ptr(), Wt2=tiles["mlp2"][1].data_ptr(), dm=dm_.data_ptr(), d_h=d_h.data_ptr(),
                                  dcls=cls.grad.data_ptr())
        with torch.cuda.device(dev):
            _check(lib.ops_tfd_front_bwd(ctypes.byref(a), _stream(dev)), "ops_tfd_front_bwd")
        train.shadow_param_grads(r2, dm_, h)
        train.shadow_param_grads(r0, d_h, xn16)
        return None, None, None, None


def _front_fused_ok(model: nn.Module, st: _State, d: int) -> bool:
    mlp = model.diffusion.mlp
    tiles = getattr(model.transformer_encoder, "_ops_extra_tiles", None) or {}
    if not (FRONT and DRAW and st.direct and "mlp0" in tiles and "mlp2" in tiles and type(mlp[0]) is nn.Linear and type(mlp[2]) is nn.Linear):
        return False
    recs = [getattr(getattr(m, "_ops_prod", None), "rec", None) for m in (mlp[0], mlp[2])]
    hid = mlp[0].out_features
    cls = model.cls_token
    return (all(r is not None and r.b_sh is not None for r in recs) and d % 8 == 0 and d <= 128 and 16 <= hid <= 256 and hid % 8 == 0
            and mlp[0].in_features == d and mlp[2].out_features == d and cls.dtype == torch.float32 and cls.is_contiguous() and cls.numel() == d
            and cls.grad is not None and cls.grad.dtype == torch.float32 and cls.grad.is_contiguous()
            and model.pos_encoder.pe.dtype == torch.float32 and model.pos_encoder.pe.is_contiguous())


class HeadFn(torch.autograd.Function):
    """The model's head on the [CLS] rows -- fc1 -> LayerNorm -> ReLU -> dropout -> fc2 (TFD:568-575) -- as one launch per direction
    (csrc/seq_layer.hip tfd_head_*_kernel; was four + four).  x16 [B S, d] bf16: the last encoder layer's output; returns [B, C] bf16.
    Backward: the gradient of the [CLS] rows goes into a persistent zero tensor of x16's shape (every other row stays zero), weight /
    bias gradients where the shadow products send them, LayerNorm gradients straight into their `.grad`."""

    @staticmethod
    def forward(ctx, x16, model, B, S, st: _State, loss_spec=None):
        """`loss_spec` = (targets [B, C] float32, TrainableL1L2Loss, alpha0 or None, running-sum tensor or None): the training loss is
        computed ON the output tile by the same launch (value finished by the backward launch) and returned beside the predictions;
        the backward pass then starts from the launch's own d loss / d out (the caller differentiates the loss with weight one)."""
        lib = _cabi.load()
        r1, r2 = model.fc1._ops_prod.rec, model.fc2._ops_prod.rec
        tiles = model.transformer_encoder._ops_extra_tiles
        d, hid, C = x16.shape[1], model.fc1.out_features, model.fc2.out_features
        dev = x16.device
        x16 = x16.contiguous()
        bf, f32 = dict(dtype=torch.bfloat16, device=dev), dict(dtype=torch.float32, device=dev)
        a16, h, out = torch.empty((B, hid), **bf), torch.empty((B, hid), **bf), torch.empty((B, C), **bf)
        mean, rstd = torch.empty(B, **f32), torch.empty(B, **f32)
        used = st.used(103)
        p = float(model.dropout.p) if st.train_mode else 0.0
        a = _cabi.TfdHeadArgs(identity_act=int(IDENTITY_ACT), B=B, S=S, d=d, hid=hid, C=C, y16=x16.data_ptr(), W1=tiles["fc1"][0].data_ptr(), b1=r1.b_sh.data_ptr(),
                              gamma=model.norm1.weight.data_ptr(), beta=model.norm1.bias.data_ptr(), eps=float(model.norm1.eps),
                              W2=tiles["fc2"][0].data_ptr(), b2=r2.b_sh.data_ptr(), p_drop=p, seed=st.seed + 7919 * 103,
                              counter=st.counter.data_ptr(), used_call=used.data_ptr(), a16=a16.data_ptr(), mean=mean.data_ptr(),
                              rstd=rstd.data_ptr(), h=h.data_ptr(), out=out.data_ptr())
        ctx.loss = None
        if loss_spec is not None:
            targets, crit, alpha0, acc = loss_spec
            sc = lambda v: None if v is None else (v if torch.is_tensor(v) else torch.tensor(float(v))).to(device=dev, dtype=torch.float32).reshape(())   # noqa: E731
            grad, part = torch.empty((B, C), **bf), torch.empty(5 * ((B + 15) // 16), dtype=torch.float64, device=dev)
            loss = torch.empty((), **f32)
            alpha, minc, maxc = crit.alpha.detach(), sc(crit.min_constraint), sc(crit.max_constraint)
            a.targets, a.grad, a.loss_part, a.alpha = targets.data_ptr(), grad.data_ptr(), part.data_ptr(), alpha.data_ptr()
            if getattr(st, "gathered", False) and _GATHER is not None:      # the front end assembled this batch: targets by its row indices
                a.targets, a.target_rows = _GATHER["targets"].data_ptr(), _GATHER["idx_out"].data_ptr()
            a.min_constraint = minc.data_ptr() if minc is not None else None
            a.max_constraint = maxc.data_ptr() if maxc is not None else None
            a.box_weight = float(crit.penalty_weight)
            ctx.loss = (grad, part, loss, alpha, float("nan") if alpha0 is None else float(alpha0), float(crit.penalty_weight), acc, targets, minc, maxc)
        with torch.cuda.device(dev):
            _check(lib.ops_tfd_head_fwd(ctypes.byref(a), _stream(dev)), "ops_tfd_head_fwd")
        ctx.save_for_backward(x16, a16, mean, rstd, h)
        ctx.cfg = (model, B, S, d, hid, C, p, st, (r1, r2))
        if ctx.loss is not None:
            # (the predictions stay differentiable: a physics term may hang on them; without one their gradient is absent -- and with
            #  materialisation off no zero tensor, i.e. no fill node per step, is made for it)
            ctx.set_materialize_grads(False)
            return out, ctx.loss[2]
        return out

    @staticmethod
    def backward(ctx, g, g_loss=None):
        from . import train
        lib = _cabi.load()
        x16, a16, mean, rstd, h = ctx.saved_tensors
        model, B, S, d, hid, C, p, st, (r1, r2) = ctx.cfg
        tiles = model.transformer_encoder._ops_extra_tiles
        dev = x16.device
        if ctx.loss is not None:
            # the forward launch's own d loss / d out (the loss enters the total with weight one) + whatever else hangs on the predictions
            g2 = None
            if g is not None:
                g2 = g.contiguous() if g.dtype == torch.bfloat16 else g.to(torch.bfloat16)
            g = ctx.loss[0]                     # (g2: summed into it by the launch below, in place)
        else:
            g2 = None
            g = g.contiguous()
            if g.dtype != torch.bfloat16:
                g = g.to(torch.bfloat16)
        d_a = torch.empty((B, hid), dtype=torch.bfloat16, device=dev)
        full = st.zeros16((B * S, d))
        a = _cabi.TfdHeadBwdArgs(B=B, S=S, d=d, hid=hid, C=C, g=g.data_ptr(), Wt2=tiles["fc2"][1].data_ptr(), Wt1=tiles["fc1"][1].data_ptr(),
                                 gamma=model.norm1.weight.data_ptr(), p_drop=p, a16=a16.data_ptr(), mean=mean.data_ptr(), rstd=rstd.data_ptr(),
                                 h=h.data_ptr(), d_a=d_a.data_ptr(), dcls_rows=full.data_ptr(), dgamma=model.norm1.weight.grad.data_ptr(),
                                 dbeta=model.norm1.bias.grad.data_ptr())
        if ctx.loss is not None:                # workgroup 0 of this launch adds the forward launch's partial sums up
            _, part, loss, alpha, alpha0, bw, acc = ctx.loss[:7]
            a.loss_part, a.alpha, a.alpha0, a.box_weight, a.loss = part.data_ptr(), alpha.data_ptr(), alpha0, bw, loss.data_ptr()
            a.loss_sum = acc.data_ptr() if acc is not None else None
        if g2 is not None:
            a.g2, a.g_sum = g2.data_ptr(), g.data_ptr()
        with torch.cuda.device(dev):
            _check(lib.ops_tfd_head_bwd(ctypes.byref(a), _stream(dev)), "ops_tfd_head_bwd")
        train.shadow_param_grads(r2, g, h)
        train.shadow_param_grads(r1, d_a, x16.view(B, S, d)[:, 0, :])       # (row-strided operand: no copy of the [CLS] rows)
        return full, None, None, None, None, None


_GATHER = None            # dict(src, order, cursor, idx_out, sigma, seed): armed by the training loop for a whole run (arm_gather)


def arm_gather(model: nn.Module, src: torch.Tensor, targets: torch.Tensor, order: torch.Tensor, cursor: torch.Tensor, idx_out: torch.Tensor,
               sigma: torch.Tensor, seed: int) -> None:
    """From now on the fused front end of THIS patched model in training mode assembles its own batch (any other model's passes are untouched:
    a run that died before `disarm_gather` must not feed its tensors to the next one): sample b = row order[cursor + b] of
    `src` [n, Nc, d] float32 + sigma * N(0, 1) from the batch-assembly stream (`seed`, the shared step counter), the tensor handed to
    `model(x)` only gives the batch size; idx_out [>= B] receives the rows, through which the head's loss-on-the-tile reads rows of
    `targets` [n, C] float32 (the step must arm it: arm_head_loss); the launch advances the step counter and `cursor` itself.
    `disarm_gather()` ends it."""
    global _GATHER
    assert src.dtype == torch.float32 and src.is_contiguous() and src.dim() == 3 and order.dtype == torch.int64 and cursor.dtype == torch.int64
    assert targets.dtype == torch.float32 and targets.is_contiguous() and targets.dim() == 2 and targets.shape[0] == src.shape[0]
    _GATHER = dict(owner=model, src=src, targets=targets, order=order, cursor=cursor, idx_out=idx_out, sigma=sigma, seed=int(seed) & 0x7FFFFFFFFFFFFFFF)


def disarm_gather() -> None:
    global _GATHER
    _GATHER = None


def gather_fusable(model: nn.Module, device, d: int) -> bool:
    """Whether a patched model's training forward will run the fused front end AND the head with the loss on its tile (the two launches that
    take over the batch assembly's work)."""
    enc = getattr(model, "transformer_encoder", None)
    if enc is None or "forward" not in model.__dict__ or getattr(enc, "_ops_step_counter", None) is None:
        return False
    st = _State(torch.device(device), 1, True, enc._ops_step_counter)
    return bool(FRONT_GATHER and HEAD_LOSS and st.external and st.counter.numel() >= 2 and _front_fused_ok(model, st, d) and _head_fused_ok(model, st, d)
                and enc.norm is None and all(_layer_fused_ok(l, st) for l in enc.layers))


_PENDING_LOSS = None      # (targets, criterion, alpha0, running sum): armed by the training step for ITS next forward pass
_TAKEN_LOSS = None


def arm_head_loss(targets: torch.Tensor, crit: nn.Module, alpha0, acc) -> bool:
    """The training step's next forward pass of a patched model may compute its TrainableL1L2Loss on the head's output tile (two launches
    of the step less); `take_head_loss()` afterwards tells whether it did.  False: not this criterion / these targets."""
    global _PENDING_LOSS, _TAKEN_LOSS
    from .surrogates import TrainableL1L2Loss
    _TAKEN_LOSS = None
    ok = (HEAD_LOSS and type(crit) is TrainableL1L2Loss and torch.is_tensor(targets) and targets.is_cuda and targets.dtype == torch.float32
          and targets.dim() == 2 and targets.is_contiguous())
    _PENDING_LOSS = (targets, crit, alpha0, acc) if ok else None
    return ok


def take_head_loss():
    """The loss the last forward pass computed on the head's tile (an autograd scalar), or None."""
    global _PENDING_LOSS, _TAKEN_LOSS
    loss, _TAKEN_LOSS, _PENDING_LOSS = _TAKEN_LOSS, None, None
    return loss


def _head_fused_ok(model: nn.Module, st: _State, d: int) -> bool:
    enc = model.transformer_encoder
    if type(model.fc1) is not nn.Linear or type(model.fc2) is not nn.Linear:
        return False
    tiles = getattr(enc, "_ops_extra_tiles", None) or {}
    recs = [getattr(getattr(m, "_ops_prod", None), "rec", None) for m in (model.fc1, model.fc2)]
    hid, C = model.fc1.out_features, model.fc2.out_features
    return (HEAD and st.direct and "fc1" in tiles and "fc2" in tiles and all(r is not None and r.b_sh is not None for r in recs) and d % 8 == 0
            and d <= 128 and 16 <= hid <= 256 and hid % 8 == 0 and 4 <= C <= 128 and C % 4 == 0 and model.fc1.in_features == d
            and all(t.grad is not None and t.grad.dtype == torch.float32 and t.grad.is_contiguous() for t in (model.norm1.weight, model.norm1.bias)))


def model_forward(model: nn.Module, x: torch.Tensor, st: _State) -> torch.Tensor:
    """ModelOnePassTransformerWithDiffusion.forward (TFD:539-575) for the training step: diffusion arithmetic in two launches around
    the MLP's shadow products, the patched encoder, the head's LayerNorm / ReLU / dropout in two.  The random step indices and the
    noise are drawn inside the first launch (DRAW; or by the framework's generators in the module's order: torch.randint, torch.randn_like).  `st` is the
    ENCODER's dropout state: its pass below advances the counter before the head's dropout (the only site out here with p > 0,
    TFD:573) draws its mask."""
    lib = _cabi.load()
    B, Nc, d = x.shape
    dm = model.diffusion
    x = x.contiguous()
    if _front_fused_ok(model, st, d):
        z, z16 = FrontFn.apply(x, model.cls_token, model, st)                                      # draws, x_noisy, MLP, combine: one launch
        return _model_tail(model, z, z16, st, B, Nc, d)
    rows = B * Nc
    xn32 = torch.empty((rows, d), dtype=torch.float32, device=x.device)
    xn16 = torch.empty((rows, d), dtype=torch.bfloat16, device=x.device)
    sa = torch.empty(rows, dtype=torch.float32, device=x.device)
    sb = torch.empty(rows, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        if DRAW:       # step indices and noise drawn inside the launch (the encoder pass below advances the counter: fresh draws every step)
            keep = st.keep_draws or KEEP_DRAWS
            t = torch.empty(rows, dtype=torch.int64, device=x.device) if keep else None
            eps = torch.empty((rows, d), dtype=torch.float32, device=x.device) if keep else None
            _check(lib.ops_diffusion_noise_draw(rows, d, int(dm.T), x.data_ptr(), dm._acp.data_ptr(), st.seed + 7919 * 100, st.counter.data_ptr(),
                                                xn32.data_ptr(), xn16.data_ptr(), sa.data_ptr(), sb.data_ptr(), t.data_ptr() if keep else None,
                                                eps.data_ptr() if keep else None, _stream(x.device)), "ops_diffusion_noise_draw")
            if keep:
                st.draws = (t.view(B, Nc), eps.view(B, Nc, d))
        else:
            t = torch.randint(0, dm.T, (B, Nc), device=x.device)
            eps = torch.randn_like(x)
            _check(lib.ops_diffusion_noise(rows, d, x.data_ptr(), t.data_ptr(), eps.data_ptr(), dm._acp.data_ptr(), xn32.data_ptr(), xn16.data_ptr(),
                                           sa.data_ptr(), sb.data_ptr(), _stream(x.device)), "ops_diffusion_noise")
    h = ActDropout.apply(dm.mlp[0](xn16), 0.0, 0.0, st, 101)                      # ReLU
    m = dm.mlp[2](h)
    z, z16 = DiffusionCombine.apply(m, xn32, sa, sb, model.cls_token, model.pos_encoder.pe, B, Nc, st)
    return _model_tail(model, z, z16, st, B, Nc, d)


def _model_tail(model: nn.Module, z: torch.Tensor, z16: torch.Tensor, st: _State, B: int, Nc: int, d: int) -> torch.Tensor:
    """Encoder + head behind the front end."""
    st.src16, st.last16 = z16, None                                                # handed to / by the patched encoder pass below
    z = model.transformer_encoder(z)
    if st.last16 is not None and model.transformer_encoder.norm is None and _head_fused_ok(model, st, d):
        last16, st.last16 = st.last16, None
        global _PENDING_LOSS, _TAKEN_LOSS
        spec, _PENDING_LOSS = _PENDING_LOSS, None
        if (spec is not None and st.train_mode and torch.is_grad_enabled() and tuple(spec[0].shape) == (B, model.fc2.out_features)
                and spec[0].device == last16.device):
            out, _TAKEN_LOSS = HeadFn.apply(last16, model, B, Nc + 1, st, spec)   # ... and the loss on its output tile
            return out
        return HeadFn.apply(last16, model, B, Nc + 1, st, None)                      # the whole head: one launch
    if st.last16 is not None and model.transformer_encoder.norm is None:
        from . import train
        train.set_next_input_grad_dest(st.zeros16((B * (Nc + 1), d)).view(B, Nc + 1, d)[:, 0, :])   # fc1's input gradient lands in the [CLS] rows
        a = model.fc1(ClsRows.apply(st.last16, B, Nc + 1, st))                     # bf16 [B, hidden], read in place (no slice copy / cast)
        train.set_next_input_grad_dest(None)
        st.last16 = None
    else:
        a = model.fc1(z[:, 0, :])
    zero = st.zeros(a.shape)
    _, y16 = DropoutAddLayerNorm.apply(a, zero, model.norm1.weight, model.norm1.bias, model.norm1.eps, 0.0, st, 102)
    return model.fc2(ActDropout.apply(y16, 0.0, model.dropout.p, st, 103))


def patch_model(model: nn.Module, seed: int, direct_param_grads: bool = False) -> bool:
    """Encoder fast path + the fused front end / head around it for ModelOnePassTransformerWithDiffusion in training mode under bf16
    autocast on the GPU; everything else (evaluation, CPU) runs the module's own forward.  Returns whether anything was patched."""
    from .surrogates import ModelOnePassTransformerWithDiffusion
    if not (ENABLED and isinstance(model, ModelOnePassTransformerWithDiffusion)):
        return False
    if not patch_encoder(model.transformer_encoder, seed, direct_param_grads):
        return False
    mlp = model.diffusion.mlp
    ok = (len(mlp) == 3 and type(mlp[0]) is nn.Linear and type(mlp[1]) is nn.ReLU and type(mlp[2]) is nn.Linear and "forward" in mlp[0].__dict__
          and "forward" in model.fc1.__dict__ and type(model.norm1) is nn.LayerNorm and model.norm1.elementwise_affine
          and model.fc1.out_features <= 256 and model.pos_encoder.pe.shape[-1] == model.feat_dim)
    if not ok:
        return True                       # the encoder alone
    if LAYER_FWD and HEAD and "forward" in model.fc2.__dict__:
        # the head's two weight matrices join the layers' tiled copies (one-launch head: HeadFn)
        enable_layer_tiles(model.transformer_encoder, extra={"fc1": model.fc1.weight, "fc2": model.fc2.weight, "mlp0": mlp[0].weight, "mlp2": mlp[2].weight})
    # ONE dropout stream for the model and its encoder: the encoder pass advances the call counter once per step (one captured
    # add_), and the head's dropout (site 103, drawn after the encoder) reads the same counter -- a state of its own that nothing
    # advanced gave it the same mask in every step of a run
    state = model.transformer_encoder._ops_dropout_state
    cls = type(model)

    def forward(self, x):
        fast = ((self.training or (EVAL_FAST and not torch.is_grad_enabled())) and x.is_cuda and x.dim() == 3 and x.dtype == torch.float32
                and x.shape[1] == self.n_cases and x.shape[1] < 8 and torch.is_autocast_enabled("cuda")
                and torch.get_autocast_dtype("cuda") == torch.bfloat16)
        if not fast:
            return cls.forward(self, x)
        st = state.get(x.device)
        if st is None:
            st = state[x.device] = _State(x.device, seed, direct_param_grads, getattr(self.transformer_encoder, "_ops_step_counter", None))
        if not self.training:
            # evaluation (no gradients): the same launches with every dropout probability 0 (the diffusion noise is NOT gated on the
            # mode, TFD:443-478) -- only when every block has its one-launch form; ~70 us per batch against ~500 through the modules
            enc = self.transformer_encoder
            if not (_front_fused_ok(self, st, x.shape[2]) and _head_fused_ok(self, st, x.shape[2]) and enc.norm is None
                    and all(_layer_fused_ok(l, st) for l in enc.layers)):
                return cls.forward(self, x)
            st.train_mode = False
            try:
                return model_forward(self, x, st)
            finally:
                st.train_mode = True
        return model_forward(self, x, st)

    model.forward = types.MethodType(forward, model)
    return True


def unpatch_model(model: nn.Module) -> None:
    if "forward" in model.__dict__:
        del model.__dict__["forward"]
    if hasattr(model, "transformer_encoder"):
        unpatch_encoder(model.transformer_encoder)


def patch_encoder(enc: nn.TransformerEncoder, seed: int, direct_param_grads: bool = False) -> bool:
    """Route `enc.forward` through the fast path whenever it applies (GPU, training, bf16 autocast, no masks, short sequences);
    anything else falls through to the framework's forward.  Needs the shadow products registered on every layer's attention
    (train.enable_shadow_linears).  `direct_param_grads`: the caller zeroes every parameter's `.grad` before each backward pass (the
    training loop's flat buffer), so LayerNorm gradients may be assigned there.  Returns whether the encoder qualifies at all."""
    if not (ENABLED and type(enc) is nn.TransformerEncoder and all(_layer_ok(l) for l in enc.layers)):
        return False
    state = {}
    enc._ops_dropout_state = state          # device -> _State; patch_model's front end / head draw from the same stream

    def forward(self, src, mask=None, src_key_padding_mask=None, is_causal=None):
        st0 = state.get(src.device)
        eval_pass = st0 is not None and not st0.train_mode and not self.training and not torch.is_grad_enabled()    # (set by the model's patched forward)
        fast = ((self.training or eval_pass) and src.is_cuda and src.dim() == 3 and src.shape[1] <= 8 and mask is None and src_key_padding_mask is None
                and not is_causal and torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16
                and src.dtype == torch.float32)
        if not fast:
            return nn.TransformerEncoder.forward(self, src, mask=mask, src_key_padding_mask=src_key_padding_mask, is_causal=is_causal)
        st = state.get(src.device)
        if st is None:
            st = state[src.device] = _State(src.device, seed, direct_param_grads, getattr(self, "_ops_step_counter", None))
        return encoder_forward(self, src, st)

    enc.forward = types.MethodType(forward, enc)
    if LAYER_FWD:
        enable_layer_tiles(enc)              # weight tiles of the one-launch layer kernels (enc._ops_tile_entries: FlatClipAdam.repack)
    return True


def unpatch_encoder(enc: nn.Module) -> None:
    if "forward" in enc.__dict__:
        del enc.__dict__["forward"]
    enc.__dict__.pop("_ops_dropout_state", None)
    enc.__dict__.pop("_ops_tile_entries", None)
    enc.__dict__.pop("_ops_extra_tiles", None)
    enc.__dict__.pop("_ops_step_counter", None)
    for layer in getattr(enc, "layers", []):
        layer.__dict__.pop("_ops_tiles", None)
