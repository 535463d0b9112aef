"""Dataset -> model tensors for the surrogate training loops (PINN and Transformer-Diffusion).

Tensor-native restatement of the data-prep block the reference copy-pastes into every model script
(/root/reference/OpenPyStruct_PINN_MultiCase.py:66-120, :198-388;
 /root/reference/OpenPyStruct_TransformerDiffusionModule_MultiCase.py:72-189, :240-371):
pad -> group `n_cases` consecutive samples -> permutation split -> per-column standardisation ->
`mean + c * std` label aggregation over the cases of a group -> standardised targets.

Differences that are deliberate: everything stays in torch tensors on the device the
records live on (no numpy / sklearn round trip, no per-batch host-to-device copies later), the split
permutation is seeded, and the standardisation moments can be all-reduced over a process group so that
every rank of a data-parallel job scales with the GLOBAL statistics although it only holds its shard.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Sequence

import torch
import torch.distributed as dist


def pad_sequences(seqs: Sequence[Sequence[float]], max_length: int, pad_val: float = 0.0) -> torch.Tensor:
    """Ragged lists -> [len(seqs), max_length] float32 (PINN:66-76)."""
    out = torch.full((len(seqs), max_length), pad_val, dtype=torch.float32)
    for i, s in enumerate(seqs):
        t = torch.as_tensor(s, dtype=torch.float32)[:max_length]
        out[i, : t.numel()] = t
    return out


def unify_label_with_c(y3: torch.Tensor, c: float = 0.5) -> torch.Tensor:
    """[G, n_cases, M] -> [G, M]: mean over the cases + c * population std (PINN:79-92)."""
    return y3.mean(dim=1) + c * y3.std(dim=1, unbiased=False)


class StandardScalerT:
    """Column-wise (x - mean) / std with sklearn's conventions (population variance, zero-variance columns
    are left unscaled).  `fit` optionally all-reduces count / sum / sum of squares over `group`."""

    def __init__(self):
        self.mean_: Optional[torch.Tensor] = None
        self.scale_: Optional[torch.Tensor] = None

    def fit(self, x2: torch.Tensor, group=None, distributed: bool = False) -> "StandardScalerT":
        x = x2.double()
        stats = torch.cat([torch.tensor([float(x.shape[0])], dtype=torch.float64, device=x.device), x.sum(0), (x * x).sum(0)])
        if distributed and dist.is_available() and dist.is_initialized():
            dist.all_reduce(stats, group=group)
        n, s, ss = stats[0], stats[1: 1 + x.shape[1]], stats[1 + x.shape[1]:]
        mean = s / n
        var = (ss / n - mean * mean).clamp_min(0.0)
        scale = var.sqrt()
        # sklearn's _handle_zeros_in_scale: (near-)constant columns keep scale 1
        scale = torch.where(scale < 10 * torch.finfo(torch.float64).eps * mean.abs().clamp_min(1.0), torch.ones_like(scale), scale)
        self.mean_, self.scale_ = mean.float(), scale.float()
        return self

    def transform(self, x: torch.Tensor) -> torch.Tensor:
        return (x - self.mean_.to(x.device)) / self.scale_.to(x.device)

    def inverse_transform(self, x: torch.Tensor) -> torch.Tensor:
        return x * self.scale_.to(x.device) + self.mean_.to(x.device)

    def fit_transform_3d(self, x3: torch.Tensor, **kw) -> torch.Tensor:
        """Fit over the flattened (group, case) axis and transform (PINN:94-108)."""
        G, C, M = x3.shape
        self.fit(x3.reshape(G * C, M), **kw)
        return self.transform(x3.reshape(G * C, M)).reshape(G, C, M)


@dataclass
class SurrogateData:
    X_train: torch.Tensor      # PINN: [G, n_cases*feat]; TFD: [G, n_cases, feat_padded]
    Y_train: torch.Tensor      # PINN: [G, nelem + 2*(nelem+1)]; TFD: [G, nelem]
    X_val: torch.Tensor
    Y_val: torch.Tensor
    scalers_inputs: Dict[str, StandardScalerT]
    scalers_Y: Dict[str, StandardScalerT]
    min_constraint: torch.Tensor   # 0-dim: min / max of the standardised I targets (PINN:377-378)
    max_constraint: torch.Tensor
    feat_dim: int
    max_lengths: Dict[str, int]
    Fy_train: Optional[torch.Tensor] = None   # [G, N] nodal loads per training group (n_cases == 1 only): physics loss
    Fy_val: Optional[torch.Tensor] = None
    v_train: Optional[torch.Tensor] = None    # [G, N] recorded deflections / rotations (float64, unscaled; n_cases == 1 only):
    theta_train: Optional[torch.Tensor] = None  # the displacement field the I-only models' physics term tests K(I_pred) against


INPUT_KEYS = ("roller_x_locations", "force_x_locations", "force_values", "node_positions")   # PINN:198-201


def _as_rows(v, width=None) -> torch.Tensor:
    if torch.is_tensor(v):
        return v.float()
    w = width if width is not None else max(len(r) for r in v)
    return pad_sequences(v, w)


def prepare(records: Dict[str, object], *, kind: str = "pinn", n_cases: int = 6, c: float = 0.5, train_split: float = 0.8,
            nheads: int = 8, seed: int = 0, device=None, distributed: bool = False, group=None,
            refit_val_scalers: Optional[bool] = None, max_lengths: Optional[Dict[str, int]] = None,
            perm: Optional[torch.Tensor] = None) -> SurrogateData:
    """records: the 13-field dataset (lists as in the reference JSON, or tensors from `generate_dataset`).

    kind = "pinn": flat inputs, targets [I, deflections, rotations] (PINN:337-369)
    kind = "fnn" : flat inputs, targets I only (the FNN sibling: same prep as PINN without the displacement targets)
    kind = "gnn" : as "fnn", but the reference re-fits the input scalers on the validation split (GNN:196-199)
    kind = "fno" : sequence inputs [G, n_cases, feat] without head padding, validation scaled with the training scalers (FNO:277-286)
    kind = "tfd" : sequence inputs padded to a multiple of `nheads`, targets I only (TFD:330-371); the
                   reference re-fits the input scalers on the validation split (TFD:325-328) -- kept behind
                   `refit_val_scalers` (default True for "tfd", False for "pinn").
    With `distributed=True` every rank passes ITS shard of records; moments and constraints are all-reduced.
    `perm`: the group permutation of the split (PINN:261 draws it from numpy's global generator); default: seeded randperm."""
    assert kind in ("pinn", "tfd", "fnn", "gnn", "fno")
    if refit_val_scalers is None:
        refit_val_scalers = kind in ("tfd", "gnn")
    ml = dict(max_lengths or {})
    feats = {}
    for k, short in zip(INPUT_KEYS, ("roller_x", "force_x", "force_values", "node_positions")):
        v = records[k]
        if short not in ml:
            ml[short] = int(v.shape[1]) if torch.is_tensor(v) else max(len(r) for r in v)
        feats[short] = _as_rows(v, ml[short])
    if distributed and dist.is_available() and dist.is_initialized():   # agree on the padded widths
        w = torch.tensor([ml[s] for s in ("roller_x", "force_x", "force_values", "node_positions")], device=device or "cpu")
        dist.all_reduce(w, op=dist.ReduceOp.MAX, group=group)
        for s, val in zip(("roller_x", "force_x", "force_values", "node_positions"), w.tolist()):
            if val != ml[s]:
                feats[s] = torch.nn.functional.pad(feats[s], (0, val - ml[s]))
                ml[s] = val
    I = _as_rows(records["I_values"])
    targets = {"I": I}
    if kind == "pinn":
        targets["deflections"] = _as_rows(records["deflections"])
        targets["rotations"] = _as_rows(records["rotations"])
    dev = torch.device(device) if device is not None else I.device
    S = I.shape[0]
    G = S // n_cases
    if G == 0:
        raise ValueError(f"n_cases={n_cases} > total samples={S}")      # PINN:228-229

    def grouped(t):
        return t[: G * n_cases].to(dev).reshape(G, n_cases, -1)

    feats = {k: grouped(v) for k, v in feats.items()}
    targets = {k: grouped(v) for k, v in targets.items()}
    if perm is None:
        gen = torch.Generator().manual_seed(seed)
        perm = torch.randperm(G, generator=gen)                          # PINN:261 (unseeded there)
    perm = torch.as_tensor(perm, dtype=torch.long).to(dev)
    if perm.numel() != G or not torch.equal(perm.sort().values, torch.arange(G, device=dev)):
        raise ValueError("perm must be a permutation of the group indices")
    n_tr = int(train_split * G)
    tr, va = perm[:n_tr], perm[n_tr:]
    kw = dict(distributed=distributed, group=group)

    sc_in = {k: StandardScalerT() for k in feats}
    Xtr = torch.cat([sc_in[k].fit_transform_3d(feats[k][tr], **kw) for k in feats], dim=2)     # PINN:291-331
    if refit_val_scalers:
        # TFD:325-328 / GNN:196-199 call fit_transform on the SAME scaler objects: from here on `scalers_inputs` -- what the
        # scripts' inference front end scales user inputs with -- holds the VALIDATION statistics (pinned by the fixtures)
        Xva = torch.cat([sc_in[k].fit_transform_3d(feats[k][va], **kw) for k in feats], dim=2)
    else:
        Xva = torch.cat([sc_in[k].transform(feats[k][va]) for k in feats], dim=2)
    feat_dim = Xtr.shape[2]
    if kind in ("pinn", "fnn", "gnn"):
        Xtr, Xva = Xtr.reshape(Xtr.shape[0], -1), Xva.reshape(Xva.shape[0], -1)               # PINN:337-338
    else:
        pad = (-feat_dim) % (1 if kind == "fno" else nheads)                                   # TFD:170-189; FNO: nheads 1
        if pad:
            Xtr = torch.nn.functional.pad(Xtr, (0, pad))
            Xva = torch.nn.functional.pad(Xva, (0, pad))
        feat_dim += pad

    sc_Y, Ytr, Yva = {}, [], []
    for k, t in targets.items():
        ytr, yva = unify_label_with_c(t[tr], c), unify_label_with_c(t[va], c)                   # PINN:341-350
        sc_Y[k] = StandardScalerT().fit(ytr, **kw)                                              # PINN:353-355
        Ytr.append(sc_Y[k].transform(ytr))
        Yva.append(sc_Y[k].transform(yva))
    nel = targets["I"].shape[2]
    Ytr, Yva = torch.cat(Ytr, dim=1), torch.cat(Yva, dim=1)
    mn, mx = Ytr[:, :nel].min(), Ytr[:, :nel].max()                                             # PINN:377-378
    if distributed and dist.is_available() and dist.is_initialized():
        dist.all_reduce(mn, op=dist.ReduceOp.MIN, group=group)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
    Fy_tr = Fy_va = None
    if n_cases == 1 and "force_nodes" in records and "force_values" in records:
        # dense nodal load vectors of every case, for the FE-residual physics term (physics.py)
        Nn = int(records["num_nodes"]) if not isinstance(records["num_nodes"], (list, tuple)) else int(records["num_nodes"][0])
        fn, fv = records["force_nodes"], records["force_values"]
        if torch.is_tensor(fn):       # zero-padded tensors (generate_dataset): slot 0 of the scratch row swallows the padding
            Fy = torch.zeros((G, Nn + 1), dtype=torch.float64, device=fn.device).scatter_add_(
                1, fn[:G].long(), fv[:G].to(torch.float64))[:, 1:]
        else:
            Fy = torch.zeros((G, Nn), dtype=torch.float64)
            for b in range(G):
                for n, f in zip(fn[b], fv[b]):
                    Fy[b, int(n) - 1] += float(f)
        Fy = Fy.to(dev)
        Fy_tr, Fy_va = Fy[tr].contiguous(), Fy[va].contiguous()
    v_tr = th_tr = None
    if n_cases == 1 and "deflections" in records and "rotations" in records:
        as64 = lambda v: (v.to(torch.float64) if torch.is_tensor(v) else pad_sequences(v, len(v[0])).to(torch.float64))  # noqa: E731
        v_tr = as64(records["deflections"])[:G].to(dev)[tr].contiguous()
        th_tr = as64(records["rotations"])[:G].to(dev)[tr].contiguous()
    return SurrogateData(Xtr.contiguous(), Ytr.contiguous(), Xva.contiguous(), Yva.contiguous(), sc_in, sc_Y, mn, mx, feat_dim, ml,
                         Fy_tr, Fy_va, v_tr, th_tr)


def user_inputs(data: SurrogateData, kind: str, roller_x, force_x, force_values, node_positions, nheads: int = 8) -> torch.Tensor:
    """The inference front end of the model scripts (`scale_user_inputs` + `build_user_input_no_agg`, FNN:138-183, :647-657;
    PINN:861-...): n_cases lists per feature -> zero-padded to the training widths, scaled with the TRAINING scalers,
    concatenated per case and flattened ("pinn", "fnn", "gnn") or kept as a sequence ("tfd": padded to a multiple of
    `nheads`; "fno").  Returns X of one sample, on the scalers' device: [1, n_cases * feat] or [1, n_cases, feat]."""
    names = ("roller_x", "force_x", "force_values", "node_positions")
    cols = []
    for name, seqs in zip(names, (roller_x, force_x, force_values, node_positions)):
        sc = data.scalers_inputs[name]
        t = pad_sequences(seqs, data.max_lengths[name]).to(sc.mean_.device)
        cols.append(sc.transform(t))
    X = torch.cat(cols, dim=1)                                   # [n_cases, feat]
    if kind in ("pinn", "fnn", "gnn"):
        return X.reshape(1, -1)
    pad = (-X.shape[1]) % (1 if kind == "fno" else nheads)
    if pad:
        X = torch.nn.functional.pad(X, (0, pad))
    return X.unsqueeze(0)


def predicted_inertia(data: SurrogateData, preds: torch.Tensor) -> torch.Tensor:
    """Standardised model output -> second moments of area: `scaler_Y.inverse_transform` on the inertia columns (FNN:692)."""
    sc = data.scalers_Y["I"]
    n = sc.mean_.numel()
    return sc.inverse_transform(preds[..., :n].float())
