"""A/B switches of the fused training pieces -- ONE registry instead of one environment variable each (r06).

Every fused piece of the surrogates' training steps can be switched back to the framework composition of the same arithmetic:
that is how each was measured against what it replaced and how the tests state "fused == framework".  Up to r05 each had its own `OPS_AMD_<NAME>`
environment variable (about fifty of them); now they are entries here, read through `get(name)` at the same places and times as before:

    from openpystruct_amd import switches
    switches.set("fused_prep", 0)            # tests, A/B scripts (returns the previous value)
    OPS_AMD_SWITCHES="fused_prep=0,tfd_front=0" python ...      # the one environment variable: parsed once, at import

Names are the old variables' without the prefix, lower case.  Unknown names raise: a typo must not silently measure the default.
The handful of process-level variables that stay variables (OPS_AMD_LIB, OPS_AMD_GRAPH, OPS_AMD_FORCE_DP, OPS_AMD_DP_*, OPS_AMD_SIZING_*,
OPS_AMD_DEBUG_NAN, the build's two) are listed in INTEGRATION.md."""
from __future__ import annotations

import os
from typing import Dict

_DEFAULTS: Dict[str, str] = {
    # PINN step (pinn_fused.py, surrogates.py, train.py)
    "pinn_layer_blocks": "1", "pinn_fused_stencil": "1", "pinn_fused_tails": "1", "pinn_norm_fold": "1", "pinn_repack_in_gather": "1",
    "pinn_engine_eval": "1", "pinn_eval_slots": "1",
    # Transformer-Diffusion step (tfd_fused.py)
    "tfd_fast_encoder": "1", "tfd_layer_fwd": "1", "tfd_layer_bwd": "1", "tfd_layer_pair": "1", "tfd_layer_pair_bwd": "1", "tfd_draw": "1",
    "tfd_eval_fast": "1", "tfd_front": "1", "tfd_front_gather": "1", "tfd_ln_partials": "1", "tfd_head": "1", "tfd_head_loss": "1", "tfd_trace_bwd": "",
    # the loops (train.py)
    "fused_loss": "1", "fused_prep": "1", "fused_physics": "1", "prep_targets": "1", "loss_acc": "1", "adam_zero": "1", "val_whole": "1", "tail_graph": "1",
    "explicit_root": "1", "gx_dest": "1", "wgrad_rows": "1", "split_wgrad_rows": "16", "group_wgrad": "1", "shadow_linear": "1",
}
_values: Dict[str, str] = dict(_DEFAULTS)


def _parse(spec: str) -> None:
    for item in spec.replace(";", ",").split(","):
        item = item.strip()
        if item:
            k, _, v = item.partition("=")
            set(k.strip(), v.strip() if _ else "1")


def get(name: str, default: str | None = None) -> str:
    """The switch's value as a string ("1" / "0" / a number), like the environment variable it replaces.  `default` is accepted for symmetry with
    os.environ.get and must equal the registry's."""
    if name not in _values:
        raise KeyError(f"unknown switch {name!r} (openpystruct_amd/switches.py)")
    return _values[name]


def on(name: str) -> bool:
    return get(name) == "1"


def set(name: str, value) -> str:          # noqa: A001 (the module's verb)
    if name not in _values:
        raise KeyError(f"unknown switch {name!r} (openpystruct_amd/switches.py)")
    old = _values[name]
    _values[name] = str(int(value)) if isinstance(value, bool) else str(value)
    return old


def reset() -> None:
    _values.clear()
    _values.update(_DEFAULTS)
    _parse(os.environ.get("OPS_AMD_SWITCHES", ""))


def snapshot() -> Dict[str, str]:
    """What differs from the defaults (for a bench line / a log)."""
    return {k: v for k, v in _values.items() if v != _DEFAULTS[k]}


_parse(os.environ.get("OPS_AMD_SWITCHES", ""))
