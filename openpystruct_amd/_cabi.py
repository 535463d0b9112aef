"""ctypes binding of the C ABI (include/openpystruct_amd.h).

The product path has NO CPU fallback: if the HIP shared library is missing or cannot be
loaded this module raises, loudly, instead of computing anything elsewhere.
"""
from __future__ import annotations

import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
FRAME_REUSE_PLAN = 1     # include/openpystruct_amd.h OPS_FRAME_REUSE_PLAN
ABI_VERSION = 13     # include/openpystruct_amd.h OPS_AMD_ABI_VERSION
# OPS_AMD_LIB lets A/B kernel experiments point at another build of the same C ABI
LIB_PATH = os.environ.get("OPS_AMD_LIB") or os.path.join(_PKG, "lib", "libopenpystruct_amd.so")

# every symbol include/openpystruct_amd.h declares
EXPORTS = (
    "ops_beam_solve_batched_f64",
    "ops_beam_solve_forces_f64",
    "ops_beam_solve_forces_f32",
    "ops_beam_sizing_step_vm32_f32",
    "ops_sizing_schedule_f32",
    "ops_sizing_draw_cases_f64",
    "ops_beam_sizing_epoch_f32",
    "ops_beam_sizing_step_f32",
    "ops_beam_residual_f64",
    "ops_beam_residual_vjp_f64",
    "ops_frame_solve_batched_f64",
    "ops_frame_workspace_bytes",
    "ops_frame_solve_batched_f64_ex",
    "ops_frame_plan_signature",
    "ops_stencil3_bn1_fwd_f32",
    "ops_stencil3_bn1_bwd_f32",
    "ops_stencil3_bn1_workspace_bytes",
    "ops_flat_clip_adam_step_f32",
    "ops_flat_adam_workspace_bytes",
    "ops_surrogate_loss_grad_f32",
    "ops_surrogate_loss_workspace_bytes",
    "ops_amd_set_option",
    "ops_amd_get_option",
    "ops_amd_max_elements",
    "ops_amd_abi_version",
    "ops_gather_rows_noise_f32",
    "ops_fused_bn_act_fwd",
    "ops_fused_bn_act_bwd",
    "ops_amd_last_error",
    "ops_beam_solve_kernel_name",
    "ops_mlp_strip_launch",
    "ops_mlp_spart_doubles",
    "ops_mlp_wgrad_group",
    "ops_mlp_wgrad_group_norm",
    "ops_mlp_gather_noise_repack",
    "ops_mlp_repack_weights",
    "ops_flat_clip_adam_step_repack_f32",
    "ops_mlp_gather_noise",
    "ops_mlp_loss_workspace_bytes",
    "ops_seq_attention_fwd",
    "ops_seq_attention_bwd",
    "ops_dropout_add_layernorm_fwd",
    "ops_dropout_add_layernorm_bwd",
    "ops_act_dropout_fwd",
    "ops_act_dropout_bwd",
    "ops_linear_wgrad_accumulate",
    "ops_linear_wgrad_accumulate_group",
    "ops_diffusion_noise",
    "ops_diffusion_combine_fwd",
    "ops_diffusion_combine_bwd",
    "ops_hbm_copy16",
    "ops_tfd_encoder_layer_fwd",
    "ops_tfd_encoder_layer_pair_fwd",
    "ops_tfd_encoder_layer_pair_bwd",
    "ops_diffusion_noise_draw",
    "ops_surrogate_loss_grad_sum_f32",
    "ops_gather_rows_noise_targets_f32",
    "ops_tfd_encoder_layer_bwd",
    "ops_tfd_head_fwd",
    "ops_tfd_head_bwd",
    "ops_tfd_front_fwd",
    "ops_tfd_front_bwd",
    "ops_physics_loss_part_doubles",
    "ops_physics_loss_fwd",
    "ops_physics_loss_bwd",
)

OK, ERR_INVALID_ARG, ERR_UNSUPPORTED, ERR_LAUNCH = 0, 1, 2, 3
FIX_UY, FIX_RZ = 1, 2

_lib = None


class SizingParams(ctypes.Structure):
    """Mirror of `ops_sizing_params` (include/openpystruct_amd.h)."""
    _fields_ = [(n, ctypes.c_double) for n in (
        "E", "G", "alpha_moment", "alpha_shear", "lr", "gamma", "beta1", "beta2", "adam_eps",
        "clamp_min", "bend_eps", "area_coef", "tolerance")] + [("patience", ctypes.c_int32), ("max_epochs", ctypes.c_int32)]


# layer blocks of the PINN training step (include/openpystruct_amd.h, csrc/mlp_block.hip)
MLP_MAX_ROWS = 128
MLP_TAIL_NONE, MLP_TAIL_ACT_DROP, MLP_TAIL_BN_ACT_DROP, MLP_TAIL_BN = 0, 1, 2, 3
MLP_TAIL_BWD_ACT_DROP, MLP_TAIL_BWD_BN, MLP_TAIL_BWD_BN_ACT_DROP, MLP_TAIL_LOSS = 4, 5, 6, 7
MLP_ADD_NONE, MLP_ADD_FWD_BLOCK, MLP_ADD_BWD_BLOCK = 0, 1, 2
MLP_SIDE_NONE, MLP_SIDE_FWD_STENCIL_STATS, MLP_SIDE_BWD_STENCIL_SUMS = 0, 1, 2
MLP_MAX_WGRAD = 8
MLP_MAX_REPACK = 16
MLP_MAX_NORM_RANGES = 32      # OPS_MLP_MAX_NORM_RANGES
FLAT_ADAM_MAX_PARTS = 1024    # OPS_FLAT_ADAM_MAX_PARTS
ADAM_NORM_READY = 4           # OPS_ADAM_NORM_READY


class MlpStripArgs(ctypes.Structure):
    """Mirror of `ops_mlp_strip_args`."""
    _vp, _i, _f = ctypes.c_void_p, ctypes.c_int32, ctypes.c_float
    _fields_ = [("B", _i), ("N", _i), ("K", _i), ("tail", _i), ("add_mode", _i), ("side", _i),
                ("A", _vp), ("lda", _i), ("W", _vp), ("ldw", _i), ("bias", _vp),
                ("Y", _vp), ("ldy", _i), ("Yt", _vp),
                ("gamma", _vp), ("beta", _vp), ("eps", _f), ("momentum", _f),
                ("running_mean", _vp), ("running_var", _vp), ("num_batches_tracked", _vp),
                ("mean", _vp), ("rstd", _vp), ("Zt", _vp), ("Yref_t", _vp),
                ("slope", _f), ("p_drop", _f), ("seed", ctypes.c_ulonglong), ("call_counter", _vp),
                ("dgamma", _vp), ("dbeta", _vp), ("dbias", _vp),
                ("Ot", _vp), ("No", _i), ("dZt", _vp),
                ("conv_w", _vp), ("conv_b", _vp), ("sgamma", _vp), ("sbeta", _vp), ("seps", _f), ("smomentum", _f),
                ("srunning_mean", _vp), ("srunning_var", _vp), ("snum_batches_tracked", _vp),
                ("ssave", _vp), ("spart", _vp), ("sdparams", _vp),
                ("P", _vp), ("ldp", _i), ("targets_t", _vp), ("nI", _i), ("nD", _i), ("alpha", _vp), ("alpha0", _f),
                ("min_constraint", _vp), ("max_constraint", _vp), ("box_weight", _f), ("rel_penalty", _f), ("loss_ws", _vp), ("loss_finish_rows", _i), ("loss_C", _i), ("loss", _vp), ("loss_sum", _vp),
                ("eval_stats", _i), ("n_slots", _i), ("slot_total_rows", _i), ("slot_stride", ctypes.c_int64)]


class MlpWgradProblem(ctypes.Structure):
    """Mirror of `ops_mlp_wgrad_problem`."""
    _fields_ = [("At", ctypes.c_void_p), ("Bt", ctypes.c_void_p), ("out", ctypes.c_void_p), ("N", ctypes.c_int32), ("K", ctypes.c_int32),
                ("ldo", ctypes.c_int32)]


class WgradProblem(ctypes.Structure):
    """Mirror of `ops_wgrad_problem`."""
    _fields_ = [("T", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32), ("dY", ctypes.c_void_p), ("X", ctypes.c_void_p),
                ("dW", ctypes.c_void_p), ("dbias", ctypes.c_void_p), ("ldy", ctypes.c_int32), ("ldx", ctypes.c_int32)]


class TfdLayerArgs(ctypes.Structure):
    """Mirror of `ops_tfd_layer_args`."""
    _vp, _i, _f, _u = ctypes.c_void_p, ctypes.c_int32, ctypes.c_float, ctypes.c_ulonglong
    _fields_ = [("Bn", _i), ("S", _i), ("H", _i), ("dh", _i), ("d", _i), ("ff", _i), ("x32", _vp),
                ("W_in", _vp), ("b_in", _vp), ("W_out", _vp), ("b_out", _vp), ("W_1", _vp), ("b_1", _vp), ("W_2", _vp), ("b_2", _vp),
                ("gamma1", _vp), ("beta1", _vp), ("eps1", _f), ("gamma2", _vp), ("beta2", _vp), ("eps2", _f),
                ("p_attn", _f), ("p_1", _f), ("p_act", _f), ("p_2", _f),
                ("seed_attn", _u), ("seed_1", _u), ("seed_act", _u), ("seed_2", _u), ("counter", _vp), ("used_call", _vp),
                ("qkv", _vp), ("ctx", _vp), ("z1", _vp), ("mean1", _vp), ("rstd1", _vp), ("y1_16", _vp), ("u", _vp), ("h", _vp),
                ("z2", _vp), ("mean2", _vp), ("rstd2", _vp), ("y32", _vp), ("y16", _vp), ("trace", _vp), ("identity_act", _i)]


class TfdLayerBwdArgs(ctypes.Structure):
    """Mirror of `ops_tfd_layer_bwd_args`."""
    _vp, _i, _f, _u = ctypes.c_void_p, ctypes.c_int32, ctypes.c_float, ctypes.c_ulonglong
    _fields_ = [("Bn", _i), ("S", _i), ("H", _i), ("dh", _i), ("d", _i), ("ff", _i), ("g32", _vp), ("g16", _vp),
                ("Wt_in", _vp), ("Wt_out", _vp), ("Wt_1", _vp), ("Wt_2", _vp), ("gamma1", _vp), ("gamma2", _vp),
                ("p_attn", _f), ("p_1", _f), ("p_act", _f), ("p_2", _f),
                ("seed_attn", _u), ("seed_1", _u), ("seed_act", _u), ("seed_2", _u), ("used_call", _vp),
                ("qkv", _vp), ("z1", _vp), ("mean1", _vp), ("rstd1", _vp), ("u", _vp), ("z2", _vp), ("mean2", _vp), ("rstd2", _vp),
                ("d_f", _vp), ("d_u", _vp), ("d_a", _vp), ("dqkv", _vp), ("dx32", _vp),
                ("dgamma1", _vp), ("dbeta1", _vp), ("dgamma2", _vp), ("dbeta2", _vp), ("trace", _vp), ("ln_part", _vp), ("identity_act", _i)]


class TfdHeadArgs(ctypes.Structure):
    """Mirror of `ops_tfd_head_args`."""
    _vp, _i, _f, _u = ctypes.c_void_p, ctypes.c_int32, ctypes.c_float, ctypes.c_ulonglong
    _fields_ = [("B", _i), ("S", _i), ("d", _i), ("hid", _i), ("C", _i), ("y16", _vp), ("W1", _vp), ("b1", _vp), ("gamma", _vp), ("beta", _vp),
                ("eps", _f), ("W2", _vp), ("b2", _vp), ("p_drop", _f), ("seed", _u), ("counter", _vp), ("used_call", _vp),
                ("a16", _vp), ("mean", _vp), ("rstd", _vp), ("h", _vp), ("out", _vp),
                ("targets", _vp), ("grad", _vp), ("loss_part", _vp), ("alpha", _vp), ("min_constraint", _vp), ("max_constraint", _vp),
                ("box_weight", _f), ("identity_act", _i), ("target_rows", _vp)]


class TfdHeadBwdArgs(ctypes.Structure):
    """Mirror of `ops_tfd_head_bwd_args`."""
    _vp, _i, _f = ctypes.c_void_p, ctypes.c_int32, ctypes.c_float
    _fields_ = [("B", _i), ("S", _i), ("d", _i), ("hid", _i), ("C", _i), ("g", _vp), ("Wt2", _vp), ("Wt1", _vp), ("gamma", _vp), ("p_drop", _f),
                ("a16", _vp), ("mean", _vp), ("rstd", _vp), ("h", _vp), ("d_a", _vp), ("dcls_rows", _vp), ("dgamma", _vp), ("dbeta", _vp),
                ("loss_part", _vp), ("alpha", _vp), ("alpha0", _f), ("box_weight", _f), ("loss", _vp), ("loss_sum", _vp), ("g2", _vp), ("g_sum", _vp)]


class PhysicsLossArgs(ctypes.Structure):
    """Mirror of `ops_physics_loss_args`."""
    _vp, _i, _f, _d = ctypes.c_void_p, ctypes.c_int32, ctypes.c_float, ctypes.c_double
    _fields_ = [("B", _i), ("Ne", _i), ("preds", _vp), ("preds_bf16", _i), ("ldp", _i), ("I_scale", _vp), ("I_mean", _vp), ("I_min", _f),
                ("v_rec", _vp), ("t_rec", _vp), ("v_scale", _vp), ("v_mean", _vp), ("t_scale", _vp), ("t_mean", _vp), ("rows", _vp),
                ("Fy", _vp), ("x", _vp), ("fix", _vp), ("E", _d), ("wy", _d), ("weight", _f), ("ev", _vp), ("et", _vp), ("part", _vp),
                ("value", _vp), ("value_sum", _vp), ("dpreds", _vp)]


class TfdFrontArgs(ctypes.Structure):
    """Mirror of `ops_tfd_front_args`."""
    _vp, _i, _u = ctypes.c_void_p, ctypes.c_int32, ctypes.c_ulonglong
    _fields_ = [("B", _i), ("Nc", _i), ("d", _i), ("hid", _i), ("T", _i), ("x", _vp), ("alpha_cumprod", _vp), ("seed", _u), ("counter", _vp),
                ("W0", _vp), ("b0", _vp), ("W2", _vp), ("b2", _vp), ("cls", _vp), ("pe", _vp),
                ("xn16", _vp), ("h", _vp), ("sa", _vp), ("sb", _vp), ("z", _vp), ("z16", _vp), ("t_out", _vp), ("eps_out", _vp), ("identity_act", _i),
                ("src", _vp), ("order", _vp), ("cursor", _vp), ("idx_out", _vp), ("sigma", _vp), ("in_seed", _u), ("n_order", ctypes.c_longlong)]


class TfdFrontBwdArgs(ctypes.Structure):
    """Mirror of `ops_tfd_front_bwd_args`."""
    _vp, _i = ctypes.c_void_p, ctypes.c_int32
    _fields_ = [("B", _i), ("Nc", _i), ("d", _i), ("hid", _i), ("g32", _vp), ("g16", _vp), ("sa", _vp), ("sb", _vp), ("h", _vp), ("Wt2", _vp),
                ("dm", _vp), ("d_h", _vp), ("dcls", _vp)]


WGRAD_MAX_GROUP = 24


class MlpRepackEntry(ctypes.Structure):
    """Mirror of `ops_mlp_repack_entry`."""
    _fields_ = [("W", ctypes.c_void_p), ("N", ctypes.c_int32), ("K", ctypes.c_int32), ("Wp", ctypes.c_void_p), ("ldw", ctypes.c_int32),
                ("Wtp", ctypes.c_void_p), ("ldwt", ctypes.c_int32)]


class ExtensionMissingError(RuntimeError):
    pass


def load():
    """Load the HIP library once; raise ExtensionMissingError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ExtensionMissingError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -m openpystruct_amd.build` "
            "(needs hipcc). openpystruct_amd has no CPU fallback for the beam solve."
        )
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the machine
        raise ExtensionMissingError(f"cannot load {LIB_PATH}: {e}") from e
    vp, lg, it = ctypes.c_void_p, ctypes.c_long, ctypes.c_int
    f = lib.ops_beam_solve_batched_f64
    f.restype = it
    f.argtypes = [it, it, vp, lg, vp, lg, vp, lg, vp, lg, vp, lg, vp, lg, vp, vp, vp, vp, vp, it, vp]
    ff = lib.ops_beam_solve_forces_f64
    ff.restype = it
    ff.argtypes = [it, it, vp, lg, vp, lg, vp, lg, vp, lg, vp, lg, vp, lg, vp, vp, vp, vp, it, vp]
    lib.ops_beam_solve_forces_f32.restype = it
    lib.ops_beam_solve_forces_f32.argtypes = ff.argtypes
    lib.ops_beam_sizing_step_vm32_f32.restype = it
    lib.ops_beam_sizing_step_vm32_f32.argtypes = [it, it] + [vp] * 11 + [ctypes.POINTER(SizingParams), vp, vp]
    lib.ops_sizing_schedule_f32.restype = None
    lib.ops_sizing_schedule_f32.argtypes = [ctypes.POINTER(SizingParams), vp]
    lib.ops_sizing_draw_cases_f64.restype = it
    _ull, _db = ctypes.c_ulonglong, ctypes.c_double
    lib.ops_sizing_draw_cases_f64.argtypes = [lg, _ull, _ull, it, it, it, it, vp, it, _db, _db, _db, _db, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.ops_beam_sizing_epoch_f32.restype = it
    lib.ops_beam_sizing_epoch_f32.argtypes = [it, it, vp, lg, vp, lg, vp, lg, vp, lg, vp, lg] + [vp] * 9 + [ctypes.POINTER(SizingParams), vp, vp, it, vp]
    g = lib.ops_beam_sizing_step_f32
    g.restype = it
    g.argtypes = [it, it] + [vp] * 13 + [ctypes.POINTER(SizingParams), vp]
    r = lib.ops_beam_residual_f64
    r.restype = it
    r.argtypes = [it, it, vp, lg, vp, lg, vp, vp, lg, vp, vp, lg, vp, vp, vp, vp, vp]
    rj = lib.ops_beam_residual_vjp_f64
    rj.restype = it
    rj.argtypes = [it, it, vp, lg, vp, lg, vp, vp, lg] + [vp] * 10
    lib.ops_tfd_encoder_layer_fwd.restype = it
    lib.ops_tfd_encoder_layer_fwd.argtypes = [ctypes.POINTER(TfdLayerArgs), vp]
    lib.ops_tfd_encoder_layer_pair_fwd.restype = it
    lib.ops_tfd_encoder_layer_pair_fwd.argtypes = [ctypes.POINTER(TfdLayerArgs), ctypes.POINTER(TfdLayerArgs), vp]
    lib.ops_tfd_encoder_layer_pair_bwd.restype = it
    lib.ops_tfd_encoder_layer_pair_bwd.argtypes = [ctypes.POINTER(TfdLayerBwdArgs), ctypes.POINTER(TfdLayerBwdArgs), vp]
    lib.ops_tfd_encoder_layer_bwd.restype = it
    lib.ops_tfd_encoder_layer_bwd.argtypes = [ctypes.POINTER(TfdLayerBwdArgs), vp]
    lib.ops_tfd_head_fwd.restype = it
    lib.ops_tfd_head_fwd.argtypes = [ctypes.POINTER(TfdHeadArgs), vp]
    lib.ops_tfd_head_bwd.restype = it
    lib.ops_tfd_head_bwd.argtypes = [ctypes.POINTER(TfdHeadBwdArgs), vp]
    lib.ops_physics_loss_part_doubles.restype = ctypes.c_size_t
    lib.ops_physics_loss_part_doubles.argtypes = [it, it]
    lib.ops_physics_loss_fwd.restype = it
    lib.ops_physics_loss_fwd.argtypes = [ctypes.POINTER(PhysicsLossArgs), vp]
    lib.ops_physics_loss_bwd.restype = it
    lib.ops_physics_loss_bwd.argtypes = [ctypes.POINTER(PhysicsLossArgs), vp]
    lib.ops_tfd_front_fwd.restype = it
    lib.ops_tfd_front_fwd.argtypes = [ctypes.POINTER(TfdFrontArgs), vp]
    lib.ops_tfd_front_bwd.restype = it
    lib.ops_tfd_front_bwd.argtypes = [ctypes.POINTER(TfdFrontBwdArgs), vp]
    lib.ops_hbm_copy16.restype = it
    lib.ops_hbm_copy16.argtypes = [vp, vp, ctypes.c_size_t, it, vp]
    fr = lib.ops_frame_solve_batched_f64
    fr.restype = it
    fr.argtypes = [it] * 5 + [vp] * 8 + [lg] + [vp] * 6 + [ctypes.c_size_t, vp]
    frx = lib.ops_frame_solve_batched_f64_ex
    frx.restype = it
    frx.argtypes = [it] * 5 + [vp] * 8 + [lg] + [vp] * 6 + [ctypes.c_size_t, vp, ctypes.c_uint]
    lib.ops_frame_plan_signature.restype = lg
    lib.ops_frame_plan_signature.argtypes = [it, it, it]
    lib.ops_frame_workspace_bytes.restype = ctypes.c_size_t
    lib.ops_frame_workspace_bytes.argtypes = [it, it, it]
    fl = ctypes.c_float
    lib.ops_stencil3_bn1_fwd_f32.restype = it
    lib.ops_stencil3_bn1_fwd_f32.argtypes = [it, it, vp, vp, vp, vp, vp, fl, fl, it, vp, vp, vp, vp, it, vp, vp, vp]
    lib.ops_stencil3_bn1_bwd_f32.restype = it
    lib.ops_stencil3_bn1_bwd_f32.argtypes = [it, it, vp, vp, it, vp, vp, vp, vp, it, vp, vp, vp, vp]
    lib.ops_stencil3_bn1_workspace_bytes.restype = ctypes.c_size_t
    lib.ops_flat_clip_adam_step_f32.restype = it
    lib.ops_flat_clip_adam_step_f32.argtypes = [lg, vp, vp, vp, vp, vp, vp, fl, fl, fl, fl, fl, fl, it, vp, vp, vp]
    lib.ops_flat_adam_workspace_bytes.restype = ctypes.c_size_t
    lib.ops_surrogate_loss_grad_f32.restype = it
    lib.ops_surrogate_loss_grad_f32.argtypes = [it, it, it, it, vp, it, vp, vp, fl, vp, vp, fl, fl, vp, vp, vp, vp]
    lib.ops_surrogate_loss_grad_sum_f32.restype = it
    lib.ops_surrogate_loss_grad_sum_f32.argtypes = [it, it, it, it, vp, it, vp, vp, fl, vp, vp, fl, fl, vp, vp, vp, vp, vp]
    lib.ops_surrogate_loss_workspace_bytes.restype = ctypes.c_size_t
    ull = ctypes.c_ulonglong
    lib.ops_fused_bn_act_fwd.restype = it
    lib.ops_fused_bn_act_fwd.argtypes = [it, it, vp, vp, vp, it, vp, vp, fl, fl, it, vp, vp, vp, fl, it, fl, ull, vp, vp, vp, vp, vp, vp, vp]
    lib.ops_gather_rows_noise_f32.restype = it
    lib.ops_gather_rows_noise_f32.argtypes = [it, lg, vp, vp, vp, ull, vp, vp, it, vp]
    lib.ops_gather_rows_noise_targets_f32.restype = it
    lib.ops_gather_rows_noise_targets_f32.argtypes = [it, lg, vp, vp, vp, ull, vp, vp, it, vp, it, vp, vp]
    lib.ops_fused_bn_act_bwd.restype = it
    lib.ops_fused_bn_act_bwd.argtypes = [it, it, vp, it, vp, vp, vp, vp, vp, fl, it, fl, vp, vp, vp, vp, vp]
    lib.ops_amd_set_option.restype = it
    lib.ops_amd_set_option.argtypes = [ctypes.c_char_p, lg]
    lib.ops_amd_get_option.restype = lg
    lib.ops_amd_get_option.argtypes = [ctypes.c_char_p]
    lib.ops_mlp_strip_launch.restype = it
    lib.ops_mlp_strip_launch.argtypes = [ctypes.POINTER(MlpStripArgs), vp]
    lib.ops_mlp_spart_doubles.restype = ctypes.c_size_t
    lib.ops_mlp_spart_doubles.argtypes = [it]
    lib.ops_mlp_wgrad_group.restype = it
    lib.ops_mlp_wgrad_group.argtypes = [it, ctypes.POINTER(MlpWgradProblem), vp]
    lib.ops_mlp_wgrad_group_norm.restype = it
    lib.ops_mlp_wgrad_group_norm.argtypes = [it, ctypes.POINTER(MlpWgradProblem), it, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_int32), fl, vp, vp,
                                             fl, fl, ctypes.POINTER(ctypes.c_int32), vp]
    lib.ops_mlp_repack_weights.restype = it
    lib.ops_mlp_repack_weights.argtypes = [it, ctypes.POINTER(MlpRepackEntry), vp]
    lib.ops_flat_clip_adam_step_repack_f32.restype = it
    lib.ops_flat_clip_adam_step_repack_f32.argtypes = [lg, vp, vp, vp, vp, vp, vp, fl, fl, fl, fl, fl, fl, it, vp, vp, it,
                                                       ctypes.POINTER(MlpRepackEntry), vp]
    lib.ops_mlp_gather_noise.restype = it
    lib.ops_mlp_gather_noise.argtypes = [it, it, vp, vp, vp, ull, vp, vp, it, vp, vp, it, vp, vp]
    lib.ops_mlp_gather_noise_repack.restype = it
    lib.ops_mlp_gather_noise_repack.argtypes = [it, it, vp, vp, vp, ull, vp, vp, it, vp, vp, it, vp, lg, vp, it, ctypes.POINTER(MlpRepackEntry), vp]
    lib.ops_mlp_loss_workspace_bytes.restype = ctypes.c_size_t
    lib.ops_seq_attention_fwd.restype = it
    lib.ops_seq_attention_fwd.argtypes = [it, it, it, it, vp, vp, fl, ull, vp, vp, vp]
    lib.ops_seq_attention_bwd.restype = it
    lib.ops_seq_attention_bwd.argtypes = [it, it, it, it, vp, vp, vp, fl, ull, vp, vp]
    lib.ops_dropout_add_layernorm_fwd.restype = it
    lib.ops_dropout_add_layernorm_fwd.argtypes = [it, it, vp, vp, it, vp, vp, fl, fl, ull, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.ops_dropout_add_layernorm_bwd.restype = it
    lib.ops_dropout_add_layernorm_bwd.argtypes = [it, it, vp, vp, vp, vp, vp, vp, fl, ull, vp, vp, vp, vp, vp, vp]
    lib.ops_act_dropout_fwd.restype = it
    lib.ops_act_dropout_fwd.argtypes = [lg, vp, vp, fl, fl, ull, vp, vp, vp]
    lib.ops_diffusion_noise.restype = it
    lib.ops_diffusion_noise.argtypes = [lg, it, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.ops_diffusion_noise_draw.restype = it
    lib.ops_diffusion_noise_draw.argtypes = [lg, it, it, vp, vp, ull, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.ops_diffusion_combine_fwd.restype = it
    lib.ops_diffusion_combine_fwd.argtypes = [it, it, it, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.ops_diffusion_combine_bwd.restype = it
    lib.ops_diffusion_combine_bwd.argtypes = [it, it, it, vp, vp, vp, vp, vp, vp, vp]
    lib.ops_linear_wgrad_accumulate_group.restype = it
    lib.ops_linear_wgrad_accumulate_group.argtypes = [it, ctypes.POINTER(WgradProblem), vp]
    lib.ops_linear_wgrad_accumulate.restype = it
    lib.ops_linear_wgrad_accumulate.argtypes = [it, it, it, vp, vp, vp, vp, vp]
    lib.ops_act_dropout_bwd.restype = it
    lib.ops_act_dropout_bwd.argtypes = [lg, vp, vp, vp, fl, fl, ull, vp, vp]
    lib.ops_amd_max_elements.restype = it
    lib.ops_amd_abi_version.restype = it
    lib.ops_amd_last_error.restype = ctypes.c_char_p
    lib.ops_beam_solve_kernel_name.restype = ctypes.c_char_p
    lib.ops_beam_solve_kernel_name.argtypes = [it, it, it]
    if lib.ops_amd_abi_version() != ABI_VERSION:     # a stale build of another ABI must not be driven with today's argument lists
        raise ExtensionMissingError(f"{LIB_PATH} has C-ABI version {lib.ops_amd_abi_version()}, "
                                    f"this package needs {ABI_VERSION}: rebuild with `python -m openpystruct_amd.build --force`")
    _lib = lib
    return lib


def set_option(name: str, value: int) -> None:
    """include/openpystruct_amd.h ops_amd_set_option: "frame_latency_batch" (-1 = the library's model, 0 = tuned kernels for every batch),
    "frame_pack" (0 = one wave per frame for every half bandwidth), "frame_coop" (0 / 1 / 2: four waves per frame for small batches never / where
    measured faster / always), "deterministic".  The library reads no environment variable."""
    if load().ops_amd_set_option(name.encode(), int(value)) != OK:
        raise ValueError(f"unknown library option {name!r}")


def get_option(name: str) -> int:
    v = int(load().ops_amd_get_option(name.encode()))
    if v == -2:
        raise ValueError(f"unknown library option {name!r}")
    return v
