"""The C-ABI shared library loads on a machine without a GPU and exports every symbol
include/openpystruct_amd.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

from openpystruct_amd import _cabi, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build()
    return _cabi.load()


def test_header_symbols_are_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "openpystruct_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(ops_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_cabi.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None


def test_introspection_calls(lib):
    assert lib.ops_amd_abi_version() == 13     # 13: ops_frame_solve_batched_f64_ex (OPS_FRAME_REUSE_PLAN), ops_frame_plan_signature; 12: ops_mlp_wgrad_group_norm, OPS_ADAM_NORM_READY; 11: ops_tfd_front_args.n_order (a walked-past cursor wraps); 10: loss on the TFD head's tile, identity_act of the TFD launches, ops_physics_loss_*; 2: ops_beam_sizing_epoch_f32 takes I_last (float32) instead of I64; 3: ops_mlp_* layer blocks; 4: diffusion combine with a bf16 copy / two gradients; 5: encoder-layer launches on tiled weights; 6: head / front-end launches, column-sum jobs and 24 problems in the grouped weight-gradient launch, ln_part; 7: ops_mlp_strip_args.eval_stats; 8: ops_sizing_draw_cases_f64; 9: evaluation slots of ops_mlp_strip_args
    assert lib.ops_amd_max_elements() >= 100
    assert b"beam_rows_kernel<16, 7" in lib.ops_beam_solve_kernel_name(10000, 100, 0)       # default: the row-staged 16-lane kernel
    assert b"beam_rows_kernel<16, 7" in lib.ops_beam_solve_kernel_name(100000, 100, 0)
    assert b"beam_solve_kernel<8, 13" in lib.ops_beam_solve_kernel_name(1 << 20, 100, 0)   # bandwidth-bound batches: flat aligned streams
    assert b"beam_solve_kernel<16, 7" in lib.ops_beam_solve_kernel_name(10000, 100, 16)     # explicit P without OPS_AMD_TILING_ROWS: beam_solve.hip
    assert b"beam_solve_kernel<32, 4" in lib.ops_beam_solve_kernel_name(10000, 120, 0)      # beyond 16 x 7 nodes: beam_solve.hip
    assert b"beam_rows_kernel<6, 17" in lib.ops_beam_solve_kernel_name(64, 100, 6)          # the fat-wave tiling (csrc/beam_fat.hip)
    assert b"beam_rows_kernel<16, 7" in lib.ops_beam_solve_kernel_name(64, 100, 16 | 0x200)  # OPS_AMD_TILING_ROWS
    assert lib.ops_beam_solve_kernel_name(64, 300, 6) == b""
    assert lib.ops_beam_solve_kernel_name(10, 5000, 0) == b""


def test_argument_validation_without_gpu(lib):
    # invalid arguments are rejected before any HIP call
    z = ctypes.c_void_p(0)
    rc = lib.ops_beam_solve_batched_f64(4, 100, z, 0, z, 0, z, 100, z, 0, z, 101, z, 0, z, z, z, z, z, 0, z)
    assert rc == _cabi.ERR_INVALID_ARG
    rc = lib.ops_beam_solve_batched_f64(-1, 100, z, 0, z, 0, z, 100, z, 0, z, 101, z, 0, z, z, z, z, z, 0, z)
    assert rc == _cabi.ERR_INVALID_ARG
    rc = lib.ops_beam_solve_batched_f64(0, 100, z, 0, z, 0, z, 100, z, 0, z, 101, z, 0, z, z, z, z, z, 0, z)
    assert rc == _cabi.OK   # empty batch is a no-op
    # the other entry points: NULL pointers / bad sizes are refused the same way
    hp = _cabi.SizingParams(E=2e11, G=7.7e10, alpha_moment=1e-2, alpha_shear=1e-2, lr=0.01, gamma=0.98, beta1=0.9, beta2=0.999,
                            adam_eps=1e-8, clamp_min=1e-8, bend_eps=1e-6, area_coef=0.03, tolerance=5e-3, patience=5, max_epochs=10)
    assert lib.ops_beam_solve_forces_f64(4, 100, z, 0, z, 0, z, 100, z, 0, z, 101, z, 0, z, z, z, z, 0, z) == _cabi.ERR_INVALID_ARG
    assert lib.ops_beam_solve_forces_f32(4, 100, z, 0, z, 0, z, 100, z, 0, z, 101, z, 0, z, z, z, z, 0, z) == _cabi.ERR_INVALID_ARG
    assert lib.ops_beam_sizing_step_f32(4, 100, *([z] * 13), ctypes.byref(hp), z) == _cabi.ERR_INVALID_ARG
    assert lib.ops_beam_sizing_step_f32(4, 600, *([z] * 13), ctypes.byref(hp), z) == _cabi.ERR_UNSUPPORTED
    assert lib.ops_beam_sizing_step_vm32_f32(4, 100, *([z] * 11), ctypes.byref(hp), z, z) == _cabi.ERR_INVALID_ARG
    assert lib.ops_beam_sizing_epoch_f32(4, 100, z, 0, z, 0, z, 0, z, 101, z, 0, *([z] * 9), ctypes.byref(hp), z, z, 0, z) == _cabi.ERR_INVALID_ARG
    assert lib.ops_beam_sizing_epoch_f32(4, 200, z, 0, z, 0, z, 0, z, 201, z, 0, *([z] * 9), ctypes.byref(hp), z, z, 0, z) == _cabi.ERR_UNSUPPORTED
    assert lib.ops_frame_solve_batched_f64(2, 4, 3, 6, 5, *([z] * 8), 0, *([z] * 6), 0, z) == _cabi.ERR_INVALID_ARG
    # large batches keep the factor in the caller's workspace: one column-major L per frame (window width 36 / 52 for these bands) + the plan;
    # a frame whose band does not fit LDS needs the workspace at any batch (15 x 16: band + rhs per frame)
    assert lib.ops_frame_workspace_bytes(5000, 330, 35) >= 5000 * 330 * 36 * 8 and lib.ops_frame_workspace_bytes(3, 768, 50) >= 3 * 768 * 53 * 8
    assert lib.ops_frame_workspace_bytes(40000, 90, 17) >= 40000 * 90 * 18 * 8      # packed kernel (half bandwidth <= 29): window width 18
    assert lib.ops_frame_plan_signature(40000, 90, 17) >> 24 == 2 and lib.ops_frame_plan_signature(5000, 330, 35) >> 24 == 1 and lib.ops_frame_plan_signature(3, 330, 35) == 0
    assert lib.ops_amd_get_option(b"frame_pack") == 1 and lib.ops_amd_get_option(b"frame_latency_batch") == -1 and lib.ops_amd_get_option(b"nope") == -2
    assert lib.ops_amd_get_option(b"frame_coop") == 1 and lib.ops_amd_set_option(b"frame_coop", 3) != _cabi.OK
    assert lib.ops_amd_set_option(b"nope", 1) == _cabi.ERR_INVALID_ARG
    assert lib.ops_frame_workspace_bytes(3, 330, 35) == 0        # r05: small batches (<= 256 .. 4 000 frames by size) whose band fits LDS take the workgroup-per-frame kernels
    assert lib.ops_frame_workspace_bytes(0, 330, 35) == 0
    assert lib.ops_stencil3_bn1_fwd_f32(2, 3, z, z, z, z, z, 1e-5, 0.1, 1, z, z, z, z, 0, z, z, z) == _cabi.ERR_INVALID_ARG
    assert lib.ops_stencil3_bn1_bwd_f32(2, 3, z, z, 0, z, z, z, z, 1, z, z, z, z) == _cabi.ERR_INVALID_ARG
    assert lib.ops_flat_clip_adam_step_f32(10, z, z, z, z, z, z, 1.0, 1.0, 0.9, 0.999, 1e-8, 0.0, 0, z, z, z) == _cabi.ERR_INVALID_ARG
    assert lib.ops_surrogate_loss_grad_f32(2, 3, 3, 0, z, 0, z, z, 0.5, z, z, 0.1, 0.0, z, z, z, z) == _cabi.ERR_INVALID_ARG
    assert lib.ops_beam_residual_f64(2, 3, *([z, 0] * 2), z, z, 0, z, z, 0, z, z, z, z, z) == _cabi.ERR_INVALID_ARG
    # layer blocks of the PINN step: broken layout contracts are refused on the host
    a = _cabi.MlpStripArgs()
    assert lib.ops_mlp_strip_launch(None, z) == _cabi.ERR_INVALID_ARG and lib.ops_mlp_strip_launch(a, z) == _cabi.ERR_INVALID_ARG
    a.B, a.N, a.K, a.A, a.lda, a.W, a.ldw, a.Y, a.ldy = 129, 16, 32, 64, 32, 64, 32, 64, 32
    assert lib.ops_mlp_strip_launch(a, z) == _cabi.ERR_INVALID_ARG       # more rows than a workgroup owns
    a.B, a.lda = 128, 24
    assert lib.ops_mlp_strip_launch(a, z) == _cabi.ERR_INVALID_ARG       # reduction not padded to whole 32-column steps
    assert lib.ops_mlp_spart_doubles(350) == 44 * 12
    assert lib.ops_mlp_wgrad_group(0, None, z) == _cabi.ERR_INVALID_ARG and lib.ops_mlp_wgrad_group(9, (_cabi.MlpWgradProblem * 9)(), z) == _cabi.ERR_INVALID_ARG
    assert lib.ops_mlp_repack_weights(1, (_cabi.MlpRepackEntry * 1)(), z) == _cabi.ERR_INVALID_ARG
    assert lib.ops_mlp_gather_noise(200, 684, z, z, z, 1, z, z, 704, z, z, 0, z, z) == _cabi.ERR_INVALID_ARG
    assert lib.ops_mlp_loss_workspace_bytes() > 0
    sched = (ctypes.c_float * 20)()
    lib.ops_sizing_schedule_f32(ctypes.byref(hp), ctypes.cast(sched, ctypes.c_void_p))       # host-only helper
    assert sched[0] == pytest.approx(0.01 / (1 - 0.9), rel=1e-6) and sched[1] == pytest.approx((1 - 0.999) ** 0.5, rel=1e-6)
    assert sched[2] == pytest.approx(0.01 * 0.98 / (1 - 0.81), rel=1e-6)
    # case draws of the generator: counts beyond the eight pick slots, too few nodes, missing outputs
    draw = lambda B, N, R, F, rb, nf, out: lib.ops_sizing_draw_cases_f64(B, 0, 1, N, R, F, rb, z, nf, 15.0, 200.0, -1e6, -1e4, *([out] * 8), z)   # noqa: E731
    assert draw(4, 101, 4, 9, 1, 0, 64) == _cabi.ERR_INVALID_ARG and draw(4, 101, 9, 4, 1, 0, 64) == _cabi.ERR_INVALID_ARG
    assert draw(4, 3, 1, 1, 1, 0, 64) == _cabi.ERR_INVALID_ARG and draw(4, 101, 4, 4, 0, 3, 64) == _cabi.ERR_INVALID_ARG      # fixed rollers without their array
    assert draw(4, 101, 4, 4, 1, 0, z) == _cabi.ERR_INVALID_ARG and draw(0, 101, 4, 4, 1, 0, z) == _cabi.OK


def test_no_cpu_fallback():
    import torch

    import openpystruct_amd as oa

    I = torch.full((2, 10), 0.1, dtype=torch.float64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        oa.beam_solve(torch.linspace(0, 1, 11, dtype=torch.float64), 2e11, I, torch.zeros(11, dtype=torch.uint8),
                      torch.zeros(2, 11, dtype=torch.float64), 0.0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "openpystruct_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(dp, fn)).read()
                assert "import oracle" not in src and "from oracle" not in src and "oracle/" not in src, fn


def test_torch_library_op_is_registered_gpu_only():
    import torch
    import openpystruct_amd  # noqa: F401
    from torch._subclasses.fake_tensor import FakeTensorMode
    op = torch.ops.openpystruct_amd.beam_solve
    with FakeTensorMode():
        I = torch.empty(7, 100, dtype=torch.float64)
        v, th, V, M, st = op(torch.empty(101, dtype=torch.float64), torch.empty((), dtype=torch.float64), I,
                             torch.empty(101, dtype=torch.uint8), torch.empty(7, 101, dtype=torch.float64),
                             torch.empty((), dtype=torch.float64))
    assert v.shape == (7, 101) and th.shape == (7, 101) and V.shape == (7, 100) and M.shape == (7, 100)
    assert st.shape == (7,) and st.dtype == torch.int32
    with pytest.raises(NotImplementedError):            # no CPU kernel behind the operator
        op(torch.zeros(3, dtype=torch.float64), torch.ones((), dtype=torch.float64), torch.ones(1, 2, dtype=torch.float64),
           torch.zeros(3, dtype=torch.uint8), torch.zeros(1, 3, dtype=torch.float64), torch.zeros((), dtype=torch.float64))


def test_frame_dispatch_table_on_the_host():
    """r06: which kernel family and register-window width a frame of the reference's random range takes (FR:17-18: bays, stories ~ U{1..10}; half
    bandwidth 3 m + 2 along the short side m = min(stories, bays + 1)), the workspace the call asks for, and the plan signature a caller keeps its
    plan by -- all host-side functions of the library (no GPU call)."""
    from openpystruct_amd import _cabi
    lib = _cabi.load()
    B = 20000
    packed = 0
    for bays in range(1, 11):
        for stories in range(1, 11):
            m = min(stories, bays + 1)
            kd, n_eq = 3 * m + 2, 3 * stories * (bays + 1)
            sig = int(lib.ops_frame_plan_signature(B, n_eq, kd))
            fam, W, G, P = sig >> 24, (sig >> 16) & 0xFF, (sig >> 8) & 0xFF, sig & 0xFF
            if kd <= 29:
                packed += 1
                assert fam == 2 and W == (kd + 2) // 2 * 2 and P == (16 if kd <= 11 else 32) and G == (4 if (P == 16 or 23 < kd <= 27) else 2 if kd > 27 else 8), (bays, stories, sig)
                assert (kd // G + 1) * G + G <= P                      # the rows in flight below an entering group fit the lane group
                assert int(lib.ops_frame_workspace_bytes(B, n_eq, kd)) >= B * (n_eq + 3) * W * 8
            else:
                assert fam == 1 and W == 36 and P == 64, (bays, stories, sig)
                assert int(lib.ops_frame_workspace_bytes(B, n_eq, kd)) >= B * (n_eq + 4) * W * 8
    assert packed == 98                                               # 98 of the 100 draws (r06, late: kd 29 -- 9 x 9 .. -- joined)
    # beyond the tuned kernels: 56..63 the workgroup kernels (no plan), 64..1024 the column-by-column fallback, both with the band in the workspace
    assert lib.ops_frame_plan_signature(B, 2000, 59) == 0 and lib.ops_frame_plan_signature(B, 2000, 300) == 0
    assert lib.ops_frame_workspace_bytes(B, 2000, 300) >= B * 2000 * 301 * 8
    # small batches: no plan; the option moves the threshold (and is put back)
    assert lib.ops_frame_plan_signature(64, 90, 17) == 0
    try:
        assert lib.ops_amd_set_option(b"frame_latency_batch", 0) == _cabi.OK and lib.ops_frame_plan_signature(64, 90, 17) >> 24 == 2
        assert lib.ops_amd_set_option(b"frame_pack", 0) == _cabi.OK and lib.ops_frame_plan_signature(64, 90, 17) >> 24 == 1
    finally:
        lib.ops_amd_set_option(b"frame_latency_batch", -1); lib.ops_amd_set_option(b"frame_pack", 1)
    assert lib.ops_amd_get_option(b"deterministic") == 0
