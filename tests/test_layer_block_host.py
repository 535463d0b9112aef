"""Host-side pieces of the layer-block paths that need no GPU: the fragment-tiled layout helpers (include/openpystruct_amd.h layout
contract), eligibility of the PINN engine, and that the CPU training loops never touch either fast path."""
import torch
import torch.nn as nn


def test_tiled_layout_is_the_mfma_fragment_order():
    from openpystruct_amd.pinn_fused import from_tiled, to_tiled
    x = torch.arange(48 * 96, dtype=torch.float32).reshape(48, 96)
    t = to_tiled(x)
    assert t.shape == x.shape and torch.equal(from_tiled(t), x)
    flat = t.reshape(-1)
    KS = 96 // 32
    for row, k in ((0, 0), (17, 40), (47, 95), (5, 33), (16, 31), (31, 32)):
        off = ((row >> 4) * KS + (k >> 5)) * 512 + ((k >> 3) & 3) * 128 + (row & 15) * 8 + (k & 7)
        assert float(flat[off]) == float(x[row, k])
    # one wave-wide 16-byte load = one tile: lane l = (k >> 3 & 3) << 4 | row & 15 holds 8 consecutive k of its row
    tile = flat[512:1024].reshape(64, 8)                      # tile (row block 0, k step 1)
    for lane in (0, 15, 16, 37, 63):
        r, kg = lane & 15, lane >> 4
        assert torch.equal(tile[lane], x[r, 32 + 8 * kg:32 + 8 * kg + 8])


def test_pinn_engine_eligibility_is_conservative():
    from openpystruct_amd import pinn_fused
    from openpystruct_amd.surrogates import CompositeLoss, FNNWithResidual, TrainableL1L2Loss
    crit = CompositeLoss(100, 101, 101)
    ok = FNNWithResidual(684, 350, 2, 302, 0.5)
    assert not pinn_fused.eligible(ok, crit, 128)                                   # CPU parameters
    assert not pinn_fused.eligible(FNNWithResidual(684, 350, 2, 302, 0.5, norm_type="layer"), crit, 128)
    assert not pinn_fused.eligible(FNNWithResidual(684, 350, 2, 302, 0.5, use_conv=False), crit, 128)
    assert not pinn_fused.eligible(ok, TrainableL1L2Loss(), 128)
    assert not pinn_fused.eligible(ok, crit, 256)                                   # more rows than a workgroup owns
    assert not pinn_fused.eligible(nn.Linear(4, 4), crit, 128)


def test_tfd_patch_needs_the_shadow_products():
    from openpystruct_amd import tfd_fused
    from openpystruct_amd.surrogates import ModelOnePassTransformerWithDiffusion
    m = ModelOnePassTransformerWithDiffusion(6, 120, 100)
    assert not tfd_fused.patch_model(m, seed=1)                                     # no shadow products registered: nothing patched
    assert "forward" not in m.__dict__ and "forward" not in m.transformer_encoder.__dict__
    x = torch.randn(3, 6, 120)
    m.eval()
    with torch.no_grad():
        assert m(x).shape == (3, 100)
