"""End-to-end on the GPU: generate a dataset with the HIP path, prepare it on the device, train the three
surrogates for two epochs under bf16 autocast."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("kind", ["pinn", "tfd", "fnn"])
def test_generate_prepare_train_on_gpu(kind):
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    from openpystruct_amd import dataprep, sizing, train
    rec = sizing.generate_dataset(600, sizing.SizingConfig(max_e=30), "cuda", seed=11)
    assert int(rec["status"].abs().sum()) == 0
    d = dataprep.prepare(rec, kind=kind, seed=0, device="cuda")
    assert d.X_train.is_cuda and d.X_train.shape[0] == 80
    if kind == "pinn":
        assert d.X_train.shape[1] == 684 and d.Y_train.shape[1] == 302
    cfg = {"pinn": train.PinnConfig, "tfd": train.TfdConfig, "fnn": train.FnnConfig}[kind](batch_size=32)
    out = train.train_surrogate(kind, d, cfg, device="cuda", max_epochs=2)
    assert out["epochs"] == 2 and np.isfinite(out["history"]["train"]).all() and np.isfinite(out["history"]["val"]).all()
    assert np.isfinite(out["r2_val_I"])
