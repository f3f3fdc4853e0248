"""CPU: host-side mirror of the reference's Python surface (no kernels involved)."""
import numpy as np
import torch

from denoising_diffusion_deep_fake_amd.dataset import ImageDataset, SyntheticFaceDataset
from denoising_diffusion_deep_fake_amd.dataset.image_dataset import NormalizeToTensor
from denoising_diffusion_deep_fake_amd.lightning import AttributeDict, LightningModule


def test_image_dataset_contract(tmp_path):
    from PIL import Image
    (tmp_path / "imgs").mkdir()
    rng = np.random.default_rng(0)
    arrs = []
    for i in range(3):
        a = rng.integers(0, 256, size=(32, 32, 3), dtype=np.uint8)
        Image.fromarray(a).save(tmp_path / "imgs" / f"{i}.png")
        arrs.append(a)
    (tmp_path / "images.txt").write_text("imgs/0.png\nimgs/1.png\nimgs/2.png\n")
    ds = ImageDataset(tmp_path / "images.txt", transform=NormalizeToTensor([0.5] * 3, [0.5] * 3))
    assert len(ds) == 3
    item = ds[1]
    assert set(item) == {"image", "index"} and item["index"] == 1
    img = item["image"]
    assert img.shape == (3, 32, 32) and img.dtype == torch.float32
    want = (arrs[1].astype(np.float32) / 255 - 0.5) / 0.5   # RGB order, like cv2 BGR->RGB in the reference
    np.testing.assert_allclose(img.numpy(), want.transpose(2, 0, 1), rtol=0, atol=1e-6)
    raw = ImageDataset(tmp_path / "images.txt")[0]["image"]
    assert raw.dtype == np.uint8 and raw.shape == (32, 32, 3)


def test_synthetic_dataset_matches_oracle_generator():
    import oracle
    ds = SyntheticFaceDataset(4, 64, seed=1234)
    x = ds[2]["image"]
    assert x.shape == (3, 64, 64) and x.abs().max() <= 1
    assert torch.equal(x, oracle.synthetic_face_crops(1, 64, seed=1236)[0])


def test_hparams_surface():
    class M(LightningModule):
        def __init__(self, **kwargs):
            super().__init__()
            self.save_hyperparameters()
            self.lin = torch.nn.Linear(2, 2)

    m = M(batch_size=4, encoder_name="resnet34", mean=[0.5, 0.5, 0.5])
    assert m.hparams.batch_size == 4 and m.hparams["encoder_name"] == "resnet34"
    assert isinstance(m.hparams, AttributeDict) and m.global_step == 0 and m.current_epoch == 0
    m.log("loss", torch.tensor(1.5))
    assert float(m._logged["loss"]) == 1.5
    assert m.device.type == "cpu"


def test_checkpoint_roundtrip(tmp_path):
    class M(LightningModule):
        def __init__(self, **kwargs):
            super().__init__()
            self.save_hyperparameters()
            self.lin = torch.nn.Linear(self.hparams.width, 2)

    m = M(width=3, learning_rate=0.1)
    torch.save({"state_dict": m.state_dict(), "hyper_parameters": dict(m.hparams)}, tmp_path / "a.ckpt")
    m2 = M.load_from_checkpoint(tmp_path / "a.ckpt", learning_rate=0.5)
    assert m2.hparams.width == 3 and m2.hparams.learning_rate == 0.5
    assert torch.equal(m2.lin.weight, m.lin.weight)


def test_ema_decay_schedule_matches_oracle():
    import oracle
    from denoising_diffusion_deep_fake_amd.optim import EMA

    class Fake(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(1))

    ours = EMA.__new__(EMA)
    ours.update_after_step, ours.inv_gamma, ours.power, ours.min_value, ours.beta = 100, 1.0, 2 / 3, 0.0, 0.9999
    ref = oracle.EMA(Fake(), beta=0.9999)
    for step in (0, 50, 101, 102, 150, 1000, 10 ** 7):
        ref.step.fill_(step)
        assert abs(ours.get_current_decay(step) - ref.get_current_decay()) < 1e-12


def test_balance_host_logic_matches_reference_golden(tmp_path):
    """balance_training_images host arithmetic (no kernels): binning against vectors produced by the reference,
    concatenation of validation outputs, and the class list file."""
    from pathlib import Path
    import oracle  # noqa: F401  (same golden file pins the oracle in test_oracle_golden.py)
    from denoising_diffusion_deep_fake_amd.balance_training_images.lit_module import LitModule
    g = np.load(Path(__file__).resolve().parent / "golden" / "balance.npz")
    losses = torch.from_numpy(g["losses"])
    for ncls in (10, 4):
        stub = type("S", (), {"hparams": AttributeDict(number_of_classes=ncls)})()
        idx = LitModule.compute_difficulty_index_for_each_loss(stub, losses)
        assert torch.equal(idx, torch.from_numpy(g[f"difficulty_index_{ncls}"]))
    outs = [{"index": torch.tensor([2, 0]), "loss": torch.tensor([0.5, 0.1])},
            {"index": torch.tensor([1]), "loss": torch.tensor([0.9])}]
    cat = LitModule.concat_validation_output(None, outs)
    assert cat["index"].tolist() == [2, 0, 1] and torch.allclose(cat["loss"], torch.tensor([0.5, 0.1, 0.9]))
    (tmp_path / "images.txt").write_text("a.png\nb.png\nc.png\n")
    stub = type("S", (), {"hparams": AttributeDict(input_image_list_path=str(tmp_path / "images.txt"), synthetic=False),
                          "_data_path": lambda self: self.hparams["input_image_list_path"]})()
    LitModule.write_output_list(stub, str(tmp_path / "out.txt"), cat["index"], torch.tensor([3, 0, 1]))
    assert (tmp_path / "out.txt").read_text() == "a.png\t0\nb.png\t1\nc.png\t3\n"


def test_cli_has_the_reference_commands():
    from denoising_diffusion_deep_fake_amd.main import cli
    assert {"train", "denoise", "balance"} <= set(cli.commands)
    assert {"new", "resume", "modify"} <= set(cli.commands["train"].commands)
