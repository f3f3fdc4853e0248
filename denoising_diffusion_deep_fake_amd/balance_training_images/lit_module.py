"""balance_training_images LitModule (d3f/balance_training_images/lit_module.py:28-193) on the HIP path.

Trains the same U-Net to denoise at ONE fixed noise ratio, then scores every image by its per-image L1
reconstruction error and bins the scores into `number_of_classes` difficulty classes:

    training_step   : image -> blend_fixed_amount_of_noise -> model -> (MSE + 1 - SSIM) / 2      (:88-107)
    validation_step : same blend, eval-mode forward, compute_difficulty_loss = mean |pred - image| per image (:123-142)
    validation_epoch_end : concatenate, compute_difficulty_index_for_each_loss (min-max, clamp, bin) (:144-193)

Device work is HIP: ops.noise_blend_fixed, the Unet engine, the fused loss, ops.l1_per_image.  Differences from the
reference, on purpose: TensorBoard image / histogram logging is dropped (no tensorboard / matplotlib on the box);
the reference accepts `--output_list` but never writes it (dead option) -- here the classes ARE written, one
"<relative image path>\\t<class>" line per image, when `output_image_list_path` is set.
"""
import torch
from torch.utils.data import DataLoader

from .. import ops
from ..dataset.image_dataset import ImageDataset, NormalizeToTensor, SyntheticFaceDataset
from ..lightning import LightningModule
from ..loss_functions import MseStructuralSimilarityLoss
from ..optim import FusedAdam
from ..unet import Unet


class LitModule(LightningModule):
    def __init__(self, **kwargs):
        super().__init__()
        self.save_hyperparameters()
        self.model = self.create_model_instance()
        self.training_criterion = MseStructuralSimilarityLoss(-1.0, 1.0)
        self.difficulty_index = None  # filled by validation_epoch_end: (image index [N], class [N])

    def create_model_instance(self):
        p = self.hparams
        return Unet(encoder_name=p["encoder_name"], encoder_weights=None, in_channels=3, classes=3, activation=None,
                    compute_dtype=p.get("precision", "f32"))

    def _data_path(self):
        p = self.hparams
        return p.get("input_image_list_path") or p.get("data_path")

    def train_dataloader(self):
        p = self.hparams
        return self.create_dataloader(self._data_path(), p.mean, p.std, shuffle=True)

    def val_dataloader(self):
        p = self.hparams
        return self.create_dataloader(self._data_path(), p.mean, p.std, shuffle=True)  # the reference shuffles here too

    def create_dataloader(self, path, mean, std, shuffle=True):
        p = self.hparams
        if p.get("synthetic", False) or path is None:
            dataset = SyntheticFaceDataset(p.get("synthetic_length", 4 * p.batch_size), p.get("image_size", 256))
        else:
            m = [v / 255.0 if max(mean) > 1 else v for v in mean]
            s = [v / 255.0 if max(std) > 1 else v for v in std]
            dataset = ImageDataset(path, transform=NormalizeToTensor(m, s))
        workers = p.get("num_workers", 0)
        extra = dict(multiprocessing_context="spawn", persistent_workers=True) if workers > 0 else {}  # never fork after HIP init
        return DataLoader(dataset=dataset, batch_size=p.batch_size, num_workers=workers, shuffle=shuffle, **extra)

    def configure_optimizers(self):
        p = self.hparams
        return FusedAdam(self.model.parameters(), lr=p.learning_rate, module=self.model)

    def training_step(self, batch, batch_idx):
        image = batch["image"]
        image_noisy = self.blend_fixed_amount_of_noise_with_each_sample(image)
        image_prediction = self.model(image_noisy)
        loss = self.training_criterion(image_prediction, image)
        self.log("loss", loss)
        return loss

    @torch.no_grad()
    def blend_fixed_amount_of_noise_with_each_sample(self, batch):
        noise = torch.randn_like(batch)
        return ops.noise_blend_fixed(batch, noise, float(self.hparams.ratio_of_noise))

    @torch.no_grad()
    def validation_step(self, batch, batch_idx):
        image = batch["image"]
        image_index = batch["index"]
        image_noisy = self.blend_fixed_amount_of_noise_with_each_sample(image)
        image_prediction = self.model(image_noisy)
        difficulty_loss = self.compute_difficulty_loss(image_prediction, image)
        return {"index": torch.as_tensor(image_index).cpu(), "loss": difficulty_loss.cpu()}

    def compute_difficulty_loss(self, predicted, target):
        return ops.l1_per_image(predicted, target)

    def validation_epoch_end(self, validation_step_output_list):
        tensors = self.concat_validation_output(validation_step_output_list)
        image_index, difficulty_loss = tensors["index"], tensors["loss"]
        difficulty_index = self.compute_difficulty_index_for_each_loss(difficulty_loss)
        self.difficulty_index = (image_index, difficulty_index)
        counts = torch.bincount(difficulty_index, minlength=int(self.hparams.number_of_classes))
        self.log("difficulty_class_max_count", counts.max().float())
        out_path = self.hparams.get("output_image_list_path")
        if out_path:
            self.write_output_list(out_path, image_index, difficulty_index)
        return difficulty_index

    def write_output_list(self, out_path, image_index, difficulty_index):
        names = None
        path = self._data_path()
        if path and not self.hparams.get("synthetic", False):
            with open(path) as f:
                names = [line.strip() for line in f if line.strip()]
        order = torch.argsort(image_index)
        with open(out_path, "w") as f:
            for i in order.tolist():
                idx = int(image_index[i])
                name = names[idx] if names is not None and idx < len(names) else str(idx)
                f.write(f"{name}\t{int(difficulty_index[i])}\n")

    def concat_validation_output(self, validation_step_output_list):
        keys = validation_step_output_list[0].keys()
        return {k: torch.concat([o[k].reshape(-1) for o in validation_step_output_list]) for k in keys}

    def compute_difficulty_index_for_each_loss(self, loss):
        # host arithmetic on the gathered [N] vector, operation for operation as the reference (:181-193)
        p = self.hparams
        loss_min = loss.min()
        loss_max = loss.max()
        loss_normalised = (loss - loss_min) / (loss_max - loss_min)
        loss_normalised = loss_normalised.clamp(0, 0.99999)
        return (loss_normalised * p.number_of_classes).long()
