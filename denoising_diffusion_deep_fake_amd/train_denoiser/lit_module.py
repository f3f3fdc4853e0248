"""`d3f.train_denoiser.lit_module.LitModule` on the HIP hot path.

Mirrors d3f/train_denoiser/lit_module.py:28-153 -- same hyper-parameter keys
(denoiser_config.yml), same method names, same training_step data flow:
    image -> (affine augmentation) -> blend noise -> self.model(image_noisy) -> criterion(pred, image)
with the U-Net, the loss, the noise blend and the optimiser running as HIP kernels.
Differences, all outside the parity-checked arithmetic:
  * the reference's dataloader transform is bit-rotted (nn.Sequential(T.Normalize) called with
    image=...); this module uses the train_deep_fake convention (Normalize + ToTensor) so the
    command actually runs; `synthetic: true` swaps in the synthetic face-crop dataset;
  * kornia's RandomAffine is replaced by the same parameter ranges drawn with torch RNG and one HIP warp
    kernel (ops.affine_warp = affine_grid + grid_sample(bilinear, zeros), parity-tested against torch on CPU);
    the random draws themselves are not part of the numerics contract; off in benchmarks (`augment: false`).
"""
import math

import torch
import torch.optim.lr_scheduler as schedulers
from torch.utils.data import DataLoader

from .. import ops
from ..dataset.image_dataset import ImageDataset, NormalizeToTensor, SyntheticFaceDataset, ToUint8Tensor
from ..lightning import LightningModule
from ..loss_functions import MseStructuralSimilarityLoss
from ..optim import FusedAdam
from ..unet import Unet


class RandomAffine(torch.nn.Module):
    """per-sample rotation U(-deg,deg), translation U(-t,t)*size, isotropic scale U(s0,s1); bilinear,
    zero padding (kornia RandomAffine(degrees=15, translate=[.2,.2], scale=[.8,1.2], p=1) stand-in)."""

    def __init__(self, degrees=15.0, translate=(0.2, 0.2), scale=(0.8, 1.2)):
        super().__init__()
        self.degrees, self.translate, self.scale = degrees, translate, scale

    def forward(self, x):
        B = x.shape[0]
        dev = x.device
        ang = (torch.rand(B, device=dev) * 2 - 1) * math.radians(self.degrees)
        sc = torch.rand(B, device=dev) * (self.scale[1] - self.scale[0]) + self.scale[0]
        tx = (torch.rand(B, device=dev) * 2 - 1) * self.translate[0] * 2
        ty = (torch.rand(B, device=dev) * 2 - 1) * self.translate[1] * 2
        cos, sin = torch.cos(ang) / sc, torch.sin(ang) / sc
        theta = torch.stack([torch.stack([cos, -sin, tx], 1), torch.stack([sin, cos, ty], 1)], 1)
        return ops.affine_warp(x, theta)  # K17: affine_grid + grid_sample(bilinear, zeros) in one HIP kernel


class LitModule(LightningModule):
    def __init__(self, **kwargs):
        super().__init__()
        self.save_hyperparameters()
        self.model = self.create_model_instance()
        self.training_criterion = MseStructuralSimilarityLoss(-1.0, 1.0)
        self.shared_augmentation_sequence = self.create_shared_augmentation_sequence()
        # graph_step: true -- the whole optimiser step as one captured hipGraph (graph_step.py).  Lightning's MANUAL
        # optimisation contract: training_step does its own backward and optimiser step, the trainer only calls it.
        # Single GPU only.  Measured on ROCm 7.0 / MI355X (profiles/README.md, round 3): bit-identical and SLOWER than the
        # eager step (the replay of a graph spanning the engine's three streams costs ~30 us per node), so it is off
        # unless asked for.
        self.automatic_optimization = not self.hparams.get("graph_step", False)
        self.__dict__["_graph_step"] = None

    def create_model_instance(self):
        p = self.hparams
        return Unet(
            encoder_name=p["encoder_name"],
            encoder_weights=None,
            in_channels=3,
            classes=3,
            activation=None,
            compute_dtype=p.get("precision", "f32"),
        )

    def create_shared_augmentation_sequence(self):
        return RandomAffine(degrees=15, translate=[0.2, 0.2], scale=[0.8, 1.2])

    def train_dataloader(self):
        p = self.hparams
        return self.create_dataloader(p.get("input_image_list_path"), p.get("mean"), p.get("std"))

    def create_dataloader(self, path, mean, std):
        p = self.hparams
        if p.get("synthetic", False) or path is None:
            dataset = SyntheticFaceDataset(p.get("synthetic_length", 64 * p.batch_size), p.get("image_size", 256))
        else:
            # config means/stds are in 0..255 units (denoiser_config.yml:10-11) -> [0,1] units here
            m, s = self.mean_std_unit()
            # uint8_batches: true -- the workers hand over the decoded HWC uint8 image, Normalize + ToTensor run on the
            # device in training_step (normalise_on_device; bit-identical, 4x fewer bytes through IPC and PCIe)
            dataset = ImageDataset(path, transform=ToUint8Tensor() if p.get("uint8_batches", False) else NormalizeToTensor(m, s))
        workers = p.get("num_workers", 0)
        # shuffle=True, ragged last batch kept (d3f/train_denoiser/lit_module.py:78-86); workers are SPAWNED: forking
        # a process that has initialised HIP is not safe
        extra = dict(multiprocessing_context="spawn", persistent_workers=True,
                     prefetch_factor=p.get("prefetch_factor", 2)) if workers > 0 else {}
        # pin_memory: the trainer's `.to(device, non_blocking=True)` is only asynchronous from page-locked memory
        return DataLoader(dataset=dataset, batch_size=p.batch_size, num_workers=workers, shuffle=True,
                          pin_memory=bool(p.get("pin_memory", True)) and torch.cuda.is_available(), **extra)

    def mean_std_unit(self):
        """config means / stds are in 0..255 units (denoiser_config.yml:10-11) -> [0, 1] units"""
        p = self.hparams
        mean, std = p.get("mean"), p.get("std")
        return ([v / 255.0 if max(mean) > 1 else v for v in mean], [v / 255.0 if max(std) > 1 else v for v in std])

    @torch.no_grad()
    def normalise_on_device(self, frames_u8):
        """[B, H, W, 3] uint8 RGB batch of a `uint8_batches: true` loader -> the normalised NCHW float batch the host
        transform would have produced (bit-identical)"""
        m, s = self.mean_std_unit()
        return ops.u8rgb_normalise(frames_u8, m, s)

    def configure_optimizers(self):
        p = self.hparams
        # optimizer_overlap_tail: true -- the update of layer3 / layer4 / decoder / head runs inside backward next to the
        # last gradient bucket (FusedAdam docstring: one step() per backward(); bit-identical values)
        optimizer = FusedAdam(self.model.parameters(), lr=p.learning_rate, module=self.model,
                              overlap_tail=bool(p.get("optimizer_overlap_tail", False)))
        scheduler = schedulers.CosineAnnealingLR(optimizer, T_max=p.cosine_scheduler_max_epoch)
        return [optimizer], [scheduler]

    def forward(self, image):
        return self.model(image)

    def training_step(self, batch, batch_idx):
        image = batch["image"]
        if image.dtype == torch.uint8:
            image = self.normalise_on_device(image)
        if self.hparams.get("augment", True):
            with torch.no_grad():
                image = self.shared_augmentation_sequence(image)
        if not self.automatic_optimization:
            if self._graph_step is None:
                from ..graph_step import GraphTrainStep
                opt = self.optimizers()
                self.__dict__["_graph_step"] = GraphTrainStep(self.model, opt, self.hparams.noise_exponential_sampling_lambda,
                                                              self.training_criterion.input_min_value,
                                                              self.training_criterion.input_max_value)
            loss = self._graph_step(image)
            self.log("loss", loss)
            return loss
        image_noisy = self.blend_random_amount_of_noise_with_each_sample(image)
        image_prediction = self.model(image_noisy)
        loss = self.training_criterion(image_prediction, image)
        self.log("loss", loss)
        return loss

    @torch.no_grad()
    def blend_random_amount_of_noise_with_each_sample(self, batch):
        p = self.hparams
        # RNG call order of the reference: randn_like(batch) first, then rand(B,1,1,1)
        noise = torch.randn_like(batch)
        y = torch.rand(size=(batch.shape[0], 1, 1, 1), device=batch.device)
        return ops.noise_blend(batch, noise, y.reshape(-1), p.noise_exponential_sampling_lambda)

    def sample_random_number_from_exponential_distribution(self, batch_size, lam):
        y = torch.rand(size=(batch_size, 1, 1, 1), device=self.device)
        zeros = torch.zeros(batch_size, 4, device=self.device)
        _, r = ops.noise_blend(zeros, zeros, y.reshape(-1), lam, return_r=True)
        return r.reshape(batch_size, 1, 1, 1)
