"""`d3f.train_deep_fake.lit_module.LitModule` on the HIP hot path.

Mirrors d3f/train_deep_fake/lit_module.py:30-300: two U-Nets (`model_a`, `model_b`), two Adam
optimisers alternated by `optimizer_idx`, `mode: "denoise"` (noisy real -> real) or `mode: "swap"`
(EMA teacher of the OTHER domain renders a fake, the student denoises the noised fake back to the
real image), and the single-frame inference entry `predict_fake`.  Same hyper-parameter keys as
denoise_config.yml / swap_config.yml; extra optional keys: `synthetic`, `image_size`, `precision`, `augment`,
`pair_fused` (denoise mode: the two nets' optimizer steps of a batch as ONE set of kernel launches -- UnetPair; default on,
`false` = Lightning's one-after-the-other loop) and `pair_plan` (the sequential loop on the pair's kernel choices: the
bit-identical reference of the fused step).

Augmentation: the reference's `A.ShiftScaleRotate(shift_limit=0.2, scale_limit=0.1, rotate_limit=15, border_mode=0,
p=0.7)` after `A.Normalize` (lit_module.py:99-111) runs on the GPU here -- per-sample Bernoulli(0.7), the same
parameter ranges drawn with torch RNG, one HIP warp kernel (ops.affine_warp, K17) with a zero border in normalised
units -- inside `training_step`, before the noise blend, so that 8 GPUs are not fed by 8 CPU warps.
`augment: false` switches it off (benchmarks, parity tests).

Reference quirks kept on purpose: the dataloaders receive `mean_x` as BOTH
mean and std (lit_module.py:75-76); `predict_fake("a")` uses model_a with B's mean/std (:253-257);
de-normalisation truncates with `.int()` BEFORE clamping (:293-294).
"""
import math
import zlib
from datetime import timedelta

import numpy as np
import torch
import torch.nn as nn
import torch.optim.lr_scheduler as schedulers
from torch.utils.data import DataLoader

from .. import ops
from ..dataset.image_dataset import ImageDataset, NormalizeToTensor, SyntheticFaceDataset, ToUint8Tensor
from ..lightning import LightningModule
from ..loss_functions import MseStructuralSimilarityLoss
from ..optim import EMA, FusedAdam
from ..trainer import LearningRateMonitor, ModelCheckpoint
from ..unet import Unet, UnetPair


class ShiftScaleRotate(nn.Module):
    """albumentations.ShiftScaleRotate(shift_limit, scale_limit, rotate_limit, border_mode=0, p) on a normalised
    NCHW batch on the HIP device: with probability p per sample, rotate by U(-rotate, rotate) degrees and scale by
    U(1 - scale, 1 + scale) about the image centre, then shift by U(-shift, shift) x (width, height); bilinear,
    constant-zero border; the other samples pass through untouched."""

    def __init__(self, shift_limit=0.2, scale_limit=0.1, rotate_limit=15.0, p=0.7):
        super().__init__()
        self.shift_limit, self.scale_limit, self.rotate_limit, self.p = shift_limit, scale_limit, rotate_limit, p

    @staticmethod
    def theta(angle_deg, scale, dx, dy, height, width):
        """[B, 2, 3] affine_grid matrices (output -> input, normalised coordinates, align_corners=False) of
        cv2.warpAffine(img, M) with M = getRotationMatrix2D(centre, angle, scale) + (dx * width, dy * height):
        about the centre x - c = (W / 2) u, so  in - c = R(-angle) / scale * ((out - c) - shift)."""
        a = angle_deg * (math.pi / 180.0)
        cos, sin = torch.cos(a) / scale, torch.sin(a) / scale
        a11, a12, a21, a22 = cos, -sin * (height / width), sin * (width / height), cos
        t1 = -2.0 * (a11 * dx + a12 * dy)
        t2 = -2.0 * (a21 * dx + a22 * dy)
        return torch.stack([torch.stack([a11, a12, t1], 1), torch.stack([a21, a22, t2], 1)], 1)

    def draw(self, B, device):
        u = lambda lim: (torch.rand(B, device=device) * 2 - 1) * lim  # noqa: E731
        return {"apply": torch.rand(B, device=device) < self.p, "angle": u(self.rotate_limit),
                "scale": 1.0 + u(self.scale_limit), "dx": u(self.shift_limit), "dy": u(self.shift_limit)}

    @torch.no_grad()
    def forward(self, x, draws=None):
        d = self.draw(x.shape[0], x.device) if draws is None else draws
        th = self.theta(d["angle"], d["scale"], d["dx"], d["dy"], x.shape[2], x.shape[3])
        warped = ops.affine_warp(x, th)
        return torch.where(d["apply"].reshape(-1, 1, 1, 1), warped, x)


class LitModule(LightningModule):
    def __init__(self, **kwargs):
        super().__init__()
        self.save_hyperparameters()
        self.augmentation = self.create_gpu_augmentation()
        self.model_a = self.create_model_instance()
        self.model_b = self.create_model_instance()
        if self.hparams.get("pair_plan", False):
            self.model_a.set_plan_nets(2)
            self.model_b.set_plan_nets(2)
        self.__dict__["_pair"] = None  # UnetPair(model_a, model_b), made on first use (not a sub-module: no state of its own)
        self.ema_model_a = self.create_ema_model(self.model_a)
        self.ema_model_b = self.create_ema_model(self.model_b)
        self.criterion = MseStructuralSimilarityLoss(-1.0, 1.0)

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

    def create_ema_model(self, model):
        p = self.hparams
        if p.mode == "swap":
            return EMA(
                model,
                beta=p.ema_beta,
                update_every=p.ema_update_every,
                include_online_model=False,
            )
        return None

    def train_dataloader(self):
        p = self.hparams
        dataloader_a = self.create_dataloader(p.get("data_path_a"), p.mean_a, p.mean_a, domain="a")
        dataloader_b = self.create_dataloader(p.get("data_path_b"), p.mean_b, p.mean_b, domain="b")
        return {"a": dataloader_a, "b": dataloader_b}

    def create_dataloader(self, path, mean, std, domain=""):
        p = self.hparams
        if p.get("synthetic", False) or path is None:
            # a stable, per-domain seed (str hashes are randomised per process; two None paths must still differ)
            seed = 1234 + zlib.crc32(f"{domain}:{path}".encode()) % 1000
            dataset = SyntheticFaceDataset(p.get("synthetic_length", 8 * p.batch_size), p.get("image_size", 256), seed=seed)
        else:
            dataset = ImageDataset(path, transform=self.create_augmentation_sequence(mean, std))
        workers = p.get("num_workers", 0)
        # shuffle=True and the ragged last batch kept, as the reference (lit_module.py:90-95); workers are SPAWNED:
        # forking a process that has initialised HIP is not safe
        extra = dict(multiprocessing_context="spawn", persistent_workers=True,
                     prefetch_factor=p.get("prefetch_factor", 2)) if workers > 0 else {}
        # pin_memory: the trainer's `.to(device, non_blocking=True)` is only asynchronous from page-locked memory
        return DataLoader(dataset=dataset, batch_size=p.batch_size, num_workers=workers, shuffle=True,
                          pin_memory=bool(p.get("pin_memory", True)) and torch.cuda.is_available(), **extra)

    def create_augmentation_sequence(self, mean, std):
        # host half of the reference's A.Compose: Normalize + ToTensorV2; its ShiftScaleRotate(p=0.7) runs on the
        # GPU inside training_step (create_gpu_augmentation)
        # uint8_batches: true -- the workers hand over the decoded HWC uint8 image and Normalize + ToTensorV2 run on the
        # device too (ops.u8rgb_normalise, bit-identical): 4x fewer bytes through worker IPC and PCIe
        return ToUint8Tensor() if self.hparams.get("uint8_batches", False) else NormalizeToTensor(mean, std)

    def create_gpu_augmentation(self):
        if not self.hparams.get("augment", True):
            return None
        return ShiftScaleRotate(shift_limit=0.2, scale_limit=0.1, rotate_limit=15, p=0.7)

    def configure_optimizers(self):
        p = self.hparams
        b1, b2 = p.adam_b1, p.adam_b2
        tail = bool(p.get("optimizer_overlap_tail", False))  # FusedAdam docstring: part of the update inside backward
        optimizer_a = FusedAdam(self.model_a.parameters(), lr=p.learning_rate, betas=(b1, b2), module=self.model_a,
                                overlap_tail=tail)
        optimizer_b = FusedAdam(self.model_b.parameters(), lr=p.learning_rate, betas=(b1, b2), module=self.model_b,
                                overlap_tail=tail)
        scheduler_a = schedulers.CosineAnnealingLR(optimizer_a, T_max=p.cosine_scheduler_max_epoch)
        scheduler_b = schedulers.CosineAnnealingLR(optimizer_b, T_max=p.cosine_scheduler_max_epoch)
        return [optimizer_a, optimizer_b], [scheduler_a, scheduler_b]

    def optimizer_streams(self, device):
        """one HIP stream per optimizer when the two optimizer steps of a batch are independent of each other -- denoise
        mode: net a learns domain a, net b learns domain b, nothing is shared (d3f/train_deep_fake/lit_module.py:142-181)
        -- so that the trainer can overlap them (trainer.optimizer_steps).  Swap mode couples them through the EMA
        teachers (each step updates and runs the OTHER net's teacher, :183-206): sequential, None.
        OPT-IN (`concurrent_optimizers: true`): bit-identical to the sequential loop (tests/test_gpu_training.py) but
        measured 50 % SLOWER on MI355X / ROCm 7.0 (10.4 -> 15.5 ms per combined batch at bs 8 x 2, 256x256: four
        streams of MFMA-bound work -- two chains, two weight-gradient streams -- serve each other worse than two; the
        third-stream experiments of round 2 pointed the same way, profiles/README.md)."""
        if self.hparams.mode != "denoise" or not self.hparams.get("concurrent_optimizers", False):
            return None
        if self.__dict__.get("_opt_streams") is None:
            self.__dict__["_opt_streams"] = [torch.cuda.Stream(device=device) for _ in range(2)]
        return self.__dict__["_opt_streams"]

    def configure_callbacks(self):
        return [
            LearningRateMonitor(logging_interval="step"),
            ModelCheckpoint(save_top_k=8, monitor="epoch", mode="max", train_time_interval=timedelta(hours=2)),
            ModelCheckpoint(filename="last", save_on_train_epoch_end=True),
        ]

    def training_step(self, batch, batch_idx, optimizer_idx):
        batch_a = batch["a"]["image"]
        batch_b = batch["b"]["image"]
        p = self.hparams
        if optimizer_idx == 0 and batch_a.dtype == torch.uint8:  # a `uint8_batches: true` loader (mean passed as std too,
            batch_a = ops.u8rgb_normalise(batch_a, p.mean_a, p.mean_a)  # like train_dataloader / the reference :75-76)
        if optimizer_idx == 1 and batch_b.dtype == torch.uint8:
            batch_b = ops.u8rgb_normalise(batch_b, p.mean_b, p.mean_b)
        if self.augmentation is not None:
            # the reference augments in the dataset; each image of a combined batch is consumed by exactly one of the
            # two optimiser steps (a by 0, b by 1), so warping the half a step uses is the same thing
            if optimizer_idx == 0:
                batch_a = self.augmentation(batch_a)
            else:
                batch_b = self.augmentation(batch_b)
        if optimizer_idx == 0:
            loss = self.training_step_for_one_model("a", batch_a, self.model_a, self.ema_model_b)
        if optimizer_idx == 1:
            loss = self.training_step_for_one_model("b", batch_b, self.model_b, self.ema_model_a)
        self.log("epoch", float(self.current_epoch))
        return loss

    # ---- the two optimizer steps of a denoise-mode batch as one set of launches ------------------------------------
    def pair_fused_active(self, batch=None, optimizers=None):
        """In `mode: "denoise"` optimizer 0 trains model_a on batch a and optimizer 1 trains model_b on batch b, nothing
        shared (d3f/train_deep_fake/lit_module.py:142-181): the trainer may run both steps as ONE forward / backward of a
        UnetPair (trainer.optimizer_steps).  Not in swap mode (each step updates and runs the OTHER net's EMA teacher,
        :183-206), not with `pair_fused: false`, not for a ragged batch pair, synchronised BatchNorm statistics or
        optimisers other than the two FusedAdam of configure_optimizers."""
        p = self.hparams
        if p.mode != "denoise" or p.get("pair_fused", None) is False or p.get("concurrent_optimizers", False):
            return False
        if optimizers is not None:
            if len(optimizers) != 2 or not all(isinstance(o, FusedAdam) for o in optimizers) or \
                    optimizers[0].module is not self.model_a or optimizers[1].module is not self.model_b:
                return False
        if batch is not None:
            a, b = batch["a"]["image"], batch["b"]["image"]
            if a.shape != b.shape or a.dtype != b.dtype or a.device.type != "cuda":
                return False
        if not (self.model_a.training and self.model_b.training):
            return False
        return not (self.model_a._rt.get("bn_sync") or self.model_b._rt.get("bn_sync"))

    def training_step_pair(self, batch, batch_idx):
        """training_step(batch, batch_idx, 0) and training_step(batch, batch_idx, 1) of a denoise-mode batch as one pass:
        the same preparation and the same random draws in the same order (a's augmentation and noise, then b's), ONE
        forward of UnetPair(model_a, model_b), the two losses.  Returns (loss_a, loss_b); the caller runs ONE backward,
        torch.autograd.backward([loss_a, loss_b]), then both optimizer steps."""
        p = self.hparams
        reals, noisy = [], []
        for key, mean in (("a", p.mean_a), ("b", p.mean_b)):
            x = batch[key]["image"]
            if x.dtype == torch.uint8:
                x = ops.u8rgb_normalise(x, mean, mean)
            if self.augmentation is not None:
                x = self.augmentation(x)
            with torch.no_grad():
                noisy.append(self.blend_random_amount_of_noise_with_each_sample(x))
            reals.append(x)
        if self._pair is None:
            self.__dict__["_pair"] = UnetPair(self.model_a, self.model_b)
        predictions = self._pair(noisy[0], noisy[1])
        losses = []
        for name, prediction, real in zip("ab", predictions, reals):
            loss = self.criterion(prediction, real)
            self.log(f"loss_denoise/train_{name}", loss)
            losses.append(loss)
        self.log("epoch", float(self.current_epoch))
        return tuple(losses)

    def training_step_for_one_model(self, name, real, real_model, fake_model):
        p = self.hparams
        if p.mode == "denoise":
            return self.training_denoise_step_for_one_model(name, real, real_model)
        elif p.mode == "swap":
            return self.training_swap_step_for_one_model(name, real, real_model, fake_model)
        raise ValueError(f"unknown mode {p.mode!r}")

    def training_denoise_step_for_one_model(self, name, real, real_model):
        with torch.no_grad():
            noisy_real = self.blend_random_amount_of_noise_with_each_sample(real)
        real_prediction = real_model(noisy_real)
        loss = self.criterion(real_prediction, real)
        self.log(f"loss_denoise/train_{name}", loss)
        return loss

    def training_swap_step_for_one_model(self, name, real, real_model, fake_model):
        fake_model.update()
        with torch.no_grad():
            fake = fake_model(real)  # train-mode BatchNorm: the EMA wrapper is a registered sub-module
            swap_diff = nn.functional.mse_loss(real, fake)
            noisy_fake = self.blend_random_amount_of_noise_with_each_sample(fake)
        real_prediction = real_model(noisy_fake)
        loss = self.criterion(real_prediction, real)
        self.log(f"swap_difference/{name}", swap_diff)
        self.log(f"loss_swap/train_{name}", loss)
        return loss

    @torch.no_grad()
    def blend_random_amount_of_noise_with_each_sample(self, batch):
        p = self.hparams
        noise = torch.randn_like(batch)  # reference RNG order: randn_like first, then rand
        y = torch.rand(size=(batch.shape[0], 1, 1, 1), device=batch.device)
        return ops.noise_blend(batch, noise, y.reshape(-1), p.noise_exponential_sampling_lambda)

    # ---- single-frame inference (script_tools/put_video_through_fake_model.py:117) ----------------
    def predict_fake(self, real_bgr, model_a_or_b):
        p = self.hparams
        if model_a_or_b == "a":
            return self.predict_fake_for_single_frame(real_bgr, self.model_a, p.mean_b, p.std_b)
        if model_a_or_b == "b":
            return self.predict_fake_for_single_frame(real_bgr, self.model_b, p.mean_a, p.std_a)

    @torch.no_grad()
    def predict_fake_for_single_frame(self, real_bgr, model, mean, std):
        if not model.training and hasattr(model, "predict_u8"):
            # eval mode (script_tools/put_video_through_fake_model.py:49-52 calls .eval()): one C call with the
            # uint8 <-> normalised-tensor conversions fused in.  hipGraph replay (hparam `inference_graph`) is
            # available but off by default: at B=1 the ~100 kernels are GPU-bound (1.24 ms eager vs 1.31 ms
            # replayed at 448x448, profiles/README.md), the host stays ahead of the device either way.
            key = (tuple(real_bgr.shape), id(model))
            bufs = self._frame_buffers.get(key) if hasattr(self, "_frame_buffers") else None
            if bufs is None:
                if not hasattr(self, "_frame_buffers"):
                    self._frame_buffers = {}
                bufs = (torch.empty(real_bgr.shape, dtype=torch.uint8, device=self.device),
                        torch.empty((1,) + tuple(real_bgr.shape), dtype=torch.uint8, device=self.device))
                self._frame_buffers[key] = bufs
            bufs[0].copy_(torch.from_numpy(np.ascontiguousarray(real_bgr)))
            out = model.predict_u8(bufs[0], mean, std, graph=bool(self.hparams.get("inference_graph", False)),
                                   out=bufs[1])
            return out.cpu().numpy()
        mean = torch.tensor(mean, device=self.device, dtype=torch.float32)
        std = torch.tensor(std, device=self.device, dtype=torch.float32)
        input_tensor = self.cv2_to_tensor_normalised(real_bgr, mean, std)
        output_tensor = model(input_tensor)
        return self.tensor_cv2_to_denormalised(output_tensor, mean, std)

    def cv2_to_tensor_normalised(self, image_bgr, mean, std):
        image_rgb = np.ascontiguousarray(image_bgr[:, :, ::-1])  # cv2.COLOR_BGR2RGB
        tensor = torch.from_numpy(image_rgb).float().to(self.device)
        tensor = tensor.permute(2, 0, 1)  # hwc to chw
        tensor = tensor - mean.reshape(3, 1, 1) * 255
        tensor = tensor / (std.reshape(3, 1, 1) * 255)
        return tensor.unsqueeze(0).contiguous()

    def tensor_cv2_to_denormalised(self, tensor, mean, std):
        tensor = tensor.squeeze(0)
        tensor = tensor * (std.reshape(3, 1, 1) * 255)
        tensor = tensor + mean.reshape(3, 1, 1) * 255
        tensor = tensor.permute(1, 2, 0)  # chw to hwc
        tensor = tensor.int()
        tensor = tensor.clamp(0, 255)
        image_rgb = tensor.cpu().numpy().astype(np.uint8)
        return np.ascontiguousarray(image_rgb[:, :, ::-1])  # cv2.COLOR_RGB2BGR
