"""Dataset plugin point of the reference (d3f/dataset/image_dataset.py:8-44), same contract:
`ImageDataset(image_list_path, transform)[i] -> {"image": transform(image=HWC uint8 RGB)["image"], "index": i}`
where `images.txt` lists paths relative to its own directory.  Images are decoded with PIL
(OpenCV is not available on the MI355X image); PIL already yields RGB, i.e. what the reference has
after its BGR->RGB conversion.

`SyntheticFaceDataset` produces the benchmark's "synthetic face crops" behind the same return
contract: spatially correlated fp32 CHW images in [-1, 1].
"""
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F
from torch.utils.data import Dataset


class ImageDataset(Dataset):
    def __init__(self, image_list_path, transform=None):
        self.image_list_path = Path(image_list_path)
        self.transform = transform
        self.image_path_list = self.read_list_of_image_paths()

    def read_list_of_image_paths(self):
        image_path_list = []
        with open(self.image_list_path) as f:
            for relative_image_path in f.readlines():
                relative_image_path = relative_image_path.strip()
                if relative_image_path:
                    image_path_list.append(self.image_list_path.parent / relative_image_path)
        return image_path_list

    def __len__(self):
        return len(self.image_path_list)

    def __getitem__(self, index):
        from PIL import Image
        with Image.open(str(self.image_path_list[index])) as im:
            image = np.asarray(im.convert("RGB"))
        if self.transform is not None:
            image = self.transform(image=image)["image"]
        return {"image": image, "index": index}


def synthetic_face_crops(batch, size, seed=1234, device="cpu"):
    h, w = (size, size) if isinstance(size, int) else size  # an int, or (height, width)
    g = torch.Generator().manual_seed(seed)
    low = torch.randn(batch, 3, h // 16, w // 16, generator=g) * 0.5
    fine = torch.randn(batch, 3, h, w, generator=g) * 0.05
    x = F.interpolate(low, size=(h, w), mode="bilinear", align_corners=False) + fine
    return torch.tanh(x).to(device)


class SyntheticFaceDataset(Dataset):
    def __init__(self, length, size, seed=1234):
        self.length, self.size, self.seed = length, size, seed

    def __len__(self):
        return self.length

    def __getitem__(self, index):
        return {"image": synthetic_face_crops(1, self.size, self.seed + index)[0], "index": index}


class ToUint8Tensor:
    """albumentations-style callable `t(image=HWC uint8)["image"] -> HWC uint8 tensor` (no arithmetic on the host): the
    Normalize + ToTensorV2 half of the reference's A.Compose then runs on the GPU inside training_step
    (ops.u8rgb_normalise, bit-identical to NormalizeToTensor) and the batch crosses worker IPC and PCIe as one byte per
    value instead of four.  Opt-in through the hyper-parameter `uint8_batches: true`."""

    def __call__(self, image):
        return {"image": torch.from_numpy(np.array(image))}  # (a writable copy: PIL's decoded buffer is read-only)


class NormalizeToTensor:
    """albumentations-style callable `t(image=HWC uint8)["image"] -> CHW float`:
    Normalize(mean, std, max_pixel_value=255) + ToTensorV2 (train_deep_fake/lit_module.py:100-110,
    without the random ShiftScaleRotate, which is not part of the parity-checked path)."""

    def __init__(self, mean, std):
        self.mean = np.asarray(mean, dtype=np.float32).reshape(1, 1, 3)
        self.std = np.asarray(std, dtype=np.float32).reshape(1, 1, 3)

    def __call__(self, image):
        x = (image.astype(np.float32) / 255.0 - self.mean) / self.std
        return {"image": torch.from_numpy(np.ascontiguousarray(x.transpose(2, 0, 1)))}
