from .image_dataset import ImageDataset, SyntheticFaceDataset, synthetic_face_crops  # noqa: F401
