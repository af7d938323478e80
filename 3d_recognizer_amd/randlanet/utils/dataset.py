"""Batched loading of point clouds for training / evaluation (reference
randlanet/utils/dataset.py:11-131): per sample -> sub-sample to n points -> optional
normalisation -> optional augmentation -> (input (n,3+F) float32, labels (n,) int64, index)."""
from typing import Optional, Sequence, Tuple

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from . import preprocessing
from .augmentation import AugmentationSettings, perturbate_point_cloud

Sample = Tuple[np.ndarray, np.ndarray, np.ndarray]


class PointCloudPreprocessor(Dataset):
    def __init__(self, dataset: Sequence[Sample], n_sample_points: int, consistent_sampling: bool = True,
                 augmentation_settings: Optional[AugmentationSettings] = None,
                 normalization: Optional[str] = None) -> None:
        self._dataset = dataset
        self._n_sample_points = n_sample_points
        self._consistent_sampling = consistent_sampling
        self._augmentation_settings = augmentation_settings
        self._normalization = normalization
        self._epoch = 0

    def __len__(self) -> int:
        return len(self._dataset)

    def __getitem__(self, idx: int, preprocess: bool = True) -> Tuple[torch.Tensor, torch.Tensor, int]:
        xyz, features, labels = self._dataset[idx]
        if preprocess:
            xyz, features, labels = self.preprocess(xyz, features, labels)
        point_input = torch.cat((torch.from_numpy(xyz).float(), torch.from_numpy(features).float()), dim=1)
        return point_input, torch.from_numpy(labels).long(), idx

    def preprocess(self, xyz: np.ndarray, features: np.ndarray, labels: np.ndarray) -> Sample:
        n = xyz.shape[0]
        assert xyz.shape[1] == 3, "Point coordinates should have shape (N, 3)!"
        assert features.shape[0] == n, "Features should have shape (N, F)!"
        assert labels.shape == (n,), "Labels should have shape (N,)!"
        keep = preprocessing.sample_points(n, self._n_sample_points, consistent=self._consistent_sampling)
        xyz, features, labels = xyz[keep], features[keep], labels[keep]
        if self._normalization is not None:
            xyz = xyz - np.mean(xyz, axis=0, keepdims=True)
            norms = np.linalg.norm(xyz, axis=1)
            radius = {"mean": np.mean, "max": np.max, "stdev": np.std}.get(self._normalization, lambda _: 1.0)(norms)
            xyz = xyz / radius
        if self._augmentation_settings:
            xyz = perturbate_point_cloud(xyz, self._augmentation_settings)
        return xyz, features, labels


def get_data_loader(dataset: Sequence[Sample], n_sample_points: int, batch_size: int, shuffle: bool = False,
                    consistent_sampling: bool = True, augmentation_settings: Optional[AugmentationSettings] = None,
                    normalization: Optional[str] = None) -> DataLoader:
    prepared = PointCloudPreprocessor(dataset, n_sample_points, consistent_sampling=consistent_sampling,
                                      augmentation_settings=augmentation_settings, normalization=normalization)
    return DataLoader(prepared, batch_size=batch_size, shuffle=shuffle)
