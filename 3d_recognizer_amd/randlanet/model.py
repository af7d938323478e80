"""`Model`: the facade train.py / predict.py talk to (reference randlanet/model.py:21-336) -
construct, load / save (zip of `config` json + `model` state_dict, interchangeable with the
reference's files), predict (confidences), upsample, train, evaluate - on the MI355X kernels."""
import json
import logging
import os
import shutil
import tempfile
from collections import OrderedDict
from dataclasses import asdict
from pathlib import Path
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _ops as ops
from .utils.augmentation import AugmentationSettings
from .utils.dataset import get_data_loader
from .utils.device_dataset import get_device_data_loader
from .utils.modules import RandLANet, RandLANetSettings, UpSampler
from .utils.preprocessing import sample_points
from .utils.trainer import Trainer, TrainingSettings

Sample = Tuple[np.ndarray, np.ndarray, np.ndarray]


def _pick_device(use_gpu: bool) -> torch.device:
    return torch.device("cuda" if (torch.cuda.is_available() and use_gpu) else "cpu")


class Model:
    def __init__(self, settings: RandLANetSettings, weights: Optional[OrderedDict] = None, use_gpu: bool = True):
        self.device = _pick_device(use_gpu)
        self._model = RandLANet(settings, self.device)
        if weights is not None:
            self._model.load_state_dict(weights)
        self._model.eval()
        self._upsampler = UpSampler(settings.upsampling, self.device)

    def __del__(self):
        try:
            torch.cuda.empty_cache()
        except AttributeError:
            pass

    def __str__(self) -> str:
        return str(self._model)

    @property
    def settings(self) -> RandLANetSettings:
        return self._model.settings

    @property
    def module(self) -> torch.nn.Module:
        return self._model

    # ------------------------------------------------------------------------- persistence
    @staticmethod
    def load(path: Path, use_gpu: bool = True, **kwargs) -> "Model":
        """Read a model zip (model.py:77-105); keyword arguments override stored settings."""
        path = Path(path)
        assert path.is_file(), f"Could not find model file at {path}!"
        device = _pick_device(use_gpu)
        with tempfile.TemporaryDirectory() as tmp:
            shutil.unpack_archive(str(path), tmp, format="zip")
            with open(os.path.join(tmp, "config")) as f:
                settings = RandLANetSettings(**json.load(f))
            state = torch.load(os.path.join(tmp, "model"), map_location=device)
        if "model" in state.keys():
            state = state["model"]
        settings.update(**kwargs)
        return Model(settings, weights=state, use_gpu=use_gpu)

    def save(self, path: Path) -> None:
        """Write `config` (json of the settings) + `model` (state_dict) as one zip at `path`."""
        path = Path(path)
        os.makedirs(path.parent, exist_ok=True)
        with tempfile.TemporaryDirectory() as payload, tempfile.TemporaryDirectory() as out:
            with open(os.path.join(payload, "config"), "w") as f:
                json.dump(asdict(self.settings), f)
            torch.save(self._model.state_dict(), os.path.join(payload, "model"))
            archive = shutil.make_archive(os.path.join(out, "file"), "zip", payload)
            shutil.move(archive, str(path))

    # --------------------------------------------------------------------------- inference
    def upsample(self, logits: torch.Tensor, xyz: torch.Tensor, xyz_upsampled: torch.Tensor) -> torch.Tensor:
        """Softmax confidences of `logits` (B,C,N1) carried to `xyz_upsampled` (B,N2,3) -> (B,C,N2)."""
        if self.device.type != "cuda":
            conf = torch.softmax(logits.to("cpu", torch.float32), dim=1)
        else:
            with torch.cuda.device(self.device):
                conf = ops.softmax_cf(logits.to(self.device, torch.float32).contiguous())
        return self._upsampler(conf.unsqueeze(3), xyz, xyz_upsampled).squeeze(-1)

    def _knn_advice(self) -> None:
        s = self.settings
        if s.n_points > 20000:
            if s.n_neighbors < 32:
                if s.knn != "kdtree":
                    logging.warning('For improved performance, it is recommended to use knn="kdtree" when '
                                    "N > 20000 and K < 32.")
            elif s.knn != "approximate":
                logging.warning('For improved performance, it is recommended to use knn="approximate" when '
                                "N > 20000 and K > 32.")
            if s.knn == "naive":
                logging.warning('Using knn="naive" for N > 20000 potentially has very low performance or '
                                "will reach an OOM!")
        elif s.knn != "naive":
            logging.warning('For improved performance, it is recommended to use knn="naive" when N < 20000.')

    def predict(self, xyz: np.ndarray, features: Optional[np.ndarray] = None,
                prepostprocess: bool = True) -> np.ndarray:
        """Class confidences (softmax) for one (N,3) or a batch (B,N,3) of clouds -> (C,N) / (B,C,N).
        With pre/post-processing the cloud is down-sampled to settings.n_points with the fixed seed 0
        and the confidences are carried back to every input point (model.py:146-235)."""
        self._knn_advice()          # the reference's messages; every choice runs the exact HIP search here
        assert xyz.shape[-1] == 3, "xyz should have shape (B) x N x 3!"
        batched = xyz.ndim != 2
        if not batched:
            xyz = xyz[None]
        if features is not None and features.ndim == 2:
            features = features[None]
        cloud = xyz
        if features is not None:
            assert xyz.shape[0] == features.shape[0], "xyz and features should have same batch size!"
            assert xyz.shape[1] == features.shape[1], "xyz and features should have same number of points!"
            cloud = np.concatenate((xyz, features), axis=-1)
        if self.settings.upsampling == "none":
            prepostprocess = False
        with torch.no_grad():
            full = torch.from_numpy(cloud.astype(np.float32))
            if prepostprocess:
                keep = sample_points(cloud.shape[1], self.settings.n_points, consistent=True)
                sampled = full[:, keep, :]
                logits = self._model(sampled.to(self._model.device))
                out = self.upsample(logits, sampled[:, :, :3], full[:, :, :3]).cpu().numpy()
            else:
                logits = self._model(full.to(self._model.device))
                # (the reference returns a device tensor in this branch, against its own annotation)
                if logits.is_cuda:
                    with torch.cuda.device(logits.device):
                        out = ops.softmax_cf(logits.contiguous()).cpu().numpy()
                else:
                    out = torch.softmax(logits, dim=1).numpy()
        return out if batched else out[0]

    # ---------------------------------------------------------------------------- training
    def _loader(self, dataset, n_points: int, batch_size: int, **kw):
        """get_data_loader (model.py:277-291, 326-332).  On a GPU the clouds live in HBM and every batch is assembled by
        one kernel (utils/device_dataset.py); the random numbers still come from numpy / torch in the reference's
        order unless RL_PIPELINE_RNG=device.  RL_HOST_PIPELINE=1 keeps the reference's host pipeline."""
        if self.device.type == "cuda" and not int(os.environ.get("RL_HOST_PIPELINE", "0")):
            return get_device_data_loader(dataset, n_points, batch_size, device=self.device,
                                          rng=os.environ.get("RL_PIPELINE_RNG", "numpy"), **kw)
        return get_data_loader(dataset, n_points, batch_size, **kw)

    def train(self, dataset_train: Sequence[Sample], dataset_validation: Sequence[Sample],
              training_settings: TrainingSettings = TrainingSettings(),
              augmentation_settings: AugmentationSettings = AugmentationSettings(),
              log_dir: Optional[Path] = None, class_names: Optional[List[str]] = None,
              callbacks: List[Callable[[int, Dict[str, float]], None]] = []):
        """Train from the current weights and keep the best ones (model.py:237-298)."""
        assert class_names is not None and len(class_names) == self.settings.n_classes, (
            "The length of given class names should correspond to the n_classes setting of the model")
        n, bs = self.settings.n_points, training_settings.batch_size
        train_loader = self._loader(dataset_train, n, bs, shuffle=True, consistent_sampling=False,
                                    augmentation_settings=augmentation_settings)
        val_loader = self._loader(dataset_validation, n, bs, shuffle=False, consistent_sampling=True)
        trainer = Trainer(train_loader, val_loader, log_dir, class_names)
        self._model = trainer.train(self._model, training_settings, callbacks=callbacks)

    def evaluate(self, dataset: Sequence[Sample], class_names: Optional[List[str]] = None, batch_size: int = 16,
                 loss_function: str = "dice", postprocess: bool = False, include_stdev: bool = False) -> Dict:
        loader = self._loader(dataset, self.settings.n_points, batch_size, shuffle=False, consistent_sampling=True)
        bag = Trainer.evaluate(self._model, loader, class_names, loss_function, postprocess)
        return bag.as_dict(include_stdev=include_stdev)
