"""MI355X-native drop-in for the `randlanet` package of matthiasverstraete/3d_recognizer
(reference randlanet/__init__.py:1-11).  Put the directory that contains this package
(3d_recognizer_amd/) on PYTHONPATH and the reference's train.py / predict.py import it unchanged.
Importing it creates no GPU context (train.py spawns its worker, train.py:108-115).
"""
from .model import Model
from .utils.augmentation import AugmentationSettings
from .utils.modules import RandLANetSettings
from .utils.trainer import TrainingSettings

__all__ = ["AugmentationSettings", "Model", "RandLANetSettings", "TrainingSettings"]
