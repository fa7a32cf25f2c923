"""Host-side mirror of the reference's model modules on the hot path:
same class names, constructor signatures and ``state_dict`` keys, so released checkpoints
(``dae_state_dict`` / ``vae_state_dict``, models/lion.py:30-35) load unchanged; every point-voxel
operator goes through ``lion_amd.functional`` (HIP).  ``import_model`` resolves the dotted class
paths that reference configs carry (``models.score_sde.resnet.PriorSEDrop`` ...) inside this package.
"""
import importlib


def import_model(path: str):
    """utils/model_helper.py:105-110 equivalent: 'models.a.b.Class' -> lion_amd.models.a.b.Class."""
    mod, cls = path.rsplit(".", 1)
    if mod.startswith("models."):
        mod = "lion_amd." + mod
    elif mod == "models":
        mod = "lion_amd.models"
    return getattr(importlib.import_module(mod), cls)
