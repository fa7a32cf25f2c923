"""Checkpoint files in the reference's two on-disk formats, so that its released ``lion_ckpt`` weights load
unchanged and a run started there can be resumed here.

* VAE trainer   (trainers/base_trainer.py:90-108 save, :122-146 resume):
  ``{'opt', 'model', 'epoch', 'step'[, 'grad_scalar']}``; older files call the weights ``'model_state'`` and may
  carry DataParallel prefixes (``module.`` / ``model.module.``, ``filter_name`` :110-120).
* prior trainer (trainers/train_prior.py:328-350 save, :294-326 resume):
  ``{'epoch' (= last finished epoch + 1), 'global_step', 'grad_scalar', 'dae_state_dict', 'dae_optimizer',
  'dae_scheduler', 'vae_state_dict', 'vae_optimizer', 'vae_scheduler'}``; ``models/lion.py:30-35`` reads
  ``dae_state_dict`` into ``ModuleList([global_prior, local_prior])`` and ``vae_state_dict`` into the VAE.

Files land in ``<save_dir>/checkpoints/epoch_<e>_iters_<s>.pt`` unless a name is given.  Only plain
``state_dict``s are stored; parameter names and shapes of the modules in ``lion_amd.models`` equal the reference's
(tests/golden/state_dict_layouts.json), which is what makes the files interchangeable.
"""
import os

import torch

_WRAPPER_PREFIXES = ("model.module.", "module.")


def checkpoint_path(save_dir, epoch, step, save_name=None):
    name = save_name if save_name is not None else "epoch_%s_iters_%s.pt" % (epoch, step)
    return os.path.join(save_dir, "checkpoints", name)


def strip_wrapper_prefixes(state):
    """Drop the DataParallel / trainer wrappers from parameter names (one prefix per key, longest first)."""
    out = {}
    for key, value in state.items():
        for prefix in _WRAPPER_PREFIXES:
            if key.startswith(prefix):
                key = key[len(prefix):]
                break
        out[key] = value
    return out


def _write(content, path):
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    # write-then-rename: a job killed mid-save must not leave a truncated file under the final name
    tmp = path + ".tmp"
    torch.save(content, tmp)
    os.replace(tmp, path)
    return path


def _opt(obj):
    return obj.state_dict() if obj is not None else None


def save_vae_checkpoint(save_dir, model, optimizer, epoch, step, grad_scalar=None, appendix=None, save_name=None):
    content = {"opt": optimizer.state_dict(), "model": model.state_dict(), "epoch": epoch, "step": step}
    if appendix:
        content.update(appendix)
    if grad_scalar is not None:
        content["grad_scalar"] = grad_scalar.state_dict()
    return _write(content, checkpoint_path(save_dir, epoch, step, save_name))


def load_vae_checkpoint(path, model, optimizer=None, grad_scalar=None, strict=True, map_location="cpu"):
    """-> (start_epoch, step).  ``optimizer`` / ``grad_scalar`` are restored when given; a grad scaler that the file
    has no state for is an error, as in the reference."""
    ckpt = torch.load(path, map_location=map_location)
    weights = ckpt["model"] if "model" in ckpt else ckpt["model_state"]
    model.load_state_dict(strip_wrapper_prefixes(weights), strict=strict)
    if optimizer is not None and "opt" in ckpt:
        optimizer.load_state_dict(ckpt["opt"])
    if grad_scalar is not None:
        if "grad_scalar" not in ckpt:
            raise KeyError("checkpoint holds no 'grad_scalar' state; train without a grad scaler to resume it")
        grad_scalar.load_state_dict(ckpt["grad_scalar"])
    return ckpt["epoch"], ckpt.get("step", 0)


def save_prior_checkpoint(save_dir, dae, vae, epoch, step, dae_optimizer=None, vae_optimizer=None,
                          dae_scheduler=None, vae_scheduler=None, grad_scalar=None, appendix=None, save_name=None):
    content = {"epoch": epoch + 1, "global_step": step, "grad_scalar": _opt(grad_scalar),
               "dae_state_dict": dae.state_dict(), "dae_optimizer": _opt(dae_optimizer),
               "dae_scheduler": _opt(dae_scheduler), "vae_state_dict": vae.state_dict(),
               "vae_optimizer": _opt(vae_optimizer), "vae_scheduler": _opt(vae_scheduler)}
    if appendix:
        content.update(appendix)
    return _write(content, checkpoint_path(save_dir, epoch, step, save_name))


def load_prior_checkpoint(path, dae, vae, dae_optimizer=None, vae_optimizer=None, dae_scheduler=None,
                          vae_scheduler=None, grad_scalar=None, map_location="cpu"):
    """-> (start_epoch, global_step).  Weights are always restored; each optimizer / scheduler / scaler passed in is
    restored from its entry (sampling-only callers pass none, like models/lion.py:30-35)."""
    ckpt = torch.load(path, map_location=map_location)
    dae.load_state_dict(ckpt["dae_state_dict"])
    vae.load_state_dict(ckpt["vae_state_dict"])
    for obj, key in ((dae_optimizer, "dae_optimizer"), (dae_scheduler, "dae_scheduler"),
                     (vae_optimizer, "vae_optimizer"), (vae_scheduler, "vae_scheduler"),
                     (grad_scalar, "grad_scalar")):
        if obj is None:
            continue
        if ckpt.get(key) is None:
            raise KeyError(f"checkpoint holds no '{key}' state")
        obj.load_state_dict(ckpt[key])
    return ckpt["epoch"], ckpt["global_step"]


def load_pretrained_vae(path, vae, map_location="cpu"):
    """``cfg.sde.vae_checkpoint`` of the prior trainer (train_prior.py:246-253): the VAE trainer's file, weights only."""
    vae.load_state_dict(torch.load(path, map_location=map_location)["model"])
    return vae
