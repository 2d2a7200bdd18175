"""Reference-format checkpoints (/root/reference/project/models/trainers/base.py:677-753): ``checkpoint_{step:05d}.pth`` /
``checkpoint_final.pth`` = ``torch.save({"models": {class_name: module.state_dict()}, "step": int})`` -- model tensors only (the
reference never restores optimiser / scheduler state: it raises NotImplementedError at :703-705).  These helpers read and write that
container for a dict of modules whose parameter names mirror the reference's (``modules.*AffineTransform``, ``envlight.EnvLight``, a
``VanillaGaussians``-shaped module), so that a scene trained with either implementation can be evaluated with the other."""
from __future__ import annotations

import os
from typing import Dict

import torch


def state_dict(models: Dict[str, torch.nn.Module], step: int) -> dict:
    """BasicTrainer.state_dict(only_model=True) (base.py:677-682)."""
    return {"models": {k: v.state_dict() for k, v in models.items()}, "step": int(step)}


def save_checkpoint(models: Dict[str, torch.nn.Module], step: int, log_dir: str, is_final: bool = False) -> str:
    """BasicTrainer.save_checkpoint (base.py:739-753): same file names."""
    path = os.path.join(log_dir, "checkpoint_final.pth" if is_final else f"checkpoint_{int(step):05d}.pth")
    torch.save(state_dict(models, step), path)
    return path


def load_checkpoint(path_or_dict, models: Dict[str, torch.nn.Module], strict: bool = True, map_location=None) -> int:
    """BasicTrainer.resume_from_checkpoint / load_state_dict (base.py:690-737): loads every class present in both, sets ``.step`` on
    the modules, returns the step.  Classes of ``models`` missing from the file are skipped with the reference's warning semantics
    (left untouched); ``strict`` is passed to each module's ``load_state_dict`` as the reference does."""
    sd = torch.load(path_or_dict, map_location=map_location) if isinstance(path_or_dict, (str, os.PathLike)) else dict(path_or_dict)
    step = sd.pop("step")
    sd.pop("optimizer", None)
    per_model = sd.pop("models")
    for name, module in models.items():
        module.step = step
        if name not in per_model:
            continue
        module.load_state_dict(per_model[name], strict=strict)
    return int(step)
