"""Checkpoint I/O in the reference's format (utils/other_tools.py:757-790): a torch file holding
``{'model_state': state_dict}`` whose keys may carry nn.DataParallel's ``module.`` prefix."""
from __future__ import annotations

import torch


def strip_module_prefix(sd: dict) -> dict:
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}


def load_checkpoints(model, save_path, load_name="model"):
    """Same call signature as the reference's loader.  Accepts prefixed and unprefixed keys."""
    states = torch.load(save_path, map_location="cpu")
    sd = states["model_state"] if isinstance(states, dict) and "model_state" in states else states
    # the reference's drivers pass an nn.DataParallel / DDP wrapper (test.py:87,208; train.py:94,282): load into the wrapped
    # module, whose keys carry no "module." prefix
    target = model
    while isinstance(target, (torch.nn.DataParallel, torch.nn.parallel.DistributedDataParallel)):
        target = target.module
    missing, unexpected = target.load_state_dict(strip_module_prefix(sd), strict=False)
    missing = [k for k in missing if not k.endswith("num_batches_tracked")]
    if missing or unexpected:
        raise KeyError(f"checkpoint/model key mismatch: missing {missing[:5]}..., unexpected {unexpected[:5]}...")
    return model


def save_checkpoints(save_path, model, opt=None, epoch=None, lrs=None):
    """utils/other_tools.py:757-769: the model state, plus `epoch + 1` and the optimizer's (and the scheduler's) state when they are given -
    the reference's trainer passes none of them (train.py:283-286)."""
    states = {"model_state": model.state_dict()}
    if lrs is not None:
        states.update(epoch=epoch + 1, opt_state=opt.state_dict(), lrs=lrs.state_dict())
    elif opt is not None:
        states.update(epoch=epoch + 1, opt_state=opt.state_dict())
    torch.save(states, save_path)
