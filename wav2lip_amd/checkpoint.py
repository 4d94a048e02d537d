"""Checkpoint files in the reference's on-disk format (`torch.save` of a dict with "state_dict", "optimizer",
"global_step", "global_epoch"; wav2lip_train.py:294-305, :308-336; inference.py:160-179), including the `module.` prefix
that released weights trained under nn.DataParallel carry.  A file written by the reference loads here and vice versa:
the mirrored modules keep the reference's state-dict keys and `wav2lip_amd.optim.Adam` keeps torch.optim.Adam's state layout.
"""
import os

import torch

from .hparams import hparams


def _load(checkpoint_path, device=None):
    """wav2lip_train.py:308-314 / inference.py:160-166: tensors are mapped to the CPU unless a device is given"""
    return torch.load(checkpoint_path, map_location=device or "cpu", weights_only=False)


def strip_module_prefix(state_dict):
    """inference.py:172-176: the same unconditional `k.replace('module.', '')`"""
    return {k.replace('module.', ''): v for k, v in state_dict.items()}


def save_checkpoint(model, optimizer, step, checkpoint_dir, epoch, prefix=''):
    """wav2lip_train.py:294-305 (hq_wav2lip_train.py:330-342 adds `prefix`)"""
    checkpoint_path = os.path.join(checkpoint_dir, "{}checkpoint_step{:09d}.pth".format(prefix, step))
    optimizer_state = optimizer.state_dict() if (optimizer is not None and hparams.save_optimizer_state) else None
    torch.save({"state_dict": model.state_dict(), "optimizer": optimizer_state, "global_step": step,
                "global_epoch": epoch}, checkpoint_path)
    return checkpoint_path


def load_checkpoint(path, model, optimizer=None, reset_optimizer=False):
    """wav2lip_train.py:316-336; returns (model, global_step, global_epoch) instead of writing module globals (the training
    loops use trainer.load_checkpoint, which keeps the counters in a `Run` and has `overwrite_global_states`)"""
    checkpoint = _load(path)
    model.load_state_dict(strip_module_prefix(checkpoint["state_dict"]))
    if not reset_optimizer and optimizer is not None and checkpoint.get("optimizer") is not None:
        optimizer.load_state_dict(checkpoint["optimizer"])
    return model, checkpoint.get("global_step", 0), checkpoint.get("global_epoch", 0)


def load_model(path, device="cuda"):
    """inference.py:168-179"""
    from .models import Wav2Lip
    model = Wav2Lip()
    print("Load checkpoint from: {}".format(path))
    checkpoint = _load(path)
    model.load_state_dict(strip_module_prefix(checkpoint["state_dict"]))
    return model.to(device).eval()
