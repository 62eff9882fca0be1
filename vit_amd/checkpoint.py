"""Checkpoint files in the reference's own dictionary layouts, so that files written by either side load in
the other:

* stage-1 pre-training (stage1_surgvu_pretraining.py:498-523): `checkpoint_epoch_XX.pth` with optimizer and
  scheduler state, and `best_encoder_surgvu.pth` without;
* V-JEPA2 fine-tuning (stage1_vjepa2_finetuning.py:371-390): adds `full_model_state_dict` and `config`;
* pretrained-encoder import (load_vjepa2_pretrained.py:292-297): `model_state_dict`, `encoder_state_dict`,
  `vjepa2_config`, `metadata`; read back by stage1_vjepa2_finetuning.py:84-128 with `strict=False`.

Tensors are written on the CPU.  Files are read with `weights_only=True` (nothing in these layouts needs
unpickling of code).
"""
from __future__ import annotations

from pathlib import Path
from typing import Any, Dict, Optional

import torch

from .surgical_world_model import SurgicalActionConditionedWorldModel

VJEPA2_DEFAULTS = {"embed_dim": 1024, "depth": 24, "num_heads": 16}      # stage1_vjepa2_finetuning.py:89-92


def _cpu(sd: Optional[Dict[str, Any]]):
    if sd is None:
        return None
    return {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in sd.items()}


def _heads(heads) -> Dict[str, Any]:
    return {
        "temporal_classifier": _cpu(heads.temporal_classifier.state_dict()),
        "contrastive_proj": _cpu(heads.contrastive_proj.state_dict()),
        "denoise_head": _cpu(heads.denoise_head.state_dict()) if heads.denoise_head is not None else None,
    }


def save_stage1_checkpoint(path, epoch: int, model, heads, loss: float, optimizer=None, scheduler=None) -> Dict[str, Any]:
    """With optimizer/scheduler: the per-epoch file; without: the best-encoder file (s1:498-523)."""
    ckpt = {"epoch": epoch, "encoder_state_dict": _cpu(model.encoder.state_dict()), **_heads(heads)}
    if optimizer is not None:
        ckpt["optimizer"] = optimizer.state_dict()
        ckpt["scheduler"] = scheduler.state_dict() if scheduler is not None else None
    ckpt["loss"] = float(loss)
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    torch.save(ckpt, path)
    return ckpt


def save_finetune_checkpoint(path, epoch: int, model, heads, loss: float, optimizer, scheduler,
                             encoder_frozen: bool) -> Dict[str, Any]:
    """ft:371-390."""
    ckpt = {"epoch": epoch, "encoder_state_dict": _cpu(model.encoder.state_dict()),
            "full_model_state_dict": _cpu(model.state_dict()), **_heads(heads),
            "optimizer": optimizer.state_dict(), "scheduler": scheduler.state_dict() if scheduler is not None else None,
            "loss": float(loss),
            "config": {"encoder_frozen": encoder_frozen, "encoder_dim": model.encoder_dim, "pretrained_source": "V-JEPA 2"}}
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    torch.save(ckpt, path)
    return ckpt


def save_pretrained_model(model, path, vjepa2_config: Dict[str, Any], metadata: Optional[Dict[str, Any]] = None):
    """lp:292-297."""
    ckpt = {"model_state_dict": _cpu(model.state_dict()), "encoder_state_dict": _cpu(model.encoder.state_dict()),
            "vjepa2_config": dict(vjepa2_config), "metadata": metadata or {}}
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    torch.save(ckpt, path)
    return ckpt


def load_checkpoint(path) -> Dict[str, Any]:
    return torch.load(path, map_location="cpu", weights_only=True)


def load_pretrained_model(path, img_size: int = 224, num_frames: int = 16, predictor_dim: int = 1024,
                          predictor_depth: int = 12, predictor_heads: int = 8, device="cuda"):
    """stage1_vjepa2_finetuning.py:57-138, same signature and return value: build the model from the file's
    `vjepa2_config` (defaults 1024/24/16) and load its weights with strict=False -- the full model when
    `model_state_dict` is present, else the encoder alone.  Returns (model, config) with config =
    {encoder_dim, encoder_depth, encoder_heads}; what `load_state_dict` reported (missing / unexpected keys) is kept on
    the model as `model.load_result`."""
    ckpt = load_checkpoint(path)
    cfg = {**VJEPA2_DEFAULTS, **(ckpt.get("vjepa2_config") or {})}
    model = SurgicalActionConditionedWorldModel(
        img_size=img_size, num_frames=num_frames, tubelet_size=(2, 16, 16), in_channels=3,
        encoder_dim=cfg["embed_dim"], encoder_depth=cfg["depth"], encoder_heads=cfg["num_heads"],
        predictor_dim=predictor_dim, predictor_depth=predictor_depth, predictor_heads=predictor_heads,
        action_dim=7, state_dim=7, num_anomaly_classes=2, use_grad_checkpoint=True)
    if "model_state_dict" in ckpt:
        result = model.load_state_dict(ckpt["model_state_dict"], strict=False)
    elif "encoder_state_dict" in ckpt:
        result = model.encoder.load_state_dict(ckpt["encoder_state_dict"], strict=False)
    else:
        raise ValueError("Checkpoint does not contain model weights")
    model = model.to(device)
    model.load_result = result
    return model, {"encoder_dim": cfg["embed_dim"], "encoder_depth": cfg["depth"], "encoder_heads": cfg["num_heads"]}
