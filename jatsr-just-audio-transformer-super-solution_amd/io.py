"""Data formats either side of the sampling path (reference infer_test_v3m2.py / prepare_dataset_v5.py).

* latent files: `torch.save({'hr_latent': fp16 [1024, T], 'lr_latent': fp16 [1024, T], 'metadata': ...})`
  (prepare_dataset_v5.py:255-264), read with `torch.load(..., mmap=True)` and `.float()` (infer_test_v3m2.py:292-294);
* normalisation statistics (infer_test_v3m2.py:300-332): JSON with hr_mean / hr_std / lr_mean / lr_std lists
  (recalculate_stats.py:113-121), or a `.pt` dict with those keys, or running sums {sum, sq_sum, count} over the
  2048 concatenated channels;
* checkpoints: `jatsr_amd.model.load_model` (infer_test_v3m2.py:33-94).

Pure host code (PyTorch only as the container / file reader); nothing here computes on the hot path.
"""
from __future__ import annotations

import json
import os

import torch


def load_latent_file(path, mmap=True):
    """-> (hr_latent or None, lr_latent) as fp32 [C, T] CPU tensors (infer_test_v3m2.py:292-294)."""
    data = torch.load(path, map_location="cpu", mmap=mmap, weights_only=False)
    if "lr_latent" not in data:
        raise KeyError(f"{path}: no 'lr_latent' (keys: {list(data.keys())})")
    hr = data["hr_latent"].float() if "hr_latent" in data else None
    return hr, data["lr_latent"].float()


def save_latent_file(path, hr_latent=None, lr_latent=None, metadata=None, **extra):
    """Write the reference's latent container (fp16 tensors, prepare_dataset_v5.py:255-264)."""
    out = dict(extra)
    if hr_latent is not None:
        out["hr_latent"] = hr_latent.detach().to("cpu", torch.float16)
    if lr_latent is not None:
        out["lr_latent"] = lr_latent.detach().to("cpu", torch.float16)
    out["metadata"] = metadata or {}
    torch.save(out, path)


def load_stats(path, channels=1024, device="cpu"):
    """Per-channel normalisation statistics -> dict(hr_mean, hr_std, lr_mean, lr_std), each fp32 [channels].

    Accepts the three formats infer_test_v3m2.py:300-332 accepts; raises ValueError on anything else."""
    if str(path).endswith(".json"):
        with open(path, "r") as f:
            raw = json.load(f)
        stats = {k: torch.tensor(raw[k], dtype=torch.float32) for k in ("hr_mean", "hr_std", "lr_mean", "lr_std")}
    else:
        raw = torch.load(path, map_location="cpu", weights_only=False)
        if "hr_mean" in raw:
            stats = {k: torch.as_tensor(raw[k]).float() for k in ("hr_mean", "hr_std", "lr_mean", "lr_std")}
        elif "sum" in raw:
            count = raw["count"]
            mean = torch.as_tensor(raw["sum"]).double() / count
            var = torch.as_tensor(raw["sq_sum"]).double() / count - mean ** 2
            std = torch.sqrt(var + 1e-8)
            # first `channels` entries are HR, the rest LR (infer_test_v3m2.py:322-326)
            stats = {"hr_mean": mean[:channels].float(), "hr_std": std[:channels].float(),
                     "lr_mean": mean[channels:].float(), "lr_std": std[channels:].float()}
        else:
            raise ValueError(f"Unknown stats format. Keys: {list(raw.keys())}")
    for k, v in stats.items():
        v = v.reshape(-1)
        if v.numel() != channels:
            raise ValueError(f"{path}: {k} has {v.numel()} entries, expected {channels}")
        stats[k] = v.to(device)
    return stats


def frames_for_seconds(seconds, sample_rate=44100, hop=512):
    """DAC 44.1 kHz latent frames for a duration (infer_test_v3m2.py:340-350): 16 s -> 1378, 2 s -> 172."""
    return int(seconds * sample_rate / hop)


def first_latent_file(val_dir):
    """Default input: the first `*.pt` of the validation directory (infer_test_v3m2.py:283-289)."""
    files = sorted(f for f in os.listdir(val_dir) if f.endswith(".pt"))
    if not files:
        raise FileNotFoundError(f"No files found in {val_dir}")
    return os.path.join(val_dir, files[0])
