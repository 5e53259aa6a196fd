"""Dataset glue (reference: src/vla_fastvlm/data/aloha_dataset.py).  Only the batch schema of `aloha_collate_fn`
(:205-222) belongs to the hot path's boundary; the HF `datasets` wrappers need network access and are thin."""
from __future__ import annotations

from typing import Any, Callable, Dict, List, Optional

import torch
from torch.utils.data import DataLoader, Dataset, IterableDataset


def default_aloha_transforms(sample: Dict[str, Any]) -> Dict[str, Any]:
    """uint8 [0,255] HWC/CHW image -> float [0,1] CHW (reference :26-37)."""
    img = torch.as_tensor(sample["image"])
    if img.ndim == 3 and img.shape[-1] in (1, 3) and img.shape[0] not in (1, 3):
        img = img.permute(2, 0, 1)
    if img.dtype == torch.uint8:
        img = img.float() / 255.0
    out = dict(sample)
    out["image"] = img.float()
    return out


def _resolve_task(sample: Dict[str, Any]) -> str:
    for key in ("task", "language_instruction", "instruction"):
        if key in sample and sample[key] is not None:
            val = sample[key]
            return val if isinstance(val, str) else str(val)
    return ""


class SyntheticAlohaDataset(Dataset):
    """Seeded random samples with the ALOHA shapes (336x336 RGB, 14-d state/action): what the benchmarks train on."""

    def __init__(self, length: int = 1024, image_size: int = 336, state_dim: int = 14, action_dim: int = 14, seed: int = 0,
                 task: str = "insert the peg into the socket") -> None:
        self.length, self.image_size, self.ds, self.da, self.seed, self.task = length, image_size, state_dim, action_dim, seed, task

    def __len__(self) -> int:
        return self.length

    def __getitem__(self, i: int) -> Dict[str, Any]:
        g = torch.Generator().manual_seed(self.seed * 1_000_003 + i)
        return {"image": torch.rand(3, self.image_size, self.image_size, generator=g), "state": torch.randn(self.ds, generator=g),
                "action": torch.randn(self.da, generator=g), "task": self.task, "index": i}


class AlohaDataset(Dataset):
    def __init__(self, split: str = "train", repo_id: str = "lerobot/aloha_sim_insertion_human_image",
                 transform: Optional[Callable] = default_aloha_transforms, limit_samples: Optional[int] = None) -> None:
        from datasets import load_dataset
        self.ds = load_dataset(repo_id, split=split)
        if limit_samples:
            self.ds = self.ds.select(range(min(limit_samples, len(self.ds))))
        self.transform = transform

    def __len__(self) -> int:
        return len(self.ds)

    def __getitem__(self, i: int) -> Dict[str, Any]:
        raw = self.ds[i]
        sample = {"image": raw.get("observation.images.top", raw.get("image")), "state": raw.get("observation.state", raw.get("state")),
                  "action": raw["action"], "task": _resolve_task(raw), "index": i}
        return self.transform(sample) if self.transform else sample


class AlohaIterableDataset(IterableDataset):
    def __init__(self, split: str = "train", repo_id: str = "lerobot/aloha_sim_insertion_human_image",
                 transform: Optional[Callable] = default_aloha_transforms) -> None:
        from datasets import load_dataset
        self.ds = load_dataset(repo_id, split=split, streaming=True)
        self.transform = transform

    def __iter__(self):
        for i, raw in enumerate(self.ds):
            sample = {"image": raw.get("observation.images.top", raw.get("image")), "state": raw.get("observation.state", raw.get("state")),
                      "action": raw["action"], "task": _resolve_task(raw), "index": i}
            yield self.transform(sample) if self.transform else sample


def aloha_collate_fn(samples: List[Dict[str, Any]]) -> Dict[str, Any]:
    return {"images": torch.stack([torch.as_tensor(s["image"]) for s in samples]),
            "states": torch.stack([torch.as_tensor(s["state"], dtype=torch.float32) for s in samples]),
            "actions": torch.stack([torch.as_tensor(s["action"], dtype=torch.float32) for s in samples]),
            "tasks": [s.get("task", "") for s in samples],
            "metadata": [{k: v for k, v in s.items() if k not in ("image", "state", "action", "task")} for s in samples]}


def create_aloha_dataloader(dataset, batch_size: int = 4, shuffle: bool = True, num_workers: int = 4) -> DataLoader:
    return DataLoader(dataset, batch_size=batch_size, shuffle=shuffle and not isinstance(dataset, IterableDataset),
                      num_workers=num_workers, collate_fn=aloha_collate_fn, pin_memory=True, drop_last=False)
