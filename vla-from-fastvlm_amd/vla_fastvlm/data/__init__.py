"""Batch schema + loaders (reference: src/vla_fastvlm/data/__init__.py).  Dataset I/O is out of scope for the HIP path;
the schema of `aloha_collate_fn` ({images, states, actions, tasks, metadata}) is the boundary and is kept."""
from .aloha_dataset import (AlohaDataset, AlohaIterableDataset, SyntheticAlohaDataset, aloha_collate_fn,
                            create_aloha_dataloader, default_aloha_transforms)

__all__ = ["AlohaDataset", "AlohaIterableDataset", "SyntheticAlohaDataset", "aloha_collate_fn", "create_aloha_dataloader",
           "default_aloha_transforms"]
