"""Deterministic stand-in tokenizer used ONLY with synthetic weights (no tokenizer files are reachable offline).

It follows the call contract `FastVLMBackbone._prep_text` relies on (reference model/fastvlm_adapter.py:361-380):
`tok(list[str], padding="longest"|"max_length", truncation=True, max_length=N, return_tensors="pt")` ->
{"input_ids", "attention_mask"} with `padding_side` honoured.  Ids are UTF-8 bytes offset into the vocabulary, so the
same prompt always maps to the same ids; it makes no claim of matching the Qwen2 BPE vocabulary.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch


class SyntheticTokenizer:
    def __init__(self, vocab_size: int, pad_token_id: int = 0, padding_side: str = "right"):
        self.vocab_size = int(vocab_size)
        self.pad_token_id = int(pad_token_id)
        self.padding_side = padding_side

    def encode(self, text: str) -> List[int]:
        span = max(self.vocab_size - 1, 1)
        return [1 + (b * 2654435761 % span) for b in text.encode("utf-8")] or [1]

    def __call__(self, texts: Sequence[str], padding="longest", truncation=True, max_length: int = 64,
                 return_tensors: str = "pt") -> Dict[str, torch.Tensor]:
        if isinstance(texts, str):
            texts = [texts]
        rows = [self.encode(t) for t in texts]
        if truncation:
            rows = [r[:max_length] for r in rows]
        width = max_length if padding == "max_length" else max(len(r) for r in rows)
        ids = torch.full((len(rows), width), self.pad_token_id, dtype=torch.long)
        mask = torch.zeros(len(rows), width, dtype=torch.long)
        for i, r in enumerate(rows):
            if self.padding_side == "left":
                ids[i, width - len(r):] = torch.tensor(r)
                mask[i, width - len(r):] = 1
            else:
                ids[i, : len(r)] = torch.tensor(r)
                mask[i, : len(r)] = 1
        return {"input_ids": ids, "attention_mask": mask}
