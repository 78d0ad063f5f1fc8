"""The two small config types the fine-tuning path shares between the datamodule and the module
(rec_retrieval/types/enums.py:4-8 NegativeSampleOption; rec_retrieval/configs/finetune.py:8-24 NegativeSampleConfig).
The reference's tyro CLI dataclasses are replaced by argparse in the entry scripts."""
from __future__ import annotations

from enum import Enum
from typing import Optional

NegativeSampleOption = Enum("NegativeSampleOption", {n: n for n in ("FULL", "IN_BATCH", "SAMPLE", "IN_BATCH_SAMPLE")})


class NegativeSampleConfig:
    """k sampled negatives per sequence (None: none) and / or the other targets of the batch as negatives; the mode follows."""

    def __init__(self, k: Optional[int] = None, in_batch: bool = False):
        self.k, self.in_batch = k, bool(in_batch)
        table = {(False, False): NegativeSampleOption.FULL, (True, False): NegativeSampleOption.SAMPLE,
                 (False, True): NegativeSampleOption.IN_BATCH, (True, True): NegativeSampleOption.IN_BATCH_SAMPLE}
        self.mode = table[(k is not None, self.in_batch)]

    def __repr__(self):
        return f"NegativeSampleConfig(k={self.k}, in_batch={self.in_batch}, mode={self.mode.name})"
