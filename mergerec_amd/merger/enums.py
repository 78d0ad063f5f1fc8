"""The three option enums of rec_retrieval/merger/enums.py:11-40.  Member names and values (value == name) are part of the drop-in
boundary: scripts select them by ``MergeType[flag.upper()]`` and weight files / configs spell them out."""
from enum import Enum

__all__ = ["MergeType", "LearnType", "LossType"]


def _named(cls_name: str, members: str) -> Enum:
    return Enum(cls_name, {m: m for m in members.split()}, module=__name__)


MergeType = _named("MergeType", "TASK_VECTOR TIES PCB LOCALIZE_AND_STITCH")
LearnType = _named("LearnType", "TASK_WISE LAYER_WISE")
LossType = _named(
    "LossType",
    "CE KD MSE ADAMERGING ADAMERGING_KD MERGED_PSEUDO_LABEL SINGLE_PSEUDO_LABEL MERGED_PSEUDO_LABEL_KD SINGLE_PSEUDO_LABEL_KD",
)
