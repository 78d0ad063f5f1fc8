from .enums import LearnType, LossType, MergeType
from .merger import ModelMerger
from .weight_learning import (
    TaskVectorMergingModuleBase,
    TaskVectorMergingModuleLayerWise,
    TaskVectorMergingModuleTaskWise,
    load_merging_module,
)

__all__ = [
    "MergeType", "LearnType", "LossType", "ModelMerger", "load_merging_module", "TaskVectorMergingModuleBase",
    "TaskVectorMergingModuleTaskWise", "TaskVectorMergingModuleLayerWise",
]
