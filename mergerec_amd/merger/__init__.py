from .enums import LearnType, LossType, MergeType
from .weight_learning import (
    TaskVectorMergingModuleBase,
    TaskVectorMergingModuleLayerWise,
    TaskVectorMergingModuleTaskWise,
    load_merging_module,
)

__all__ = [
    "MergeType", "LearnType", "LossType", "load_merging_module", "TaskVectorMergingModuleBase",
    "TaskVectorMergingModuleTaskWise", "TaskVectorMergingModuleLayerWise",
]
