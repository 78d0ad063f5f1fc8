from .callbacks import ItemEncoderMixin, ItemEncodingCallback, MultiDatasetItemEncodingCallback, SaveWeightsCallback
from .distiller import DistillSequenceModule, teacher_scores
from .loss_fn import distill_loss_factory
from .models import ModelType
from .recommender import RecModule

__all__ = ["ModelType", "RecModule", "ItemEncodingCallback", "ItemEncoderMixin", "SaveWeightsCallback", "MultiDatasetItemEncodingCallback", "DistillSequenceModule",
           "teacher_scores", "distill_loss_factory"]
