from .callbacks import ItemEncoderMixin, ItemEncodingCallback
from .models import ModelType
from .recommender import RecModule

__all__ = ["ModelType", "RecModule", "ItemEncodingCallback", "ItemEncoderMixin"]
