"""mfcnet_amd -- MI355X-native MFCNet multi-frame forward/backward hot path.

Drop-in for the reference's `models.get_multiframe_segmentation_model` on the
`HRNetMulti-Large` / `HRNetMulti-Basic` model types (models/__init__.py:79-84).
"""
from ._lib import BF16, F16, F32, MfcError, lib  # noqa: F401  (importing loads libmfcnet_hip.so or raises)
from .model import (HighResolutionNetHIP, HRNetMultiBasic, HRNetMultiLarge, get_multiframe_segmentation_model,  # noqa: F401
                    get_tooltip_segmentation_model)
from .resunet import ResUnet_VB  # noqa: F401
from .optim import FlatAdam  # noqa: F401
from .loss import get_loss, mfc_loss  # noqa: F401
from .metrics import confusion_counts, get_metrics  # noqa: F401
from .checkpoint import load_base_model_weights, load_model_weights, save_model  # noqa: F401
from .engine import LossScaler, eval_step, train_step  # noqa: F401
from .dist import DataParallel, ShardedStep  # noqa: F401
