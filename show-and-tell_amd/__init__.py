"""show-and-tell_amd: MI355X-native (gfx950) Show-and-Tell training hot path behind the reference's model API.

The directory name is not a Python identifier; import it with
    sat = importlib.import_module("show-and-tell_amd")
or through the repo-root `models.py` drop-in shim (`from models import EncoderCNN, DecoderRNN`).
"""
from . import _lib  # noqa: F401
from .models import (CaptionModel, Decoder, DecoderRNN, Encoder, EncoderCNN, ShowAndTell)  # noqa: F401
from .attend import ShowAttendTellModel, VggFeatures  # noqa: F401
from .input import DevicePrefetcher, collate_batch, collate_on_device  # noqa: F401
from .optim import FusedClampAdam  # noqa: F401
from .pack import PackInfo, pack_targets  # noqa: F401
from .resnet import RESNET152, conv_flops  # noqa: F401
from .trainer import (DataParallelStep, FlatParams, TrainStep, decode_shard, dp_shard, gather_decoded,  # noqa: F401
                      lr_for_epoch)
