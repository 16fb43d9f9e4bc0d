"""show-and-tell_amd: MI355X-native (gfx950) Show-and-Tell training hot path behind the reference's model API.

The directory name is not a Python identifier; import it with
    sat = importlib.import_module("show-and-tell_amd")
or through the repo-root `models.py` drop-in shim (`from models import EncoderCNN, DecoderRNN`).
"""
import os as _os

# HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The training step uses the main stream, two
# look-ahead streams (EncoderCNN.prefetch) and, with several GPUs, RCCL's stream: a fifth stream shares a queue with one of the
# others and serialises behind it (measured with RCCL present: 11.3 k -> 12.0 k img/s with 8 queues).  Only effective when set
# before the HIP runtime initialises, hence at import; an explicit setting by the user wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import _lib  # noqa: F401,E402
from . import watch  # noqa: F401,E402
from .models import (CaptionModel, Decoder, DecoderRNN, Encoder, EncoderCNN, ShowAndTell)  # noqa: F401
from .attend import ShowAttendTellModel, VggFeatures  # noqa: F401
from .input import DevicePrefetcher, collate_batch, collate_on_device  # noqa: F401
from .evaluate import kept_tokens, mean_cross_entropy, pack_validation_targets, sentences, validation_step  # noqa: F401
from .optim import FusedClampAdam  # noqa: F401
from .pack import PackInfo, pack_targets  # noqa: F401
from .resnet import RESNET152, conv_flops  # noqa: F401
from .trainer import (DataParallelStep, FlatParams, TrainStep, decode_shard, dp_shard, gather_decoded,  # noqa: F401
                      lr_for_epoch)
