"""Drop-in for the reference's `models.py` (`/root/reference/models.py`): `from models import EncoderCNN, DecoderRNN`
resolves to the MI355X-native implementations in `show-and-tell_amd/models.py`."""
import importlib as _il
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.abspath(__file__))
if _root not in _sys.path:
    _sys.path.insert(0, _root)
_m = _il.import_module("show-and-tell_amd.models")
EncoderCNN, DecoderRNN, ShowAndTell = _m.EncoderCNN, _m.DecoderRNN, _m.ShowAndTell
Encoder, Decoder, CaptionModel = _m.Encoder, _m.Decoder, _m.CaptionModel
ShowAttendTellModel = _il.import_module("show-and-tell_amd.attend").ShowAttendTellModel      # model2.py:9 (what train.py:37 builds)
