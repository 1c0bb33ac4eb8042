"""vosk_tts_amd — MI355X-native VITS2 inference path behind the vosk_tts.Model / Synth API.

    from vosk_tts_amd import Model, Synth
    Synth(Model(model_path="...")).synth("прив+ет м+ир", "out.wav", speaker_id=2)
"""
from .model import Model  # noqa: F401
from .synth import Synth  # noqa: F401
