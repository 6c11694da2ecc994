"""Mirror of `fam/quantiser/audio/speaker_encoder/` (`model.SpeakerEncoder`, `audio.wav_to_mel_spectrogram`)."""
from .model import SpeakerEncoder  # noqa: F401
