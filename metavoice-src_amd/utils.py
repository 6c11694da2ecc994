"""Mirror of the inference-side helpers of `fam/llm/utils.py` (host-side, not on the GPU path)."""
from __future__ import annotations

import hashlib
import json
import re

import torch

# fam/llm/utils.py:13-42: unicode punctuation folded to Latin-1 look-alikes
_FOLD = {
    8175: "'", 8189: "'", 8190: "'", 8208: "-", 8209: "-", 8210: "-", 8211: "-", 8212: "-", 8213: "-", 8214: "||",
    8216: "'", 8217: "'", 8218: ",", 8219: "`", 8220: '"', 8221: '"', 8222: ",,", 8223: '"', 8228: ".", 8229: "..",
    8230: "...", 8242: "'", 8243: '"', 8245: "'", 8246: '"', 180: "'", 2122: "TM",
}


def normalize_text(text: str) -> str:
    """`fam/llm/utils.py:12-52`: fold unicode punctuation, reject anything outside Latin-1 (ValueError), turn tabs, newlines
    and `*` into spaces, strip, squeeze runs of whitespace."""
    text = text.translate(_FOLD)
    offenders = sorted({(ch, ord(ch)) for ch in text if ord(ch) >= 256})
    if offenders:
        raise ValueError(f"Non-supported character found: {offenders}")
    for ch in "\t\n\r*":
        text = text.replace(ch, " ")
    return re.sub(r"\s\s+", " ", text.strip())


def check_audio_file(path_or_uri, threshold_s: int = 30):
    """`fam/llm/utils.py:55-74`: the speaker reference must be at least `threshold_s` long (the reference measures it with
    librosa; here with the .wav reader of speaker_encoder/audio.py)."""
    from .speaker_encoder import audio

    wav = audio.load_audio(str(path_or_uri), sr=audio.sampling_rate)
    duration_s = len(wav) / float(audio.sampling_rate)
    if duration_s < threshold_s:
        raise Exception(f"The audio file is too short. Please provide an audio file that is at least {threshold_s} seconds long "
                        f"to proceed. (Provided file is {duration_s:.2f}s)")


def get_default_dtype() -> str:
    """`fam/llm/utils.py:77-86`.  MI355X (gfx950) reports compute capability major 9 -> bfloat16."""
    if torch.cuda.is_available():
        return "float16" if torch.cuda.get_device_properties(0).major <= 7 else "bfloat16"
    return "float16"


def get_device() -> str:
    """`fam/llm/utils.py:89-91`."""
    return "cuda" if torch.cuda.is_available() else "cpu"


def hash_dictionary(d: dict) -> str:
    """`fam/llm/utils.py:94-103`."""
    return hashlib.sha256(json.dumps(d, sort_keys=True).encode("utf-8")).hexdigest()
