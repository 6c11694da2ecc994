"""MI355X-native MetaVoice-1B synthesis engine: host-side mirror of the reference's `fam.llm` fast-inference API on top of
libmetavoice_hip.so (hand-written gfx950 HIP kernels, C ABI in include/metavoice_hip.h).

The directory name is hyphenated (`metavoice-src_amd`), so import it with
    importlib.import_module("metavoice-src_amd")
Nothing here imports `oracle/` and nothing falls back to CPU/eager PyTorch: the first use of an engine raises if the
HIP library has not been built.
"""
from . import synthetic  # noqa: F401
from ._native import LIB_PATH, NativeLibraryMissing  # noqa: F401
from .engine import SamplingParams, Stage1Engine  # noqa: F401
from .fast_model import ModelArgs, Transformer, find_multiple, transformer_configs  # noqa: F401
from .model import GPT, GPTConfig  # noqa: F401
