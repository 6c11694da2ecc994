"""Host-side mirrors of the reference's integer / text code (no GPU): the PRODUCT package's adapters, text normaliser and
second-stage input packing against the golden vectors minted from the reference (tests/golden/adapters.json)."""
import json
import os

import pytest

from conftest import GOLDEN, load_pkg

adapters = load_pkg("adapters")
utils = load_pkg("utils")
inference = load_pkg("inference")


@pytest.fixture(scope="module")
def golden():
    return json.load(open(os.path.join(GOLDEN, "adapters.json")))


def test_adapters_match_reference(golden):
    n = {"flattened": 0, "tilted": 0}
    for c in golden["cases"]:
        if c["kind"] == "flattened":
            text, audio = adapters.FlattenedInterleavedEncodec2Codebook(end_of_audio_token=1024).decode([c["tokens"]])
        elif c["kind"] == "tilted":
            text, audio = adapters.TiltedEncodec(end_of_audio_token=1024).decode(c["tokens"])
        else:
            continue
        assert text == c["text"] and audio == c["audio"]
        n[c["kind"]] += 1
    assert n["flattened"] > 0 and n["tilted"] > 0


def test_normalize_text_matches_reference(golden):
    seen = 0
    for c in golden.get("normalize", []):
        if c.get("raises"):
            with pytest.raises(ValueError):
                utils.normalize_text(c["text"])
        else:
            assert utils.normalize_text(c["text"]) == c["out"]
        seen += 1
    assert seen > 0


def test_pack_second_stage_input():
    rows = inference.pack_second_stage_input([1030, 1031, 1537], [[1, 2, 3], [4, 5, 6]], ctx_window=10)
    assert rows[0] == [1030, 1031, 1537, 1, 2, 3, 1024, 1024, 1024, 1024]
    assert rows[1] == [1024, 1024, 1024, 4, 5, 6, 1024, 1024, 1024, 1024]
    rows = inference.pack_second_stage_input([1030, 1031, 1537], [[1, 2, 3], [4, 5, 6]], ctx_window=5)
    assert rows == [[1030, 1031, 1537, 1, 2], [1024, 1024, 1024, 4, 5]]  # truncation, fam/llm/inference.py:299-300


def test_tokeniser_round_trip_and_layout():
    tok_mod = load_pkg("tokenise")
    info = tok_mod.byte_level_tokenizer_info(offset=2049)
    tok = tok_mod.TrainedBPETokeniser(**info)
    ids = tok.encode("Hello there, this is a test.")
    assert ids[-1] == 2561 and min(ids) >= 2049 and max(ids) <= 2561  # 513 text ids after the audio ids + EOA
    assert tok.decode(ids[:-1]) == "Hello there, this is a test."
