"""Host-side mirrors of the reference's integer / text code (no GPU): the PRODUCT package's adapters, text normaliser and
second-stage input packing against the golden vectors minted from the reference (tests/golden/adapters.json)."""
import json
import os

import pytest

from conftest import GOLDEN, ROOT, load_pkg

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


def test_loudness_normalisation():
    """loudness.py (BS.1770-4 as torchaudio / audiocraft evaluate it, restated; parity unpinned): the standard's calibration
    tone, gain linearity, the -14 LKFS target of `audio_write(strategy="loudness")`, silence and short clips."""
    import math
    import torch
    L = load_pkg("loudness")
    sr = 24000
    t = torch.arange(sr * 3) / sr
    tone = torch.sin(2 * math.pi * 997 * t)
    # a 997 Hz sine 20 dB below full scale in one channel reads -23.01 LKFS (BS.1770 calibration: 0 dBFS -> -3.01)
    assert abs(L.integrated_loudness(0.1 * tone, sr) - (-23.01)) < 0.1
    # +6.02 dB of gain is +6.02 LU
    a, b = L.integrated_loudness(0.05 * tone, sr), L.integrated_loudness(0.1 * tone, sr)
    assert abs((b - a) - 20 * math.log10(2)) < 1e-3
    g = torch.Generator().manual_seed(0)
    noise = 0.03 * torch.randn(1, sr * 4, generator=g)
    out = L.normalize_loudness(noise, sr, loudness_compressor=False)
    assert abs(L.integrated_loudness(out, sr) - (-14.0)) < 1e-3
    comp = L.normalize_loudness(noise, sr, loudness_compressor=True)
    assert float(comp.abs().max()) < 1.0 and torch.allclose(comp, torch.tanh(out))
    quiet = 1e-4 * torch.randn(1, sr, generator=g)
    assert L.normalize_loudness(quiet, sr) is quiet                      # below the energy floor: untouched
    assert L.integrated_loudness(torch.zeros(sr), sr) == float("-inf")   # digital silence
    assert L.integrated_loudness(tone[: sr // 10], sr) == float("-inf")  # shorter than one 400 ms block


def test_bench_byte_accounting():
    """bench.py prices a decode step with SURVEY 8d's figure: 2 476 945 408 B of weights + 393 216 B per cached position."""
    import importlib.util
    import types
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cfg = types.SimpleNamespace(n_layer=24, n_head=16, dim=2048, intermediate_size=5632, vocab_size=2562)
    assert bench.step_bytes(cfg, -1) == 2_476_945_408
    assert bench.step_bytes(cfg, 99) - bench.step_bytes(cfg, 98) == 393_216
    assert bench.TOKENS_PER_AUDIO_SECOND == 150 and (bench.T_PROMPT, bench.N_NEW) == (128, 1920)


def test_second_stage_packing_matches_the_reference():
    """a15: `pack_second_stage_input` against what the reference's own `Model.non_causal_sample` hands to `model.generate`
    (tests/golden/packing.json, minted by executing fam/llm/inference.py:264-306 with audiocraft stubbed)."""
    cases = json.load(open(os.path.join(GOLDEN, "packing.json")))
    assert len(cases) >= 3
    truncated = 0
    for c in cases:
        for ids, cbs, want in zip(c["text_ids"], c["codebooks"], c["packed"]):
            got = inference.pack_second_stage_input(ids, cbs, c["ctx"])
            assert got == want
            truncated += int(len(ids) + len(cbs[0]) + 1 > c["ctx"])
    assert truncated >= 1  # the truncating branch (:299-300) is covered


def test_long_form_chunker():
    longform = load_pkg("longform")
    text = ("This is the first sentence. Here is a second one, with a clause; and another! Is this the third? " * 40).strip()
    chunks = longform.split_text(text)
    assert all(0 < len(c) <= 220 for c in chunks)
    assert " ".join(chunks) == " ".join(text.split())  # nothing dropped, nothing reordered
    assert all(c[-1] in ".!?" for c in chunks)           # sentences are kept whole when they fit
    assert len(chunks) >= len(text) // 220
    # a sentence longer than the limit falls back to clause / word boundaries, a giant "word" to hard cuts
    monster = "word " * 100 + "x" * 500 + ", tail."
    parts = longform.split_text(monster, max_chars=120)
    assert all(0 < len(p) <= 120 for p in parts) and "".join(parts).replace(" ", "") == monster.replace(" ", "")
    assert longform.split_text("   ") == [] and longform.split_text("Short.") == ["Short."]
    ten_k = ("The quick brown fox jumps over the lazy dog near the quiet river bank. " * 150)[:10000]
    n = len(longform.split_text(ten_k))
    assert 46 <= n <= 60  # ~46-50 chunks for 10k characters


def test_bench_work_lists_for_every_baseline_config():
    """bench.make_work (host logic, no GPU): configs[1] one 128-token prompt + 1920 tokens; configs[2] ragged prompts 96..160 that
    differ per rank and never overrun the context window; configs[4] ~46-60 sentence chunks of <= 220 characters."""
    import importlib
    import sys

    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    mv = load_pkg()
    args = mv.ModelArgs.from_name("metavoice-1B")
    one = bench.make_work(mv, args, long_form=False, long_form_chars=0, utterances_per_gpu=1, rank=0, world=1)
    assert len(one) == 1 and len(one[0]) == 1 and len(one[0][0][0]) == 128 and one[0][0][1] == 1920
    seen = set()
    for rank in range(8):
        (group,) = bench.make_work(mv, args, long_form=False, long_form_chars=0, utterances_per_gpu=4, rank=rank, world=8)
        assert len(group) == 4
        for prompt, n_new in group:
            assert 96 <= len(prompt) <= 160 and len(prompt) + n_new <= args.block_size and n_new >= 1888
            seen.add(tuple(prompt.tolist()))
    assert len(seen) == 32  # batch 32: every utterance of every rank is different
    long = bench.make_work(mv, args, long_form=True, long_form_chars=10000, utterances_per_gpu=1, rank=0, world=1)
    assert 46 <= len(long) <= 60
    for (ids, n_new), in long:
        assert len(ids) + n_new <= args.block_size and n_new >= 100


def test_bench_counts_the_launches_of_the_multi_step_decode_kernel():
    """`roofline.avg_launch_us` = HIP-event time of the decode region / this count: it has to equal what rocprofv3 sees as calls of
    decode_mega_kernel<.., true> (profiles/r02b_fold: 4 for the default utterance)."""
    import bench
    p = [0] * 128
    assert bench.decode_launches([[(p, 1920)]]) == 4                      # 1919 steps = 512 + 512 + 512 + 383
    assert bench.decode_launches([[(p, 513)]]) == 1 and bench.decode_launches([[(p, 514)]]) == 2
    assert bench.decode_launches([[(p, 40), (p, 700)], [(p, 2)]]) == 2 + 1  # a group runs as long as its longest utterance
    assert bench.decode_launches([[(p, 1920)]], steps_per_launch=16) == 120



def test_split_text_keeps_closing_quotes_and_brackets_with_their_sentence():
    """the sentence-end pattern used to swallow the quotes / brackets that follow the punctuation, so the text fed to
    the first stage differed from the input ('He said "Stop." Then left.' lost its closing quote)."""
    import random

    longform = load_pkg("longform")
    text = 'He said "Stop." Then left. (Really?) She asked: \'Why?\' No answer... [sic.] The end!'
    chunks = longform.split_text(text, max_chars=24)
    assert " ".join(chunks) == text and all(len(c) <= 24 for c in chunks)
    assert chunks[0] == 'He said "Stop."'
    rng = random.Random(7)
    alphabet = 'ab c.!?,;:"\')]( \n'
    for _ in range(3000):
        t = "".join(rng.choice(alphabet) for _ in range(rng.randint(0, 90)))
        mc = rng.randint(8, 40)
        ch = longform.split_text(t, mc)
        norm = " ".join(t.split())
        assert all(0 < len(c) <= mc for c in ch)
        if all(len(w) <= mc for w in norm.split()):
            assert " ".join(ch) == norm, (t, mc, ch)   # nothing dropped, nothing reordered; only hard cuts of over-long words add spaces
        else:
            assert "".join(ch).replace(" ", "") == norm.replace(" ", "")


def test_fast_inference_command_line():
    """`fast_inference.py:198-199`: the module builds a TTS from the command line; here with the constructor's keywords as flags."""
    fi = load_pkg("fast_inference")
    a = fi._parse_cli(["--model_name", "/models/mv", "--quantisation_mode", "int8", "--seed", "7"])
    assert (a.model_name, a.quantisation_mode, a.seed, a.output_dir, a.text) == ("/models/mv", "int8", 7, "outputs", None)
    a = fi._parse_cli(["--text", "hello", "--spk_ref_path", "ref.wav", "--top_p", "0.9"])
    assert (a.text, a.spk_ref_path, a.top_p, a.guidance_scale, a.temperature) == ("hello", "ref.wav", 0.9, 3.0, 1.0)
    with pytest.raises(SystemExit):
        fi._parse_cli(["--text", "no speaker"])


def test_bpe_matches_independent_formulations_on_random_vocabularies():
    """tokenise.py restates tiktoken 0.5.1's `_byte_pair_merge` (fam/quantiser/text/tokenise.py:5-12; tiktoken is not in the image, the
    real ranks live in the unreachable checkpoint).  Pinned here against two formulations written independently of it, on random
    TRAINED-LIKE vocabularies (every merge joins two existing tokens, ranks in creation order -- the only kind BPE training emits):
      (a) the incremental form tiktoken 0.5.1 itself uses (`_byte_pair_merge`: a vector of (start, rank of the pair starting here),
          only the two ranks next to a merge are recomputed);
      (b) a span-based brute force of "fuse the adjacent pair of lowest rank, leftmost first" that never builds byte strings for the
          parts (boundaries into the original piece + a rank lookup on slices).
    All three must give the same segmentation; a bug in the merge loop (wrong tie order, stale pair after a
    merge, off-by-one at the ends) separates them on random byte strings."""
    import random

    tok_mod = load_pkg("tokenise")
    rng = random.Random(1234)

    def random_vocab(alphabet, n_merges):
        toks = [bytes([c]) for c in alphabet]
        ranks = {bytes([c]): c for c in range(256)}
        merges = []
        nxt = 256
        while len(merges) < n_merges:
            a, b = rng.choice(toks), rng.choice(toks)
            if a + b in ranks or len(a + b) > 6:
                continue
            ranks[a + b] = nxt
            merges.append((a, b))
            toks.append(a + b)
            nxt += 1
        return ranks, merges

    def incremental(ranks, piece):
        MAX = 1 << 60
        parts = [[i, MAX] for i in range(len(piece) + 1)]

        def get_rank(start_idx, skip):
            if start_idx + skip + 2 < len(parts):
                return ranks.get(piece[parts[start_idx][0]:parts[start_idx + skip + 2][0]], MAX)
            return MAX

        for i in range(len(parts) - 2):
            parts[i][1] = get_rank(i, 0)
        while len(parts) > 1:
            best_rank, best_i = MAX, -1
            for i in range(len(parts) - 1):
                if parts[i][1] < best_rank:
                    best_rank, best_i = parts[i][1], i
            if best_rank == MAX:
                break
            i = best_i
            parts[i][1] = get_rank(i, 1)
            if i > 0:
                parts[i - 1][1] = get_rank(i - 1, 1)
            parts.pop(i + 1)
        return [piece[parts[k][0]:parts[k + 1][0]] for k in range(len(parts) - 1)]

    def span_brute_force(ranks, piece):
        cuts = list(range(len(piece) + 1))   # boundaries: part k = piece[cuts[k]:cuts[k+1]]
        while len(cuts) > 2:
            best = min(((ranks.get(piece[cuts[k]:cuts[k + 2]], 1 << 60), k) for k in range(len(cuts) - 2)))
            if best[0] == 1 << 60:
                break
            del cuts[best[1] + 1]
        return [piece[cuts[k]:cuts[k + 1]] for k in range(len(cuts) - 1)]

    checked = 0
    for trial in range(60):
        alphabet = rng.sample(range(97, 123), rng.randint(2, 5))
        ranks, merges = random_vocab(alphabet, rng.randint(3, 40))
        for _ in range(40):
            piece = bytes(rng.choice(alphabet) for _ in range(rng.randint(1, 24)))
            got = tok_mod._bpe(ranks, piece)
            # tiktoken's `_encode_native` looks the whole piece up before it merges anything (src/lib.rs: `if let Some(token) = encoder.get(piece)`)
            want_a = [ranks[piece]] if piece in ranks else [ranks[p] for p in incremental(ranks, piece)]
            want_b = [ranks[piece]] if piece in ranks else [ranks[p] for p in span_brute_force(ranks, piece)]
            assert got == want_b, (piece, got, want_b)
            assert got == want_a, (piece, got, want_a)
            checked += 1
    assert checked == 2400
    # the classic overlap case: "aaa" with the merge (a, a) fuses the LEFT pair
    assert tok_mod._bpe({b"a": 0, b"aa": 1}, b"aaa") == [1, 0]


def test_grouped_vocoder_call_keeps_a_failure_with_its_utterance(tmp_path):
    """`decode_wavs` vocodes a whole group in one call; with `return_exceptions` a failing utterance -- its token
    grid, a native error on one vocoder lane, a waveform under 400 ms (fam/llm/decoders.py:88-91) -- must leave its exception in its own
    slot and the rest of the group must still come out; without it the first failure propagates, as `decode()` of the reference does."""
    import torch

    decoders = load_pkg("decoders")

    class Vocoder:
        def __init__(self):
            self.single_calls = 0

        def __call__(self, grid):
            self.single_calls += 1
            if int(grid[0, 0, 0]) == 13:
                raise RuntimeError("native lane error")
            return torch.zeros(1, 1, 320 * grid.shape[-1])

        def tokens_to_wav_many(self, grids):
            if any(int(g[0, 0, 0]) == 13 for g in grids):
                raise RuntimeError("native lane error")     # the grouped call fails as a whole ...
            return [torch.zeros(1, 1, 320 * g.shape[-1]) for g in grids]

    def adapter(tokens):
        if tokens == "bad":
            raise ValueError("adapter refused")
        return [0], tokens

    voc = Vocoder()
    dec = decoders.EncodecDecoder(lambda ids: "t", adapter, str(tmp_path), vocoder=voc)
    ok = [[7] * 40] * 8       # 40 frames = 12 800 samples
    short = [[7] * 10] * 8    # 3 200 samples < 9 600
    lane = [[13] * 40] * 8
    out = dec.decode_wavs([ok, "bad", short, lane, ok], return_exceptions=True)
    assert [type(o).__name__ for o in out] == ["Tensor", "ValueError", "Exception", "RuntimeError", "Tensor"]
    assert "shorter than 400ms" in str(out[2]) and out[0].shape == (1, 1, 12800)
    assert voc.single_calls == 4   # ... and the four utterances with a grid were vocoded one by one
    with pytest.raises(ValueError):
        dec.decode_wavs([ok, "bad"], return_exceptions=False)
    with pytest.raises(Exception, match="shorter than 400ms"):
        dec.decode_wavs([ok, short], return_exceptions=False)
    with pytest.raises(RuntimeError):
        dec.decode_wavs([ok, lane], return_exceptions=False)
    assert [o.shape[-1] for o in dec.decode_wavs([ok, ok])] == [12800, 12800]
