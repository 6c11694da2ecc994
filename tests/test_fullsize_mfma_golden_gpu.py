"""The matrix-core decode kernels (`decode_mfma_kernel<4>`, `<8>`: csrc/decode_mfma.hip) against the REFERENCE at MetaVoice-1B size.

Three of the bench line's legs (configs[2] at 4 and at 8 utterances per GPU, configs[4] long-form) decode on these kernels.  What they
must hold is `Transformer.forward` per CFG row pair (fam/llm/fast_model.py:150-227) and per-row independence inside a batch
(fam/llm/mixins/causal.py:179-287: every utterance keeps its own position, its own stop, its own stream).  The goldens are outputs of the
reference's own `generate()` at 24 layers x 2048 (`tests/golden/make_golden.py --full24 / --long24`): T = 128 (one key range), T = 1100
(8 key ranges per head + the split combine) and T = 2016 (ends at sequence index 2047, the last row of the cache).

One engine launch holds ALL THREE attention regimes side by side: the slots of an engine of 4 (8 CFG rows) and of an engine of 8 (16 rows)
each carry one of the three golden streams, teacher-forced (`prefill_mode=0`: every position of every stream is a decode step of the
persistent matrix-core kernel, `decode_path() == (3, 0)`), and every slot is held to the assertions of
`test_fullsize_golden_gpu.py::test_full_size_engine_follows_the_reference`:
  raw / CFG-mixed logits within NOISE_FACTOR x what the reference's own bf16 evaluation loses against its fp32 one on that stream,
  argmax equal wherever the reference's top-2 margin exceeds twice the measured error of that step,
  agreement with the fp32 stream no worse than the reference-in-bf16's own, minus LAG.
Then the free-running leg: ragged MFMA prefill of the three contexts + the kernel's own greedy decode, forks margin-gated.
The sigma figures printed here are the ones BASELINE.md quotes for configs[2] / [2'] / [4].
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_pkg

pytestmark = pytest.mark.gpu

mv = load_pkg()
synthetic = load_pkg("synthetic")
fiu = load_pkg("fast_inference_utils")

FULL = dict(n_layer=24, n_head=16, dim=2048, vocab_size=2562, block_size=2048)
NOISE_FACTOR = 1.5  # as test_fullsize_golden_gpu.py
LAG = 3
N_FREE = 32         # free-running tokens per slot (the T = 2016 stream ends at the block size after 32)

# which golden stream each slot carries: neighbours in one MFMA row tile always sit in different attention regimes
SLOTS = {4: [128, 1100, 2016, 1100], 8: [2016, 128, 1100, 2016, 128, 1100, 2016, 128]}


def mix(lg):  # (n, 2, V) -> (n, V): the reference's CFG formula at guidance 3 (fast_inference_utils.py:113-117)
    return 3.0 * lg[:, 0] + (1.0 - 3.0) * lg[:, 1]


@pytest.fixture(scope="module")
def weights():
    return fiu.remap_checkpoint_keys(synthetic.stage1_state_dict(seed=0, **FULL))


@pytest.fixture(scope="module")
def streams():
    """T -> the reference's stream, the logits of its forward calls, its bf16-vs-fp32 floors and its own bf16 picks."""
    out = {}
    g = np.load(os.path.join(GOLDEN, "stage1_full24.npz"))
    rb = np.load(os.path.join(GOLDEN, "stage1_full24_refbf16.npz"))
    assert json.loads(str(g["cfg"])) == FULL
    out[128] = dict(T=128, n_new=int(g["n_new"]), seq=g["seq_bf16"].tolist(), logits=torch.from_numpy(g["logits_bf16"].astype(np.float32)),
                    raw_floor=g["noise_raw_refbf16"], mix_floor=g["noise_mix_refbf16"], pick=rb["pick_refbf16_bf16"],
                    prompt=synthetic.prompt_tokens(128, seed=0))
    for T in (1100, 2016):
        g = np.load(os.path.join(GOLDEN, f"stage1_long24_T{T}.npz"))
        assert json.loads(str(g["cfg"])) == FULL and int(g["T"]) == T
        out[T] = dict(T=T, n_new=int(g["n_new"]), seq=g["seq_bf16"].tolist(), logits=torch.from_numpy(g["logits_bf16"].astype(np.float32)),
                      raw_floor=g["noise_raw_refbf16_bf16"], mix_floor=g["noise_mix_refbf16_bf16"], pick=g["pick_refbf16_bf16"],
                      prompt=synthetic.long_prompt_tokens(T, seed=0))
    for s in out.values():
        assert s["seq"][:s["T"]] == s["prompt"].tolist() and len(s["seq"]) == s["T"] + s["n_new"]
    return out


@pytest.mark.parametrize("n_utt", [4, 8])
def test_matrix_core_kernels_follow_the_reference_at_full_size(weights, streams, n_utt):
    slots = SLOTS[n_utt]
    engine = mv.Stage1Engine(mv.ModelArgs(**FULL), max_utterances=n_utt)
    engine.load_state_dict(weights)
    engine.finalize()
    engine.capture_logits(True)
    spk1 = synthetic.speaker_embedding(0)
    spks = torch.cat([spk1] * n_utt)
    sp = mv.SamplingParams(top_k=1, guidance_scale=3.0, seed=1)

    # (i) teacher-forced: the reference's streams are the "prompts"; every position of every slot is a decode step of the kernel
    forced = [streams[T]["seq"][:T + streams[T]["n_new"] - 1] for T in slots]
    engine.begin(forced, spks, sp, max_new_tokens=1, end_of_audio_token=9999, prefill_mode=0)
    engine.run()
    assert engine.decode_path() == (3, 0), "decode_mfma.hip did not run (or a hand-off timed out): these numbers would be another kernel's"
    margins, worst = [], dict(raw=0.0, mix=0.0, raw_floor=0.0, mix_floor=0.0)
    for u, T in enumerate(slots):
        s = streams[T]
        n_new, glog, gseq = s["n_new"], s["logits"], s["seq"]
        got = engine.captured_logits(T - 1, n_new)[:, 2 * u:2 * u + 2]
        raw_sigma, mix_sigma = float(glog.std()), float(mix(glog).std())
        d_raw = (got - glog).abs().amax(dim=(1, 2))
        gm, em = mix(glog), mix(got)
        d_mix = (em - gm).abs().amax(dim=1)
        top2 = gm.topk(2, dim=-1).values
        margin = top2[:, 0] - top2[:, 1]
        margins.append(margin)
        decided = margin > 2 * d_mix
        same = em.argmax(-1) == torch.tensor(gseq[T:])
        ref_same = int((torch.from_numpy(s["pick"]).long() == torch.tensor(gseq[T:])).sum())
        worst["raw"] = max(worst["raw"], float(d_raw.max()) / raw_sigma)
        worst["mix"] = max(worst["mix"], float(d_mix.max()) / mix_sigma)
        worst["raw_floor"] = max(worst["raw_floor"], float(s["raw_floor"].max()) / raw_sigma)
        worst["mix_floor"] = max(worst["mix_floor"], float(s["mix_floor"].max()) / mix_sigma)
        print(f"[mfma x {n_utt} full24, slot {u}: T={T}] teacher-forced positions {T - 1}..{T + n_new - 2}: max|draw| {float(d_raw.max()):.4f} "
              f"({float(d_raw.max()) / raw_sigma:.3f} sigma; reference-bf16 floor {float(s['raw_floor'].max()):.4f}), max|dmix| {float(d_mix.max()):.4f} "
              f"({float(d_mix.max()) / mix_sigma:.3f} sigma; floor {float(s['mix_floor'].max()):.4f}); agreement with the reference's fp32 stream: "
              f"engine {int(same.sum())}/{n_new}, the reference evaluated in bf16 {ref_same}/{n_new}; {int(decided.sum())} steps decided")
        assert float(d_raw.max()) <= NOISE_FACTOR * float(s["raw_floor"].max()), f"slot {u} (T={T}): raw logits off by {float(d_raw.max()):.4f}"
        assert float(d_mix.max()) <= NOISE_FACTOR * float(s["mix_floor"].max()), f"slot {u} (T={T}): mixed logits off by {float(d_mix.max()):.4f}"
        assert float(d_raw.mean()) <= NOISE_FACTOR * float(s["raw_floor"].mean())
        assert bool(same[decided].all()), f"slot {u} (T={T}): argmax differs from the reference at a step the reference decides clearly"
        assert int(same.sum()) >= ref_same - LAG, f"slot {u} (T={T}): leaves the fp32 stream more often than the reference's own bf16 evaluation"
    # slots carrying the same stream are the same arithmetic in another row of the tile: bit-identical
    for T in set(slots):
        us = [u for u, t in enumerate(slots) if t == T]
        n_new = streams[T]["n_new"]
        lg = engine.captured_logits(T - 1, n_new)
        for u in us[1:]:
            assert torch.equal(lg[:, 2 * u:2 * u + 2], lg[:, 2 * us[0]:2 * us[0] + 2]), f"slots {us[0]} and {u} carry the same stream and differ"
    print(f"[mfma x {n_utt} full24] worst slot: raw {worst['raw']:.3f} sigma (reference-bf16 floor {worst['raw_floor']:.3f}), "
          f"CFG mix {worst['mix']:.3f} sigma (floor {worst['mix_floor']:.3f})")

    # (ii) free running: ragged MFMA prefill of the three contexts, then the kernel's own greedy decode of every slot at its own position
    engine.begin([streams[T]["prompt"] for T in slots], spks, sp, max_new_tokens=N_FREE, end_of_audio_token=9999, prefill_mode=1)
    engine.run()
    assert engine.decode_path() == (3, 0)
    agree = []
    for u, T in enumerate(slots):
        s = streams[T]
        seq = engine.tokens(u)
        assert len(seq) == T + N_FREE and seq[:T] == s["prompt"].tolist()
        fork = next((i for i, (a, b) in enumerate(zip(seq, s["seq"])) if a != b), None)
        agree.append(N_FREE if fork is None else fork - T)
        d_pre = float((engine.captured_logits(T - 1, 1)[:, 2 * u:2 * u + 2] - s["logits"][:1]).abs().max())
        assert d_pre <= NOISE_FACTOR * float(s["raw_floor"].max()), f"slot {u} (T={T}): MFMA-prefill first-token logits off by {d_pre:.4f}"
        if fork is not None:
            m = float(margins[u][fork - T])
            assert m <= 2 * NOISE_FACTOR * float(s["mix_floor"].max()), \
                f"slot {u} (T={T}): stream forks at generated token {fork - T} where the reference's top-2 margin is {m:.4f}"
    print(f"[mfma x {n_utt} full24] free-running greedy streams identical to the reference's for {agree} of {N_FREE} tokens (slots {slots})")
    engine.close()
