"""The oracle (oracle/) against the golden vectors minted from the reference itself
(tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_pkg
from oracle import adapters as o_adapters
from oracle import quantize as o_quant
from oracle import rng as o_rng
from oracle import stage1 as o_s1
from oracle import stage2 as o_s2

synthetic = load_pkg("synthetic")


def _stage1(name, rounding="fp32"):
    g = np.load(os.path.join(GOLDEN, f"stage1_{name}.npz"))
    kw = json.loads(str(g["cfg"]))
    sd = o_s1.remap_checkpoint_keys(synthetic.stage1_state_dict(seed=int(g["seed"]), **kw))
    cfg = o_s1.Stage1Config(**kw)
    model = o_s1.Stage1Oracle({k: v.float() for k, v in sd.items()}, cfg, rounding=rounding, max_seq_length=cfg.block_size)
    prompt = synthetic.prompt_tokens(int(g["T"]), seed=int(g["seed"]))
    spk = synthetic.speaker_embedding(int(g["seed"]))
    return g, model, prompt, spk


@pytest.mark.parametrize("name", ["small", "wide2"])
def test_stage1_fp32_matches_reference(name):
    g, model, prompt, spk = _stage1(name)
    seq, logits = o_s1.generate(model, prompt, spk, max_new_tokens=int(g["n_new"]), end_of_audio_token=9999,
                                top_k=1, noise=False, return_logits=True)
    assert seq == g["seq"].tolist()  # greedy token stream of the reference's generate(top_k=1)
    assert float((logits - torch.from_numpy(g["logits_fp32"])).abs().max()) < 2e-5


def test_stage1_bf16_points_close_to_reference_bf16():
    g, model, prompt, spk = _stage1("small", rounding="bf16")
    seq = g["seq"].tolist()
    T = int(g["T"])
    out = [model.forward(prompt.view(1, -1).repeat(2, 1), spk, torch.arange(T))[:, -1]]
    for i in range(int(g["n_new"]) - 1):
        out.append(model.forward(torch.full((2, 1), seq[T + i]), spk, torch.tensor([T + i]))[:, -1])
    out = torch.stack(out)
    ref16 = torch.from_numpy(g["logits_bf16"])
    ref32 = torch.from_numpy(g["logits_fp32"])
    # bf16 noise floor: the reference's own bf16 run is this far from its fp32 run
    floor = float((ref16 - ref32).abs().max())
    assert float((out - ref32).abs().max()) <= 2.0 * floor
    assert float((out - ref16).abs().max()) <= 3.0 * floor
    assert torch.equal(out, out.bfloat16().float())  # logits are bf16-representable


def test_sampling_matches_reference():
    g = np.load(os.path.join(GOLDEN, "sampling.npz"))
    logits = torch.from_numpy(g["logits"])
    for ci, c in enumerate(json.loads(str(g["combos"]))):
        for li in range(logits.shape[0]):
            p = o_s1.logits_to_probs(logits[li].clone(), c["temperature"], top_p=c["top_p"], top_k=c["top_k"])
            assert float((p - torch.from_numpy(g["probs"][ci, li])).abs().max()) < 1e-6, (c, li)
    _, p = o_s1.sample(torch.from_numpy(g["cfg_pair"]), 3.0, 1.0, top_p=0.95)
    assert float((p - torch.from_numpy(g["cfg_probs"])).abs().max()) < 1e-6


def test_greedy_tie_keeps_both_and_race_is_deterministic():
    logits = torch.zeros(2, 1, 16)
    logits[:, :, 3] = 5.0
    logits[:, :, 9] = 5.0
    q = torch.ones(16)
    idx, probs = o_s1.sample(logits, 3.0, 1.0, top_k=1, q=q)
    assert probs[3] == probs[9] == 0.5 and idx == 3  # both survive (`logits < pivot`), lowest id wins a dead heat
    q[3] = 2.0
    assert o_s1.sample(logits, 3.0, 1.0, top_k=1, q=q)[0] == 9


def test_quantisers_match_reference():
    g = np.load(os.path.join(GOLDEN, "quant.npz"))
    w = torch.from_numpy(g["w"]).to(torch.bfloat16)
    q, sz = o_quant.group_quantize_tensor(w)
    assert torch.equal(q, torch.from_numpy(g["q4"]).to(torch.int32))
    assert torch.equal(sz.float(), torch.from_numpy(g["scales_and_zeros"]))
    assert torch.equal(o_quant.group_dequantize(q, sz, out_dtype=torch.bfloat16).float(), torch.from_numpy(g["deq"]))
    q8, s8 = o_quant.quantize_per_channel_int8(w.float())
    assert torch.equal(q8, torch.from_numpy(g["q8"])) and torch.equal(s8, torch.from_numpy(g["s8"]))
    assert o_quant.int4_eligible(6144, 2048) and not o_quant.int4_eligible(2562, 2048)


def test_adapters_and_normalize_text_match_reference():
    with open(os.path.join(GOLDEN, "adapters.json")) as f:
        g = json.load(f)
    for c in g["cases"]:
        fn = o_adapters.flattened_interleaved_decode if c["kind"] == "flattened" else o_adapters.tilted_decode
        text, audio = fn(c["tokens"])
        assert text == c["text"] and audio == c["audio"]
    for c in g["normalize"]:
        if c.get("raises"):
            with pytest.raises(ValueError):
                o_adapters.normalize_text(c["text"])
        else:
            assert o_adapters.normalize_text(c["text"]) == c["out"]


def test_pack_second_stage_input():
    rows = o_adapters.pack_second_stage_input([1030, 1031, 1537], [[1, 2, 3], [4, 5, 6]], ctx_window=10)
    assert rows[0] == [1030, 1031, 1537, 1, 2, 3, 1024, 1024, 1024, 1024]
    assert rows[1] == [1024, 1024, 1024, 4, 5, 6, 1024, 1024, 1024, 1024]
    rows = o_adapters.pack_second_stage_input([1030, 1031, 1537], [[1, 2, 3], [4, 5, 6]], ctx_window=5)
    assert rows == [[1030, 1031, 1537, 1, 2], [1024, 1024, 1024, 4, 5]]  # truncation, inference.py:299-300


@pytest.mark.parametrize("name", ["rms_swiglu", "ln_gelu_bias"])
def test_stage2_matches_reference(name):
    g = np.load(os.path.join(GOLDEN, f"stage2_{name}.npz"))
    args = json.loads(str(g["model_args"]))
    sd = synthetic.stage2_state_dict(args, speaker_emb_dim=256, seed=int(g["seed"]))
    cfg = o_s2.Stage2Config(**{k: v for k, v in args.items() if k in o_s2.Stage2Config.__dataclass_fields__})
    orc = o_s2.Stage2Oracle(sd, cfg)
    idx = torch.from_numpy(g["idx"])
    spk = synthetic.speaker_embedding(2)
    logits = torch.stack(orc.forward(idx, spk))[:, :, ::4]
    assert float((logits - torch.from_numpy(g["logits_sub"])).abs().max()) < 5e-6
    assert torch.equal(orc.non_causal_sample(idx, spk, 1.0, top_k=1), torch.from_numpy(g["tokens"]))


def test_philox_known_answer():
    # Random123 kat_vectors: philox4x32-10, counter = key = 0 and all-ones
    out = o_rng.philox4x32_10(np.uint32(0), np.uint32(0), np.uint32(0), np.uint32(0), 0, 0)
    assert [int(x) for x in out] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    f = 0xFFFFFFFF
    out = o_rng.philox4x32_10(np.uint32(f), np.uint32(f), np.uint32(f), np.uint32(f), f, f)
    assert [int(x) for x in out] == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    q = o_rng.exp1_noise(1234, 5, 2562)
    assert q.dtype == np.float32 and np.all(q > 0) and np.isfinite(q).all() and abs(float(q.mean()) - 1.0) < 0.1


@pytest.mark.parametrize("T", [1100, 2016])
def test_long_context_goldens_are_self_consistent(T):
    """`stage1_long24_T*.npz` (make_golden.py --long24; 24 layers, minted by the reference's `generate()`): the stored stream is the
    argmax of the stored CFG-mixed fp32 logits, the context is `synthetic.long_prompt_tokens`, and the reference-in-bf16 picks /
    noise floors cover every step.  (The 24-layer oracle needs ~1 min per forward pass on a CPU: the engine-vs-golden comparison
    itself is tests/test_fullsize_golden_gpu.py.)"""
    g = np.load(os.path.join(GOLDEN, f"stage1_long24_T{T}.npz"))
    n_new = int(g["n_new"])
    assert int(g["T"]) == T and T + n_new <= 2048
    prompt = synthetic.long_prompt_tokens(T, seed=int(g["seed"]))
    lg = torch.from_numpy(g["logits_bf16"])
    assert lg.shape == (n_new, 2, 2562) and lg.dtype == torch.float32
    seq = g["seq_bf16"].tolist()
    assert seq[:T] == prompt.tolist()
    assert (3.0 * lg[:, 0] - 2.0 * lg[:, 1]).argmax(-1).tolist() == seq[T:]
    for mode in ("bf16", "int4", "int8"):
        assert g[f"seq_{mode}"].shape == (T + n_new,) and g[f"seq_{mode}"][:T].tolist() == prompt.tolist()
        for k in ("noise_raw_refbf16", "noise_mix_refbf16", "pick_refbf16", "ties_refbf16", "margin_refbf16"):
            assert g[f"{k}_{mode}"].shape == (n_new,)
        # the reference's own bf16 evaluation stays within ~0.2 sigma of its fp32 logits and mostly picks the fp32 token
        assert 0.05 < float(g[f"noise_raw_refbf16_{mode}"].max()) < 0.25
        assert int((g[f"pick_refbf16_{mode}"] == g[f"seq_{mode}"][T:]).sum()) >= n_new - 6
