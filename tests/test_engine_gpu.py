"""First-stage engine (hipGraph decode loop in libmetavoice_hip.so) against the oracle, on the GPU.

Parity criteria: the engine computes in bf16 with fp32 accumulation; the oracle runs the reference
algorithm in fp32 with bf16 rounding at the same tensor boundaries (rounding="bf16"), on identical bf16 weights.  The two
differ only by fp32 summation order, so
  (i)  logits at every position agree to a few bf16 ulps of the logit scale, and
  (ii) the free-running token stream is identical unless the oracle's own top-1/top-2 gap at the first mismatch is inside
       that noise (then the streams legitimately fork and the test stops comparing there).
The fp32-arithmetic oracle (what the reference computes on CPU, golden-pinned) is compared too, with the bf16 noise floor
measured on the reference itself (tests/golden: |bf16 - fp32| ~ 1e-2 on logits of std ~0.3 for the small model).
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_pkg
from oracle import stage1 as o_s1

pytestmark = pytest.mark.gpu

mv = load_pkg()
synthetic = load_pkg("synthetic")

SMALL = dict(n_layer=2, n_head=2, dim=256, vocab_size=2562, block_size=64)
WIDE2 = dict(n_layer=2, n_head=16, dim=2048, vocab_size=2562, block_size=2048)


def build(cfg_kwargs, seed, max_utterances=1):
    sd = o_s1.remap_checkpoint_keys(synthetic.stage1_state_dict(seed=seed, **cfg_kwargs))
    args = mv.ModelArgs(**cfg_kwargs)
    engine = mv.Stage1Engine(args, max_utterances=max_utterances)
    engine.load_state_dict(sd)
    engine.finalize()
    cfg = o_s1.Stage1Config(**cfg_kwargs)
    return engine, sd, cfg


def oracle_for(sd, cfg, rounding):
    return o_s1.Stage1Oracle({k: v.float() for k, v in sd.items()}, cfg, rounding=rounding, max_seq_length=cfg.block_size)


def first_fork(a, b):
    for i, (x, y) in enumerate(zip(a, b)):
        if x != y:
            return i
    return None if len(a) == len(b) else min(len(a), len(b))


def check_stream(engine_seq, oracle_seq, oracle_logits, T, gap_tol):
    """Identical, or forked at a position where the oracle's CFG-mixed top-2 gap is below `gap_tol`."""
    fork = first_fork(engine_seq, oracle_seq)
    if fork is None:
        return len(oracle_seq) - T
    assert fork >= T, "prompt was altered"
    lg = oracle_logits[fork - T]  # logits that decided sequence index `fork`
    mixed = o_s1.cfg_mix(lg[:, None, :], 3.0, o_s1._round_bf16)[0]
    top2 = torch.topk(mixed, 2).values
    gap = float(top2[0] - top2[1])
    assert gap <= gap_tol, f"streams fork at index {fork} although the oracle's top-2 gap is {gap:.4f} > {gap_tol}"
    return fork - T


@pytest.mark.parametrize("name,cfgk,seed,T,n_new", [("small", SMALL, 0, 12, 20), ("wide2", WIDE2, 1, 16, 6)])
def test_greedy_against_oracle_and_golden(name, cfgk, seed, T, n_new):
    engine, sd, cfg = build(cfgk, seed)
    prompt = synthetic.prompt_tokens(T, seed=seed)
    spk = synthetic.speaker_embedding(seed)
    engine.capture_logits(True)
    engine.begin([prompt], spk, mv.SamplingParams(top_k=1, seed=5), max_new_tokens=n_new, end_of_audio_token=9999)
    steps = engine.run()
    assert steps == T + n_new - 1
    seq = engine.tokens(0)
    assert len(seq) == T + n_new and seq[:T] == prompt.tolist()
    got = engine.captured_logits(0, T + n_new - 1)  # (pos, rows, V)

    # (a) oracle with the engine's rounding points, free-running with the same Philox noise
    ob = oracle_for(sd, cfg, "bf16")
    oseq, ologits = o_s1.generate(ob, prompt, spk, max_new_tokens=n_new, end_of_audio_token=9999, top_k=1, seed=5,
                                  return_logits=True)
    scale = float(ologits.std())
    agree = check_stream(seq, oseq, ologits, T, gap_tol=0.05 * scale)
    # logits of every position up to the fork (the prefill positions are teacher-forced in both)
    ob2 = oracle_for(sd, cfg, "bf16")
    pre = ob2.forward(prompt.view(1, -1).repeat(2, 1), spk, torch.arange(T))  # (2, T, V)
    d_pre = float((got[:T].transpose(0, 1) - pre).abs().max())
    d_dec = float((got[T - 1:T - 1 + max(agree, 1)] - ologits[:max(agree, 1)]).abs().max())
    print(f"[{name}] stream agreement {agree}/{n_new} tokens; max|dlogit| prefill {d_pre:.4f} decode {d_dec:.4f} (logit std {scale:.3f})")
    assert d_pre <= 0.06 * scale and d_dec <= 0.06 * scale

    # (b) the fp32 reference numbers (golden, minted from the reference itself)
    g = np.load(os.path.join(GOLDEN, f"stage1_{name}.npz"))
    assert json.loads(str(g["cfg"])) == cfgk
    gold = torch.from_numpy(g["logits_fp32"])  # (n_new, 2, V): prefill-last then teacher-forced decode steps
    gseq = g["seq"].tolist()
    fork = first_fork(seq, gseq)
    n_cmp = n_new if fork is None else fork - T + 1
    d32 = float((got[T - 1:T - 1 + n_cmp] - gold[:n_cmp]).abs().max())
    print(f"[{name}] vs reference fp32 golden: first fork {fork}, max|dlogit| {d32:.4f}")
    assert d32 <= 0.12 * scale
    engine.close()


def race_check(lg, q, token, nat, top_p, top_k, tol):
    """The Exp(1) race of `fast_inference_utils.py:61-65,107-120` on the oracle's logits `lg` (2, V) with the shared Philox noise
    `q`: returns (oracle winner, ok) where ok = `token` IS the winner, or loses to it by less than a perturbation of `tol` on
    each of the two temperature-scaled mixed logits can explain -- and then it must also sit inside, or within `tol` below,
    the kept set (a token the oracle filtered out clearly can never be drawn)."""
    mixed = o_s1.cfg_mix(lg[:, None, :], nat.guidance_scale, o_s1._round_bf16)[0]
    temp = max(nat.temperature, 1e-5)
    scaled = o_s1._round_bf16(mixed / temp).float()
    probs = o_s1.logits_to_probs(mixed, nat.temperature, top_p=top_p, top_k=top_k, rnd=o_s1._round_bf16)
    score = torch.where(probs > 0, scaled - torch.log(q), torch.full_like(scaled, float("-inf")))  # log(p / q) up to a constant
    winner = int(score.argmax())
    if winner == token:
        return winner, True
    floor = float(scaled[probs > 0].min())
    near_kept = float(scaled[token]) >= floor - tol / temp
    near_tie = float(score[winner] - (scaled[token] - torch.log(q[token]))) <= 2 * tol / temp
    return winner, near_kept and near_tie


@pytest.mark.parametrize("combo", [dict(top_p=0.95), dict(top_p=0.9, temperature=0.8), dict(top_k=50), dict()])
def test_stochastic_stream_matches_oracle(combo):
    """Every sampled token is checked, not only the first half of the stream: the oracle is teacher-forced along the ENGINE's
    stream and, at every step, the engine's token must win the oracle's race under the same Philox noise -- or be a near-tie
    in that race (the `check_stream` rule carried over to the stochastic sampler)."""
    from oracle import rng as o_rng

    engine, sd, cfg = build(SMALL, 3)
    T, n_new = 10, 40
    prompt = synthetic.prompt_tokens(T, seed=3)
    spk = synthetic.speaker_embedding(3)
    sp = mv.SamplingParams(guidance_scale=3.0, seed=99, **combo)
    nat = sp.to_native(torch.bfloat16)
    top_p = nat.top_p if sp.top_p is not None else None
    engine.begin([prompt], spk, sp, max_new_tokens=n_new, end_of_audio_token=9999)
    engine.run()
    seq = engine.tokens(0)
    assert len(seq) == T + n_new
    ob = oracle_for(sd, cfg, "bf16")
    logits = ob.forward(prompt.view(1, -1).repeat(2, 1), spk, torch.arange(T))[:, -1]
    scale = None
    exact = 0
    for i in range(n_new):
        pos = T + i  # sequence index of the token being drawn
        if scale is None:
            scale = float(logits.std())
        q = torch.from_numpy(o_rng.exp1_noise(99, pos, cfg.vocab_size))
        winner, ok = race_check(logits, q, seq[pos], nat, top_p, sp.top_k, tol=0.05 * scale)
        exact += int(winner == seq[pos])
        assert ok, f"sampled token {i} ({seq[pos]}) neither wins nor nearly ties the oracle's race (winner {winner})"
        if i + 1 < n_new:
            logits = ob.forward(torch.full((2, 1), seq[pos], dtype=torch.long), spk, torch.tensor([pos]))[:, -1]
    print(f"[stochastic {combo}] {exact}/{n_new} sampled tokens are the oracle's race winners, the rest near-ties")
    assert exact >= n_new - 4
    assert len(set(seq[T:])) > 3  # it really is sampling
    engine.close()


def test_end_of_audio_stops_and_is_included():
    engine, sd, cfg = build(SMALL, 0)
    T = 12
    prompt = synthetic.prompt_tokens(T, seed=0)
    spk = synthetic.speaker_embedding(0)
    engine.begin([prompt], spk, mv.SamplingParams(top_k=1), max_new_tokens=20, end_of_audio_token=9999)
    engine.run()
    full = engine.tokens(0)
    eoa = full[T + 5]  # declare the 6th generated token to be end-of-audio
    first = full[T:].index(eoa)
    engine.begin([prompt], spk, mv.SamplingParams(top_k=1), max_new_tokens=20, end_of_audio_token=eoa)
    steps = engine.run(poll_every=4)
    seq = engine.tokens(0)
    assert seq == full[: T + first + 1] and seq[-1] == eoa  # EOA itself is part of the result (fast_inference_utils.py:161)
    assert steps <= T + first + 4
    engine.close()


def test_prompt_too_long_and_length_cap():
    engine, sd, cfg = build(SMALL, 0)
    spk = synthetic.speaker_embedding(0)
    with pytest.raises(ValueError, match="Prompt is too long"):
        engine.begin([synthetic.prompt_tokens(64, seed=0)], spk, mv.SamplingParams(top_k=1))
    # max_new_tokens=None -> run to block_size (fast_inference_utils.py:195-201)
    engine.begin([synthetic.prompt_tokens(60, seed=0)], spk, mv.SamplingParams(top_k=1), end_of_audio_token=9999)
    engine.run()
    assert len(engine.tokens(0)) == 64
    engine.close()


def test_two_utterances_with_ragged_prompts_match_single_runs():
    """Rows of different utterances never interact: a batch of 2 (different prompt lengths, speakers, seeds) must give
    exactly the streams each utterance gives alone."""
    engine2, sd, cfg = build(SMALL, 4, max_utterances=2)
    prompts = [synthetic.prompt_tokens(9, seed=10), synthetic.prompt_tokens(17, seed=11)]
    spks = torch.cat([synthetic.speaker_embedding(10), synthetic.speaker_embedding(11)])
    sps = [mv.SamplingParams(top_p=0.95, seed=1), mv.SamplingParams(top_k=1, seed=2)]
    engine2.begin(prompts, spks, sps, max_new_tokens=24, end_of_audio_token=9999)
    engine2.run()
    both = [engine2.tokens(0), engine2.tokens(1)]
    engine2.close()
    engine1, _, _ = build(SMALL, 4, max_utterances=1)
    for u in range(2):
        engine1.begin([prompts[u]], spks[u:u + 1], sps[u], max_new_tokens=24, end_of_audio_token=9999)
        engine1.run()
        single = engine1.tokens(0)
        assert single == both[u]  # the noise stream is keyed by (seed, token id, sequence index), not by the slot
        assert len(both[u]) == len(prompts[u]) + 24
    engine1.close()


def test_batch_of_two_utterances_follows_the_oracle():
    """SURVEY 8f row 2 against the ORACLE, not the engine itself: two utterances with ragged prompts, their own speaker embeddings
    and seeds decoded together (persistent kernel, 2 utterances per launch) -- each stream must be the oracle's single-utterance
    stream, or fork only at an oracle near-tie (`check_stream`)."""
    engine, sd, cfg = build(SMALL, 6, max_utterances=2)
    lens, n_new = [9, 17], 24
    prompts = [synthetic.prompt_tokens(t, seed=20 + u) for u, t in enumerate(lens)]
    spks = torch.cat([synthetic.speaker_embedding(20 + u) for u in range(2)])
    sps = [mv.SamplingParams(top_k=1, seed=3 + u) for u in range(2)]
    engine.begin(prompts, spks, sps, max_new_tokens=n_new, end_of_audio_token=9999, prefill_mode=1)
    engine.run()
    for u in range(2):
        seq = engine.tokens(u)
        ob = oracle_for(sd, cfg, "bf16")
        oseq, ologits = o_s1.generate(ob, prompts[u], spks[u:u + 1], max_new_tokens=n_new, end_of_audio_token=9999, top_k=1, seed=3 + u,
                                      return_logits=True)
        agree = check_stream(seq, oseq, ologits, lens[u], gap_tol=0.05 * float(ologits.std()))
        print(f"[batch of 2] utterance {u}: {agree}/{n_new} tokens identical to the oracle's single-utterance stream")
        assert len(seq) == lens[u] + n_new and agree >= 1
    engine.close()


@pytest.mark.parametrize("name,cfgk,seed,T,n_new", [("small", SMALL, 0, 12, 20), ("wide2", WIDE2, 1, 130, 8)])
def test_batched_prefill_matches_teacher_forcing_and_oracle(name, cfgk, seed, T, n_new):
    """`prefill()` (fast_inference_utils.py:123-132) as MFMA GEMMs + causal attention == the same prompt pushed token by
    token through the decode graph: same first-token logits (to fp32 summation order) and the same greedy stream."""
    engine, sd, cfg = build(cfgk, seed)
    prompt = synthetic.prompt_tokens(T, seed=seed)
    spk = synthetic.speaker_embedding(seed)
    sp = mv.SamplingParams(top_k=1, seed=5)
    engine.capture_logits(True)
    runs = {}
    for mode in (0, 1):
        engine.begin([prompt], spk, sp, max_new_tokens=n_new, end_of_audio_token=9999, prefill_mode=mode)
        steps = engine.run()
        assert steps == (T + n_new - 1 if mode == 0 else n_new - 1)
        runs[mode] = (engine.tokens(0), engine.captured_logits(T - 1, n_new))
    ob = oracle_for(sd, cfg, "bf16")
    oseq, ologits = o_s1.generate(ob, prompt, spk, max_new_tokens=n_new, end_of_audio_token=9999, top_k=1, seed=5, return_logits=True)
    scale = float(ologits.std())
    d01 = float((runs[0][1][0] - runs[1][1][0]).abs().max())
    d1o = float((runs[1][1][0] - ologits[0]).abs().max())
    print(f"[{name}] first-token logits: prefill vs teacher-forced {d01:.4f}, prefill vs oracle {d1o:.4f} (logit std {scale:.3f})")
    assert d01 <= 0.06 * scale and d1o <= 0.06 * scale
    assert len(runs[1][0]) == T + n_new and runs[1][0][:T] == prompt.tolist()
    agree = check_stream(runs[1][0], oseq, ologits, T, gap_tol=0.05 * scale)
    print(f"[{name}] prefill-mode stream agreement with the oracle: {agree}/{n_new}")
    engine.close()


def test_batched_prefill_ragged_two_utterances():
    engine2, sd, cfg = build(SMALL, 4, max_utterances=2)
    prompts = [synthetic.prompt_tokens(9, seed=10), synthetic.prompt_tokens(17, seed=11)]
    spks = torch.cat([synthetic.speaker_embedding(10), synthetic.speaker_embedding(11)])
    sps = [mv.SamplingParams(top_k=1, seed=1), mv.SamplingParams(top_k=1, seed=2)]
    out = {}
    for mode in (0, 1):
        engine2.begin(prompts, spks, sps, max_new_tokens=12, end_of_audio_token=9999, prefill_mode=mode)
        engine2.run()
        out[mode] = [engine2.tokens(0), engine2.tokens(1)]
        assert [len(t) for t in out[mode]] == [9 + 12, 17 + 12]
    # greedy streams agree unless a near-tie flips on fp32 summation order: require the first 4 generated tokens at least
    for u in range(2):
        T = len(prompts[u])
        assert out[0][u][: T + 4] == out[1][u][: T + 4]
    engine2.close()


@pytest.mark.parametrize("n_layer,T,n_new,combo,poll", [(2, 16, 24, dict(top_k=1), 16), (4, 40, 48, dict(top_p=0.95), 1 << 20), (3, 300, 40, dict(top_p=0.9, temperature=0.8), 16),
                                                        (2, 250, 16, dict(top_k=1), 5),       # crosses 256 keys: one key range -> 8 ranges + combine
                                                        (2, 2030, 17, dict(top_p=0.95), 16),  # up to the last cache row (block_size - 1)
                                                        (2, 16, 600, dict(top_p=0.95), 1 << 20)])  # more steps than one launch takes (512)
def test_persistent_decode_kernel_is_bit_identical_to_per_layer_launches(n_layer, T, n_new, combo, poll):
    """decode_mega.hip (24 layers + head + -- mega_fold=1, the default -- the sampler and the next input rows of up to 512 tokens per
    launch, granule hand-offs) runs the same arithmetic in the same order as gemv.hip / attention.hip / sample_kernel: logits and
    token streams must be IDENTICAL, not merely close -- any difference is a hand-off bug (stale or torn activation), which a
    tolerance would hide.  Three paths: several steps per launch with the token drawn inside, one step per launch followed by
    sample_kernel (the captured graph), per-layer launches."""
    cfgk = dict(WIDE2, n_layer=n_layer)
    engine, sd, cfg = build(cfgk, 7)
    prompt = synthetic.prompt_tokens(T, seed=7)
    spk = synthetic.speaker_embedding(7)
    sp = mv.SamplingParams(guidance_scale=3.0, seed=21, **combo)
    engine.capture_logits(True)
    out = {}
    for path in ((1, 1), (0, 0), (1, 0), (1, 1)):  # (mega, mega_fold)
        engine.tune(mega=path[0], mega_fold=path[1])
        engine.begin([prompt], spk, sp, max_new_tokens=n_new, end_of_audio_token=9999, prefill_mode=1)
        assert engine.run(poll_every=poll) == n_new - 1
        res = (engine.tokens(0), engine.captured_logits(T - 1, n_new))
        if path in out:
            assert res[0] == out[path][0] and torch.equal(res[1], out[path][1])  # run-to-run reproducible
        out[path] = res
    engine.tune(mega=1, mega_fold=1)  # the defaults
    # a persistent launch that is abandoned falls back to the launches and produces the same bits: without this the test could pass
    # on the fall-back alone
    assert engine.decode_path() == (2, 0), "the persistent kernel was abandoned (hand-off timed out) during the test"
    for path in ((1, 1), (1, 0)):
        assert out[path][0] == out[(0, 0)][0], f"token streams differ between the persistent kernel {path} and the per-layer launches"
        assert torch.equal(out[path][1], out[(0, 0)][1]), f"logits differ {path}: max {float((out[path][1] - out[(0, 0)][1]).abs().max())}"
    assert len(out[(1, 1)][0]) == T + n_new
    engine.close()


@pytest.mark.parametrize("n_utt,n_layer,lens,n_new,combo", [
    (2, 3, [40, 23], 30, dict(top_p=0.95)),
    (2, 2, [250, 700], 20, dict(top_k=1)),                    # one utterance crosses 256 keys (1 -> 8 key ranges), the other is beyond
    (4, 3, [96, 160, 131, 17], 40, dict(top_p=0.95)),
    (4, 2, [300, 12, 2000, 255], 24, dict(top_p=0.9, temperature=0.8)),  # every attention regime at once
    (3, 2, [70, 300, 19], 20, dict(top_k=1)),                 # an engine "of 3": four slots, one idle (Stage1Engine rounds up to a persistent-kernel size)
])
def test_persistent_kernel_with_several_utterances_is_bit_identical_to_launches(n_utt, n_layer, lens, n_new, combo):
    """The persistent kernel decoding 2 or 4 utterances together (the weight rows a wave holds are applied to every utterance;
    per-utterance positions, key ranges, seeds and end-of-audio) against the per-layer launches on the same batch, and against
    each utterance decoded ALONE: identical tokens and logits."""
    cfgk = dict(WIDE2, n_layer=n_layer)
    engine, sd, cfg = build(cfgk, 13, max_utterances=n_utt)
    assert engine.slots == (4 if n_utt == 3 else n_utt)
    prompts = [synthetic.prompt_tokens(t, seed=30 + u) for u, t in enumerate(lens)]
    spks = torch.cat([synthetic.speaker_embedding(40 + u) for u in range(n_utt)])
    sps = [mv.SamplingParams(guidance_scale=3.0, seed=50 + u, **combo) for u in range(n_utt)]
    engine.capture_logits(True)
    out = {}
    engine.tune(mfma=0)   # engines of 4 utterances default to the matrix-core instance (decode_mfma.hip: another summation order, its own test below)
    for mega in (1, 0):
        engine.tune(mega=mega)
        engine.begin(prompts, spks, sps, max_new_tokens=n_new, end_of_audio_token=9999, prefill_mode=1)
        engine.run()
        toks = [engine.tokens(u) for u in range(n_utt)]
        first = min(lens) - 1
        out[mega] = (toks, engine.captured_logits(max(lens) - 1, n_new - 1)[:, :2 * n_utt])  # steps every utterance is decoding in; an idle slot's rows are never written
    engine.tune(mega=1)
    assert engine.decode_path() == (1, 0), "the persistent kernel was abandoned (hand-off timed out) during the test"
    engine.tune(mfma=1)   # process-wide: back to the default
    for u in range(n_utt):
        assert len(out[1][0][u]) == lens[u] + n_new
        assert out[1][0][u] == out[0][0][u], f"utterance {u}: token streams differ between the persistent kernel and the launches"
    assert torch.equal(out[1][1], out[0][1]), f"logits differ: max {float((out[1][1] - out[0][1]).abs().max())}"
    engine.close()
    # each utterance on its own (single-utterance persistent kernel): same stream as inside the batch
    single, _, _ = build(cfgk, 13, max_utterances=1)
    for u in (0, n_utt - 1):
        single.begin([prompts[u]], spks[u:u + 1], sps[u], max_new_tokens=n_new, end_of_audio_token=9999, prefill_mode=1)
        single.run()
        assert single.tokens(0) == out[1][0][u], f"utterance {u} decodes differently alone and in a batch of {n_utt}"
    single.close()


@pytest.mark.parametrize("n_layer,lens,n_new,combo", [
    (3, [96, 160, 131, 17], 40, dict(top_k=1)),
    (2, [300, 12, 2000, 255], 24, dict(top_k=1)),    # every attention regime at once: one key range, 8 ranges + combine, the last cache rows
    (4, [600, 1100, 40, 258], 32, dict(top_p=0.95)),
])
def test_matrix_core_persistent_kernel_for_four_utterances(n_layer, lens, n_new, combo):
    """decode_mfma.hip: the persistent step of 4-utterance engines on v_mfma_f32_16x16x32_bf16 (tiled weight copy, K split over the 8 waves
    of a workgroup, packed half tiles).  Another ORDER of the fp32 additions inside every projection and (round 6) the RMSNorm weights folded
    into the tiled matrices (one more bf16 rounding of a weight, two fewer of every normalised activation):
    not bit-identical, so (i) every position of every utterance TEACHER-FORCED through both paths (`prefill_mode=0`: the prompt goes through
    the decode kernels token by token) must give logits within 0.1 sigma of each other (engine vs oracle in this file: 0.06 sigma at 2 layers), (ii) the
    free-running streams agree with the launches' wherever the launches' own top-2 gap of the CFG mix exceeds that noise, and (iii) the
    kernel is the one that ran (`decode_path() == (3, 0)`) and is run-to-run reproducible."""
    cfgk = dict(WIDE2, n_layer=n_layer)
    engine, sd, cfg = build(cfgk, 17, max_utterances=4)
    prompts = [synthetic.prompt_tokens(t, seed=90 + u) for u, t in enumerate(lens)]
    spks = torch.cat([synthetic.speaker_embedding(95 + u) for u in range(4)])
    sps = [mv.SamplingParams(guidance_scale=3.0, seed=70 + u, **combo) for u in range(4)]
    engine.capture_logits(True)
    L = max(lens)
    forced = {}
    for mfma, mega in ((1, 1), (0, 0), (1, 1)):
        engine.tune(mfma=mfma, mega=mega)
        engine.begin(prompts, spks, sps, max_new_tokens=2, end_of_audio_token=9999, prefill_mode=0)
        engine.run()
        lg = engine.captured_logits(0, L)   # (L, 8 rows, vocab): entry p = logits computed at input position p
        if mfma:   # (first: a run that fell back to the launches half way would show as "not reproducible" below)
            assert engine.decode_path() == (3, 0), "decode_mfma.hip did not run (or a hand-off timed out)"
        for u, t in enumerate(lens):        # only the positions an utterance really decoded: the rest of the capture buffer is never written
            if (mfma, mega) in forced:
                assert torch.equal(lg[:t, 2 * u:2 * u + 2], forced[(mfma, mega)][:t, 2 * u:2 * u + 2]), "the matrix-core kernel is not reproducible run to run"
        forced[(mfma, mega)] = lg
    worst = 0.0
    for u, t in enumerate(lens):
        a, b = forced[(1, 1)][:t, 2 * u:2 * u + 2], forced[(0, 0)][:t, 2 * u:2 * u + 2]   # the positions utterance u really decoded
        scale = float(b.std())
        d = float((a - b).abs().max())
        worst = max(worst, d / scale)
        # two bf16 evaluations that differ in the order of fp32 additions: 1-2 bf16 ulps of a logit (0.016-0.031 at |x| ~ 2-4, sigma 0.9) per
        # layer pair; measured 0.039-0.059 sigma at 2-4 layers (profiles/r05e), bound 0.1 sigma
        assert d <= 0.1 * scale, f"utterance {u}: teacher-forced logits differ by {d:.4f} (logit std {scale:.3f})"
    # free running
    free = {}
    for mfma, mega in ((1, 1), (0, 0), (1, 1)):   # ragged prompts behind the MFMA prefill: every utterance at its own position
        engine.tune(mfma=mfma, mega=mega)
        engine.begin(prompts, spks, sps, max_new_tokens=n_new, end_of_audio_token=9999, prefill_mode=1)
        engine.run()
        # logits per utterance at the positions IT decoded (the capture is indexed by each utterance's own input position)
        res = ([engine.tokens(u) for u in range(4)], [engine.captured_logits(t, n_new - 1)[:, 2 * u:2 * u + 2].clone() for u, t in enumerate(lens)])
        if mfma in free:
            assert res[0] == free[mfma][0] and all(torch.equal(x, y) for x, y in zip(res[1], free[mfma][1])), "free-running: the matrix-core kernel is not reproducible run to run"
        free[mfma] = res
    engine.tune(mfma=1, mega=1)
    agree = []
    for u, t in enumerate(lens):
        ta, tb = free[1][0][u], free[0][0][u]
        assert len(ta) == len(tb) == t + n_new and ta[:t] == tb[:t]
        fork = first_fork(ta, tb)
        agree.append(n_new if fork is None else fork - t)
        if fork is not None and "top_k" in combo and fork - 1 >= t:   # greedy, and the deciding logits were captured: the launches' gap there must be inside the noise
            lg = free[0][1][u][fork - 1 - t]
            mixed = o_s1.cfg_mix(lg[:, None, :], 3.0, o_s1._round_bf16)[0]
            top2 = torch.topk(mixed, 2).values
            assert float(top2[0] - top2[1]) <= 0.25, f"utterance {u} forks at {fork} although the launches' top-2 gap is {float(top2[0] - top2[1]):.3f}"
    print(f"[mfma x 4, {n_layer} layers, prompts {lens}] teacher-forced max |dlogit| / sigma {worst:.4f}; free-running agreement with the launches {agree} of {n_new}")
    if "top_k" in combo:
        # greedy: a fork is margin-gated above (the launches' top-2 gap at the fork is inside the noise); sampled streams (top-p) may fork at the
        # first near-tie of the race.  (Until round 6 this also asked for two common tokens; with the norm weight folded into the tiled matrices
        # -- decode_mfma.hip MF_FOLD_NORM: other rounding points, not only another order -- one of these random-weight streams meets its first
        # near-tie one token earlier.  The 24-layer reference goldens are the criterion: tests/test_fullsize_mfma_golden_gpu.py.)
        assert min(agree) >= 1
    engine.close()


def test_matrix_core_persistent_kernel_for_eight_utterances():
    """decode_mfma_kernel<8>: 16 CFG rows = one full 16 x 16 MFMA tile of activations; the w2 phase in two passes over the column halves (16
    whole activation rows would be 180 KB of LDS), one workgroup per (utterance, row, head) walking all of its key ranges.  There are no
    per-layer launches for 16 rows, so the reference is the SAME eight utterances decoded four at a time by an engine of 4 on the per-layer
    launches (the bit-identical family), teacher-forced; bound as for the 4-utterance instance (0.1 sigma)."""
    cfgk = dict(WIDE2, n_layer=2)
    lens = [300, 12, 1030, 255, 40, 700, 17, 1500]
    sd = o_s1.remap_checkpoint_keys(synthetic.stage1_state_dict(seed=19, **cfgk))
    prompts = [synthetic.prompt_tokens(t, seed=110 + u) for u, t in enumerate(lens)]
    spks = torch.cat([synthetic.speaker_embedding(120 + u) for u in range(8)])
    sps = [mv.SamplingParams(guidance_scale=3.0, seed=130 + u, top_k=1) for u in range(8)]
    ref = []
    e4 = mv.Stage1Engine(mv.ModelArgs(**cfgk), max_utterances=4)
    e4.load_state_dict(sd); e4.finalize()
    e4.capture_logits(True)
    e4.tune(mega=0)
    for g0 in (0, 4):
        e4.begin(prompts[g0:g0 + 4], spks[g0:g0 + 4], sps[g0:g0 + 4], max_new_tokens=2, end_of_audio_token=9999, prefill_mode=0)
        e4.run()
        lg = e4.captured_logits(0, max(lens[g0:g0 + 4]))
        ref += [lg[:lens[g0 + k], 2 * k:2 * k + 2].clone() for k in range(4)]
    e4.tune(mega=1)
    e4.close()
    e8 = mv.Stage1Engine(mv.ModelArgs(**cfgk), max_utterances=8)
    e8.load_state_dict(sd); e8.finalize()
    e8.capture_logits(True)
    got = None
    for rep in range(2):
        e8.begin(prompts, spks, sps, max_new_tokens=2, end_of_audio_token=9999, prefill_mode=0)
        e8.run()
        assert e8.decode_path() == (3, 0), "decode_mfma_kernel<8> did not run (or a hand-off timed out)"
        lg = e8.captured_logits(0, max(lens))
        cur = [lg[:t, 2 * u:2 * u + 2].clone() for u, t in enumerate(lens)]
        if got is not None:
            assert all(torch.equal(x, y) for x, y in zip(cur, got)), "not reproducible run to run"
        got = cur
    worst = 0.0
    for u in range(8):
        scale, d = float(ref[u].std()), float((got[u] - ref[u]).abs().max())
        worst = max(worst, d / scale)
        assert d <= 0.1 * scale, f"utterance {u} (prompt {lens[u]}): teacher-forced logits differ by {d:.4f} (logit std {scale:.3f})"
    # free running behind the MFMA prefill: ragged positions, every attention regime, length caps honoured
    e8.begin(prompts, spks, sps, max_new_tokens=20, end_of_audio_token=9999, prefill_mode=1)
    e8.run()
    assert [len(e8.tokens(u)) for u in range(8)] == [t + 20 for t in lens] and e8.decode_path() == (3, 0)
    print(f"[mfma x 8, prompts {lens}] teacher-forced max |dlogit| / sigma vs the launches (engines of 4) {worst:.4f}")
    full = [e8.tokens(u) for u in range(8)]
    e8.close()
    # an engine "of 6": Stage1Engine rounds up to 8 slots, two stay idle (done from the start: no K/V reads, their rows never written) -- the six
    # utterances must decode exactly as they did in the full batch (every utterance's arithmetic is its own: same kernel, same order)
    e6 = mv.Stage1Engine(mv.ModelArgs(**cfgk), max_utterances=6)
    assert e6.slots == 8
    e6.load_state_dict(sd); e6.finalize()
    e6.begin(prompts[:6], spks[:6], sps[:6], max_new_tokens=20, end_of_audio_token=9999, prefill_mode=1)
    e6.run()
    assert e6.decode_path() == (3, 0)
    assert [e6.tokens(u) for u in range(6)] == full[:6], "six utterances on an engine of 8 slots decode differently from the full batch"
    e6.close()


def test_eight_utterance_engine_has_per_layer_launches_behind_the_matrix_core_kernel():
    """An engine of 8 utterances with `mv_tune("mega", 0)`: the per-layer launches issued once per half of the 16 rows (csrc/engine.hip
    enqueue_step).  Utterances 0..3 and 4..7 must decode exactly -- tokens and logits, bit for bit -- as they do four at a time on an engine
    of 4 on the same launches (every row's arithmetic is its own), ragged prompts behind the MFMA prefill and teacher-forced."""
    cfgk = dict(WIDE2, n_layer=2)
    lens = [300, 12, 1030, 255, 40, 700, 17, 1500]
    sd = o_s1.remap_checkpoint_keys(synthetic.stage1_state_dict(seed=23, **cfgk))
    prompts = [synthetic.prompt_tokens(t, seed=210 + u) for u, t in enumerate(lens)]
    spks = torch.cat([synthetic.speaker_embedding(220 + u) for u in range(8)])
    sps = [mv.SamplingParams(guidance_scale=3.0, seed=230 + u, top_p=0.95) for u in range(8)]
    n_new = 24
    want_tok, want_lg = [], []
    e4 = mv.Stage1Engine(mv.ModelArgs(**cfgk), max_utterances=4)
    e4.load_state_dict(sd); e4.finalize()
    e4.capture_logits(True)
    e4.tune(mega=0)
    for g0 in (0, 4):
        e4.begin(prompts[g0:g0 + 4], spks[g0:g0 + 4], sps[g0:g0 + 4], max_new_tokens=n_new, end_of_audio_token=9999, prefill_mode=1)
        e4.run()
        assert e4.decode_path() == (0, 0)
        want_tok += [e4.tokens(k) for k in range(4)]
        want_lg += [e4.captured_logits(lens[g0 + k], n_new - 1)[:, 2 * k:2 * k + 2].clone() for k in range(4)]
    e4.tune(mega=1)
    e4.close()
    e8 = mv.Stage1Engine(mv.ModelArgs(**cfgk), max_utterances=8)
    e8.load_state_dict(sd); e8.finalize()
    e8.capture_logits(True)
    try:
        e8.tune(mega=0)
        e8.begin(prompts, spks, sps, max_new_tokens=n_new, end_of_audio_token=9999, prefill_mode=1)
        e8.run()
        assert e8.decode_path() == (0, 0)
        for u, t in enumerate(lens):
            assert e8.tokens(u) == want_tok[u], f"utterance {u}: the halved launches of an engine of 8 draw other tokens than an engine of 4"
            assert torch.equal(e8.captured_logits(t, n_new - 1)[:, 2 * u:2 * u + 2], want_lg[u])
        # and the matrix-core kernel against these launches on the SAME engine (free-running greedy would fork at near-ties: teacher-forced)
        short = [p[:200] for p in prompts]
        forced = {}
        for mega in (0, 1):
            e8.tune(mega=mega)
            e8.begin(short, spks, sps, max_new_tokens=2, end_of_audio_token=9999, prefill_mode=0)
            e8.run()
            assert e8.decode_path() == ((3, 0) if mega else (0, 0))
            lg = e8.captured_logits(0, 200)
            forced[mega] = [lg[:len(p), 2 * u:2 * u + 2].clone() for u, p in enumerate(short)]
        worst = max(float((a - b).abs().max()) / float(b.std()) for a, b in zip(forced[1], forced[0]))
        print(f"[engine of 8] decode_mfma_kernel<8> vs its own per-layer launches (two halves), teacher-forced: max |dlogit| / sigma {worst:.4f}")
        assert worst <= 0.1
    finally:
        e8.tune(mega=1)
    e8.close()


@pytest.mark.parametrize("n_utt", [4, 8])
def test_matrix_core_kernel_times_out_cleanly_and_the_batch_finishes_on_the_launches(n_utt):
    """a hand-off time-out of `decode_mfma_kernel<4>` / `<8>` (grid not co-resident) must not fail the batch.  The
    `mega_fault` hook makes one workgroup withhold one publish in every launch: the first chunk aborts, every utterance is rolled back to
    the last poll and the batch -- that chunk and everything after it -- is decoded by the per-layer launches (engines of 8: once per half of
    the rows).  `mv_run` returns success, `mv_decode_path` counts the abandoned launch, and since nothing of the aborted chunk counted the
    tokens and logits are those of a run on the launches from the start, bit for bit.  Then the back-off / retry cycle of the one-utterance
    test, and the kernel comes back once the obstacle is gone."""
    cfgk = dict(WIDE2, n_layer=2)
    lens = [40, 300, 17, 1030, 12, 700, 255, 90][:n_utt]
    engine, sd, cfg = build(cfgk, 29, max_utterances=n_utt)
    prompts = [synthetic.prompt_tokens(t, seed=310 + u) for u, t in enumerate(lens)]
    spks = torch.cat([synthetic.speaker_embedding(320 + u) for u in range(n_utt)])
    sps = [mv.SamplingParams(guidance_scale=3.0, seed=330 + u, top_p=0.9) for u in range(n_utt)]
    n_new = 40
    engine.capture_logits(True)

    def run(poll_every=16):
        engine.begin(prompts, spks, sps, max_new_tokens=n_new, end_of_audio_token=9999, prefill_mode=1)
        steps = engine.run(poll_every=poll_every)
        assert steps == n_new - 1
        return [engine.tokens(u) for u in range(n_utt)], [engine.captured_logits(t, n_new - 1)[:, 2 * u:2 * u + 2].clone() for u, t in enumerate(lens)]

    try:
        engine.tune(mega=0)
        want = run()
        assert engine.decode_path() == (0, 0)
        engine.tune(mega=1)
        clean = run()
        assert engine.decode_path() == (3, 0)
        engine.tune(mega=1, mega_fault=8)
        got = run()
        assert engine.decode_path() == (0, 1), "one launch abandoned, the engine continues on the per-layer launches"
        assert got[0] == want[0] and all(torch.equal(a, b) for a, b in zip(got[1], want[1]))
        for _ in range(3):   # stays on the launches for the next three batches (no further time-out) ...
            assert run()[0] == want[0] and engine.decode_path() == (0, 1)
        assert run()[0] == want[0] and engine.decode_path() == (0, 2)   # ... tries again with the fourth: the fault is still there
        engine.tune(mega_fault=0)
        for _ in range(7):
            run()
            assert engine.decode_path() == (0, 2)
        back = run()   # the obstacle is gone: the eighth batch after the second failure is the matrix-core kernel's again
        assert engine.decode_path() == (3, 2)
        assert back[0] == clean[0] and all(torch.equal(a, b) for a, b in zip(back[1], clean[1])), "the kernel came back with other bits than before the fault"
    finally:
        engine.tune(mega_fault=0)
        engine.tune(mega=1)
    engine.close()


def test_matrix_core_kernel_selected_between_begin_and_run():
    """`mv_tune` between `mv_begin` and `mv_run` (the graph is re-captured in `mv_run`): an engine of 4 that began on the per-layer launches and is
    switched to the matrix-core kernel before it runs -- behind an EARLIER run of that kernel, whose last launch left real values in its x rows -- must
    find its tag-free hand-off rows armed (sentinel), i.e. decode exactly as a run that began on the kernel."""
    cfgk = dict(WIDE2, n_layer=2)
    lens = [40, 300, 17, 260]
    engine, sd, cfg = build(cfgk, 31, max_utterances=4)
    prompts = [synthetic.prompt_tokens(t, seed=410 + u) for u, t in enumerate(lens)]
    spks = torch.cat([synthetic.speaker_embedding(420 + u) for u in range(4)])
    sps = [mv.SamplingParams(guidance_scale=3.0, seed=430 + u, top_p=0.9) for u in range(4)]
    try:
        engine.tune(mega=1, mfma=1)
        engine.begin(prompts, spks, sps, max_new_tokens=24, end_of_audio_token=9999, prefill_mode=1)
        engine.run()
        assert engine.decode_path() == (3, 0)
        want = [engine.tokens(u) for u in range(4)]
        engine.tune(mega=0)
        engine.begin(prompts, spks, sps, max_new_tokens=24, end_of_audio_token=9999, prefill_mode=1)
        engine.tune(mega=1)          # selected after mv_begin
        engine.run()
        assert engine.decode_path() == (3, 0)
        assert [engine.tokens(u) for u in range(4)] == want
    finally:
        engine.tune(mega=1, mfma=1)
    engine.close()


@pytest.mark.parametrize("n_utt", [4, 8])
def test_matrix_core_hand_offs_survive_skewed_workgroups(n_utt):
    """The hand-offs of `decode_mfma.hip` (tag-free rows pulled with LDS-DMA, a sentinel word for "not yet published") under UNEVEN arrival: the
    `mega_fault = 1000` hook makes every workgroup write back its XCD's L2 behind the last layer's x publish, so that the workgroups reach the
    following hand-offs microseconds apart, first looks come back incomplete and the repeated looks run.  Every run must stay on the kernel
    (`decode_path() == (3, 0)`: no wait ran out) and give the bits of the first run.  History (profiles/r06m): the first version of the row hand-off
    put the last layer's activation slots back to the sentinel as soon as ONE WAVE had seen its share of the x vector -- a fast workgroup then
    reset slots a slow one was still reading, which looked at sentinels until its wait ran out: 5-7 time-outs per 10^6 launches, 14 runs of 30 under
    this hook, none since the reset sits behind the workgroup's barrier.  `tools/mfma_repro.py` is the long form of this test."""
    cfgk = dict(WIDE2, n_layer=2)
    lens = [300, 12, 1000, 255, 40, 800, 17, 450][:n_utt]
    engine, sd, cfg = build(cfgk, 17, max_utterances=n_utt)
    prompts = [synthetic.prompt_tokens(t, seed=90 + u) for u, t in enumerate(lens)]
    spks = torch.cat([synthetic.speaker_embedding(95 + u) for u in range(n_utt)])
    sps = [mv.SamplingParams(guidance_scale=3.0, seed=70 + u, top_k=1) for u in range(n_utt)]
    engine.capture_logits(True)
    first = None
    try:
        engine.tune(mega=1, mega_fault=1000)
        for rep in range(30):
            engine.begin(prompts, spks, sps, max_new_tokens=2, end_of_audio_token=9999, prefill_mode=0)
            engine.run()
            assert engine.decode_path() == (3, 0), f"run {rep}: a hand-off wait ran out under skew"
            lg = engine.captured_logits(0, max(lens))
            cur = [lg[:t, 2 * u:2 * u + 2].clone() for u, t in enumerate(lens)]
            if first is None:
                first = cur
            else:
                assert all(torch.equal(a, b) for a, b in zip(cur, first)), f"run {rep}: other bits under skew (a stale value was taken for a published one)"
    finally:
        engine.tune(mega_fault=0)
    engine.close()


def test_persistent_kernel_times_out_cleanly_and_the_run_falls_back_to_launches():
    """A persistent kernel whose grid is not co-resident (or whose peer dies) must not hang and must not fail the run: every
    wait is bounded, the first workgroup to give up raises a device flag, the rest leave within a polling interval, and mv_run
    rolls the utterance back to its last poll and finishes it with per-layer launches -- the same bits.  The `mega_fault` hook
    makes one workgroup withhold one publish in every launch."""
    engine, sd, cfg = build(WIDE2, 11)
    prompt, spk = synthetic.prompt_tokens(16, seed=11), synthetic.speaker_embedding(11)
    sp = mv.SamplingParams(top_p=0.9, seed=3)
    engine.capture_logits(True)
    engine.begin([prompt], spk, sp, max_new_tokens=40, end_of_audio_token=9999, prefill_mode=1)
    engine.run()
    good, good_logits = engine.tokens(0), engine.captured_logits(15, 40)
    try:
        engine.tune(mega=1, mega_fault=8)
        engine.begin([prompt], spk, sp, max_new_tokens=40, end_of_audio_token=9999, prefill_mode=1)
        steps = engine.run(poll_every=16)   # the first chunk aborts, is rolled back and redone by the launches
        assert steps == 39
        assert engine.decode_path() == (0, 1)  # one launch abandoned, the engine stays on the per-layer launches
        assert engine.tokens(0) == good and torch.equal(engine.captured_logits(15, 40), good_logits)
        # the engine stays on the launches for the next three utterances (no further time-out) ...
        for _ in range(3):
            engine.begin([prompt], spk, sp, max_new_tokens=40, end_of_audio_token=9999, prefill_mode=1)
            engine.run()
            assert engine.tokens(0) == good and engine.decode_path() == (0, 1)
        # ... and gives the persistent kernel another chance with the fourth: the fault is still there, so that attempt is abandoned
        # too (same tokens) and the next retry is 8 utterances away
        engine.begin([prompt], spk, sp, max_new_tokens=40, end_of_audio_token=9999, prefill_mode=1)
        engine.run(poll_every=16)
        assert engine.tokens(0) == good and engine.decode_path() == (0, 2)
        engine.tune(mega_fault=0)
        for _ in range(7):
            engine.begin([prompt], spk, sp, max_new_tokens=40, end_of_audio_token=9999, prefill_mode=1)
            engine.run()
            assert engine.decode_path() == (0, 2)
        # the obstacle is gone: the eighth utterance after the second failure runs on the persistent kernel again
        engine.begin([prompt], spk, sp, max_new_tokens=40, end_of_audio_token=9999, prefill_mode=1)
        engine.run()
        assert engine.tokens(0) == good and engine.decode_path() == (2, 2)
    finally:
        engine.tune(mega_fault=0)
    engine.tune(mega=1)  # explicit re-enable clears the sticky fallback
    engine.begin([prompt], spk, sp, max_new_tokens=40, end_of_audio_token=9999, prefill_mode=1)
    engine.run()
    assert engine.tokens(0) == good
    engine.close()


def test_engine_of_eight_next_to_a_competing_kernel_finishes_the_batch():
    """the competing-kernel situation for an engine of 8 utterances.  Large matmuls are queued on another stream right before the
    decode steps; whatever the device does -- the grid of `decode_mfma_kernel<8>` becomes resident in time, or a hand-off wait runs out and the batch is
    rolled back and finished on the per-layer launches (two halves of the rows) -- `mv_run` must return success with every utterance at its full
    length.  Without an abandoned launch the streams are the clean run's, bit for bit; with one they are a valid draw from the launches' logits
    (DESIGN.md 6 "Reproducibility") and only their shape is asserted.  The outcome of every attempt is printed."""
    cfgk = dict(WIDE2, n_layer=3)
    lens = [280, 40, 700, 17, 300, 12, 1030, 90]
    engine, sd, cfg = build(cfgk, 37, max_utterances=8)
    prompts = [synthetic.prompt_tokens(t, seed=510 + u) for u, t in enumerate(lens)]
    spks = torch.cat([synthetic.speaker_embedding(520 + u) for u in range(8)])
    sps = [mv.SamplingParams(guidance_scale=3.0, seed=530 + u, top_p=0.95) for u in range(8)]
    n_new = 48
    engine.begin(prompts, spks, sps, max_new_tokens=n_new, end_of_audio_token=9999, prefill_mode=1)
    engine.run()
    assert engine.decode_path() == (3, 0)
    clean = [engine.tokens(u) for u in range(8)]
    a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
    side = torch.cuda.Stream()
    outcomes = []
    for attempt in range(3):
        with torch.cuda.stream(side):
            for _ in range(12):
                a = (a @ a).clamp_(-1, 1)
        before = engine.decode_path()[1]
        engine.begin(prompts, spks, sps, max_new_tokens=n_new, end_of_audio_token=9999, prefill_mode=1)
        steps = engine.run(poll_every=8)
        assert steps == n_new - 1
        toks = [engine.tokens(u) for u in range(8)]
        assert [len(t) for t in toks] == [t + n_new for t in lens]
        assert all(t[:n] == p.tolist() for t, n, p in zip(toks, lens, prompts))
        path = engine.decode_path()
        outcomes.append(path)
        if path[1] == before:
            assert toks == clean, "no launch was abandoned, yet the streams differ from the clean run"
        side.synchronize()
        engine.tune(mega=1)
    print(f"[competing kernel, engine of 8] (path, abandoned launches so far) per attempt: {outcomes}")
    engine.close()


def test_hand_off_epoch_advances_across_an_utterance_that_ended_mid_chunk():
    """Regression: the sampler used to return before bumping the hand-off epoch once the utterance was done, so
    the trailing graph replays of that mv_run chunk -- and the first launch of the NEXT utterance when its prompt is teacher-
    forced (no sampler launch in mv_begin) -- ran under one epoch and consumed each other's stale granules.  An utterance that
    ends on its end-of-audio token in the middle of a chunk, followed by begin(prefill_mode=0): persistent kernel == launches."""
    engine, sd, cfg = build(dict(WIDE2, n_layer=3), 5)
    prompt, spk = synthetic.prompt_tokens(20, seed=5), synthetic.speaker_embedding(5)
    sp = mv.SamplingParams(top_k=1, seed=9)
    engine.tune(mega=0)
    engine.begin([prompt], spk, sp, max_new_tokens=12, end_of_audio_token=9999, prefill_mode=1)
    engine.run()
    eoa = engine.tokens(0)[20 + 5]  # the 6th generated token becomes the end-of-audio token: the run ends mid-chunk
    prompt2, spk2 = synthetic.prompt_tokens(9, seed=6), synthetic.speaker_embedding(6)
    engine.capture_logits(True)
    out = {}
    for path in ((0, 0), (1, 0), (1, 1)):  # (mega, mega_fold): launches | one step per launch + sample_kernel | tokens drawn inside the launch
        engine.tune(mega=path[0], mega_fold=path[1])
        engine.begin([prompt], spk, sp, max_new_tokens=40, end_of_audio_token=eoa, prefill_mode=1)
        engine.run(poll_every=16)   # mega_fold: the launch leaves its step loop when the control word says "done"
        first = engine.tokens(0)
        assert first[-1] == eoa and len(first) <= 20 + 6
        engine.begin([prompt2], spk2, sp, max_new_tokens=6, end_of_audio_token=9999, prefill_mode=0)  # teacher-forced prompt
        engine.run()
        out[path] = (first, engine.tokens(0), engine.captured_logits(0, 9 + 6 - 1))
    engine.tune(mega=1, mega_fold=1)
    assert engine.decode_path() == (2, 0)
    for path in ((1, 0), (1, 1)):
        assert out[(0, 0)][0] == out[path][0] and out[(0, 0)][1] == out[path][1], path
        assert torch.equal(out[(0, 0)][2], out[path][2]), f"{path}: max {float((out[(0, 0)][2] - out[path][2]).abs().max())}"
    engine.close()


def test_persistent_kernel_next_to_a_competing_kernel_finishes_with_the_same_tokens():
    """The persistent kernel needs all 256 workgroups resident; a plain launch gives no such guarantee.  With another stream keeping
    CUs busy (large matmuls queued right before the decode steps) the run must neither hang nor change its output: either the grid
    became resident in time (no launch abandoned) or a wait ran out, the chunk was rolled back and the per-layer launches finished
    it -- same tokens, same logits either way.  (The fault-injection test above forces the second outcome; this one takes what the
    device gives and prints it.)"""
    engine, sd, cfg = build(dict(WIDE2, n_layer=3), 17)
    prompt, spk = synthetic.prompt_tokens(280, seed=17), synthetic.speaker_embedding(17)
    sp = mv.SamplingParams(top_p=0.95, guidance_scale=3.0, seed=4)
    engine.capture_logits(True)
    engine.tune(mega=0)
    engine.begin([prompt], spk, sp, max_new_tokens=48, end_of_audio_token=9999, prefill_mode=1)
    engine.run()
    want, want_logits = engine.tokens(0), engine.captured_logits(279, 48)
    engine.tune(mega=1)
    a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
    side = torch.cuda.Stream()
    outcomes = []
    for attempt in range(3):
        with torch.cuda.stream(side):
            for _ in range(12):          # ~1.1 TFLOP each: tens of milliseconds of every CU, queued BEFORE the decode steps
                a = (a @ a).clamp_(-1, 1)
        engine.begin([prompt], spk, sp, max_new_tokens=48, end_of_audio_token=9999, prefill_mode=1)
        steps = engine.run(poll_every=8)
        assert steps == 47
        assert engine.tokens(0) == want and torch.equal(engine.captured_logits(279, 48), want_logits)
        outcomes.append(engine.decode_path())
        side.synchronize()
        engine.tune(mega=1)  # re-arm after a fall-back
    print(f"[competing kernel] (path, abandoned launches so far) per attempt: {outcomes}")
    engine.close()
