"""int4 group-128 weight-only path (fam/llm/fast_quantize.py) on the GPU: the HIP quantiser must reproduce the reference's
codes / scales / zeros EXACTLY (oracle/quantize.py is pinned to the reference by tests/golden/quant.npz), the int4 GEMV must
match x @ dequant(W)^T, and the quantised engine must follow the oracle run on the dequantised weights."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_pkg
from oracle import quantize as o_q
from oracle import stage1 as o_s1

pytestmark = pytest.mark.gpu

N = load_pkg("_native")
mv = load_pkg()
synthetic = load_pkg("synthetic")


def rb(t):
    return t.to(torch.bfloat16).to(torch.float32)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def hip_quantize(w_bf16):
    n, k = w_bf16.shape
    dw = w_bf16.cuda().contiguous()
    codes = torch.zeros(n, k // 2, dtype=torch.uint8, device="cuda")
    sz = torch.zeros(n, k // 128, dtype=torch.int32, device="cuda")
    deq = torch.zeros(n, k, dtype=torch.bfloat16, device="cuda")
    N.check(N.lib().mv_op_quantize_int4(stream(), n, k, ptr(dw), ptr(codes), ptr(sz), ptr(deq)))
    torch.cuda.synchronize()
    return codes, sz, deq


def unpack(codes, sz):
    c = codes.cpu().to(torch.int32)
    q = torch.stack([c & 15, c >> 4], dim=-1).reshape(c.shape[0], -1)
    s = sz.cpu()
    scale = (s & 0xFFFF).to(torch.int16).view(torch.bfloat16)
    zero = ((s >> 16) & 0xFFFF).to(torch.int16).view(torch.bfloat16)
    return q, scale, zero


@pytest.mark.parametrize("n,k,kind", [(64, 2048, "normal"), (24, 5632, "normal"), (16, 256, "normal"), (32, 2048, "golden"), (8, 2048, "flat")])
def test_quantiser_is_bit_exact(n, k, kind):
    g = torch.Generator().manual_seed(n + k)
    if kind == "golden":
        w = torch.from_numpy(np.load(os.path.join(GOLDEN, "quant.npz"))["w"]).to(torch.bfloat16)
        w = w[:, : (w.shape[1] // 128) * 128]
    elif kind == "flat":  # constant groups exercise the clamp(min=1e-6) branch
        w = torch.full((n, k), 0.0371).to(torch.bfloat16)
        w[1] = 0.0
    else:
        w = (torch.randn(n, k, generator=g) * 0.02).to(torch.bfloat16)
        w[0, :128] *= 40.0
    q_ref, sz_ref = o_q.group_quantize_tensor(w)  # (N, K) int32, (G, N, 2) bf16
    codes, sz, deq = hip_quantize(w)
    q, scale, zero = unpack(codes, sz)
    assert torch.equal(scale, sz_ref[..., 0].T.contiguous()) and torch.equal(zero, sz_ref[..., 1].T.contiguous())
    assert torch.equal(q, q_ref)
    ref_deq = (o_q.group_dequantize(q_ref, sz_ref)).to(torch.bfloat16)  # fp32 (q-8)*s+z, one rounding: the matmul's view
    assert torch.equal(deq.cpu(), ref_deq)


@pytest.mark.parametrize("rows", [2, 4])
@pytest.mark.parametrize("N_,K", [(6144, 2048), (2048, 2048), (2048, 5632), (40, 1024)])
@pytest.mark.parametrize("mode", ["plain", "norm", "resid", "swiglu"])
def test_gemv_int4(rows, N_, K, mode):
    """x @ dequant(W)^T on the int4 GEMV.  Round 5: the kernel applies a group's (scale, zero) to fp32 partial sums instead of rounding every
    dequantised weight to bf16 first (gemv_i4_core.h), so its exact counterpart is `exact` = x @ ((q - 8) s + z)^T with UNROUNDED weights;
    the reference's evaluation (`aten::_weight_int4pack_mm`: one bf16 rounding per weight, fam/llm/fast_quantize.py:204-216) differs from
    that by at most sum |x| |w - bf16(w)| per output, and the test checks both: the kernel within the bf16 output rounding of `exact`, and
    within that hard bound of the reference's."""
    g = torch.Generator().manual_seed(N_ + K + rows)
    x = rb(torch.randn(rows, K, generator=g))
    w = (torch.randn(N_, K, generator=g) * 0.05).to(torch.bfloat16)
    w2 = (torch.randn(N_, K, generator=g) * 0.05).to(torch.bfloat16) if mode == "swiglu" else None
    nw = rb(1 + 0.1 * torch.randn(K, generator=g)) if mode in ("norm", "swiglu") else None
    res = rb(torch.randn(rows, N_, generator=g)) if mode == "resid" else None
    q, sz = o_q.group_quantize_tensor(w)
    xin = x
    if nw is not None:
        xf = x.float()
        xin = rb(rb(xf * torch.rsqrt((xf * xf).mean(-1, keepdim=True) + 1e-5)) * nw)

    def linear(qq, ss, rounded):
        wd = o_q.group_dequantize(qq, ss, out_dtype=torch.float64)
        if rounded:
            wd = wd.to(torch.bfloat16).double()
        return xin.double() @ wd.T

    def slack(qq, ss):   # |x| . |w - bf16(w)|: what the per-weight rounding can move an output by
        wd = o_q.group_dequantize(qq, ss, out_dtype=torch.float64)
        return xin.double().abs() @ (wd - wd.to(torch.bfloat16).double()).abs().T

    refs = {}
    for rounded in (False, True):
        y = rb(linear(q, sz, rounded).float())
        if mode == "swiglu":
            q2, sz2 = o_q.group_quantize_tensor(w2)
            y = rb(rb(torch.nn.functional.silu(y)) * rb(linear(q2, sz2, rounded).float()))
        if res is not None:
            y = rb(res + y)
        refs[rounded] = y
    codes, szd, _ = hip_quantize(w)
    codes2, szd2 = (None, None)
    if w2 is not None:
        codes2, szd2, _ = hip_quantize(w2)
    dx = x.bfloat16().cuda()
    dn = nw.bfloat16().cuda() if nw is not None else None
    dr = res.bfloat16().cuda() if res is not None else None
    out = torch.empty(rows, N_, dtype=torch.bfloat16, device="cuda")
    N.check(N.lib().mv_op_gemv_int4(stream(), rows, N_, K, ptr(dx), ptr(codes), ptr(szd), ptr(codes2), ptr(szd2), ptr(dn), 1e-5, ptr(dr), ptr(out)))
    torch.cuda.synchronize()
    got = out.cpu().float()
    ref = refs[False]
    tol = (3.0 if mode in ("norm", "swiglu") else 1.01) * torch.clamp(ref.abs(), min=1e-30) * 2.0 ** -7 + (float(ref.abs().max()) * 2.0 ** -7 if mode == "resid" else 3e-3)
    bad = (got - ref).abs() > tol
    d_exact, d_ref = float((got - ref).abs().max()), float((got - refs[True]).abs().max())
    print(f"[int4 gemv {mode} rows={rows} {N_}x{K}] max |d| vs unrounded-weight evaluation {d_exact:.4f}, vs the reference's bf16-rounded weights {d_ref:.4f} "
          f"(|y| max {float(ref.abs().max()):.2f})")
    assert int(bad.sum()) == 0, f"{int(bad.sum())} outputs off, worst {float(((got - ref).abs() / tol).max()):.2f}"
    if mode in ("plain", "resid"):   # linear in the weights: the hard bound applies directly (+ one more output rounding)
        bound = tol + slack(q, sz).float() + refs[True].abs() * 2.0 ** -7
        assert bool(((got - refs[True]).abs() <= bound).all())


def test_int4_engine_follows_the_oracle_on_dequantised_weights():
    cfgk = dict(n_layer=2, n_head=16, dim=2048, vocab_size=2562, block_size=2048)
    sd = o_s1.remap_checkpoint_keys(synthetic.stage1_state_dict(seed=1, **cfgk))
    args = mv.ModelArgs(**cfgk)
    engine = mv.Stage1Engine(args, max_utterances=1, quantisation_mode="int4")
    engine.load_state_dict(sd)
    engine.finalize()
    bf16_bytes = sum(v.numel() * 2 for k, v in sd.items())
    assert engine.model_bytes < 0.45 * bf16_bytes  # 2 layers: embeddings and the bf16 LM head weigh in
    # the oracle with every quantised Linear replaced by its dequantised weights (fast_quantize.py:220-252: out % 8 == 0)
    sdq = {}
    for k, v in sd.items():
        if v.dim() == 2 and "embeddings" not in k and o_q.int4_eligible(v.shape[0], v.shape[1], inner_k_tiles=8) and k != "output.weight":
            q, sz = o_q.group_quantize_tensor(v)
            sdq[k] = o_q.group_dequantize(q, sz).to(torch.bfloat16).float()
        else:
            sdq[k] = v.float()
    cfg = o_s1.Stage1Config(**cfgk)
    orc = o_s1.Stage1Oracle(sdq, cfg, rounding="bf16", max_seq_length=64)
    T, n_new = 16, 10
    prompt, spk = synthetic.prompt_tokens(T, seed=1), synthetic.speaker_embedding(1)
    oseq, ologits = o_s1.generate(orc, prompt, spk, max_new_tokens=n_new, end_of_audio_token=9999, top_k=1, seed=5, return_logits=True)
    scale = float(ologits.std())
    engine.capture_logits(True)
    for mode in (0, 1):  # teacher-forced through the int4 decode kernels / MFMA prefill on the dequantised arena
        engine.begin([prompt], spk, mv.SamplingParams(top_k=1, seed=5), max_new_tokens=n_new, end_of_audio_token=9999, prefill_mode=mode)
        engine.run()
        seq = engine.tokens(0)
        got = engine.captured_logits(T - 1, n_new)
        d = float((got[0] - ologits[0]).abs().max())
        same = sum(1 for a, b in zip(seq, oseq) if a == b) - T
        print(f"[int4 prefill_mode={mode}] first-token max|dlogit| {d:.4f} (logit std {scale:.3f}); {same}/{n_new} greedy tokens identical")
        assert d <= 0.06 * scale and same >= n_new // 2
    engine.close()


@pytest.mark.parametrize("quant", ["int4", "int8"])
@pytest.mark.parametrize("n_layer,T,n_new", [(2, 16, 24), (3, 300, 24)])
def test_quantised_persistent_kernel_is_bit_identical_to_the_quantised_launches(n_layer, T, n_new, quant):
    """decode_mega.hip instantiated for int4 / int8 weights shares i4_dots / i8_dots (gemv_i4_core.h, gemv_i8_core.h) with
    gemv_i4.hip / gemv_i8.hip: same bits (int8: the LM head is quantised too)."""
    cfgk = dict(n_layer=n_layer, n_head=16, dim=2048, vocab_size=2562, block_size=2048)
    sd = o_s1.remap_checkpoint_keys(synthetic.stage1_state_dict(seed=2, **cfgk))
    engine = mv.Stage1Engine(mv.ModelArgs(**cfgk), max_utterances=1, quantisation_mode=quant)
    engine.load_state_dict(sd)
    engine.finalize()
    prompt, spk = synthetic.prompt_tokens(T, seed=2), synthetic.speaker_embedding(2)
    engine.capture_logits(True)
    out = {}
    for mega in (1, 0, 1):
        engine.tune(mega=mega)
        engine.begin([prompt], spk, mv.SamplingParams(top_p=0.95, seed=9), max_new_tokens=n_new, end_of_audio_token=9999, prefill_mode=1)
        engine.run()
        res = (engine.tokens(0), engine.captured_logits(T - 1, n_new).clone())
        if mega in out:
            assert res[0] == out[mega][0] and torch.equal(res[1], out[mega][1])
        out[mega] = res
    engine.tune(mega=1)
    # both quantised formats draw their tokens inside the launch (path 2; int8 since the end of round 3: profiles/r03w);
    # no launch may have been abandoned, or the comparison below would be launches against launches
    assert engine.decode_path() == (2, 0)
    assert out[1][0] == out[0][0], f"token streams differ between the {quant} persistent kernel and the {quant} launches"
    assert torch.equal(out[1][1], out[0][1]), f"logits differ: max {float((out[1][1].float() - out[0][1].float()).abs().max())}"
    engine.close()


def test_int8_engine_follows_the_oracle():
    """`quantisation_mode="int8"` (fast_quantize.py:137-198): every Linear incl. the LM head and the speaker projection;
    greedy stream and logits against the oracle evaluating WeightOnlyInt8Linear.forward on the same bf16 checkpoint."""
    for cfgk, T, n_new in ((dict(n_layer=2, n_head=2, dim=256, vocab_size=2562, block_size=64), 10, 16),
                           (dict(n_layer=2, n_head=16, dim=2048, vocab_size=2562, block_size=2048), 16, 8)):
        sd = o_s1.remap_checkpoint_keys(synthetic.stage1_state_dict(seed=2, **cfgk))
        engine = mv.Stage1Engine(mv.ModelArgs(**cfgk), max_utterances=1, quantisation_mode="int8")
        engine.load_state_dict(sd)
        engine.finalize()
        assert engine.model_bytes < 0.62 * sum(v.numel() * 2 for v in sd.values())
        cfg = o_s1.Stage1Config(**cfgk)
        orc = o_s1.Stage1Oracle({k: v.float() for k, v in sd.items()}, cfg, rounding="bf16", max_seq_length=64)
        orc.quantise_int8()
        prompt, spk = synthetic.prompt_tokens(T, seed=2), synthetic.speaker_embedding(2)
        oseq, ologits = o_s1.generate(orc, prompt, spk, max_new_tokens=n_new, end_of_audio_token=9999, top_k=1, seed=5, return_logits=True)
        scale = float(ologits.std())
        engine.capture_logits(True)
        for mode in (0, 1):  # prompt teacher-forced through the int8 decode kernels / MFMA prefill on the codes + row-scale epilogue
            engine.begin([prompt], spk, mv.SamplingParams(top_k=1, seed=5), max_new_tokens=n_new, end_of_audio_token=9999, prefill_mode=mode)
            steps = engine.run()
            assert steps == (T + n_new - 1 if mode == 0 else n_new - 1)  # mode 1: the prompt did not go through decode steps
            seq = engine.tokens(0)
            got = engine.captured_logits(T - 1, n_new)
            d = float((got[0] - ologits[0]).abs().max())
            same = sum(1 for a, b in zip(seq, oseq) if a == b) - T
            print(f"[int8 dim={cfgk['dim']} prefill_mode={mode}] first-token max|dlogit| {d:.4f} (logit std {scale:.3f}); {same}/{n_new} greedy tokens identical")
            assert d <= 0.06 * scale and same >= n_new // 2
        engine.close()
