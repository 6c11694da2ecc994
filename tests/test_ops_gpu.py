"""Single-kernel parity on the GPU, through the C ABI (mv_op_*), against plain torch fp32 restatements of the same op with
the same bf16 rounding points, and (sampler) against the oracle with the same Philox noise."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import load_pkg
from oracle import rng as o_rng
from oracle import stage1 as o_s1

pytestmark = pytest.mark.gpu

N = load_pkg("_native")
eng = load_pkg("engine")


def rb(t):
    return t.to(torch.bfloat16).to(torch.float32)


def dev(t):
    return t.to("cuda").contiguous()


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def bf16_close(got, ref, ulps=2.0, atol=1e-6):
    """|got - ref| <= ulps * (bf16 spacing at |ref|) + atol."""
    ref = ref.float()
    got = got.float()
    spacing = torch.clamp(ref.abs(), min=1e-30) * 2.0 ** -7
    bad = (got - ref).abs() > ulps * spacing + atol
    return int(bad.sum()), float(((got - ref).abs() / (spacing + atol)).max())


def ref_norm(x, w, eps=1e-5):
    xf = x.float()
    return rb(rb(xf * torch.rsqrt((xf * xf).mean(-1, keepdim=True) + eps)) * w.float())


@pytest.mark.parametrize("rows", [2, 4, 8])
@pytest.mark.parametrize("N_,K", [(6144, 2048), (2048, 2048), (2048, 5632), (2562, 2048), (768, 256), (256, 768), (40, 1024)])
@pytest.mark.parametrize("mode", ["plain", "norm", "resid"])
def test_gemv(rows, N_, K, mode):
    g = torch.Generator().manual_seed(N_ * 7 + K + rows)
    x = rb(torch.randn(rows, K, generator=g))
    w = rb(torch.randn(N_, K, generator=g) * 0.05)
    nw = rb(1 + 0.1 * torch.randn(K, generator=g)) if mode == "norm" else None
    res = rb(torch.randn(rows, N_, generator=g)) if mode == "resid" else None
    xin = ref_norm(x, nw) if nw is not None else x
    y = rb(xin.double() @ w.double().T)
    if res is not None:
        y = rb(res + y)
    dx, dw = dev(x.bfloat16()), dev(w.bfloat16())
    dnw = dev(nw.bfloat16()) if nw is not None else None
    dres = dev(res.bfloat16()) if res is not None else None
    out = torch.empty(rows, N_, dtype=torch.bfloat16, device="cuda")
    N.check(N.lib().mv_op_gemv(stream(), rows, N_, K, ptr(dx), ptr(dw), ptr(dnw), 1e-5, ptr(dres), ptr(out)))
    torch.cuda.synchronize()
    nbad, worst = bf16_close(out.cpu(), y, ulps=1.01 if mode != "norm" else 3.0, atol=2e-3 if mode == "norm" else 1e-4)
    assert nbad == 0, f"{nbad} outputs off, worst {worst:.2f} tolerance units"


@pytest.mark.parametrize("rows", [2, 4])
@pytest.mark.parametrize("N_,K", [(5632, 2048), (768, 256), (100, 512)])
def test_swiglu_gemv(rows, N_, K):
    g = torch.Generator().manual_seed(N_ + K)
    x = rb(torch.randn(rows, K, generator=g))
    w1 = rb(torch.randn(N_, K, generator=g) * 0.05)
    w3 = rb(torch.randn(N_, K, generator=g) * 0.05)
    nw = rb(1 + 0.1 * torch.randn(K, generator=g))
    xin = ref_norm(x, nw).double()
    gate, up = rb(xin @ w1.double().T), rb(xin @ w3.double().T)
    y = rb(rb(torch.nn.functional.silu(gate)) * up)
    out = torch.empty(rows, N_, dtype=torch.bfloat16, device="cuda")
    # keep every device tensor referenced until the kernel has run: a temporary would be freed (and its block reused by
    # the next temporary) before the asynchronous launch reads it
    dx, d1, d3, dn = dev(x.bfloat16()), dev(w1.bfloat16()), dev(w3.bfloat16()), dev(nw.bfloat16())
    N.check(N.lib().mv_op_swiglu_gemv(stream(), rows, N_, K, ptr(dx), ptr(d1), ptr(d3), ptr(dn), 1e-5, ptr(out)))
    torch.cuda.synchronize()
    nbad, worst = bf16_close(out.cpu(), y, ulps=4.0, atol=3e-3)
    assert nbad == 0, f"{nbad} outputs off, worst {worst:.2f}"


@pytest.mark.parametrize("n_keys", [[1, 1], [5, 64], [63, 65], [200, 129], [256, 257], [258, 300], [520, 513], [1000, 897], [2048, 2047]])
def test_attention_decode(n_keys):
    H, S, rows = 16, 2048, 4
    g = torch.Generator().manual_seed(sum(n_keys))
    q = rb(torch.randn(rows, H * 128, generator=g))
    k = rb(torch.randn(rows, H, S, 128, generator=g))
    v = rb(torch.randn(rows, H, S, 128, generator=g))
    k[0, 0, 3] *= 8.0  # a dominant key: exercises the running-max rescale across lane groups / waves / splits
    nk = torch.tensor(n_keys, dtype=torch.int32)
    ref = torch.empty(rows, H * 128)
    for r in range(rows):
        P = n_keys[r // 2]
        s = torch.einsum("hd,hsd->hs", q[r].view(H, 128).double(), k[r, :, :P].double()) / np.sqrt(128.0)
        ref[r] = torch.einsum("hs,hsd->hd", torch.softmax(s, -1), v[r, :, :P].double()).reshape(-1).float()
    scratch = torch.zeros(int(N.lib().mv_op_attention_scratch_bytes(rows, H)), dtype=torch.uint8, device="cuda")
    out = torch.empty(rows, H * 128, dtype=torch.bfloat16, device="cuda")
    dnk, dq, dk, dv = dev(nk), dev(q.bfloat16()), dev(k.bfloat16()), dev(v.bfloat16())
    for _ in range(2):  # twice: the in-launch arrival counters must re-arm themselves
        out.zero_()
        N.check(N.lib().mv_op_attention_decode(stream(), rows, H, S, ptr(dnk), ptr(dq), ptr(dk), ptr(dv), ptr(scratch), ptr(out)))
        torch.cuda.synchronize()
        err = (out.cpu().float() - ref).abs().max().item()
        assert err < 2e-2, f"max abs err {err}"
        nbad, worst = bf16_close(out.cpu(), rb(ref), ulps=2.0, atol=2e-3)
        assert nbad == 0, f"{nbad} off, worst {worst:.2f}"


COMBOS = [dict(top_p=None, top_k=None, temperature=1.0), dict(top_p=0.95, top_k=None, temperature=1.0),
          dict(top_p=0.9, top_k=None, temperature=0.7), dict(top_p=None, top_k=50, temperature=1.0),
          dict(top_p=0.95, top_k=200, temperature=1.3), dict(top_p=None, top_k=1, temperature=1.0)]


@pytest.mark.parametrize("combo", COMBOS)
@pytest.mark.parametrize("V", [2562, 300])
def test_sampler_matches_oracle(combo, V):
    g = torch.Generator().manual_seed(V)
    token = torch.zeros(1, dtype=torch.int32, device="cuda")
    probs = torch.zeros(V, dtype=torch.float32, device="cuda")
    mismatch = 0
    for trial in range(6):
        pair = rb(torch.randn(2, 1, V, generator=g) * 2.0)
        sp = eng.SamplingParams(guidance_scale=3.0, seed=1234 + trial, **combo)
        nat = sp.to_native(torch.bfloat16)
        step = 100 + trial
        dpair = dev(pair.view(2, V).bfloat16())
        N.check(N.lib().mv_op_sample(stream(), V, ptr(dpair), C.byref(nat), step, ptr(token), ptr(probs)))
        torch.cuda.synchronize()
        qn = torch.from_numpy(o_rng.exp1_noise(sp.seed, step, V))
        top_p = None if combo["top_p"] is None else nat.top_p
        idx, oprobs = o_s1.sample(pair, nat.guidance_scale, nat.temperature, top_p=top_p, top_k=combo["top_k"], q=qn,
                                  rnd=o_s1._round_bf16)
        got = probs.cpu()
        # the kept set must be identical unless a token sits within 1e-5 of the top-p threshold
        kept_diff = ((got > 0) != (oprobs > 0)).sum().item()
        assert kept_diff <= 1, f"kept sets differ in {kept_diff} tokens"
        if kept_diff == 0:
            assert float((got - oprobs).abs().max()) < 1e-5
            r = oprobs / qn
            top2 = torch.topk(r, 2).values
            if int(token.item()) != idx:
                assert (top2[0] - top2[1]) / top2[0] < 1e-4, "race winner differs outside a numerical dead heat"
                mismatch += 1
    assert mismatch <= 1
