#!/usr/bin/env python3
"""MFMA vs VALU whole-row attention on the same inputs (tight statistics; the op test's tolerance is 2 bf16 ulps + 2e-3)."""
import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
N = importlib.import_module("metavoice-src_amd._native")
def run(qkv, batch, T, H, hd, causal, valu):
    if valu: os.environ["MV_ATTN_VALU_OP"] = "1"
    else: os.environ.pop("MV_ATTN_VALU_OP", None)
    d = qkv.bfloat16().cuda(); out = torch.empty(batch, T, H * hd, dtype=torch.bfloat16, device="cuda")
    N.check(N.lib().mv_op_attention_rows(C.c_void_p(torch.cuda.current_stream().cuda_stream), batch, T, H, hd, causal, C.c_void_p(d.data_ptr()), C.c_void_p(out.data_ptr())))
    torch.cuda.synchronize(); return out.float().cpu()
for (batch, T, H, hd, causal, amp) in [(2, 32, 2, 64, 0, 1.0), (2, 32, 2, 64, 0, 3.0), (2, 128, 16, 128, 1, 1.0), (1, 1100, 4, 128, 1, 2.0), (2, 2047, 2, 64, 0, 1.0)]:
    g = torch.Generator().manual_seed(T + hd)
    qkv = (torch.randn(batch, T, 3, H, hd, generator=g) * amp).bfloat16().float()
    q, k, v = (qkv[:, :, j].transpose(1, 2).double() for j in range(3))
    s = q @ k.transpose(-1, -2) / np.sqrt(hd)
    if causal: s = s.masked_fill(~torch.tril(torch.ones(T, T, dtype=torch.bool)), float("-inf"))
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(batch, T, H * hd).float()
    a, b = run(qkv, batch, T, H, hd, causal, False), run(qkv, batch, T, H, hd, causal, True)
    rms = float(ref.pow(2).mean().sqrt())
    print(f"batch {batch} T {T} H {H} hd {hd} causal {causal} amp {amp}: |ref| rms {rms:.4f}; mfma vs ref max {float((a - ref).abs().max()):.5f} rms {float((a - ref).pow(2).mean().sqrt()):.6f}; "
          f"valu vs ref max {float((b - ref).abs().max()):.5f} rms {float((b - ref).pow(2).mean().sqrt()):.6f}; mfma vs valu max {float((a - b).abs().max()):.5f}")
