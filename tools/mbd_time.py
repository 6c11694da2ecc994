#!/usr/bin/env python3
"""Time the multi-band diffusion stage at BASELINE size: 12.8 s of audio (960 frames -> 307 200 samples), 4 band models of the
recalled audiocraft basic.yaml sizes (hidden 48, depth 4, growth 4), 20 evaluations each, processor + re_eq."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
syn = importlib.import_module("metavoice-src_amd.synthetic")
mbd = importlib.import_module("metavoice-src_amd.mbd")
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 960
t0 = time.time()
pkg = syn.mbd_package(seed=0)
print(f"package drawn in {time.time() - t0:.1f} s; parameters per band: {sum(v.numel() for v in pkg[0]['model_state'].values()) / 1e6:.1f} M")
g = torch.Generator().manual_seed(0)
lat = torch.randn(1, 128, frames, generator=g)
ref = 0.1 * torch.randn(1, 1, frames * 320, generator=g)
for precision in [int(x) for x in os.environ.get("MBD_PRECISIONS", "1,3").split(",")]:
    t0 = time.time()
    eng = mbd.MultiBandDiffusionBands.from_package(pkg, precision=precision)
    print(f"precision {precision}: loaded in {time.time() - t0:.1f} s")
    for rep in range(int(os.environ.get("MBD_REPS", "2"))):
        t0 = time.time()
        wav = eng.tokens_to_wav(lat, ref)
        dt = time.time() - t0
        import hashlib
        digest = hashlib.sha1(wav.detach().float().cpu().contiguous().numpy().tobytes()).hexdigest()[:12]  # schedule variants must not change a bit
        print(f"  tokens_to_wav({frames} frames): {dt * 1e3:.1f} ms, rms {float(wav.pow(2).mean().sqrt()):.4f}, sha1 {digest}", flush=True)
    eng.close()
