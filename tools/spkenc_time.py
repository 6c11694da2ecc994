#!/usr/bin/env python3
"""Speaker encoder forward (3 LSTM layers of 256 units over N partial utterances x 160 mel frames, one persistent launch per layer):
time per call and a sha1 of the embeddings (`MV_LSTM_ROWS_IN_LDS=1` = the general kernel with the gate rows in LDS)."""
import hashlib, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
syn = importlib.import_module("metavoice-src_amd.synthetic")
spk = importlib.import_module("metavoice-src_amd.speaker_encoder.model")
enc = spk.SpeakerEncoder(state_dict=syn.speaker_encoder_state_dict(seed=4), verbose=False)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 37   # a 30 s reference: 1.6 s partials, 50 % overlap
mels = torch.rand(n, 160, 40, generator=torch.Generator().manual_seed(12)) * 3.0
for _ in range(4):
    t0 = time.time(); emb = enc(mels); dt = time.time() - t0
    print(f"speaker encoder, {n} partials: {dt * 1e3:.2f} ms, sha1 {hashlib.sha1(emb.detach().float().cpu().contiguous().numpy().tobytes()).hexdigest()[:12]}")
