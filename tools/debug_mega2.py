import ctypes as C, importlib, os, sys
os.environ["MV_DEBUG_VECTORS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
mv = importlib.import_module("metavoice-src_amd")
N = importlib.import_module("metavoice-src_amd._native")
fiu = importlib.import_module("metavoice-src_amd.fast_inference_utils")
n_layer, stop = 3, int(sys.argv[1]) if len(sys.argv) > 1 else 11
cfgk = dict(n_layer=n_layer, n_head=16, dim=2048, vocab_size=2562, block_size=2048)
sd = fiu.remap_checkpoint_keys(mv.synthetic.stage1_state_dict(seed=7, **cfgk))
e = mv.Stage1Engine(mv.ModelArgs(**cfgk), max_utterances=1)
e.load_state_dict(sd); e.finalize()
T = 16
prompt, spk = mv.synthetic.prompt_tokens(T, seed=7), mv.synthetic.speaker_embedding(7)
dumps = {}
for mega in (1, 0):
    e.tune(mega=mega)
    e.begin([prompt], spk, mv.SamplingParams(top_k=1, seed=21), max_new_tokens=60, end_of_audio_token=9999, prefill_mode=1)
    e.run(max_steps=stop)
    buf = np.zeros(n_layer * 5 * 2 * 5632, dtype=np.uint16)
    N.check(N.lib().mv_debug_vectors(e._h, buf.ctypes.data_as(C.POINTER(C.c_uint16)), buf.size))
    dumps[mega] = torch.from_numpy(buf.astype(np.int32) << 16).view(torch.float32).view(n_layer, 5, 2, 5632)
names = ["q", "attn", "h", "act", "x"]
for l in range(n_layer):
    for v in range(5):
        d = (dumps[1][l, v] - dumps[0][l, v]).abs()
        nz = [(int((d[r] > 0).sum()), float(d[r].max())) for r in range(2)]
        print(f"after {stop} decode steps, layer {l} {names[v]:5s}: differing elems/max per row {nz}", flush=True)
