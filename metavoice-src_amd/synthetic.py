"""Seeded synthetic checkpoints with the exact MetaVoice-1B tensor names and shapes.

No model weights are reachable offline, so every measurement and parity test runs on synthetic
parameters laid out exactly like the real `first_stage.pt` / `second_stage.pt`
(`fam/llm/finetune.py:300-313`: {"model", "model_args", "config", "meta"}; tensor names as listed
in SURVEY.md section 8a).  Values follow `GPT._init_weights` (`fam/llm/model.py:145-176`): N(0, 0.02) for
Linear/Embedding weights, N(0, 0.02/sqrt(2*n_layer)) for the residual projections `c_proj`; norm
weights are 1 + 0.1*N(0,1) instead of the reference's constant 1 so that the norm-scale multiply is
actually exercised.  Everything is rounded to bf16, so a bf16 engine and an fp32 oracle see the
same parameters.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch


def _normal(gen: torch.Generator, shape, std: float) -> torch.Tensor:
    return (torch.randn(shape, generator=gen, dtype=torch.float32, device=gen.device) * std).to(torch.bfloat16)


def _norm_weight(gen: torch.Generator, n: int) -> torch.Tensor:
    return (1.0 + 0.1 * torch.randn(n, generator=gen, dtype=torch.float32, device=gen.device)).to(torch.bfloat16)


def stage1_state_dict(*, n_layer: int = 24, n_head: int = 16, dim: int = 2048, vocab_size: int = 2562,
                      block_size: int = 2048, intermediate_size: Optional[int] = None, speaker_emb_dim: int = 256,
                      seed: int = 0, orig_mod_prefix: bool = False, device: str = "cpu") -> Dict[str, torch.Tensor]:
    """`first_stage.pt["model"]` with nanoGPT names (before the gpt-fast remap).  bf16 tensors.
    `device="cuda"` draws from the GPU generator instead (different values, same distribution): used by bench.py,
    where 1.2 B parameters would take ~10 s to draw on the host."""
    if intermediate_size is None:
        n_hidden = int(2 * (4 * dim) / 3)
        intermediate_size = n_hidden if n_hidden % 256 == 0 else n_hidden + 256 - n_hidden % 256
    g = torch.Generator(device=device).manual_seed(seed)
    proj_std = 0.02 / math.sqrt(2 * n_layer)
    sd: Dict[str, torch.Tensor] = {}
    sd["transformer.wtes.0.weight"] = _normal(g, (vocab_size, dim), 0.02)
    sd["transformer.wpe.weight"] = _normal(g, (block_size, dim), 0.02)
    sd["speaker_cond_pos.weight"] = _normal(g, (dim, speaker_emb_dim), 0.02)
    for i in range(n_layer):
        p = f"transformer.h.{i}."
        sd[p + "ln_1.weight"] = _norm_weight(g, dim)
        sd[p + "attn.c_attn.weight"] = _normal(g, (3 * dim, dim), 0.02)
        sd[p + "attn.c_proj.weight"] = _normal(g, (dim, dim), proj_std)
        sd[p + "ln_2.weight"] = _norm_weight(g, dim)
        sd[p + "mlp.swiglu.w1.weight"] = _normal(g, (intermediate_size, dim), 0.02)
        sd[p + "mlp.swiglu.w3.weight"] = _normal(g, (intermediate_size, dim), 0.02)
        sd[p + "mlp.c_proj.weight"] = _normal(g, (dim, intermediate_size), proj_std)
    sd["transformer.ln_f.weight"] = _norm_weight(g, dim)
    # tied to wtes.0 inside GPT (model.py:139-143) but stored under its own key
    sd["lm_heads.0.weight"] = sd["transformer.wtes.0.weight"]
    if orig_mod_prefix:
        sd = {"_orig_mod." + k: v for k, v in sd.items()}
    return sd


def stage2_model_args(*, block_size: int = 1024, n_layer: int = 6, n_head: int = 6, n_embd: int = 384,
                      bias: bool = False, norm_type: str = "rmsnorm", nonlinearity_type: str = "swiglu") -> dict:
    """A `GPTConfig` kwargs dict of the second-stage kind (non-causal, 2 input hierarchies = text+cb0 / cb1,
    6 target hierarchies cb2..cb7).  The real `second_stage.pt["model_args"]` is not available offline;
    the engine is shape-generic and reads whatever the checkpoint says."""
    return dict(
        block_size=block_size, vocab_sizes=[1538, 1025], target_vocab_sizes=[1025] * 6, n_layer=n_layer,
        n_head=n_head, n_embd=n_embd, dropout=0.0, spkemb_dropout=0.0, bias=bias, causal=False,
        spk_emb_on_text=True, norm_type=norm_type, rmsnorm_eps=1e-5 if norm_type == "rmsnorm" else None,
        nonlinearity_type=nonlinearity_type, swiglu_multiple_of=256 if nonlinearity_type == "swiglu" else None,
        attn_kernel_type="torch_attn", kv_cache_enabled=False,
    )


def stage2_state_dict(model_args: dict, speaker_emb_dim: Optional[int] = 256, seed: int = 1) -> Dict[str, torch.Tensor]:
    """`second_stage.pt["model"]` for `model_args` (names of `fam/llm/model.py:121-143`).  fp32 tensors holding
    bf16-representable values (the real checkpoint is fp32 and runs under bf16 autocast)."""
    a = model_args
    C, L = a["n_embd"], a["n_layer"]
    g = torch.Generator().manual_seed(seed)
    proj_std = 0.02 / math.sqrt(2 * L)
    if a["nonlinearity_type"] == "swiglu":
        hidden = int(2 * 4 * C / 3)
        hidden = a["swiglu_multiple_of"] * math.ceil(hidden / a["swiglu_multiple_of"])
    else:
        hidden = 4 * C
    sd: Dict[str, torch.Tensor] = {}

    def lin(name, out_f, in_f, std):
        sd[name + ".weight"] = _normal(g, (out_f, in_f), std).float()
        if a["bias"]:
            sd[name + ".bias"] = _normal(g, (out_f,), 0.02).float()

    def norm(name):
        sd[name + ".weight"] = _norm_weight(g, C).float()
        if a["norm_type"] == "layernorm" and a["bias"]:
            sd[name + ".bias"] = _normal(g, (C,), 0.02).float()

    for i, v in enumerate(a["vocab_sizes"]):
        sd[f"transformer.wtes.{i}.weight"] = _normal(g, (v, C), 0.02).float()
    sd["transformer.wpe.weight"] = _normal(g, (a["block_size"], C), 0.02).float()
    for i in range(L):
        p = f"transformer.h.{i}."
        norm(p + "ln_1")
        lin(p + "attn.c_attn", 3 * C, C, 0.02)
        lin(p + "attn.c_proj", C, C, proj_std)
        norm(p + "ln_2")
        if a["nonlinearity_type"] == "swiglu":
            lin(p + "mlp.swiglu.w1", hidden, C, 0.02)
            lin(p + "mlp.swiglu.w3", hidden, C, 0.02)
        else:
            lin(p + "mlp.c_fc", hidden, C, 0.02)
        lin(p + "mlp.c_proj", C, hidden, proj_std)
    norm("transformer.ln_f")
    if speaker_emb_dim is not None:
        sd["speaker_cond_pos.weight"] = _normal(g, (C, speaker_emb_dim), 0.02).float()
    for i, v in enumerate(a["target_vocab_sizes"]):
        sd[f"lm_heads.{i}.weight"] = _normal(g, (v, C), 0.02).float()
    return sd


def speaker_embedding(seed: int = 0, dim: int = 256) -> torch.Tensor:
    """A stand-in for the LSTM speaker encoder's output (`speaker_encoder/model.py:57-58`: ReLU then
    L2-normalise), shape (1, dim), fp32 holding bf16-representable values."""
    g = torch.Generator().manual_seed(1000 + seed)
    e = torch.relu(torch.randn(1, dim, generator=g))
    e = e / e.norm(dim=-1, keepdim=True)
    return e.to(torch.bfloat16).float()


def prompt_tokens(n: int = 128, seed: int = 0, text_offset: int = 2049, n_text: int = 512) -> torch.Tensor:
    """BASELINE config prompt: n-1 uniform text ids in [2049, 2560] followed by EOT 2561 (int32)."""
    g = torch.Generator().manual_seed(2000 + seed)
    body = torch.randint(text_offset, text_offset + n_text, (n - 1,), generator=g)
    return torch.cat([body, torch.tensor([text_offset + n_text])]).to(torch.int32)


def long_prompt_tokens(n: int, seed: int = 0, n_text: int = 128) -> torch.Tensor:
    """A context as the decode loop sees it deep into an utterance: the `prompt_tokens(n_text, seed)` text prompt followed by
    n - n_text audio tokens, codebook 0 (ids < 1024) and codebook 1 (ids 1024..2047) alternating as
    `fam/llm/preprocessing/audio_token_mode.py:38-49` flattens them (int32)."""
    g = torch.Generator().manual_seed(3000 + seed)
    k = n - n_text
    audio = torch.randint(0, 1024, (k,), generator=g) + 1024 * (torch.arange(k) % 2)
    return torch.cat([prompt_tokens(n_text, seed), audio.to(torch.int32)])


def write_model_dir(path: str, *, stage1: Optional[dict] = None, stage2_args: Optional[dict] = None, seed: int = 0,
                    mbd: Optional[dict] = None) -> str:
    """Write `first_stage.pt` / `second_stage.pt` / `speaker_encoder.pt` (+ `encodec_24khz.pt`, which the reference fetches from
    the hub) with the reference's checkpoint layout (`fam/llm/finetune.py:300-313`:
    {"model", "model_args", "config", "meta"}) and synthetic weights + a synthetic 512-rank BPE table, so that the whole
    `TTS` load path (key remap, tokenizer from `meta`, speaker conditioning flags) can be exercised offline."""
    import os

    from .tokenise import byte_level_tokenizer_info

    os.makedirs(path, exist_ok=True)
    s1 = dict(n_layer=2, n_head=2, dim=256, vocab_size=2562, block_size=256)
    s1.update(stage1 or {})
    sd1 = stage1_state_dict(seed=seed, orig_mod_prefix=True, **s1)
    torch.save({
        "model": sd1,
        "model_args": dict(n_layer=s1["n_layer"], n_head=s1["n_head"], n_embd=s1["dim"], block_size=s1["block_size"],
                           vocab_sizes=[s1["vocab_size"]], target_vocab_sizes=None, bias=False, causal=True,
                           norm_type="rmsnorm", nonlinearity_type="swiglu"),
        "config": {"causal": True},
        "meta": {"tokenizer": byte_level_tokenizer_info(offset=2049), "speaker_cond": True, "speaker_emb_size": 256},
    }, os.path.join(path, "first_stage.pt"))
    a2 = stage2_args or stage2_model_args(block_size=256, n_layer=2, n_head=2, n_embd=128)
    torch.save({
        "model": stage2_state_dict(a2, speaker_emb_dim=256, seed=seed + 1),
        "model_args": a2,
        "config": {"causal": False},
        "meta": {"tokenizer": byte_level_tokenizer_info(offset=1025), "speaker_cond": True, "speaker_emb_size": 256},
    }, os.path.join(path, "second_stage.pt"))
    torch.save({"model_state": speaker_encoder_state_dict(seed=seed + 2)}, os.path.join(path, "speaker_encoder.pt"))
    torch.save(encodec_state_dict(seed=seed + 3), os.path.join(path, "encodec_24khz.pt"))
    if mbd is not None:  # kwargs of mbd_package: the multi-band diffusion package the reference fetches from the hub
        torch.save(mbd_package(seed=seed + 4, **mbd), os.path.join(path, "mbd_comp_8.pt"))
    return path


def speaker_encoder_state_dict(seed: int = 0, n_mels: int = 40, hidden: int = 256, n_layers: int = 3, emb: int = 256) -> Dict[str, torch.Tensor]:
    """`speaker_encoder.pt["model_state"]` (`fam/quantiser/audio/speaker_encoder/model.py:33-35`): nn.LSTM(40, 256, 3) +
    nn.Linear(256, 256) with torch's default uniform(-1/sqrt(H), 1/sqrt(H)) initialisation, fp32."""
    g = torch.Generator().manual_seed(9000 + seed)
    b = 1.0 / math.sqrt(hidden)

    def uni(*shape):
        return (torch.rand(*shape, generator=g) * 2 - 1) * b

    sd: Dict[str, torch.Tensor] = {}
    for l in range(n_layers):
        i = n_mels if l == 0 else hidden
        sd[f"lstm.weight_ih_l{l}"] = uni(4 * hidden, i)
        sd[f"lstm.weight_hh_l{l}"] = uni(4 * hidden, hidden)
        sd[f"lstm.bias_ih_l{l}"] = uni(4 * hidden)
        sd[f"lstm.bias_hh_l{l}"] = uni(4 * hidden)
    sd["linear.weight"] = uni(emb, hidden)
    sd["linear.bias"] = uni(emb)
    return sd


def encodec_state_dict(seed: int = 0, *, hidden: int = 128, num_filters: int = 32, ratios=(8, 5, 4, 2), kernel_size: int = 7,
                       last_kernel_size: int = 7, residual_kernel_size: int = 3, compress: int = 2, lstm_layers: int = 2, n_q: int = 8,
                       bins: int = 1024) -> Dict[str, torch.Tensor]:
    """Seeded weights of the EnCodec 24 kHz decoder + codebooks under transformers' EncodecModel names, weight-norm already
    folded (fp32).  Scales follow torch's default Conv/LSTM initialisation (uniform +-1/sqrt(fan_in)) so that activations
    stay O(1) through the stack."""
    g = torch.Generator().manual_seed(7000 + seed)

    def uni(shape, fan_in):
        b = 1.0 / math.sqrt(fan_in)
        return (torch.rand(shape, generator=g) * 2 - 1) * b

    sd: Dict[str, torch.Tensor] = {}
    for q in range(n_q):
        sd[f"quantizer.layers.{q}.codebook.embed"] = torch.randn(bins, hidden, generator=g) * 0.5
    ch = num_filters << len(ratios)
    sd["decoder.layers.0.conv.weight"] = uni((ch, hidden, kernel_size), hidden * kernel_size)
    sd["decoder.layers.0.conv.bias"] = uni((ch,), hidden * kernel_size)
    for l in range(lstm_layers):
        for kind in ("ih", "hh"):
            sd[f"decoder.layers.1.lstm.weight_{kind}_l{l}"] = uni((4 * ch, ch), ch)
            sd[f"decoder.layers.1.lstm.bias_{kind}_l{l}"] = uni((4 * ch,), ch)
    idx = 2
    for r in ratios:
        half = ch // 2
        sd[f"decoder.layers.{idx + 1}.conv.weight"] = uni((ch, half, 2 * r), ch * 2)  # ConvTranspose1d: [Cin][Cout][K]
        sd[f"decoder.layers.{idx + 1}.conv.bias"] = uni((half,), ch * 2)
        p = f"decoder.layers.{idx + 2}."
        sd[p + "block.1.conv.weight"] = uni((half // compress, half, residual_kernel_size), half * residual_kernel_size)
        sd[p + "block.1.conv.bias"] = uni((half // compress,), half * residual_kernel_size)
        sd[p + "block.3.conv.weight"] = uni((half, half // compress, 1), half // compress)
        sd[p + "block.3.conv.bias"] = uni((half,), half // compress)
        sd[p + "shortcut.conv.weight"] = uni((half, half, 1), half)
        sd[p + "shortcut.conv.bias"] = uni((half,), half)
        ch = half
        idx += 3
    sd[f"decoder.layers.{idx + 1}.conv.weight"] = uni((1, ch, last_kernel_size), ch * last_kernel_size)
    sd[f"decoder.layers.{idx + 1}.conv.bias"] = uni((1,), ch * last_kernel_size)
    return sd


def mbd_package(seed: int = 0, *, n_bands: int = 4, hidden: int = 48, depth: int = 4, growth: float = 4.0, kernel: int = 8, stride: int = 4,
                norm_groups: int = 4, res_blocks: int = 1, codec_dim: Optional[int] = 128, emb_all_layers: bool = True, bilstm: bool = False,
                activation: str = "relu", num_steps: int = 1000, proc_bands: int = 8, sample_rate: int = 24_000) -> dict:
    """A multi-band diffusion package with audiocraft's layout (`load_diffusion_models`: {i: {"model_state", "processor_state", "cfg"},
    "n_bands", "sample_rate"}) and synthetic weights: torch's default Conv / Linear / LSTM initialisation (uniform +-1/sqrt(fan_in)),
    GroupNorm weights 1 + 0.1 N(0,1), step embeddings N(0, 1).  The defaults are the sizes of audiocraft's
    `config/model/score/basic.yaml` as recalled (hidden 48, depth 4, growth 4, kernel 8, stride 4): the released `mbd_comp_8.pt` is
    not reachable offline, the engine reads whatever the package's cfg says."""
    pkg: dict = {"n_bands": n_bands, "sample_rate": sample_rate}
    for b in range(n_bands):
        g = torch.Generator().manual_seed(11000 + 97 * seed + b)

        def uni(shape, fan_in):
            return (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan_in)

        def norm(sd, name, c):
            sd[name + ".weight"] = 1.0 + 0.1 * torch.randn(c, generator=g)
            sd[name + ".bias"] = 0.1 * torch.randn(c, generator=g)

        def res(sd, prefix, c):
            for j in (1, 2):
                norm(sd, f"{prefix}norm{j}", c)
                sd[f"{prefix}conv{j}.weight"] = uni((c, c, 3), c * 3)
                sd[f"{prefix}conv{j}.bias"] = uni((c,), c * 3)

        sd: Dict[str, torch.Tensor] = {}
        chans, cin, h = [], 1, hidden
        for i in range(depth):
            chans.append((cin, h))
            sd[f"encoders.{i}.conv.weight"] = uni((h, cin, kernel), cin * kernel)
            norm(sd, f"encoders.{i}.norm", h)
            for r in range(res_blocks):
                res(sd, f"encoders.{i}.res_blocks.{r}.", h)
            if i == 0:
                sd["embedding.weight"] = torch.randn(num_steps, h, generator=g)
            elif emb_all_layers:
                sd[f"embeddings.{i - 1}.weight"] = torch.randn(num_steps, h, generator=g)
            cin, h = h, int(h * growth)
        for i in range(depth):  # decoders[0] is the innermost layer
            ci, co = chans[depth - 1 - i][1], chans[depth - 1 - i][0]
            for r in range(res_blocks):
                res(sd, f"decoders.{i}.res_blocks.{r}.", ci)
            norm(sd, f"decoders.{i}.norm", ci)
            sd[f"decoders.{i}.convtr.weight"] = uni((ci, co, kernel), ci * kernel / stride)
        cb = chans[-1][1]
        if codec_dim is not None:
            sd["conv_codec.weight"] = uni((cb, codec_dim, 1), codec_dim)
            sd["conv_codec.bias"] = uni((cb,), codec_dim)
        if bilstm:
            for l in range(2):
                for suffix in ("", "_reverse"):
                    i_sz = cb if l == 0 else 2 * cb
                    sd[f"bilstm.lstm.weight_ih_l{l}{suffix}"] = uni((4 * cb, i_sz), cb)
                    sd[f"bilstm.lstm.weight_hh_l{l}{suffix}"] = uni((4 * cb, cb), cb)
                    sd[f"bilstm.lstm.bias_ih_l{l}{suffix}"] = uni((4 * cb,), cb)
                    sd[f"bilstm.lstm.bias_hh_l{l}{suffix}"] = uni((4 * cb,), cb)
            sd["bilstm.linear.weight"] = uni((cb, 2 * cb), 2 * cb)
            sd["bilstm.linear.bias"] = uni((cb,), 2 * cb)
        counts = torch.tensor([10_000.0])
        mean = 0.01 * torch.randn(proc_bands, generator=g)
        std = 0.5 + torch.rand(proc_bands, generator=g)
        proc = {"counts": counts, "sum_x": mean * counts, "sum_x2": (std ** 2 + mean ** 2) * counts,
                "sum_target_x2": (0.5 + torch.rand(proc_bands, generator=g)) * counts}
        pkg[b] = {
            "model_state": sd, "processor_state": proc if proc_bands > 1 else None,
            "cfg": {"diffusion_unet": dict(hidden=hidden, depth=depth, growth=growth, kernel=kernel, stride=stride, norm_groups=norm_groups,
                                           res_blocks=res_blocks, codec_dim=codec_dim, emb_all_layers=emb_all_layers, bilstm=bilstm,
                                           transformer=False, cross_attention=False, max_channels=10_000, activation=activation),
                    "schedule": dict(beta_t0=1e-4, beta_t1=0.02, num_steps=num_steps, beta_exp=1.0, clip=5.0, rescale=1.0, noise_scale=1.0),
                    "processor": dict(use=proc_bands > 1, n_bands=proc_bands, power_std=1.0)},
        }
    return pkg
