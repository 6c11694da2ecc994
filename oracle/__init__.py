"""CPU oracle for the MetaVoice-1B synthesis hot path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU restatement (torch-CPU fp32 /
numpy) of the reference algorithms on the `TTS.synthesise` path, written so the
HIP engine's results can be checked.  Only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg may import it, and there only as the
checker / the timed CPU baseline -- never from the product package
(`metavoice-src_amd/`), which must fail loudly when its HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * stage-1 forward, sampling (deterministic part), int4/int8 quantisers,
    token adapters, stage-2 forward, `normalize_text`: PINNED against the
    reference code itself, imported from /root/reference in the build container
    by `tests/golden/make_golden.py`; the resulting vectors are committed under
    `tests/golden/` and re-checked by `tests/test_oracle_golden.py`.
  * stage-2 input packing (`fam/llm/inference.py:264-306` cannot be imported:
    it pulls in audiocraft) and the vocoder: restated from source / from the
    published EnCodec architecture -- PARITY UNPINNED by execution.
"""
