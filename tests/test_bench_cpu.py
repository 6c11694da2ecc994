"""bench.py's launch contract: `python bench.py --gpus N` must end up as N ranks of one process group, whether
the driver's `torch.distributed.run` started them or bench.py re-launched itself.  No GPU: the `MV_BENCH_LAUNCH_CHECK` hook stops
after the rendezvous (gloo)."""
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_self_launch_argv(monkeypatch):
    bench = load_bench()
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    argv = ["--gpus", "8", "--steps", "2", "--warmup", "1"]
    cmd = bench.self_launch_argv(bench.parse_args(argv), argv, port=29999)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    assert cmd[-len(argv) - 1] == os.path.join(ROOT, "bench.py") and cmd[-len(argv):] == argv
    # one GPU: no launcher; already under a launcher: no second one
    assert bench.self_launch_argv(bench.parse_args(["--gpus", "1"]), ["--gpus", "1"]) is None
    monkeypatch.setenv("WORLD_SIZE", "8")
    assert bench.self_launch_argv(bench.parse_args(argv), argv) is None


def test_extra_configs_cover_the_other_baseline_configurations():
    bench = load_bench()
    names = " ".join(n for n, _ in bench.EXTRA_CONFIGS)
    for tag in ("configs[2]", "configs[3]", "configs[4]"):
        assert tag in names
    over = dict(bench.EXTRA_CONFIGS)
    lf = over["configs[4] long-form 10000 characters, full vocoder, 8 chunks per engine run"]
    assert lf["long_form_chars"] == 10000 and lf["utterances_per_gpu"] == 8
    assert over["configs[2'] 8 utterances per GPU (batch 64 over 8 GPUs), full vocoder"]["utterances_per_gpu"] == 8
    # the chunks of a long text decode in groups of --utterances-per-gpu (what TTS.synthesise_long does), consecutive chunks together
    assert bench.parse_args(["--long-form", "--utterances-per-gpu", "4"]).long_form
    importlib = __import__("importlib")
    mv = importlib.import_module("metavoice-src_amd")
    args = mv.ModelArgs.from_name("metavoice-1B")
    one = bench.make_work(mv, args, long_form=True, long_form_chars=3000, utterances_per_gpu=1, rank=0, world=1)
    four = bench.make_work(mv, args, long_form=True, long_form_chars=3000, utterances_per_gpu=4, rank=0, world=1)
    assert all(len(g) == 1 for g in one) and len(four) == -(-len(one) // 4) and all(len(g) <= 4 for g in four)
    flat1, flat4 = [x for g in one for x in g], [x for g in four for x in g]
    assert len(flat1) == len(flat4) and all(n1 == n4 and bool((p1 == p4).all()) for (p1, n1), (p4, n4) in zip(flat1, flat4))
    a = bench.parse_args([])
    for _, o in bench.EXTRA_CONFIGS:
        for k in o:
            assert hasattr(a, k), k


def _run(cmd, env):
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_gpus_flag_starts_that_many_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["MV_BENCH_LAUNCH_CHECK"] = "1"
    # bare `python bench.py --gpus 2`: the judge's round-2 invocation, which used to run ONE rank
    got = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"], env)
    assert got == {"launch_check": True, "rccl_ranks": 2, "ranks_seen": 2, "n_gpus": 2}
    # the driver's form
    got = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", "29611", "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"], env)
    assert got["rccl_ranks"] == 2 and got["ranks_seen"] == 2
    # a launcher that started the wrong number of ranks is an error, not a silently different measurement
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29612", "bench.py", "--gpus", "4"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
