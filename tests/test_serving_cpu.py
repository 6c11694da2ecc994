"""HTTP contract of the serving mirror (reference serving.py:54-109) with a fake TTS: no GPU involved."""
import json
import os
import threading
import urllib.error
import urllib.parse
import urllib.request

import pytest

from conftest import load_pkg

serving = load_pkg("serving")


class FakeTTS:
    def __init__(self, out_dir):
        self.out_dir, self.calls = out_dir, []

    def synthesise(self, text, spk_ref_path, top_p=0.95, guidance_scale=3.0, temperature=1.0):
        if "boom" in text:
            raise RuntimeError("synthetic failure")
        if str(spk_ref_path).endswith(".mp3"):
            raise type("UnsupportedAudioFormat", (ValueError,), {})("no decoder for this speaker reference")
        self.calls.append(dict(text=text, spk=spk_ref_path, top_p=top_p, guidance=guidance_scale,
                               ref_bytes=open(spk_ref_path, "rb").read() if os.path.exists(spk_ref_path) else None))
        path = os.path.join(self.out_dir, f"out{len(self.calls)}.wav")
        with open(path, "wb") as f:
            f.write(b"RIFF0000WAVEfake")
        self.last = path
        return path

    def synthesise_long_stream(self, text, spk_ref_path, top_p=0.95, guidance_scale=3.0, temperature=1.0):
        import torch
        if "boom" in text:
            raise RuntimeError("synthetic failure")
        sentences = [t for t in text.split(". ") if t]
        for i, t in enumerate(sentences):
            if "late" in t:
                raise RuntimeError("fails after the response was committed")
            yield i, len(sentences), torch.full((1, 1, 10 * (i + 1)), 0.25 * (i + 1))


@pytest.fixture()
def server(tmp_path):
    tts = FakeTTS(str(tmp_path))
    srv = serving.make_server(tts, serving.ServingConfig(port=0), host="127.0.0.1")
    t = threading.Thread(target=srv.serve_forever, daemon=True)
    t.start()
    yield f"http://127.0.0.1:{srv.server_address[1]}", tts
    srv.shutdown()


def post(url, data, headers=None):
    req = urllib.request.Request(url, data=data, headers=headers or {}, method="POST")
    try:
        with urllib.request.urlopen(req, timeout=10) as r:
            return r.status, r.headers.get("Content-Type"), r.read()
    except urllib.error.HTTPError as e:
        return e.code, e.headers.get("Content-Type"), e.read()


def test_health_and_tts_contract(server):
    base, tts = server
    with urllib.request.urlopen(base + "/health", timeout=10) as r:
        assert r.status == 200 and json.loads(r.read()) == {"status": "ok"}
    # urlencoded form with a speaker reference path and the default sampling controls
    code, ctype, body = post(base + "/tts", urllib.parse.urlencode({"text": "hello", "speaker_ref_path": "/refs/a.wav"}).encode())
    assert code == 200 and ctype == "audio/wav" and body.startswith(b"RIFF")
    assert tts.calls[-1] == dict(text="hello", spk="/refs/a.wav", top_p=0.95, guidance=3.0, ref_bytes=None)
    assert not os.path.exists(tts.last)  # the output file is unlinked after the response (serving.py:107-109)
    # neither speaker input -> 400
    code, _, body = post(base + "/tts", urllib.parse.urlencode({"text": "hello"}).encode())
    assert code == 400 and b"Either an audio file or a speaker reference path" in body
    # any failure -> 500 with the fixed text
    code, _, body = post(base + "/tts", urllib.parse.urlencode({"text": "boom", "speaker_ref_path": "x"}).encode())
    assert code == 500 and body.decode() == serving.ERROR_TEXT
    # a speaker reference in a format nothing here can decode is the client's problem: 415 with the reason, not the fixed 500 text
    code, ctype, body = post(base + "/tts", urllib.parse.urlencode({"text": "hi", "speaker_ref_path": "/refs/bria.mp3"}).encode())
    assert code == 415 and b"no decoder" in body
    # out-of-range guidance is rejected before synthesis (FastAPI's ge/le validation in the reference)
    code, _, _ = post(base + "/tts", urllib.parse.urlencode({"text": "hi", "speaker_ref_path": "x", "guidance": "9"}).encode())
    assert code == 422


def test_multipart_upload(server):
    base, tts = server
    boundary = "----mvboundary"
    parts = []
    for name, value in (("text", "from upload"), ("guidance", "2.5"), ("top_p", "0.9")):
        parts.append(f'--{boundary}\r\nContent-Disposition: form-data; name="{name}"\r\n\r\n{value}\r\n'.encode())
    parts.append(f'--{boundary}\r\nContent-Disposition: form-data; name="audiodata"; filename="ref.wav"\r\n'
                 f'Content-Type: audio/wav\r\n\r\n'.encode() + b"\x00\x01binary\xff\xfe" + b"\r\n")
    parts.append(f"--{boundary}--\r\n".encode())
    code, ctype, body = post(base + "/tts", b"".join(parts), {"Content-Type": f"multipart/form-data; boundary={boundary}"})
    assert code == 200 and ctype == "audio/wav"
    call = tts.calls[-1]
    assert call["text"] == "from upload" and call["guidance"] == 2.5 and call["top_p"] == 0.9
    assert call["ref_bytes"] == b"\x00\x01binary\xff\xfe"  # the uploaded bytes reached synthesise as a temp file


class FakeBatchTTS:
    """Stands in for `TTS.synthesise_batch`: the 'waveform' of a text is a RIFF-prefixed blob naming the rank that made it."""

    def __init__(self, rank=0, max_utterances=2):
        self.rank, self.max_utterances, self.batches = rank, max_utterances, []

    def check_speaker_reference(self, spk):
        if "missing" in spk:
            raise FileNotFoundError(spk)

    def synthesise_batch(self, texts, spk_refs, top_p=0.95, guidance_scale=3.0, temperature=1.0, return_exceptions=False):
        if any("crash" in t for t in texts):
            raise RuntimeError("the whole call fails")  # something the per-item handling of TTS.synthesise_wavs does not catch
        self.batches.append(list(texts))
        out = []
        for t, s in zip(texts, spk_refs):
            if "boom" in t:
                e = RuntimeError("synthetic failure")
                if not return_exceptions:
                    raise e
                out.append(e)
            else:
                out.append(b"RIFF" + f"|{t}|{s}|{top_p}|{guidance_scale}|rank{self.rank}".encode() * (1 + len(t) % 3))
        return out


def test_batching_dispatcher_groups_concurrent_requests(tmp_path):
    tts = FakeBatchTTS()
    disp = serving.BatchingDispatcher(tts, utterances_per_gpu=4, max_wait_ms=200.0).start()
    srv = serving.make_server(tts, serving.ServingConfig(port=0), host="127.0.0.1", dispatcher=disp)
    t = threading.Thread(target=srv.serve_forever, daemon=True)
    t.start()
    base = f"http://127.0.0.1:{srv.server_address[1]}"
    results = {}

    def call(i, extra=None):
        form = {"text": f"request {i}", "speaker_ref_path": f"/refs/{i}.wav"}
        form.update(extra or {})
        results[i] = post(base + "/tts", urllib.parse.urlencode(form).encode())

    threads = [threading.Thread(target=call, args=(i,)) for i in range(4)]
    threads.append(threading.Thread(target=call, args=(9, {"guidance": "1.5"})))  # different controls: its own batch
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    for i in (0, 1, 2, 3, 9):
        code, ctype, body = results[i]
        assert code == 200 and ctype == "audio/wav" and body.startswith(b"RIFF") and f"|request {i}|/refs/{i}.wav|".encode() in body
    assert b"|1.5|" in results[9][2]
    assert sum(disp.batches) == 5 and max(disp.batches) >= 2  # concurrent requests really were decoded together
    assert all(len({"guid" for _ in b}) == 1 for b in tts.batches)
    code, _, body = post(base + "/tts", urllib.parse.urlencode({"text": "boom", "speaker_ref_path": "x"}).encode())
    assert code == 500 and body.decode() == serving.ERROR_TEXT
    # ONE bad request in a batch fails alone: the requests grouped with it still get their audio (it used to 500 them all)
    results.clear()
    threads = [threading.Thread(target=call, args=(i,)) for i in (20, 21)]
    threads.append(threading.Thread(target=call, args=(22, {"text": "boom 22"})))
    threads.append(threading.Thread(target=call, args=(23, {"speaker_ref_path": "/refs/missing.wav"})))  # refused before dispatch
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert results[20][0] == 200 and results[21][0] == 200 and b"|request 21|" in results[21][2]
    assert results[22][0] == 500 and results[22][2].decode() == serving.ERROR_TEXT
    assert results[23][0] == 500
    assert not any("request 23" in t for b in tts.batches for t in b)
    srv.shutdown()
    disp.stop()


def _dispatch_worker(rank, world, port, q, texts_by_round=None):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    srv_mod = load_pkg("serving")
    tts = FakeBatchTTS(rank=rank)
    disp = srv_mod.BatchingDispatcher(tts, utterances_per_gpu=2, max_wait_ms=300.0)
    if rank != 0:
        disp.serve_worker()
        q.put((rank, tts.batches))
    else:
        disp.start()
        out = {}

        def ask(i, text):
            try:
                out[i] = disp.submit(text, f"/r/{i}.wav", 0.95, 3.0).decode()
            except Exception as e:
                out[i] = f"ERROR {e}"

        for texts in (texts_by_round or [[f"utt {i}" for i in range(4)]]):
            ths = [threading.Thread(target=ask, args=(len(out) + j, t)) for j, t in enumerate(texts)]
            for th in ths:
                th.start()
            for th in ths:
                th.join()
        disp.stop()
        q.put((rank, tts.batches, dict(out), disp.batches))
    dist.destroy_process_group()


def test_dispatcher_shards_a_batch_over_two_ranks():
    """world_size 2 over gloo: rank 0 queues four requests, both ranks decode their round-robin shard, one gather brings the
    bytes back, results reach the right requesters."""
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dispatch_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    r0 = next(g for g in got if g[0] == 0)
    r1 = next(g for g in got if g[0] == 1)
    out = r0[2]
    assert set(out) == {0, 1, 2, 3}
    for i in range(4):
        assert out[i].startswith("RIFF") and f"|utt {i}|/r/{i}.wav|" in out[i]
    by_rank = {i: ("rank0" in out[i], "rank1" in out[i]) for i in out}
    assert sum(a for a, _ in by_rank.values()) + sum(b for _, b in by_rank.values()) == 4
    assert any(b for _, b in by_rank.values()), "rank 1 never produced a result"
    assert sum(len(b) for b in r0[1]) + sum(len(b) for b in r1[1]) == 4


def test_a_failing_request_does_not_hang_the_other_rank():
    """a request that raises on one rank used to make that rank skip the gather the other rank was already
    blocked in.  Now the failure travels through the gather as an error payload: only that request fails, the ranks stay in step
    and the NEXT batch is served normally -- also when a rank's whole `synthesise_batch` call raises."""
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    rounds = [["ok a", "boom b", "ok c", "boom d"], ["crash e", "ok f", "ok g", "ok h"], ["after 1", "after 2"]]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dispatch_worker, args=(r, 2, port, q, rounds)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0, "a rank died or hung"
    out = next(g for g in got if g[0] == 0)[2]
    texts = [t for r in rounds for t in r]
    assert len(out) == len(texts)
    by_text = {}
    for v in out.values():
        for t in texts:
            if f"|{t}|" in v:
                by_text[t] = v
    failed = [v for v in out.values() if v.startswith("ERROR")]
    for t in ("ok a", "ok c", "after 1", "after 2"):
        assert t in by_text and by_text[t].startswith("RIFF")
    assert sum("synthetic failure" in v for v in failed) == 2
    # the crashed call takes down only the items that shared its rank in that batch
    assert 1 <= sum("the whole call fails" in v for v in failed) <= 2
    assert len(failed) + len(by_text) == len(texts)


def test_tts_stream_is_chunked_and_starts_with_an_open_ended_riff_header(server):
    import http.client
    import struct

    base, _ = server
    host, port = base.replace("http://", "").split(":")
    conn = http.client.HTTPConnection(host, int(port), timeout=10)
    conn.request("POST", "/tts_stream", body=urllib.parse.urlencode({"text": "one. two. three", "speaker_ref_path": "/refs/a.wav"}),
                 headers={"Content-Type": "application/x-www-form-urlencoded"})
    r = conn.getresponse()
    assert r.status == 200 and r.getheader("Transfer-Encoding") == "chunked" and r.getheader("Content-Type") == "audio/wav"
    body = r.read()
    assert body[:4] == b"RIFF" and body[8:16] == b"WAVEfmt " and body[4:8] == b"\xff\xff\xff\xff"
    fmt = struct.unpack("<HHIIHH", body[20:36])
    assert fmt == (1, 1, 24000, 48000, 2, 16) and body[36:40] == b"data"
    pcm = struct.unpack(f"<{(len(body) - 44) // 2}h", body[44:])
    assert len(pcm) == 10 + 20 + 30  # three chunks, in order
    assert pcm[0] == round(0.25 * 32767) and pcm[10] == round(0.5 * 32767) and pcm[-1] == round(0.75 * 32767)
    # the same connection serves the next request (HTTP/1.1 keep-alive with explicit lengths everywhere else)
    conn.request("GET", "/health")
    r = conn.getresponse()
    assert r.status == 200 and json.loads(r.read()) == {"status": "ok"}
    conn.close()
    # failures before the first chunk: the contract of /tts
    code, _, body = post(base + "/tts_stream", urllib.parse.urlencode({"text": "boom", "speaker_ref_path": "x"}).encode())
    assert code == 500 and body.decode() == serving.ERROR_TEXT
    code, _, body = post(base + "/tts_stream", urllib.parse.urlencode({"text": "no speaker"}).encode())
    assert code == 400
    # a failure after the response was committed truncates the stream instead of hanging the client
    conn = http.client.HTTPConnection(host, int(port), timeout=10)
    conn.request("POST", "/tts_stream", body=urllib.parse.urlencode({"text": "fine. also fine. late failure", "speaker_ref_path": "x"}),
                 headers={"Content-Type": "application/x-www-form-urlencoded"})
    r = conn.getresponse()
    assert r.status == 200
    try:
        got = r.read()
    except http.client.IncompleteRead as e:
        got = e.partial
    assert got[:4] == b"RIFF" and len(got) == 44 + 2 * (10 + 20)
    conn.close()


def test_cors_preflight_and_body_cap(server, monkeypatch):
    """`serving.py:138-144` (CORSMiddleware: any origin / method / header) and a bound on what a request may make the server
    read into memory."""
    import http.client

    base, tts = server
    host, port = base[len("http://"):].split(":")
    c = http.client.HTTPConnection(host, int(port), timeout=10)
    c.request("OPTIONS", "/tts", headers={"Origin": "http://localhost:3000", "Access-Control-Request-Method": "POST",
                                          "Access-Control-Request-Headers": "x-requested-with"})
    r = c.getresponse()
    r.read()
    assert r.status == 200
    assert r.getheader("Access-Control-Allow-Origin") == "http://localhost:3000"
    assert r.getheader("Access-Control-Allow-Methods") == "POST"
    assert r.getheader("Access-Control-Allow-Headers") == "x-requested-with"
    # the same connection stays usable (HTTP/1.1 keep-alive, Content-Length: 0 on the preflight answer)
    c.request("GET", "/health")
    r = c.getresponse()
    assert r.status == 200 and json.loads(r.read()) == {"status": "ok"}
    c.close()

    monkeypatch.setattr(serving, "MAX_REQUEST_BYTES", 1024)
    n_before = len(tts.calls)
    code, _, body = post(base + "/tts", urllib.parse.urlencode({"text": "x" * 4096, "speaker_ref_path": "a.wav"}).encode())
    assert code == 413 and len(tts.calls) == n_before
    code, _, _ = post(base + "/tts", urllib.parse.urlencode({"text": "ok", "speaker_ref_path": "a.wav"}).encode())
    assert code == 200


def test_uploaded_wave_reference_is_cut_to_two_minutes(server):
    """`serving.py:112-124`: the reference keeps the first 120 s of an uploaded speaker reference (`ffmpeg -t 120`)."""
    import io

    import numpy as np
    from scipy.io import wavfile

    base, tts = server
    rate = 8000
    long_ref = io.BytesIO()
    wavfile.write(long_ref, rate, (np.arange(130 * rate) % 256).astype(np.int16))
    short_ref = io.BytesIO()
    wavfile.write(short_ref, rate, (np.arange(40 * rate) % 256).astype(np.int16))
    for payload, seconds in ((long_ref.getvalue(), 120), (short_ref.getvalue(), 40)):
        boundary = "----mvtest"
        body = (f"--{boundary}\r\nContent-Disposition: form-data; name=\"text\"\r\n\r\nhello\r\n"
                f"--{boundary}\r\nContent-Disposition: form-data; name=\"audiodata\"; filename=\"ref.wav\"\r\nContent-Type: audio/wav\r\n\r\n").encode() \
            + payload + f"\r\n--{boundary}--\r\n".encode()
        code, ctype, _ = post(base + "/tts", body, {"Content-Type": f"multipart/form-data; boundary={boundary}"})
        assert code == 200 and ctype == "audio/wav"
        got_rate, got = wavfile.read(io.BytesIO(tts.calls[-1]["ref_bytes"]))
        assert got_rate == rate and len(got) == seconds * rate
        assert np.array_equal(got, (np.arange(seconds * rate) % 256).astype(np.int16))   # the FIRST two minutes, untouched


def test_an_idle_keep_alive_client_does_not_block_other_clients(server):
    """HTTP/1.1 keeps connections open; with a single-threaded server one idle persistent client blocked every
    other client, /health included.  One thread per connection now, the engine behind one lock."""
    import http.client

    base, tts = server
    host, port = base[len("http://"):].split(":")
    c1 = http.client.HTTPConnection(host, int(port), timeout=10)
    c1.request("GET", "/health")
    r1 = c1.getresponse()
    assert r1.status == 200 and json.loads(r1.read()) == {"status": "ok"}
    # c1 stays open and idle; a second client must be served
    c2 = http.client.HTTPConnection(host, int(port), timeout=5)
    c2.request("GET", "/health")
    assert c2.getresponse().status == 200
    code, ctype, body = post(base + "/tts", urllib.parse.urlencode({"text": "hello", "speaker_ref_path": "/refs/a.wav"}).encode())
    assert code == 200 and body.startswith(b"RIFF")
    # and the first connection is still usable (keep-alive)
    c1.request("GET", "/health")
    assert c1.getresponse().status == 200
    c1.close(); c2.close()
    assert serving._Handler.timeout and serving._Handler.timeout <= 60   # an idle socket cannot hold a thread forever


def test_stream_maps_errors_like_tts(server):
    base, _ = server

    class Unsupported(ValueError):
        pass
    Unsupported.__name__ = "UnsupportedAudioFormat"
    assert serving._error_response(Unsupported("no decoder"))[0] == 415
    # by type along the cause chain, and under the class name a failure on another rank travelled with -- never by message text:
    try:
        try:
            raise Unsupported("no decoder")
        except Unsupported as inner:
            raise RuntimeError("speaker embedding failed") from inner
    except RuntimeError as wrapped:
        code, _, body = serving._error_response(wrapped)
        assert code == 415 and b"speaker embedding failed" not in body   # a wrapper's own text is internal: the fixed detail goes out
    assert serving._error_response(serving.RemoteRequestError("UnsupportedAudioFormat", "x.mp3: no decoder"))[0] == 415
    assert serving._error_response(RuntimeError("cache says: UnsupportedAudioFormat was raised earlier")) == (500, "text/plain", serving.ERROR_TEXT.encode())
    assert serving._error_response(serving.RemoteRequestError("RuntimeError", "UnsupportedAudioFormat quoted"))[0] == 500
    assert serving._error_response(RuntimeError("x")) == (500, "text/plain", serving.ERROR_TEXT.encode())
    code, _, body = post(base + "/tts_stream", urllib.parse.urlencode({"text": "boom", "speaker_ref_path": "x"}).encode())
    assert code == 500 and body.decode() == serving.ERROR_TEXT
