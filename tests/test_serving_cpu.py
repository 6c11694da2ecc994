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
        self.calls.append(dict(text=text, spk=spk_ref_path, top_p=top_p, guidance=guidance_scale,
                               ref_bytes=open(spk_ref_path, "rb").read() if os.path.exists(spk_ref_path) else None))
        path = os.path.join(self.out_dir, f"out{len(self.calls)}.wav")
        with open(path, "wb") as f:
            f.write(b"RIFF0000WAVEfake")
        self.last = path
        return path


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
