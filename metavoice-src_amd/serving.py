"""Mirror of the reference's `serving.py` HTTP surface on the MI355X engine.

Same contract (`serving.py:54-109`): `GET /health -> {"status": "ok"}`; `POST /tts` with form fields `text` (required),
`speaker_ref_path` (URL / path) xor `audiodata` (uploaded file), `guidance` in [0, 5] (default 3.0), `top_p` in [0, 1]
(default 0.95) -> `audio/wav` body; 400 when neither speaker input is given; 500 with the reference's fixed text on any
failure; the synthesised file is deleted after the response.  Same flags (`ServingConfig`, `:29-42`):
--huggingface_repo_id --temperature --seed --port (58003) --quantisation_mode.  Requests are handled one at a time, like
the reference (its `async def` handler calls the blocking `synthesise`, `:89`).

The reference builds this on FastAPI + python-multipart + tyro + ffmpeg; of those only FastAPI is in this image and its
`Form(...)` needs python-multipart, so the same surface is served here with the standard library (`http.server` + `email`
for multipart bodies).  Uploaded `audiodata` is written to a temp file and handed to `TTS.synthesise` as the speaker
reference; transcoding arbitrary formats through ffmpeg (`:112-124`) is out of scope.
"""
from __future__ import annotations

import argparse
import json
import logging
import tempfile
from dataclasses import dataclass
from email import message_from_bytes
from http.server import BaseHTTPRequestHandler, HTTPServer
from pathlib import Path
from typing import Dict, Optional, Tuple
from urllib.parse import parse_qs

logger = logging.getLogger(__name__)

ERROR_TEXT = "Something went wrong. Please try again in a few mins or contact us on Discord"  # serving.py:103


@dataclass
class ServingConfig:
    """`serving.py:29-42`."""

    huggingface_repo_id: str = "metavoiceio/metavoice-1B-v0.1"
    temperature: float = 1.0
    seed: int = 1337
    port: int = 58003
    quantisation_mode: Optional[str] = None


class _GlobalState:
    config: ServingConfig
    tts = None


GlobalState = _GlobalState()


def _parse_form(content_type: str, body: bytes) -> Tuple[Dict[str, str], Dict[str, bytes]]:
    """-> (text fields, uploaded files) for urlencoded and multipart/form-data bodies."""
    fields: Dict[str, str] = {}
    files: Dict[str, bytes] = {}
    if content_type.startswith("multipart/form-data"):
        msg = message_from_bytes(b"Content-Type: " + content_type.encode() + b"\r\nMIME-Version: 1.0\r\n\r\n" + body)
        for part in msg.walk():
            if part.is_multipart():
                continue
            name = part.get_param("name", header="content-disposition")
            if name is None:
                continue
            payload = part.get_payload(decode=True) or b""
            if part.get_filename() is not None:
                files[name] = payload
            else:
                fields[name] = payload.decode("utf-8")
    else:
        fields = {k: v[0] for k, v in parse_qs(body.decode("utf-8"), keep_blank_values=True).items()}
    return fields, files


def text_to_speech(fields: Dict[str, str], files: Dict[str, bytes]) -> Tuple[int, str, bytes]:
    """The `/tts` handler body (`serving.py:59-109`) -> (status, content type, body)."""
    text = fields.get("text")
    if text is None:
        return 422, "application/json", json.dumps({"detail": "field 'text' is required"}).encode()
    try:
        guidance = float(fields.get("guidance", 3.0))
        top_p = float(fields.get("top_p", 0.95))
    except ValueError:
        return 422, "application/json", json.dumps({"detail": "guidance / top_p must be numbers"}).encode()
    if not (0.0 <= guidance <= 5.0) or not (0.0 <= top_p <= 1.0):
        return 422, "application/json", json.dumps({"detail": "guidance must be in [0, 5] and top_p in [0, 1]"}).encode()
    speaker_ref_path = fields.get("speaker_ref_path")
    audiodata = files.get("audiodata")
    if not audiodata and not speaker_ref_path:
        return 400, "application/json", json.dumps(
            {"detail": "Either an audio file or a speaker reference path must be provided."}).encode()
    wav_out_path = None
    try:
        with tempfile.NamedTemporaryFile(suffix=".wav") as wav_tmp:
            if speaker_ref_path is None:
                wav_tmp.write(audiodata)
                wav_tmp.flush()
                wav_path = wav_tmp.name
            else:
                wav_path = speaker_ref_path
            wav_out_path = GlobalState.tts.synthesise(text=text, spk_ref_path=wav_path, top_p=top_p, guidance_scale=guidance)
        with open(wav_out_path, "rb") as f:
            return 200, "audio/wav", f.read()
    except Exception:
        logger.exception(f"Error processing request. text: {text}, speaker_ref_path: {speaker_ref_path}, guidance: {guidance}, top_p: {top_p}")
        return 500, "text/plain", ERROR_TEXT.encode()
    finally:
        if wav_out_path is not None:
            Path(wav_out_path).unlink(missing_ok=True)


class _Handler(BaseHTTPRequestHandler):
    def _send(self, status: int, ctype: str, body: bytes):
        self.send_response(status)
        self.send_header("Content-Type", ctype)
        self.send_header("Content-Length", str(len(body)))
        self.send_header("Access-Control-Allow-Origin", "*")  # CORS: serving.py:139-145
        self.end_headers()
        self.wfile.write(body)

    def do_GET(self):
        if self.path.split("?")[0] == "/health":
            self._send(200, "application/json", json.dumps({"status": "ok"}).encode())
        else:
            self._send(404, "application/json", b'{"detail": "Not Found"}')

    def do_POST(self):
        if self.path.split("?")[0] != "/tts":
            self._send(404, "application/json", b'{"detail": "Not Found"}')
            return
        body = self.rfile.read(int(self.headers.get("Content-Length", 0)))
        fields, files = _parse_form(self.headers.get("Content-Type", "application/x-www-form-urlencoded"), body)
        self._send(*text_to_speech(fields, files))

    def log_message(self, fmt, *args):
        logger.info("%s - %s", self.address_string(), fmt % args)


def make_server(tts, config: Optional[ServingConfig] = None, host: str = "0.0.0.0") -> HTTPServer:
    """One engine, one request at a time (an `mv_engine` is not re-entrant, like the reference's single global TTS)."""
    GlobalState.config = config or ServingConfig()
    GlobalState.tts = tts
    return HTTPServer((host, GlobalState.config.port), _Handler)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--huggingface_repo_id", default=ServingConfig.huggingface_repo_id)
    ap.add_argument("--temperature", type=float, default=1.0)
    ap.add_argument("--seed", type=int, default=1337)
    ap.add_argument("--port", type=int, default=58003)
    ap.add_argument("--quantisation_mode", choices=["int4", "int8"], default=None)
    a = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO)
    from .fast_inference import TTS

    config = ServingConfig(**vars(a))
    tts = TTS(config.huggingface_repo_id, seed=config.seed, quantisation_mode=config.quantisation_mode)
    make_server(tts, config).serve_forever()


if __name__ == "__main__":
    main()
