"""Mirror of the reference's `serving.py` HTTP surface on the MI355X engine.

Same contract (`serving.py:54-109`): `GET /health -> {"status": "ok"}`; `POST /tts` with form fields `text` (required),
`speaker_ref_path` (URL / path) xor `audiodata` (uploaded file), `guidance` in [0, 5] (default 3.0), `top_p` in [0, 1]
(default 0.95) -> `audio/wav` body; 400 when neither speaker input is given; 500 with the reference's fixed text on any
failure; the synthesised file is deleted after the response.  Same flags (`ServingConfig`, `:29-42`):
--huggingface_repo_id --temperature --seed --port (58003) --quantisation_mode.
Addition: `POST /tts_stream`, the same form for texts of any length, answered with chunked transfer encoding -- a RIFF header
with open-ended sizes, then each sentence chunk's PCM as soon as its vocoder pass ends (BASELINE configs[4] "streaming decode").

Dispatch (the part SURVEY.md 8a row a20 / 8e asks to be new): the reference serialises requests on one global TTS (its
`async def` handler calls the blocking `synthesise`, `:89`).  Here requests wait in a queue; a dispatcher thread groups the
pending ones (same guidance / top_p) into a batch of up to `utterances_per_gpu x n_gpus`, shards it round-robin over the ranks
(`batch_dispatch.run_sharded`: one process per GPU, weights replicated, no collective during generation), every rank decodes its
utterances together (`TTS.synthesise_batch`), and ONE all-gather of the finished RIFF bytes (RCCL over xGMI; gloo in the CPU
tests) brings them back to rank 0, which answers the HTTP requests in arrival order.  With one process and one utterance per
GPU this degenerates to the reference's behaviour.  Launch: `python -m torch.distributed.run --nproc-per-node N -m ...serving`.

The reference builds this on FastAPI + python-multipart + tyro + ffmpeg; of those only FastAPI is in this image and its
`Form(...)` needs python-multipart, so the same surface is served here with the standard library (`http.server` + `email`
for multipart bodies).  Uploaded `audiodata` is written to a temp file and handed to `TTS.synthesise` as the speaker
reference (RIFF/WAVE uploads cut to the reference's two minutes); transcoding arbitrary formats through ffmpeg (`:112-124`) is out
of scope.
"""
from __future__ import annotations

import argparse
import json
import logging
import queue
import tempfile
import threading
import time
from dataclasses import dataclass
from email import message_from_bytes
from http.server import BaseHTTPRequestHandler, HTTPServer, ThreadingHTTPServer
from pathlib import Path
from typing import Dict, List, Optional, Tuple
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
    dispatcher = None
    engine_lock = threading.Lock()   # without a dispatcher: serialises the handler threads on the one engine


class BatchingDispatcher:
    """Request queue -> batches -> ranks.  `submit()` blocks the calling (HTTP handler) thread until its result is back.

    Rank 0 owns the queue; ranks > 0 sit in `serve_worker()`.  Every batch starts with a broadcast of the work list (a few
    hundred bytes), then `run_sharded` (generation, no collective) and the single gather of the results."""

    STOP = "__stop__"

    def __init__(self, tts, utterances_per_gpu: int = 1, max_wait_ms: float = 5.0, group=None):
        import torch.distributed as dist

        self.tts = tts
        self.group = group
        self.dist = dist if (dist.is_available() and dist.is_initialized()) else None
        self.world = self.dist.get_world_size(group) if self.dist else 1
        self.rank = self.dist.get_rank(group) if self.dist else 0
        self.max_batch = max(1, utterances_per_gpu) * self.world
        self.max_wait_s = max_wait_ms / 1e3
        self.q: "queue.Queue" = queue.Queue()
        self.batches: List[int] = []  # sizes of the batches dispatched so far (observability / tests)
        self._thread: Optional[threading.Thread] = None
        # an mv_engine is not re-entrant: the dispatcher thread holds this around its shard, `/tts_stream` around every chunk
        self.engine_lock = threading.Lock()

    # ---- rank 0 --------------------------------------------------------------------------------------------------
    def start(self):
        assert self.rank == 0
        self._thread = threading.Thread(target=self._loop, daemon=True)
        self._thread.start()
        return self

    def submit(self, text: str, spk_ref: str, top_p: float, guidance: float) -> bytes:
        # The speaker reference is checked HERE, on the calling (HTTP handler) thread: the check may fetch a URI (60 s per read) and decodes
        # the whole file -- on the dispatcher thread it would hold up every queued batch on every rank.  A request that fails it never
        # reaches the queue.
        item = dict(text=text, spk=spk_ref, top_p=top_p, guidance=guidance, checked=True)
        err = self._validate(item)
        if err is not None:
            raise err
        done = threading.Event()
        slot: Dict[str, object] = {}
        self.q.put((item, done, slot))
        done.wait()
        if "error" in slot:
            raise slot["error"]  # type: ignore[misc]
        return slot["wav"]  # type: ignore[return-value]

    def stop(self):
        self.q.put(None)
        if self._thread is not None:
            self._thread.join(timeout=30)

    def _take_batch(self):
        first = self.q.get()
        if first is None:
            return None
        batch = [first]
        deadline = time.monotonic() + self.max_wait_s
        held = []
        while len(batch) < self.max_batch:
            try:
                nxt = self.q.get(timeout=max(0.0, deadline - time.monotonic()))
            except queue.Empty:
                break
            if nxt is None:
                held.append(nxt)
                break
            same = (nxt[0]["top_p"], nxt[0]["guidance"]) == (first[0]["top_p"], first[0]["guidance"])
            (batch if same else held).append(nxt)  # other sampling controls wait for the next batch
        for h in held:
            self.q.put(h)
        return batch

    def _validate(self, item: dict) -> Optional[Exception]:
        """Rank 0, before the work list is announced: a speaker reference that cannot be used fails its own request here and
        never reaches the other ranks (`TTS.check_speaker_reference`: the reference's checks, `fam/llm/utils.py:55-70`)."""
        check = getattr(self.tts, "check_speaker_reference", None)
        if check is None:
            return None
        try:
            check(item["spk"])
        except Exception as e:
            return e
        return None

    def _loop(self):
        while True:
            batch = self._take_batch()
            if batch is None:
                self._broadcast([self.STOP])
                return
            live = []
            for entry in batch:
                err = None if entry[0].get("checked") else self._validate(entry[0])   # entries that did not come through submit()
                if err is None:
                    live.append(entry)
                else:
                    entry[2]["error"] = err
                    entry[1].set()
            if not live:
                continue
            items = [b[0] for b in live]
            self.batches.append(len(items))
            try:
                # Every rank enters BOTH collectives of a batch whatever happens to its requests: a failing request travels
                # as an error payload through the same gather (a rank that skipped the gather would leave the others blocked in
                # it, and its next broadcast would meet their gather: the process group hangs).
                self._broadcast(items)
                results = self._run(items)
                for (_, done, slot), r in zip(live, results):
                    slot["error" if isinstance(r, Exception) else "wav"] = r
                    done.set()
            except Exception as e:  # the collectives themselves failed: every waiter of the batch gets the failure
                logger.exception("batch failed")
                for _, done, slot in live:
                    slot["error"] = e
                    done.set()

    # ---- all ranks -----------------------------------------------------------------------------------------------
    def _broadcast(self, items: list) -> list:
        if self.dist is None or self.world == 1:
            return items
        box = [items]
        self.dist.broadcast_object_list(box, src=0, group=self.group)
        return box[0]

    OK, FAILED = b"\x00", b"\x01"  # first byte of every gathered payload: RIFF bytes follow / an error message follows

    def _run(self, items: list) -> list:
        """This rank's shard of `items` -> one payload per item, then THE gather.  Returns, for every item in input order, the RIFF
        bytes or the exception of that item.  Never raises before the gather."""
        import torch

        from .batch_dispatch import gather_variable, shard_indices

        mine = shard_indices(len(items), self.rank, self.world)
        payloads: List[bytes] = []
        if mine:
            try:
                with self.engine_lock:
                    blobs = self.tts.synthesise_batch([items[i]["text"] for i in mine], [items[i]["spk"] for i in mine],
                                                      top_p=items[0]["top_p"], guidance_scale=items[0]["guidance"], return_exceptions=True)
            except Exception as e:  # anything per-item handling did not catch: this rank's items fail, the batch goes on
                logger.exception("rank %d failed on its shard", self.rank)
                blobs = [e] * len(mine)
            for b in blobs:
                if isinstance(b, Exception):
                    logger.error("request failed on rank %d: %r", self.rank, b)
                    payloads.append(self.FAILED + f"{type(b).__name__}: {b}".encode("utf-8", "replace"))
                else:
                    payloads.append(self.OK + b)
        local = [torch.frombuffer(bytearray(b), dtype=torch.uint8) for b in payloads]
        device = None
        if self.dist is not None and self.world > 1 and self.dist.get_backend(self.group) == "nccl":
            device = torch.device("cuda", torch.cuda.current_device())
            local = [t.to(device) for t in local]
        out = gather_variable(local, len(items), group=self.group, device=device, dtype=torch.uint8)
        results: list = []
        for t in out:
            raw = bytes(t.cpu().numpy().tobytes())
            if raw[:1] == self.OK:
                results.append(raw[1:])
            else:   # "<class name>: <text>" (see above): the class name decides the status, the text goes to the log
                kind, _, msg = raw[1:].decode("utf-8", "replace").partition(": ")
                results.append(RemoteRequestError(kind, msg))
        return results

    def serve_worker(self):
        """Ranks > 0: take part in every batch rank 0 announces until it announces the end."""
        assert self.rank != 0
        while True:
            items = self._broadcast([])
            if items and items[0] == self.STOP:
                return
            self._run(items)  # per-request failures travel through the gather; only a broken process group raises here


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


def _validate_form(fields: Dict[str, str], files: Dict[str, bytes]):
    """The form checks of `/tts` (FastAPI's `Form(..., ge=, le=)` validation and `serving.py:69-73` in the reference) ->
    either (status, content type, body) of the refusal, or None."""
    if fields.get("text") is None:
        return 422, "application/json", json.dumps({"detail": "field 'text' is required"}).encode()
    try:
        guidance = float(fields.get("guidance", 3.0))
        top_p = float(fields.get("top_p", 0.95))
    except ValueError:
        return 422, "application/json", json.dumps({"detail": "guidance / top_p must be numbers"}).encode()
    if not (0.0 <= guidance <= 5.0) or not (0.0 <= top_p <= 1.0):
        return 422, "application/json", json.dumps({"detail": "guidance must be in [0, 5] and top_p in [0, 1]"}).encode()
    if not files.get("audiodata") and not fields.get("speaker_ref_path"):
        return 400, "application/json", json.dumps(
            {"detail": "Either an audio file or a speaker reference path must be provided."}).encode()
    return None


STREAM_WAV_HEADER = (b"RIFF" + b"\xff\xff\xff\xff" + b"WAVE" + b"fmt " + (16).to_bytes(4, "little") + (1).to_bytes(2, "little") +
                     (1).to_bytes(2, "little") + (24000).to_bytes(4, "little") + (48000).to_bytes(4, "little") + (2).to_bytes(2, "little") +
                     (16).to_bytes(2, "little") + b"data" + b"\xff\xff\xff\xff")  # 24 kHz mono PCM16, sizes unknown (streamed)


UPLOAD_MAX_SECONDS = 120.0   # `_convert_audiodata_to_wav_path` (serving.py:112-124): `ffmpeg -t 120`, "arbitrary 2 minute cutoff"
UPLOAD_PREFIX = "mv_upload_"   # the temp files uploads are stored under: inference.get_cached_embedding keeps no cache entry for them



def _store_upload(audiodata: bytes, wav_tmp) -> str:
    """The uploaded speaker reference as a temp file for `TTS.synthesise`.  The reference transcodes every upload through ffmpeg and keeps
    its first two minutes; there is no ffmpeg here, uploads are handed on in the format they came in (speaker_encoder/audio.py sniffs
    it) -- but a RIFF/WAVE upload longer than two minutes is cut like the reference's."""
    wav_tmp.write(audiodata)
    wav_tmp.flush()
    if audiodata[:4] in (b"RIFF", b"RIFX") and audiodata[8:12] == b"WAVE":
        try:
            from scipy.io import wavfile

            rate, data = wavfile.read(wav_tmp.name)
            keep = int(UPLOAD_MAX_SECONDS * rate)
            if len(data) > keep:
                wavfile.write(wav_tmp.name, rate, data[:keep])
        except Exception:   # an odd header: leave the file to load_audio, which reports what it cannot read
            pass
    return wav_tmp.name


def text_to_speech_stream(fields: Dict[str, str], files: Dict[str, bytes]):
    """`POST /tts_stream` (no counterpart in the reference, which stops at 220 characters: BASELINE configs[4] "streaming decode"):
    the same form as `/tts`, any text length.  Returns (refusal, None) or (None, generator of body pieces): the RIFF header with
    open-ended sizes, then the 16-bit PCM of every sentence chunk as soon as its vocoder pass ends (`TTS.synthesise_long_stream`).
    The first piece is produced before the response is committed, so a request that fails early still gets the 500 of `/tts`."""
    refusal = _validate_form(fields, files)
    if refusal is not None:
        return refusal, None
    text, guidance, top_p = fields["text"], float(fields.get("guidance", 3.0)), float(fields.get("top_p", 0.95))
    lock = _engine_lock()

    def pieces():
        with tempfile.NamedTemporaryFile(prefix=UPLOAD_PREFIX, suffix=".wav") as wav_tmp:
            if fields.get("speaker_ref_path") is None:
                spk = _store_upload(files["audiodata"], wav_tmp)
            else:
                spk = fields["speaker_ref_path"]
            gen = GlobalState.tts.synthesise_long_stream(text, spk, top_p=top_p, guidance_scale=guidance, temperature=GlobalState.config.temperature)
            yield STREAM_WAV_HEADER
            while True:
                with lock:  # one chunk at a time on the engine; batches of the dispatcher interleave between chunks
                    try:
                        _, _, wav = next(gen)
                    except StopIteration:
                        return
                x = wav.detach().reshape(-1).float().cpu().clamp(-1.0, 1.0)
                yield (x * 32767.0).round().short().numpy().tobytes()

    it = pieces()
    try:
        head = next(it)
        first = next(it, b"")
    except Exception as e:
        logger.exception(f"Error processing streaming request. text: {text[:80]!r}")
        return _error_response(e), None

    def body():
        yield head
        if first:
            yield first
        yield from it

    return None, body()


def text_to_speech(fields: Dict[str, str], files: Dict[str, bytes]) -> Tuple[int, str, bytes]:
    """The `/tts` handler body (`serving.py:59-109`) -> (status, content type, body)."""
    refusal = _validate_form(fields, files)
    if refusal is not None:
        return refusal
    text, guidance, top_p = fields["text"], float(fields.get("guidance", 3.0)), float(fields.get("top_p", 0.95))
    speaker_ref_path = fields.get("speaker_ref_path")
    audiodata = files.get("audiodata")
    wav_out_path = None
    try:
        with tempfile.NamedTemporaryFile(prefix=UPLOAD_PREFIX, suffix=".wav") as wav_tmp:
            if speaker_ref_path is None:
                wav_path = _store_upload(audiodata, wav_tmp)
            else:
                wav_path = speaker_ref_path
            if GlobalState.dispatcher is not None:  # queued, batched, sharded over the GPUs; the waveform never touches the disk
                return 200, "audio/wav", GlobalState.dispatcher.submit(text, wav_path, top_p, guidance)
            with _engine_lock():  # handler threads are concurrent (keep-alive clients must not block each other); the engine is not re-entrant
                wav_out_path = GlobalState.tts.synthesise(text=text, spk_ref_path=wav_path, top_p=top_p, guidance_scale=guidance)
        with open(wav_out_path, "rb") as f:
            return 200, "audio/wav", f.read()
    except Exception as e:
        logger.exception(f"Error processing request. text: {text}, speaker_ref_path: {speaker_ref_path}, guidance: {guidance}, top_p: {top_p}")
        return _error_response(e)
    finally:
        if wav_out_path is not None:
            Path(wav_out_path).unlink(missing_ok=True)


def _engine_lock():
    """The one lock around the engine: the dispatcher's when there is one, else the server's own."""
    return GlobalState.dispatcher.engine_lock if GlobalState.dispatcher is not None else GlobalState.engine_lock


class RemoteRequestError(RuntimeError):
    """A request that failed on another rank: the gathered payload carries the exception's class name and text (`remote_type`, message)."""

    def __init__(self, remote_type: str, message: str):
        super().__init__(f"{remote_type}: {message}")
        self.remote_type, self.message = remote_type, message


UNSUPPORTED_AUDIO_DETAIL = ("the speaker reference is in an audio format this server cannot decode; send RIFF/WAVE (or install a decoder: "
                            "soundfile / ffmpeg)")


def _is_unsupported_audio(e: BaseException) -> bool:
    """By TYPE, along the cause / context chain (and the class name a failure on another rank travelled under) -- never by searching the
    message text: an unrelated internal error that merely quotes the name stays a 500."""
    seen = 0
    while e is not None and seen < 8:
        if any(c.__name__ == "UnsupportedAudioFormat" for c in type(e).__mro__) or getattr(e, "remote_type", None) == "UnsupportedAudioFormat":
            return True
        e, seen = e.__cause__ or e.__context__, seen + 1
    return False


def _error_response(e: Exception) -> Tuple[int, str, bytes]:
    """Exception -> (status, content type, body), shared by `/tts` and `/tts_stream`: a speaker reference in a format nothing here can
    decode is the client's to fix (415, with the decoder's own reason when the exception IS that type: it names the file the client
    sent, nothing internal), everything else is the reference's fixed 500 text (`serving.py:101-105`)."""
    if _is_unsupported_audio(e):
        own = any(c.__name__ == "UnsupportedAudioFormat" for c in type(e).__mro__)
        detail = str(e) if own else (getattr(e, "message", None) if getattr(e, "remote_type", None) == "UnsupportedAudioFormat" else None)
        return 415, "application/json", json.dumps({"detail": detail or UNSUPPORTED_AUDIO_DETAIL}).encode()
    return 500, "text/plain", ERROR_TEXT.encode()


MAX_REQUEST_BYTES = 256 << 20  # text + an uploaded speaker reference (the reference's ffmpeg step keeps the first 2 minutes of it)


class _Handler(BaseHTTPRequestHandler):
    protocol_version = "HTTP/1.1"  # chunked transfer for /tts_stream; every other response carries its Content-Length
    timeout = 30                   # an idle keep-alive socket gives its handler thread back after 30 s (StreamRequestHandler.timeout)

    def _send_chunked(self, ctype: str, pieces):
        self.send_response(200)
        self.send_header("Content-Type", ctype)
        self.send_header("Transfer-Encoding", "chunked")
        self.send_header("Access-Control-Allow-Origin", "*")
        self.end_headers()
        try:
            for p in pieces:
                if p:
                    self.wfile.write(f"{len(p):x}\r\n".encode() + p + b"\r\n")
                    self.wfile.flush()
            self.wfile.write(b"0\r\n\r\n")
        except Exception:  # a chunk failed after the response was committed: cut the stream, the client sees a truncated body
            logger.exception("streaming response aborted")
            self.close_connection = True

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

    def do_OPTIONS(self):
        """CORS preflight, answered the way the reference's CORSMiddleware is configured (`serving.py:138-144`: any origin, method
        and header)."""
        self.send_response(200)
        self.send_header("Access-Control-Allow-Origin", self.headers.get("Origin", "*"))
        self.send_header("Access-Control-Allow-Credentials", "true")
        self.send_header("Access-Control-Allow-Methods", self.headers.get("Access-Control-Request-Method", "GET, POST, OPTIONS"))
        self.send_header("Access-Control-Allow-Headers", self.headers.get("Access-Control-Request-Headers", "*"))
        self.send_header("Access-Control-Max-Age", "600")
        self.send_header("Vary", "Origin")
        self.send_header("Content-Length", "0")
        self.end_headers()

    def do_POST(self):
        route = self.path.split("?")[0]
        try:
            length = int(self.headers.get("Content-Length", 0))
        except ValueError:
            length = -1
        if length < 0 or length > MAX_REQUEST_BYTES:  # the body is read into memory: refuse before reading, then drop the connection
            self.close_connection = True
            self._send(413 if length > 0 else 400, "application/json", b'{"detail": "Request body too large or of unknown length"}')
            return
        body = self.rfile.read(length)
        if route not in ("/tts", "/tts_stream"):
            self._send(404, "application/json", b'{"detail": "Not Found"}')
            return
        fields, files = _parse_form(self.headers.get("Content-Type", "application/x-www-form-urlencoded"), body)
        if route == "/tts_stream":
            refusal, pieces = text_to_speech_stream(fields, files)
            if refusal is not None:
                self._send(*refusal)
            else:
                self._send_chunked("audio/wav", pieces)
            return
        self._send(*text_to_speech(fields, files))

    def log_message(self, fmt, *args):
        logger.info("%s - %s", self.address_string(), fmt % args)


def make_server(tts, config: Optional[ServingConfig] = None, host: str = "0.0.0.0", dispatcher: Optional[BatchingDispatcher] = None) -> HTTPServer:
    """Without a dispatcher: one engine, one request at a time (the reference's single global TTS).  With one: handler threads
    enqueue and wait, the dispatcher thread is the only one that touches the engine (an `mv_engine` is not re-entrant)."""
    GlobalState.config = config or ServingConfig()
    GlobalState.tts = tts
    GlobalState.dispatcher = dispatcher
    GlobalState.engine_lock = threading.Lock()
    # one thread per connection in BOTH modes: HTTP/1.1 keeps connections open, and with a single-threaded server one idle persistent
    # client (a browser tab, a load-balancer probe) would block every other client, /health included
    return ThreadingHTTPServer((host, GlobalState.config.port), _Handler)


def main(argv=None):
    import os

    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--huggingface_repo_id", default=ServingConfig.huggingface_repo_id)
    ap.add_argument("--temperature", type=float, default=1.0)
    ap.add_argument("--seed", type=int, default=1337)
    ap.add_argument("--port", type=int, default=58003)
    ap.add_argument("--quantisation_mode", choices=["int4", "int8"], default=None)
    ap.add_argument("--utterances_per_gpu", type=int, default=1, help="requests decoded together on one GPU (not in the reference)")
    a = ap.parse_args(argv)
    if not 1 <= a.utterances_per_gpu <= 8 or (a.utterances_per_gpu > 4 and a.quantisation_mode is not None):
        ap.error("--utterances_per_gpu: 1..8 (3 and 5..7 run on engines of 4 / 8 slots; more than 4 needs bf16 weights: the 8-slot engine runs on the matrix-core kernel)")
    logging.basicConfig(level=logging.INFO)
    import torch

    from .fast_inference import TTS

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:  # one process per GPU (torch.distributed.run); RCCL carries the single gather of finished audio
        import torch.distributed as dist

        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")
    config = ServingConfig(huggingface_repo_id=a.huggingface_repo_id, temperature=a.temperature, seed=a.seed, port=a.port,
                           quantisation_mode=a.quantisation_mode)
    tts = TTS(config.huggingface_repo_id, seed=config.seed, quantisation_mode=config.quantisation_mode,
              max_utterances=a.utterances_per_gpu)
    if world > 1 or a.utterances_per_gpu > 1:
        disp = BatchingDispatcher(tts, utterances_per_gpu=a.utterances_per_gpu)
        if disp.rank != 0:
            disp.serve_worker()
            return
        make_server(tts, config, dispatcher=disp.start()).serve_forever()
    else:
        make_server(tts, config).serve_forever()


if __name__ == "__main__":
    main()
