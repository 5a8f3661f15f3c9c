"""TF-Serving-compatible REST front end for the HIP forward (SURVEY.md section 8(b) "Wire", 8(f) rank 1).

The reference's Jetty server ranks candidates by POSTing to TensorFlow Serving
(``RecForYouProcess.callNeuralCFTFServing``, src/main/java/com/sparrowrecsys/online/recprocess/
RecForYouProcess.java:113-138; HTTP client HttpClient.java:21-40):

    POST http://localhost:8501/v1/models/recmodel:predict
    {"instances": [{"userId": 123, "movieId": 456}, ... 800 of them ...]}
    -> {"predictions": [[0.73], [0.12], ...]}            # parsed with getJSONArray(i).getDouble(0)

This module answers the same route with the same JSON in front of any ``CTRModel``, so the Jetty
server can call it unchanged.  Also accepted, as TF Serving does: the columnar form
``{"inputs": {"userId": [...], "movieId": [...]}}`` (answered with ``{"outputs": [[p], ...]}``) and
``GET /v1/models/<name>`` (model status).  Invalid ids (TF: InvalidArgumentError from
assert_less_than_num_buckets) -> HTTP 400 ``{"error": "..."}``, which the Jetty side turns into an empty
recommendation list exactly as it does for a failed TF Serving call (HttpClient.java:36-39,
RecForYouService.java:49-52).

Concurrent requests (Jetty servlet threads, one blocking call each) are merged by a micro-batcher into
one forward: a request of 800 candidates is far below what keeps an MI355X busy.

    python -m sparrowrecsys_amd.serving --model neuralcf --weights weights.npz --port 8501
"""
from __future__ import annotations

import argparse
import json
import os
import queue
import threading
import time
from http.server import BaseHTTPRequestHandler, ThreadingHTTPServer
from typing import Dict, List, Mapping, Optional, Sequence

import numpy as np

_STRING_PREFIXES = ("userGenre", "movieGenre")          # string feature columns of the reference schema


def _columns_from_instances(instances: Sequence[Mapping], fast: bool = False) -> Dict[str, list]:
    if not isinstance(instances, list):
        raise ValueError('"instances" must be a list of objects')
    if fast and instances and isinstance(instances[0], dict):
        # the common request (the Jetty ranker's): every instance carries the same keys with scalar values -- one list
        # comprehension per key instead of a Python-level double loop (0.30 -> 0.04 ms for 800 instances); anything else
        # (missing keys, the [x] spelling of a scalar -- numeric or string) raises here or in _to_feature_arrays and takes
        # the general path
        k0 = list(instances[0])
        try:
            return {k: [i[k] for i in instances] for k in k0}
        except (KeyError, TypeError):
            raise ValueError("instances are not uniform")
    if not all(isinstance(i, dict) for i in instances):
        raise ValueError('"instances" must be a list of objects')
    keys: List[str] = []
    for inst in instances:
        for k in inst:
            if k not in keys:
                keys.append(k)
    cols: Dict[str, list] = {k: [] for k in keys}
    for inst in instances:
        for k in keys:
            v = inst.get(k)
            if isinstance(v, list):                       # TF Serving also accepts [x] for a scalar feature
                v = v[0] if len(v) == 1 else v
            cols[k].append(v)
    return cols


_DIGITS = b"0123456789-"
_TO_SPACE = bytes(c if (48 <= c <= 57 or c == 45) else 32 for c in range(256))


def _fast_uniform_int_instances(body: bytes) -> Optional[Dict[str, np.ndarray]]:
    """The Jetty ranker's request, byte for byte: {"instances": [{"userId": 7, "movieId": 1}, {"userId": 7, "movieId": 2}, ...]} --
    800 flat objects, the same integer-valued keys in the same order (RecForYouProcess.java:113-127 builds them in a loop; org.json
    writes them without spaces, Python clients with ", " / ": ").  json.loads + the column loop cost 0.37 ms of such a request's
    ~0.6; here the numbers are read in one numpy call and the STRUCTURE is proven by comparing the body with its digits deleted
    against the skeleton the first object predicts -- any difference (another key, another order, a float, a string, a nested list,
    pretty-printing) returns None and the request takes the general path, errors included.  Returns {key: int64 array} or None."""
    try:
        skel = body.translate(None, _DIGITS)
        i = skel.find(b"[")
        if i < 0 or skel[:i + 1].replace(b" ", b"") != b'{"instances":[':
            return None
        j = skel.find(b"}", i)
        unit = skel[i + 1:j + 1].lstrip()
        if not unit.startswith(b"{") or b"[" in unit or len(unit) < 6:
            return None
        after = skel[j + 1:j + 3]
        sep = b", " if after == b", " else (b"," if after[:1] == b"," else b"")
        tail = skel.rstrip()[-2:]
        if tail != b"]}":
            return None
        lead = len(skel[:i + 1]) + (len(skel[i + 1:j + 1]) - len(unit))
        span = len(skel.rstrip()) - lead - 2
        step = len(unit) + len(sep)
        if sep == b"":
            n = 1 if span == len(unit) else 0
        else:
            n = (span + len(sep)) // step if (span + len(sep)) % step == 0 else 0
        if n < 1 or skel.rstrip() != skel[:lead] + sep.join([unit] * n) + b"]}":
            return None
        # the first object WITH its numbers: a flat object of integers, keys as they will be named
        b0 = body.find(b"{", body.find(b"["))
        first = json.loads(body[b0:body.find(b"}", b0) + 1].decode("utf-8"))
        keys = list(first)
        if not keys or any(type(v) is not int for v in first.values()) or any(any(c.isdigit() or c == "-" for c in k) for k in keys):
            return None
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("error")                       # (a number numpy cannot read to its end: not ours)
            vals = np.fromstring(body.translate(_TO_SPACE).decode("ascii"), dtype=np.int64, sep=" ")
        if vals.size != n * len(keys) or int(vals.max()) >= 1 << 62 or int(vals.min()) <= -(1 << 62):   # (numpy saturates what overflows int64)
            return None
        vals = vals.reshape(n, len(keys))
        return {k: np.ascontiguousarray(vals[:, c]) for c, k in enumerate(keys)}
    except Exception:
        return None


def _to_feature_arrays(cols: Mapping[str, list], n: int) -> Dict[str, np.ndarray]:
    feats = {}
    for k, v in cols.items():
        if len(v) != n:
            raise ValueError("feature %r has %d values for %d instances" % (k, len(v), n))
        if k.startswith(_STRING_PREFIXES):
            # a string column takes scalars only: str() of a list never fails, so the TF-Serving "[x]" spelling would
            # otherwise become the text "['Action']" -- out of vocabulary, a wrong score with HTTP 200.
            # Raising here sends the fast path to the general one, which unwraps [x]; what is still a list there is a 400.
            if any(isinstance(x, (list, dict, tuple)) for x in v):
                raise ValueError("feature %r must be a list of scalar strings" % k)
            feats[k] = np.array(["" if x is None else str(x) for x in v], dtype=object)
        else:
            try:
                feats[k] = np.array([0 if x is None else x for x in v])
            except Exception as e:                        # ragged / non-numeric
                raise ValueError("feature %r: %s" % (k, e))
            if feats[k].dtype == object or feats[k].ndim != 1:
                raise ValueError("feature %r must be a list of scalars" % k)
    return feats


class _MicroBatcher:
    """Merges concurrent predict calls into one forward (rows concatenated, results split back)."""

    def __init__(self, predict_fn, max_rows: int = 1 << 16, max_wait_s: float = 0.0005):
        self.predict_fn, self.max_rows, self.max_wait_s = predict_fn, max_rows, max_wait_s
        self.q: "queue.Queue" = queue.Queue()
        self.batches = 0
        self.requests = 0
        # requests a handler thread has started to parse but not submitted yet: the batcher only waits (up to max_wait_s) when
        # there is somebody to wait for.  Round 2 waited unconditionally -- half of a lone client's 1.03 ms per 800-candidate
        # request (RecForYouProcess.java:113-138 sends them one at a time per servlet thread) was this timer.
        self.arriving = 0
        self._lock = threading.Lock()
        # one forward at a time (the engine's staging buffers are not re-entrant).  A request that finds nobody else arriving or queued
        # and the engine free runs its forward INLINE on the handler's thread: the two thread hand-offs through the queue cost a lone
        # client ~0.1 ms per request under the GIL
        self._run_lock = threading.Lock()
        self.inline = 0
        self._allow_inline = os.environ.get("SPRK_SERVING_INLINE", "1") != "0"          # (A/B switches of scripts/r04/09_serving.sh)
        self._adaptive = os.environ.get("SPRK_SERVING_ADAPTIVE_WAIT", "1") != "0"
        self._stop = False
        self.thread = threading.Thread(target=self._run, daemon=True, name="sparrow-batcher")
        self.thread.start()

    def announce(self):
        with self._lock:
            self.arriving += 1

    def withdraw(self):
        with self._lock:
            self.arriving -= 1

    def submit(self, feats: Dict[str, np.ndarray], n: int) -> np.ndarray:
        if self._allow_inline and self.arriving == 0 and self.q.empty() and self._run_lock.acquire(False):
            try:
                self.inline += 1
                self.requests += 1
                self.batches += 1                                # (an inline forward is a batch of one request: requests / batches stays the mean group size)
                return np.asarray(self.predict_fn(feats), dtype=np.float32).reshape(-1)
            finally:
                self._run_lock.release()
        done = threading.Event()
        slot = {"feats": feats, "n": n, "done": done, "out": None, "err": None}
        self.q.put(slot)
        done.wait()
        if slot["err"] is not None:
            raise slot["err"]
        return slot["out"]

    def close(self):
        self._stop = True
        self.q.put(None)
        self.thread.join(timeout=5)

    def _run(self):
        while not self._stop:
            first = self.q.get()
            if first is None:
                return
            group, rows = [first], first["n"]
            deadline = time.monotonic() + self.max_wait_s
            while rows < self.max_rows:
                try:
                    # whatever is queued already joins the group; the timer only runs while another request is on its way
                    wait = max(0.0, deadline - time.monotonic()) if (self.arriving > 0 or not self._adaptive) else 0.0
                    nxt = self.q.get(timeout=wait) if wait > 0.0 else self.q.get_nowait()
                except queue.Empty:
                    break
                if nxt is None:
                    self._stop = True
                    break
                if set(nxt["feats"]) != set(first["feats"]):     # different feature sets: next round
                    self.q.put(nxt)
                    break
                group.append(nxt)
                rows += nxt["n"]
            self._serve(group)

    def _serve(self, group):
        with self._run_lock:
            self._serve_locked(group)

    def _serve_locked(self, group):
        self.batches += 1
        self.requests += len(group)
        try:
            if len(group) == 1:
                merged = group[0]["feats"]
            else:
                merged = {k: np.concatenate([g["feats"][k] for g in group]) for k in group[0]["feats"]}
            out = np.asarray(self.predict_fn(merged), dtype=np.float32).reshape(-1)
            pos = 0
            for g in group:
                g["out"] = out[pos:pos + g["n"]]
                pos += g["n"]
        except Exception as e:
            if len(group) == 1:
                group[0]["err"] = e
            else:
                # one bad request must not fail its neighbours: serve each on its own
                for g in group:
                    try:
                        g["out"] = np.asarray(self.predict_fn(g["feats"]), dtype=np.float32).reshape(-1)
                    except Exception as e1:
                        g["err"] = e1
        for g in group:
            g["done"].set()


class _Headers(dict):
    """Lower-cased header names -> values: the part of email.message.Message the handler uses."""

    def get(self, name, default=None):
        return dict.get(self, name.lower(), default)


class PredictServer:
    """``POST /v1/models/<name>:predict`` in front of ``model.predict`` (any object with a Keras-shaped
    ``predict(dict) -> [N,1]``).  ``defaults`` fills feature columns a request does not send (the Jetty
    ranker only sends userId / movieId)."""

    def __init__(self, model, name: str = "recmodel", host: str = "127.0.0.1", port: int = 8501,
                 defaults: Optional[Mapping[str, object]] = None, max_wait_s: float = 0.0005,
                 max_body_bytes: int = 64 << 20, reuse_port: bool = False):
        self.model, self.name, self.defaults = model, name, dict(defaults or {})
        self.max_body_bytes = int(max_body_bytes)                # (800 candidates are 25 KB; 64 MiB is a few million instances)
        self.fast_parse = os.environ.get("SPRK_SERVING_FAST_PARSE", "1") != "0"
        self.batcher = _MicroBatcher(self._predict, max_wait_s=max_wait_s)
        outer = self

        class Handler(BaseHTTPRequestHandler):
            protocol_version = "HTTP/1.1"
            # headers and body leave in ONE write: the stock handler's unbuffered wfile sends them as two segments, and on a
            # keep-alive connection the second one then waits for the client's delayed ACK of the first (Nagle) -- a load test
            # (scripts/bench_serving.py) showed a flat 50 ms per request for a forward that takes microseconds
            wbufsize = 1 << 16

            def setup(self):
                super().setup()
                try:
                    import socket
                    self.connection.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                except OSError:
                    pass

            def log_message(self, fmt, *args):               # quiet
                pass

            def parse_request(self):
                """The stock parser hands the header block to email.parser (0.12 ms of a 0.56 ms request, profiles/r04): the three
                things this shim needs of a request -- method, path, Content-Length (and whether the connection stays open) -- are
                read here directly.  Anything unusual (a request line that is not `METHOD path HTTP/1.x`, over-long lines, chunked
                bodies, Expect) goes to the stock parser, whose state this method otherwise reproduces."""
                line = self.raw_requestline
                parts = line.split()
                if len(parts) != 3 or parts[2] not in (b"HTTP/1.1", b"HTTP/1.0") or len(line) > 4096:
                    return self._stock_parse()
                hdr, pending = {}, []
                for _ in range(64):
                    h = self.rfile.readline(8193)
                    pending.append(h)
                    if h in (b"\r\n", b"\n", b""):
                        break
                    k, sep, v = h.partition(b":")
                    if not sep or len(h) > 8192 or h[:1] in b" \t":
                        return self._stock_parse(pending)
                    hdr[k.strip().lower().decode("latin-1")] = v.strip().decode("latin-1")
                else:
                    return self._stock_parse(pending)
                if "transfer-encoding" in hdr or "expect" in hdr:
                    return self._stock_parse(pending)
                self.command, self.path = parts[0].decode("latin-1"), parts[1].decode("latin-1")
                self.request_version = parts[2].decode("latin-1")
                self.requestline = line.rstrip(b"\r\n").decode("latin-1")
                self.headers = _Headers(hdr)
                conn = hdr.get("connection", "").lower()
                self.close_connection = conn == "close" or (self.request_version == "HTTP/1.0" and conn != "keep-alive")
                return True

            def _stock_parse(self, pending=()):
                if pending:                                      # lines already taken off the stream go back in front of it
                    import io
                    rest = self.rfile
                    class _Chain(io.RawIOBase):
                        def __init__(s2): s2.buf = io.BytesIO(b"".join(pending))
                        def readline(s2, n=-1):
                            b = s2.buf.readline(n)
                            return b if b else rest.readline(n)
                        def read(s2, n=-1):
                            b = s2.buf.read(n)
                            return b if b or n == 0 else rest.read(n)
                    self.rfile = _Chain()
                return BaseHTTPRequestHandler.parse_request(self)

            def _send(self, code: int, obj):
                body = json.dumps(obj).encode("utf-8")
                self.send_response(code)
                self.send_header("Content-Type", "application/json")
                self.send_header("Content-Length", str(len(body)))
                self.end_headers()
                self.wfile.write(body)
                self.wfile.flush()

            def _send_scores(self, key: str, scores):
                # [[p], [p], ...] written directly: 9 significant digits reproduce a float32 exactly, and formatting 800 of
                # them this way costs 0.2 ms against 0.84 ms for json.dumps of the nested list (shortest-repr of doubles)
                arr = np.asarray(scores, dtype=np.float32)
                if not np.isfinite(arr).all():
                    # "%.9g" would print nan / inf, which is not JSON; json.dumps spells them NaN / Infinity, what Python
                    # and Java clients' lenient parsers accept (and what this shim sent before the fast formatter)
                    self._send(200, {key: [[float(v)] for v in arr.tolist()]})
                    return
                vals = arr.tolist()                              # (one format call for all of them: 0.15 ms instead of 0.21 for 800)
                body = ('{"%s": [[' % key + ("%.9g], [" * (len(vals) - 1) + "%.9g") % tuple(vals) + "]]}").encode("utf-8")
                self.send_response(200)
                self.send_header("Content-Type", "application/json")
                self.send_header("Content-Length", str(len(body)))
                self.end_headers()
                self.wfile.write(body)
                self.wfile.flush()

            def do_GET(self):
                if self.path.rstrip("/") == "/v1/models/" + outer.name:
                    self._send(200, {"model_version_status": [{"version": "1", "state": "AVAILABLE",
                                                                "status": {"error_code": "OK", "error_message": ""}}],
                                     "worker_pid": os.getpid()})   # (not TF Serving's: which worker of serve_workers answered)
                else:
                    self._send(404, {"error": "Not found: %s" % self.path})

            def do_POST(self):
                if self.path != "/v1/models/%s:predict" % outer.name:
                    self._send(404, {"error": "Not found: %s" % self.path})
                    return
                announced = True
                outer.batcher.announce()
                try:
                    # [r5] the length is validated before a byte of the body is read: int("-1") used to reach rfile.read(-1) -- "until
                    # EOF", i.e. a handler thread parked on the connection for as long as the client keeps it open -- and nothing
                    # bounded what one request could make the process buffer
                    cl = self.headers.get("Content-Length")
                    if cl is None or not cl.strip().isdigit():
                        self.close_connection = True             # (the body cannot be skipped: the stream is out of step from here on)
                        self._send(411 if cl is None else 400, {"error": "a Content-Length header with a non-negative decimal value is required"})
                        return
                    n = int(cl)
                    if n > outer.max_body_bytes:
                        self.close_connection = True
                        self._send(413, {"error": "request body of %d bytes exceeds the limit of %d" % (n, outer.max_body_bytes)})
                        return
                    raw = self.rfile.read(n)
                    if len(raw) != n:
                        self.close_connection = True
                        raise ValueError("request body ended after %d of %d bytes" % (len(raw), n))
                    feats = _fast_uniform_int_instances(raw) if outer.fast_parse else None
                    cols = key = None
                    if feats is not None:                        # the Jetty ranker's request shape: no json.loads at all
                        key, cols = "predictions", feats
                        if any(k.startswith(_STRING_PREFIXES) for k in feats):
                            feats = None
                    req = json.loads(raw.decode("utf-8")) if feats is None else {}
                    if not isinstance(req, dict):
                        raise ValueError("request body must be a JSON object")
                    if feats is not None:
                        pass
                    elif "instances" in req:
                        key = "predictions"
                        try:                                     # uniform scalar instances: the fast conversion
                            cols = _columns_from_instances(req["instances"], fast=True)
                            rows = len(req["instances"])
                            feats = _to_feature_arrays(cols, rows) if rows and sum(map(len, req["instances"])) == rows * len(cols) else None
                        except ValueError:
                            feats = None
                        if feats is None:
                            cols = _columns_from_instances(req["instances"])
                    elif "inputs" in req:
                        if not isinstance(req["inputs"], dict):
                            raise ValueError('"inputs" must be an object of feature columns')
                        cols, key = {k: (v if isinstance(v, list) else [v]) for k, v in req["inputs"].items()}, "outputs"
                    else:
                        raise ValueError('request needs "instances" or "inputs"')
                    rows = len(next(iter(cols.values()))) if cols else 0
                    if rows == 0:
                        self._send(200, {key: []})
                        return
                    if feats is None:
                        feats = _to_feature_arrays(cols, rows)
                    for k, v in outer.defaults.items():
                        if k not in feats:
                            feats[k] = np.array([v] * rows, dtype=object if isinstance(v, str) else None)
                    outer.batcher.withdraw()                    # (submit() queues it: no longer "arriving")
                    announced = False
                    scores = outer.batcher.submit(feats, rows)
                    self._send_scores(key, scores)
                except (ValueError, KeyError, json.JSONDecodeError) as e:
                    self._send(400, {"error": str(e)})
                except Exception as e:                           # engine failure: TF Serving answers 500 too
                    self._send(500, {"error": "%s: %s" % (type(e).__name__, e)})
                finally:
                    if announced:
                        outer.batcher.withdraw()

        class _Server(ThreadingHTTPServer):
            def server_bind(s2):
                if reuse_port:                                   # several worker PROCESSES accept on one port (serve_workers): the kernel spreads the connections
                    import socket
                    s2.socket.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEPORT, 1)
                ThreadingHTTPServer.server_bind(s2)

        self.httpd = _Server((host, port), Handler)
        self.httpd.daemon_threads = True
        self.port = self.httpd.server_address[1]
        self.thread: Optional[threading.Thread] = None

    def _predict(self, feats):
        return self.model.predict(feats)

    def start(self):
        self.thread = threading.Thread(target=self.httpd.serve_forever, daemon=True, name="sparrow-serving")
        self.thread.start()
        return self

    def serve_forever(self):
        self.httpd.serve_forever()

    def close(self):
        self.httpd.shutdown()
        self.httpd.server_close()
        self.batcher.close()


# ---------------------------------------------------------------------------------------------------------------------------
# [r5] Several worker PROCESSES behind one port.  One Python process is one GIL: however many clients, the HTTP + JSON work of this
# shim tops out near 4 000 requests of 800 candidates per second (profiles/r04/experiments/r04_24), while the forward behind it takes
# microseconds.  serve_workers starts N FRONT processes -- each a PredictServer bound to the same (host, port) with SO_REUSEPORT, so the
# kernel hands each new connection to one of them (Jetty's HttpClient opens one per request thread, RecForYouProcess.java:113-138) --
# and ONE engine process that owns the model and the GPU.  A front parses the request and hands the packed feature columns to the
# engine over a Unix-domain connection; the engine loop takes whatever requests are waiting on ANY front, merges those with the same
# feature set into one model.predict, and sends each front its slice (so concurrent clients are micro-batched ACROSS fronts too).
# Measured first and not kept (profiles/r05/experiments/r05_05): an engine per worker process -- eight HIP contexts time-slicing one
# GPU: 5 993 requests/s from 8 clients against 4 416 from one process, p99 13 ms, and a lone client fell from 3 364 to 1 514 requests/s.
# ---------------------------------------------------------------------------------------------------------------------------
class _EngineProxy:
    """What a front process hands PredictServer as its model: predict(feats) = one round trip to the engine process.  [r6, ADVICE r05] Every
    request carries an id and the reply must echo it: a reply that is out of step (it cannot be, the engine answers each request once -- but a
    bug there would hand a client another client's scores, same candidate count, nothing else to notice it by) is an error, not an answer.  A dead
    engine (EOF / reset on the connection) stops the front: the pool's parent sees a front exit instead of a port that answers 500 forever."""

    def __init__(self, address, on_engine_lost=None):
        from multiprocessing.connection import Client
        self.conn = Client(address, family="AF_UNIX")
        self._lock = threading.Lock()
        self._seq = 0
        self._on_engine_lost = on_engine_lost

    def predict(self, feats):
        with self._lock:                                         # (PredictServer runs one forward at a time anyway)
            self._seq += 1
            rid = (os.getpid(), self._seq)
            try:
                self.conn.send((rid, feats))
                got, kind, payload = self.conn.recv()
            except (EOFError, OSError) as e:
                if self._on_engine_lost:
                    self._on_engine_lost()
                raise RuntimeError("the engine process is gone (%s: %s)" % (type(e).__name__, e))
        if got != rid:
            raise RuntimeError("engine reply out of step: sent request %r, got the answer to %r" % (rid, got))
        if kind == "ok":
            return payload
        if kind == "value_error":
            raise ValueError(payload)
        raise RuntimeError(payload)


def _merge_key(feats):
    """Requests merge into one forward only when np.concatenate cannot change what any of them means: same keys, same dtype kind and item
    size, same trailing shape per key ([r6, ADVICE r05]: an int and a float column of the same name used to be upcast together)."""
    return tuple((k, np.asarray(v).dtype.kind, np.asarray(v).dtype.itemsize, np.asarray(v).shape[1:]) for k, v in sorted(feats.items()))


def _engine_main(factory, fargs, address, ready, stop):
    """The engine process: builds the model, then serves the fronts' connections -- every pass takes ALL requests that are waiting,
    merges the ones with equal feature sets into one forward, answers each front ONCE.  [r6, ADVICE r05] computing and sending are
    separate failures: a front that died with a request in flight loses its connection and nothing else happens -- round 5's blanket
    handler re-answered every member of the merged group, so live fronts got a second reply and were one reply out of step from then on."""
    from multiprocessing.connection import Listener, wait
    try:
        model = factory(*fargs)
        listener = Listener(address, family="AF_UNIX")
    except Exception as e:
        ready.put((os.getpid(), 0, "%s: %s" % (type(e).__name__, e)))
        return
    ready.put((os.getpid(), 0, None))
    conns = []
    accept_q = queue.Queue()

    def acceptor():
        while not stop.is_set():
            try:
                accept_q.put(listener.accept())
            except Exception:
                return
    threading.Thread(target=acceptor, daemon=True).start()

    def drop(conn):
        if conn in conns:
            conns.remove(conn)
        try:
            conn.close()
        except Exception:
            pass

    def send(conn, rid, kind, payload):
        try:
            conn.send((rid, kind, payload))
        except (OSError, EOFError, ValueError, BrokenPipeError):  # the front is gone: its request dies with it
            drop(conn)

    def compute(feats):
        """-> (kind, payload): never raises"""
        try:
            return "ok", np.asarray(model.predict(feats), dtype=np.float32).reshape(-1)
        except ValueError as e:
            return "value_error", str(e)
        except Exception as e:
            return "error", "%s: %s" % (type(e).__name__, e)
    parent = os.getppid()
    while not stop.is_set() and os.getppid() == parent:          # (the parent killed without close(): do not outlive it)
        while not accept_q.empty():
            conns.append(accept_q.get())
        if not conns:
            time.sleep(0.01)
            continue
        got = []
        for c in wait(conns, timeout=0.05):
            try:
                rid, feats = c.recv()
                got.append((c, rid, feats))
            except (EOFError, OSError):
                drop(c)
            except Exception:                                    # not a (rid, feats) pair: a front of another version; nothing to answer to
                drop(c)
        if not got:
            continue
        groups = {}
        for c, rid, feats in got:
            try:
                key = _merge_key(feats)
            except Exception:
                key = ("unmergeable", id(feats))
            groups.setdefault(key, []).append((c, rid, feats))
        for members in groups.values():
            if len(members) == 1:
                c, rid, feats = members[0]
                send(c, rid, *compute(feats))
                continue
            sizes = [len(next(iter(f.values()))) for _, _, f in members]
            kind, out = compute({k: np.concatenate([np.asarray(f[k]) for _, _, f in members]) for k in members[0][2]})
            if kind == "ok" and out.shape[0] == sum(sizes):
                pos = 0
                for (c, rid, _), n in zip(members, sizes):
                    send(c, rid, "ok", out[pos:pos + n])
                    pos += n
            elif kind == "value_error":                          # one member's id is out of range: each on its own, the others get their scores
                for c, rid, f in members:
                    send(c, rid, *compute(f))
            else:                                                # the forward itself failed: the same answer for all, no second forward each
                for c, rid, _ in members:
                    send(c, rid, "error", out if kind == "error" else "merged forward returned %d scores for %d rows" % (out.shape[0], sum(sizes)))
    listener.close()


def _pid_runs(pid) -> bool:
    """The process exists and is not a zombie waiting for its parent."""
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    except PermissionError:
        return True
    try:
        with open("/proc/%d/stat" % pid) as f:
            return f.read().rsplit(")", 1)[1].split()[0] != "Z"
    except OSError:
        return True


def _front_main(address, name, host, port, defaults, ready, stop, engine_pid=0):
    try:
        lost = threading.Event()
        srv = PredictServer(_EngineProxy(address, on_engine_lost=lost.set), name=name, host=host, port=port, defaults=defaults, reuse_port=True).start()
        ready.put((os.getpid(), srv.port, None))
        parent = os.getppid()
        while not stop.wait(0.5):
            if lost.is_set() or os.getppid() != parent or (engine_pid and not _pid_runs(engine_pid)):
                break                                            # the engine died, or the parent did (a plain `kill`): do not stay on the port
        srv.close()
    except Exception as e:                                       # the parent raises it
        ready.put((os.getpid(), 0, "%s: %s" % (type(e).__name__, e)))


class WorkerPool:
    """``serve_workers``' handle: ``port``, ``pids`` (the fronts), ``engine_pid``; ``close()`` stops everything."""

    def __init__(self, procs, stop, port, pids, engine_pid, tmpdir):
        self.procs, self._stop, self.port, self.pids, self.engine_pid, self._tmpdir = procs, stop, port, pids, engine_pid, tmpdir

    def alive(self) -> bool:
        """Every process of the pool (the engine first) still runs."""
        return all(p.is_alive() for p in self.procs)

    def close(self):
        self._stop.set()
        for p in self.procs:
            p.join(timeout=10)
            if p.is_alive():
                p.terminate()
        if self._tmpdir:
            import shutil
            shutil.rmtree(self._tmpdir, ignore_errors=True)


def serve_workers(factory, fargs=(), n_workers: int = 4, name: str = "recmodel", host: str = "127.0.0.1", port: int = 8501,
                  defaults: Optional[Mapping[str, object]] = None, start_method: str = "spawn", timeout_s: float = 300.0) -> WorkerPool:
    """Starts ONE engine process that calls ``factory(*fargs)`` (picklable: a module-level function) for the model, and ``n_workers`` front
    processes that serve it on the same ``host:port`` (``port=0``: the first front picks a free one, the others join it).  Returns
    when every process answers; raises what one of them raised."""
    import multiprocessing as mp
    import tempfile
    ctx = mp.get_context(start_method)
    ready, stop = ctx.Queue(), ctx.Event()
    tmpdir = tempfile.mkdtemp(prefix="sprk_serving_")
    address = os.path.join(tmpdir, "engine.sock")
    procs, pids = [], []

    def take():
        pid, got, err = ready.get(timeout=timeout_s)
        if err:
            raise RuntimeError("serving process %d failed: %s" % (pid, err))
        return pid, got
    try:
        eng = ctx.Process(target=_engine_main, args=(factory, tuple(fargs), address, ready, stop), daemon=True)
        eng.start()
        procs.append(eng)
        engine_pid, _ = take()                                   # (the model is built, the engine listens)
        for i in range(n_workers):
            p = ctx.Process(target=_front_main, args=(address, name, host, port, dict(defaults or {}), ready, stop, engine_pid), daemon=True)
            p.start()
            procs.append(p)
            if i == 0 or port == 0:                              # the port must be known before the next front binds it
                pid, port = take()
                pids.append(pid)
        while len(pids) < n_workers:
            pids.append(take()[0])
    except Exception:
        stop.set()
        for p in procs:
            p.join(timeout=5)
            if p.is_alive():
                p.terminate()
        raise
    return WorkerPool(procs, stop, port, pids, engine_pid, tmpdir)


def _cli_model(model_name, weights_path):
    from . import models as M
    cls = {"neuralcf": M.NeuralCF, "embedding_mlp": M.EmbeddingMLP, "wide_n_deep": M.WideNDeep, "deepfm": M.DeepFM,
           "deepfm_v2": M.DeepFMv2, "din": M.DIN, "dien": M.DIEN}[model_name]
    weights = dict(np.load(weights_path)) if weights_path else None
    model = cls(weights=weights, seed=None if weights else 0)
    model.engine                                           # fail loudly now if the HIP library / device is missing
    return model


def _main():
    from . import models as M
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--model", default="neuralcf", choices=["neuralcf", "embedding_mlp", "wide_n_deep", "deepfm", "deepfm_v2", "din", "dien"])
    ap.add_argument("--weights", help=".npz of reference-layout weights (keys as CTRModel.weight_shapes()); default: seeded random")
    ap.add_argument("--name", default="recmodel")
    ap.add_argument("--host", default="127.0.0.1")
    ap.add_argument("--port", type=int, default=8501)
    ap.add_argument("--workers", type=int, default=1, help="front processes sharing the port (SO_REUSEPORT) in front of ONE engine process that owns the model and the GPU")
    args = ap.parse_args()
    if args.workers > 1:
        import signal
        pool = serve_workers(_cli_model, (args.model, args.weights), n_workers=args.workers, name=args.name, host=args.host, port=args.port)
        print("serving %s on http://%s:%d/v1/models/%s:predict from %d workers (pids %s)" % (args.model, args.host, pool.port, args.name, args.workers, pool.pids), flush=True)
        # [r6, ADVICE r05] a plain `kill` (SIGTERM) must take the fronts and the engine along, and a dead engine must end the service
        def _term(signum, frame):
            raise KeyboardInterrupt
        signal.signal(signal.SIGTERM, _term)
        try:
            while pool.alive():
                time.sleep(1.0)
            print("a serving process exited: shutting down", flush=True)
        except KeyboardInterrupt:
            pass
        finally:
            pool.close()
        return
    cls = {"neuralcf": M.NeuralCF, "embedding_mlp": M.EmbeddingMLP, "wide_n_deep": M.WideNDeep, "deepfm": M.DeepFM,
           "deepfm_v2": M.DeepFMv2, "din": M.DIN, "dien": M.DIEN}[args.model]
    weights = dict(np.load(args.weights)) if args.weights else None
    model = cls(weights=weights, seed=None if weights else 0)
    model.engine                                           # fail loudly now if the HIP library / device is missing
    srv = PredictServer(model, name=args.name, host=args.host, port=args.port)
    print("serving %s on http://%s:%d/v1/models/%s:predict" % (args.model, args.host, srv.port, args.name), flush=True)
    try:
        srv.serve_forever()
    except KeyboardInterrupt:
        pass
    finally:
        srv.close()


if __name__ == "__main__":
    _main()
