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
        self._stop = False
        self.thread = threading.Thread(target=self._run, daemon=True, name="sparrow-batcher")
        self.thread.start()

    def submit(self, feats: Dict[str, np.ndarray], n: int) -> np.ndarray:
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
                    nxt = self.q.get(timeout=max(0.0, deadline - time.monotonic()))
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


class PredictServer:
    """``POST /v1/models/<name>:predict`` in front of ``model.predict`` (any object with a Keras-shaped
    ``predict(dict) -> [N,1]``).  ``defaults`` fills feature columns a request does not send (the Jetty
    ranker only sends userId / movieId)."""

    def __init__(self, model, name: str = "recmodel", host: str = "127.0.0.1", port: int = 8501,
                 defaults: Optional[Mapping[str, object]] = None, max_wait_s: float = 0.0005):
        self.model, self.name, self.defaults = model, name, dict(defaults or {})
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
                body = ('{"%s": [[' % key + "], [".join(["%.9g" % v for v in arr.tolist()]) + "]]}").encode("utf-8")
                self.send_response(200)
                self.send_header("Content-Type", "application/json")
                self.send_header("Content-Length", str(len(body)))
                self.end_headers()
                self.wfile.write(body)
                self.wfile.flush()

            def do_GET(self):
                if self.path.rstrip("/") == "/v1/models/" + outer.name:
                    self._send(200, {"model_version_status": [{"version": "1", "state": "AVAILABLE",
                                                                "status": {"error_code": "OK", "error_message": ""}}]})
                else:
                    self._send(404, {"error": "Not found: %s" % self.path})

            def do_POST(self):
                if self.path != "/v1/models/%s:predict" % outer.name:
                    self._send(404, {"error": "Not found: %s" % self.path})
                    return
                try:
                    n = int(self.headers.get("Content-Length", "0"))
                    req = json.loads(self.rfile.read(n).decode("utf-8"))
                    if not isinstance(req, dict):
                        raise ValueError("request body must be a JSON object")
                    feats = None
                    if "instances" in req:
                        key = "predictions"
                        try:                                     # uniform scalar instances: the fast conversion
                            cols = _columns_from_instances(req["instances"], fast=True)
                            rows = len(req["instances"])
                            feats = _to_feature_arrays(cols, rows) if rows and all(len(i) == len(cols) for i in req["instances"]) else None
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
                    scores = outer.batcher.submit(feats, rows)
                    self._send_scores(key, scores)
                except (ValueError, KeyError, json.JSONDecodeError) as e:
                    self._send(400, {"error": str(e)})
                except Exception as e:                           # engine failure: TF Serving answers 500 too
                    self._send(500, {"error": "%s: %s" % (type(e).__name__, e)})

        self.httpd = ThreadingHTTPServer((host, port), Handler)
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


def _main():
    from . import models as M
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--model", default="neuralcf", choices=["neuralcf", "embedding_mlp", "wide_n_deep", "deepfm", "deepfm_v2", "din", "dien"])
    ap.add_argument("--weights", help=".npz of reference-layout weights (keys as CTRModel.weight_shapes()); default: seeded random")
    ap.add_argument("--name", default="recmodel")
    ap.add_argument("--host", default="127.0.0.1")
    ap.add_argument("--port", type=int, default=8501)
    args = ap.parse_args()
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
