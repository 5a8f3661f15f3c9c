"""The REST shim under load: what the Jetty server does per recommendation request (RecForYouProcess.java:113-138 -- one POST of
800 {"userId", "movieId"} instances to /v1/models/recmodel:predict), from C concurrent keep-alive clients for a few seconds.
Reports requests/s, candidates/s and the latency percentiles; the forward itself is microseconds, so this measures the Python
HTTP + JSON path and the micro-batcher in front of the GPU.

    python scripts/bench_serving.py [--clients 16] [--seconds 5] [--instances 800]
"""
import argparse
import http.client
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _client(port, bodies, seconds, instances, k, q):
    c = http.client.HTTPConnection("127.0.0.1", port, timeout=30)
    c.request("POST", "/v1/models/recmodel:predict", body=bodies[0], headers={"Content-Type": "application/json"})   # warm-up
    c.getresponse().read()
    mine, i = [], k
    stop = time.perf_counter() + seconds
    data = b""
    while time.perf_counter() < stop:
        t0 = time.perf_counter()
        c.request("POST", "/v1/models/recmodel:predict", body=bodies[i % len(bodies)], headers={"Content-Type": "application/json"})
        r = c.getresponse()
        data = r.read()
        assert r.status == 200, data[:200]
        mine.append(time.perf_counter() - t0)
        i += 1
    assert len(json.loads(data)["predictions"]) == instances
    q.put(mine)


def _neuralcf():
    from sparrowrecsys_amd import models as M
    m = M.NeuralCF(seed=7)
    m.engine
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clients", type=int, default=16)
    ap.add_argument("--seconds", type=float, default=5.0)
    ap.add_argument("--instances", type=int, default=800)
    ap.add_argument("--workers", type=int, default=1, help="front processes behind one port (serving.serve_workers: SO_REUSEPORT fronts, ONE engine process with the model)")
    a = ap.parse_args()
    import numpy as np
    from sparrowrecsys_amd import models as M
    from sparrowrecsys_amd.serving import PredictServer, serve_workers
    if a.workers > 1:
        srv = serve_workers(_neuralcf, (), n_workers=a.workers, port=0)
        port = srv.port
    else:
        model = M.NeuralCF(seed=7)
        srv = PredictServer(model, port=0)
        srv.start()
        port = srv.httpd.server_address[1] if hasattr(srv, "httpd") else srv.port
    rng = np.random.default_rng(1)
    bodies = []
    for _ in range(32):
        u = int(rng.integers(1, 30000))
        inst = [{"userId": u, "movieId": int(m)} for m in rng.integers(1, 1000, a.instances)]
        bodies.append(json.dumps({"instances": inst}).encode())
    # clients live in their own processes: in this one they would share the server's GIL
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_client, args=(port, bodies, a.seconds, a.instances, k, q)) for k in range(a.clients)]
    t0 = time.perf_counter()
    for p_ in procs:
        p_.start()
    lat = []
    for _ in procs:
        lat.extend(q.get(timeout=a.seconds + 120))
    for p_ in procs:
        p_.join(timeout=30)
    wall = a.seconds                                             # every client measures for exactly this long
    srv.close()
    lat.sort()
    pct = lambda p: round(lat[min(len(lat) - 1, int(p * len(lat)))] * 1e3, 2)
    print(json.dumps({"clients": a.clients, "instances_per_request": a.instances, "requests": len(lat), "requests_per_sec": round(len(lat) / wall, 1),
                      "candidates_per_sec": round(len(lat) * a.instances / wall), "latency_ms": {"p50": pct(0.5), "p90": pct(0.9), "p99": pct(0.99)},
                      "workers": a.workers,
                      "server": "sparrowrecsys_amd.serving.PredictServer (ThreadingHTTPServer + micro-batcher)%s, model NeuralCF"
                                % (" x %d front processes on one port (SO_REUSEPORT) + one engine process" % a.workers if a.workers > 1 else "")}))


if __name__ == "__main__":
    main()
