#!/usr/bin/env python
"""Host <-> device copy rates for a 25 MB POI queue (config B): pageable vs pinned, whole vs in 4 chunks, and the
host-side memcpy rate into a pinned buffer with 1..4 threads.  (GPU box; informs capi.hip's host-queue pipeline.)"""
import json
import threading
import time

import numpy as np
import torch

n = 250000 * 25
dev = torch.device("cuda", 0)
pageable = torch.from_numpy(np.random.default_rng(0).random(n, dtype=np.float32))
pinned = torch.empty(n, dtype=torch.float32).pin_memory()
pinned.copy_(pageable)
d = torch.empty(n, dtype=torch.float32, device=dev)
out = {}


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best


mb = n * 4 / 1e6
out["h2d_pageable_GBs"] = mb / 1e3 / timed(lambda: d.copy_(pageable, non_blocking=True))
out["h2d_pinned_GBs"] = mb / 1e3 / timed(lambda: d.copy_(pinned, non_blocking=True))
back_pg = torch.empty(n, dtype=torch.float32)
out["d2h_pageable_GBs"] = mb / 1e3 / timed(lambda: back_pg.copy_(d, non_blocking=True))
out["d2h_pinned_GBs"] = mb / 1e3 / timed(lambda: pinned.copy_(d, non_blocking=True))
src = pageable.numpy(); dst = pinned.numpy()
for nthreads in (1, 2, 4, 8):
    def work(k):
        lo, hi = k * n // nthreads, (k + 1) * n // nthreads
        np.copyto(dst[lo:hi], src[lo:hi])
    def run():
        ts = [threading.Thread(target=work, args=(k,)) for k in range(nthreads)]
        [t.start() for t in ts]; [t.join() for t in ts]
    best = 1e9
    for _ in range(10):
        t0 = time.perf_counter(); run(); best = min(best, time.perf_counter() - t0)
    out["host_memcpy_%d_threads_GBs" % nthreads] = mb / 1e3 / best
print(json.dumps(out))
