"""Batches of independent QPs (BASELINE configs[3]): the path shards trivially over problems.

Across GPUs: rank r of N takes problems ``r, r+N, ...`` (no communication).  Inside one GPU the
solver of a small QP is launch bound, so several problems are kept in flight: each worker thread owns
one context = one HIP stream (threading) and drives its own :class:`MPCSolver`; the
C ABI calls release the GIL, kernels of different problems overlap on the device.
"""
from __future__ import annotations

import queue
import threading

import torch

from .backend import HipBackend
from .solver import MPCSolver


def shard(items, rank: int, world: int):
    """Round-robin assignment of problems to ranks (QP b -> GPU b mod N)."""
    return list(items)[rank::world]


def solve_batch(make_qp, indices, device_index: int = 0, streams: int = 8, **opts):
    """Solve the QPs ``make_qp(backend, i)`` for i in ``indices`` with ``streams`` concurrent contexts.

    Returns ``{i: result}`` (the dict of :meth:`MPCSolver.result`, per-problem status and iteration
    count -- problems converge independently)."""
    todo = queue.Queue()
    for i in indices:
        todo.put(i)
    results, errors = {}, []
    lock = threading.Lock()

    def worker():
        stream = torch.cuda.Stream(device=device_index)
        with torch.cuda.stream(stream):
            be = HipBackend(device_index, stream=stream)
            try:
                while True:
                    try:
                        i = todo.get_nowait()
                    except queue.Empty:
                        break
                    solver = MPCSolver(make_qp(be, i), be, **opts)
                    r = solver.solve()
                    solver.close()
                    with lock:
                        results[i] = r
            except Exception as e:  # surfaced to the caller: no silent fallback
                with lock:
                    errors.append(e)
            finally:
                stream.synchronize()
                be.close()

    threads = [threading.Thread(target=worker) for _ in range(max(1, min(streams, todo.qsize())))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return results
