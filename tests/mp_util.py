"""One process per rank for the multi-process tests, with a bounded, fail-fast join.

A rank that throws reports its traceback and leaves WITHOUT taking part in any further collective; the parent ends the
run at the first failure (a reported exception, a rank that died without reporting, the deadline) and kills the ranks it
started -- healthy ranks are never left waiting inside a collective for a peer that is gone."""
import os
import queue as _queue
import time
import traceback


def report_ok(q, rank):
    q.put((rank, "ok"))


def report_failure_and_leave(q, rank, exc):
    """called from a worker's except block: ship the traceback, flush the queue's feeder thread, exit hard"""
    q.put((rank, repr(exc) + "\n" + traceback.format_exc(limit=8)))
    q.close()
    q.join_thread()
    os._exit(3)


def run_ranks(ctx, target, world, args=(), deadline_s=420.0):
    """start target(rank, world, *args, q) for every rank; returns (ok, reports, exitcodes)"""
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world) + tuple(args) + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    got, deadline = [], time.time() + deadline_s
    while len(got) < world and time.time() < deadline:
        try:
            got.append(q.get(timeout=1))
        except _queue.Empty:
            reported = {r for r, _ in got}
            if any(p.exitcode not in (None, 0) and r not in reported for r, p in enumerate(procs)):
                break                           # a rank died without a report (segfault, kill)
            continue
        if got[-1][1] != "ok":
            break
    ok = len(got) == world and all(msg == "ok" for _, msg in got)
    for p in procs:
        p.join(timeout=30 if ok else 0.1)
        if p.is_alive():
            p.kill()                            # exactly the processes started above
            p.join(timeout=10)
    codes = [p.exitcode for p in procs]
    return ok and all(c == 0 for c in codes), got, codes
