"""world_size-2 gloo run of the frame-sharded stream driver on CPU: shard arithmetic, the
recomputed halo chunks and the single gather must reproduce the unsharded result (golden 'stream',
generated from the reference's own functions)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden, rel_err


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_backend import OracleBackend
    from passiveradar_amd.stream import StreamProcessor
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = load_golden("stream")
    C, R, F = int(g["C"]), int(g["R"]), int(g["F"])
    sp = StreamProcessor(OracleBackend(2 * C, R, F, float(g["fs"])), rank, world)
    frames, sh = sp.process_local(g["ref"], g["srv"])
    full = sp.process(g["ref"], g["srv"])
    # the async form used by bench.py returns the same result once its work handle completed
    from passiveradar_amd.stream import gather_frames
    res2, work = gather_frames(frames, sh, async_op=True)
    work.wait()
    if rank == 0:
        assert torch.equal(res2, full)
    if rank == 0:
        q.put((full.numpy(), (sh.frame_lo, sh.frame_hi, sh.chunk_lo, sh.chunk_hi)))
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_stream_matches_golden(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    full, sh0 = q.get(timeout=240)
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    g = load_golden("stream")
    ref = np.moveaxis(g["out"], 2, 0)           # (F, R+1, nframes) -> [nframes][F][R+1]
    assert full.shape == ref.shape
    assert rel_err(full, ref) < 1e-5
    assert sh0[0] == 0 and sh0[2] == 0


class _CheapBackend:
    """A stand-in backend whose per-chunk 'canceller' and per-frame 'map' are cheap but sensitive to exactly
    the things sharding can get wrong: which chunk a sample belongs to (per-chunk statistics, like the per-chunk
    LS taps) and which three chunks a frame touches (zeros beyond the stream ends)."""

    def __init__(self, C, F, cols):
        self.C, self.cpi, self.F, self.R = C, 2 * C, F, cols - 1
        self.device = "cpu"

    def padded(self, chunks):
        x = np.asarray(chunks, dtype=np.complex64)
        z = np.zeros(self.C // 2, np.complex64)
        return np.concatenate((z, x, z))

    def clean(self, ref_pad, srv_pad, nlocal):
        C, h = self.C, self.C // 2
        out = np.zeros_like(srv_pad)
        for c in range(nlocal):
            sl = slice(h + c * C, h + (c + 1) * C)
            out[sl] = srv_pad[sl] - srv_pad[sl].mean() * ref_pad[sl]          # per-chunk coefficient
        return out

    def frames(self, ref_pad, clean_pad, first, nframes):
        k = np.arange(self.F * (self.R + 1)).reshape(self.F, self.R + 1)
        fr = []
        for i in range(nframes):
            a = ref_pad[first + i * self.C:first + i * self.C + self.cpi]
            s = clean_pad[first + i * self.C:first + i * self.C + self.cpi]
            w = np.arange(1, self.cpi + 1)
            fr.append(((a * np.conj(s) * w).sum() * np.exp(1j * 0.01 * k)).astype(np.complex64))
        return torch.from_numpy(np.stack(fr))


def _stream_1199(C=8):
    rng = np.random.default_rng(1199)
    n = 1199 * C
    ref = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    srv = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64) + ref
    return ref, srv


def _worker_1199(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from passiveradar_amd.stream import StreamProcessor, gather_frames, plan_shard, shard_sizes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ref, srv = _stream_1199()
    sp = StreamProcessor(_CheapBackend(8, 4, 3), rank, world)
    frames, sh = sp.process_local(ref, srv)
    assert sh == plan_shard(1199, rank, world) and frames.shape[0] == shard_sizes(sh)[rank]
    full = sp.process(ref, srv)
    # two gathers into caller-owned buffers (what bench.py does every step): results stay distinct
    outs = [torch.empty((1199, 4, 3), dtype=torch.complex64) if rank == 0 else None for _ in range(2)]
    r0, w0 = gather_frames(frames, sh, async_op=True, out=outs[0])
    r1, w1 = gather_frames(frames * 2, sh, async_op=True, out=outs[1])
    w0.wait(); w1.wait()
    if rank == 0:
        assert r0 is outs[0] and torch.equal(r0, full) and torch.equal(r1, full * 2)
        q.put(full.numpy())
    else:
        assert full is None and r0 is None
    dist.barrier()
    dist.destroy_process_group()


def test_cfg4_stream_of_1199_frames_sharded_over_8_ranks():
    """BASELINE config 4's shape of work: 1199 frames (600 s of cfg-2 IQ), contiguous shards over 8 ranks
    (7 x 150 + 149), halo chunks re-filtered locally, one gather -- equals the unsharded pass."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_1199, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    full = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    from passiveradar_amd.stream import StreamProcessor
    ref, srv = _stream_1199()
    single = StreamProcessor(_CheapBackend(8, 4, 3)).process(ref, srv).numpy()
    assert full.shape == (1199, 4, 3) and np.array_equal(full, single)


def _part_worker(rank, world, port, q):
    from passiveradar_amd.stream import PartGather, plan_shard, shard_sizes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total, F, cols, nparts = 1199 if world == 8 else 11, 3, 2, 4
    sh = plan_shard(total, rank, world)
    mine = torch.arange(sh.frame_lo, sh.frame_hi, dtype=torch.float32).reshape(-1, 1, 1).repeat(1, F, cols).to(torch.complex64)
    mine = mine + 1j * float(rank)
    pg = PartGather(sh, nparts)
    result = torch.zeros((total, F, cols), dtype=torch.complex64) if rank == 0 else None
    works = []
    for p in range(nparts):
        a, b = pg.part_range(p)
        if p % 2:                                   # both forms of the call
            pg.gather_part(p, mine[a:b], result)
        else:
            works.append(pg.gather_part(p, mine[a:b], result, async_op=True))
    for w in works:
        w.wait()
    if rank == 0:
        sizes = shard_sizes(sh)
        owner = torch.cat([torch.full((m,), float(r)) for r, m in enumerate(sizes)])
        exp = torch.arange(total, dtype=torch.float32).reshape(-1, 1, 1).repeat(1, F, cols).to(torch.complex64)
        exp = exp + 1j * owner.reshape(-1, 1, 1)
        q.put(bool(torch.equal(result, exp)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [3, 8])
def test_part_gather_assembles_the_frame_ordered_array(world):
    """PartGather: a frame-sharded array moved to the root in four rounds (what bench.py --workload cfg4 does so that
    finished frames travel under the compute of the rest): ragged shards (1199 frames over 8 ranks; 11 over 3, where
    parts are empty), sync and async rounds, every frame at its place with its owner's mark"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_part_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok = q.get(timeout=240)
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    assert ok


def _failing_worker(rank, world, port, q):
    """rank 1 throws before the collective the other ranks are already waiting in"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from mp_util import report_failure_and_leave, report_ok
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if rank == 1:
            raise RuntimeError("rank 1 fails before the barrier")
        dist.barrier()                          # never completes: rank 1 is gone
        report_ok(q, rank)
    except Exception as e:                      # noqa: BLE001
        report_failure_and_leave(q, rank, e)


def test_a_failing_rank_ends_the_run_at_once():
    """the multi-rank GPU test (tests/test_gpu_gather_rccl.py) runs through mp_util.run_ranks: a rank that throws must
    fail the test within seconds, with its traceback, and leave no process behind -- not block the healthy ranks in a
    collective until a 600 s timeout"""
    import time
    from mp_util import run_ranks
    t0 = time.time()
    ok, got, codes = run_ranks(mp.get_context("spawn"), _failing_worker, 3, args=(_free_port(),), deadline_s=120)
    took = time.time() - t0
    assert not ok and took < 60, (took, got, codes)
    assert any(r == 1 and "rank 1 fails before the barrier" in msg for r, msg in got), got
    assert all(c is not None for c in codes), codes          # every process is gone (killed or exited)


def test_part_gather_edge_cases_single_rank():
    """a world of one with async_op (no work handle to wait for) and a round in which nobody has frames"""
    from passiveradar_amd.stream import PartGather, Shard, gather_frames
    sh = Shard(0, 1, 3, 0, 3, 0, 3)
    mine = torch.arange(3 * 4, dtype=torch.float32).reshape(3, 2, 2).to(torch.complex64)
    res, work = gather_frames(mine, sh, async_op=True)
    work.wait()                                  # must not raise although nothing was communicated
    assert torch.equal(res, mine)
    pg = PartGather(sh, 5)                       # more rounds than frames: two of them are empty
    result = torch.zeros_like(mine)
    works = [pg.gather_part(p, mine[slice(*pg.part_range(p))], result, async_op=True) for p in range(5)]
    for w in works:
        w.wait()
    assert torch.equal(result, mine)
