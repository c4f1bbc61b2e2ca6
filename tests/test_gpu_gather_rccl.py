"""prc_gather_frames between REAL ranks: two (or more) processes, one GPU each, RCCL through the C ABI.

Skipped on a box with fewer than two GPUs (the driver's GPU test box has one); on a multi-GPU node it is the
only test that runs the multi-rank branch of passiveradar_amd/csrc/gather.hip (ncclSend on the peers, the grouped
ncclRecv set on the root): ragged and empty blocks, a root other than rank 0, a FrameComm over a torch sub-group whose
ranks differ from the global ones, and the sharded stream driver end to end against the unsharded pass."""
import os
import socket

import numpy as np
import pytest

from mp_util import report_failure_and_leave, report_ok, run_ranks

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from passiveradar_amd import scene
    from passiveradar_amd.stream import FrameComm, HipBackend, Shard, StreamProcessor, gather_frames, shard_sizes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        comm = FrameComm.from_torch_distributed()
        # what RCCL itself says about the communicator (prc_comm_count): the number of ranks it connected, this rank's id
        assert comm.count() == (world, rank), comm.count()
        probe = torch.arange(1000, dtype=torch.float32, device=dev) + rank
        echo = torch.zeros_like(probe)
        comm.loopback(probe, echo)                # the self-peer link check on a communicator of several ranks
        torch.cuda.synchronize()
        assert torch.equal(probe, echo)
        F, cols = 8, 5
        # ragged blocks (ceil split of 2 * world + 1 frames), root 0 and root world - 1
        total = 2 * world + 1
        per = -(-total // world)
        lo, hi = min(rank * per, total), min((rank + 1) * per, total)
        sh = Shard(rank, world, total, lo, hi, lo, hi)
        mine = torch.full((hi - lo, F, cols), float(rank + 1), dtype=torch.complex64, device=dev)
        mine += torch.arange(lo, hi, device=dev, dtype=torch.float32).reshape(-1, 1, 1) * 1j
        for root in (0, world - 1):
            res = gather_frames(mine, sh, dst=root, comm=comm)
            if rank == root:
                sizes = shard_sizes(sh)
                exp = torch.cat([torch.full((m, F, cols), float(r + 1), dtype=torch.complex64) for r, m in enumerate(sizes)])
                exp += torch.arange(total, dtype=torch.float32).reshape(-1, 1, 1) * 1j
                assert torch.equal(res.cpu(), exp), f"root {root}"
            else:
                assert res is None
        # the same array in three rounds (PartGather: finished frames travel while the rest is computed)
        from passiveradar_amd.stream import PartGather
        pg = PartGather(sh, 3, comm=comm)
        result = torch.zeros((total, F, cols), dtype=torch.complex64, device=dev) if rank == 0 else None
        works = [pg.gather_part(p_, mine[slice(*pg.part_range(p_))], result, async_op=True) for p_ in range(3)]
        for w_ in works:
            w_.wait()
        if rank == 0:
            sizes = shard_sizes(sh)
            exp = torch.cat([torch.full((m, F, cols), float(r + 1), dtype=torch.complex64) for r, m in enumerate(sizes)])
            exp += torch.arange(total, dtype=torch.float32).reshape(-1, 1, 1) * 1j
            assert torch.equal(result.cpu(), exp), "PartGather"
        # the sharded stream driver against the unsharded pass on rank 0
        C, R, Fd, fs = 16384, 24, 64, 1.0e5
        ref, srv = scene.make_stream(3 * world + 1, C, fs, R, 2026)
        be = HipBackend(2 * C, R, Fd, fs, batch=8, device=dev)
        full = StreamProcessor(be, rank, world, comm=comm).process(ref, srv)
        if rank == 0:
            one = StreamProcessor(HipBackend(2 * C, R, Fd, fs, batch=8, device=dev)).process(ref, srv)
            err = float((full - one).abs().max() / one.abs().max())
            assert err < 1e-6, err
        comm.close()
        report_ok(q, rank)
        # a clean exit only: every rank got here, so this barrier cannot wait for a rank that threw (a failing rank
        # reports and leaves at once below; the parent then ends the others instead of letting them sit in a collective)
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:                      # noqa: BLE001 -- reported to the parent, which fails the test
        report_failure_and_leave(q, rank, e)    # no collective on the way out: the other ranks may be inside one


def test_prc_gather_frames_between_ranks(gpu_ready):
    import torch
    import torch.multiprocessing as mp
    world = min(torch.cuda.device_count(), 4)
    if world < 2:
        pytest.skip("needs at least two GPUs: the multi-rank RCCL branch of prc_gather_frames")
    ctx = mp.get_context("spawn")
    ok, got, codes = run_ranks(ctx, _worker, world, args=(_free_port(),), deadline_s=420)
    assert ok, (got, codes)
