"""Block pipeline of the reference's main.py:169-194, batched on one GPU and sharded over GPUs.

What a frame is: the IF stream is cut into chunks of C = cpi_samples/2
(config.py:71); the LS canceller runs per chunk, independently (main.py:169-176); frame i is the
CAF over stream[i*C - C/2 : i*C + 3C/2] of (ref, cleaned srv) with zeros beyond the stream ends
(da.overlap.overlap(depth=cpi/4, boundary=0), main.py:178-181), Kaiser(5.0) window (main.py:183).
nframes == nchunks, frames stacked on the last axis of the saved array (main.py:200-224).

Device layout: ref and cleaned-srv live in HBM as ONE zero-padded stream each,
[C/2 zeros | chunk 0 | chunk 1 | ... | C/2 zeros]; the LS kernels write chunk c straight into the
cleaned stream and the CAF kernels read frame i at element offset i*C with frame stride C, so the
50 % overlap costs no copy.  Frames shard contiguously over ranks; each rank re-filters the one
extra chunk each side its first/last frame touches (LS taps are per chunk, so there is no halo
exchange) and the only collective is the final gather of the (F, R+1) maps (RCCL via
torch.distributed; gloo in the CPU tests).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

__all__ = ["Shard", "plan_shard", "shard_sizes", "gather_frames", "part_bounds", "PartGather", "FrameComm", "HipBackend",
           "StreamProcessor"]


@dataclass(frozen=True)
class Shard:
    rank: int
    world: int
    nchunks: int       # chunks (== frames) in the whole stream
    frame_lo: int      # frames [frame_lo, frame_hi) belong to this rank
    frame_hi: int
    chunk_lo: int      # chunks [chunk_lo, chunk_hi) must be resident + LS-filtered on this rank
    chunk_hi: int

    @property
    def nframes(self):
        return self.frame_hi - self.frame_lo

    @property
    def nlocal_chunks(self):
        return self.chunk_hi - self.chunk_lo

    def frame_offset(self, i, C):
        """element offset of global frame i inside the local zero-padded stream
        [C/2 zeros | chunks chunk_lo..chunk_hi | C/2 zeros]"""
        return (i - self.chunk_lo) * C


def plan_shard(nchunks, rank=0, world=1):
    """Contiguous frame ranges of ceil(nchunks/world); frame i touches chunks i-1..i+1."""
    per = -(-nchunks // world)
    lo = min(rank * per, nchunks)
    hi = min(lo + per, nchunks)
    if hi <= lo:
        return Shard(rank, world, nchunks, lo, lo, lo, lo)
    return Shard(rank, world, nchunks, lo, hi, max(lo - 1, 0), min(hi + 1, nchunks))


class FrameComm:
    """prc_comm of include/prcore.h: the RCCL communicator behind prc_gather_frames, one per process
    (one process per GPU).  The 128-byte id is made on rank 0 and shipped to the other ranks through
    whatever side channel the host has; ``from_torch_distributed`` uses the already initialised
    torch.distributed group for that (object broadcast) and nothing else."""

    def __init__(self, rank, world, id_bytes):
        import ctypes as C
        from . import _lib
        self.rank, self.world = int(rank), int(world)
        h = C.c_void_p()
        buf = C.create_string_buffer(bytes(id_bytes), _lib.COMM_ID_BYTES)
        _lib.check(_lib.lib().prc_comm_create(C.byref(h), buf, self.rank, self.world))
        self._h = h

    @staticmethod
    def unique_id():
        import ctypes as C
        from . import _lib
        buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
        _lib.check(_lib.lib().prc_comm_unique_id(buf))
        return bytes(buf.raw)

    @classmethod
    def from_torch_distributed(cls, group=None):
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [None]
        if rank == 0:
            try:
                box[0] = cls.unique_id()
            except Exception as e:              # noqa: BLE001 -- every rank must still take part in the broadcast
                box[0] = e
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        if not isinstance(box[0], (bytes, bytearray)):
            raise RuntimeError(f"rank 0 could not make an RCCL id: {box[0]}")
        return cls(rank, world, box[0])

    def gather(self, send, frames_per_rank, frame_elems, recv, root=0, stream=None):
        """enqueue prc_gather_frames on ``stream`` (device pointers / tensors; recv only on the root)"""
        import ctypes as C
        from . import _lib
        from .engine import _ptr
        cnt = (C.c_int64 * self.world)(*[int(c) for c in frames_per_rank])
        _lib.check(_lib.lib().prc_gather_frames(self._h, _ptr(send), cnt, int(frame_elems), _ptr(recv), int(root),
                                                stream))

    def count(self):
        """(ranks, this rank) as RCCL reports them for the communicator it built (prc_comm_count)"""
        import ctypes as C
        from . import _lib
        n, r = C.c_int32(0), C.c_int32(0)
        _lib.check(_lib.lib().prc_comm_count(self._h, C.byref(n), C.byref(r)))
        return int(n.value), int(r.value)

    def loopback(self, send, recv, stream=None):
        """prc_comm_loopback: ``send`` -> ``recv`` (device tensors of equal size, float32 or complex64) through RCCL's
        point-to-point path with this rank as its own peer; enqueued on ``stream`` (default: the current torch stream)"""
        from . import _lib
        if send.numel() * send.element_size() != recv.numel() * recv.element_size() or not (send.is_contiguous() and recv.is_contiguous()):
            raise ValueError("loopback: send and recv must be contiguous and of equal size")
        nfloats = send.numel() * send.element_size() // 4
        import ctypes as C
        sp = _lib.torch_stream_ptr(send.device) if stream is None else C.c_void_p(stream.cuda_stream)
        _lib.check(_lib.lib().prc_comm_loopback(self._h, send.data_ptr(), recv.data_ptr(), nfloats, sp))

    def close(self):
        if getattr(self, "_h", None):
            from . import _lib
            _lib.lib().prc_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _NoWork:
    """work handle of a gather that had nothing to move (a world of one, an empty round)"""

    def wait(self):
        return None


class _StreamWork:
    """``work.wait()`` for a gather that was enqueued on a HIP stream (the C-ABI path has no work handle)"""

    def __init__(self, event, keep):
        self.event, self.keep = event, keep

    def wait(self):
        self.event.synchronize()
        self.keep = None


def shard_sizes(shard):
    """frames owned by every rank of ``shard``'s world (contiguous ceil(nchunks/world) blocks, plan_shard)"""
    per = -(-shard.nchunks // shard.world)
    return [max(min((r + 1) * per, shard.nchunks) - min(r * per, shard.nchunks), 0) for r in range(shard.world)]


def gather_frames(local, shard, group=None, dst=0, async_op=False, out=None, comm=None, stream=None, counts=None):
    """Gather per-rank frame blocks [m_r][F][R+1] (torch tensors, complex64) to rank ``dst`` -- a rank of ``comm``
    when a FrameComm is given (0 .. comm.world-1; torch.distributed is not touched), else a global
    torch.distributed rank.  Returns the full [sum m_r][F][R+1] tensor on dst (rank blocks in rank order), None
    elsewhere.  ONE collective.

    counts: frames sent by every rank (the same list on every rank); default: the contiguous ceil(nchunks/world)
    split of ``shard`` (plan_shard), i.e. the whole frame-sharded array.

    comm: a FrameComm -> prc_gather_frames (RCCL point-to-point group through the C ABI, ragged blocks
    land at their place, nothing is padded); it is enqueued on ``stream`` (a torch stream, default the
    current one).  comm=None -> torch.distributed.gather on ``group`` (gloo in the CPU tests; complex
    tensors travel as float pairs, blocks padded to the largest count).

    out: optional preallocated [sum m_r][F][R+1] complex64 result on dst -- callers that gather every
    step pass their own (ping-pong) buffers; by default a fresh tensor is returned, never a cached one.

    async_op=True returns ``(result_or_None, work)``: ``work.wait()`` before touching the result (and
    before reusing ``local``)."""
    import torch
    if shard.world == 1 and comm is None:
        if out is not None and out.data_ptr() != local.data_ptr():
            if tuple(out.shape) != tuple(local.shape):
                raise ValueError(f"gather_frames: out has shape {tuple(out.shape)}, the frames {tuple(local.shape)}")
            out.copy_(local)
            local = out
        return (local, _NoWork()) if async_op else local
    F, cols = local.shape[1], local.shape[2]
    counts = shard_sizes(shard) if counts is None else [int(c) for c in counts]
    if len(counts) != shard.world:
        raise ValueError(f"gather_frames: {len(counts)} counts for a world of {shard.world}")
    total = sum(counts)
    if comm is not None:
        # the C-ABI path knows nothing of torch.distributed: ranks, the root and the world are the communicator's own
        # (a FrameComm built over a sub-group, or over any other side channel, numbers its ranks from 0)
        if comm.world != shard.world:
            raise ValueError(f"gather_frames: the communicator has {comm.world} ranks, the shard plan {shard.world}")
        me = comm.rank
    else:
        import torch.distributed as dist
        me = dist.get_rank(group) if group is not None else dist.get_rank()
        on_dst_global = dist.get_rank() == dst           # dst is a GLOBAL rank on the torch path, also for sub-groups
    on_dst = (me == dst) if comm is not None else on_dst_global
    if total == 0:
        # a round in which no rank has frames (more rounds than frames): nothing to move, no collective, no buffers; the
        # documented contract holds all the same -- the (empty) result on dst, None elsewhere
        empty = None
        if on_dst:
            empty = out[:0] if out is not None else torch.empty((0, F, cols), dtype=torch.complex64, device=local.device)
        return (empty, _NoWork()) if async_op else empty
    if int(local.shape[0]) != counts[me]:
        raise ValueError(f"gather_frames: rank {me} holds {int(local.shape[0])} frames, counts say {counts[me]}")
    res = None
    if on_dst:
        res = out if out is not None else torch.empty((total, F, cols), dtype=torch.complex64, device=local.device)
        assert tuple(res.shape) == (total, F, cols) and res.is_contiguous()
    if comm is not None:
        import ctypes
        st = torch.cuda.current_stream(local.device) if stream is None else stream
        send = local if local.is_contiguous() else local.contiguous()
        with torch.cuda.device(local.device):
            comm.gather(send, counts, F * cols, res, dst, ctypes.c_void_p(st.cuda_stream))
        if not async_op:
            st.synchronize()
            return res
        ev = torch.cuda.Event()
        ev.record(st)
        return res, _StreamWork(ev, send)
    per = max(counts)
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # REHEARSAL path (bench.py --dist-backend gloo): gloo gathers host tensors only, so the maps go through the host,
        # synchronously, on the current stream -- the control flow of a multi-rank run where RCCL cannot be used (ranks
        # sharing one GPU); never the product path
        send_h = torch.zeros((per, F, cols, 2), dtype=torch.float32)
        if counts[me]:
            send_h[:counts[me]] = torch.view_as_real(local).cpu()
        recv_h = [torch.empty((per, F, cols, 2), dtype=torch.float32) for _ in range(shard.world)] if on_dst else None
        dist.gather(send_h, recv_h, dst=dst, group=group)
        if on_dst:
            resr, lo = torch.view_as_real(res), 0
            for r in range(shard.world):
                if counts[r]:
                    resr[lo:lo + counts[r]].copy_(recv_h[r][:counts[r]])
                lo += counts[r]
        return (res, _NoWork()) if async_op else res
    if counts[me] == per and local.is_contiguous():
        send = torch.view_as_real(local)
    else:
        send = torch.zeros((per, F, cols, 2), dtype=torch.float32, device=local.device)
        if counts[me]:
            send[:counts[me]] = torch.view_as_real(local)
    if on_dst:
        resr = torch.view_as_real(res)
        recv, tail, lo = [], [], 0
        for r in range(shard.world):
            if counts[r] == per:
                recv.append(resr[lo:lo + per])                  # lands in place
            else:                                               # shorter / empty blocks: staged
                tmp = torch.empty((per, F, cols, 2), dtype=torch.float32, device=local.device)
                recv.append(tmp)
                tail.append((lo, counts[r], tmp))
            lo += counts[r]
        work = dist.gather(send, recv, dst=dst, group=group, async_op=async_op)

        def finish():
            for lo_, m, tmp in tail:
                if m:
                    resr[lo_:lo_ + m].copy_(tmp[:m])
        if not async_op:
            finish()
            return res
        return res, _TorchWork(work, finish)
    work = dist.gather(send, None, dst=dst, group=group, async_op=async_op)
    return (None, work) if async_op else None


def part_bounds(m, nparts):
    """frame offsets of ``nparts`` near-equal consecutive parts of a block of m frames (the same rule on every rank)"""
    return [(m * p) // nparts for p in range(nparts + 1)]


class PartGather:
    """A frame-sharded array gathered to the root in ``nparts`` rounds, so that the transfer of the frames that are
    finished rides under the compute of the ones that are not (a 600 s stream at 8 ranks: one gather of 158 MB per rank
    after the compute is a quarter of the step; four gathers hide three of them).

    Round p moves frames [b_p, b_{p+1}) of EVERY rank's block (b = part_bounds of that rank's block) with one
    gather_frames call -- rank blocks concatenated in a staging buffer on the root -- and the root copies each block to
    its place in the frame-ordered result (device copies on the gather's stream / after its work handle)."""

    def __init__(self, shard, nparts, group=None, dst=0, comm=None, stream=None):
        self.shard, self.nparts, self.group, self.dst, self.comm, self.stream = shard, int(nparts), group, dst, comm, stream
        self.sizes = shard_sizes(shard)
        self.bounds = [part_bounds(m, self.nparts) for m in self.sizes]
        self.first = [sum(self.sizes[:r]) for r in range(shard.world)]
        self._stage = {}

    def part_range(self, p, rank=None):
        """(lo, hi) inside the block of ``rank`` (default: this shard's rank) that round p moves"""
        b = self.bounds[self.shard.rank if rank is None else rank]
        return b[p], b[p + 1]

    def gather_part(self, p, local_part, result=None, async_op=False):
        """local_part: this rank's frames part_range(p) ([m][F][R+1]); result: the root's [nchunks][F][R+1] array the
        parts are assembled into (None elsewhere).  Returns a work handle (async_op) or None."""
        import torch
        counts = [self.bounds[r][p + 1] - self.bounds[r][p] for r in range(self.shard.world)]
        F, cols = local_part.shape[1], local_part.shape[2]
        stage = None
        if result is not None:
            # one staging block per round (rounds of one pass may be in flight together; the same round of the next
            # pass reuses its block: the caller waits for a round's previous work before issuing it again)
            shape = (sum(counts), F, cols)
            stage = self._stage.get(p)
            if stage is None or tuple(stage.shape) != shape or stage.device != result.device:
                stage = self._stage[p] = torch.empty(shape, dtype=torch.complex64, device=result.device)
        got = gather_frames(local_part, self.shard, self.group, self.dst, async_op, stage, self.comm, self.stream, counts)
        res, work = got if async_op else (got, None)

        def place():
            if res is None:
                return
            off = 0
            for r in range(self.shard.world):
                if counts[r]:
                    lo = self.first[r] + self.bounds[r][p]
                    result[lo:lo + counts[r]].copy_(res[off:off + counts[r]], non_blocking=True)
                off += counts[r]
        if not async_op:
            place()
            return None
        return _PlacedWork(work, place, self.comm, self.stream, result)


class _PlacedWork:
    """work handle of one PartGather round: the root's placement copies follow the gather (on its stream for the C-ABI
    path, after the handle for torch's)"""

    def __init__(self, work, place, comm, stream, result):
        self.work, self.place, self.done = work, place, False
        if comm is not None and result is not None:
            import torch
            st = torch.cuda.current_stream(result.device) if stream is None else stream
            with torch.cuda.stream(st):                 # ordered after prc_gather_frames on the same stream
                place()
            self.done = True
            self.ev = torch.cuda.Event()
            self.ev.record(st)

    def wait(self):
        if self.work is not None:
            self.work.wait()
        if not self.done:
            self.place()
            self.done = True
        elif hasattr(self, "ev"):
            self.ev.synchronize()


class _TorchWork:
    def __init__(self, work, finish):
        self.work, self.finish = work, finish

    def wait(self):
        self.work.wait()
        self.finish()


class HipBackend:
    """The compute of one shard on one MI355X: batched LS chunks + batched overlapped CAF frames.

    ``run`` software-pipelines the two stages over sub-batches on two HIP streams: while the LS
    chain of sub-batch k+1 sits in its latency-bound solves (one Levinson-Durbin per block, a few
    wavefronts busy), the VALU-bound CAF kernel of sub-batch k fills the chip.
    """

    LS_HALO = 2        # spare blocks in every LS plan (the two halo chunks of a sharded stream)

    def __init__(self, cpi_samples, num_range_cells, num_doppler_cells, IF_sample_rate,
                 doppler_bins=(0, 1, -1, 2, -2), window=("kaiser", 5.0), clutter="ls",
                 batch=16, device=None, caf_method=0, doppler_method=0, nlms_mu=0.02, overlap=True,
                 ls_method=0, nsub=1, ls_streams=3, nref=1, ls_reg=1.0, caf_multi="auto", caf_lanes=1):
        """clutter: "ls" = LS_Filter_Multiple over ``doppler_bins`` (main.py:169-176, the reference's choice),
        "ls_direct" = LS_Filter (clutter_removal.py:6-56: circular data matrix, ``ls_reg`` on the Gram diagonal, one
        bin -- SURVEY 8's config-2 "LS_Filter variant"), "nlms" = NLMS_filter with step ``nlms_mu``, None = no canceller."""
        import torch
        from . import engine
        from .range_doppler_processing import _named_window
        self.torch = torch
        self.engine = engine
        self.cpi = int(cpi_samples)
        self.C = self.cpi // 2
        self.R, self.F = int(num_range_cells), int(num_doppler_cells)
        self.fs = float(IF_sample_rate)
        self.bins = tuple(float(b) for b in doppler_bins)
        if clutter not in ("ls", "ls_direct", "nlms", None):
            raise ValueError(f"HipBackend: unknown clutter canceller {clutter!r}")
        self.clutter = clutter
        self.ls_like = clutter in ("ls", "ls_direct")       # block least-squares cancellers: plans, sub-batches, chains
        self.ls_reg = float(ls_reg)
        self.batch = int(batch)
        self.nref = max(1, int(nref))      # reference channels per surveillance channel (run_multi / frames_multi)
        self.nlms_mu = float(nlms_mu)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        # LS launches of batch/nsub chunks.  Measured on MI355X (config 2, two LS chains in flight): 256-chunk launches
        # (nsub = 1) 20.45 k frames/s, 128-chunk launches (nsub = 2) 19.56 k -- the latency-bound Durbin / solve kernels
        # are per launch, so fewer, fuller launches win now that the second chain provides the overlap
        self.overlap = bool(overlap) and self.ls_like and self.batch >= 128
        # chunks per LS launch; NLMS is one wavefront per chunk, so splitting a batch would only idle SIMDs
        self.sub = -(-self.batch // max(int(nsub), 1)) if (self.overlap and self.ls_like) else self.batch
        with torch.cuda.device(self.device):
            self.caf = engine.CafPlan(self.cpi, self.R, self.F, self.batch * self.nref, caf_method, doppler_method,
                                      multi=caf_multi)
            # The LS chain of a sub-batch is a strictly sequential string of kernels, a third of them latency-bound
            # (one Levinson-Durbin per block, per-bin solves: a few wavefronts busy).  Several plans on as many streams
            # run alternate sub-batches concurrently, so one chain's solves sit under the others' HBM-bound passes
            # (measured at config 2: 1 chain 18.7 k frames/s, 2: 20.25 k, 3: 20.46 k, 4: 20.51 k).
            self.nls = max(1, int(ls_streams)) if (self.overlap and self.ls_like) else 1
            # LS_HALO spare blocks per plan: a shard of m frames filters m + 2 chunks (one halo chunk each side,
            # plan_shard); when the last sub-batch would be no more than the halo it rides in the launch before it
            # instead of paying a latency-bound chain of its own (config 4 at 8 ranks: 150 frames, 152 chunks)
            self.ls_plans = [engine.LsPlan(self.C, self.R, 10, clutter == "ls_direct", self.sub + self.LS_HALO, ls_method)
                             for _ in range(self.nls)] if self.ls_like else []
            self.ls = self.ls_plans[0] if self.ls_plans else None
            if isinstance(window, (tuple, str)):
                w = _named_window(window, self.cpi)
            else:
                w = None if window is None else np.ascontiguousarray(window, dtype=np.float32)
            self.window = None if w is None else torch.from_numpy(w).to(self.device)
            self.s_ls_all = [torch.cuda.Stream(device=self.device) for _ in range(self.nls)] if self.overlap else []
            self.s_ls = self.s_ls_all[0] if self.s_ls_all else None
            self.s_caf = torch.cuda.Stream(device=self.device) if self.overlap else None
        self._clean_buf = None
        self._fe_plans = {}
        # frames_multi in two lanes (config 5): alternate pieces of batch/2 frames go to two plans on two streams, so the
        # Doppler kernel of one piece (issue-stalled on its LDS / memory instructions) runs under the segment kernel of the
        # next (latency-bound) and no launch ramp or tail is exposed between the stages
        self.caf_lanes = 2 if (int(caf_lanes) >= 2 and self.batch >= 2) else 1
        self._lane_plans, self._lane_streams = [], []
        if self.caf_lanes == 2:
            with torch.cuda.device(self.device):
                per = -(-self.batch // 2)
                self._lane_plans = [engine.CafPlan(self.cpi, self.R, self.F, per * self.nref, caf_method, doppler_method,
                                                   multi=caf_multi) for _ in range(2)]
                self._lane_streams = [torch.cuda.Stream(device=self.device) for _ in range(2)]

    def _stream(self):
        from . import _lib
        return _lib.torch_stream_ptr(self.device)

    def padded(self, chunks):
        """[C/2 zeros | chunks | C/2 zeros] complex64 on the device (chunks: 1-D host or device)."""
        torch = self.torch
        t = chunks if torch.is_tensor(chunks) else torch.from_numpy(np.ascontiguousarray(chunks, dtype=np.complex64))
        n = t.shape[0]
        buf = torch.zeros(n + self.C, dtype=torch.complex64, device=self.device)
        buf[self.C // 2:self.C // 2 + n].copy_(t, non_blocking=True)
        return buf

    def front_end(self, raw, input_chunk_length, offset_freq, input_sample_rate, up, dn, max_blocks=32, block0=0, out=None):
        """SURVEY 8f next #1 -- main.py:105-166 for one channel, on the device: raw interleaved scalars
        (host array or device tensor; int8 / uint8 / int16 / float32) -> IF complex64 stream (device
        tensor, blocks concatenated).  One fused kernel per batch of blocks.  block0: index of the first block within
        its recording (the block phases of main.py:125-131 continue across calls on consecutive pieces of one recording);
        out: optional device tensor of nblocks * output_chunk_length complex64 to write into."""
        return self._front_end((raw,), input_chunk_length, offset_freq, input_sample_rate, up, dn, max_blocks, block0,
                               (out,))[0]

    def front_end2(self, raw_ref, raw_srv, input_chunk_length, offset_freq, input_sample_rate, up, dn, max_blocks=32,
                   block0=0, out_ref=None, out_srv=None):
        """main.py:133-149: BOTH recordings through the front end in one launch per batch of blocks -- they are tuned
        with the same frequency and the same block phases, so one rotation factor per input sample serves the two
        (prc_frontend_execute2).  Same arguments as front_end; returns (ref IF stream, srv IF stream), bit-identical to
        two front_end calls."""
        return self._front_end((raw_ref, raw_srv), input_chunk_length, offset_freq, input_sample_rate, up, dn, max_blocks,
                               block0, (out_ref, out_srv))

    def _front_end(self, raws, input_chunk_length, offset_freq, input_sample_rate, up, dn, max_blocks, block0, outs):
        torch = self.torch
        from . import _lib
        ts = []
        for raw in raws:
            t = raw if torch.is_tensor(raw) else torch.from_numpy(np.ascontiguousarray(raw))
            if str(t.dtype).replace("torch.", "") not in _lib.RAW_DTYPES:
                t = t.to(torch.float32)
            ts.append(t.to(self.device, non_blocking=True).contiguous())
        if len(ts) == 2 and (ts[0].dtype != ts[1].dtype or ts[0].shape != ts[1].shape):
            raise ValueError("front_end2: the two recordings must have the same sample type and length")
        t = ts[0]
        icl = int(input_chunk_length)
        nblocks = int(t.shape[0]) // icl
        n_in = icl // 2
        key = (n_in, str(t.dtype), int(up), int(dn), int(max_blocks))
        plan = self._fe_plans.get(key)
        if plan is None:
            with torch.cuda.device(self.device):
                plan = self._fe_plans[key] = self.engine.FrontendPlan(n_in, str(t.dtype).replace("torch.", ""),
                                                                      up, dn, max_blocks)
        per_block = n_in % (input_sample_rate // offset_freq)                       # main.py:125-130
        phases = 2 * np.pi * (np.arange(nblocks) + int(block0)) * per_block * (offset_freq / input_sample_rate)
        outs = list(outs)
        for i, out in enumerate(outs):
            if out is None:
                outs[i] = torch.empty(nblocks * plan.n_out, dtype=torch.complex64, device=self.device)
                continue
            # the kernel writes nblocks * n_out complex64 straight into the caller's tensor: anything else than a
            # contiguous complex64 tensor of at least that size on this device would be a silent out-of-bounds write
            if not (torch.is_tensor(out) and out.is_cuda and out.device == self.device):
                raise ValueError(f"front_end: out must be a tensor on {self.device}")
            if out.dtype != torch.complex64 or out.dim() != 1 or not out.is_contiguous():
                raise ValueError("front_end: out must be a contiguous one-dimensional complex64 tensor")
            if out.numel() < nblocks * plan.n_out:
                raise ValueError(f"front_end: out holds {out.numel()} samples, {nblocks} blocks of {plan.n_out} need "
                                 f"{nblocks * plan.n_out}")
        with torch.cuda.device(self.device):
            for b0 in range(0, nblocks, max_blocks):
                nb = min(max_blocks, nblocks - b0)
                if len(ts) == 2:
                    plan.execute2(ts[0][b0 * icl:], ts[1][b0 * icl:], outs[0][b0 * plan.n_out:], outs[1][b0 * plan.n_out:],
                                  nb, icl, plan.n_out, offset_freq, input_sample_rate, phases[b0:b0 + nb], True, self._stream())
                else:
                    plan.execute(t[b0 * icl:], outs[0][b0 * plan.n_out:], nb, icl, plan.n_out, offset_freq,
                                 input_sample_rate, phases[b0:b0 + nb], True, self._stream())
        return tuple(outs)

    def _clean_target(self, srv_pad):
        # the cleaned stream buffer is reused across calls; only its two C/2 pads must be zero and
        # the kernels never write them
        out = self._clean_buf
        if out is None or out.shape != srv_pad.shape or out.device != srv_pad.device:
            out = self._clean_buf = self.torch.zeros_like(srv_pad)
        return out

    def _clean_range(self, ref_pad, srv_pad, out, c0, nb, stream, plan=None):
        C, off = self.C, self.C // 2 + c0 * self.C
        if self.clutter == "ls":
            (plan or self.ls).execute(ref_pad[off:], srv_pad[off:], out[off:], nb, C, C, self.fs, self.bins, 0.0,
                                      None, stream)
        elif self.clutter == "ls_direct":
            (plan or self.ls).execute(ref_pad[off:], srv_pad[off:], out[off:], nb, C, C, self.fs, (0.0,), self.ls_reg,
                                      None, stream)
        else:
            self.engine.nlms_execute(ref_pad[off:], srv_pad[off:], out[off:], C, self.R, self.nlms_mu, 10,
                                     None, None, nb, C, C, stream)

    def clean(self, ref_pad, srv_pad, nlocal):
        """LS_Filter_Multiple / NLMS per chunk, chunk c of the padded stream -> same place in the
        returned padded cleaned stream (main.py:169-176)."""
        if self.clutter is None:
            return srv_pad
        out = self._clean_target(srv_pad)
        # NLMS is one wavefront per hop chunk and needs no plan workspace: every local chunk goes into ONE launch
        # (a 256-stream launch would leave three SIMDs in four idle); LS launches are bounded by the plan's workspace
        with self.torch.cuda.device(self.device):
            for c0, c1 in self._ls_ranges(nlocal):
                self._clean_range(ref_pad, srv_pad, out, c0, c1 - c0, self._stream())
        return out

    def _ls_ranges(self, nlocal, sub=None):
        """[c0, c1) chunk ranges of the clutter launches: sub-batches of ``sub`` chunks, a remainder of at most
        LS_HALO chunks folded into the launch before it; NLMS takes every chunk in one launch"""
        if not self.ls_like:
            return [(0, nlocal)] if nlocal > 0 else []
        sub = self.sub if sub is None else sub
        out, c0 = [], 0
        while c0 < nlocal:
            c1 = min(c0 + sub, nlocal)
            if nlocal - c1 <= self.LS_HALO:
                c1 = nlocal
            out.append((c0, c1))
            c0 = c1
        return out

    def frames(self, ref_pad, clean_pad, offsets_first, nframes, out=None, f_lo=0, f_hi=None, stream=None,
               cuts=None, on_frames=None, tstream=None):
        """fast_xambg on overlapped frames [f_lo, f_hi) (stride C; frame 0 at element offsets_first).
        cuts: frame indices at which a CAF launch must end (besides every ``batch`` frames); on_frames(lo, hi, event)
        is called after the launch that completes frames [lo, hi) has been enqueued, with an event recorded behind it
        on the launch stream (``tstream``: the torch stream object when ``stream`` is given as a raw pointer) -- together
        they let a caller start moving finished frames (PartGather) while the rest is still being computed."""
        torch = self.torch
        if out is None:
            out = torch.empty((nframes, self.F, self.R + 1), dtype=torch.complex64, device=self.device)
        f_hi = nframes if f_hi is None else f_hi
        stops = sorted({c for c in (cuts or ()) if f_lo < c < f_hi} | {f_hi})
        with torch.cuda.device(self.device):
            f0 = f_lo
            for stop in stops:
                while f0 < stop:
                    nb = min(self.batch, stop - f0)
                    off = offsets_first + f0 * self.C
                    self.caf.execute(ref_pad[off:], clean_pad[off:], out[f0:], nb, self.C, self.cpi,
                                     self.window, self._stream() if stream is None else stream)
                    f0 += nb
                if on_frames is not None:
                    ev = torch.cuda.Event()
                    ev.record(tstream if tstream is not None else torch.cuda.current_stream(self.device))
                    on_frames(f_lo, stop, ev)
                    f_lo = stop
        return out

    def frames_multi(self, ref_pads, clean_pad, offsets_first, nframes, outs=None, stream=None, join=True):
        """fast_xambg of EVERY reference channel in ``ref_pads`` against the one (cleaned) surveillance stream, frame
        by overlapped frame: a multi-illuminator frame (BASELINE config 5) through prc_caf_execute_multi, which
        transforms the surveillance pieces once per segment for all illuminators.  Returns one
        [nframes][F][R+1] tensor per reference channel.

        With ``caf_lanes=2`` (and no explicit ``stream``) pieces of batch/2 frames alternate between two plans on two
        streams; the lanes start behind whatever the current stream holds, and ``join`` makes the current stream wait for
        them at the end (join=False: the caller synchronises -- a resident pipeline whose next call may then start under
        this call's last Doppler launch)."""
        torch = self.torch
        nref = len(ref_pads)
        if nref > self.nref:
            raise ValueError(f"frames_multi: {nref} reference channels, the backend was built for nref={self.nref}")
        if outs is None:
            outs = [torch.empty((nframes, self.F, self.R + 1), dtype=torch.complex64, device=self.device)
                    for _ in ref_pads]
        if self.caf_lanes == 2 and stream is None and nframes > 1:
            import ctypes
            main = torch.cuda.current_stream(self.device)
            ev = torch.cuda.Event()
            ev.record(main)
            per = -(-self.batch // 2)
            # the lane streams read and write tensors the caching allocator handed out on OTHER streams: tell it, so that a
            # tensor freed by the caller (join=False, outs=None) is not handed out again while a lane still works on it
            # (ADVICE r5).  With join=False the caller must still order its own reads of `outs` behind the lanes
            # (torch.cuda.synchronize, or wait_stream on HipBackend._lane_streams).
            for st in self._lane_streams[:min(2, -(-nframes // per))]:
                for t in list(outs) + list(ref_pads) + [clean_pad]:
                    t.record_stream(st)
            with torch.cuda.device(self.device):
                for i, f0 in enumerate(range(0, nframes, per)):
                    st, plan = self._lane_streams[i & 1], self._lane_plans[i & 1]
                    if i < 2:
                        st.wait_event(ev)
                    nb = min(per, nframes - f0)
                    off = offsets_first + f0 * self.C
                    plan.execute_multi([r[off:] for r in ref_pads], clean_pad[off:], [o[f0:] for o in outs], nb,
                                       self.C, self.cpi, self.window, ctypes.c_void_p(st.cuda_stream))
            if join:
                for st in self._lane_streams:
                    main.wait_stream(st)
            return outs
        with torch.cuda.device(self.device):
            for f0 in range(0, nframes, self.batch):
                nb = min(self.batch, nframes - f0)
                off = offsets_first + f0 * self.C
                self.caf.execute_multi([r[off:] for r in ref_pads], clean_pad[off:], [o[f0:] for o in outs], nb,
                                       self.C, self.cpi, self.window, self._stream() if stream is None else stream)
        return outs

    def run_multi(self, ref_pads, srv_pad, nlocal, offsets_first, nframes, outs=None, clutter_ref=0):
        """clean (against reference channel ``clutter_ref``, when the backend has a canceller) + frames_multi"""
        clean = self.clean(ref_pads[clutter_ref], srv_pad, nlocal)
        return self.frames_multi(ref_pads, clean, offsets_first, nframes, outs)

    def run(self, ref_pad, srv_pad, nlocal, offsets_first, nframes, out=None, cuts=None, on_frames=None):
        """clean + frames for one resident shard.  With ``overlap`` the LS chain runs sub-batch by
        sub-batch on one stream and the CAF of every frame whose three chunks are already clean
        follows on a second stream (frame j needs local chunk j + offsets_first/C + 1).
        out: optional preallocated [>= nframes][F][R+1] complex64 tensor for the maps."""
        # a shard that fits ONE LS launch (config 4 at 8 ranks: 152 chunks) is still cut in two, so that one half's
        # latency-bound solves and the first frames' CAF run under the other half's HBM-bound passes (measured on one
        # GPU, `bench.py --workload cfg4 --shard-of 3/8`: 8.0 -> 7.7 ms per 150-frame shard)
        sub = self.sub
        if self.overlap and self.nls >= 2 and 128 <= nlocal <= self.sub + self.LS_HALO:
            sub = -(-nlocal // 2)
        if not self.overlap or nlocal <= sub + self.LS_HALO:
            clean = self.clean(ref_pad, srv_pad, nlocal)
            return self.frames(ref_pad, clean, offsets_first, nframes, out, cuts=cuts, on_frames=on_frames)
        import ctypes
        torch = self.torch
        clean = self._clean_target(srv_pad)
        if out is None:
            out = torch.empty((nframes, self.F, self.R + 1), dtype=torch.complex64, device=self.device)
        main = torch.cuda.current_stream(self.device)
        for st in self.s_ls_all:
            st.wait_stream(main)
        self.s_caf.wait_stream(main)
        first_chunk = offsets_first // self.C
        done = 0
        with torch.cuda.device(self.device):
            for idx, (c0, c1) in enumerate(self._ls_ranges(nlocal, sub)):
                k = idx % self.nls
                st = self.s_ls_all[k]
                self._clean_range(ref_pad, srv_pad, clean, c0, c1 - c0, ctypes.c_void_p(st.cuda_stream),
                                  self.ls_plans[k])
                ev = torch.cuda.Event()
                ev.record(st)
                # the CAF stream waits for every sub-batch in order, so "chunks below c1 are clean" holds there
                self.s_caf.wait_event(ev)
                ready = nframes if c1 == nlocal else max(min(nframes, c1 - 1 - first_chunk), 0)
                if ready > done:
                    self.frames(ref_pad, clean, offsets_first, nframes, out, done, ready,
                                ctypes.c_void_p(self.s_caf.cuda_stream), cuts=cuts, on_frames=on_frames,
                                tstream=self.s_caf)
                    done = ready
        for st in self.s_ls_all:
            main.wait_stream(st)
        main.wait_stream(self.s_caf)
        return out


class StreamProcessor:
    """main.py:169-194 for an in-memory IF stream; ``backend`` does the per-shard compute."""

    def __init__(self, backend, rank=0, world=1, group=None, comm=None):
        self.backend = backend
        self.rank, self.world, self.group = rank, world, group
        self.comm = comm            # FrameComm: gather through prc_gather_frames instead of torch.distributed

    @classmethod
    def from_config(cls, config, **kw):
        """config: the dict of passiveradar_amd.config.getConfiguration (reference keys)."""
        be = HipBackend(config["cpi_samples"], config["num_range_cells"], config["num_doppler_cells"],
                        config["IF_sample_rate"], **kw)
        return cls(be)

    def process_local(self, ref, srv):
        """ref, srv: the whole stream (host arrays or device tensors).  Returns (frames [m][F][R+1]
        for this rank's shard, shard)."""
        C = self.backend.C
        nchunks = int(ref.shape[0]) // C
        sh = plan_shard(nchunks, self.rank, self.world)
        if sh.nframes == 0:         # more ranks than frames: an empty block still takes part in the gather
            import torch
            return torch.zeros((0, self.backend.F, self.backend.R + 1), dtype=torch.complex64,
                               device=getattr(self.backend, "device", "cpu")), sh
        lo, hi = sh.chunk_lo * C, sh.chunk_hi * C
        ref_pad = self.backend.padded(ref[lo:hi])
        srv_pad = self.backend.padded(srv[lo:hi])
        if hasattr(self.backend, "run"):
            frames = self.backend.run(ref_pad, srv_pad, sh.nlocal_chunks, sh.frame_offset(sh.frame_lo, C), sh.nframes)
        else:
            clean_pad = self.backend.clean(ref_pad, srv_pad, sh.nlocal_chunks)
            frames = self.backend.frames(ref_pad, clean_pad, sh.frame_offset(sh.frame_lo, C), sh.nframes)
        return frames, sh

    def process(self, ref, srv, gather=True):
        """Returns [nframes][F][R+1] (rank 0 when sharded; None on other ranks if gather)."""
        frames, sh = self.process_local(ref, srv)
        if self.world == 1 or not gather:
            return frames
        return gather_frames(frames, sh, self.group, comm=self.comm)

    def process_raw(self, raw_ref, raw_srv, config, gather=True):
        """main.py:105-194 from the raw interleaved recordings: front end per channel (device), then
        the LS + CAF pipeline.  ``config`` is the dict of passiveradar_amd.config.getConfiguration."""
        args = (config["input_chunk_length"], config["offset_freq"], config["input_sample_rate"],
                config["resamp_up"], config["resamp_dn"])
        shape = lambda x: tuple(x.shape) if hasattr(x, "shape") else (len(x),)      # host arrays, lists or device tensors
        same_type = str(getattr(raw_ref, "dtype", "")) == str(getattr(raw_srv, "dtype", ""))
        if hasattr(self.backend, "front_end2") and shape(raw_ref) == shape(raw_srv) and same_type:
            ref, srv = self.backend.front_end2(raw_ref, raw_srv, *args)       # one launch per batch of blocks for both
        else:
            ref = self.backend.front_end(raw_ref, *args)
            srv = self.backend.front_end(raw_srv, *args)
        return self.process(ref, srv, gather=gather)

    @staticmethod
    def to_reference_layout(frames):
        """[nframes][F][R+1] -> the reference's saved array (F, R+1, nframes) (main.py:200-224)."""
        if hasattr(frames, "permute"):
            return frames.permute(1, 2, 0)
        return np.moveaxis(np.asarray(frames), 0, 2)
