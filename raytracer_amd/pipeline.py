"""Frames in flight: a sequence of frames rendered with several frames of every rank under way at once.

Why: a rank's share of one frame is bound by the *latency* of its longest pixels, not by throughput -- a pixel's
samples are serial through its generator (reference src/main.cpp:304-318), so the share of an eighth of the 4K frame
takes 3.3 ms although its arithmetic is worth 0.75 ms.  A sequence of frames hides that latency
the way the reference hides the latency of its tiles: producers and one consumer around a queue (its worker threads
push tiles to a Redis list while the main thread pops and recomposes them, src/main.cpp:268-331, 375-395,
src/redis.h:27-48).  Here the producers are `in_flight` host threads, each with a Renderer context (own HIP streams
and device buffers) and its own band block; the consumer is the calling thread, which issues the gathers in frame
order -- every rank issues the same collectives in the same order -- and reassembles the frame on rank 0.  There
are more band blocks than producers (two per producer by default), so a producer whose frame finished early starts its
next frame although the consumer, who takes frames in order, has not come to its block yet.

Frames are independent (each pixel of each frame re-seeds from (x, y, seed_base)), so the bytes of every frame are
those of an unpipelined render.
"""
import os
import threading
import time

import torch

from . import bands


class FramePipeline:
    """render(worker, block, k) must fill `block` with this rank's packed bands of frame k, using the resources of
    producer `worker` (0 <= worker < workers: its Renderer context), and return when they are in memory
    (Renderer.render_rows does).  blocks: [rows_per_rank, W, C] tensors, same shape on every rank, at least `workers`
    of them.  Not re-entrant: one run() at a time."""

    def __init__(self, render, blocks, height, band_rows, renderer=None, group=None, workers=None, comm=None):
        if not blocks:
            raise ValueError("FramePipeline needs at least one band block")
        self.render = render
        self.blocks = list(blocks)
        self.workers = len(self.blocks) if workers is None else int(workers)
        if not 1 <= self.workers <= len(self.blocks):
            raise ValueError("FramePipeline: 1 <= workers <= number of blocks")
        self.height = height
        self.band_rows = band_rows
        self.renderer = renderer
        self.group = group
        # comm = (ncclComm_t handle from Renderer.comm_create, rank, nranks): the gather goes through the library's
        # rt_gather_rows on `renderer`'s stream (RCCL, no torch.distributed on the data path); None: torch.distributed
        # (gloo in the CPU tests) or a single rank
        self.comm = comm
        self._gathered = None
        self._frame = None
        # per frame of the last run(): time of the gather + reassembly step -- device time from HIP events when it went
        # through rt_gather_rows, wall time of the torch.distributed gather + rt_deinterleave_bands otherwise
        self.gather_ms = []
        self.pace_ms = float(os.environ.get("RT_PACE_MS", "0")) or None

    def _buffers(self):
        """rank 0's gather and frame buffers, allocated once (no allocation inside the frame loop)"""
        import torch.distributed as dist
        on = dist.is_initialized()
        nranks = dist.get_world_size(self.group) if on else 1
        rank = dist.get_rank(self.group) if on else 0
        if self.comm is not None:
            _, rank, nranks = self.comm
        b = self.blocks[0]
        if rank == 0 and self._frame is None:
            if nranks > 1:
                self._gathered = torch.empty((nranks,) + tuple(b.shape), dtype=b.dtype, device=b.device)
            self._frame = torch.empty((self.height,) + tuple(b.shape[1:]), dtype=b.dtype, device=b.device)
        return nranks, rank

    def run(self, n_frames, on_frame=None, first_frame=0):
        """Renders frames first_frame .. first_frame + n_frames - 1.  On rank 0, on_frame(k, frame) is called in frame
        order; `frame` ([H, W, C]) is reused for the next frame, so the callback copies what it wants to keep.
        Returns the last frame on rank 0 (None elsewhere)."""
        nranks, rank = self._buffers()
        B = len(self.blocks)
        # per block: free = the consumer is done with the frame that was in it, ready = the frame in it is rendered
        free = [threading.Semaphore(1) for _ in range(B)]
        ready = [threading.Semaphore(0) for _ in range(B)]
        failed = []
        stop = threading.Event()
        take = threading.Lock()
        next_frame = [0]
        last_start = [0.0]

        def producer(worker):
            # frames are taken in order; frame k lives in block k % B, which is free once frame k - B was consumed
            try:
                while True:
                    with take:
                        k = next_frame[0]
                        next_frame[0] = k + 1
                    if k >= n_frames:
                        return
                    slot = k % B
                    free[slot].acquire()
                    if stop.is_set():
                        return
                    if self.pace_ms:
                        with take:
                            t = max(time.perf_counter(), last_start[0] + self.pace_ms * 1e-3)
                            last_start[0] = t
                        while time.perf_counter() < t:
                            time.sleep(max(0.0, min(t - time.perf_counter(), 0.0005)))
                    self.render(worker, self.blocks[slot], first_frame + k)
                    ready[slot].release()
            except BaseException as e:          # handed to the consumer, which raises it
                failed.append(e)
                for s in ready:
                    s.release()

        threads = [threading.Thread(target=producer, args=(w,), name="rt-frame-%d" % w, daemon=True)
                   for w in range(min(self.workers, max(n_frames, 1)))]
        for t in threads:
            t.start()
        last = None
        self.gather_ms = []
        try:
            for k in range(n_frames):
                slot = k % B
                ready[slot].acquire()
                if failed:
                    raise failed[0]
                block = self.blocks[slot]
                t_gather = time.perf_counter()
                if self.comm is not None and nranks > 1:
                    W = block.shape[1]
                    bpp = block.element_size() * (block.shape[2] if block.dim() > 2 else 1)
                    self.renderer.gather_rows(self.comm[0], rank, nranks, block, self._gathered, self._frame, W, self.height, bpp,
                                              self.band_rows, block.shape[0])
                    last = self._frame
                elif nranks == 1:
                    last = bands.deinterleave(block.unsqueeze(0), self._frame, self.band_rows, 1, self.renderer)
                else:
                    import torch.distributed as dist
                    dist.gather(block, list(self._gathered.unbind(0)) if rank == 0 else None, dst=0, group=self.group)
                    if rank == 0:
                        last = bands.deinterleave(self._gathered, self._frame, self.band_rows, nranks, self.renderer)
                    elif block.is_cuda:
                        # the block goes back to the producers: the send must have read it
                        torch.cuda.current_stream(block.device).synchronize()
                if self.comm is not None and nranks > 1:
                    self.gather_ms.append(self.renderer.last_gather_ms)
                else:
                    self.gather_ms.append((time.perf_counter() - t_gather) * 1e3)
                free[slot].release()
                if rank == 0 and on_frame is not None:
                    on_frame(first_frame + k, last)
        finally:
            stop.set()
            for s in free:
                s.release()
            for t in threads:
                t.join()
        if failed:
            raise failed[0]
        return last
