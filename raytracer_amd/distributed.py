"""Image partition across ranks and the framebuffer gather (one process per GPU, torch.distributed).

The image is cut into bands of 8 pixel rows (one row of 8x8 wavefront tiles).  Band b belongs to rank
b % world; the scene is replicated; nothing is exchanged while rendering.  After rendering, each rank holds
its bands inside a full-size frame buffer; `gather_bands` moves them to rank 0 with one collective
(`torch.distributed.gather`: RCCL on GPUs, gloo on CPU) and interleaves them back into rank 0's frame.
"""
import queue
import threading

import torch

BAND_ROWS = 8


def num_bands(height):
    return (height + BAND_ROWS - 1) // BAND_ROWS


def padded_bands(height, world):
    """Bands rounded up to a multiple of world so that every rank contributes an equal-size block."""
    b = num_bands(height)
    return (b + world - 1) // world * world


def bands_of_rank(height, rank, world):
    return list(range(rank, num_bands(height), world))


def rows_of_rank(height, rank, world):
    rows = []
    for b in bands_of_rank(height, rank, world):
        rows.extend(range(b * BAND_ROWS, min((b + 1) * BAND_ROWS, height)))
    return rows


def alloc_frame(height, width, world, device):
    """Frame buffer with the row count padded to whole bands for every rank (the renderer only writes rows < height)."""
    return torch.zeros((padded_bands(height, world) * BAND_ROWS, width, 3), dtype=torch.float32, device=device)


class BandGather:
    """Pre-allocated gather of the bands to rank 0.  Without a process group (dist=None) the frame is already whole."""

    def __init__(self, height, width, rank, world, device, dist=None):
        self.height, self.width, self.rank, self.world, self.dist = height, width, rank, world, dist
        self.bands_pad = padded_bands(height, world)
        self.row_floats = BAND_ROWS * width * 3
        self.per_rank = self.bands_pad // world
        self.recv = None
        self.parts = None
        if dist is not None and rank == 0:
            # one receive buffer [rank][band of that rank][band bytes]; the gather list is its per-rank views
            self.recv = torch.empty((world, self.per_rank, self.row_floats), dtype=torch.float32, device=device)
            self.parts = list(self.recv.unbind(0))

        self.mine = None
        if dist is not None:
            self.mine = torch.empty((self.per_rank, self.row_floats), dtype=torch.float32, device=device)
        self._work = None
        self._frame = None

    def start(self, frame):
        """Begins the gather of `frame` (alloc_frame(...) tensor holding this rank's bands) without waiting for it: the
        collective runs on the backend's own stream, so the caller may render the next frame -- into ANOTHER buffer, with
        ANOTHER BandGather -- meanwhile.  finish() completes it."""
        if self.dist is None:
            return
        assert self._work is None, "finish() the previous gather first"
        # band b = k * world + r  ->  view [k][r][band bytes]
        fv = frame.view(self.per_rank, self.world, self.row_floats)
        self.mine.copy_(fv[:, self.rank])
        self._work = self.dist.gather(self.mine, self.parts, dst=0, async_op=True)
        self._frame = frame

    def finish(self):
        """Waits for the gather begun by start(); on return rank 0's frame is complete (on the current stream)."""
        if self._work is None:
            return
        self._work.wait()
        if self.rank == 0:
            fv = self._frame.view(self.per_rank, self.world, self.row_floats)
            fv.copy_(self.recv.permute(1, 0, 2))  # one interleaving copy (rank 0's own bands are rewritten with themselves)
        self._work = None
        self._frame = None

    def __call__(self, frame):
        """frame: alloc_frame(...) tensor holding this rank's bands.  On return rank 0's frame is complete."""
        self.start(frame)
        self.finish()
        return frame


class GatherThread:
    """Runs the gathers of a ring of (frame buffer, BandGather) pairs on a worker thread with its own stream, so that
    neither the host side of the collective (tens of microseconds per call) nor its copies sit between two frames of
    the rendering thread.  submit(k) hands over buffer k, whose frame must be complete (the renderer's call has
    returned); wait(k) returns once that gather is complete and the buffer may be overwritten.  All collectives of the
    process group are issued by this one thread, in submission order -- the same order on every rank."""

    def __init__(self, frames, gathers, device="cuda"):
        self.frames, self.gathers = frames, gathers
        self.done = [threading.Event() for _ in frames]
        for e in self.done:
            e.set()
        self.error = None
        self.jobs = queue.Queue()
        self.stream = torch.cuda.Stream(device=device) if torch.cuda.is_available() and str(device).startswith("cuda") else None
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def _run(self):
        if self.stream is not None:
            torch.cuda.set_device(self.stream.device)  # a new thread starts on device 0
        while True:
            k = self.jobs.get()
            if k is None:
                return
            try:
                if self.stream is not None:
                    with torch.cuda.stream(self.stream):
                        self.gathers[k](self.frames[k])
                    self.stream.synchronize()
                else:
                    self.gathers[k](self.frames[k])
            except BaseException as e:  # reported by the next wait()
                self.error = e
            self.done[k].set()

    def submit(self, k):
        self.done[k].clear()
        self.jobs.put(k)

    def wait(self, k=None):
        for e in (self.done if k is None else [self.done[k]]):
            e.wait()
        if self.error is not None:
            raise self.error

    def close(self):
        self.wait()
        self.jobs.put(None)
        self.thread.join()
