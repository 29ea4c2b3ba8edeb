"""Multi-rank path on CPU (gloo, world_size 2 and 3): the band partition and the gather to rank 0 used by
bench.py.  The per-rank renderer is replaced by the oracle (the GPU renderer needs a GPU; the partition-
independence of the GPU renderer itself is covered by test_gpu_parity.py::test_render_is_partition_independent)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import CAMERAS, GOLDEN, ROOT

from raytracer_amd import distributed as rd

W, H = 40, 37  # not multiples of 8 on purpose


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = orc.Scene.from_npz(os.path.join(GOLDEN, "scenes", "cornell_original.npz"))
        tree = orc.Octree(sc, 16, 10)
        frame = rd.alloc_frame(H, W, world, "cpu")
        kw = dict(fov=0.6, width=W, height=H, spt=2, render_type="blinn_phong", shading="flat", aspect_mode=orc.ASPECT_TRUE,
                  **CAMERAS["original"])
        for b in rd.bands_of_rank(H, rank, world):
            r0, r1 = b * rd.BAND_ROWS, min((b + 1) * rd.BAND_ROWS, H)
            img, _, _, _ = orc.render(sc, tree, orc.make_cfg(rows=(r0, r1), **kw), want_hits=False)
            frame[r0:r1] = torch.from_numpy(img[r0:r1])
        rd.BandGather(H, W, rank, world, "cpu", dist)(frame)
        if rank == 0:
            full, _, _, _ = orc.render(sc, tree, orc.make_cfg(**kw), want_hits=False)
            np.save(out_path, np.stack([frame[:H].numpy(), full]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_band_partition_and_gather(world, tmp_path):
    out = str(tmp_path / "frames.npy")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got, ref = np.load(out)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert ref.max() > 0.05


def _pipeline_worker(rank, world, port, out_path):
    """The frame loop of bench.py: two frame buffers, the gather of frame k still in flight while frame k + 1 is
    produced.  Frame k is filled with a value that identifies (k, row), so a late or mixed-up gather shows."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        frames = [rd.alloc_frame(H, W, world, "cpu") for _ in range(2)]
        gathers = [rd.BandGather(H, W, rank, world, "cpu", dist) for _ in range(2)]
        rows = rd.rows_of_rank(H, rank, world)
        seen = []
        n_frames = 5
        for k in range(n_frames + 2):
            b = k & 1
            gathers[b].finish()
            if rank == 0 and k >= 2:
                seen.append(frames[b][:H].clone().numpy())  # frame k - 2 has arrived
            if k < n_frames:
                frames[b].zero_()
                for r in rows:
                    frames[b][r] = float(1000 * k + r)
                gathers[b].start(frames[b])
        if rank == 0:
            np.save(out_path, np.stack(seen))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_pipelined_gather(world, tmp_path):
    out = str(tmp_path / "pipe.npy")
    mp.spawn(_pipeline_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out)
    assert got.shape[0] == 5
    for k in range(5):
        want = (1000 * k + np.arange(H, dtype=np.float32))[:, None, None] * np.ones((1, W, 3), np.float32)
        assert np.array_equal(got[k], want), k


def _thread_worker(rank, world, port, out_path):
    """The same frame loop with the gathers on rd.GatherThread, as bench.py runs it by default."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        frames = [rd.alloc_frame(H, W, world, "cpu") for _ in range(2)]
        gathers = [rd.BandGather(H, W, rank, world, "cpu", dist) for _ in range(2)]
        worker = rd.GatherThread(frames, gathers, device="cpu")
        rows = rd.rows_of_rank(H, rank, world)
        seen = []
        n_frames = 6
        for k in range(n_frames + 2):
            b = k & 1
            worker.wait(b)
            if rank == 0 and k >= 2:
                seen.append(frames[b][:H].clone().numpy())  # frame k - 2 has arrived
            if k < n_frames:
                frames[b].zero_()
                for r in rows:
                    frames[b][r] = float(1000 * k + r)
                worker.submit(b)
        worker.close()
        if rank == 0:
            np.save(out_path, np.stack(seen))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_thread(world, tmp_path):
    out = str(tmp_path / "thread.npy")
    mp.spawn(_thread_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out)
    assert got.shape[0] == 6
    for k in range(6):
        want = (1000 * k + np.arange(H, dtype=np.float32))[:, None, None] * np.ones((1, W, 3), np.float32)
        assert np.array_equal(got[k], want), k


class _StandInRenderer:
    """What bench.FrameLoop needs of the renderer: frame number f, row r -> every float of the row is 1000 f + r, written
    into this rank's rows when the frame is waited for (a frame in flight is not complete before)."""

    def __init__(self, rank, world):
        self.rows = rd.rows_of_rank(H, rank, world)
        self.loop = None
        self.submitted = 0
        self.pending = {}
        self.order = []  # (frame number, buffer index) in submission order

    def _buffer(self, ptr):
        return next(i for i, f in enumerate(self.loop.frames) if f.data_ptr() == ptr)

    def _fill(self, k, number):
        for r in self.rows:
            self.loop.frames[k][r] = float(1000 * number + r)

    def render(self, cfg, out_ptr=None, stream=None):
        k = self._buffer(out_ptr)
        self.order.append((self.submitted, k))
        self._fill(k, self.submitted)
        self.submitted += 1
        return None, None

    def render_submit(self, cfg, out_ptr, stream=None):
        k = self._buffer(out_ptr)
        ticket = min(set(range(8)) - set(self.pending))
        self.pending[ticket] = (self.submitted, k)
        self.order.append((self.submitted, k))
        self.loop.frames[k].fill_(-1.0)  # the buffer of a frame in flight holds nothing usable
        self.submitted += 1
        return ticket

    def render_wait(self, ticket):
        number, k = self.pending.pop(ticket)
        self._fill(k, number)
        return None


def _frame_loop_worker(rank, world, port, out_path, depth):
    """bench.FrameLoop itself (ring of buffers, frames in flight, gathers on the worker thread) over gloo."""
    import bench
    import raytracer_amd as rt
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        wl = dict(bench.WORKLOADS["cornell_original_blinn_spt4_256"], width=W, height=H)
        bench.PREWARM_S = 0.0  # (the frames are counted below: no warm-up by the clock)
        ads = _StandInRenderer(rank, world)
        loop = bench.FrameLoop(ads, rt, wl, rank, world, dist, torch, depth=depth, device="cpu")
        ads.loop = loop
        assert len(loop.frames) == (2 if depth == 0 else depth + 2)
        loop.warm(1)
        loop.timed(7)
        assert not loop.in_flight and not ads.pending
        if rank == 0:
            # after the fence every buffer of the ring holds the complete frame that was rendered into it last
            last = {}
            for number, k in ads.order:
                last[k] = number
            np.save(out_path, np.stack([np.concatenate([[float(last[k])], loop.frames[k][:H, 0, 0].numpy()]) for k in sorted(last)]))
        loop.close()
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,depth", [(2, 0), (2, 3), (3, 4)])
def test_bench_frame_loop_with_frames_in_flight(world, depth, tmp_path):
    out = str(tmp_path / "loop.npy")
    mp.spawn(_frame_loop_worker, args=(world, _free_port(), out, depth), nprocs=world, join=True)
    got = np.load(out)
    assert got.shape[0] == (2 if depth == 0 else depth + 2)
    numbers = sorted(int(row[0]) for row in got)
    total = 7 + max(1, 2 * depth)
    assert numbers == list(range(total - len(numbers), total))  # the newest frames, one per buffer
    for row in got:
        assert np.array_equal(row[1:], 1000 * row[0] + np.arange(H, dtype=np.float32)), int(row[0])


def test_band_bookkeeping():
    for h in (1, 7, 8, 9, 1080, 2160):
        for world in (1, 2, 3, 8):
            rows = sorted(r for k in range(world) for r in rd.rows_of_rank(h, k, world))
            assert rows == list(range(h))
            assert rd.padded_bands(h, world) % world == 0 and rd.padded_bands(h, world) >= rd.num_bands(h)
