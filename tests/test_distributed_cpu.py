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


def test_band_bookkeeping():
    for h in (1, 7, 8, 9, 1080, 2160):
        for world in (1, 2, 3, 8):
            rows = sorted(r for k in range(world) for r in rd.rows_of_rank(h, k, world))
            assert rows == list(range(h))
            assert rd.padded_bands(h, world) % world == 0 and rd.padded_bands(h, world) >= rd.num_bands(h)
