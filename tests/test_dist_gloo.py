"""N > 1 path on CPU: POI sharding + the all-gather of POI records (gloo, world_size 2).

The compute between shard and gather is the CPU oracle here (tests may use it); on the GPU
box the same `opencorr_amd.dist` helpers wrap the HIP engines with backend "nccl" (= RCCL).
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import oracle
    from opencorr_amd import synth
    from opencorr_amd.dist import allgather_pois, shard_bounds
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ref, tar = synth.speckle_pair_2d(200, 220, seed=5)
    xs, ys = synth.poi_grid_2d(200, 220, 9, 7, 26)
    xs, ys = xs[:n_total], ys[:n_total]
    lo, hi = shard_bounds(n_total, world, rank)
    pois = oracle.make_pois2d(xs[lo:hi], ys[lo:hi])
    if hi > lo:
        oracle.fftcc2d(ref, tar, 12, 12, pois, threads=1)
        prep = oracle.Prepared2D(ref, tar, threads=1)
        oracle.icgn2d1(prep, 12, 12, 0.001, 10, pois, order=oracle.ORDER_LANES, threads=1)
    full = allgather_pois(torch.from_numpy(pois), n_total)
    # the pipelined form bench.py uses: preallocated buffer, async handle
    per = -(-n_total // world)
    buf = torch.empty((world * per, pois.shape[1]), dtype=torch.float32)
    full2, work = allgather_pois(torch.from_numpy(pois), n_total, out=buf, async_op=True)
    work.wait()
    assert torch.equal(full, full2) and full2.data_ptr() == buf.data_ptr()
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), full.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [63, 62, 1])
def test_shard_allgather_equals_single_process(tmp_path, n_total):
    import torch.multiprocessing as mp
    import oracle
    from opencorr_amd import synth
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_total, str(tmp_path)), nprocs=world, join=True)
    ref, tar = synth.speckle_pair_2d(200, 220, seed=5)
    xs, ys = synth.poi_grid_2d(200, 220, 9, 7, 26)
    want = oracle.make_pois2d(xs[:n_total], ys[:n_total])
    oracle.fftcc2d(ref, tar, 12, 12, want, threads=1)
    oracle.icgn2d1(oracle.Prepared2D(ref, tar, threads=1), 12, 12, 0.001, 10, want, order=oracle.ORDER_LANES, threads=1)
    for r in range(world):
        got = np.load(tmp_path / ("rank%d.npy" % r))
        assert got.shape == want.shape
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "rank %d" % r


def test_shard_bounds_cover_queue_exactly():
    from opencorr_amd.dist import shard_bounds
    for n in (0, 1, 7, 8, 9, 250000, 1999396):
        for g in (1, 2, 3, 4, 8):
            cuts = [shard_bounds(n, g, r) for r in range(g)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(g - 1))
            assert all(hi - lo <= -(-n // g) for lo, hi in cuts)
