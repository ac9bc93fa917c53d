"""Device groups (oc_hip_set_devices) and the chunked host-queue pipeline.

A one-GPU box exercises the whole sharding path by naming device 0 more than once: separate engines, separate
streams, blocks of the queue, peer copies and the all-gather emulation are all real; what a one-GPU box cannot show is
the RCCL all-gather and xGMI traffic (test_group_on_distinct_devices needs >= 2 GPUs).  Bar everywhere: the group's
result is BIT-IDENTICAL to the single engine's (same kernel, per-POI arithmetic independent of the block cut) --
SURVEY 8(e)'s equivalence test.
"""
import os
import struct
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _download(ptr, nbytes, device=0):
    """hipMemcpy of a raw device pointer (a group member's mirror) to a float32 array."""
    import ctypes
    from opencorr_amd import capi
    hip = capi.hip_runtime()  # THE runtime of this process, never a second copy by name
    hip.hipSetDevice.argtypes = [ctypes.c_int]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    assert hip.hipSetDevice(device) == 0
    host = np.empty(nbytes // 4, dtype=np.float32)
    assert hip.hipMemcpy(host.ctypes.data, ptr, nbytes, 2) == 0  # hipMemcpyDeviceToHost
    assert hip.hipSetDevice(0) == 0
    return host


@pytest.fixture(scope="module")
def case2d():
    import opencorr_amd
    from opencorr_amd import synth
    ref, tar = synth.speckle_pair_2d(420, 460, seed=11)
    xs, ys = synth.poi_grid_2d(420, 460, 24, 21, 26)   # 504 POIs
    pois = opencorr_amd.make_pois2d(xs, ys)
    f = opencorr_amd.FFTCC2D(16, 16)
    f.set_images(ref, tar)
    f.compute(pois)
    single = opencorr_amd.ICGN2D1(16, 16, 0.001, 10)
    single.set_images(ref, tar)
    single.prepare()
    want = single.compute(pois.copy())
    return ref, tar, pois, want


@pytest.mark.parametrize("members", [2, 3])
def test_group_host_queue_same_bits(case2d, members):
    import opencorr_amd
    ref, tar, pois, want = case2d
    g = opencorr_amd.ICGN2D1(16, 16, 0.001, 10)
    g.set_devices([0] * members)
    assert g.devices() == [0] * members
    g.set_images(ref, tar)
    g.prepare()
    got = g.compute(pois.copy())
    assert np.array_equal(_bits(got), _bits(want))
    # settings reach every member: another radius / iteration limit through the group handle
    g.set_subset(12, 14)
    g.set_iteration(0.01, 3)
    one = opencorr_amd.ICGN2D1(12, 14, 0.01, 3)
    one.set_images(ref, tar)
    one.prepare()
    assert np.array_equal(_bits(g.compute(pois.copy())), _bits(one.compute(pois.copy())))
    # dissolving the group leaves a working single engine
    g.set_devices([0])
    g.set_images(ref, tar)
    g.prepare()
    assert np.array_equal(_bits(g.compute(pois.copy())), _bits(one.compute(pois.copy())))


def test_group_fftcc_and_shared_images(case2d):
    """FFTCC2D group -> ICGN2D2 group with share_images (member-to-member sharing on the same devices)."""
    import opencorr_amd
    from opencorr_amd import synth
    ref, tar, _, _ = case2d
    xs, ys = synth.poi_grid_2d(420, 460, 24, 21, 26)
    start = opencorr_amd.make_pois2d(xs, ys)
    f1 = opencorr_amd.FFTCC2D(16, 16)
    f1.set_images(ref, tar)
    i1 = opencorr_amd.ICGN2D2(16, 16, 0.001, 10)
    i1.share_images(f1)
    i1.prepare()
    want = i1.compute(f1.compute(start.copy()))
    fg = opencorr_amd.FFTCC2D(16, 16)
    fg.set_devices([0, 0])
    fg.set_images(ref, tar)
    ig = opencorr_amd.ICGN2D2(16, 16, 0.001, 10)
    ig.set_devices([0, 0])
    ig.share_images(fg)
    ig.prepare()
    got = ig.compute(fg.compute(start.copy()))
    assert np.array_equal(_bits(got), _bits(want))


def test_group_device_queue_offsets_and_allgather(case2d):
    import torch
    import opencorr_amd
    ref, tar, pois, want = case2d
    dev = torch.device("cuda", 0)
    g = opencorr_amd.ICGN2D1(16, 16, 0.001, 10)
    g.set_devices([0, 0, 0])
    g.set_images(torch.from_numpy(ref).to(dev), torch.from_numpy(tar).to(dev))
    g.prepare()
    q = torch.from_numpy(pois).to(dev)
    g.compute(q)
    assert np.array_equal(_bits(q.cpu().numpy()), _bits(want))
    # centre offsets travel with their blocks
    rng = np.random.default_rng(5)
    off = rng.uniform(-2, 2, (len(pois), 2)).astype(np.float32)
    one = opencorr_amd.ICGN2D1(16, 16, 0.001, 10)
    one.set_images(ref, tar)
    one.prepare()
    want_off = one.compute_with_offsets(pois.copy(), off)
    q2 = torch.from_numpy(pois).to(dev)
    g.compute_with_offsets(q2, torch.from_numpy(off).to(dev))
    assert np.array_equal(_bits(q2.cpu().numpy()), _bits(want_off))
    # all-gather: every member ends up with the complete result queue in its own mirror
    g.set_tuning("group_allgather", 1)
    q3 = torch.from_numpy(pois).to(dev)
    g.compute(q3)
    torch.cuda.synchronize()
    n, stride = pois.shape[0], pois.shape[1] * 4
    per = -(-n // 3)
    for member in range(3):
        ptr, block = g.group_queue(member)
        assert block == per * stride
        mirror = _download(ptr, 3 * block).reshape(3 * per, pois.shape[1])[:n]
        assert np.array_equal(_bits(mirror), _bits(want)), member
    assert np.array_equal(_bits(q3.cpu().numpy()), _bits(want))


def test_group_3d(tmp_path):
    import opencorr_amd
    from opencorr_amd import synth
    dz, dy, dx = 60, 64, 68
    ref, tar = synth.speckle_pair_3d(dz, dy, dx, seed=3)
    xs, ys, zs = synth.poi_grid_3d(dz, dy, dx, 8, 6, 4, 20)   # 192 POIs
    start = opencorr_amd.make_pois3d(xs, ys, zs)
    f = opencorr_amd.FFTCC3D(8, 8, 8)
    f.set_images(ref, tar)
    i = opencorr_amd.ICGN3D1(8, 8, 8, 0.001, 20)
    i.share_images(f)
    i.prepare()
    want = i.compute(f.compute(start.copy()))
    fg = opencorr_amd.FFTCC3D(8, 8, 8)
    fg.set_devices([0, 0])
    fg.set_images(ref, tar)
    ig = opencorr_amd.ICGN3D1(8, 8, 8, 0.001, 20)
    ig.set_devices([0, 0])
    ig.share_images(fg)
    ig.prepare()
    got = ig.compute(fg.compute(start.copy()))
    assert np.array_equal(_bits(got), _bits(want))


def test_group_rejects_what_it_cannot_do():
    import opencorr_amd
    from opencorr_amd import capi
    s = opencorr_amd.Strain(20.0, 5)
    ids = (__import__("ctypes").c_int * 2)(0, 0)
    assert capi.lib().oc_hip_set_devices(s._h, ids, 2) == capi.ERR_UNSUPPORTED
    g = opencorr_amd.ICGN2D1(16, 16, 0.001, 10)
    with pytest.raises(capi.OpenCorrHipError):
        g.set_devices([0, 99])
    with pytest.raises(capi.OpenCorrHipError):
        g.set_devices([])


def test_group_on_distinct_devices(case2d):
    """The real thing: one member per GPU, RCCL all-gather over xGMI.  Needs >= 2 GPUs."""
    import torch
    import opencorr_amd
    from opencorr_amd import capi
    ndev = capi.device_count()
    if ndev < 2:
        pytest.skip("one GPU visible")
    ref, tar, pois, want = case2d
    ids = list(range(min(ndev, 8)))
    g = opencorr_amd.ICGN2D1(16, 16, 0.001, 10)
    g.set_devices(ids)
    g.set_images(ref, tar)
    g.prepare()
    assert np.array_equal(_bits(g.compute(pois.copy())), _bits(want))
    g.set_tuning("group_allgather", 1)
    q = torch.from_numpy(pois).to(torch.device("cuda", 0))
    g.compute(q)
    torch.cuda.synchronize()
    assert np.array_equal(_bits(q.cpu().numpy()), _bits(want))
    n, floats = pois.shape
    per = -(-n // len(ids))
    for member, d in enumerate(ids):
        ptr, block = g.group_queue(member)
        host = _download(ptr, len(ids) * block, device=d)
        assert np.array_equal(_bits(host.reshape(-1, floats)[:n]), _bits(want)), member


def test_rccl_allgather_on_a_one_rank_communicator(case2d):
    """The RCCL binding itself -- dlopen of librccl, the version check, ncclCommInitAll, ncclGroupStart/End and an
    ncclAllGather of ncclUint8 blocks -- executed on ONE GPU: with "group_force_rccl" a lone engine's DEVICE queue goes
    down the group path as a group of one, whose all-gather is a one-rank ncclAllGather (queue -> mirror).  The mirror
    is written by RCCL alone (no copy of the leader's block precedes it), so equal bits prove the call moved the
    records.  The peer-copy emulation used by the [0, 0, 0] groups is NOT involved here."""
    import torch
    import opencorr_amd
    ref, tar, pois, want = case2d
    dev = torch.device("cuda", 0)
    g = opencorr_amd.ICGN2D1(16, 16, 0.001, 10)
    g.set_images(torch.from_numpy(ref).to(dev), torch.from_numpy(tar).to(dev))
    g.prepare()
    g.set_tuning("group_allgather", 1)
    g.set_tuning("group_force_rccl", 1)
    n, floats = pois.shape
    for rep in range(2):  # the second call reuses the cached communicator
        q = torch.from_numpy(pois).to(dev)
        g.compute(q)
        torch.cuda.synchronize()
        assert np.array_equal(_bits(q.cpu().numpy()), _bits(want))
        ptr, block = g.group_queue(0)
        assert block == n * floats * 4
        assert np.array_equal(_bits(_download(ptr, block).reshape(n, floats)), _bits(want)), rep
    # 3D records (124 B, not a multiple of 8) through the same call
    from opencorr_amd import synth
    ref3, tar3 = synth.speckle_pair_3d(64, 72, 80, seed=3)
    xs, ys, zs = synth.poi_grid_3d(64, 72, 80, 4, 4, 4, 22)
    p3 = opencorr_amd.make_pois3d(xs, ys, zs)
    f3 = opencorr_amd.FFTCC3D(8, 8, 8)
    f3.set_images(ref3, tar3)
    want3 = f3.compute(p3.copy())
    g3 = opencorr_amd.FFTCC3D(8, 8, 8)
    g3.set_images(ref3, tar3)
    g3.set_tuning("group_allgather", 1)
    g3.set_tuning("group_force_rccl", 1)
    q3 = torch.from_numpy(p3).to(dev)
    g3.compute(q3)
    torch.cuda.synchronize()
    ptr, block = g3.group_queue(0)
    assert block == p3.size * 4
    assert np.array_equal(_bits(_download(ptr, block).reshape(p3.shape)), _bits(want3))
    # dissolving / destroying drops the communicator without complaint
    g.set_devices([0])
    g.close()
    g3.close()


def test_bench_nccl_backend_at_world_size_one():
    """bench.py's N > 1 control flow (init_process_group(backend="nccl"), double-buffered queues, async all-gather,
    barriers, max-over-ranks) with WORLD_SIZE = 1 and OC_BENCH_FORCE_DIST=1: torch.distributed's RCCL backend executes
    on the one GPU of this box.  Small workload; the numbers mean nothing."""
    import json
    import socket
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, OC_BENCH_FORCE_DIST="1", OC_BENCH_STRICT="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--size", "1024", "--pois", "64", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 1 and rec["config"]["collective"].startswith("RCCL")
    assert rec["config"]["all_gather_alone_ms"] is not None and rec["config"]["all_gather_alone_ms"] > 0
    assert rec["multi_gpu_check"]["backend"] == "nccl" and rec["multi_gpu_check"]["gathered_equals_local_bits"]
    assert rec["multi_gpu_check"]["ok"] and rec["multi_gpu_check"]["problems"] == []
    assert rec["config"]["converged_pois"] >= 0.99 * rec["config"]["total_pois"]


@pytest.mark.parametrize("nranks", [2, 8])
def test_bench_control_flow_at_n2_and_n8_on_one_device(nranks):
    """The WHOLE N > 1 control flow of bench.py with N = 2 and N = 8 ranks on the one GPU of this box (OC_BENCH_ONE_DEVICE=1:
    every rank on device 0, gloo instead of RCCL -- the RCCL backend itself runs in the test above): the pitch-preserving
    weak layout (a x b image tiles), the block cuts of an N-rank queue, the broadcast image pair, double-buffered queues
    with overlapped all-gathers, barriers, max-over-ranks, and -- strict mode -- multi_gpu_check for N ranks: every rank's
    block of the gathered queue equals what it computed and rank 0 re-solves a sample of every other rank's block bit for
    bit.  No scaling number comes out of this (N processes share one GPU); it shows the N = 8 path EXECUTES."""
    import json
    import socket
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, OC_BENCH_ONE_DEVICE="1", OC_BENCH_STRICT="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    tile, per_side = 512, 40
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", str(nranks), "--steps", "2", "--warmup", "1",
           "--size", str(tile), "--pois", str(per_side), "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    chk = rec["multi_gpu_check"]
    assert rec["n_gpus"] == nranks and rec["scaling"] == "weak" and chk["world_size"] == nranks and chk["backend"] == "gloo"
    assert chk["ok"] and chk["problems"] == [] and chk["gathered_equals_local_bits"] and chk["resolved_sample_bit_identical"]
    assert chk["resolved_sample_of_other_ranks"] >= (nranks - 1) * 500
    assert not chk["devices_distinct"]                       # one GPU: said so, not hidden
    assert rec["config"]["total_pois"] == nranks * per_side * per_side
    a = {2: 1, 8: 2}[nranks]
    assert "%dx%d speckle pair" % (nranks // a * tile, a * tile) in rec["config"]["workload"]   # width x height
    assert rec["config"]["converged_pois"] >= 0.99 * rec["config"]["total_pois"]
    assert chk["ms_per_step_gather_not_overlapped"] > 0


@pytest.mark.parametrize("nranks", [2, 8])
def test_bench_workload_e_control_flow_at_n2_and_n8_on_one_device(nranks):
    """`bench.py --workload E` (BASELINE configs[4]: the DVC path, FFTCC3D -> ICGN3D1, the queue cut into N contiguous blocks,
    one all-gather of POI3D records; src/oc_icgn.cpp:1492-1500) with N = 2 and N = 8 ranks on the one GPU of this box, on a
    shrunk volume (96^3, 9^3 = 729 POIs): the broadcast volume pair, the block cuts (the last rank's block is short), the
    double-buffered queues with overlapped all-gathers of 124-byte records, and -- strict mode -- multi_gpu_check: every rank's
    block of the gathered queue equals what it computed, rank 0 re-solves every other rank's POIs bit for bit."""
    import json
    import socket
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, OC_BENCH_ONE_DEVICE="1", OC_BENCH_STRICT="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", str(nranks), "--steps", "2", "--warmup", "1",
           "--workload", "E", "--size", "96", "--pois", "9", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    chk = rec["multi_gpu_check"]
    assert rec["n_gpus"] == nranks and rec["scaling"] == "strong" and chk["world_size"] == nranks and chk["backend"] == "gloo"
    assert "FFTCC3D" in rec["metric"] and rec["config"]["total_pois"] == 729
    assert chk["ok"] and chk["problems"] == [] and chk["gathered_equals_local_bits"] and chk["resolved_sample_bit_identical"]
    per = -(-729 // nranks)
    assert chk["resolved_sample_of_other_ranks"] == 729 - per          # blocks of <= 512 POIs are re-solved whole
    assert rec["config"]["converged_pois"] >= 0.95 * 729
    assert rec["roofline"]["kernel"].startswith("icgn3d1") and rec["roofline"]["frac"] > 0


def test_group_of_eight_on_config_d_queue():
    """oc_hip_set_devices with EIGHT members over BASELINE config D's whole queue (8192^2 pair, 1414 x 1414 = 1 999 396 POIs,
    the 8-GPU configuration), device 0 named eight times: eight engines with their own tables (8 x 4.3 GB), eight streams,
    eight blocks of ceil(n / 8) records pulled and pushed with peer copies, the all-gather into every member's mirror
    (emulated by peer copies: a communicator cannot hold a device twice) -- bit-identical to the single engine, for the
    FFTCC2D -> ICGN2D1 sequence on a DEVICE queue.  The shard cuts and the padded last block of an 8-way split of this
    queue are thereby executed under the driver's eyes; what stays unexecuted without a node is only RCCL over distinct
    devices."""
    import torch
    import opencorr_amd
    from opencorr_amd import synth
    dev = torch.device("cuda", 0)
    side, r, nside = 8192, 16, 1414
    ref, tar = synth.speckle_pair_2d(side, side, seed=20260925, device=dev)
    xs, ys = synth.poi_grid_2d(side, side, nside, nside, r + 8)
    n = len(xs)
    assert n == 1999396 and n % 8 != 0
    pristine = torch.from_numpy(opencorr_amd.make_pois2d(xs, ys)).to(dev)
    f1 = opencorr_amd.FFTCC2D(r, r)
    f1.set_images(ref, tar)
    g1 = opencorr_amd.ICGN2D1(r, r, 0.001, 10)
    g1.share_images(f1)
    g1.prepare()
    want = pristine.clone()
    f1.compute(want)
    g1.compute(want)
    torch.cuda.synchronize()
    want = want.cpu().numpy()
    g1.close()
    f1.close()
    f8 = opencorr_amd.FFTCC2D(r, r)
    f8.set_devices([0] * 8)
    f8.set_images(ref, tar)
    g8 = opencorr_amd.ICGN2D1(r, r, 0.001, 10)
    g8.set_devices([0] * 8)
    g8.set_images(ref, tar)
    g8.prepare()
    g8.set_tuning("group_allgather", 1)
    assert g8.devices() == [0] * 8
    q = pristine.clone()
    f8.compute(q)
    g8.compute(q)
    torch.cuda.synchronize()
    assert np.array_equal(_bits(q.cpu().numpy()), _bits(want))
    floats = want.shape[1]
    per = -(-n // 8)
    for member in (0, 3, 7):
        ptr, block = g8.group_queue(member)
        assert block == per * floats * 4
        mirror = _download(ptr, 8 * block).reshape(8 * per, floats)[:n]
        assert np.array_equal(_bits(mirror), _bits(want)), member
    g8.close()
    f8.close()


def test_host_pipeline_chunks_change_no_bits():
    """Host queues travel in chunks (H2D / kernels / D2H of neighbouring chunks overlap): same bits as one piece and as
    the device-resident queue."""
    import torch
    import opencorr_amd
    from opencorr_amd import synth
    dev = torch.device("cuda", 0)
    ref, tar = synth.speckle_pair_2d(1024, 1024, seed=20260925, device=dev)
    xs, ys = synth.poi_grid_2d(1024, 1024, 192, 192, 24)   # 36 864 POIs -> 3 chunks of 16 384
    start = opencorr_amd.make_pois2d(xs, ys)
    f = opencorr_amd.FFTCC2D(16, 16)
    f.set_images(ref, tar)
    g = opencorr_amd.ICGN2D1(16, 16, 0.001, 10)
    g.share_images(f)
    g.prepare()
    whole = start.copy()
    f.set_tuning("host_chunk", 0)
    g.set_tuning("host_chunk", 0)
    g.compute(f.compute(whole))
    chunked = start.copy()
    f.set_tuning("host_chunk", 16384)
    g.set_tuning("host_chunk", 16384)
    g.compute(f.compute(chunked))
    assert np.array_equal(_bits(chunked), _bits(whole))
    resident = torch.from_numpy(start).to(dev)
    g.compute(f.compute(resident))
    assert np.array_equal(_bits(resident.cpu().numpy()), _bits(whole))
    assert (whole[:, 16] > 0.9).mean() > 0.99


def test_cpp_shim_with_device_group_env(tmp_path, speckle_small):
    """An unmodified OpenCorr-style main with OC_HIP_DEVICES=0,0: every engine becomes a two-member group; the output
    file must equal the plain run's byte for byte."""
    from opencorr_amd import synth
    from tests.test_cpp_shim import _build_driver
    ref, tar = speckle_small
    h, w = ref.shape
    xs, ys = synth.poi_grid_2d(h, w, 14, 12, 28)
    inp = tmp_path / "in.bin"
    with open(inp, "wb") as f:
        f.write(struct.pack("<5i2f", h, w, 16, 16, len(xs), 0.001, 10.0))
        f.write(np.ascontiguousarray(ref, np.float32).tobytes())
        f.write(np.ascontiguousarray(tar, np.float32).tobytes())
        f.write(xs.astype(np.float32).tobytes())
        f.write(ys.astype(np.float32).tobytes())
    exe = _build_driver(tmp_path)
    subprocess.check_call([exe, str(inp), str(tmp_path / "plain.bin")])
    env = dict(os.environ, OC_HIP_DEVICES="0,0")
    subprocess.check_call([exe, str(inp), str(tmp_path / "group.bin")], env=env)
    assert open(tmp_path / "plain.bin", "rb").read() == open(tmp_path / "group.bin", "rb").read()
