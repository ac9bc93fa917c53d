#!/usr/bin/env python
"""bench.py -- converged POIs/s of the FFTCC2D -> ICGN2D1 hot path on N MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over the rank's POI block with everything
resident in HBM: reset the POI records, FFTCC2D::compute (integer-pixel guess),
ICGN2D1::compute (sub-pixel refinement), then -- for N > 1 -- the RCCL all-gather of
the POI records.  prepare() (gradients + 64 B/px bicubic LUT) is done once before the
timed region and reported separately.

Workload (BASELINE.json configs[1], SURVEY.md 8d "B"): 4096 x 4096 synthetic speckle
pair, r = 16 (33 x 33 subset, 32 x 32 FFTCC window), 500 x 500 = 250 000 POIs,
conv 1e-3, stop 10.  N > 1 is weak scaling: 250 000 POIs per GPU cut from one
N*250 000-POI queue over a pair replicated on every GPU.  The image grows with N so that
the POI pitch -- i.e. how much neighbouring subsets overlap, which sets the cache behaviour
of the kernel -- stays what it is at N = 1: 4096 x 8192 with 1000 x 500 POIs at N = 2,
8192 x 8192 with 1000 x 1000 at N = 4; N = 8 is BASELINE config "D" as written (8192 x 8192,
1414 x 1414 POIs; a denser grid, 8192^2 being the largest image the 32-bit LUT offsets address).

For N > 1 the all-gather of step k overlaps the correlation of step k+1 (double-buffered
queues, `async_op=True`): a production pipeline streams image pairs, and xGMI moving one
pair's records while the next pair is being correlated is how it would run.  All K gathers
complete inside the timed region.

Besides the contract fields the JSON line carries
  roofline     -- ICGN2D1 kernel: algorithmic bytes (3*N2*4 + k*N2*64 + 200 per
                  POI with each POI's own iteration count k) / hipEvent-timed kernel duration
                  vs the 8 TB/s HBM3E peak,
  cpu_baseline -- the CPU oracle (float32 restatement of the reference, OpenMP) timed on a
                  bounded sample of the same workload on this box's host cores (rank 0, N = 1).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
# HBM bytes per ICGN2D1 launch of THIS workload from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE in their
# own runs, corrected as the guide prescribes); written by tools/pmc_traffic.py, see tools/gpu_round.sh
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "icgn2d1_hbm_traffic_configB.json")
RX = RY = 16
CONV, STOP = 0.001, 10.0
POIS_PER_GPU_SIDE = 500


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=0, help="override image side (debug)")
    ap.add_argument("--pois", type=int, default=0, help="override POIs per GPU side (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true",
                    help="skip the side measurements (CPU baseline, host-queue rate): profiler runs")
    ap.add_argument("--cpu-sample", type=int, default=125000)
    return ap.parse_args()


def algorithmic_bytes_icgn2d1(pois_np, rx, ry):
    """B1 = 3*N2*4 + k*N2*64 + 200 per POI that ran k iterations; 200 B otherwise."""
    n2 = (2 * rx + 1) * (2 * ry + 1)
    it = pois_np[:, 17].astype(np.float64)
    ran = it > 0
    return float(ran.sum() * (3 * n2 * 4 + 200) + it[ran].sum() * n2 * 64 + (~ran).sum() * 200), float(it[ran].mean() if ran.any() else 0.0)


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    import opencorr_amd
    from opencorr_amd import synth
    from opencorr_amd.dist import allgather_pois, shard_bounds

    # ---- workload ---------------------------------------------------------------------
    per_side = args.pois or POIS_PER_GPU_SIDE
    n_total = world * per_side * per_side
    # image (height, width) and POI grid (nx, ny) per world size: constant POI pitch up to N = 4
    if args.size:
        height = width = args.size
        nx = int(np.floor(np.sqrt(n_total)))
        ny = -(-n_total // nx)
    elif world == 1:
        height, width, nx, ny = 4096, 4096, per_side, per_side
    elif world == 2:
        height, width, nx, ny = 4096, 8192, 2 * per_side, per_side
    elif world == 4:
        height, width, nx, ny = 8192, 8192, 2 * per_side, 2 * per_side
    else:
        height = width = 8192
        nx = int(np.floor(np.sqrt(n_total)))
        ny = -(-n_total // nx)
    t0 = time.time()
    ref, tar = synth.speckle_pair_2d(height, width, seed=20260925, device=dev)
    xs, ys = synth.poi_grid_2d(height, width, nx, ny, RX + 8)
    xs, ys = xs[:n_total], ys[:n_total]
    n_total = len(xs)
    lo, hi = shard_bounds(n_total, world, rank)
    pristine = torch.from_numpy(opencorr_amd.make_pois2d(xs[lo:hi], ys[lo:hi])).to(dev)
    # two queues (and two gather buffers): step k+1 fills one while the all-gather of step k reads the other
    queues = [pristine.clone(), pristine.clone()] if world > 1 else [pristine.clone()]
    per_rank = -(-n_total // world)
    gather_bufs = [torch.empty((world * per_rank, pristine.shape[1]), dtype=pristine.dtype, device=dev)
                   for _ in queues] if world > 1 else []
    pending = [None] * len(queues)
    pois = queues[0]
    gen_s = time.time() - t0

    stream = torch.cuda.current_stream().cuda_stream
    fftcc = opencorr_amd.FFTCC2D(RX, RY, device=local_rank)
    fftcc.set_stream(stream)
    fftcc.set_images(ref, tar)
    icgn = opencorr_amd.ICGN2D1(RX, RY, CONV, STOP, device=local_rank)
    icgn.set_stream(stream)
    icgn.share_images(fftcc)
    torch.cuda.synchronize()
    t0 = time.time()
    icgn.prepare()
    torch.cuda.synchronize()
    prepare_ms = (time.time() - t0) * 1e3

    gathered = None
    step_no = 0

    def step():
        nonlocal gathered, pois, step_no
        b = step_no % len(queues)
        step_no += 1
        if pending[b] is not None:
            pending[b].wait()  # the gather that last read this queue / wrote this buffer has finished
            pending[b] = None
        pois = queues[b]
        pois.copy_(pristine)
        fftcc.compute(pois)
        icgn.compute(pois)
        if world > 1:
            gathered, pending[b] = allgather_pois(pois, n_total, out=gather_bufs[b], async_op=True)

    def drain():
        for b in range(len(pending)):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    icgn.profile_enable(True)
    fftcc.profile_enable(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    icgn_ms, icgn_launches = icgn.profile_read()
    fftcc_ms, fftcc_launches = fftcc.profile_read()

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    full = gathered if world > 1 else pois
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    full_np = full.cpu().numpy()
    converged = int((full_np[:, 16] >= 0).sum())
    local_np = pois.cpu().numpy()

    if rank == 0:
        value = converged * args.steps / elapsed
        alg_bytes, mean_iter = algorithmic_bytes_icgn2d1(local_np, RX, RY)
        icgn_avg_ms = icgn_ms / max(icgn_launches, 1)
        achieved = alg_bytes / (icgn_avg_ms * 1e-3) / 1e9 if icgn_avg_ms > 0 else 0.0
        out = {
            "metric": "converged POIs/sec (FFTCC+ICGN2D1, 33x33 subset)",
            "value": value,
            "unit": "POI/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": ("B: %dx%d speckle pair, r=16 (33x33 subset, 32x32 FFTCC window), %d POIs/GPU, "
                             "FFTCC2D init -> ICGN2D1 conv=1e-3 stop=10" % (width, height, hi - lo)),
                "total_pois": n_total,
                "converged_pois": converged,
                "mean_iterations": mean_iter,
                "collective": ("RCCL all_gather of POI records, overlapped with the next step's kernels"
                               if world > 1 else "none"),
            },
            "roofline": {
                "kernel": "icgn2d1_kernel",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic(world),
                "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE; compare with algorithmic_bytes_per_launch)",
                "algorithmic_bytes_per_launch": alg_bytes,
                "avg_launch_ms": icgn_avg_ms,
                "launches_timed": icgn_launches,
            },
            "stage_ms": {
                "fftcc_pipeline_avg": fftcc_ms / max(fftcc_launches, 1),
                "icgn_kernel_avg": icgn_avg_ms,
                "prepare_once": prepare_ms,
                "generate_inputs_s": gen_s,
            },
        }
        # side measurements, skipped with --no-cpu-baseline so that a profiler sees only warm-up + timed steps
        if world == 1 and not args.no_cpu_baseline:
            out["pcie_inclusive"] = host_queue_rate(fftcc, icgn, pristine, converged)
            out["cpu_baseline"] = cpu_baseline(ref, tar, xs, ys, args.cpu_sample)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def host_queue_rate(fftcc, icgn, pristine, converged, reps=3):
    """The same step when the caller hands over a HOST queue (what the C++ shim's compute(std::vector<POI2D>&)
    does): every compute() then copies the 25 MB AoS to the GPU and back.  Reported beside `value`, never as it."""
    host0 = pristine.cpu().numpy()
    best = None
    for _ in range(reps + 1):
        q = host0.copy()
        t0 = time.perf_counter()
        fftcc.compute(q)
        icgn.compute(q)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return {"ms_per_step": best * 1e3, "value": converged / best, "unit": "POI/s",
            "note": "pageable host POI queue: H2D + D2H of the AoS around FFTCC2D and around ICGN2D1"}


def pmc_traffic(world):
    """HBM bytes per ICGN launch from the committed PMC record (collected on the N = 1 workload)."""
    if world != 1 or not os.path.exists(TRAFFIC_JSON):
        return None
    with open(TRAFFIC_JSON) as f:
        return float(json.load(f)["hbm_bytes_per_launch"])


def cpu_baseline(ref, tar, xs, ys, sample):
    """The CPU oracle (OC_ORDER_SEQ, i.e. the reference's loop order) on a strided sample of the
    same POI queue, all host cores, best of 3, FFTCC + ICGN compute only (prepare excluded like
    on the GPU side)."""
    import oracle
    ref_h = ref.cpu().numpy()
    tar_h = tar.cpu().numpy()
    stride = max(1, len(xs) // sample)
    sx, sy = xs[::stride], ys[::stride]
    cores = oracle.max_threads()
    prep = oracle.Prepared2D(ref_h, tar_h)
    best, conv = None, 0
    for _ in range(3):
        p = oracle.make_pois2d(sx, sy)
        t0 = time.perf_counter()
        oracle.fftcc2d(ref_h, tar_h, RX, RY, p, threads=cores)
        t1 = time.perf_counter()
        oracle.icgn2d1(prep, RX, RY, CONV, STOP, p, order=oracle.ORDER_SEQ, threads=cores)
        t2 = time.perf_counter()
        if best is None or (t2 - t0) < best[0]:
            best = (t2 - t0, t1 - t0, t2 - t1)
            conv = int((p[:, 16] >= 0).sum())
    return {
        "value": conv / best[0],
        "unit": "POI/s",
        "cores": cores,
        "kind": "port",
        "sample": "every %d-th POI of the same queue (%d POIs), FFTCC2D+ICGN2D1 compute, best of 3; "
                  "fftcc %.3f s, icgn %.3f s" % (stride, len(sx), best[1], best[2]),
    }


if __name__ == "__main__":
    main()
