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
N*250 000-POI queue over a pair replicated on every GPU (rendered on rank 0, broadcast to the others).  The image grows with N so that
the POI pitch -- i.e. how much neighbouring subsets overlap, which sets the cache behaviour
of the kernel -- stays what it is at N = 1 for EVERY N: a x b tiles of 4096 x 4096 px with 500 x 500 POIs each, a = the largest
divisor of N up to sqrt(N), b = N / a (`weak_layout`): 4096 x 8192 with 1000 x 500 POIs at N = 2, 8192 x 8192 with 1000 x 1000 at
N = 4, 8192 x 16384 with 2000 x 1000 at N = 8 (2^27 px: inside the 2^28 px the per-plane 32-bit LUT offsets address,
check_image2d_limits in capi.hip; N <= 16).  BASELINE config "D" as written (8192 x 8192, 1414 x 1414 POIs: a denser grid) is
`--scaling strong`.

For N > 1 the all-gather of step k overlaps the correlation of step k+1 (double-buffered
queues, `async_op=True`): a production pipeline streams image pairs, and xGMI moving one
pair's records while the next pair is being correlated is how it would run.  All K gathers
complete inside the timed region.

`--scaling strong` fixes the total work instead (BASELINE config D: 8192 x 8192 pair, 1414 x 1414 POIs, cut into N
blocks); the default is weak scaling as described above.

Warm-up is exactly W steps: nothing else runs the hot path before the timed region (round 2 ran 25 extra "settle"
steps first; `--settle` still exists for experiments, defaults to 0, and is reported in the line when used).

Besides the contract fields the JSON line carries
  value_pcie_inclusive -- SURVEY 8(d)'s own definition of the metric (host POI queue: H2D + D2H of the AoS inside the
                  timed region) as a top-level scalar next to `value` (which the bench contract defines on HBM-resident
                  inputs); details in `pcie_inclusive`,
  roofline_secondary -- the same kind of statement for the dominant kernels of the other BASELINE configs
                  (fftcc2d_fused32x2 on B, icgn2d_kernel<12> on C, fftcc3d_fused32 and icgn3d1_kernel on E), each with the
                  SURVEY 8(d) byte formula, hipEvent-timed on the engines' stream in this run,
  multi_gpu_check -- N > 1 only: world size and device distinctness asserted, rank 0 re-solves a strided sample of every
                  other rank's block and compares it bit for bit with the gathered records, and the step is timed once more
                  with the all-gather NOT overlapped,
  roofline     -- ICGN2D1 kernel.  `combined` (round 4) is the ONE ceiling: max(gather side, VALU side) of the kernel's own
                  instruction stream, from a micro-benchmark of its sweep (tools/ubench/coissue_ubench.hip: gathers + VALU mix =
                  gathers alone, i.e. perfect overlap inside the sweep) and the kernel's VALU wave-instruction count (PMC) --
                  frac = ceiling / measured = 0.70 (what is missing is overlap BETWEEN the kernel's phases: DESIGN.md 4.1); the byte-rate fractions below are kept for continuity.  HBM is not it (traffic <= 9 % of peak:
                  neighbouring subsets share their table entries in L1/L2, so the SURVEY 8(d) byte count / time exceeds
                  the HBM peak); ablations show the gather path of the 64-byte table entries to be the
                  larger limiter (-13 % without two thirds of the gathers, -3 % without two thirds of the polynomials),
                  so `achieved` = the SURVEY 8(d) bytes (3*N2*4 + k*N2*64 + 200 per POI, k = that POI's iteration
                  count) / the hipEvent-timed kernel duration against `peak` = the guide's aggregate L2 figure
                  (34.5 TB/s); `gather_ubench` sets the same rate against a compute-free gather of the same pattern,
                  `valu` the reference's own floating point operations (50*N2 + 75*N2*k per POI, every multiply, add,
                  subtract counted once: the parity contract forbids FMA contraction) against the chip's fp32 vector
                  rate for separately rounded operations (256 CUs x 4 SIMD32 x 2.4 GHz = 78.6 Tflop/s, half the
                  157.3 Tflop/s FMA figure of MI355X_MICROARCH.md); `traffic` = HBM bytes per launch from PMC counters
                  (FETCH_SIZE x 2 + WRITE_SIZE, separate passes) and `traffic_l2` = L2-side bytes (TCC_REQ x a request
                  size calibrated on a gather of known byte count), both collected by tools/gpu_profiles.sh over THIS
                  command and committed as profiles/icgn2d1_traffic_configB.json (counters cannot be read from inside
                  the process; `traffic_source` names the file and its age),
  cpu_baseline -- the CPU oracle (float32 restatement of the reference, OpenMP; pinned bit for bit on the reference's
                  own sources, tests/test_oracle_vs_ref.py) timed on a bounded sample of the same workload on this
                  box's host cores (rank 0, N = 1), built -O3 -march=native for timing,
  oht_pair     -- the reference's own example pair (examples/2d_dic/oht_cfrp_{0,4}.bmp, 30 000 POIs, r = 16: mean 4.5
                  iterations, harder than the synthetic field) through the same engines and the same CPU build.
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
# fp32 vector peak for separately rounded multiplies / adds: 256 CUs x 4 SIMD32 x 32 lanes x 2.4 GHz (the guide's
# 157.3 Tflop/s counts an FMA as two operations; FMA contraction is excluded by the parity contract)
VALU_PEAK_TFLOPS = 256 * 4 * 32 * 2.4e9 / 1e12
L2_PEAK_GBS = 34500.0       # aggregate L2 bandwidth, MI355X_MICROARCH.md ("4 MiB per XCD ... ~34.5 TB/s")
# compute-free gather of the kernel's own access pattern in 8-wave workgroups that re-align every two passes (what the
# kernel does since round 3): profiles/r3j_gather_ubench_lockstep.txt; free-running waves reach 20 350 GB/s (r02a)
GATHER_UBENCH_GBS = 29780.0
GATHER_UBENCH_FREE_GBS = 20350.0
# what the CUs' vector L1 ports can deliver at 64 bytes per clock and CU (every byte of the gather passes through them)
L1_PORT_PEAK_GBS = 256 * 64 * 2.4
# HBM bytes per ICGN2D1 launch of THIS workload from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE in their
# own runs, corrected as the guide prescribes); written by tools/pmc_traffic.py, see tools/gpu_profiles.sh
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "icgn2d1_traffic_configB.json")
TRAFFIC_JSON_OLD = os.path.join(ROOT, "profiles", "icgn2d1_hbm_traffic_configB.json")  # rounds 1-2: HBM side only
# round 4: ONE script (tools/gpu_profiles.sh) collects kernel stats + PMC traffic of the dominant kernels of configs B, C and E;
# its records fill roofline.traffic and roofline_secondary[*].traffic (tools/pmc_traffic.py --kernels: `per_kernel`)
TRAFFIC_BY_CONFIG = {c: os.path.join(ROOT, "profiles", "traffic_config%s.json" % c) for c in "BCE"}
# the same collection under the fused arithmetic contract (OC_BENCH_ARITH_FMA=1 tools/gpu_profiles.sh, round 5)
TRAFFIC_BY_CONFIG_FMA = {c: os.path.join(ROOT, "profiles", "traffic_config%s_fma.json" % c) for c in "BCE"}
# sweep of icgn2d_kernel<6> as a micro-benchmark: its gather pattern AND its VALU mix, nothing else (tools/ubench/coissue_ubench.hip)
COISSUE_JSON = os.path.join(ROOT, "profiles", "coissue_ubench.json")
# ds_read2_b32 serves 128 B per clock and CU (MI355X_MICROARCH.md, LDS table): 256 CUs x 128 B x 2.4 GHz
LDS_READ2_PEAK_GBS = 256 * 128 * 2.4
CLOCK_GHZ = 2.4
RX = RY = 16
CONV, STOP = 0.001, 10.0
POIS_PER_GPU_SIDE = 500


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)    # (the defaults are the driver's command line: --steps 20 --warmup 5;
    ap.add_argument("--warmup", type=int, default=5)   #  two warm-up steps -- 7 ms -- leave the clocks 3 % short of their plateau)
    ap.add_argument("--settle", type=int, default=0,
                    help="extra untimed steps BEFORE the W warm-up steps (experiments only: the default run warms up with exactly "
                         "W steps, as the bench contract says; a non-zero value is reported in the JSON line)")
    ap.add_argument("--size", type=int, default=0, help="override the side of one GPU's image tile (debug; default 4096)")
    ap.add_argument("--pois", type=int, default=0, help="override POIs per GPU side (debug; default 500)")
    ap.add_argument("--no-cpu-baseline", action="store_true",
                    help="skip the side measurements (CPU baseline, host-queue rate): profiler runs")
    ap.add_argument("--cpu-sample", type=int, default=125000)
    ap.add_argument("--arith", choices=["sep", "fma"], default="fma" if os.environ.get("OC_BENCH_ARITH_FMA") == "1" else "sep",
                    help="arithmetic contract of the ICGN kernels in the TIMED region: sep = every multiply and add rounds on its own "
                         "(the default build, oracle OC_ORDER_LANES), fma = the per-sample multiply-adds are fused "
                         "(oc_hip_set_tuning arith_fma, oracle OC_ORDER_LANES_FMA).  The default line is `sep`; it reports the "
                         "fused build next to it as `arith_fma`")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="strong: BASELINE config D (8192^2, 1414 x 1414 POIs) cut into N blocks, whatever N is")
    ap.add_argument("--workload", choices=["B", "E"], default="B",
                    help="B (default): the metric's configuration, FFTCC2D -> ICGN2D1 (BASELINE configs[1]; N > 1: weak scaling, or config D "
                         "with --scaling strong).  E: BASELINE configs[4], the DVC path -- 512^3 volume pair, r = 16 (33^3 subvolumes, 32^3 "
                         "FFTCC windows), 37^3 = 50 653 POIs, FFTCC3D -> ICGN3D1 (conv 1e-3, stop 20), the queue cut into N contiguous "
                         "blocks (strong scaling, as the config is written: 'sharded over 8 x MI355X'), volumes replicated, one all-gather "
                         "of the POI3D records (src/oc_icgn.cpp:1492-1500)")
    return ap.parse_args()


def weak_layout(world, tile=4096, per_side=POIS_PER_GPU_SIDE):
    """Weak scaling that keeps the POI pitch (and with it the per-POI cache behaviour) of N = 1 at every N: the image is
    a x b tiles of `tile` x `tile` pixels, each carrying per_side x per_side POIs; a = the largest divisor of N that is
    <= sqrt(N).  Returns (height, width, nx, ny)."""
    if world < 1 or world > 16:
        raise SystemExit("bench.py: weak scaling is defined for 1 <= N <= 16 (2^28-pixel image limit), got %d" % world)
    a = max(d for d in range(1, int(world ** 0.5) + 1) if world % d == 0)
    b = world // a
    return a * tile, b * tile, b * per_side, a * per_side


def algorithmic_bytes_icgn2d1(pois_np, rx, ry):
    """B1 = 3*N2*4 + k*N2*64 + 200 per POI that ran k iterations; 200 B otherwise."""
    n2 = (2 * rx + 1) * (2 * ry + 1)
    it = pois_np[:, 17].astype(np.float64)
    ran = it > 0
    return float(ran.sum() * (3 * n2 * 4 + 200) + it[ran].sum() * n2 * 64 + (~ran).sum() * 200), float(it[ran].mean() if ran.any() else 0.0)


def algorithmic_flops_icgn2d1(pois_np, rx, ry):
    """Floating point operations of ICGN2D1::compute(POI2D*) as the reference writes it (src/oc_icgn.cpp:144-341), every
    multiply / add / subtract once, per sample of the (2rx+1)(2ry+1) subset:
      once:  reference subset mean + zero-mean + norm 4 (src/oc_subset.cpp:46-52); steepest-descent row 4 and the 21
             Hessian terms 42 (:191-205)                                                                    = 50
      per iteration: warp 8 + centre 2 (src/oc_deformation.cpp:94-105, :240); fractional offsets 2, powers 4, the
             16-term polynomial 24 * and 15 + (src/oc_cubic_bspline.cpp:147-177); mean 1, zero-mean + norm 3; error
             image 2 (:260); ZNSSD 2 (:263); numerator 12 (:266-276)                                        = 75
    (the 6 x 6 inverse, warp update and convergence norm are O(1) per iteration and left out)."""
    n2 = (2 * rx + 1) * (2 * ry + 1)
    it = pois_np[:, 17].astype(np.float64)
    ran = it > 0
    return float(ran.sum() * 50 * n2 + it[ran].sum() * 75 * n2)


def sample_slots_icgn2d1(pois_np, rx, ry):
    """Sample slots the interpolation sweeps of a launch occupy: iterations x ceil(N2 / 64) passes x 64 lanes, summed over the POIs."""
    n2 = (2 * rx + 1) * (2 * ry + 1)
    it = pois_np[:, 17].astype(np.float64)
    return float(it[it > 0].sum() * ((n2 + 63) // 64) * 64)


# Mandated VALU work per sample of a subset (once | per iteration), every floating point INSTRUCTION the reference's
# arithmetic needs -- no address arithmetic, no walks, no selects, no reductions:
#   separately rounded (one instruction per multiply / add / subtract):
#     2d1: 50 | 75     (algorithmic_flops_icgn2d1 below lists them)
#     2d2: once  reference mean / zero-mean / norm 4, steepest-descent row 15 (x*x*0.5, x*y, y*y*0.5: 5; ten products), the
#          78 Hessian terms 156 = 175; per iteration  warp 25 (three monomials + 2 x (6 products + 5 adds),
#          src/oc_deformation.cpp:268-282) + centre 2, fractions 2, powers 4, polynomial 39, mean 1, zero-mean + norm 3, error 2,
#          ZNSSD 2, numerator 24 = 104
#     3d1: once  4 + 9 (steepest-descent products) + 156 = 169; per iteration  warp 18 + centre 3, floor / fraction 6, the four
#          basis polynomials per axis 22 x 3 (src/oc_cubic_bspline.cpp:35-53), the 21 four-tap sums 147 (:390-401), mean 1,
#          zero-mean + norm 3, error 2, ZNSSD 2, numerator 24 = 272
#   fused contract (arith_fma: a multiply-add is ONE instruction): 2d1 28 | 49, 2d2 96 | 64, 3d1 90 | 167
MANDATED_INSTR = {("2d1", False): (50, 75), ("2d1", True): (28, 49), ("2d2", False): (175, 104), ("2d2", True): (96, 64),
                  ("3d1", False): (169, 272), ("3d1", True): (90, 167)}
SIMDS = 256 * 4
VALU_ISSUE_CYCLES = 2.0   # one wave64 VALU instruction occupies a SIMD32 for two cycles (MI355X_MICROARCH.md: 64 flop / clk / SIMD with FMA)


def mandated_instr_icgn(pois_np, iter_col, samples, kind, fma):
    """Lane-level floating point instructions the reference's arithmetic needs for this launch (MANDATED_INSTR x samples x
    POIs / iterations that ran)."""
    once, per_it = MANDATED_INSTR[(kind, bool(fma))]
    it = pois_np[:, iter_col].astype(np.float64)
    ran = it > 0
    return float(ran.sum() * once * samples + it[ran].sum() * per_it * samples)


def valu_hardware_block(mandated_lane_instr, retired_wave_instr, measured_ms):
    """ceilings from HARDWARE rates only.  valu_floor_ms = the mandated arithmetic at the chip's VALU issue
    rate (wave-instructions x 2 cycles / (1024 SIMDs x 2.4 GHz)); valu_retired_ms = what the kernel actually retires (PMC
    SQ_INSTS_VALU, committed profile of the same workload and arithmetic mode) at the same rate; overhead = retired / mandated."""
    per_ms = SIMDS * CLOCK_GHZ * 1e9 / VALU_ISSUE_CYCLES * 1e-3          # wave-instructions per millisecond, whole chip
    mandated_wave = mandated_lane_instr / 64.0
    floor = mandated_wave / per_ms
    blk = {"valu_floor_ms": floor, "valu_floor_frac": floor / measured_ms if measured_ms > 0 else None,
           "mandated_wave_instr_per_launch": mandated_wave,
           "valu_retired_ms": None, "valu_retired_frac": None, "overhead": None,
           "issue_rate": "one wave64 VALU instruction per 2 cycles and SIMD32, 1024 SIMDs, 2.4 GHz"}
    if retired_wave_instr:
        blk["retired_wave_instr_per_launch"] = retired_wave_instr
        blk["valu_retired_ms"] = retired_wave_instr / per_ms
        blk["valu_retired_frac"] = blk["valu_retired_ms"] / measured_ms if measured_ms > 0 else None
        blk["overhead"] = retired_wave_instr / mandated_wave
    return blk


def roofline_block(alg_bytes, alg_flops, icgn_avg_ms, icgn_launches, prof, sample_slots=None, hw=None):
    """The `roofline` object of the JSON line (a function so that the CPU tests can exercise it)."""
    secs = icgn_avg_ms * 1e-3
    alg_rate = alg_bytes / secs / 1e9 if secs > 0 else 0.0      # GB/s
    achieved = alg_flops / secs / 1e12 if secs > 0 else 0.0     # Tflop/s
    prof = prof or {}
    stale = prof.get("stale")
    l2_bytes, hbm_bytes = prof.get("l2_bytes_per_launch"), prof.get("hbm_bytes_per_launch")
    l2_rate = l2_bytes / secs / 1e9 if (l2_bytes and secs > 0) else None
    # `frac` is what the hardware counters measure at the level `bound` names: L2 request bytes per launch /
    # the hipEvent-timed launch / the guide's aggregate L2 figure.  The SURVEY 8(d) byte count (which the L1s partly absorb) is
    # `frac_algorithmic`.  Without a PMC record of THIS tree's kernel (`traffic_stale`) only the algorithmic figure exists.
    main_rate = l2_rate if l2_rate is not None else alg_rate
    return {
        "kernel": "icgn2d_kernel<6,...> (ICGN2D1)",
        "bound": "l2",
        "achieved": main_rate,
        "achieved_source": ("PMC: TCC_REQ x the calibrated request size per launch (committed record of this tree's kernel) / the launch "
                            "duration measured with hipEvents in this run" if l2_rate is not None else
                            "SURVEY 8(d) algorithmic bytes / the measured launch duration (no PMC record of this tree's kernel)"),
        "peak": L2_PEAK_GBS,
        "unit": "GB/s",
        "frac": main_rate / L2_PEAK_GBS,
        "achieved_algorithmic": alg_rate,
        "frac_algorithmic": alg_rate / L2_PEAK_GBS,
        "hbm_frac_by_counters": (hbm_bytes / secs / 1e9 / HBM_PEAK_GBS) if (hbm_bytes and secs > 0) else None,
        # HBM bytes per launch by PMC counters -- collected over this very command by tools/gpu_profiles.sh in separate
        # rocprofv3 passes (a process cannot read them itself) and committed with a fingerprint of the kernel's sources; None when
        # no record exists for the workload or the record belongs to other sources (traffic_stale says which)
        "traffic": hbm_bytes,
        "traffic_l2": l2_bytes,
        "traffic_source": prof.get("source"),
        "traffic_stale": stale,
        "l2_counter_frac": (l2_rate or 0.0) / L2_PEAK_GBS,
        "algorithmic_bytes_per_launch": alg_bytes,
        "avg_launch_ms": icgn_avg_ms,
        "launches_timed": icgn_launches,
        "why_not_hbm": ("the 64 B/sample table gather is served by L1/L2 (neighbouring subsets overlap): HBM sees a few percent "
                        "of the algorithmic bytes, so bytes / time exceeds the HBM peak ({:.1f}x) and says nothing"
                        .format(alg_rate / HBM_PEAK_GBS)),
        "gather_ubench": {"value": GATHER_UBENCH_GBS, "unit": "GB/s", "frac": alg_rate / GATHER_UBENCH_GBS,
                          "source": "profiles/r3j_gather_ubench_lockstep.txt",
                          "free_running_value": GATHER_UBENCH_FREE_GBS,
                          "note": "the same gather pattern with no arithmetic at all (tools/ubench/gather_ubench.hip, planar table, "
                                  "8-wave workgroups re-aligned every two passes like the kernel's): the kernel's gather rate as a "
                                  "fraction of that ceiling; free-running waves (the round-2 ceiling) reach free_running_value"},
        "l1_port": {"peak": L1_PORT_PEAK_GBS, "unit": "GB/s", "frac": alg_rate / L1_PORT_PEAK_GBS,
                    "note": "algorithmic bytes over 64 B per clock and CU of vector-L1 bandwidth (256 CUs, 2.4 GHz): since the lockstep "
                            "sweeps the L2s see 0.68 x the algorithmic bytes (traffic_l2) -- the rest are L1 hits"},
        "valu": {"achieved": achieved, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / VALU_PEAK_TFLOPS,
                 "algorithmic_flops_per_launch": alg_flops,
                 "note": "the reference's own fp32 operations over the chip's rate for separately rounded operations"},
        # hardware-rate ceilings (round 5): the mandated arithmetic and the retired VALU stream at the chip's issue rate
        "valu_hw": hw,
        # the gather side of `combined` is a measured hardware ceiling (the sweep's own gather pattern, compute-free); its VALU
        # side is an OWN-MIX ESTIMATE (the kernel's retired count x the cycles its own mix sustains) -- not a hardware ceiling:
        # read `valu_hw` for that
        "combined": (combined_ceiling(icgn_avg_ms, sample_slots, prof.get("valu_wave_instr_per_launch"))
                     if sample_slots else None),
        "hbm_traffic_profiled": prof or None,
    }


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # the driver launches N ranks for --gpus N; anything else is a mis-launch, and a number from it would be mislabelled
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d (launch with torch.distributed.run --nproc-per-node %d)"
                         % (args.gpus, world, args.gpus))
    # OC_BENCH_ONE_DEVICE=1 (tests only): every rank uses GPU 0 and the collective runs over gloo, so that the N > 1
    # control flow (sharding, double-buffered queues, overlapped gathers, barriers) can be exercised on a one-GPU box.
    # Numbers of such a run mean nothing.
    one_device = os.environ.get("OC_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    # OC_BENCH_FORCE_DIST=1 (tests only): the N > 1 control flow -- process group, all-gather, barriers -- also at
    # WORLD_SIZE = 1, so that torch.distributed's RCCL backend executes on a one-GPU box
    force_dist = os.environ.get("OC_BENCH_FORCE_DIST") == "1"
    dist_on = world > 1 or force_dist
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = None
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = "gloo" if one_device else "nccl"
        if one_device:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", device_id=dev, rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)

    import opencorr_amd
    from opencorr_amd import synth
    from opencorr_amd.dist import allgather_pois, shard_bounds

    # ---- workload ---------------------------------------------------------------------
    is3d = args.workload == "E"
    strong = args.scaling == "strong" or is3d
    t0 = time.time()
    if is3d:
        # E: the volume side and the POI grid can be shrunk for tests (--size, --pois); the default is BASELINE configs[4]
        dim = args.size or 512
        per_side = args.pois or 37
        r3 = 16
        zncc_col, iter_col = 18, 19   # POI3D: result.zncc, result.iteration (src/oc_poi.h:187-222)
        # ONE volume pair for all ranks (rank 0 renders, the others receive it: replicas must be bit-identical to be cross-checked)
        if not dist_on or rank == 0:
            ref, tar = synth.speckle_pair_3d(dim, dim, dim, seed=20260927, device=dev)
        else:
            ref = torch.empty((dim, dim, dim), dtype=torch.float32, device=dev)
            tar = torch.empty((dim, dim, dim), dtype=torch.float32, device=dev)
        if dist_on:
            dist.broadcast(ref, src=0)
            dist.broadcast(tar, src=0)
            torch.cuda.synchronize()
        gx3, gy3, gz3 = synth.poi_grid_3d(dim, dim, dim, per_side, per_side, per_side, r3 + 8)
        n_total = len(gx3)

        def make_pois(idx):
            return opencorr_amd.make_pois3d(gx3[idx], gy3[idx], gz3[idx])
        height = width = dim
    else:
        zncc_col, iter_col = 16, 17
        per_side = args.pois or POIS_PER_GPU_SIDE
        n_total = world * per_side * per_side
        if strong:
            n_total = 1414 * 1414
        # image (height, width) and POI grid (nx, ny) per world size: constant POI pitch at every N (weak_layout)
        if strong:
            height = width = 8192
            nx = ny = 1414
        else:
            height, width, nx, ny = weak_layout(world, args.size or 4096, per_side)
        # ONE image pair for all ranks: rank 0 renders it, the others receive it (the GPU renderer adds its speckles with float
        # atomics, so two renderings of the same seed differ in the last bits of a few pixels -- and with them ~40 of 250 000
        # POIs; ranks working on private renderings could not be cross-checked bit for bit, and would not be "replicas")
        if not dist_on or rank == 0:
            ref, tar = synth.speckle_pair_2d(height, width, seed=20260925, device=dev)
        else:
            ref = torch.empty((height, width), dtype=torch.float32, device=dev)
            tar = torch.empty((height, width), dtype=torch.float32, device=dev)
        if dist_on:
            dist.broadcast(ref, src=0)
            dist.broadcast(tar, src=0)
            torch.cuda.synchronize()
        xs, ys = synth.poi_grid_2d(height, width, nx, ny, RX + 8)
        xs, ys = xs[:n_total], ys[:n_total]
        n_total = len(xs)

        def make_pois(idx):
            return opencorr_amd.make_pois2d(xs[idx], ys[idx])
    lo, hi = shard_bounds(n_total, world, rank)
    pristine = torch.from_numpy(make_pois(np.arange(lo, hi))).to(dev)
    # two queues (and two gather buffers): step k+1 fills one while the all-gather of step k reads the other
    queues = [pristine.clone(), pristine.clone()] if dist_on else [pristine.clone()]
    per_rank = -(-n_total // world)
    gather_bufs = [torch.empty((world * per_rank, pristine.shape[1]), dtype=pristine.dtype, device=dev)
                   for _ in queues] if dist_on else []
    pending = [None] * len(queues)
    pois = queues[0]
    gen_s = time.time() - t0

    stream = torch.cuda.current_stream().cuda_stream
    if is3d:
        fftcc = opencorr_amd.FFTCC3D(r3, r3, r3, device=local_rank)
        icgn = opencorr_amd.ICGN3D1(r3, r3, r3, CONV, 20.0, device=local_rank)
    else:
        fftcc = opencorr_amd.FFTCC2D(RX, RY, device=local_rank)
        icgn = opencorr_amd.ICGN2D1(RX, RY, CONV, STOP, device=local_rank)
    fftcc.set_stream(stream)
    fftcc.set_images(ref, tar)
    icgn.set_stream(stream)
    icgn.share_images(fftcc)
    if args.arith == "fma":
        icgn.set_tuning("arith_fma", 1)
    torch.cuda.synchronize()
    t0 = time.time()
    icgn.prepare()
    torch.cuda.synchronize()
    prepare_first_ms = (time.time() - t0) * 1e3   # includes the one-off allocation of the gradient images and the 64 B/px table
    t0 = time.time()
    icgn.prepare()                                # what every further image pair of a sequence costs (buffers are grow-only)
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
        if dist_on:
            gathered, pending[b] = allgather_pois(pois, n_total, out=gather_bufs[b], async_op=True)

    def drain():
        for b in range(len(pending)):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    for _ in range(max(args.settle, 0)):
        step()
    drain()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    icgn.profile_enable(True)
    fftcc.profile_enable(True)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    icgn_ms, icgn_launches = icgn.profile_read()
    fftcc_ms, fftcc_launches = fftcc.profile_read()
    icgn.profile_enable(False)
    fftcc.profile_enable(False)
    # one all-gather on its own, nothing overlapping it (what a caller that needs the field at once would wait for)
    gather_alone_ms = None
    serial_ms = None
    if dist_on:
        # the same K steps with the all-gather NOT overlapped: every step waits for its own gather before the next starts
        dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
            drain()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        serial_ms = (time.perf_counter() - t1) / args.steps * 1e3
        dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            _, w = allgather_pois(queues[0], n_total, out=gather_bufs[0], async_op=True)
            w.wait()
        torch.cuda.synchronize()
        gather_alone_ms = (time.perf_counter() - t1) / 3 * 1e3

    t = torch.tensor([elapsed, serial_ms or 0.0], dtype=torch.float64, device=dev)
    full = gathered if dist_on else pois
    if dist_on:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t[0].item())
    serial_ms = float(t[1].item()) if dist_on else None
    full_np = full.cpu().numpy()
    converged = int((full_np[:, zncc_col] >= 0).sum())
    local_np = pois.cpu().numpy()
    check = multi_gpu_check(args, dist, torch, dev, rank, world, local_rank, one_device, backend, fftcc, icgn, make_pois, n_total,
                            full_np, local_np, lo, hi, serial_ms) if dist_on else None
    if is3d:
        if rank == 0:
            print(json.dumps(line_workload_e(args, torch, dev, local_rank, world, dist_on, elapsed, converged, n_total, hi - lo, dim, per_side,
                                             local_np, icgn_ms, icgn_launches, fftcc_ms, fftcc_launches, prepare_ms, prepare_first_ms,
                                             gen_s, gather_alone_ms, check, ref, tar, gx3, gy3, gz3)), flush=True)
        if dist_on:
            dist.destroy_process_group()
        return

    if rank == 0:
        value = converged * args.steps / elapsed
        alg_bytes, mean_iter = algorithmic_bytes_icgn2d1(local_np, RX, RY)
        alg_flops = algorithmic_flops_icgn2d1(local_np, RX, RY)
        icgn_avg_ms = icgn_ms / max(icgn_launches, 1)
        prof = pmc_profile(world, args.arith == "fma")
        out = {
            "metric": "converged POIs/sec (FFTCC+ICGN2D1, 33x33 subset)",
            "value": value,
            # `value` = the bench contract's definition: whole-job throughput with POIs and images resident in HBM when the timed
            # region starts (the same number as value_device_resident).  SURVEY 8(d) defines the metric over a HOST queue (H2D of
            # the POI records and D2H of the results inside): that rate is value_pcie_inclusive (N = 1, side measurement below)
            "value_device_resident": value,
            "value_definition": "HBM-resident inputs (bench contract); SURVEY 8(d)'s PCIe-inclusive rate: value_pcie_inclusive",
            "unit": "POI/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup + max(args.settle, 0),
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": ("B: %dx%d speckle pair, r=16 (33x33 subset, 32x32 FFTCC window), %d POIs/GPU, "
                             "FFTCC2D init -> ICGN2D1 conv=1e-3 stop=10" % (width, height, hi - lo)),
                "arithmetic": ("fp32, per-sample multiply-adds fused (arith_fma = 1; GPU == oracle OC_ORDER_LANES_FMA bit for bit)"
                               if args.arith == "fma" else
                               "fp32, every multiply and add rounded separately like the reference built for baseline x86-64 "
                               "(GPU == oracle OC_ORDER_LANES bit for bit)"),
                "total_pois": n_total,
                "converged_pois": converged,
                "mean_iterations": mean_iter,
                "collective": ("RCCL all_gather of POI records, overlapped with the next step's kernels"
                               if dist_on else "none"),
                "all_gather_alone_ms": gather_alone_ms,
            },
            "roofline": roofline_block(alg_bytes, alg_flops, icgn_avg_ms, icgn_launches, prof, sample_slots_icgn2d1(local_np, RX, RY),
                                       hw=valu_hardware_block(mandated_instr_icgn(local_np, 17, (2 * RX + 1) * (2 * RY + 1), "2d1", args.arith == "fma"),
                                                              (prof or {}).get("valu_wave_instr_per_launch"), icgn_avg_ms)),
            "stage_ms": {
                "fftcc_pipeline_avg": fftcc_ms / max(fftcc_launches, 1),
                "icgn_kernel_avg": icgn_avg_ms,
                "prepare_once": prepare_ms,
                "prepare_first_call": prepare_first_ms,
                "generate_inputs_s": gen_s,
            },
        }
        if args.settle > 0:
            out["settle_steps"] = args.settle
        if check is not None:
            out["multi_gpu_check"] = check
        # side measurements, skipped with --no-cpu-baseline so that a profiler sees only warm-up + timed steps
        if world == 1 and not dist_on and not args.no_cpu_baseline:
            out["pcie_inclusive"] = host_queue_rate(fftcc, icgn, pristine, converged)
            out["value_pcie_inclusive"] = out["pcie_inclusive"]["value"]
            out["arith_fma"] = other_arith_leg(torch, args, fftcc, icgn, queues[0], pristine, icgn_avg_ms, elapsed / args.steps * 1e3)
            out["paths_8f_row1"] = variant_paths(torch, dev, local_rank, ref, tar, xs, ys, icgn_avg_ms, local_np)
            fftcc_block = secondary_block(
                "fftcc2d_fused32x2_kernel (FFTCC2D, 32x32 window)", "B: 4096^2, r=16, 250 000 POIs",
                (2 * (2 * RX) * (2 * RY) * 4 + 20) * float(hi - lo), fftcc_ms / max(fftcc_launches, 1), fftcc_launches,
                "hbm", HBM_PEAK_GBS, "SURVEY 8(d): 2*M2*4 B in + 20 B out = 8 212 B per POI; the kernel itself is VALU-bound "
                "(about 1.1 k wave-instructions per POI, two POIs per wave, DESIGN.md 4.2)",
                traffic=kernel_traffic("B", "fftcc2d_fused32x2_kernel"))
            del queues, gather_bufs
            out["roofline_secondary"] = [fftcc_block] + secondary_rooflines(dev, local_rank)
            out["cpu_baseline"] = cpu_baseline(ref, tar, xs, ys, args.cpu_sample)
            out["oht_pair"] = oht_pair(local_rank)
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.destroy_process_group()


def line_workload_e(args, torch, dev, local_rank, world, dist_on, elapsed, converged, n_total, n_local, dim, per_side, local_np,
                    icgn_ms, icgn_launches, fftcc_ms, fftcc_launches, prepare_ms, prepare_first_ms, gen_s, gather_alone_ms, check,
                    ref, tar, gx3, gy3, gz3):
    """The JSON line of `--workload E` (BASELINE configs[4]: DVC, FFTCC3D -> ICGN3D1, the queue cut into N blocks)."""
    r = 16
    n3 = (2 * r + 1) ** 3
    it = local_np[:, 19].astype(np.float64)
    ran = it > 0
    alg = float(ran.sum() * (4 * n3 * 4 + 248) + it[ran].sum() * n3 * 256 + (~ran).sum() * 248)
    icgn_avg = icgn_ms / max(icgn_launches, 1)
    fma = args.arith == "fma"
    tr = kernel_traffic("E", "icgn3d1", fma=fma) if world == 1 else None
    roof = secondary_block("icgn3d1_kernel (ICGN3D1)", "E: %d^3, r=16 (33^3), %d POIs on this rank" % (dim, n_local), alg, icgn_avg,
                           icgn_launches, "lds", LDS_READ2_PEAK_GBS,
                           "SURVEY 8(d): 4*N3*4 + k*N3*256 + 248 B per POI; the 256 B per sample and iteration are the 64 tricubic taps, "
                           "served from the LDS-staged coefficient box: judged against the LDS read rate (ds_read2_b32: 128 B per clock and CU)",
                           {"mean_iterations": float(it[ran].mean()) if ran.any() else 0.0,
                            "valu_hw": valu_hardware_block(mandated_instr_icgn(local_np, 19, n3, "3d1", fma),
                                                           (tr or {}).get("valu_wave_instr_per_launch"), icgn_avg)}, traffic=tr)
    out = {
        "metric": "converged POIs/sec (FFTCC3D+ICGN3D1, 33^3 subvolume)",
        "value": converged * args.steps / elapsed,
        "unit": "POI/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup + max(args.settle, 0),
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": ("E: %d^3 blob volume pair, r=16 (33^3 subvolume, 32^3 FFTCC window), %d POIs in all, %d on this rank, "
                         "FFTCC3D init -> ICGN3D1 conv=1e-3 stop=20" % (dim, n_total, n_local)),
            "arithmetic": "fp32, fused per-sample multiply-adds (arith_fma = 1)" if fma else "fp32, every multiply and add rounded separately",
            "total_pois": n_total,
            "converged_pois": converged,
            "collective": ("RCCL all_gather of POI3D records (124 B), overlapped with the next step's kernels" if dist_on else "none"),
            "all_gather_alone_ms": gather_alone_ms,
        },
        "roofline": roof,
        "stage_ms": {"fftcc_pipeline_avg": fftcc_ms / max(fftcc_launches, 1), "icgn_kernel_avg": icgn_avg, "prepare_once": prepare_ms,
                     "prepare_first_call": prepare_first_ms, "generate_inputs_s": gen_s},
    }
    if check is not None:
        out["multi_gpu_check"] = check
    if world == 1 and not dist_on and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_3d(ref, tar, gx3, gy3, gz3, r, min(args.cpu_sample, 6000))
    return out


def cpu_baseline_3d(ref, tar, xs, ys, zs, r, sample):
    """The CPU oracle (the reference's loop order) on a strided sample of config E's queue, all host cores: FFTCC3D + ICGN3D1."""
    import oracle
    build = oracle.use_timing_build()
    ref_h, tar_h = ref.cpu().numpy(), tar.cpu().numpy()
    stride = max(1, len(xs) // sample)
    cores = oracle.max_threads()
    p = oracle.make_pois3d(xs[::stride], ys[::stride], zs[::stride])
    t0 = time.perf_counter()
    oracle.fftcc3d(ref_h, tar_h, r, r, r, p)
    t1 = time.perf_counter()
    prep = oracle.Prepared3D(ref_h, tar_h)
    t2 = time.perf_counter()
    oracle.icgn3d1(prep, r, r, r, CONV, 20.0, p, order=oracle.ORDER_SEQ)
    t3 = time.perf_counter()
    conv = int((p[:, 18] >= 0).sum())
    return {"value": conv / ((t1 - t0) + (t3 - t2)), "unit": "POI/s", "cores": cores, "kind": "port", "build": build,
            "sample": "every %d-th POI of the same queue (%d POIs), FFTCC3D+ICGN3D1 compute, one run; fftcc %.2f s, icgn %.2f s "
                      "(prepare %.2f s excluded)" % (stride, len(p), t1 - t0, t3 - t2, t2 - t1),
            "icgn_only_value": conv / (t3 - t2)}


def multi_gpu_check(args, dist, torch, dev, rank, world, local_rank, one_device, backend, fftcc, icgn, make_pois, n_total, full_np,
                    local_np, lo, hi, serial_ms, sample_per_rank=512):
    """N > 1 self-check (results for G in {1, 2, 4, 8} must be bitwise identical).  Every rank: its own block of
    the gathered queue equals what it computed.  Rank 0 additionally RE-SOLVES a strided sample of every other rank's block
    on its own GPU (same images, same engines) and compares with the gathered records bit for bit -- a rank that computed on
    a stale image, a mis-cut block or a gather that landed records in the wrong slot cannot pass.  Ranks must sit on
    distinct devices (PCI bus ids gathered over the process group)."""
    import opencorr_amd
    # A failed check is reported in the line (multi_gpu_check.ok = false, the offending items named in .problems) and on
    # stderr -- a scaling run keeps its numbers and shows what was wrong with them.  OC_BENCH_STRICT=1 (the tests) turns
    # every failure into an immediate exception instead.
    strict = os.environ.get("OC_BENCH_STRICT") == "1"
    problems = []

    def expect(cond, what):
        if cond:
            return True
        if strict:
            raise AssertionError(what)
        problems.append(what)
        print("bench.py multi_gpu_check: " + what, file=sys.stderr, flush=True)
        return False

    expect(dist.get_world_size() == world == args.gpus, "world size %d != --gpus %d" % (dist.get_world_size(), args.gpus))
    props = torch.cuda.get_device_properties(dev)
    ident = "%s/%s" % (getattr(props, "pci_bus_id", "?"), getattr(props, "uuid", local_rank))
    idents = [None] * world
    dist.all_gather_object(idents, (rank, local_rank, ident))
    distinct = len({i[2] for i in idents}) == world
    if not one_device:
        expect(distinct, "ranks share a device: %r" % (idents,))
    own_ok = bool(np.array_equal(full_np[lo:hi].view(np.uint32), local_np.view(np.uint32)))
    expect(own_ok, "rank %d: its block of the gathered queue differs from what it computed" % rank)
    checked, ok = 0, True
    if rank == 0:
        per = -(-n_total // world)
        for r in range(1, world):
            rlo, rhi = min(r * per, n_total), min((r + 1) * per, n_total)
            if rhi <= rlo:
                continue
            stride = max(1, (rhi - rlo) // sample_per_rank)
            idx = np.arange(rlo, rhi, stride)
            q = torch.from_numpy(make_pois(idx)).to(dev)
            fftcc.compute(q)
            icgn.compute(q)
            torch.cuda.synchronize()
            same = bool(np.array_equal(q.cpu().numpy().view(np.uint32), full_np[idx].view(np.uint32)))
            expect(same, "records gathered from rank %d differ from rank 0's own solution of the same POIs" % r)
            ok = ok and same
            checked += len(idx)
    flag = torch.tensor([1 if (ok and own_ok) else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    all_ok = int(flag.item()) == 1 and not problems
    expect(int(flag.item()) == 1 or bool(problems), "another rank reports a failed check")
    return {"ok": all_ok, "problems": problems, "world_size": dist.get_world_size(), "backend": backend, "devices": [i[2] for i in idents],
            "devices_distinct": distinct, "gathered_equals_local_bits": own_ok,
            "resolved_sample_of_other_ranks": checked, "resolved_sample_bit_identical": ok,
            "ms_per_step_gather_not_overlapped": serial_ms}


def traffic_record_is_current(rec, kernel_regex):
    """Staleness guard: a PMC record belongs to the instruction stream it was collected on.  tools/pmc_traffic.py
    stores opencorr_amd.build.kernel_fingerprint() -- a hash of the kernel family's sources and the compiler flags -- in the record;
    a record without one, or with another tree's, is not reported.  Returns (ok, reason)."""
    from opencorr_amd import build as hip_build
    have = (rec or {}).get("source_fingerprint")
    if not have:
        return False, "the PMC record carries no source fingerprint (collected before round 6): re-run tools/gpu_profiles.sh"
    want = hip_build.kernel_fingerprint(kernel_regex)
    if have != want:
        return False, ("the PMC record was collected on other kernel sources (fingerprint %s, this tree %s): re-run tools/gpu_profiles.sh"
                       % (have, want))
    return True, None


def kernel_traffic(config, kernel_regex, fma=False):
    """HBM / L2-side bytes per launch of one kernel from the committed PMC record of its config (tools/gpu_profiles.sh), or None;
    a record that belongs to other kernel sources comes back as {"stale": reason, "source": file} with no numbers."""
    path = (TRAFFIC_BY_CONFIG_FMA if fma else TRAFFIC_BY_CONFIG).get(config)
    if not path or not os.path.exists(path):
        return None
    with open(path) as f:
        rec = json.load(f)
    k = (rec.get("per_kernel") or {}).get(kernel_regex)
    if not k:
        return None
    ok, why = traffic_record_is_current(k, kernel_regex)
    if not ok:
        return {"stale": why, "source": os.path.relpath(path, ROOT)}
    return {"hbm_bytes_per_launch": k.get("hbm_bytes_per_launch"), "l2_bytes_per_launch": k.get("l2_bytes_per_launch"),
            "l2_hit_rate": k.get("l2_hit_rate"), "rocprof_avg_ms": (k["avg_us"] * 1e-3 if k.get("avg_us") else None),
            "valu_wave_instr_per_launch": k.get("SQ_INSTS_VALU_per_launch"),
            "rocprof_median_ms": (k["median_us"] * 1e-3 if k.get("median_us") else None), "scratch_bytes": k.get("scratch_bytes"),
            "vgpr": k.get("vgpr"), "source": os.path.relpath(path, ROOT) + (" (%s)" % rec["collected"] if rec.get("collected") else "")}


def secondary_block(kernel, config, alg_bytes, avg_ms, launches, bound, peak_gbs, note, extra=None, traffic=None):
    secs = avg_ms * 1e-3
    rate = alg_bytes / secs / 1e9 if secs > 0 else 0.0
    blk = {"kernel": kernel, "config": config, "bound": bound, "achieved": rate, "peak": peak_gbs, "unit": "GB/s",
           "frac": rate / peak_gbs, "traffic": (traffic or {}).get("hbm_bytes_per_launch"), "traffic_stale": (traffic or {}).get("stale"),
           "traffic_l2": (traffic or {}).get("l2_bytes_per_launch"), "traffic_source": (traffic or {}).get("source"),
           "rocprof_avg_ms": (traffic or {}).get("rocprof_avg_ms"), "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": avg_ms,
           "launches_timed": launches, "hbm_frac_of_algorithmic_bytes": rate / HBM_PEAK_GBS,
           "hbm_frac_by_counters": ((traffic or {}).get("hbm_bytes_per_launch") or 0.0) / secs / 1e9 / HBM_PEAK_GBS if secs > 0 else None,
           "note": note}
    if extra:
        blk.update(extra)
    return blk


def combined_ceiling(icgn_avg_ms, sample_slots, valu_instr_per_launch):
    """ONE ceiling for the metric kernel.  tools/ubench/coissue_ubench.hip runs the kernel's interpolation sweep --
    its exact gather pattern in lockstep 8-wave workgroups AND its per-sample VALU mix (the kernel's own device functions) -- and
    nothing else, in three builds: both, gathers only, VALU only.  Measured: `both` = `gathers only` (the VALU work hides under
    the gather), and the VALU-only build sustains 3.2 cycles per VALU wave-instruction and SIMD at the kernel's occupancy.
    Scaled to this run: gather side = both_ms x (this run's sample slots / the benchmark's), VALU side = the kernel's VALU
    wave-instructions (PMC, committed) x that cycle cost; the larger one is the ceiling no schedule of THIS instruction stream
    can beat, frac = ceiling / measured."""
    if not os.path.exists(COISSUE_JSON):
        return None
    with open(COISSUE_JSON) as f:
        u = json.load(f)
    lock = u["lockstep2"]
    per_slot_ms = lock["both_ms"] / (u["samples"] / 64.0)          # ms per wave-pass (64 sample slots)
    gather_ms = per_slot_ms * sample_slots / 64.0
    valu_instr_ubench = u.get("valu_wave_instr_valu_only")          # VALU wave-instructions of the VALU-only build (ISA count x passes)
    cyc = lock["valu_only_ms"] * 1e-3 * CLOCK_GHZ * 1e9 * 1024 / valu_instr_ubench if valu_instr_ubench else None
    valu_ms = valu_instr_per_launch * cyc / (1024 * CLOCK_GHZ * 1e9) * 1e3 if (cyc and valu_instr_per_launch) else None
    ceiling = max(gather_ms, valu_ms or 0.0)
    return {"ceiling_ms": ceiling, "frac": ceiling / icgn_avg_ms if icgn_avg_ms > 0 else None,
            "gather_side_ms": gather_ms, "gather_side_frac": gather_ms / icgn_avg_ms if icgn_avg_ms > 0 else None,
            "valu_side_own_mix_estimate_ms": valu_ms, "valu_cycles_per_instr_measured": cyc,
            "kernel_valu_wave_instr_per_launch": valu_instr_per_launch,
            "sweep_ubench": {"both_ms": lock["both_ms"], "gather_only_ms": lock["gather_only_ms"], "valu_only_ms": lock["valu_only_ms"],
                             "sample_slots": u["samples"]},
            "source": os.path.relpath(COISSUE_JSON, ROOT),
            "note": "both = gathers only: the sweep's VALU work hides completely under its gathers (perfect overlap INSIDE the sweep); "
                    "over the whole kernel the VALU side is the larger one: the kernel's VALU wave-instructions x the cycles per "
                    "instruction its own sweep mix sustains when run alone (an extrapolation from the sweep to the set-up, numerator and "
                    "reduction phases).  What separates the kernel from this ceiling is the overlap BETWEEN its phases (texture-bound "
                    "sweeps, latency-bound set-up / solve) with three lockstep workgroups per CU: DESIGN.md 4.1 lists what was tried"}


def _timed_launches(torch, eng, fn, reps):
    fn()
    torch.cuda.synchronize()
    eng.profile_reset()
    eng.profile_enable(True)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    ms, n = eng.profile_read()
    eng.profile_enable(False)
    return ms / max(n, 1), n


def secondary_rooflines(dev, device, reps=3):
    """Dominant kernels of BASELINE configs C and E on this GPU, each against the SURVEY 8(d) byte formula (hipEvents on the
    engines' stream around every launch, like the main block).  Side measurement: runs after the timed region."""
    import torch
    import opencorr_amd
    from opencorr_amd import synth
    out = []
    stream = torch.cuda.current_stream().cuda_stream
    # ---- C: 4096^2, r = 20, ICGN2D2, 316 x 316 POIs
    r = 20
    ref, tar = synth.speckle_pair_2d(4096, 4096, seed=20260925, device=dev, second_order=dict(uxx=2e-6, vyy=-1e-6))
    xs, ys = synth.poi_grid_2d(4096, 4096, 316, 316, r + 8)
    f = opencorr_amd.FFTCC2D(r, r, device=device)
    f.set_stream(stream)
    f.set_images(ref, tar)
    g = opencorr_amd.ICGN2D2(r, r, CONV, STOP, device=device)
    g.set_stream(stream)
    g.share_images(f)
    g.prepare()
    guess = torch.from_numpy(opencorr_amd.make_pois2d(xs, ys)).to(dev)
    f.compute(guess)
    q = guess.clone()
    avg, n = _timed_launches(torch, g, lambda: (q.copy_(guess), g.compute(q)), 2 * reps)
    res = q.cpu().numpy()
    it = res[:, 17].astype(np.float64)
    ran = it > 0
    n2 = (2 * r + 1) ** 2
    alg = float(ran.sum() * (3 * n2 * 4 + 200) + it[ran].sum() * n2 * 64 + (~ran).sum() * 200)
    tr = kernel_traffic("C", "icgn2d_kernel")
    out.append(secondary_block("icgn2d_kernel<12,...> (ICGN2D2)", "C: 4096^2, r=20 (41x41), %d POIs" % len(xs), alg, avg, n, "l2",
                               L2_PEAK_GBS, "SURVEY 8(d): 3*N2*4 + k*N2*64 + 200 B per POI, N2 = 1681; same L1/L2 table gather as ICGN2D1",
                               {"mean_iterations": float(it[ran].mean()), "converged": int((res[:, 16] >= 0).sum()),
                                "valu_hw": valu_hardware_block(mandated_instr_icgn(res, 17, n2, "2d2", False),
                                                               (tr or {}).get("valu_wave_instr_per_launch"), avg),
                                "arith_fma": fused_leg(torch, g, q, guess, 2 * reps, 17, n2, "2d2", kernel_traffic("C", "icgn2d_kernel", fma=True), avg)},
                               traffic=tr))
    del f, g, ref, tar, guess, q
    # ---- E: 512^3, r = 16, FFTCC3D + ICGN3D1, 37^3 POIs
    r = 16
    ref, tar = synth.speckle_pair_3d(512, 512, 512, seed=20260927, device=dev)
    xs, ys, zs = synth.poi_grid_3d(512, 512, 512, 37, 37, 37, r + 8)
    f = opencorr_amd.FFTCC3D(r, r, r, device=device)
    f.set_stream(stream)
    f.set_images(ref, tar)
    g = opencorr_amd.ICGN3D1(r, r, r, CONV, 20.0, device=device)
    g.set_stream(stream)
    g.share_images(f)
    g.prepare()
    pristine = torch.from_numpy(opencorr_amd.make_pois3d(xs, ys, zs)).to(dev)
    guess = pristine.clone()
    avg_f, n_f = _timed_launches(torch, f, lambda: (guess.copy_(pristine), f.compute(guess)), reps)
    m3 = (2 * r) ** 3
    out.append(secondary_block("fftcc3d_fused32_kernel (FFTCC3D, 32^3 window)", "E: 512^3, r=16, %d POIs" % len(xs),
                               (2 * m3 * 4 + 28) * float(len(xs)), avg_f, n_f, "hbm", HBM_PEAK_GBS,
                               "SURVEY 8(d): 2*M3*4 B in + 28 B out = 262 172 B per POI; the kernel is VALU + LDS-exchange bound "
                               "(six 32-point FFT passes per thread, DESIGN.md 4.3)", traffic=kernel_traffic("E", "fftcc3d_fused32_kernel")))
    q = guess.clone()
    avg, n = _timed_launches(torch, g, lambda: (q.copy_(guess), g.compute(q)), reps)
    res = q.cpu().numpy()
    it = res[:, 19].astype(np.float64)   # POI3D: result.iteration (float 19; zncc is float 18)
    ran = it > 0
    n3 = (2 * r + 1) ** 3
    alg = float(ran.sum() * (4 * n3 * 4 + 248) + it[ran].sum() * n3 * 256 + (~ran).sum() * 248)
    tr = kernel_traffic("E", "icgn3d1")
    out.append(secondary_block("icgn3d1_kernel (ICGN3D1)", "E: 512^3, r=16 (33^3), %d POIs" % len(xs), alg, avg, n, "lds",
                               LDS_READ2_PEAK_GBS, "SURVEY 8(d): 4*N3*4 + k*N3*256 + 248 B per POI; the 256 B per sample and iteration are "
                               "the 64 tricubic taps, served from the LDS-staged coefficient box: judged against the LDS read rate "
                               "(ds_read2_b32: 128 B per clock and CU)",
                               {"mean_iterations": float(it[ran].mean()), "converged": int((res[:, 18] >= 0).sum()),
                                "valu_hw": valu_hardware_block(mandated_instr_icgn(res, 19, n3, "3d1", False),
                                                               (tr or {}).get("valu_wave_instr_per_launch"), avg),
                                "arith_fma": fused_leg(torch, g, q, guess, max(2, reps - 1), 19, n3, "3d1", kernel_traffic("E", "icgn3d1", fma=True), avg)},
                               traffic=tr))
    return out


def other_arith_leg(torch, args, fftcc, icgn, pois, pristine, this_kernel_ms, this_step_ms):
    """The WHOLE step (reset, FFTCC2D, ICGN2D1) of the bench line once more under the other arithmetic contract -- the fused
    build when the line is the default one, and vice versa -- same queue, same steps: what oc_hip_set_tuning("arith_fma") buys
    on the metric, measured in the same process."""
    fma = args.arith != "fma"
    icgn.set_tuning("arith_fma", 1 if fma else 0)
    try:
        def step():
            pois.copy_(pristine)
            fftcc.compute(pois)
            icgn.compute(pois)
        for _ in range(max(2, args.warmup)):
            step()
        torch.cuda.synchronize()
        icgn.profile_reset()
        icgn.profile_enable(True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ms, n = icgn.profile_read()
        icgn.profile_enable(False)
        res = pois.cpu().numpy()
    finally:
        icgn.set_tuning("arith_fma", 1 if args.arith == "fma" else 0)
    conv = int((res[:, 16] >= 0).sum())
    k_ms = ms / max(n, 1)
    prof = pmc_profile(1, fma)
    return {"contract": "fused per-sample multiply-adds (arith_fma = 1, oracle OC_ORDER_LANES_FMA)" if fma else "separately rounded (default build)",
            "value": conv * args.steps / dt, "unit": "POI/s", "ms_per_step": dt / args.steps * 1e3, "icgn_kernel_avg_ms": k_ms,
            "converged_pois": conv, "mean_iterations": float(res[res[:, 17] > 0, 17].astype(np.float64).mean()),
            "kernel_time_ratio_to_the_line": k_ms / this_kernel_ms if this_kernel_ms > 0 else None,
            "step_time_ratio_to_the_line": (dt / args.steps * 1e3) / this_step_ms if this_step_ms > 0 else None,
            "valu_hw": valu_hardware_block(mandated_instr_icgn(res, 17, (2 * RX + 1) * (2 * RY + 1), "2d1", fma),
                                           (prof or {}).get("valu_wave_instr_per_launch"), k_ms),
            "traffic_source": (prof or {}).get("source"),
            "why_not_the_default": "adopted as the default only if every parity bar holds AND ICGN3D1 gains >= 10 % (VERDICT r4 item 1): the "
                                   "bars hold on all five configs, ICGN3D1 gains 5 % -- so the separately rounded build stays the default and "
                                   "the fused one is one oc_hip_set_tuning call away (DESIGN.md section 3)"}


def fused_leg(torch, eng, q, guess, reps, iter_col, samples, kind, traffic, sep_ms):
    """The same launches under the fused arithmetic contract (oc_hip_set_tuning arith_fma = 1): kernel time, speed-up over the
    default build measured a moment ago on the same queue, hardware-rate VALU figures of the fused instruction stream."""
    eng.set_tuning("arith_fma", 1)
    try:
        avg, n = _timed_launches(torch, eng, lambda: (q.copy_(guess), eng.compute(q)), reps)
        res = q.cpu().numpy()
    finally:
        eng.set_tuning("arith_fma", 0)
    return {"avg_launch_ms": avg, "launches_timed": n, "speedup_over_default_build": sep_ms / avg if avg > 0 else None,
            "valu_hw": valu_hardware_block(mandated_instr_icgn(res, iter_col, samples, kind, True),
                                           (traffic or {}).get("valu_wave_instr_per_launch"), avg),
            "traffic_source": (traffic or {}).get("source"),
            "parity": "GPU == oracle OC_ORDER_LANES_FMA bit for bit; against the reference's separately rounded loop order the bars of "
                      "tests/test_gpu_fullsize.py hold (flags equal, >= 99.5 % equal iteration counts, <= 1e-4 px, <= 1e-5 ZNCC)"}


def variant_paths(torch, dev, device, ref, tar, xs, ys, base_ms, base_np, reps=3):
    """SURVEY 8(f) row 1 on config B's grid: `B-OFF` = compute(poi_queue, center_offset_queue) with offsets
    in [-2, 2] px; `B-SA` = setSelfAdaptive(true) with per-POI radii 12 ... 20 (src/oc_icgn.cpp:152-158, 353-557).  Self-adaptive
    queues cannot share a per-workgroup coordinate table nor run lockstep sweeps, so they use the 4-wave table-free variant:
    the cost per sample-iteration next to config B's is what this reports."""
    import opencorr_amd
    stream = torch.cuda.current_stream().cuda_stream
    f = opencorr_amd.FFTCC2D(RX, RY, device=device)
    f.set_stream(stream)
    f.set_images(ref, tar)
    g = opencorr_amd.ICGN2D1(RX, RY, CONV, STOP, device=device)
    g.set_stream(stream)
    g.share_images(f)
    g.prepare()
    guess = torch.from_numpy(opencorr_amd.make_pois2d(xs, ys)).to(dev)
    f.compute(guess)
    q = guess.clone()

    def work(res):   # sample-iterations of a launch: what the interpolation sweeps and numerator passes scale with
        it = res[:, 17].astype(np.float64)
        ran = it > 0
        n2 = (2 * res[ran, 23] + 1) * (2 * res[ran, 24] + 1)
        return float((it[ran] * n2).sum()), float(it[ran].mean())

    base_work, _ = work(base_np)
    out = {}
    rng = np.random.default_rng(20260927)
    off = torch.from_numpy(rng.uniform(-2.0, 2.0, (len(xs), 2)).astype(np.float32)).to(dev)
    avg, n = _timed_launches(torch, g, lambda: (q.copy_(guess), g.compute_with_offsets(q, off)), reps)
    w, k = work(q.cpu().numpy())
    out["B-OFF"] = {"what": "ICGN2D1::compute(poi_queue, center_offset_queue), offsets uniform in [-2, 2] px", "avg_launch_ms": avg,
                    "launches_timed": n, "mean_iterations": k, "ns_per_sample_iteration": avg * 1e6 / w,
                    "relative_to_config_B_per_sample_iteration": (avg / w) / (base_ms / base_work)}
    sa = guess.clone()
    radii = torch.from_numpy(rng.integers(12, 21, (len(xs), 2)).astype(np.float32)).to(dev)
    sa[:, 23:25] = radii
    g.set_self_adaptive(True)
    try:
        avg, n = _timed_launches(torch, g, lambda: (q.copy_(sa), g.compute(q)), reps)
        w, k = work(q.cpu().numpy())
    finally:
        g.set_self_adaptive(False)
    out["B-SA"] = {"what": "ICGN2D1 with setSelfAdaptive(true), per-POI radii uniform in 12 ... 20 (mean subset 33 x 33)",
                   "avg_launch_ms": avg, "launches_timed": n, "mean_iterations": k, "ns_per_sample_iteration": avg * 1e6 / w,
                   "relative_to_config_B_per_sample_iteration": (avg / w) / (base_ms / base_work)}
    out["config_B"] = {"avg_launch_ms": base_ms, "ns_per_sample_iteration": base_ms * 1e6 / base_work}
    return out


def host_queue_rate(fftcc, icgn, pristine, converged, reps=3):
    """The same step when the caller hands over a HOST queue (what the C++ shim's compute(std::vector<POI2D>&)
    does): every compute() then copies the 25 MB AoS to the GPU and back.  Reported beside `value`, never as it."""
    import opencorr_amd
    host0 = pristine.cpu().numpy()
    best, best2 = None, None
    for _ in range(reps + 1):
        q = host0.copy()
        t0 = time.perf_counter()
        opencorr_amd.compute_chain([fftcc, icgn], q)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        q = host0.copy()
        t0 = time.perf_counter()
        fftcc.compute(q)
        icgn.compute(q)
        dt = time.perf_counter() - t0
        best2 = dt if best2 is None else min(best2, dt)
    return {"ms_per_step": best * 1e3, "value": converged / best, "unit": "POI/s",
            "two_calls_ms_per_step": best2 * 1e3, "two_calls_value": converged / best2,
            "note": "SURVEY 8(d)'s definition: pageable host POI queue (what a std::vector<POI2D> is), H2D and D2H of the AoS inside "
                    "the timed region.  `value`: FFTCC2D + ICGN2D1 as ONE chain (oc_hip_compute_chain / computeChain in the C++ "
                    "shim): one copy in, both engines' kernels, one copy out, chunk by chunk.  `two_calls_*`: the reference's "
                    "unmodified call sequence fftcc->compute(q); icgn->compute(q); -- the queue crosses PCIe four times"}


def pmc_profile(world, fma=False):
    """HBM bytes per ICGN launch from the COMMITTED PMC record of this workload (separate rocprofv3 --pmc passes,
    tools/gpu_profiles.sh); a constant read from profiles/, not something measured in this run."""
    if world != 1:
        return None
    cands = (TRAFFIC_BY_CONFIG_FMA["B"],) if fma else (TRAFFIC_BY_CONFIG["B"], TRAFFIC_JSON, TRAFFIC_JSON_OLD)
    path = next((p for p in cands if os.path.exists(p)), None)
    if path is None:
        return None
    return pmc_profile_from(path)


def pmc_profile_from(path, kernel_regex="icgn2d_kernel"):
    with open(path) as f:
        rec = json.load(f)
    ok, why = traffic_record_is_current(rec, kernel_regex)
    if not ok:
        return {"stale": why, "source": os.path.relpath(path, ROOT) + (" (%s)" % rec["collected"] if rec.get("collected") else "")}
    # run-to-run spread of the HBM counters: the same passes collected a second time by the same script
    second = os.path.join(os.path.dirname(path), "traffic_configB_second_run.json")
    spread = None
    if os.path.exists(second):
        with open(second) as f:
            spread = float(json.load(f)["hbm_bytes_per_launch"])
    return {"hbm_bytes_per_launch": (float(rec["hbm_bytes_per_launch"]) if rec.get("hbm_bytes_per_launch") else None),
            "hbm_bytes_per_launch_second_collection": spread,
            "l2_bytes_per_launch": (float(rec["l2_bytes_per_launch"]) if rec.get("l2_bytes_per_launch") else None),
            "l2_request_bytes_calibrated": rec.get("l2_request_bytes"), "l2_hit_rate": rec.get("l2_hit_rate"),
            "valu_wave_instr_per_launch": rec.get("SQ_INSTS_VALU_per_launch"),
            "rocprof_avg_ms": (rec["avg_us"] * 1e-3 if rec.get("avg_us") else None),
            "source": os.path.relpath(path, ROOT) + (" (%s)" % rec["collected"] if rec.get("collected") else ""),
            "note": "PMC counters over `python bench.py --no-cpu-baseline`, one counter set per rocprofv3 run (tools/gpu_profiles.sh): "
                    "HBM = FETCH_SIZE x 2 + WRITE_SIZE; L2 side = TCC_REQ_sum x a request size calibrated on a gather of known byte "
                    "count; compulsory HBM traffic (images + table once) is 1.27 GB"}


def cpu_baseline(ref, tar, xs, ys, sample):
    """The CPU oracle (OC_ORDER_SEQ, i.e. the reference's loop order) on a strided sample of the
    same POI queue, all host cores, best of 3, FFTCC + ICGN compute only (prepare excluded like
    on the GPU side)."""
    import oracle
    build = oracle.use_timing_build()
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
    out = {
        "value": conv / best[0],
        "unit": "POI/s",
        "cores": cores,
        "kind": "port",
        "build": build,
        "sample": "every %d-th POI of the same queue (%d POIs), FFTCC2D+ICGN2D1 compute, best of 3; "
                  "fftcc %.3f s, icgn %.3f s" % (stride, len(sx), best[1], best[2]),
        "icgn_only_value": conv / best[2],
    }
    # OpenCorr's OWN loops beside the port: oracle/_ref/liboc_ref.so = the reference's src/*.cpp compiled
    # unmodified (its float**** table, per-thread instance pool and omp loop, src/oc_icgn.cpp:61-69, 343-351) against the
    # stand-in Eigen of oracle/ref_stubs; prebuilt where the reference tree is mounted (it travels to this box, it cannot
    # be rebuilt here), so -O2 -march=x86-64-v2 rather than this host's native flags.  ICGN leg only: the stand-in FFTW is
    # an O(N^2) DFT, so the reference's FFTCC is not timed; the queue holds the port's FFTCC output.
    try:
        from oracle import ref as oracle_ref
        if oracle_ref.available():
            g = oracle.make_pois2d(sx, sy)
            oracle.fftcc2d(ref_h, tar_h, RX, RY, g, threads=cores)
            t = oracle_ref.time_icgn2d1(ref_h, tar_h, RX, RY, CONV, STOP, g, threads=cores, reps=3)
            if t is not None:
                rconv = int((g[:, 16] >= 0).sum())
                same = bool(np.array_equal(g.view(np.uint32), p.view(np.uint32)))
                out["reference_sources"] = {
                    "icgn_only_value": rconv / t[1], "unit": "POI/s", "cores": cores, "kind": "reference",
                    "icgn_seconds": t[1], "prepare_seconds": t[0],
                    "build": "the reference's src/oc_{icgn,cubic_bspline,gradient,subset,deformation,...}.cpp unmodified, g++ -O2 "
                             "-march=x86-64-v2 -fopenmp -ffp-contract=off (prebuilt: oracle/Makefile `ref`), stand-in Eigen",
                    "sample": "the same %d POIs, ICGN2D1::compute(queue) only (initial guesses = the port's FFTCC output), best of 3" % len(sx),
                    "bit_identical_to_the_port": same,
                    "port_icgn_only_value": conv / best[2]}
    except Exception as exc:   # the baseline leg never takes the bench line down
        out["reference_sources"] = {"error": repr(exc)[:200]}
    return out


def oht_pair(device):
    """The reference's own example (examples/test_2d_dic_fftcc_icgn1.cpp: oht_cfrp_0 / _4.bmp, r = 16, 30 000 POIs,
    conv 1e-3, stop 10) from the committed golden fixture: GPU engines (device-resident queue) and the CPU build above."""
    import torch
    import opencorr_amd
    import oracle
    path = os.path.join(ROOT, "tests", "golden", "oht_cfrp_r16.npz")
    if not os.path.exists(path):
        return None
    g = np.load(path)
    ref, tar, table = g["ref"].astype(np.float32), g["tar"].astype(np.float32), g["table"]
    dev = torch.device("cuda", device)
    pristine = torch.from_numpy(opencorr_amd.make_pois2d(table[:, 0], table[:, 1])).to(dev)
    f = opencorr_amd.FFTCC2D(RX, RY, device=device)
    f.set_images(torch.from_numpy(ref).to(dev), torch.from_numpy(tar).to(dev))
    i = opencorr_amd.ICGN2D1(RX, RY, CONV, STOP, device=device)
    i.share_images(f)
    i.prepare()
    q = pristine.clone()
    best = None
    for _ in range(6):
        q.copy_(pristine)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f.compute(q)
        i.compute(q)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    res = q.cpu().numpy()
    conv = int((res[:, 16] >= 0).sum())
    cores = oracle.max_threads()
    prep = oracle.Prepared2D(ref, tar)
    cbest = None
    for _ in range(3):
        p = opencorr_amd.make_pois2d(table[:, 0], table[:, 1])
        t0 = time.perf_counter()
        oracle.fftcc2d(ref, tar, RX, RY, p, threads=cores)
        oracle.icgn2d1(prep, RX, RY, CONV, STOP, p, order=oracle.ORDER_SEQ, threads=cores)
        dt = time.perf_counter() - t0
        cbest = dt if cbest is None else min(cbest, dt)
    return {"pois": int(len(table)), "converged": conv, "mean_iterations": float(res[res[:, 17] > 0, 17].mean()),
            "gpu_ms": best * 1e3, "gpu_value": conv / best, "cpu_value": int((p[:, 16] >= 0).sum()) / cbest, "cpu_cores": cores,
            "unit": "POI/s", "note": "280 x 900 px image pair of the reference's example: 30 000 POIs keep one MI355X busy "
                                       "for a fraction of a millisecond, launch latencies included"}


if __name__ == "__main__":
    main()
