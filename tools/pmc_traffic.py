#!/usr/bin/env python
"""Turns the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over bench.py into the per-launch HBM
traffic record that bench.py reports as roofline.traffic.

    python tools/pmc_traffic.py <dir with pmc_fetch/ and pmc_write/> <kernel regex> profiles/rNN_icgn2d1_hbm_traffic.json

Units and corrections (/opt/skills/guides/MI355X_MICROARCH.md, section HBM): FETCH_SIZE and
WRITE_SIZE are kilobytes summed over the L2 channels; on gfx950 FETCH_SIZE counts 128-byte fabric
requests as 64 bytes, so it is doubled.  WRITE_SIZE is taken as reported (uncalibrated on gfx950).
Warm-up launches are dropped: only the last `--launches` dispatches of the kernel are averaged.

L2 side (when <dir> also holds pmc_tcc/ and pmc_calib/, tools/gpu_profiles.sh): TCC_REQ_sum of the kernel x the request
size measured on tools/ubench/l2_req_calib (a read-once stream of 2^30 bytes with the same 16-byte-per-lane loads:
bytes / TCC_REQ_sum); FETCH_SIZE of that stream checks the x2 correction of the HBM side on this box.
"""
import datetime
import argparse
import csv
import glob
import json
import os
import re


def per_dispatch(path, counter, rx):
    out = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or not rx.search(r["Kernel_Name"]):
            continue
        d = int(r["Dispatch_Id"])
        out[d] = out.get(d, 0.0) + float(r["Counter_Value"])
    return [out[k] for k in sorted(out)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("kernel")
    ap.add_argument("out")
    ap.add_argument("--launches", type=int, default=5)
    ap.add_argument("--note", default="")
    ap.add_argument("--command", default="", help="the profiled command, recorded in the output")
    ap.add_argument("--kernels", nargs="*", default=[], help="several kernels of one profiled command: one record each under "
                    "`per_kernel` (the top-level fields describe the FIRST one, what bench.py's main block reads)")
    ap.add_argument("--stats", default="", help="kernel-stats csv of the same command (tools/rocpd_summary.py): its average and "
                    "median durations are copied into the per-kernel records")
    a = ap.parse_args()
    if a.kernels:
        recs = {}
        for k in a.kernels:
            recs[k] = one_kernel(a, re.compile(k))
        rec = dict(recs[a.kernels[0]])
        rec["kernel_regex"] = a.kernels[0]
        rec["per_kernel"] = recs
        if a.stats and os.path.exists(a.stats):
            for row in csv.DictReader(open(a.stats)):
                for k in a.kernels:
                    if re.search(k, row["kernel"]) and "avg_us" not in recs[k]:
                        recs[k].update({"kernel_name": row["kernel"][:120], "calls": int(row["calls"]), "avg_us": float(row["avg_us"]),
                                        "median_us": float(row["median_us"]), "vgpr": int(row["vgpr"]), "lds_bytes": int(row["lds_bytes"]),
                                        "scratch_bytes": int(row["scratch_bytes"])})
        with open(a.out, "w") as f:
            json.dump(rec, f, indent=1)
        print(json.dumps({k: {kk: v.get(kk) for kk in ("hbm_bytes_per_launch", "l2_bytes_per_launch", "avg_us")} for k, v in recs.items()}))
        return
    rec = one_kernel(a, re.compile(a.kernel))
    with open(a.out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec))


def one_kernel(a, rx):
    fpath = glob.glob(os.path.join(a.root, "pmc_fetch", "*_counter_collection.csv"))
    wpath = glob.glob(os.path.join(a.root, "pmc_write", "*_counter_collection.csv"))
    rec = {"kernel_regex": rx.pattern, "launches_averaged": a.launches}
    if fpath and wpath:   # (a collection limited to other counter sets -- OC_PROFILE_PASSES -- has no HBM record)
        fetch = per_dispatch(fpath[0], "FETCH_SIZE", rx)[-a.launches:]
        write = per_dispatch(wpath[0], "WRITE_SIZE", rx)[-a.launches:]
        f_kb = sum(fetch) / len(fetch)
        w_kb = sum(write) / len(write)
        rec.update({
            "launches_averaged": len(fetch),
            "FETCH_SIZE_kb_per_launch_raw": f_kb,
            "WRITE_SIZE_kb_per_launch_raw": w_kb,
            "fetch_correction": 2.0,
            "hbm_bytes_per_launch": 2.0 * f_kb * 1024.0 + w_kb * 1024.0,
            "note": a.note or "FETCH_SIZE x2 (gfx950 counts 128 B fabric requests as 64 B); WRITE_SIZE as reported",
        })
    rec["collected"] = datetime.date.today().isoformat()
    # which instruction stream these counters belong to: bench.py drops the record when the kernel's sources have changed since
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from opencorr_amd import build as _build
        rec["source_fingerprint"] = _build.kernel_fingerprint(rx.pattern)
    except Exception as exc:   # (a record without a fingerprint is treated as stale)
        rec["source_fingerprint_error"] = repr(exc)[:200]
    if a.command:
        rec["command"] = a.command
    tcc = glob.glob(os.path.join(a.root, "pmc_tcc", "*_counter_collection.csv"))
    cal = glob.glob(os.path.join(a.root, "pmc_calib", "*_counter_collection.csv"))
    if tcc:
        avg = lambda v: sum(v[-a.launches:]) / max(len(v[-a.launches:]), 1)
        req = avg(per_dispatch(tcc[0], "TCC_REQ_sum", rx))
        hit = avg(per_dispatch(tcc[0], "TCC_HIT_sum", rx))
        miss = avg(per_dispatch(tcc[0], "TCC_MISS_sum", rx))
        rd = avg(per_dispatch(tcc[0], "TCP_TCC_READ_REQ_sum", rx))
        rec.update({"TCC_REQ_per_launch": req, "TCC_HIT_per_launch": hit, "TCC_MISS_per_launch": miss,
                    "TCP_TCC_READ_REQ_per_launch": rd, "l2_hit_rate": hit / (hit + miss) if hit + miss else None})
        if cal:
            crx = re.compile("stream_read")
            creq = per_dispatch(cal[0], "TCC_REQ_sum", crx)
            cbytes = float(1 << 30)
            per_req = cbytes / (sum(creq[1:]) / max(len(creq[1:]), 1))   # first launch dropped (cold)
            rec.update({"l2_request_bytes": per_req, "l2_request_calibration": "tools/ubench/l2_req_calib: 2^30 bytes read once / TCC_REQ_sum",
                        "l2_bytes_per_launch": req * per_req})
            calf = glob.glob(os.path.join(a.root, "pmc_calib_fetch", "*_counter_collection.csv"))
            if calf:
                cf = per_dispatch(calf[0], "FETCH_SIZE", crx)
                rec["fetch_size_kb_of_the_1GiB_stream"] = sum(cf[1:]) / max(len(cf[1:]), 1)
                rec["fetch_correction_measured"] = cbytes / 1024.0 / rec["fetch_size_kb_of_the_1GiB_stream"]
    sq = glob.glob(os.path.join(a.root, "pmc_sq", "*_counter_collection.csv"))
    if sq:
        avg = lambda v: sum(v[-a.launches:]) / max(len(v[-a.launches:]), 1)
        for name in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES",
                     "SQ_ACTIVE_INST_VALU2", "SQ_LDS_BANK_CONFLICT"):
            v = per_dispatch(sq[0], name, rx)
            if v:
                rec[name + "_per_launch"] = avg(v)
    return rec


if __name__ == "__main__":
    main()
