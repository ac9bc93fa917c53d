#!/usr/bin/env python
"""Turns the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over bench.py into the per-launch HBM
traffic record that bench.py reports as roofline.traffic.

    python tools/pmc_traffic.py <dir with pmc_fetch/ and pmc_write/> <kernel regex> profiles/rNN_icgn2d1_hbm_traffic.json

Units and corrections (/opt/skills/guides/MI355X_MICROARCH.md, section HBM): FETCH_SIZE and
WRITE_SIZE are kilobytes summed over the L2 channels; on gfx950 FETCH_SIZE counts 128-byte fabric
requests as 64 bytes, so it is doubled.  WRITE_SIZE is taken as reported (uncalibrated on gfx950).
Warm-up launches are dropped: only the last `--launches` dispatches of the kernel are averaged.
"""
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
    a = ap.parse_args()
    rx = re.compile(a.kernel)
    fetch = per_dispatch(glob.glob(os.path.join(a.root, "pmc_fetch", "*_counter_collection.csv"))[0], "FETCH_SIZE", rx)
    write = per_dispatch(glob.glob(os.path.join(a.root, "pmc_write", "*_counter_collection.csv"))[0], "WRITE_SIZE", rx)
    fetch, write = fetch[-a.launches:], write[-a.launches:]
    f_kb = sum(fetch) / len(fetch)
    w_kb = sum(write) / len(write)
    rec = {
        "kernel_regex": a.kernel,
        "launches_averaged": len(fetch),
        "FETCH_SIZE_kb_per_launch_raw": f_kb,
        "WRITE_SIZE_kb_per_launch_raw": w_kb,
        "fetch_correction": 2.0,
        "hbm_bytes_per_launch": 2.0 * f_kb * 1024.0 + w_kb * 1024.0,
        "note": a.note or "FETCH_SIZE x2 (gfx950 counts 128 B fabric requests as 64 B); WRITE_SIZE as reported",
    }
    with open(a.out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
