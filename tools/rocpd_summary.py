#!/usr/bin/env python
"""Turns a rocprofv3 rocpd SQLite database (`rocprofv3 --kernel-trace --stats`) into the
per-kernel summary committed under profiles/ (name, calls, total/avg/min/max duration in us, %).

VGPR columns: rocprofv3's `vgpr_count` for a wave64 kernel on gfx950 is HALF the registers a lane holds
(a 76-VGPR kernel, allocated in granules of 8 = 80, is reported as 40; 128 as 64 -- checked against the code objects'
.vgpr_count with tools/kernel_resources.py).  `vgpr` keeps rocprofv3's number (files of rounds 1 - 4 hold only that one);
`vgpr_per_lane` = 2 x that = the allocated architected VGPRs per lane, the number occupancy is computed from.

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db profiles/r01_x_kernel_stats.csv
"""
import csv
import sqlite3
import sys


def main(db_path, out_path, top=40):
    db = sqlite3.connect(db_path)
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) from kernels group by name "
        "order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    # median per kernel: robust against the cold first launch when comparing with bench.py's timed launches
    median = {}
    for name, dur in db.execute("select name, duration from kernels"):
        median.setdefault(name, []).append(dur)
    median = {k: sorted(v)[len(v) // 2] for k, v in median.items()}
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "median_us", "min_us", "max_us", "percent", "vgpr", "sgpr", "lds_bytes",
                    "scratch_bytes", "vgpr_per_lane"])
        # the top kernels by time, plus every kernel of this library however small (torch's synthetic-data kernels can crowd
        # them out of the top rows)
        for r in [r for i, r in enumerate(rows) if i < top or "ochip::" in r[0]]:
            name = r[0] if len(r[0]) < 160 else r[0][:157] + "..."
            w.writerow([name, r[1], "%.3f" % (r[2] / 1e3), "%.3f" % (r[3] / 1e3), "%.3f" % (median[r[0]] / 1e3), "%.3f" % (r[4] / 1e3),
                        "%.3f" % (r[5] / 1e3), "%.2f" % (100.0 * r[2] / total), r[6], r[7], r[8], r[9], 2 * (r[6] or 0)])
    for r in rows[:12]:
        print("%-90s calls %5d  avg %10.3f us  median %10.3f us  %5.1f%%" % (r[0][:90], r[1], r[3] / 1e3, median[r[0]] / 1e3,
                                                                              100.0 * r[2] / total))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
