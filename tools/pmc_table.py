#!/usr/bin/env python
"""Condenses the rocprofv3 --pmc CSVs of tools/gpu_sweep.sh into one table: a row per ICGN2D
kernel variant and XCD mapping (the timed dispatch of each: dispatches 2 and 4 of a kernel).

    python tools/pmc_table.py gpurun_out/sweep1 [profiles/r01b_icgn2d_pmc.csv]
"""
import collections
import csv
import glob
import os
import re
import sys


def main(root, out=None):
    per = collections.defaultdict(lambda: collections.defaultdict(dict))  # kernel -> dispatch -> counter -> value
    dur = collections.defaultdict(dict)
    for path in glob.glob(os.path.join(root, "pmc_*", "*_counter_collection.csv")):
        tag = os.path.basename(os.path.dirname(path))
        seq = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            m = re.search(r"icgn2d_kernel<([\d, ]+)>", r["Kernel_Name"])
            if not m:
                continue
            key = tuple(int(x) for x in m.group(1).split(","))
            d = int(r["Dispatch_Id"])
            if d not in seq[key]:
                seq[key].append(d)
            ordinal = seq[key].index(d)
            per[key][ordinal].setdefault(r["Counter_Name"], 0.0)
            per[key][ordinal][r["Counter_Name"]] += float(r["Counter_Value"])
            dur[key][(tag, ordinal)] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    counters = sorted({c for k in per for o in per[k] for c in per[k][o]})
    rows = []
    for key in sorted(per):
        # every (variant, mapping) is dispatched twice by tests/fullsize/icgn_sweep.py --launches 1: warm-up, then the timed one
        for n, ordinal in enumerate(sorted(per[key])[1::2]):
            c = per[key][ordinal]
            ms = [v for (t, o), v in dur[key].items() if o == ordinal]
            row = dict(dof=key[0], G=key[1], mode=key[2], pipe=key[3], wpb=key[4], occ=key[5], xcd=n,
                       ms_under_pmc=sum(ms) / len(ms))
            row.update({n: c.get(n, float("nan")) for n in counters})
            rows.append(row)
    if out:
        with open(out, "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows)
    for r in rows:
        wc = r.get("SQ_WAVE_CYCLES", float("nan"))
        l2 = r.get("TCC_HIT_sum", 0) / max(r.get("TCC_HIT_sum", 0) + r.get("TCC_MISS_sum", 0), 1)
        print("G%d m%d p%d w%d o%d x%d | %6.2f ms | valu_inst %.3g act_valu/wave_cyc %.2f wait_any %.2f wait_inst %.2f | "
              "vmem_rd %.3g lds %.3g | L2 hit %.3f req %.3g | tcp->tcc rd %.3g tcp acc %.3g | fetch %.3g MB write %.3g MB | "
              "ta_busy %.3g tcp_pend %.3g ta_stall %.3g | bank_conf %.3g lds_act %.3g"
              % (r["G"], r["mode"], r["pipe"], r["wpb"], r["occ"], r["xcd"], r["ms_under_pmc"],
                 r.get("SQ_INSTS_VALU", 0), r.get("SQ_ACTIVE_INST_VALU", 0) / wc, r.get("SQ_WAIT_ANY", 0) / wc,
                 r.get("SQ_WAIT_INST_ANY", 0) / wc, r.get("SQ_INSTS_VMEM_RD", 0), r.get("SQ_INSTS_LDS", 0), l2,
                 r.get("TCC_REQ_sum", 0), r.get("TCP_TCC_READ_REQ_sum", 0), r.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0),
                 r.get("FETCH_SIZE", 0) / 1024, r.get("WRITE_SIZE", 0) / 1024, r.get("TA_TA_BUSY_sum", 0),
                 r.get("TCP_PENDING_STALL_CYCLES_sum", 0), r.get("TCP_TCP_TA_DATA_STALL_CYCLES_sum", 0),
                 r.get("SQ_LDS_BANK_CONFLICT", 0), r.get("SQ_LDS_IDX_ACTIVE", 0)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
