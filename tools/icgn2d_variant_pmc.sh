# PMC counter sets (four passes) + a kernel trace per icgn2d launch shape, via tools/variant_ab.py (GPU box; the A/B library for variants 0, 6, 8, 9).
# usage: bash tools/icgn2d_variant_pmc.sh <tag> "<variants, e.g. 5 9>"   -> gpurun_out/<tag>/   (profiles/r6c_icgn2d_band_vs_variant5_pmc.txt)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; VARS=${2:-"5 9"}
export OPENCORR_HIP_LIB=$ROOT/opencorr_amd/lib/libopencorr_hip.so
export TMPDIR=/tmp
export PYTHONPATH=$ROOT
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp
for v in $VARS; do
  for c in "a:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
           "b:SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS" \
           "c:SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_FLAT SQ_INSTS_VMEM_WR" \
           "d:SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_GDS SQ_INSTS_BRANCH"; do
    name=${c%%:*}; ctr=${c#*:}
    timeout 300 rocprofv3 --pmc $ctr --kernel-include-regex "icgn2d" --output-format csv -d $OUT/v${v}_$name -o p -- python $ROOT/tools/variant_ab.py $v 1 2 > $OUT/v${v}_$name.log 2>&1
  done
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/v${v}_trace -o t -- python $ROOT/tools/variant_ab.py $v 1 2 > $OUT/v${v}_trace.log 2>&1
done
python - <<PY
import csv, glob, collections, os
out = "$OUT"
for v in "$VARS".split():
    tot = collections.defaultdict(float); n = collections.defaultdict(int)
    for p in glob.glob(out + "/v%s_[abcd]/**/*counter_collection.csv" % v, recursive=True):
        disp = set()
        for r in csv.DictReader(open(p)):
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
        for r in csv.DictReader(open(p)):
            n[r["Counter_Name"]] = len(disp)
    print("variant", v, {k: "%.4g" % (tot[k] / max(n[k], 1)) for k in sorted(tot)})
    for p in glob.glob(out + "/v%s_trace/**/*kernel_trace.csv" % v, recursive=True):
        rows = [r for r in csv.DictReader(open(p)) if "icgn2d" in r["Kernel_Name"]]
        if rows:
            r = rows[-1]
            print("  trace:", {k: r[k] for k in r if k in ("LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Workgroup_Size", "Grid_Size")},
                  "ms", sorted((int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e6 for x in rows)[len(rows)//2])
PY
