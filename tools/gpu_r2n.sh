#!/bin/bash
# PMC passes over the final round-2 kernels: ICGN2D1 (config B, variant 2) and ICGN3D1 (config Es)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02n}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
bash $ROOT/tools/gpu_pmc.sh $TAG/pmc2d 2
python $ROOT/tools/pmc_table.py $OUT/pmc2d $OUT/icgn2d1_pmc_table.csv 2>&1 | tail -3
bash $ROOT/tools/gpu_pmc3d.sh $TAG/pmc3d Es
python - <<'PY'
import csv, glob, json, os, sys
out = os.environ.get("OUT3D", "")
root = sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "%s" % sys.argv[1] if len(sys.argv) > 1 else "r02n", "pmc3d", "pmc_*", "*_counter_collection.csv")))
tot = {}
for path in root:
    per = {}
    for r in csv.DictReader(open(path)):
        if "icgn3d1_kernel" not in r["Kernel_Name"]:
            continue
        per.setdefault(r["Dispatch_Id"], {}).setdefault(r["Counter_Name"], 0.0)
        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    if per:
        last = per[sorted(per, key=int)[-1]]
        tot.update(last)
print(json.dumps(tot)); open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r02n", "icgn3d1_pmc_Es.json"), "w").write(json.dumps(tot, indent=1))
PY
