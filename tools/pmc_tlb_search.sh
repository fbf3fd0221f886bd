#!/bin/bash
# Address-translation counters of the persistent l2-lpips search at 64 x 64 (1 MB rows) and 256 x 256 (16 MB rows): is the lower rate at 256 x 256
# (0.49 per busy cluster against 0.58) a TLB effect?      bash tools/pmc_tlb_search.sh   (on the GPU box; writes gpurun_out/pmc_tlb_*)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
C="TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"
rocprofv3 --pmc $C --output-format csv -d $ROOT/gpurun_out/pmc_tlb_64 -- python3 $ROOT/tools/bench_pairwise.py --feat --variants 3 --rounds 1 --res 64 --queries 4096 --feat-bank 8192 > /dev/null
rocprofv3 --pmc $C --output-format csv -d $ROOT/gpurun_out/pmc_tlb_256 -- python3 $ROOT/tools/bench_pairwise.py --feat --variants 3 --rounds 1 --res 256 --queries 4096 --feat-bank 3072 > /dev/null
cd $ROOT
python3 - <<'PY'
import csv, glob, json
for tag in ("64", "256"):
    tot = {}
    n = 0
    for f in glob.glob("gpurun_out/pmc_tlb_%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if "feat_knn_h1c" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                n += 1
    launches = max(1, n // max(1, len(tot)))
    print(json.dumps({"res": tag, "launches": launches, **{k: v / launches for k, v in tot.items()},
                      "utcl1_miss_rate": tot.get("TCP_UTCL1_TRANSLATION_MISS_sum", 0) / max(1.0, tot.get("TCP_UTCL1_REQUEST_sum", 1.0)),
                      "utcl2_busy_frac": tot.get("GRBM_UTCL2_BUSY", 0) / max(1.0, tot.get("GRBM_GUI_ACTIVE", 1.0))}))
PY
