#!/bin/bash
# Address-translation counters per kernel family over the headline step, the l2-lpips step and the generator table (after the 256 x 256 search
# turned out to be TLB-bound, round 3): which other kernels miss in the UTCL1 / keep the UTCL2 busy?   bash tools/pmc_tlb_all.sh   (GPU box)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
C="TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"
ONE="python3 $ROOT/bench.py --secondary off --steps 1 --warmup 0 --cpu-queries 0 --check-queries 0"
rocprofv3 --pmc $C --output-format csv -d $ROOT/gpurun_out/pmc_tlb_l2 -- $ONE > /dev/null
rocprofv3 --pmc $C --output-format csv -d $ROOT/gpurun_out/pmc_tlb_lp -- $ONE --distance l2-lpips > /dev/null
rocprofv3 --pmc $C --output-format csv -d $ROOT/gpurun_out/pmc_tlb_gen -- python3 $ROOT/tools/bench_generators.py > /dev/null
cd $ROOT
python3 - <<'PY'
import csv, glob, json, re
for tag in ("l2", "lp", "gen"):
    fam = {}
    for f in glob.glob("gpurun_out/pmc_tlb_%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))[:70]
            d = fam.setdefault(name, {})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for name, d in sorted(fam.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
        if d.get("GRBM_GUI_ACTIVE", 0) < 1e6:
            continue
        print(json.dumps({"run": tag, "kernel": name, "gui_active": d.get("GRBM_GUI_ACTIVE"),
                          "utcl1_miss_rate": round(d.get("TCP_UTCL1_TRANSLATION_MISS_sum", 0) / max(1.0, d.get("TCP_UTCL1_REQUEST_sum", 1.0)), 6),
                          "utcl2_busy_frac": round(d.get("GRBM_UTCL2_BUSY", 0) / max(1.0, d.get("GRBM_GUI_ACTIVE", 1.0)), 4)}))
PY
