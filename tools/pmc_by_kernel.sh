#!/bin/bash
# Per kernel INSTANCE (template arguments kept) counters of one headline step: matrix-pipe occupancy, wait breakdown, L2 hit rate.
#   bash tools/pmc_by_kernel.sh [extra bench flags]      (GPU box; prints JSON lines)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
ONE="python3 $ROOT/bench.py --secondary off --live-traffic off --steps 1 --warmup 0 --cpu-queries 0 --check-queries 0 $*"
rm -rf $ROOT/gpurun_out/pmc_k1 $ROOT/gpurun_out/pmc_k2
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $ROOT/gpurun_out/pmc_k1 -- $ONE > /dev/null
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --output-format csv -d $ROOT/gpurun_out/pmc_k2 -- $ONE > /dev/null || true
cd $ROOT
python3 - <<'PY'
import csv, glob, json, re
fam = {}
for tag in ("k1", "k2"):
    for f in glob.glob("gpurun_out/pmc_%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(GlGatherConv.*|\(signed char.*|\(char const.*|\(float const.*|\(unsigned char.*", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))[:80]
            d = fam.setdefault(name, {})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                d["_n"] = d.get("_n", 0) + 1
for name, d in sorted(fam.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    g = d.get("GRBM_GUI_ACTIVE", 0)
    if g < 5e6:
        continue
    wc = max(1.0, d.get("SQ_WAVE_CYCLES", 1.0))
    out = {"kernel": name, "launches": d.get("_n"), "cycles_per_launch_per_xcd": round(g / 8 / max(1, d.get("_n", 1))),
           "mfma_busy": round(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * g / 8), 3),
           "wait_any": round(d.get("SQ_WAIT_ANY", 0) / wc, 3), "wait_inst_any": round(d.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
           "wait_inst_lds": round(d.get("SQ_WAIT_INST_LDS", 0) / wc, 3), "active_inst": round(d.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3)}
    if "TCC_HIT_sum" in d:
        out["l2_hit_rate"] = round(d["TCC_HIT_sum"] / max(1.0, d["TCC_HIT_sum"] + d.get("TCC_MISS_sum", 0)), 4)
    for k in ("SQ_LDS_IDX_ACTIVE", "SQ_INST_CYCLES_VMEM", "SQ_ACTIVE_INST_LDS"):
        if k in d:
            out[k.lower() + "_per_wave_cycle"] = round(d[k] / wc, 4)
    print(json.dumps(out))
PY
