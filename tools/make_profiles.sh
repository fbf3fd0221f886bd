#!/bin/bash
# Regenerates the measurement set of profiles/rNN on an MI355X box (run from the repository root, e.g. through gpurun):
#   tools/make_profiles.sh r02
# bench line, rocprofv3 kernel statistics of the same command, PMC passes (FETCH_SIZE / WRITE_SIZE / SQ, each in its own run and never
# together with a trace), the config-3 (l2-lpips) line, the generator table and the end-to-end parity check against the CPU replay.
set -eo pipefail
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT" "$ROOT/profiles/$TAG"
cd /tmp && export TMPDIR=/tmp
ONE="python3 $ROOT/bench.py --steps 1 --warmup 0 --cpu-queries 0 --check-queries 0"
python3 "$ROOT/bench.py" > "$OUT/bench.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --cpu-queries 0 > "$OUT/bench_under_rocprof.json"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $ONE > /dev/null
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $ONE > /dev/null
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
          --output-format csv -d "$OUT/pmc_sq" -- $ONE > /dev/null
cd "$ROOT"
python3 bench.py --distance l2-lpips --steps 2 --warmup 1 --cpu-queries 0 > "$OUT/bench_config3_l2lpips.json"
python3 tools/bench_generators.py > "$OUT/generators.jsonl"
python3 tools/auroc_delta_full.py > "$OUT/auroc_delta_full_config2.json"
cp "$OUT/bench.json" "profiles/$TAG/${TAG}_bench.json"
cp "$OUT/bench_under_rocprof.json" "profiles/$TAG/${TAG}_bench_under_rocprof.json"
cp "$OUT"/trace/*/*_kernel_stats.csv "profiles/$TAG/${TAG}_bench_kernel_stats.csv"
cp "$OUT/bench_config3_l2lpips.json" "profiles/$TAG/${TAG}_bench_config3_l2lpips_10kx100k.json"
cp "$OUT/generators.jsonl" "profiles/$TAG/${TAG}_generators.jsonl"
cp "$OUT/auroc_delta_full_config2.json" "profiles/$TAG/${TAG}_auroc_delta_full_config2.json"
python3 tools/pmc_summary.py "profiles/$TAG/pmc_traffic_default.json" "$OUT/pmc_fetch" "$OUT/pmc_write" > /dev/null
python3 tools/pmc_summary.py "profiles/$TAG/${TAG}_pmc_sq_default.json" "$OUT/pmc_sq" > /dev/null
echo "profiles/$TAG refreshed (bench.py takes roofline.traffic from the newest profiles/r*/pmc_traffic_default.json)"
