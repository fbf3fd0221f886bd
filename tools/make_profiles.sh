#!/bin/bash
# Regenerates the measurement set of profiles/rNN on an MI355X box (run from the repository root, e.g. through gpurun):
#   tools/make_profiles.sh r02 [quick]
# bench line (headline + configs[2] + fp32 generator), rocprofv3 kernel statistics of the headline and of the l2-lpips workload, PMC passes
# (FETCH_SIZE / WRITE_SIZE / SQ, each in its own run and never together with a trace) for the headline, the l2-lpips and the fp32-generator
# workloads, the generator table, the one-rank shares of configs[3] / configs[4] and the end-to-end parity check against the CPU replay
# (the last two are skipped with `quick`).
set -eo pipefail
TAG=${1:-r02}
QUICK=${2:-}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles_$TAG
# the finished set goes to $OUT/final (gpurun brings back gpurun_out/ only): copy it into profiles/$TAG/ afterwards
DEST=$OUT/final
mkdir -p "$OUT" "$DEST"
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --secondary off"
ONE="$B --steps 1 --warmup 0 --cpu-queries 0 --check-queries 0"
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
echo "[profiles] bench (headline + secondaries)"; python3 "$ROOT/bench.py" > "$OUT/bench.json"
echo "[profiles] kernel trace, headline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $B --cpu-queries 0 > "$OUT/bench_under_rocprof.json"
rm -f "$OUT"/trace/*/*kernel_trace.csv
echo "[profiles] PMC, headline"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $ONE > /dev/null
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $ONE > /dev/null
rocprofv3 --pmc $SQ --output-format csv -d "$OUT/pmc_sq" -- $ONE > /dev/null
echo "[profiles] kernel trace + PMC, l2-lpips (configs[2])"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_lp" -- $B --distance l2-lpips --steps 2 --warmup 1 --cpu-queries 0 > "$OUT/bench_l2lpips_under_rocprof.json"
rm -f "$OUT"/trace_lp/*/*kernel_trace.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_lp" -- $ONE --distance l2-lpips > /dev/null
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_lp" -- $ONE --distance l2-lpips > /dev/null
rocprofv3 --pmc $SQ --output-format csv -d "$OUT/pmc_sq_lp" -- $ONE --distance l2-lpips > /dev/null
echo "[profiles] PMC, fp32 generator"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_f32" -- $ONE --gen-precision 0 > /dev/null
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_f32" -- $ONE --gen-precision 0 > /dev/null
cd "$ROOT"
echo "[profiles] generators"; python3 tools/bench_generators.py > "$OUT/generators.jsonl"
echo "[profiles] pairwise kernels A/B"; python3 tools/bench_pairwise.py --variants 0,1 --rounds 5 > "$OUT/pairwise_ab.jsonl"
if [ -z "$QUICK" ]; then
  echo "[profiles] one-rank shares of configs[3] / configs[4]"; python3 tools/bench_shard_configs.py > "$OUT/shard_configs.jsonl"
  echo "[profiles] end-to-end parity at configs[1] size"; python3 tools/auroc_delta_full.py > "$OUT/auroc_delta_full_config2.json"
  cp "$OUT/shard_configs.jsonl" "$DEST/${TAG}_shard_configs.jsonl"
  cp "$OUT/auroc_delta_full_config2.json" "$DEST/${TAG}_auroc_delta_full_config2.json"
fi
cp "$OUT/bench.json" "$DEST/${TAG}_bench.json"
cp "$OUT/bench_under_rocprof.json" "$DEST/${TAG}_bench_under_rocprof.json"
cp "$OUT"/trace/*/*_kernel_stats.csv "$DEST/${TAG}_bench_kernel_stats.csv"
cp "$OUT/bench_l2lpips_under_rocprof.json" "$DEST/${TAG}_bench_l2lpips_under_rocprof.json"
cp "$OUT"/trace_lp/*/*_kernel_stats.csv "$DEST/${TAG}_bench_l2lpips_kernel_stats.csv"
cp "$OUT/generators.jsonl" "$DEST/${TAG}_generators.jsonl"
cp "$OUT/pairwise_ab.jsonl" "$DEST/${TAG}_pairwise_ab.jsonl"
python3 tools/pmc_summary.py "$DEST/pmc_traffic_default.json" "$OUT/pmc_fetch" "$OUT/pmc_write" > /dev/null
python3 tools/pmc_summary.py "$DEST/${TAG}_pmc_sq_default.json" "$OUT/pmc_sq" > /dev/null
python3 tools/pmc_summary.py "$DEST/pmc_traffic_l2lpips.json" "$OUT/pmc_fetch_lp" "$OUT/pmc_write_lp" > /dev/null
python3 tools/pmc_summary.py "$DEST/${TAG}_pmc_sq_l2lpips.json" "$OUT/pmc_sq_lp" > /dev/null
python3 tools/pmc_summary.py "$DEST/pmc_traffic_fp32.json" "$OUT/pmc_fetch_f32" "$OUT/pmc_write_f32" > /dev/null
echo "$DEST written: cp gpurun_out/profiles_$TAG/final/* profiles/$TAG/ (bench.py takes roofline.traffic from the newest profiles/r*/pmc_traffic_*.json)"
