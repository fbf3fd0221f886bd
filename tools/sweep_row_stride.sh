#!/bin/bash
# Row-stride experiment for the persistent l2-lpips search at 256 x 256 (8.2 M halves per row, 16 MB): the same K slice of the 512 rows of a tile
# is read together, so the row stride decides how those 128-byte pieces spread over the memory channels.  The search row already carries 64 extra
# halves (lp_search_pad); this adds more.   bash tools/sweep_row_stride.sh > gpurun_out/row_stride.jsonl
set -e
cd "$(dirname "$0")/.."
for pad in 0 64 192 448 960 1984 4032 8128; do
  echo "{\"extra_pad_halves\": $pad}"
  python tools/bench_pairwise.py --feat --variants 3 --rounds 3 --res 256 --queries 4096 --feat-bank 3072 --pad $pad 2>/dev/null | grep median
done
