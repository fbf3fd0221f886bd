#!/bin/bash
# tools/make_profiles.sh only when this box runs the DCGAN-64 generator at the upper end of the pool's spread (boxes differ by ~10 % under
# matrix-core load): sets of different boxes are not comparable, so the committed set should come from one kind of box.
#   tools/profile_if_fast_box.sh r02 [min_images_per_s]
TAG=${1:-r02}
MIN=${2:-515000}
mkdir -p gpurun_out
python3 tools/bench_generators.py 2>/dev/null | head -1 > gpurun_out/box_probe.jsonl
RATE=$(python3 -c "import json; print(int(json.loads(open('gpurun_out/box_probe.jsonl').read())['images_per_s']))")
echo "DCGAN-64: $RATE img/s (threshold $MIN)"
if [ "$RATE" -ge "$MIN" ]; then
  bash tools/make_profiles.sh "$TAG" > gpurun_out/make_profiles.log 2>&1
  tail -2 gpurun_out/make_profiles.log
else
  echo "slow box: nothing done"
fi
