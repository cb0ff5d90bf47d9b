#!/bin/bash
# PTZ-IBA + georeferencing on the ten synthetic scenes, then the accuracy report (the reference's run_ptzba_synthetic.sh).
set -e
source "$(dirname "$0")/_parallel.sh"
DATA=${DATA:-data/synthetic}
OUT=${OUT:-output-synthetic-offline}
for s in 01 02 03 04 05 06 07 08 09 10; do
  run_on_next_gpu "$BIN/run_ptz_ba" -i $DATA/offline/scene_$s -f $DATA/offline_matches/scene_$s -a $DATA/offline/scene_$s/scene_$s.json -o $OUT
done
wait_all
for s in 01 02 03 04 05 06 07 08 09 10; do
  python3 "$(dirname "$0")/../tools/eval_synthetic.py" --pred $OUT/scene_$s.json --gt $DATA/gt/scene_$s.json
done
