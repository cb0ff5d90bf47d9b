#!/bin/bash
# PTZ-Reloc of the online images of the ten synthetic scenes against the offline results (the reference's run_reloc_synthetic.sh).
set -e
source "$(dirname "$0")/_parallel.sh"
DATA=${DATA:-data/synthetic}
REF=${REF:-output-synthetic-offline}
OUT=${OUT:-output-synthetic-online}
for s in 01 02 03 04 05 06 07 08 09 10; do
  run_on_next_gpu "$BIN/run_ptz_reloc" --ref_images $DATA/offline/scene_$s --ref_features $DATA/offline_matches/scene_$s --ref_params $REF/scene_$s.json \
    --test_images $DATA/online/scene_$s --test_features $DATA/online_matches/scene_$s --output $OUT
done
wait_all
for s in 01 02 03 04 05 06 07 08 09 10; do
  python3 "$(dirname "$0")/../tools/eval_synthetic.py" --pred $OUT/scene_$s.json --gt $DATA/gt/scene_$s.json
done
