#!/bin/bash
# PTZ-Reloc of the seven WorldCup14 test sequences against their reference matches (the reference's run_reloc_worldcup14.sh).
# The IoU evaluation of the reference (scripts/eval_worldcup.py: OpenCV warps + shapely polygons) is not part of this repository.
set -e
source "$(dirname "$0")/_parallel.sh"
DATA=${DATA:-data/worldcup14}
REF=${REF:-output-worldcup14-offline}
OUT=${OUT:-output-worldcup14-online}
while read ref test; do
  run_on_next_gpu "$BIN/run_ptz_reloc" --ref_images $DATA/offline/$ref --ref_features $DATA/offline_matches/$ref --ref_params $REF/$ref.json \
    --test_images $DATA/online/$test --test_features $DATA/online_matches/$test --output $OUT
done <<PAIRS
GER_ARG ESP_CHI
GER_ARG FRA_GER
GER_POR SUI_FRA
NED_ARG ARG_SUI
NED_ARG BRA_CRO
NED_ARG URU_ENG
USA_GER CRO_MEX
PAIRS
wait_all
