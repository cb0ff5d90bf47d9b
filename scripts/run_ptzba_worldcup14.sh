#!/bin/bash
# PTZ-IBA + georeferencing on the four WorldCup14 matches (the reference's run_ptzba_worldcup14.sh); at most four GPUs are busy.
set -e
source "$(dirname "$0")/_parallel.sh"
DATA=${DATA:-data/worldcup14}
OUT=${OUT:-output-worldcup14-offline}
for m in GER_ARG GER_POR NED_ARG USA_GER; do
  run_on_next_gpu "$BIN/run_ptz_ba" -i $DATA/offline/$m -f $DATA/offline_matches/$m -a $DATA/offline/$m/$m.json -o $OUT
done
wait_all
