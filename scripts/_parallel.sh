# Sourced by the run_* scripts: the reference runs its scenes / matches one after the other on the CPU; they are independent,
# so here every job is handed to the next GPU of the node (one process per GPU, HIP_VISIBLE_DEVICES), NGPU at a time.
NGPU=${NGPU:-$(python3 -c "import torch; print(max(torch.cuda.device_count(), 1))" 2>/dev/null || echo 1)}
BIN=${BIN:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)/ptz-calib_amd/bin}
_slot=0
_pids=()
run_on_next_gpu() {
  HIP_VISIBLE_DEVICES=$((_slot % NGPU)) "$@" &
  _pids+=($!)
  _slot=$((_slot + 1))
  if [ $((_slot % NGPU)) -eq 0 ]; then wait_all; fi
}
wait_all() {
  local rc=0
  for p in "${_pids[@]}"; do wait "$p" || rc=$?; done
  _pids=()
  return $rc
}
