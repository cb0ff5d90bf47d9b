#!/bin/bash
# probe build: the library with -DPTZ_CHOL_TIMELINE (tile stamps of the chain kernel in the replayed graph) -> tools/probes/hip/lib_chain_tl.so
set -e
cd "$(dirname "$0")/../../ptz-calib_amd/csrc"
O=/tmp/ptz_tl_build; mkdir -p $O
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DPTZ_CHOL_TIMELINE"
/opt/rocm/bin/hipcc $F -c ptz_chol.hip -o $O/ptz_chol.o &
/opt/rocm/bin/hipcc $F -c ptz_ba.hip -o $O/ptz_ba.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/probes/hip/lib_chain_tl.so $O/ptz_ba.o $O/ptz_chol.o ptz_krt.o
