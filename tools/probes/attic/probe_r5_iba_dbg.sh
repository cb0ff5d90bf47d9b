cd $GRAFT_REPO_ROOT
PTZ_BA_DEBUG_TIMING=1 timeout 600 python tools/probes/probe_iba_batch.py 64 200 2> /tmp/iba_dbg.txt | grep -E "rigs" | tail -1 | cut -c1-200
grep "ptz_ba_create" /tmp/iba_dbg.txt | tail -200 | awk 'NR%20==1' | cut -c1-250
grep -v "ptz_ba_create" /tmp/iba_dbg.txt | grep -i "solve\|pass\|enq" | tail -150 | awk 'NR%15==1' | cut -c1-300
