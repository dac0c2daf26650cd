#!/bin/bash
# round 5, call 1: price the all-gather primitive; a same-box baseline of the headline
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c1; mkdir -p $O; cd $R
hipcc --offload-arch=gfx950 -O3 -Wno-unused-result tools/cap_allgather_probe.hip -o /tmp/cap_ag_probe 2>/dev/null
timeout 120 /tmp/cap_ag_probe > $O/cap_allgather_probe.txt 2>&1
cat $O/cap_allgather_probe.txt


