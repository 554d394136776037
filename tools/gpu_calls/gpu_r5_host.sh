#!/bin/bash
# round 5: where the host time of one submitted batch goes (cProfile of forward_async / result on C2, default mode), and the bench's own host_enqueue figure
cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r05h}
mkdir -p $O
timeout 300 python tools/host_profile.py > $O/host_profile.txt 2>&1; grep -A55 "Ordered by: internal" $O/host_profile.txt | cut -c1-180
timeout 300 python tools/host_overhead.py > $O/host_overhead.txt 2>&1; tail -15 $O/host_overhead.txt | cut -c1-250
