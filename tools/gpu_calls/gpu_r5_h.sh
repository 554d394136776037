#!/bin/bash
# round 5, call H: the submit's host time split by C entry point
cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r05h3}
mkdir -p $O
timeout 300 python tools/host_overhead.py > $O/host_overhead.txt 2>&1; tail -6 $O/host_overhead.txt
