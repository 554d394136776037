#!/bin/bash
# round 4, call j: LDS / MFMA microbenchmark (tools/lds_mfma_bench.hip): what one CU delivers to the fragment loop of the resident-weights 3x3 kernels
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04j
timeout 120 tools/_bin/lds_mfma_bench 2>&1 | tee gpurun_out/r04j/lds_mfma_bench.txt
