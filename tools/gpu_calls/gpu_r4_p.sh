#!/bin/bash
# round 4, call p: the C5 at-spec test's label-sequence disagreement with the oracle after the re-tune (different logits): dump the case for a CPU look
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04p
timeout 600 python tools/dump_c5_case.py gpurun_out/r04p 2>&1 | grep -v amdgpu.ids | tail -5
