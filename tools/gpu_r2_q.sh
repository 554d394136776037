#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
NBLK=448 YOLORT_AMD_LIB=$PWD/tools/_bin/libyolort_amd_h8stamps.so timeout 300 python tools/stamp_h8.py 32,128,128,40,40,93 2>&1 | grep -E "shader clock|==|CUs used"
NBLK=2048 YOLORT_AMD_LIB=$PWD/tools/_bin/libyolort_amd_h8stamps.so timeout 300 python tools/stamp_h8.py 32,32,32,160,160,94 2>&1 | grep -E "shader clock|==|CUs used"
