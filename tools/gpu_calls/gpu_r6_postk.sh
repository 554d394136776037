#!/bin/bash
# round 6: post-process kernel durations (rocprofv3 kernel stats of tools/profile_serial.py --config c2) for the shipped library and tools/_bin/libyolort_amd_<alt>.so
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r06postk}
O=gpurun_out/$TAG
mkdir -p $O
for alt in shipped $ALTS; do
  if [ $alt = shipped ]; then unset YOLORT_AMD_LIB; else export YOLORT_AMD_LIB=$PWD/tools/_bin/libyolort_amd_$alt.so; fi
  (cd /tmp && rm -rf /tmp/prof_pk && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_pk -o r -- python $GRAFT_REPO_ROOT/tools/profile_serial.py --config c2 --steps 8 > /tmp/ps_pk.log 2>&1)
  echo "== $alt" >> $O/postk.txt
  python tools/rocprof_summary.py $(find /tmp/prof_pk -name "*.db" | head -1) 2>/dev/null | grep -i "rank_image\|scatter_ranks\|sort_image\|nms_segments\|select_prefix\|find_segments\|gather_topk\|post_reset" | python -c "
import sys,re
for l in sys.stdin:
    m=re.match(r'\"(?:void )?ymi::(\w+).*\",(\d+),([\d.]+),([\d.]+),([\d.]+)', l)
    if m: print('  %-24s calls %s avg_us %s' % (m.group(1), m.group(2), m.group(4)))
" >> $O/postk.txt
done
cat $O/postk.txt
