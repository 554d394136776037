#!/bin/bash
# round 6: the whole GPU suite + smoke
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r06n}
O=gpurun_out/$TAG
mkdir -p $O
date
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $O/pytest_all.log 2>&1
echo "suite rc $?"; grep -v "^  File" $O/pytest_all.log | tail -40 | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
date
