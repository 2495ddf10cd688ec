#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2p; mkdir -p $O
export TMPDIR=/tmp
python __graft_entry__.py smoke > $O/smoke_main.log 2>&1; echo "main rc=$?"
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke_both.log 2>&1; echo "both rc=$?"
tail -n 2 $O/smoke_main.log
timeout 1200 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -n 3 $O/tests.log
