#!/bin/bash
# round 5, call 45: the whole GPU suite twice on the final code (flakiness check) + smoke
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for i in 1; do timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -3; done
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
