#!/bin/bash
# round 5, call 33: the whole GPU suite on the final kernels + smoke
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
