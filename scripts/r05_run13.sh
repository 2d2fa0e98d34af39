#!/bin/bash
# round 5, call 13: the whole GPU suite on the final library, smoke()
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 3400 python -m pytest tests -q -m gpu --maxfail=10 2>&1 | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
