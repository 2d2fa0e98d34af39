#!/bin/bash
# round 5, call 28: the whole GPU suite and smoke() on the round's last library
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 3000 python -m pytest tests -q -m gpu --maxfail=10 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
