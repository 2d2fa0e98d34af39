#!/bin/bash
# round 5, call 24: the whole GPU suite on the library with the small submissions' path, smoke(), the randomised sweep
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 3400 python -m pytest tests -q -m gpu --maxfail=10 2>&1 | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
bash scripts/r05_fuzz.sh 2>&1 | tail -6
