#!/bin/bash
# the retry attempts gated by the pending count (AH_RETRY_GATE): A/B on one box, uniform / clustered / N(0,1) rows, 10M x 768 x 100 trees
for D in 1 4 2; do for G in 0 1 0 1; do AH_RETRY_GATE=$G timeout 300 python scripts/exp_build.py 10000000 100 3 768 $D 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('dist $D gate $G wall %.4f device %.4f margin %.4f after %.4f retries %d' % (d['wall'], d['seconds_device'], d['seconds_margin'], d['seconds_after_device'], d['retries']))
" | tail -2; done; done
