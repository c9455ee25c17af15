#!/bin/bash
# tracking epoch time vs channel count: flat = latency-bound, linear = throughput-bound
export BDS_LIB_PATH="${BDS_LIB_PATH:-$(cd "$(dirname "${BASH_SOURCE[0]}")" && git rev-parse --show-toplevel 2>/dev/null || echo "$PWD")/bds-3-b1c-b2a-sdr-receiver_amd/libbds_mi355x_hooks.so}"  # the tuning switches exist in the test-hooks build only
cd "$GRAFT_REPO_ROOT"
for M in ${MODES:-WB B2A}; do
for env in "BDS_TRK_PERSAMPLE=1" "BDS_X=0"; do
for c in 1 3 6 12 24 48; do
  E=200; [ $M = B2A ] && E=1000
  env $env timeout 300 python tools/bench_track.py --mode $M --epochs $E --channels $c 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$M $env channels', j['channels'], 'us/epoch', round(j['ms_per_epoch']*1e3,2), 'GB/s', round(j['int8_read_GBps'],1))
"
done; done; done
