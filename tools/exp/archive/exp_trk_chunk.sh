export BDS_LIB_PATH="${BDS_LIB_PATH:-$(cd "$(dirname "${BASH_SOURCE[0]}")" && git rev-parse --show-toplevel 2>/dev/null || echo "$PWD")/bds-3-b1c-b2a-sdr-receiver_amd/libbds_mi355x_hooks.so}"  # the tuning switches exist in the test-hooks build only
for c in 1024 2048 4096 8192; do
  touch bds-3-b1c-b2a-sdr-receiver_amd/csrc/bds_track.hip
  BDS_HIPCC_EXTRA="-DBDS_TRK_CHUNK=$c" ./build.sh 2>&1 | grep -q built || { echo "build failed: $c"; continue; }
  for m in WB B2A; do ep=100; [ $m = B2A ] && ep=1000; echo -n "chunk=$c $m: "; python tools/bench_track.py --mode $m --epochs $ep | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['ms_per_epoch']*1e3,1),'us/epoch')"; done; done
