#!/bin/bash
# Same-box A/B of one handle option: bash tools/opt_ab.sh "three_plane=1" [rounds] [bench args]
O=$1; R=${2:-2}; shift; shift || true
ARGS="--steps 100 --cpu-seconds 0 --host-variants 0 --file-variants 0 --secondary 0 --resident-steps 0 $@"
brief() { python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('$1', d['ms_per_step'], 'ms/step', round(d['value']/1e6,2), 'M/s  kernel', r.get('avg_launch_ms'))"; }
for i in $(seq $R); do
  python3 bench.py $ARGS 2>/dev/null | brief default
  SGX_BENCH_OPTS="$O" python3 bench.py $ARGS 2>/dev/null | brief "$O"
done
