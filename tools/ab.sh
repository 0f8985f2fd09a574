#!/bin/bash
# Same-box A/B of two builds of the library, alternating, the bench's step only.
#   bash tools/ab.sh [rounds=2] [extra bench args]      A = $AB_A (default tools/ab/libsaigehip_cur.so: a copy of an earlier build), B = $AB_B (default the tree's build)
R=${1:-2}; shift || true
A=${AB_A:-$(pwd)/tools/ab/libsaigehip_cur.so}; B=${AB_B:-$(pwd)/saigegds_amd/libsaigehip.so}
ARGS="--steps 100 --cpu-seconds 0 --host-variants 0 --file-variants 0 --secondary 0 --resident-steps 0 $@"
brief() { python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('$1', d['ms_per_step'], 'ms/step', round(d['value']/1e6,2), 'M/s  kernel', r.get('avg_launch_ms'))"; }
for i in $(seq $R); do
  SAIGEHIP_LIB=$A python3 bench.py $ARGS 2>/dev/null | brief A
  SAIGEHIP_LIB=$B python3 bench.py $ARGS 2>/dev/null | brief B
done
