#!/bin/bash
# Same-box A/B of two builds of the library: tools/ab/libsaigehip_base.so (a copy of an earlier build) against
# saigegds_amd/libsaigehip.so, alternating, the bench's step only.   bash tools/ab.sh [rounds=2] [extra bench args]
R=${1:-2}; shift || true
ARGS="--steps 100 --cpu-seconds 0 --host-variants 0 --file-variants 0 --secondary 0 --resident-steps 0 $@"
brief() { python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('$1', d['ms_per_step'], 'ms/step', round(d['value']/1e6,2), 'M/s  kernel', r.get('avg_launch_ms'), 'lists', (r.get('stages') or {}).get('lists',{}).get('ms'))"; }
for i in $(seq $R); do
  SAIGEHIP_LIB=$(pwd)/tools/ab/libsaigehip_base.so python3 bench.py $ARGS 2>/dev/null | brief base
  python3 bench.py $ARGS 2>/dev/null | brief new
done
