#!/bin/bash
# one-lane kernel stats of one build: bash tools/one_lane_lib.sh tag lib.so [bench args]
T=$1; L=$2; shift; shift; R=$(pwd); mkdir -p gpurun_out/$T; cd /tmp; export TMPDIR=/tmp
SAIGEHIP_LIB=$L rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/p -- python3 $R/bench.py --steps 8 --warmup 2 --cpu-seconds 0 --host-variants 0 --file-variants 0 --secondary 0 --resident-steps 0 --lanes 1 "$@" > $R/gpurun_out/$T/bench.json 2>&1
cd $R; find gpurun_out/$T/p -name "*kernel_stats.csv" -exec cp {} gpurun_out/$T/ks.csv \; ; rm -rf gpurun_out/$T/p
python3 profiles/show_stats.py gpurun_out/$T/ks.csv | grep "spa4_moments"
