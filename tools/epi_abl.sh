#!/bin/bash
# one-lane kernel stats of the epilogue under the debug bits (spa_abl >> 16)
mkdir -p gpurun_out/q4; R=$(pwd); cd /tmp; export TMPDIR=/tmp
for d in 0 1 2 4 8 6; do
  SGX_BENCH_OPTS="spa_abl=$((d<<16))" rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/q4/p$d -- python3 $R/bench.py --steps 6 --warmup 1 --cpu-seconds 0 --host-variants 0 --file-variants 0 --secondary 0 --resident-steps 0 --lanes 1 "$@" > $R/gpurun_out/q4/b$d.json 2>&1
  f=$(find $R/gpurun_out/q4/p$d -name "*kernel_stats.csv"); echo "dbg=$d: $(python3 $R/profiles/show_stats.py $f | grep epilogue)"
  rm -rf $R/gpurun_out/q4/p$d
done
