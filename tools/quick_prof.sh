#!/bin/bash
# Bench line + per-kernel durations (one lane, two lanes) into gpurun_out/$1/ -- a short form of refresh_profiles.sh
#   bash tools/quick_prof.sh tag [extra bench args]
set -e -o pipefail
R=${1:-q}; shift || true
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py --steps 30 --cpu-seconds 4 --host-variants 0 --file-variants 0 --secondary 0 "$@" > $OUT/bench.json 2> $OUT/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof1 -- python3 $ROOT/bench.py --steps 10 --warmup 2 --cpu-seconds 0 --host-variants 0 --file-variants 0 --secondary 0 --resident-steps 0 --lanes 1 "$@" > $OUT/bench_one_lane.json 2> $OUT/prof1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof2 -- python3 $ROOT/bench.py --steps 10 --warmup 2 --cpu-seconds 0 --host-variants 0 --file-variants 0 --secondary 0 --resident-steps 0 "$@" > $OUT/bench_two_lanes.json 2> $OUT/prof2.err
cd $ROOT
find $OUT/prof2 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_two_lanes.csv \;
find $OUT/prof1 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_one_lane.csv \;
find $OUT/prof1 -name "*kernel_trace.csv" -exec cp {} $OUT/kernel_trace_one_lane.csv \;
find $OUT/prof2 -name "*kernel_trace.csv" -exec cp {} $OUT/kernel_trace_two_lanes.csv \;
rm -rf $OUT/prof1 $OUT/prof2
python3 tools/bench_brief.py $OUT/bench.json || true
echo "--- one lane"; python3 profiles/show_stats.py $OUT/kernel_stats_one_lane.csv | head -24
echo "--- one lane, one step"; python3 profiles/show_timeline.py $OUT/kernel_trace_one_lane.csv || true
echo "--- two lanes"; python3 profiles/show_stats.py $OUT/kernel_stats_two_lanes.csv | head -24
