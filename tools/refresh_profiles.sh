#!/bin/bash
# Regenerates the round's evidence under gpurun_out/rNN/ on the GPU box (copy what is to be kept into profiles/):
#   bash tools/refresh_profiles.sh r04
# One MI355X; the commands are the ones profiles/README.md lists.
set -e -o pipefail
R=${1:-r04}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof2 -- python3 $ROOT/bench.py --steps 20 --warmup 2 --cpu-seconds 0 --host-variants 0 --file-variants 0 --secondary 0 --resident-steps 0 > $OUT/bench_under_rocprof.json 2> $OUT/prof2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof1 -- python3 $ROOT/bench.py --steps 20 --warmup 2 --cpu-seconds 0 --host-variants 0 --file-variants 0 --secondary 0 --resident-steps 0 --lanes 1 > $OUT/bench_under_rocprof_one_lane.json 2> $OUT/prof1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof13 -- python3 $ROOT/bench.py --k 13 --steps 10 --warmup 2 --cpu-seconds 0 --host-variants 0 --file-variants 0 --secondary 0 --resident-steps 0 --lanes 1 > $OUT/bench_k13_under_rocprof_one_lane.json 2> $OUT/prof13.err
echo "kernel stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/pmc_rd -- python3 $ROOT/bench.py --steps 4 --warmup 1 --cpu-seconds 0 --host-variants 0 --file-variants 0 --secondary 0 --resident-steps 0 --lanes 1 > $OUT/pmc_rd.json 2> $OUT/pmc_rd.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_wr -- python3 $ROOT/bench.py --steps 4 --warmup 1 --cpu-seconds 0 --host-variants 0 --file-variants 0 --secondary 0 --resident-steps 0 --lanes 1 > $OUT/pmc_wr.json 2> $OUT/pmc_wr.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/bench.py --steps 4 --warmup 1 --cpu-seconds 0 --host-variants 0 --file-variants 0 --secondary 0 --resident-steps 0 --lanes 1 > $OUT/pmc_sq.json 2> $OUT/pmc_sq.err
echo "pmc done"
cd $ROOT
python3 profiles/digest_pmc.py $OUT/pmc_rd $OUT/pmc_wr 5 $OUT/pmc_stages.json workload=c3 trait=binary n_samples=430000 variants_per_launch=50000 n_covariates=3 > $OUT/pmc_digest.txt
# (the bench line quotes the traffic of THIS build: the digest goes where bench.py looks for it before the line is made)
cp $OUT/pmc_stages.json profiles/${R}_pmc_stages.json
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"
for w in c2 c4; do python3 bench.py --workload $w --cpu-seconds 0 --host-variants 0 --file-variants 0 --secondary 0 > $OUT/bench_$w.json 2> $OUT/bench_$w.err; done
for k in 5 8 13 16; do python3 bench.py --k $k --cpu-seconds 0 --host-variants 0 --file-variants 0 --secondary 0 --resident-steps 0 > $OUT/bench_k$k.json 2> $OUT/bench_k$k.err; done
echo "configs done"
python3 profiles/show_pmc.py $OUT/pmc_rd $OUT/pmc_wr > $OUT/pmc_tcc.txt
python3 profiles/show_pmc.py $OUT/pmc_sq > $OUT/pmc_sq.txt
find $OUT/prof2 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_two_lanes.csv \;
find $OUT/prof1 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_one_lane.csv \;
find $OUT/prof13 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_k13_one_lane.csv \;
find $OUT/prof1 -name "*kernel_trace.csv" -exec python3 profiles/show_timeline.py {} \; > $OUT/timeline_one_lane.txt || true
find $OUT/prof2 -name "*kernel_trace.csv" -exec python3 profiles/show_timeline.py {} 13 3 \; > $OUT/timeline_two_lanes.txt || true
rm -rf $OUT/prof1 $OUT/prof2 $OUT/prof13 $OUT/pmc_rd $OUT/pmc_wr $OUT/pmc_sq
ls -la $OUT
