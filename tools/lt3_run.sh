#!/bin/bash
# list + T3 pass of the tool: checks, timing, vector-instruction counts of the large launches
mkdir -p gpurun_out/lt3
RM=1 LT3=1 timeout -k 10 300 ./tools/score3_bench_small 430000 50000 5 > gpurun_out/lt3/b.txt 2>&1
echo "ok lines: $(grep -c ': ok' gpurun_out/lt3/b.txt)"; grep -i "fail\|differ" gpurun_out/lt3/b.txt | head
tail -4 gpurun_out/lt3/b.txt
R=$(pwd); cd /tmp; export TMPDIR=/tmp
RM=1 LT3=1 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/lt3/pmc -- $R/tools/score3_bench_small 430000 50000 1 > $R/gpurun_out/lt3/pmc.txt 2>&1
cd $R
python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/lt3/pmc/**/*counter_collection.csv",recursive=True)[0]
per=collections.defaultdict(float); gs={}
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] not in ("SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_INSTS_LDS"): continue
    k=(r["Kernel_Name"].split("(")[0][:45],r["Dispatch_Id"],r["Counter_Name"]); per[k]+=float(r["Counter_Value"]); gs[k]=r["Grid_Size"]
seen=set()
for k,v in per.items():
    if int(gs[k])>1e7 and (k[0],k[2]) not in seen: seen.add((k[0],k[2])); print(k[0],k[2],"%.3e"%v)
PY
rm -rf gpurun_out/lt3/pmc
