#!/bin/bash
# usage: bash tools/prof_stats.sh <tag> -- <python script args...> : rocprofv3 --kernel-trace --stats, prints our kernels' averages
set -u
# both libraries are built BEFORE the first rocprofv3 line: a profiled process has the GPU initialised by the profiler's preload
# and must not start a compiler chain (bitnuc_amd.build.ensure_built refuses to build there and says so)
python3 -m bitnuc_amd.build > /dev/null && python3 -m bitnuc_amd.build --sweep > /dev/null || { echo "build failed"; exit 1; }
TAG=$1; shift 2
OUT=$PWD/gpurun_out/stats_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o t -- python3 "$@" > "$OUT/run.log" 2>&1
echo "rc=$?"
python3 - "$OUT" <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/t_kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if "bitnuc" in r["Name"]:
        print(f'{r["Name"].split("(")[0].replace("void ","").replace("bitnuc_dev::","")[:60]:62s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:8.1f} us  min {float(r["MinNs"])/1e3:8.1f}')
PY
