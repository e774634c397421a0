#!/bin/bash
# usage: bash tools/prof_pmc.sh <tag> "<counters>" -- <python script args...>
# runs `python3 <args>` under rocprofv3 --pmc (counters only, no tracing domains) and prints per-kernel averages
set -u
# both libraries are built BEFORE the first rocprofv3 line: a profiled process has the GPU initialised by the profiler's preload
# and must not start a compiler chain (bitnuc_amd.build.ensure_built refuses to build there and says so)
python3 -m bitnuc_amd.build > /dev/null && python3 -m bitnuc_amd.build --sweep > /dev/null || { echo "build failed"; exit 1; }
TAG=$1; CTRS=$2; shift 3
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 --pmc $CTRS --output-format csv -d "$OUT" -o pmc -- python3 "$@" > "$OUT/run.log" 2>&1
echo "rc=$?"
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/pmc_counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("bitnuc_dev::", "")[:60]
    if "bitnuc" in r["Kernel_Name"]:
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "n=%d" % len(next(iter(d.values()))))
PY
