#!/bin/bash
# rocprofv3 recipe for one round: kernel-trace stats of the timed step, HBM counters in separate --pmc passes (counters only,
# never combined with a trace domain), then kernel-trace stats of the full default run (side measurements included).
# usage (on the GPU box, from the repo root): bash tools/prof.sh r02
set -u
# both libraries are built BEFORE the first rocprofv3 line: a profiled process has the GPU initialised by the profiler's preload
# and must not start a compiler chain (bitnuc_amd.build.ensure_built refuses to build there and says so)
python3 -m bitnuc_amd.build > /dev/null && python3 -m bitnuc_amd.build --sweep > /dev/null || { echo "build failed"; exit 1; }
TAG=${1:-r00}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# --no-traffic: bench.py would otherwise start its own rocprofv3 children; --no-extras: only the step's two kernels run at full
# size, so the per-kernel averages of --stats are the step's (the side measurements launch the same kernels on small chunks)
STEP="python3 $PWD/bench.py --no-cpu-baseline --no-traffic --no-extras"
FULL="python3 $PWD/bench.py --no-cpu-baseline --no-traffic"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $STEP > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o pmc -- $STEP > "$OUT/pmc_fetch.log" 2>&1
echo "pmc fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o pmc -- $STEP > "$OUT/pmc_write.log" 2>&1
echo "pmc write rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_full" -o trace -- $FULL > "$OUT/trace_full.log" 2>&1
echo "full trace rc=$?"
find "$OUT" -name "*.csv" | head -20
