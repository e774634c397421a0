#!/bin/bash
# rocprofv3 recipe for one round: kernel-trace stats, then HBM counters in separate --pmc passes.
# usage (on the GPU box, from the repo root): bash tools/prof.sh r01
set -u
TAG=${1:-r00}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --no-cpu-baseline"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $BENCH > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o pmc -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
echo "pmc fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o pmc -- $BENCH > "$OUT/pmc_write.log" 2>&1
echo "pmc write rc=$?"
find "$OUT" -name "*.csv" | head -20
