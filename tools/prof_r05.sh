#!/bin/bash
# Round-5 profile set, one gpurun call: rocprofv3 kernel stats + HBM PMC passes of bench.py's timed step (tools/prof.sh), then the same
# three passes for BASELINE config 3 (10^8 dense 31-mers), config 5 (10^9-base scan: the matrix-core form, a queue of 96 launches from an
# idle chip) and config 5's fused count ALONE at full size (tools/run_cfg35.py), so that their fractions can be recomputed from profiles/
# without small launches of the same kernels mixed in.  Counters only in --pmc passes (never combined with a trace domain); the program
# itself follows `--`.
# usage (on the GPU box, from the repo root): bash tools/prof_r05.sh
set -u
python3 -m bitnuc_amd.build > /dev/null && python3 -m bitnuc_amd.build --sweep > /dev/null || { echo "build failed"; exit 1; }
bash tools/prof.sh r05
OUT=$PWD/gpurun_out/prof_r05
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
for CFG in cfg3 cfg5 cfg5count; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${CFG}_trace" -o trace -- python3 "$ROOT/tools/run_cfg35.py" $CFG > "$OUT/${CFG}_trace.log" 2>&1
  echo "$CFG trace rc=$?"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/${CFG}_pmc_fetch" -o pmc -- python3 "$ROOT/tools/run_cfg35.py" $CFG 12 > "$OUT/${CFG}_pmc_fetch.log" 2>&1
  echo "$CFG pmc fetch rc=$?"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/${CFG}_pmc_write" -o pmc -- python3 "$ROOT/tools/run_cfg35.py" $CFG 12 > "$OUT/${CFG}_pmc_write.log" 2>&1
  echo "$CFG pmc write rc=$?"
done
cd "$ROOT"
python3 tools/prof_summary.py "$OUT" r05 > "$OUT/summary.log" 2>&1; tail -30 "$OUT/summary.log"
