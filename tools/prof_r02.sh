#!/bin/bash
# Round-2 profile set, one gpurun call: rocprofv3 kernel stats + HBM PMC passes of bench.py, SQ/LDS counters of the batch kernels.
# usage (on the GPU box, from the repo root): bash tools/prof_r02.sh
set -u
# both libraries are built BEFORE the first rocprofv3 line: a profiled process has the GPU initialised by the profiler's preload
# and must not start a compiler chain (bitnuc_amd.build.ensure_built refuses to build there and says so)
python3 -m bitnuc_amd.build > /dev/null && python3 -m bitnuc_amd.build --sweep > /dev/null || { echo "build failed"; exit 1; }
bash tools/prof.sh r02
OUT=$PWD/gpurun_out/prof_r02
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/batch_trace" -o trace -- python3 "$ROOT/tools/run_batch.py" > "$OUT/batch_trace.log" 2>&1
echo "batch trace rc=$?"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/batch_pmc1" -o pmc -- python3 "$ROOT/tools/run_batch.py" > "$OUT/batch_pmc1.log" 2>&1
echo "batch pmc1 rc=$?"
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$OUT/batch_pmc2" -o pmc -- python3 "$ROOT/tools/run_batch.py" > "$OUT/batch_pmc2.log" 2>&1
echo "batch pmc2 rc=$?"
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
for sub in ("batch_pmc1", "batch_pmc2"):
    f = glob.glob(out + "/" + sub + "/**/*counter_collection.csv", recursive=True)
    if not f:
        print(sub, "no csv"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("bitnuc_dev::", "")[:60]
        if "bitnuc" in r["Kernel_Name"]:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(out + "/" + sub + ".txt", "w") as fo:
        for k, d in agg.items():
            line = k + " " + str({c: round(sum(v) / len(v)) for c, v in d.items()}) + " n=%d" % len(next(iter(d.values())))
            print(line); fo.write(line + "\n")
f = glob.glob(out + "/batch_trace/**/trace_kernel_stats.csv", recursive=True)
with open(out + "/batch_trace.txt", "w") as fo:
    for r in csv.DictReader(open(f[0])):
        if "bitnuc" in r["Name"]:
            line = f'{r["Name"].split("(")[0].replace("void ","").replace("bitnuc_dev::","")[:60]:62s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:8.1f} us  min {float(r["MinNs"])/1e3:8.1f}'
            print(line); fo.write(line + "\n")
PY
python3 tools/prof_summary.py "$OUT" r02 > "$OUT/summary.log" 2>&1; tail -5 "$OUT/summary.log"
