#!/bin/bash
# LDS counters of the ragged-batch kernels (one --pmc pass, no trace domain): usage on the GPU box: bash tools/pmc_batch_lds.sh OUTDIR
set -u
# both libraries are built BEFORE the first rocprofv3 line: a profiled process has the GPU initialised by the profiler's preload
# and must not start a compiler chain (bitnuc_amd.build.ensure_built refuses to build there and says so)
python3 -m bitnuc_amd.build > /dev/null && python3 -m bitnuc_amd.build --sweep > /dev/null || { echo "build failed"; exit 1; }
OUT=$PWD/${1:-gpurun_out/pmc_lds}
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d "$OUT/pmc" -o pmc -- python3 "$ROOT/tools/run_batch.py" > "$OUT/pmc.log" 2>&1
echo "pmc rc=$?"
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
f = glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    if "bitnuc" in r["Kernel_Name"]:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("bitnuc_dev::", "")[:60]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/lds.txt", "w") as fo:
    for k, d in agg.items():
        m = {c: round(sum(v) / len(v)) for c, v in d.items()}
        ratio = m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_INSTS_LDS"] if m.get("SQ_INSTS_LDS") else 0
        line = f"{k:44s} {m}  conflict cycles / LDS instruction = {ratio:.3f}"
        print(line); fo.write(line + "\n")
PY
