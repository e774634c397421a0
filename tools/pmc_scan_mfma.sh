#!/bin/bash
# Where does a round of the matrix-core scan / count spend its cycles?  SQ counters for every config-5 form (tools/run_scan_forms.py),
# counters only (no trace domains beside --pmc), one pass per counter group.  usage on the GPU box: bash tools/pmc_scan_mfma.sh OUTDIR
set -u
python3 -m bitnuc_amd.build > /dev/null && python3 -m bitnuc_amd.build --sweep > /dev/null || { echo "build failed"; exit 1; }
OUT=$PWD/${1:-gpurun_out/pmc_scan_mfma}
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 -L > "$OUT/counters_available.txt" 2>&1
pass() { # name counters...
    local name=$1; shift
    timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o pmc -- python3 "$ROOT/tools/run_scan_forms.py" 3 > "$OUT/$name.log" 2>&1
    echo "$name rc=$?"
}
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE &&
pass b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS &&
pass c SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "scan" not in name and "kmer_count" not in name:
            continue
        k = name.split("(")[0].replace("void ", "").replace("bitnuc_dev::", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fo:
    for k, d in sorted(agg.items()):
        m = {c: sum(v) / len(v) for c, v in d.items()}
        line = k + "\n   " + "  ".join(f"{c}={m[c]:.4g}" for c in sorted(m))
        print(line); fo.write(line + "\n")
PY
