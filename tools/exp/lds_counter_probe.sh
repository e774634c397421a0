#!/bin/bash
# usage on the GPU box: bash tools/exp/lds_counter_probe.sh OUTDIR
set -u
OUT=$PWD/${1:-gpurun_out/lds_probe}
mkdir -p "$OUT"
ROOT=$PWD
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o "$OUT/lds_counter_probe" "$ROOT/tools/exp/lds_counter_probe.hip" || exit 1
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/pmc" -o pmc -- "$OUT/lds_counter_probe" > "$OUT/run.log" 2>&1
echo "rc=$?"
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
f = glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True)
names = ["ds_read_b32 stride 1", "ds_read_b32 stride 2 (2-way)", "ds_write_b32 stride 1", "ds_write_b32 stride 2 (2-way)", "ds_or_b32 stride 1 (no return)",
         "ds_or_b32 stride 2 (2-way)", "ds_add_u32 stride 1 (no return)", "ds_or_rtn_b32 stride 1"]
agg = collections.defaultdict(dict)
for r in csv.DictReader(open(f[0])):
    if "probe<" in r["Kernel_Name"]:
        agg[r["Kernel_Name"]][r["Counter_Name"]] = float(r["Counter_Value"])
pat = 1024 * 4 * 64
with open(out + "/lds_counter_probe.txt", "w") as fo:
    for k in sorted(agg):
        m = int(k.split("<")[1].split(">")[0])
        d = agg[k]
        # the fill loop and the final read add 8 + 1 LDS instructions per wave; the pattern adds 64
        line = f"{names[m]:34s} LDS instr {d['SQ_INSTS_LDS']:.0f}  BANK_CONFLICT {d['SQ_LDS_BANK_CONFLICT']:.0f}  IDX_ACTIVE {d['SQ_LDS_IDX_ACTIVE']:.0f}  -> conflict cycles per pattern instruction {d['SQ_LDS_BANK_CONFLICT'] / pat:.2f}, array cycles per LDS instruction {d['SQ_LDS_IDX_ACTIVE'] / d['SQ_INSTS_LDS']:.2f}"
        print(line); fo.write(line + "\n")
PY
