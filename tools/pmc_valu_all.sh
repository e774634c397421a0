#!/bin/bash
# Vector-ALU utilisation of every kernel of the default bench run (one --pmc pass, counters only).
# usage on the GPU box: bash tools/pmc_valu_all.sh OUTDIR
set -u
# both libraries are built BEFORE the first rocprofv3 line: a profiled process has the GPU initialised by the profiler's preload
# and must not start a compiler chain (bitnuc_amd.build.ensure_built refuses to build there and says so)
python3 -m bitnuc_amd.build > /dev/null && python3 -m bitnuc_amd.build --sweep > /dev/null || { echo "build failed"; exit 1; }
OUT=$PWD/${1:-gpurun_out/pmc_valu}
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc" -o pmc -- python3 "$ROOT/bench.py" --no-traffic --no-cpu-baseline --steps 20 > "$OUT/run.log" 2>&1
echo "rc=$?"
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
f = glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    if "bitnuc" in r["Kernel_Name"]:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("bitnuc_dev::", "")[:64]
        agg[(k, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for (k, grid), d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    clocks = m.get("GRBM_GUI_ACTIVE", 0) / 8  # summed over the 8 XCDs
    if clocks < 20000:  # < ~10 us: launch-dominated
        continue
    busy = m.get("SQ_INSTS_VALU", 0) / 1024 * 4 / clocks  # wave64 instruction = 4 clocks of a 16-lane SIMD; 1024 SIMDs
    rows.append((busy, k, grid, m))
with open(out + "/valu.txt", "w") as fo:
    for busy, k, grid, m in sorted(rows, reverse=True):
        line = f"{k:66s} grid {grid:>10s}  VALU {m.get('SQ_INSTS_VALU',0)/1e6:8.1f} M  SALU {m.get('SQ_INSTS_SALU',0)/1e6:7.1f} M  LDS {m.get('SQ_INSTS_LDS',0)/1e6:6.1f} M  clocks/XCD {m.get('GRBM_GUI_ACTIVE',0)/8/1e3:7.0f} K  vector ALUs busy {100*busy:5.1f} %"
        print(line); fo.write(line + "\n")
PY
