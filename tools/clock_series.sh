#!/bin/bash
# Effective shader clock of every launch of a queue: GRBM_GUI_ACTIVE (busy cycles, summed over the 8 XCDs) / 8 / launch duration, one --pmc pass
# (counters only) over tools/run_cfg35.py.  Shows whether a kernel's launch-to-launch slowdown is the CLOCK.
# usage (GPU box, repo root): bash tools/clock_series.sh cfg5 96   |   bash tools/clock_series.sh cfg3 24
set -u
# both libraries are built BEFORE the first rocprofv3 line: a profiled process must not start a compiler chain
python3 -m bitnuc_amd.build > /dev/null && python3 -m bitnuc_amd.build --sweep > /dev/null || { echo "build failed"; exit 1; }
CFG=${1:-cfg5}; N=${2:-96}
OUT=$PWD/gpurun_out/clock_$CFG
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc" -o pmc -- python3 "$ROOT/tools/run_cfg35.py" $CFG $N > "$OUT/run.log" 2>&1
echo "rc=$?"
cd "$ROOT"
python3 - "$OUT" "$CFG" <<'PY'
import csv, glob, sys
out, cfg = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "kmer_" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
us = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
mhz = [float(r["Counter_Value"]) / 8 / u for r, u in zip(rows, us)]
print(f"{cfg}: {len(rows)} launches of {rows[0]['Kernel_Name'].split('(')[0][-40:]}")
print("duration us:", " ".join(f"{u:.0f}" for u in us))
print("busy cycles per XCD / duration = MHz:", " ".join(f"{m:.0f}" for m in mhz))
k = max(4, len(us) // 6)
print(f"first 4: {sum(us[:4])/4:.1f} us at {sum(mhz[:4])/4:.0f} MHz; slowest: {max(us):.1f} us at {mhz[us.index(max(us))]:.0f} MHz; last {k}: {sum(us[-k:])/k:.1f} us at {sum(mhz[-k:])/k:.0f} MHz")
PY
