#!/bin/bash
# Round-3 profile set, one gpurun call: rocprofv3 kernel stats + HBM PMC passes of bench.py (tools/prof.sh), the kernel trace of
# the ragged-batch kernels (plan_emit_kernel's own duration), TCC write-request counters of the bulk encode's store shapes, and
# SQ / LDS counters of the two every-window kernels.  Counters only in --pmc passes (never combined with a trace domain).
# usage (on the GPU box, from the repo root): bash tools/prof_r03.sh
set -u
# both libraries are built BEFORE the first rocprofv3 line: a profiled process has the GPU initialised by the profiler's preload
# and must not start a compiler chain (bitnuc_amd.build.ensure_built refuses to build there and says so)
python3 -m bitnuc_amd.build > /dev/null && python3 -m bitnuc_amd.build --sweep > /dev/null || { echo "build failed"; exit 1; }
bash tools/prof.sh r03
OUT=$PWD/gpurun_out/prof_r03
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 -L > "$OUT/counters_avail.txt" 2>&1
grep -i -o "TCC_EA0_WRREQ[A-Z0-9_]*\|TCC_EA0_WR_[A-Z0-9_]*\|TCC_WRREQ[A-Z0-9_]*\|TCC_EA0_RDREQ[A-Z0-9_]*" "$OUT/counters_avail.txt" | sort -u > "$OUT/tcc_counters.txt"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/batch_trace" -o trace -- python3 "$ROOT/tools/run_batch.py" > "$OUT/batch_trace.log" 2>&1
echo "batch trace rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/enc_trace" -o trace -- python3 "$ROOT/tools/run_encode_variants.py" > "$OUT/enc_trace.log" 2>&1
echo "encode trace rc=$?"
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d "$OUT/enc_pmc_wr" -o pmc -- python3 "$ROOT/tools/run_encode_variants.py" > "$OUT/enc_pmc_wr.log" 2>&1
echo "encode pmc wr rc=$?"
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d "$OUT/enc_pmc_rd" -o pmc -- python3 "$ROOT/tools/run_encode_variants.py" > "$OUT/enc_pmc_rd.log" 2>&1
echo "encode pmc rd rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/win_trace" -o trace -- python3 "$ROOT/tools/run_windows.py" > "$OUT/win_trace.log" 2>&1
echo "windows trace rc=$?"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d "$OUT/win_pmc1" -o pmc -- python3 "$ROOT/tools/run_windows.py" > "$OUT/win_pmc1.log" 2>&1
echo "windows pmc1 rc=$?"
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LEVEL_WAVES --output-format csv -d "$OUT/win_pmc2" -o pmc -- python3 "$ROOT/tools/run_windows.py" > "$OUT/win_pmc2.log" 2>&1
echo "windows pmc2 rc=$?"
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
def short(n):
    return n.split("(")[0].replace("void ", "").replace("bitnuc_dev::", "")[:70]
for sub in ("enc_pmc_wr", "enc_pmc_rd", "win_pmc1", "win_pmc2"):
    f = glob.glob(out + "/" + sub + "/**/*counter_collection.csv", recursive=True)
    if not f:
        print(sub, "no csv"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if "bitnuc" in r["Kernel_Name"] and "nucgen" not in r["Kernel_Name"]:
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(out + "/" + sub + ".txt", "w") as fo:
        for k, d in agg.items():
            line = k + " " + str({c: round(sum(v) / len(v)) for c, v in d.items()}) + " n=%d" % len(next(iter(d.values())))
            print(line); fo.write(line + "\n")
for sub in ("batch_trace", "enc_trace", "win_trace"):
    f = glob.glob(out + "/" + sub + "/**/trace_kernel_stats.csv", recursive=True)
    if not f:
        print(sub, "no stats"); continue
    with open(out + "/" + sub + ".txt", "w") as fo:
        for r in csv.DictReader(open(f[0])):
            if "bitnuc" in r["Name"]:
                line = f'{short(r["Name"]):72s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:8.1f} us  min {float(r["MinNs"])/1e3:8.1f}'
                print(line); fo.write(line + "\n")
PY
python3 tools/prof_summary.py "$OUT" r03 > "$OUT/summary.log" 2>&1; tail -5 "$OUT/summary.log"
