#!/bin/bash
# Process-to-process stability of the timed step for a few encode variants: N separate bench.py processes each (fresh allocations, so a
# different physical placement every time), same box.  usage on the GPU box: bash tools/enc_variant_stability.sh "39 2 14 41" 5
# both libraries are built BEFORE the first rocprofv3 line: a profiled process has the GPU initialised by the profiler's preload
# and must not start a compiler chain (bitnuc_amd.build.ensure_built refuses to build there and says so)
python3 -m bitnuc_amd.build > /dev/null && python3 -m bitnuc_amd.build --sweep > /dev/null || { echo "build failed"; exit 1; }
VARS=${1:-"39 2 14"}
N=${2:-5}
for i in $(seq 1 $N); do
  for v in $VARS; do
    python3 bench.py --evidence-build --enc-variant $v --no-extras --no-cpu-baseline --no-traffic --steps 200 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('run $i e%-3d step %.4f ms  value %7.1f  encode %.4f ms  decode %.4f ms' % ($v, d['ms_per_step'], d['value'], d['roofline_encode']['avg_launch_ms'], d['roofline_decode']['avg_launch_ms']))"
  done
done
