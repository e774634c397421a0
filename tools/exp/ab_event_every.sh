#!/bin/bash
# does recording the per-kernel HIP events on EVERY timed step cost step time?  (bench.py --event-every k)
for k in 1 10 1 10 50 1; do
  timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline --no-traffic --event-every $k 2>/dev/null > /tmp/line.json
  python - "$k" <<'PY'
import json, sys
d = json.load(open("/tmp/line.json"))
print("event-every", sys.argv[1], "value", d["value"], "ms/step", d["ms_per_step"], "enc ms", d["roofline_encode"]["avg_launch_ms"], "dec ms", d["roofline_decode"]["avg_launch_ms"])
PY
done
