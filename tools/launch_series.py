#!/usr/bin/env python3
"""Per-launch durations of N back-to-back launches of one kernel (HIP events between consecutive launches on the launch stream, no
host wait inside the series): does a kernel keep its rate when the queue never drains?  rocprofv3 showed the config-5 scan at
~312 us for its first five launches and 380-470 us from the sixth on (profiles/r04_rocprof/kernel_stats_cfg5.csv); this prints the
same series without a profiler, for the scan and for the other full-size kernels, together with the GPU's clock / power files when
the box lets an ordinary user read them.
usage: launch_series.py [N=48] [kernel ...]   kernels: scan dense encode decode count windows"""
import glob
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bitnuc_amd

N = int(sys.argv[1]) if len(sys.argv) > 1 else 48
which = sys.argv[2:] or ["scan", "dense", "encode", "decode", "count"]
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
ctx = bitnuc_amd.Context(0, stream=stream.cuda_stream)
SEED, k = 0xB17C0DE, 31
n = 10**9
ref = torch.empty(n, dtype=torch.uint8, device=dev)
ctx.nucgen_dev(ref, n, SEED)
q = 0x1B1B1B1B1B1B1B1B & ((1 << 62) - 1)
bufs = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(2)]
words = [torch.empty((n + 31) // 32, dtype=torch.int64, device=dev) for _ in range(2)]
kout = [torch.empty(10**8, dtype=torch.int64, device=dev) for _ in range(2)]
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
ctx.encode_dev(ref, n, words[0])
ctx.encode_dev(ref, n, words[1])
ctx.sync()
KERNELS = {
    "scan": (lambda i: ctx.kmer_hdist_scan_dev(ref, n, k, q, bufs[i & 1]), 2 * (n - 30)),
    "count": (lambda i: ctx.kmer_hdist_count_dev(ref, n, k, q, 8, cnt), n - 30),
    "dense": (lambda i: ctx.as_2bit_batch_dev(ref, k, k, n // k, kout[i & 1]), (n // k) * 39),
    "encode": (lambda i: ctx.encode_dev(ref, n, words[i & 1]), 1.25 * n),
    "decode": (lambda i: ctx.decode_dev(words[i & 1], (n + 31) // 32, n, bufs[i & 1]), 1.25 * n),
    "windows": (lambda i: ctx.as_2bit_batch_dev(ref, k, 1, 10**8, kout[i & 1]), 9e8),
}


def sensors():
    out = {}
    for pat, name in (("/sys/class/drm/card*/device/pp_dpm_sclk", "sclk"), ("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average", "power_uW"),
                      ("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input", "freq1_Hz"), ("/sys/class/drm/card*/device/hwmon/hwmon*/temp1_input", "temp_mC")):
        for f in glob.glob(pat)[:1]:
            try:
                txt = open(f).read().strip()
                out[name] = next((l for l in txt.splitlines() if l.endswith("*")), txt.splitlines()[-1] if txt else "")
            except Exception as e:  # noqa: BLE001
                out[name] = f"unreadable ({type(e).__name__})"
    return out


print("sensors at start:", sensors(), flush=True)
for name in which:
    fn, alg = KERNELS[name]
    for gap_ms in (0, 5):  # 5: a 5 ms host sleep + sync after every 8 launches (how bench.py's bursts run)
        torch.cuda.synchronize()
        time.sleep(0.5)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
        samples = []
        stop = threading.Event()

        def sample():
            while not stop.is_set():
                samples.append(sensors())
                time.sleep(0.002)
        th = threading.Thread(target=sample)
        th.start()
        fn(0)
        ev[0].record(stream)
        for i in range(N):
            fn(i + 1)
            ev[i + 1].record(stream)
            if gap_ms and i % 8 == 7:
                torch.cuda.synchronize()
                time.sleep(gap_ms * 1e-3)
        torch.cuda.synchronize()
        stop.set()
        th.join()
        us = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(N)]
        if gap_ms:
            us = [u for i, u in enumerate(us) if i % 8 != 0]  # the first interval after a gap contains the gap
        head, tail = us[:4], us[-8:]
        print(f"{name:8s} {'bursts of 8 + 5 ms gaps' if gap_ms else 'one queue, no gaps      '}: first 4 avg {sum(head)/len(head):6.1f} us ({alg/(sum(head)/len(head))/8e6*100:4.1f} % of 8 TB/s)  "
              f"last 8 avg {sum(tail)/len(tail):6.1f} us ({alg/(sum(tail)/len(tail))/8e6*100:4.1f} %)  series: " + " ".join(f"{u:.0f}" for u in us), flush=True)
        keys = sorted({k2 for s in samples for k2 in s})
        for k2 in keys:
            vals = [s.get(k2) for s in samples]
            uniq = []
            for v in vals:
                if not uniq or uniq[-1] != v:
                    uniq.append(v)
            print(f"           {k2}: " + " -> ".join(str(u) for u in uniq[:12]), flush=True)
ctx.close()
