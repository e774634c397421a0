#!/usr/bin/env python3
"""Where does the GPU sit relative to the host's NUMA nodes, and how much does the pipelined host path vary from one fresh
context to the next (thread and page placement lottery)?"""
import glob
import os

os.environ.setdefault("BITNUC_PIPE_IMPL", "staged")  # this tool studies the STAGED engine's thread budget / placement (the direct engine ships: tools/ab_pipe_impl.py)
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
print(subprocess.run("lscpu | grep -i -E 'numa|socket|model name|^CPU\\(s\\)'", shell=True, capture_output=True, text=True).stdout)
for d in sorted(glob.glob("/sys/class/drm/card*/device")):
    try:
        print(d, "numa_node", open(d + "/numa_node").read().strip(), "local_cpulist", open(d + "/local_cpulist").read().strip(), "vendor", open(d + "/vendor").read().strip())
    except OSError as e:
        print(d, e)
print("affinity:", len(os.sched_getaffinity(0)), "cpus; cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "n/a")
print("mems_allowed:", [l.strip() for l in open("/proc/self/status") if "Mems_allowed_list" in l or "Cpus_allowed_list" in l])
import numpy as np
import torch
import bitnuc_amd
p = torch.cuda.get_device_properties(0)
print("pci:", getattr(p, "pci_bus_id", "?"), getattr(p, "pci_device_id", "?"), getattr(p, "pci_domain_id", "?"))
n = 10**9
seq = np.repeat(np.frombuffer(b"ACGT", dtype=np.uint8)[np.frombuffer(np.random.default_rng(1).bytes(n // 4 + 1), dtype=np.uint8) & 3], 4)[:n].copy()
w = np.zeros((n + 31) // 32, dtype=np.uint64)
back = np.zeros(n, dtype=np.uint8)
for trial in range(12):
    os.environ["BITNUC_PIPE_NUMA"] = "1" if trial % 2 == 0 else "0"  # alternate: workers bound to the GPU's node / free
    ctx = bitnuc_amd.Context(0)
    info = ctx.host_pipe_info()
    te, td = [], []
    for _ in range(3):
        t = time.perf_counter(); ctx.encode_into(seq, w); te.append(time.perf_counter() - t)
        t = time.perf_counter(); ctx.decode_into(w, n, back); td.append(time.perf_counter() - t)
    print(f"fresh context {trial} (node {info['gpu_numa_node']}, workers bound to {info['workers_bound_to_cpus']} cpus): encode {n / min(te[1:]) / 1e9:5.1f}  decode {n / min(td[1:]) / 1e9:5.1f} Gbases/s   (all: enc {[round(n / x / 1e9, 1) for x in te]} dec {[round(n / x / 1e9, 1) for x in td]})", flush=True)
    ctx.close()
