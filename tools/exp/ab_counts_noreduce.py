import os, statistics, sys
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT)
import torch, bitnuc_amd
dev = torch.device("cuda:0"); stream = torch.cuda.current_stream(); n = 10**9
ctxs = {k: bitnuc_amd.Context(0, stream=stream.cuda_stream, lib_path=os.path.join(ROOT, "bitnuc_amd", f"libbitnuc_hip_{k}.so")) for k in ("cu4", "cu_noreduce")}
c0 = ctxs["cu4"]
words = [torch.empty(n // 32, dtype=torch.int64, device=dev) for _ in range(2)]
seq = torch.empty(n, dtype=torch.uint8, device=dev)
for r in range(2):
    c0.nucgen_dev(seq, n, 5 + r); c0.encode_dev(seq, n, words[r])
c0.sync()
counts = torch.zeros(4, dtype=torch.int64, device=dev)
res = {}
for rnd in range(6):
  for mult in (2, 4, 8, 16):
    for k0, c in ctxs.items():
        k = f"{k0} x{mult}"
        c.require_variant("reduce_mult", mult)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c.base_counts_dev(words[0], n // 32, n, counts)
        a.record(stream)
        for i in range(8): c.base_counts_dev(words[i & 1], n // 32, n, counts)
        b.record(stream); torch.cuda.synchronize()
        if rnd: res.setdefault(k, []).append(a.elapsed_time(b) / 8)
for k, v in res.items():
    m = statistics.median(v); print(f"{k:18s} {m*1e3:6.1f} us  {0.25*n/m/1e6:6.0f} GB/s")
