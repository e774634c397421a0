#!/usr/bin/env python3
"""Rewrite the numeric cells of README.md's results table from profiles/r05_bench_n1.json and profiles/hbm_traffic.json (the two files a round's final
evidence run leaves), so that the table cannot drift from the files it quotes.  Rows are found by their first cell; prose around the numbers stays as
written here.  usage: python3 tools/readme_results.py [--check]   (--check: exit 1 if README.md would change)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_n1.json")))
t = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
X = {}


def walk(o):
    if isinstance(o, dict):
        for k, v in o.items():
            if isinstance(v, dict) and k not in X:
                X[k] = v
            walk(v)


walk(d)
pc = lambda x: f"{100 * x:.1f}"
scan, cnt = X["kmer_hdist_scan"], X["kmer_hdist_count"]
sq, cq = scan["from_idle_queue_of_96"], cnt["from_idle_queue_of_96"]
rb, rt, rf, cpu = X["reads_batch"], X["reads_batch_tables"], X["reads_fixed"], d["cpu_baseline"]
c5, c5c = t["cfg5"], t["cfg5count"]
enc_p, dec_p = 1.25e9 / t["encode_avg_ns"] / 8000, 1.25e9 / t["decode_avg_ns"] / 8000
settled = c5["last16_avg_ns"] / 1e3
rows = {
    "| bulk encode, 10⁹ bases (cfg 2) |": f"| bulk encode, 10⁹ bases (cfg 2) | {d['encode_gbases_s']:.0f} Gbases/s, {d['roofline']['avg_launch_ms']:.3f} ms; PMC traffic 1.0001 × algorithmic | {pc(d['configs']['cfg2_encode'])} (rocprofv3, kernel alone: {pc(enc_p)}) |",
    "| bulk decode, 10⁹ bases (cfg 2), input cache-cold |": f"| bulk decode, 10⁹ bases (cfg 2), input cache-cold | {d['decode_gbases_s']:.0f} Gbases/s, {1e3 / d['decode_gbases_s']:.3f} ms; PMC 1.0001 × | {pc(d['configs']['cfg2_decode'])} ({pc(dec_p)}) |",
    "| encode + decode step (`bench.py` `value` = encoded + decoded bases) |": f"| encode + decode step (`bench.py` `value` = encoded + decoded bases) | **{d['value']:.0f} Gbases/s**, {d['ms_per_step']:.4f} ms/step | {100 * 2.5e9 / (d['ms_per_step'] * 1e-3) / 8e12:.1f} |",
    "| 10⁸ dense 31-mers → u64 (cfg 3) |": f"| 10⁸ dense 31-mers → u64 (cfg 3) | {X['kmer_batch']['gkmers_s']:.0f} G k-mers/s, {X['kmer_batch']['ms']:.3f} ms | {pc(d['configs']['cfg3_kmer_batch'])} (rocprofv3 {pc(t['cfg3']['frac_of_8tb_s'])}) |",
    "| every 31-base window → u64 (stride 1), 10⁹ bases |": f"| every 31-base window → u64 (stride 1), 10⁹ bases | {X['kmer_windows']['gwindows_s'] / 1e3:.2f} T windows/s, {X['kmer_windows']['ms']:.3f} ms (≈ the fill rate of its 9 GB footprint; 1.55–1.64 over the round's boxes) | {pc(X['kmer_windows']['roofline']['frac'])} |",
    "| `hdist` of two 10⁹-base packed buffers |": f"| `hdist` of two 10⁹-base packed buffers | {X['hdist_bulk']['ms']:.4f} ms | {pc(X['hdist_bulk']['roofline']['frac'])} |",
    "| A/C/G/T counts of 10⁹ packed bases (no decode) |": f"| A/C/G/T counts of 10⁹ packed bases (no decode) | {X['base_counts']['ms']:.4f} ms | {pc(X['base_counts']['roofline']['frac'])} |",
    "| one packed 32-mer vs 3.1·10⁷ packed 32-mers (`hdist_query`) |": f"| one packed 32-mer vs 3.1·10⁷ packed 32-mers (`hdist_query`) | {X['hdist_query']['ms']:.4f} ms | {pc(X['hdist_query']['roofline']['frac'])} |",
    "| `split_packed` of 10⁹ packed bases, mid-word |": f"| `split_packed` of 10⁹ packed bases, mid-word | {X['split_packed']['ms']:.4f} ms | {pc(X['split_packed']['roofline']['frac'])} |",
    "| 6.7 M × 150-base reads with a layout plan, encode / decode |": f"| 6.7 M × 150-base reads with a layout plan, encode / decode | {rb['encode_ms']:.4f} / {rb['decode_ms']:.4f} ms | {pc(rb['encode_frac'])} / {pc(rb['decode_frac'])} (with the plan's own bytes: {pc(rb['encode_frac_with_plan_bytes'])} / {pc(rb['decode_frac_with_plan_bytes'])}) |",
    "| same batch from the offset tables alone (tables counted: every call reads them) |": f"| same batch from the offset tables alone (tables counted: every call reads them) | {rt['encode_ms']:.4f} / {rt['decode_ms']:.4f} ms | {pc(rt['encode_frac'])} / {pc(rt['decode_frac'])} (bases + words only: {pc(rt['encode_frac_without_tables'])} / {pc(rt['decode_frac_without_tables'])}) |",
    "| same reads, `encode_fixed` / `decode_fixed` (no tables) |": f"| same reads, `encode_fixed` / `decode_fixed` (no tables) | {rf['encode_ms']:.4f} / {rf['decode_ms']:.4f} ms | {pc(rf['encode_frac'])} / {pc(rf['decode_frac'])} |",
    "| CPU: the reference's AVX2 algorithm restated in C, 1 core of the box's AMD EPYC 9575F |": f"| CPU: the reference's AVX2 algorithm restated in C, 1 core of the box's AMD EPYC 9575F | {cpu['value']:.2f} Gbases/s (`-march=x86-64-v3`), {cpu['native_value']:.2f} (`-march=native`); 16 cores: {cpu['all_cores']['value']:.1f} | — |",
}
path = os.path.join(ROOT, "README.md")
old = s = open(path).read()
for key, row in rows.items():
    i = s.index(key)
    s = s[:i] + row + s[s.index("\n", i):]
print(f"scan: bench bursts {scan['ms']:.4f} ms, from idle {sq['mean_ms']:.4f} (three queues {sq['mean_ms_of_the_three_queues']}), slowest group {sq['slowest_group_over_settled']:.2f} x settled; "
      f"rocprofv3 mean {c5['avg_ns'] / 1e6:.4f} ms = {pc(c5['frac_of_8tb_s'])} %, first launch {c5['launch_series_us'][0] / settled:.3f} x settled, later ones <= {max(c5['launch_series_us'][1:]) / settled:.3f} x")
print(f"count: bench from idle {cq['mean_ms']:.4f} ms; rocprofv3 mean {c5c['avg_ns'] / 1e6:.4f} ms = {pc(c5c['frac_of_8tb_s'])} %, last 16 {c5c['last16_avg_ns'] / 1e3:.1f} us, slowest {c5c['max_ns'] / 1e3:.1f} us")
print("(the scan and count rows of the table and the paragraph under it carry ranges over several boxes: edit them by hand from the two lines above)")
if "--check" in sys.argv:
    sys.exit(0 if s == old else 1)
open(path, "w").write(s)
