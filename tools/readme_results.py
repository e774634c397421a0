#!/usr/bin/env python3
"""The results table of README.md from a bench.py JSON line: python tools/readme_results.py profiles/r04_bench_n1.json"""
import json
import sys

d = json.load(open(sys.argv[1]))


def g(k, *ks):
    v = d[k]
    for kk in ks:
        v = v[kk]
    return v


q = g("kmer_hdist_scan", "one_queue_of_64")
rows = [
    ("bulk encode, 10⁹ bases (cfg 2)", f'{g("encode_gbases_s"):.0f} Gbases/s, {g("roofline_encode","avg_launch_ms"):.3f} ms; PMC traffic {g("roofline_encode","traffic_over_algorithmic"):.4f} × algorithmic', f'{100*g("roofline_encode","frac"):.1f}'),
    ("bulk decode, 10⁹ bases (cfg 2), input cache-cold", f'{g("decode_gbases_s"):.0f} Gbases/s, {g("roofline_decode","avg_launch_ms"):.3f} ms; PMC {g("roofline_decode","traffic_over_algorithmic"):.4f} ×', f'{100*g("roofline_decode","frac"):.1f}'),
    ("encode + decode step (`bench.py` `value` = encoded + decoded bases)", f'**{g("value"):.0f} Gbases/s**, {g("ms_per_step"):.4f} ms/step', f'{100*g("roofline_step","frac"):.1f}'),
    ("10⁸ dense 31-mers → u64 (cfg 3)", f'{g("kmer_batch","gkmers_s"):.0f} G k-mers/s, {g("kmer_batch","ms"):.3f} ms', f'{100*g("kmer_batch","roofline","frac"):.1f}'),
    ("sliding 31-mer pack + Hamming scan, 10⁹ bases (cfg 5)", f'{g("kmer_hdist_scan","gwindows_s")/1e3:.2f} T windows/s, {g("kmer_hdist_scan","ms"):.3f} ms in sustained bursts; one queue of 64 on the busy chip: mean {q["mean_ms"]:.3f}, settled {q["last16_ms"]:.3f} ms (VALU-issue bound: follows the clock, `DESIGN.md` §3.4)', f'{100*g("kmer_hdist_scan","roofline","frac"):.1f}'),
    ("every 31-base window → u64 (stride 1), 10⁹ bases", f'{g("kmer_windows","gwindows_s")/1e3:.2f} T windows/s, {g("kmer_windows","ms"):.3f} ms (≈ the fill rate of its 9 GB footprint)', f'{100*g("kmer_windows","roofline","frac"):.1f}'),
    ("`hdist` of two 10⁹-base packed buffers", f'{g("hdist_bulk","ms"):.4f} ms', f'{100*g("hdist_bulk","roofline","frac"):.1f}'),
    ("A/C/G/T counts of 10⁹ packed bases (no decode)", f'{g("base_counts","ms"):.4f} ms', f'{100*g("base_counts","roofline","frac"):.1f}'),
    ("one packed 32-mer vs 3.1·10⁷ packed 32-mers (`hdist_query`)", f'{g("hdist_query","ms"):.4f} ms', f'{100*g("hdist_query","roofline","frac"):.1f}'),
    ("`split_packed` of 10⁹ packed bases, mid-word", f'{g("split_packed","ms"):.4f} ms', f'{100*g("split_packed","roofline","frac"):.1f}'),
    ("6.7 M × 150-base reads with a layout plan, encode / decode", f'{g("reads_batch","encode_ms"):.4f} / {g("reads_batch","decode_ms"):.4f} ms', f'{100*g("reads_batch","encode_frac"):.1f} / {100*g("reads_batch","decode_frac"):.1f} (with the plan\'s own bytes: {100*g("reads_batch","encode_frac_with_plan_bytes"):.1f} / {100*g("reads_batch","decode_frac_with_plan_bytes"):.1f})'),
    ("same batch from the offset tables alone (tables counted: every call reads them)", f'{g("reads_batch_tables","encode_ms"):.4f} / {g("reads_batch_tables","decode_ms"):.4f} ms', f'{100*g("reads_batch_tables","encode_frac"):.1f} / {100*g("reads_batch_tables","decode_frac"):.1f} (bases + words only: {100*g("reads_batch_tables","encode_frac_without_tables"):.1f} / {100*g("reads_batch_tables","decode_frac_without_tables"):.1f})'),
    ("same reads, `encode_fixed` / `decode_fixed` (no tables)", f'{g("reads_fixed","encode_ms"):.4f} / {g("reads_fixed","decode_ms"):.4f} ms', f'{100*g("reads_fixed","encode_frac"):.1f} / {100*g("reads_fixed","decode_frac"):.1f}'),
    ("CPU: the reference's AVX2 algorithm restated in C, 1 core of the box's " + g("cpu_baseline", "cpu").replace(" 64-Core Processor", ""), f'{g("cpu_baseline","value"):.2f} Gbases/s (`-march=x86-64-v3`), {g("cpu_baseline","native_value"):.2f} (`-march=native`); 16 cores: {g("cpu_baseline","all_cores","value"):.1f}', "—"),
]
print("| Workload (BASELINE config) | Result | % of 8 TB/s (algorithmic bytes) |\n|---|---|---|")
for r in rows:
    print("| " + " | ".join(r) + " |")
p = g("parity_vs_oracle")
print(f'\n`parity_vs_oracle`: {p["encode_words_compared"]:,} words and {p["decode_bases_compared"]:,} bases of the timed step compared with `oracle/bitnuc_avx2.c`\'s output, ok = {p["ok"]}; library `csrc:{g("config","library_csrc_sha16")}` = sources `csrc:{g("config","csrc_sha16")}`.')
