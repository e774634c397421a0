#!/usr/bin/env python3
"""One compact row per bench.py JSON line (for box-to-box tables in profiles/)."""
import json
import sys

for path in sys.argv[1:]:
    d = json.load(open(path))
    hp = d.get("host_path", {})
    sp = d.get("stream_probe_gb_s", {}).get("sustained", {})
    pp = d.get("shape_probe_pair", {})
    print(f"{d['value']:8.1f} Gbases/s  step {d['ms_per_step']:.4f} ms (same shapes, no arithmetic, no events: {pp.get('ms_per_pair_without_events')})  roofline {d['roofline']['kernel'].split('_')[0]} {d['roofline']['frac']:.3f}  enc {d['roofline_encode']['frac']:.3f} dec {d['roofline_decode']['frac']:.3f} step {d['roofline_step']['frac']:.3f} | "
          f"kmer {d['kmer_batch']['roofline']['frac']:.3f} win {d['kmer_windows']['roofline']['frac']:.3f} scan {d['kmer_hdist_scan']['roofline']['frac']:.3f} hdist {d['hdist_bulk']['roofline']['frac']:.3f} "
          f"counts {d['base_counts']['roofline']['frac']:.3f} query {d['hdist_query']['roofline']['frac']:.3f} split {d['split_packed']['roofline']['frac']:.3f} | "
          f"plan {d['reads_batch']['encode_frac']:.3f}/{d['reads_batch']['decode_frac']:.3f} tables {d['reads_batch_tables']['encode_frac']:.3f}/{d['reads_batch_tables']['decode_frac']:.3f} "
          f"fixed {d['reads_fixed']['encode_frac']:.3f}/{d['reads_fixed']['decode_frac']:.3f} | host enc {hp.get('encode_frac_of_pinned_h2d')} dec {hp.get('decode_frac_of_pinned_d2h')} "
          f"kmer {hp.get('kmer_batch_host', {}).get('frac_of_pinned_h2d')} scan {hp.get('kmer_scan_host', {}).get('gwindows_s')} Gw/s numa {hp.get('pipe', {}).get('gpu_numa_node')} | "
          f"sustained read {sp.get('read')} fill {sp.get('fill_nt')} copy {sp.get('copy')} | cpu {d.get('cpu_baseline', {}).get('value')}")
