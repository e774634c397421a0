#!/usr/bin/env python3
"""Summarise a tools/prof.sh output directory into profiles/<tag>_summary.md and
profiles/hbm_traffic.json (read by bench.py for roofline.traffic).

HBM bytes per launch follow MI355X_MICROARCH.md section HBM: FETCH_SIZE and WRITE_SIZE are in
KiB, collected in separate --pmc passes; on gfx950 FETCH_SIZE reports exactly half of the
bytes of a coalesced streaming read, so it is doubled; WRITE_SIZE is taken as is."""
import collections
import csv
import datetime
import hashlib
import json
import os
import subprocess
import sys


def csrc_sha16(root):
    """Same identity bench.py reports and the library carries (bitnuc_amd.build.csrc_sha16): a stored traffic figure is only valid
    for these kernel sources."""
    sys.path.insert(0, root)
    from bitnuc_amd import build
    return build.csrc_sha16()


def kernel_short(name):
    n = name.split("(")[0]
    return n.replace("void ", "").replace("bitnuc_dev::", "")


def main():
    src, tag = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_md = os.path.join(root, "profiles", f"{tag}_summary.md")
    lines = [f"# rocprofv3 summary {tag}", "", "Command: `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-traffic --no-extras` (the timed step only)",
             "(PMC passes: same command with `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, separately; no trace domains combined with --pmc.)", "",
             "## kernel stats (--kernel-trace --stats)", "", "| kernel | calls | avg us | min us | max us | % |", "|---|---|---|---|---|---|"]
    lines[3] = "(PMC passes: same command with `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, separately; no trace domains combined with --pmc.  `kernel_stats_full_run.csv`: the same trace of the default run, side measurements included.)"
    stats = list(csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_stats.csv"))))
    avg = {}
    for r in stats:
        k = kernel_short(r["Name"])
        avg[k] = float(r["AverageNs"])
        lines.append(f"| `{k[:90]}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | {float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} | {r['Percentage']} |")
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        path = os.path.join(src, sub, "pmc_counter_collection.csv")
        if not os.path.exists(path):
            continue
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == ctr:
                pmc[kernel_short(r["Kernel_Name"])][ctr].append(float(r["Counter_Value"]))
    lines += ["", "## HBM traffic per launch (PMC, corrected per MI355X_MICROARCH.md)", "",
              "| kernel | launches | FETCH_SIZE KiB (raw) | WRITE_SIZE KiB | HBM bytes/launch = (2*FETCH + WRITE)*1024 |", "|---|---|---|---|---|"]
    traffic = {}
    for k, d in pmc.items():
        if not any(t in k for t in ("encode_kernel", "decode_kernel", "kmer", "batch", "fixed")):
            continue
        f = sum(d["FETCH_SIZE"]) / max(1, len(d["FETCH_SIZE"]))
        w = sum(d["WRITE_SIZE"]) / max(1, len(d["WRITE_SIZE"]))
        b = (2 * f + w) * 1024
        traffic[k] = b
        lines.append(f"| `{k[:90]}` | {len(d['FETCH_SIZE'])} | {f:.1f} | {w:.1f} | {b:.4g} |")
    enc = next((v for k, v in traffic.items() if k.startswith("encode_kernel")), None)
    dec = next((v for k, v in traffic.items() if k.startswith("decode_kernel")), None)
    lines += ["", f"Algorithmic bytes per launch (10^9 bases x 1.25 B): 1.25e9.  encode HBM/algorithmic = {enc/1.25e9:.4f}; decode = {dec/1.25e9:.4f}" if enc and dec else ""]
    enc_us = next((v for k, v in avg.items() if k.startswith("encode_kernel")), None)
    dec_us = next((v for k, v in avg.items() if k.startswith("decode_kernel")), None)
    if enc_us and dec_us:
        lines += ["", f"Achieved (algorithmic 1.25e9 B / avg duration): encode {1.25e9/enc_us:.1f} GB/s = {1.25e9/enc_us/8000*100:.1f}% of 8 TB/s; "
                      f"decode {1.25e9/dec_us:.1f} GB/s = {1.25e9/dec_us/8000*100:.1f}% of 8 TB/s"]
    os.makedirs(os.path.dirname(out_md), exist_ok=True)
    open(out_md, "w").write("\n".join(lines) + "\n")
    try:
        commit = subprocess.run(["git", "-C", root, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except Exception:  # noqa: BLE001
        commit = None
    json.dump({"tag": tag, "commit": commit, "date": datetime.date.today().isoformat(), "csrc_sha16": csrc_sha16(root),
               "encode_bytes_per_launch": enc, "decode_bytes_per_launch": dec,
               "encode_avg_ns": enc_us, "decode_avg_ns": dec_us,
               "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE counts half of a coalesced stream)"},
              open(os.path.join(root, "profiles", "hbm_traffic.json"), "w"), indent=1)
    print(open(out_md).read())


if __name__ == "__main__":
    main()
