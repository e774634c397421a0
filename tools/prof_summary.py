#!/usr/bin/env python3
"""Summarise a tools/prof.sh output directory into profiles/<tag>_summary.md and
profiles/hbm_traffic.json (read by bench.py for roofline.traffic).

HBM bytes per launch follow MI355X_MICROARCH.md section HBM: FETCH_SIZE and WRITE_SIZE are in
KiB, collected in separate --pmc passes; on gfx950 FETCH_SIZE reports exactly half of the
bytes of a coalesced streaming read, so it is doubled; WRITE_SIZE is taken as is."""
import collections
import csv
import datetime
import glob
import shutil
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
    stats = list(csv.DictReader(open((glob.glob(os.path.join(src, "trace", "**", "trace_kernel_stats.csv"), recursive=True) or [os.path.join(src, "trace", "trace_kernel_stats.csv")])[0])))
    avg = {}
    for r in stats:
        k = kernel_short(r["Name"])
        avg[k] = float(r["AverageNs"])
        lines.append(f"| `{k[:90]}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | {float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} | {r['Percentage']} |")
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        path = (glob.glob(os.path.join(src, sub, "**", "pmc_counter_collection.csv"), recursive=True) or [None])[0]
        if not path:
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
    # ---- BASELINE configs 3 and 5 alone at full size (tools/run_cfg35.py under the same three passes) ----
    side = {}
    CFG = {"cfg3": ("kmer_dense_kernel", 10**8 * 39, "10^8 dense 31-mers: 39 B per k-mer (31 read + 8 written)"),
           "cfg5": ("kmer_scan", 2 * (10**9 - 30), "10^9-base scan: 2 B per window (1 read + 1 written)"),  # kmer_scan_mfma_kernel since round 5, kmer_scan2_kernel before
           "cfg5count": ("kmer_count", 10**9 - 30, "10^9-base scan, fused d <= tau count: 1 B per window (read only)")}  # kmer_count3_mfma_kernel (round 5)
    for cfg, (kname, alg, what) in CFG.items():
        tpath = glob.glob(os.path.join(src, cfg + "_trace", "**", "trace_kernel_stats.csv"), recursive=True)
        if not tpath:
            continue
        row = next((r for r in csv.DictReader(open(tpath[0])) if kname in r["Name"]), None)
        if row is None:
            continue
        vals = {}
        for sub, ctr in ((cfg + "_pmc_fetch", "FETCH_SIZE"), (cfg + "_pmc_write", "WRITE_SIZE")):
            ppath = glob.glob(os.path.join(src, sub, "**", "pmc_counter_collection.csv"), recursive=True)
            v = [float(r["Counter_Value"]) for r in csv.DictReader(open(ppath[0])) if r["Counter_Name"] == ctr and kname in r["Kernel_Name"]] if ppath else []
            vals[ctr] = sum(v) / len(v) if v else None
        avg_ns = float(row["AverageNs"])
        # per-launch series of the queue (the VALU-bound scan runs slower while the chip's power management settles: DESIGN 3.4)
        series = []
        kpath = glob.glob(os.path.join(src, cfg + "_trace", "**", "trace_kernel_trace.csv"), recursive=True)
        if kpath:
            series = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(kpath[0])) if kname in r["Kernel_Name"]]
        settled = sum(series[-16:]) / 16 if len(series) >= 32 else None
        hbm_bytes = (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024 if vals["FETCH_SIZE"] is not None and vals["WRITE_SIZE"] is not None else None
        side[cfg] = {"kernel": kernel_short(row["Name"])[:60], "calls": int(row["Calls"]), "avg_ns": avg_ns, "min_ns": float(row["MinNs"]), "max_ns": float(row["MaxNs"]),
                     "algorithmic_bytes_per_launch": alg, "achieved_gb_s": round(alg / avg_ns, 1), "frac_of_8tb_s": round(alg / avg_ns / 8000, 4),
                     "first4_avg_ns": sum(series[:4]) / 4 if len(series) >= 4 else None, "last16_avg_ns": settled,
                     "last16_frac_of_8tb_s": round(alg / settled / 8000, 4) if settled else None,
                     "launch_series_us": [round(x / 1e3, 1) for x in series],
                     "hbm_bytes_per_launch": hbm_bytes, "traffic_over_algorithmic": round(hbm_bytes / alg, 4) if hbm_bytes else None, "what": what}
        if not lines[-1] == "":
            lines.append("")
        lines += [f"## {cfg}: {what}", "", f"`{kernel_short(row['Name'])[:100]}`: {row['Calls']} launches alone at full size (tools/run_cfg35.py {cfg}), avg {avg_ns/1e3:.1f} us "
                  f"(min {float(row['MinNs'])/1e3:.1f}, max {float(row['MaxNs'])/1e3:.1f}) -> {alg/avg_ns:.1f} GB/s on algorithmic bytes = {alg/avg_ns/80:.1f} % of 8 TB/s; "
                  + (f"HBM traffic (2*FETCH_SIZE + WRITE_SIZE) * 1024 = {hbm_bytes:.4g} B per launch = {hbm_bytes/alg:.4f} x algorithmic" if hbm_bytes else "no PMC pass"),
                  (f"Queue of {len(series)} launches: first 4 avg {sum(series[:4])/4e3:.1f} us, last 16 avg {settled/1e3:.1f} us = {alg/settled/80:.1f} % of 8 TB/s; series (us): " + " ".join(f"{x/1e3:.0f}" for x in series)) if settled else "", ""]
    # the raw CSVs the numbers above come from, tracked under profiles/<tag>_rocprof/
    dst = os.path.join(root, "profiles", f"{tag}_rocprof")
    os.makedirs(dst, exist_ok=True)
    copies = [("trace/**/trace_kernel_stats.csv", "kernel_stats.csv"), ("trace_full/**/trace_kernel_stats.csv", "kernel_stats_full_run.csv"),
              ("pmc_fetch/**/pmc_counter_collection.csv", "pmc_fetch.csv"), ("pmc_write/**/pmc_counter_collection.csv", "pmc_write.csv")]
    for cfg in CFG:
        copies += [(f"{cfg}_trace/**/trace_kernel_stats.csv", f"kernel_stats_{cfg}.csv"), (f"{cfg}_pmc_fetch/**/pmc_counter_collection.csv", f"pmc_fetch_{cfg}.csv"),
                   (f"{cfg}_pmc_write/**/pmc_counter_collection.csv", f"pmc_write_{cfg}.csv")]
    for pat, name in copies:
        hit = glob.glob(os.path.join(src, pat), recursive=True)
        if hit:
            shutil.copy(hit[0], os.path.join(dst, name))
    os.makedirs(os.path.dirname(out_md), exist_ok=True)
    open(out_md, "w").write("\n".join(lines) + "\n")
    try:
        commit = subprocess.run(["git", "-C", root, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except Exception:  # noqa: BLE001
        commit = None
    # the identity of the binary that WAS PROFILED: the bench line of the trace pass carries the hash its library reports for itself;
    # only when that is missing (older logs) the hash of the sources beside this script, which is right only if they are unchanged
    sha, sha_src = csrc_sha16(root), "sources on disk when the summary was written"
    try:
        import re
        m = re.search(r'"library_csrc_sha16": "([0-9a-f]{16})"', open(os.path.join(src, "trace.log")).read())
        if m:
            sha, sha_src = m.group(1), "bitnuc_version() of the profiled library (bench line in trace.log)"
    except OSError:
        pass
    json.dump({"tag": tag, "commit": commit, "date": datetime.date.today().isoformat(), "csrc_sha16": sha, "csrc_sha16_from": sha_src,
               "encode_bytes_per_launch": enc, "decode_bytes_per_launch": dec,
               "encode_avg_ns": enc_us, "decode_avg_ns": dec_us, **side,
               "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE counts half of a coalesced stream)"},
              open(os.path.join(root, "profiles", "hbm_traffic.json"), "w"), indent=1)
    print(open(out_md).read())


if __name__ == "__main__":
    main()
