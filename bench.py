#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE configs[1]:
"Gbases/s encode+decode on 10^9-base synthetic; % of HBM3E roofline".

A step = one pass of the hot path over one batch: bulk-encode 10^9 device-resident
ASCII bases to 2-bit words and bulk-decode 10^9 bases of packed words (two kernel launches
through the C ABI's device-pointer entry points).  Three buffer sets rotate and the decode
reads the set encoded two steps earlier, so every input of every kernel comes from HBM, not
from the 256 MiB Infinity Cache (--warm-decode gives the literal same-step round trip).  Weak scaling: every rank (one process per GPU)
owns its own 10^9-base shard; there is no data-path collective (the optional
concatenating all-gather of config 4 is timed separately, outside the step).

value = bases pushed through the codec per second, whole job:
        N_gpus * (10^9 encoded + 10^9 decoded) / step time / 1e9   [Gbases/s]
        (x 1.25 algorithmic bytes per base = aggregate algorithmic GB/s).
The two kernels of a step are timed live by HIP events on the launch stream, on every 10th timed step (--event-every; on every step
the three markers cost 2.5 % of the step itself, profiles/r03_ab_event_every.txt); `config.hip_events` in the line says what was done.

What the line carries beside the contract's keys: `roofline` (the step's dominant kernel; `traffic` = HBM bytes per launch from live
rocprofv3 PMC passes), `cpu_baseline` (oracle/bitnuc_avx2.c on one core: `value` from the tracked portable build, `native_value` from a
-march=native build made on this host), `parity_vs_oracle` (EVERY word and base the timed step left on rank 0's GPU against the
oracle's output for the same seeded stream; a mismatch makes the run exit 3), `config.library_csrc_sha16` (the hash the timed
library reports for its sources) and, as the LAST key, `configs` (the fraction of every BASELINE config measured in this run).

Process model
  python bench.py --gpus N            (no launcher)  the parent touches no GPU: it starts
        `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child process,
        relays rank 0's single JSON line and exits with the child's return code.
  torchrun ... bench.py --gpus N      (the driver's form) WORLD_SIZE is set: this process is a rank.
Failure is never silent: a stalled collective prints the line with "stalled": true and exits 3;
RCCL that cannot come up on EVERY rank (decided collectively over a gloo group) is reported as
top-level "rccl_ok": false.
"""
import argparse
import datetime
import hashlib
import json
import os
import socket
import statistics
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s)
BYTES_PER_BASE = 1.25          # encode: 1 read + 0.25 write; decode: 0.25 read + 1 write
SEED = 0xB17C0DE
METRIC = "Gbases/s encode+decode on 10^9-base synthetic; % of HBM3E roofline"
EXIT_STALLED = 3


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--bases", type=int, default=10**9, help="bases per GPU per step (BASELINE configs[1]: 1e9)")
    ap.add_argument("--event-every", type=int, default=10, help="record the per-kernel HIP events on every k-th timed step (1 = every step, the method of rounds 1-2: three event records per step cost 2.5 %% of the step, profiles/r03_ab_event_every.txt); never fewer than 4 sampled steps")
    ap.add_argument("--separate-allocations", action="store_true", help="allocate the rotating buffers one by one (rounds 1-3) instead of carving them from one allocation (profiles/r04_ab_arena.txt: separate allocations vary by +-4 %% per buffer with where the allocator puts them)")
    ap.add_argument("--rotate", type=int, default=3, help="buffer sets rotated so the 256 MiB Infinity Cache cannot serve a step (>= 2)")
    ap.add_argument("--evidence-build", action="store_true", help="load libbitnuc_hip_sweep.so (every kernel variant) instead of the product library: for --enc-variant / --dec-variant studies only")
    ap.add_argument("--enc-variant", type=int, default=-1)
    ap.add_argument("--dec-variant", type=int, default=-1)
    ap.add_argument("--grid-mult", type=int, default=-1)
    ap.add_argument("--cpu-sample", type=int, default=10**9, help="bases timed on the CPU (default: the whole configs[1] workload)")
    ap.add_argument("--cpu-reps", type=int, default=15, help="repetitions of the 10^9-base CPU round trip (median reported): about 12 s of single-core work")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the config-3 (k-mer batch) and config-5 (scan) side measurements")
    ap.add_argument("--no-traffic", action="store_true", help="skip the live rocprofv3 --pmc pass that measures roofline.traffic (N=1 only)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); gloo only to rehearse the N>1 control flow")
    ap.add_argument("--force-dist", action="store_true", help="rehearsal only: initialise torch.distributed and run the collectives even at world size 1")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0 (1-GPU box)")
    ap.add_argument("--rehearse-cpu", action="store_true", help="rehearsal only (CPU, no GPU touched): exercise launcher, process groups, watchdog and the JSON relay; value is null")
    ap.add_argument("--dist-timeout", type=float, default=600.0, help="seconds a rank may spend in process-group setup / a collective before it reports a stall")
    ap.add_argument("--warm-decode", action="store_true", help="decode the words the same step just encoded (a literal round trip: 20 %% of decode's input then comes from the Infinity Cache)")
    ap.add_argument("--probe", action="store_true", help="also time pure streaming kernels (the box's own HBM rates) when --no-extras is given")
    ap.add_argument("--traffic-child", choices=["codec"], default=None, help=argparse.SUPPRESS)  # the workload run under rocprofv3 --pmc
    ap.add_argument("--inject-stall", type=int, default=-1, help=argparse.SUPPRESS)  # test hook: this rank never joins the side collectives
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------
# parent mode: start the ranks (no GPU call in this process)
# ---------------------------------------------------------------------------------------------
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args, argv):
    """`python bench.py --gpus N` with no launcher: spawn N fresh rank processes through
    torch.distributed.run, relay rank 0's JSON line, return the children's worst return code."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    print("[bench] launching: " + " ".join(cmd), file=sys.stderr)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:  # rank 0 prints exactly one JSON line; anything else on stdout goes to stderr
        if out.startswith('{"metric"') and line is None:
            line = out.rstrip("\n")
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    else:
        print(json.dumps({"metric": METRIC, "value": None, "unit": "Gbases/s", "n_gpus": args.gpus, "error": f"ranks exited with code {rc} before printing a line"}), flush=True)
    return rc if rc != 0 else (0 if line is not None else 1)


# ---------------------------------------------------------------------------------------------
# roofline.traffic: HBM bytes per launch from PMC counters, measured for THIS run's kernels
# ---------------------------------------------------------------------------------------------
def csrc_sha16():
    """Identity of the kernel sources (bitnuc_amd.build.csrc_sha16: csrc/ + include/bitnuc_hip.h).  The library carries the same
    hash (compiled in, bitnuc_version()); a stored traffic figure is only valid for the same sources."""
    from bitnuc_amd import build as bn_build
    return bn_build.csrc_sha16()


def library_identity():
    """What was timed: the hash the loaded product library reports for itself next to the hash of the sources on disk (equal, or
    bitnuc_amd refuses to load it), and whether this run had to rebuild it."""
    from bitnuc_amd import _lib, build as bn_build
    ver = _lib.load().bitnuc_version().decode()
    return {"csrc_sha16": bn_build.csrc_sha16(), "library_csrc_sha16": ver.split("csrc:")[-1].split()[0] if "csrc:" in ver else None,
            "library": bn_build.LAST_ACTION.get(bn_build.LIB, "as shipped"), "version": ver}


def git_head():
    """HEAD of the repository, or -- on a box that received a snapshot without history -- the commit the product library was
    built at (bitnuc_amd/BUILD_INFO.json, written by bitnuc_amd.build), marked as such."""
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip()
        if head:
            return head
    except Exception:  # noqa: BLE001
        pass
    try:
        with open(os.path.join(ROOT, "bitnuc_amd", "BUILD_INFO.json")) as f:
            info = json.load(f)
        return f"{info['commit']}{'+uncommitted csrc' if info.get('csrc_dirty') else ''} (library built at)"
    except Exception:  # noqa: BLE001
        return None


def measure_traffic_live(n):
    """Run the codec pair under `rocprofv3 --pmc` in child processes (FETCH_SIZE and WRITE_SIZE in
    separate passes, counters only: MI355X_MICROARCH.md section HBM) BEFORE this process touches the GPU,
    and return HBM bytes per launch per kernel: (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 counts half
    of a wide coalesced read; WRITE_SIZE is exact for 16-B-per-lane stores)."""
    import csv
    import glob
    import shutil
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return {"error": "rocprofv3 not found"}
    tmp = tempfile.mkdtemp(prefix="bitnuc_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    sums = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, ctr)
            cmd = [rocprof, "--pmc", ctr, "--output-format", "csv", "-d", out, "-o", "pmc", "--",
                   sys.executable, os.path.abspath(__file__), "--traffic-child", "codec", "--bases", str(n)]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return {"error": f"rocprofv3 --pmc {ctr} failed (rc {r.returncode}): {r.stderr[-300:]}"}
            for row in csv.DictReader(open(files[0])):
                if row.get("Counter_Name") != ctr:
                    continue
                k = row["Kernel_Name"]
                name = "encode_kernel" if "encode_kernel" in k else "decode_kernel" if "decode_kernel" in k else None
                if name:
                    sums.setdefault((name, ctr), []).append(float(row["Counter_Value"]))
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)[:300]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), measured by this run before the timed region",
           "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch (gfx950: FETCH_SIZE counts half of a wide coalesced read)"}
    for name in ("encode_kernel", "decode_kernel"):
        f, w = sums.get((name, "FETCH_SIZE")), sums.get((name, "WRITE_SIZE"))
        if not f or not w:
            return {"error": f"no PMC rows for {name}"}
        res[name] = round((2 * sum(f) / len(f) + sum(w) / len(w)) * 1024)
        res[name + "_launches"] = min(len(f), len(w))
    return res


def stored_traffic():
    """profiles/hbm_traffic.json from an earlier PMC pass, accepted only for the same kernel sources."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        t = json.load(open(path))
    except Exception:  # noqa: BLE001
        return {"error": "no live PMC pass and no profiles/hbm_traffic.json"}
    if t.get("csrc_sha16") != csrc_sha16():
        return {"error": "profiles/hbm_traffic.json was measured on different kernel sources (csrc_sha16 mismatch): refused"}
    return {"source": f"profiles/hbm_traffic.json (commit {t.get('commit')}, {t.get('date')}): NOT measured in this run",
            "encode_kernel": round(t["encode_bytes_per_launch"]), "decode_kernel": round(t["decode_bytes_per_launch"])}


def generator_words(torch, first_word, count, seed, dev):
    """Closed form of the packed words of the seeded stream (bitnuc_nucgen_dev: base(i) is 2-bit field i % 32 of
    splitmix64(seed + (i / 32 + 1) * 0x9E3779B97F4A7C15), so the generator's words ARE the packed words of upper-case input):
    words [first_word, first_word + count) as int64, computed with torch integer arithmetic -- independent of the library."""
    idx = torch.arange(first_word + 1, first_word + count + 1, dtype=torch.int64, device=dev)
    z = idx * (-7046029254386353131) + seed  # 0x9E3779B97F4A7C15 as i64, wraps mod 2^64

    def lsr(x, sh):
        return (x >> sh) & ((1 << (64 - sh)) - 1)
    z = (z ^ lsr(z, 30)) * (-4658895280553007687)   # 0xBF58476D1CE4E5B9
    z = (z ^ lsr(z, 27)) * (-7723592293110705685)   # 0x94D049BB133111EB
    return z ^ lsr(z, 31)


def all_slots_check(torch, full, lw, world, slot_first_word, seed, rehearse=False, chunk=1 << 22):
    """Config 4's result held against what it must be, EVERY slot of the gathered buffer (not this rank's own, not one run against
    another): slot s = the lw packed words of rank s's shard = the generator's closed form from word slot_first_word(s) on
    (rehearsal without a GPU: arange(lw) + s).  Returns {"all_slots_ok": bool, "first_bad_slot": s or None}."""
    for s_ in range(world):
        part = full[s_ * lw:(s_ + 1) * lw]
        if rehearse:
            good = bool(torch.equal(part, torch.arange(lw, dtype=part.dtype, device=part.device) + s_))
        else:
            good = True
            for a in range(0, lw, chunk):
                b = min(lw, a + chunk)
                if not torch.equal(part[a:b], generator_words(torch, slot_first_word(s_) + a, b - a, seed, part.device)):
                    good = False
                    break
        if not good:
            return {"all_slots_ok": False, "first_bad_slot": s_}
    return {"all_slots_ok": True, "first_bad_slot": None}


def carve_buffers(torch, dev, n, nw, R, separate):
    """The R rotating buffer sets (ASCII in, packed words, ASCII out).  Default: ONE allocation, every buffer at a 2 MiB boundary --
    how a resident pipeline lays its buffers out, and what makes the codec's per-launch time independent of where an allocator
    happens to put each buffer (tools/ab_placement.py, tools/ab_arena.py, profiles/r04_ab_arena.txt: inside one allocation the
    kernels do not care where their buffers sit; nine separate allocations vary by +-4 % per buffer, 1 % on the step).  Returns
    torch tensors (views of the arena) so that everything downstream is unchanged."""
    if separate:
        return ([torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(R)], [torch.empty(nw, dtype=torch.int64, device=dev) for _ in range(R)],
                [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(R)], None)
    A = 2 << 20

    def up(x):
        return (x + A - 1) // A * A
    arena = torch.empty(R * (2 * up(n) + up(8 * nw)) + A, dtype=torch.uint8, device=dev)
    off = up(arena.data_ptr()) - arena.data_ptr()
    seqs, words, backs = [], [], []
    for _ in range(R):
        seqs.append(arena[off:off + n]); off += up(n)
    for _ in range(R):
        words.append(arena[off:off + 8 * nw].view(torch.int64)); off += up(8 * nw)
    for _ in range(R):
        backs.append(arena[off:off + n]); off += up(n)
    return seqs, words, backs, arena


def traffic_child(args):
    """The workload profiled by measure_traffic_live: the same two launches as a step, cache-cold rotation."""
    import torch
    import bitnuc_amd
    dev = torch.device("cuda", 0)
    n, nw = args.bases, (args.bases + 31) // 32
    stream = torch.cuda.current_stream()
    ctx = bitnuc_amd.Context(0, stream=stream.cuda_stream)
    R = 3
    seqs, words, backs, _arena = carve_buffers(torch, dev, n, nw, R, args.separate_allocations)
    for r in range(R):
        ctx.nucgen_dev(seqs[r], n, SEED + r)
    for i in range(2 * R):
        ctx.encode_dev(seqs[i % R], n, words[i % R])
        ctx.decode_dev(words[(i + 1) % R], nw, n, backs[(i + 1) % R])
    ctx.sync()
    ctx.close()


# ---------------------------------------------------------------------------------------------
# CPU baseline (oracle/bitnuc_avx2.c) -- the checker timed beside the product, never inside it
# ---------------------------------------------------------------------------------------------
def cpu_baseline(n_sample, reps, all_cores, seed=SEED, gpu=None):
    """Reference-algorithm restatement (oracle/bitnuc_avx2.c, the reference's AVX2 path as
    written) timed on this host.  kind = "port": the Rust reference cannot be built here.
    gpu = {"words": ndarray u64, "back": ndarray u8} -- what the timed GPU launches left for the SAME seeded stream -- makes the
    line's parity claim literal: the oracle's words for the stream (it encodes them here anyway) are compared with the GPU's word
    for word, and the GPU's decoded bases with the stream (the oracle's own decode is checked against the stream inside
    orc_avx2_time_roundtrip).  The result is returned under "parity"; the oracle is the checker here, never the thing shipped."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_py
    seq = oracle_py.nucgen(n_sample, seed)

    def time_with(L):
        enc, dec = [], []
        for _ in range(reps):
            e, d = oracle_py.avx2_time_roundtrip(seq, L)
            enc.append(e)
            dec.append(d)
        return statistics.median(enc), statistics.median(dec)
    e1, d1 = time_with(None)
    out = {"value": round(2 * n_sample / (e1 + d1) / 1e9, 4), "unit": "Gbases/s", "cores": 1, "kind": "port",
           "sample": f"{n_sample:.3g} bases of the same seeded stream, encode+decode, median of {reps}",
           "encode_gbases_s": round(n_sample / e1 / 1e9, 4), "decode_gbases_s": round(n_sample / d1 / 1e9, 4),
           "what": "C restatement of the reference's AVX2 path as written (oracle/bitnuc_avx2.c); "
                   "the reference itself is single-threaded Rust and cannot be built in this image",
           "build_flags": oracle_py.build_flags() if hasattr(oracle_py, "build_flags") else None}
    # BASELINE.md section 3 / the reference's .cargo/config.toml:1-2 time a target-cpu=native build: the same sources compiled
    # -march=native on THIS host (into a temporary directory), timed the same way, reported beside the portable build's figure
    L_nat, nat_flags = oracle_py.native_lib()
    if L_nat is not None:
        e2, d2 = time_with(L_nat)
        out.update({"native_value": round(2 * n_sample / (e2 + d2) / 1e9, 4), "native_encode_gbases_s": round(n_sample / e2 / 1e9, 4),
                    "native_decode_gbases_s": round(n_sample / d2 / 1e9, 4), "native_build_flags": nat_flags})
    else:
        out.update({"native_value": None, "native_build_flags": f"-march=native build failed on this host: {nat_flags}"})
    if gpu is not None:
        try:
            o_words = oracle_py.encode(seq, avx2=True)
            nws = o_words.size
            g_words, g_back = gpu["words"][:nws], gpu["back"][:n_sample]
            if g_words.size != nws or g_back.size != n_sample:
                out["parity"] = {"ok": None, "note": f"the GPU step ran on fewer bases than the CPU sample ({g_back.size} < {n_sample}): not compared"}
            else:
                ok_w, ok_b = bool(np.array_equal(g_words, o_words)), bool(np.array_equal(g_back, seq))
                # localise a failure: 64-bit sum per 1 MiB block of words
                first_bad = None
                if not ok_w:
                    first_bad = int(np.nonzero(g_words != o_words)[0][0])
                out["parity"] = {"ok": ok_w and ok_b, "encode_words_compared": int(nws), "encode_ok": ok_w, "decode_bases_compared": int(n_sample), "decode_ok": ok_b,
                                 "first_differing_word": first_bad, "stream_seed": hex(seed),
                                 "how": "the words / bases the timed launches left on the GPU (copied to the host after the timed region) against the words "
                                        "oracle/bitnuc_avx2.c produces for the same seeded stream and against the stream itself, every element"}
        except Exception as e:  # noqa: BLE001
            out["parity"] = {"ok": None, "error": repr(e)[:300]}
    if all_cores:
        # a one-GPU box's CPU share is 16 cores (os.cpu_count() reports the whole host)
        cores = min(len(os.sched_getaffinity(0)), 16)
        per = (n_sample // cores) // 32 * 32
        if per > 0:
            res = [None] * cores

            def work(i):
                res[i] = oracle_py.avx2_time_roundtrip(seq[i * per:(i + 1) * per], L_nat)
            t0 = time.perf_counter()
            th = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
            [t.start() for t in th]
            [t.join() for t in th]
            wall = time.perf_counter() - t0
            out["all_cores"] = {"value": round(2 * per * cores / wall / 1e9, 4), "cores": cores, "build": "native" if L_nat is not None else "portable",
                                "note": "courtesy split at 32-base boundaries over threads; the reference has no threading"}
    try:
        with open("/proc/cpuinfo") as f:
            out["cpu"] = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except Exception:  # noqa: BLE001
        pass
    return out


# ---------------------------------------------------------------------------------------------
# rank mode
# ---------------------------------------------------------------------------------------------
class Watchdog:
    """One deadline for everything a rank does together with other ranks.  When it fires the rank
    prints what it has (rank 0: the JSON line with "stalled": true) and leaves with EXIT_STALLED, so a
    hung collective is a visible failure, never rc 0."""

    def __init__(self, on_fire):
        self._on_fire = on_fire
        self._timer = None
        self.stage = "start"

    def arm(self, seconds, stage):
        self.disarm()
        self.stage = stage
        self._timer = threading.Timer(seconds, self._fire)
        self._timer.daemon = True
        self._timer.start()

    def disarm(self):
        if self._timer is not None:
            self._timer.cancel()
            self._timer = None

    def _fire(self):
        try:
            self._on_fire(self.stage)
        finally:
            os._exit(EXIT_STALLED)


def run_rank(args, real_stdout, traffic):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    state = {"line_extra": {}, "headline": None}  # headline = dict of measured step numbers once the timed loop is done
    emit_lock = threading.Lock()
    emitted = []

    def emit(line):
        with emit_lock:
            if emitted:
                return
            emitted.append(True)
            if rank == 0:
                sys.stdout.flush()
                os.write(real_stdout, (json.dumps(line) + "\n").encode())

    def on_stall(stage):
        base = {"metric": METRIC, "value": None, "unit": "Gbases/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup}
        if state["headline"] is not None:
            base = state["headline"](dict(state["line_extra"]), with_cpu=False)
        base["stalled"] = True
        base["stalled_stage"] = stage
        print(f"[bench] rank {rank}: stalled in '{stage}' after {args.dist_timeout:.0f} s; exiting {EXIT_STALLED}", file=sys.stderr)
        emit(base)

    wd = Watchdog(on_stall)

    import torch
    import torch.distributed as dist
    use_dist = world > 1 or args.force_dist
    rehearse = args.rehearse_cpu

    if not rehearse:
        import bitnuc_amd
        from bitnuc_amd import build as bn_build
        # a fresh checkout has no libbitnuc_hip.so (git-ignored): local rank 0 compiles it, the others wait for the file
        # ... and a library that travelled with the tree but was built from other sources is rebuilt the same way (the line says so)
        if local_rank == 0:
            bn_build.ensure_built()
        else:
            t_wait = time.time()
            while bn_build.is_stale(bn_build.LIB):
                if time.time() - t_wait > 600:
                    raise RuntimeError(f"{bn_build.LIB} (csrc:{bn_build.csrc_sha16()}) did not appear: there is no CPU fallback")
                time.sleep(1.0)
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
        ndev = torch.cuda.device_count()
        if args.share_gpu:
            local_rank = 0
        elif local_rank >= ndev:
            # fewer visible devices than ranks is only legitimate when a launcher narrowed each rank to its own device
            narrowed = any(os.environ.get(v) for v in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"))
            if ndev == 1 and narrowed:
                local_rank = 0
            else:
                raise SystemExit(f"rank {rank}: local rank {local_rank} but only {ndev} visible device(s); pass --share-gpu to rehearse on one GPU")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    else:
        dev = torch.device("cpu")

    # ---- process groups: gloo carries the decision, RCCL carries the collectives iff EVERY rank has it ----
    ctl = None           # group used for barrier / max-reduce / side collectives (None = default gloo group)
    control_backend = None
    rccl_ok = None       # None = not attempted (single process, or --backend gloo)
    probe_thread = None
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")  # a hung probe must not abort the process: the watchdog reports it
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        tmo = datetime.timedelta(seconds=args.dist_timeout)
        wd.arm(args.dist_timeout, "init_process_group(gloo)")
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=tmo)
        control_backend = "gloo"
        if args.backend == "nccl" and not rehearse:
            wd.arm(args.dist_timeout + 60, "rccl probe")
            probe = {}

            def try_nccl():
                try:
                    g = dist.new_group(backend="nccl", timeout=tmo)
                    t = torch.ones(1, device=dev)
                    dist.all_reduce(t, group=g)
                    torch.cuda.synchronize()
                    probe["ok"] = int(t.item()) == world
                    probe["group"] = g
                except Exception as e:  # noqa: BLE001
                    probe["err"] = repr(e)[:300]
            probe_thread = th = threading.Thread(target=try_nccl, daemon=True)
            th.start()
            th.join(min(args.dist_timeout, 180.0))
            mine = 1 if probe.get("ok") else 0
            if not mine:
                print(f"[bench] rank {rank}: RCCL probe failed: {probe.get('err', 'timed out')}", file=sys.stderr)
            flag = torch.tensor([mine], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)  # collective decision over gloo: all ranks or none
            rccl_ok = bool(flag.item())
            if rccl_ok:
                ctl = probe["group"]
                control_backend = "nccl"
        wd.disarm()
    on_gpu_collectives = use_dist and control_backend == "nccl"

    def barrier():
        if use_dist:
            if on_gpu_collectives:
                dist.barrier(group=ctl, device_ids=[local_rank])  # no rank -> device guessing inside the NCCL barrier
            else:
                dist.barrier(group=ctl)

    def fence():
        barrier()
        if not rehearse:
            torch.cuda.synchronize()

    # physical identity of every rank's device: two ranks on one GPU would double-count the weak-scaling value
    identities = None
    if not rehearse:
        p = torch.cuda.get_device_properties(dev)
        ident = str(getattr(p, "uuid", "")) or f"{getattr(p, 'pci_bus_id', '?')}:{getattr(p, 'pci_device_id', '?')}"
        ident = f"{socket.gethostname()}/{ident}"
        identities = [ident]
        if use_dist:
            wd.arm(args.dist_timeout, "all_gather(device identity)")
            identities = [None] * world
            dist.all_gather_object(identities, ident)
            wd.disarm()
        if len(set(identities)) != len(identities) and not args.share_gpu:
            raise SystemExit(f"rank {rank}: ranks share a physical GPU ({identities}); pass --share-gpu to rehearse on one GPU")

    n = args.bases
    nw = (n + 31) // 32
    R = max(2, args.rotate)

    if rehearse:
        # no GPU: the "step" is a sleep; everything around it (barriers, max-reduce, side collective, relay) is real
        def step(i, ev=None):
            time.sleep(0.001)
        ctx = stream = None
    else:
        stream = torch.cuda.current_stream()
        if args.evidence_build:
            from bitnuc_amd import build as _build
            ctx = bitnuc_amd.Context(local_rank, stream=stream.cuda_stream, lib_path=_build.ensure_built(sweep=True))
            state["line_extra"]["library"] = "evidence build (libbitnuc_hip_sweep.so): not the product"
        else:
            ctx = bitnuc_amd.Context(local_rank, stream=stream.cuda_stream)
        ctx.set_variant("force_gpu", 1)  # the bench measures kernels only; the small-input host path is reported separately
        if args.enc_variant >= 0:
            ctx.set_variant("encode", args.enc_variant)
        if args.dec_variant >= 0:
            ctx.set_variant("decode", args.dec_variant)
        if args.grid_mult >= 0:
            ctx.set_variant("grid_mult", args.grid_mult)
        seqs, words, backs, _arena = carve_buffers(torch, dev, n, nw, R, args.separate_allocations)
        for r in range(R):  # rank-disjoint slices of one seeded stream, generated in place on the device
            ctx.nucgen_dev(seqs[r], n, SEED + r, first=rank * n)
        ctx.sync()
        # Step i encodes buffer set r = i % R and decodes set (r + 1) % R, whose words were written
        # R - 1 steps earlier: >= 2 x 2.25 GB of traffic ago, so no part of the decode's input can still
        # sit in the 256 MiB Infinity Cache (decoding the words the same step just wrote would read
        # 20 % of its bytes from cache and flatter the HBM fraction).
        for r in range(R):
            ctx.encode_dev(seqs[r], n, words[r])
        ctx.sync()

        def step(i, ev=None):
            r = i % R
            d = r if args.warm_decode else (r + 1) % R
            if ev:
                ev[0].record(stream)
            ctx.encode_dev(seqs[r], n, words[r])
            if ev:
                ev[1].record(stream)
            ctx.decode_dev(words[d], nw, n, backs[d])
            if ev:
                ev[2].record(stream)

    wd.arm(args.dist_timeout, "timed loop")
    for i in range(args.warmup):
        step(i)
    if ctx:
        ctx.sync()
    ev_every = max(1, min(args.event_every, args.steps // 4))  # at least 4 sampled steps (every step when K < 8)
    events = None if rehearse else [[torch.cuda.Event(enable_timing=True) for _ in range(3)] if i % ev_every == 0 else None for i in range(args.steps)]
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, events[i] if events else None)
    fence()
    t1 = time.perf_counter()
    if ctx:
        ctx.sync()  # raises if any launch latched an InvalidBase
        # parity guard inside the bench: the last step's round trip must be the identity
        r_last = (args.steps - 1) % R if args.warm_decode else ((args.steps - 1) % R + 1) % R
        assert torch.equal(seqs[r_last], backs[r_last]), "decode(encode(x)) != x"
        # what the timed launches left for buffer set r_last (stream seed SEED + r_last), copied out now -- after the timed region,
        # before the side measurements reuse the buffers -- for the word-for-word comparison with the CPU oracle (cpu_baseline)
        if rank == 0 and not args.no_cpu_baseline and args.steps + args.warmup >= R:
            state["gpu_result"] = {"words": words[r_last].cpu().numpy().view("uint64"), "back": backs[r_last].cpu().numpy(), "seed": SEED + r_last}

    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev if on_gpu_collectives else "cpu")
    if use_dist:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX, group=ctl)
    wd.disarm()
    sec_per_step = float(elapsed.item()) / args.steps
    if events:
        enc_ms = [e[0].elapsed_time(e[1]) for e in events if e]
        dec_ms = [e[1].elapsed_time(e[2]) for e in events if e]
        enc_avg, dec_avg = sum(enc_ms) / len(enc_ms), sum(dec_ms) / len(dec_ms)
    else:
        enc_avg = dec_avg = None

    def roof(kernel, alg_bytes, ms, traffic_bytes=None, probe=None):
        gbs = alg_bytes / (ms * 1e-3) / 1e9
        r = {"kernel": kernel, "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic_bytes,
             "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(ms, 4)}
        if traffic_bytes:
            r["traffic_over_algorithmic"] = round(traffic_bytes / alg_bytes, 4)
        if probe:  # a kernel with the same access shape and no arithmetic, timed in the same sustained rotation: a rate, not a bound
            r["same_shape_no_arithmetic_gb_s"] = probe
            r["vs_same_shape_no_arithmetic"] = round(gbs / probe, 4)
        return r

    def make_line(extra, with_cpu=True):
        total_bases = world * 2 * n  # encoded + decoded, all ranks, per step
        line = {
            "metric": METRIC,
            "value": None if rehearse else round(total_bases / sec_per_step / 1e9, 2),
            "unit": "Gbases/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(sec_per_step * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: bulk encode + decode of 10^9 random bases per GPU, device-resident",
                       "bases_per_gpu_per_step": n, "bases_counted_per_step": "encoded + decoded = 2 x bases_per_gpu_per_step x n_gpus",
                       "seed": hex(SEED), "rotating_buffer_sets": R,
                       "buffers": "separate allocations" if args.separate_allocations else "carved from one allocation at 2 MiB boundaries (profiles/r04_ab_arena.txt)",
                       "hip_events": (f"per-kernel HIP events recorded on every {ev_every}th timed step ({len([e for e in (events or []) if e])} of {args.steps} steps): recording them on every step "
                                      "inserts three markers per step and costs 2.5 % of it") if ev_every > 1 else "per-kernel HIP events recorded on every timed step",
                       "decode_input": "words encoded in the same step (Infinity-Cache warm)" if args.warm_decode else f"words encoded {R - 1} steps earlier (HBM resident, cache cold)",
                       "parallelism": f"shard{world}" if world > 1 else "single",
                       "control_backend": control_backend, "devices": identities,
                       "commit": git_head(), "csrc_sha16": csrc_sha16()},
        }
        if use_dist:
            line["rccl_ok"] = rccl_ok
        if rehearse:
            line["rehearsal"] = "no GPU touched: launcher / process-group / relay rehearsal only, value is null"
            line.update(extra)
            return line
        line["config"].update({"encode_variant": ctx.get("encode"), "decode_variant": ctx.get("decode"), "grid_mult": ctx.get("grid_mult")})
        line["config"].update({k: v for k, v in library_identity().items() if k != "csrc_sha16"})  # the binary that was timed, held against the sources
        # the same measurement, three readings (value is the first): bases through the codec per
        # second (encoded + decoded), and the per-kernel rates from the HIP events of rank 0
        line.update({"codec_gbases_s": round(total_bases / sec_per_step / 1e9, 2),
                     "roundtrip_gbases_s": round(world * n / sec_per_step / 1e9, 2),
                     "encode_gbases_s": round(world * n / (enc_avg * 1e-3) / 1e9, 1),
                     "decode_gbases_s": round(world * n / (dec_avg * 1e-3) / 1e9, 1)})
        pr = extra.get("shape_probe_pair", {})
        tr = traffic or {}
        alg = n * BYTES_PER_BASE
        r_enc = roof("encode_kernel", alg, enc_avg, tr.get("encode_kernel"), pr.get("encode_shape_gb_s"))
        r_dec = roof("decode_kernel", alg, dec_avg, tr.get("decode_kernel"), pr.get("decode_shape_gb_s"))
        both = tr.get("encode_kernel") and tr.get("decode_kernel")
        r_step = roof("encode_kernel + decode_kernel (the whole step)", 2 * alg, enc_avg + dec_avg,
                      tr["encode_kernel"] + tr["decode_kernel"] if both else None)
        dominant = r_dec if dec_avg >= enc_avg else r_enc  # the kernel that takes the larger share of the step
        line["roofline"] = dict(dominant, dominant_because="largest share of the step's kernel time",
                                traffic_source=tr.get("source") or tr.get("error"))
        line["roofline_encode"], line["roofline_decode"], line["roofline_step"] = r_enc, r_dec, r_step
        line.update(extra)
        if with_cpu and rank == 0 and not args.no_cpu_baseline:
            try:  # north_star: the CPU SIMD path "in the same run" at every GPU count (N > 1: 3 repetitions on rank 0's host cores, the other ranks idle)
                g = state.get("gpu_result")
                cb = cpu_baseline(args.cpu_sample, args.cpu_reps if world == 1 else min(args.cpu_reps, 3), all_cores=True,
                                  seed=g["seed"] if g else SEED, gpu=g)
                par = cb.pop("parity", None)
                line["cpu_baseline"] = cb
                if par is not None:
                    line["parity_vs_oracle"] = par
                    if par.get("ok"):
                        line["config"]["workload"] += ", bit-exact vs CPU oracle (parity_vs_oracle: every word and base of the timed step's last buffer set)"
                    elif par.get("ok") is False:
                        state["rc"] = 3  # a fast kernel whose results differ from the reference's is not done
            except Exception as e:  # noqa: BLE001
                line["cpu_baseline"] = {"error": repr(e)[:300]}
        # LAST key (the driver keeps the tail of the line): the roofline fraction of every BASELINE config measured in this run
        def frac(block):
            return (extra.get(block) or {}).get("roofline", {}).get("frac")
        scan = extra.get("kmer_hdist_scan") or {}
        line["configs"] = {"cfg2_encode": r_enc["frac"], "cfg2_decode": r_dec["frac"], "cfg3_kmer_batch": frac("kmer_batch"),
                           # config 5: the CONSERVATIVE reading -- the mean of a 96-launch queue that starts on an idle chip (= a rocprofv3 trace of the scan alone)
                           "cfg5_kmer_hdist_scan": (scan.get("from_idle_queue_of_96") or {}).get("mean_frac"),
                           "cfg5_from_idle_last16": (scan.get("from_idle_queue_of_96") or {}).get("last16_frac"),
                           "cfg5_sustained_bursts": frac("kmer_hdist_scan"),
                           "cfg5_one_queue_of_64_mean": (scan.get("one_queue_of_64") or {}).get("mean_frac"),
                           "cfg5_one_queue_of_64_last16": (scan.get("one_queue_of_64") or {}).get("last16_frac"),
                           # ... and its fused count read the same way: the mean of a 96-launch queue from an idle chip (the first launches run on a cool chip, the middle in the dip)
                           "cfg5_fused_count": ((extra.get("kmer_hdist_count") or {}).get("from_idle_queue_of_96") or {}).get("mean_frac"),
                           "cfg5_fused_count_first8": ((extra.get("kmer_hdist_count") or {}).get("from_idle_queue_of_96") or {}).get("first8_frac"),
                           "cfg5_fused_count_last16": ((extra.get("kmer_hdist_count") or {}).get("from_idle_queue_of_96") or {}).get("last16_frac"),
                           "cfg5_fused_count_burst_after_the_scan_queues": frac("kmer_hdist_count"),
                           "unit": "fraction of 8 TB/s HBM3E on algorithmic bytes; null = block not run"}
        return line

    state["headline"] = make_line
    extra = state["line_extra"]

    if use_dist and (on_gpu_collectives or rehearse or args.backend == "gloo"):
        # config 4's concatenation, reported beside the step, never inside it.  These are collectives: a rank that
        # fails alone leaves the others waiting, and the watchdog then prints the headline with "stalled": true and
        # exits non-zero.
        wd.arm(args.dist_timeout if args.inject_stall < 0 else min(args.dist_timeout, 20.0), "allgather_packed (config 4 side measurement)")
        try:
            if args.inject_stall == rank:
                time.sleep(10**6)
            from bitnuc_amd.dist import allgather_packed, encode_allgather_overlapped
            if rehearse:
                local = torch.arange(1024, dtype=torch.int64) + rank
            elif on_gpu_collectives:
                local = words[0]
            else:  # gloo rehearsal on a GPU box: a small CPU copy, the point is the control flow
                local = words[0][:1 << 20].cpu()
            lw = local.numel()
            allgather_packed(local, group=ctl)
            fence()
            t = time.perf_counter()
            reps = 5
            for _ in range(reps):
                full = allgather_packed(local, group=ctl)
            fence()
            ag = (time.perf_counter() - t) / reps
            ok = bool(torch.equal(full[rank * lw:(rank + 1) * lw], local))
            # every slot against the closed form of the seeded stream (rank s's shard starts at base s * n, word s * n / 32)
            closed_form = rehearse or n % 32 == 0

            def slots(buf, words_per_slot):
                if not closed_form:
                    return {"all_slots_ok": None, "reason": "--bases is not a multiple of 32: the shards do not start at word boundaries of the seeded stream"}
                return all_slots_check(torch, buf, words_per_slot, world, lambda s_: s_ * (n // 32), SEED, rehearse=rehearse)
            if os.environ.get("BITNUC_BENCH_PLANT_BAD_SLOT") and world > 1:  # test hook: a wrong word in a PEER's slot must fail the run
                peer = (rank + 1) % world
                full[peer * lw + lw // 2] ^= 1
            chk = slots(full, lw)
            del full
            if chk["all_slots_ok"] is False:
                state["rc"] = 3  # the gather delivered something else than the shards' packed words: not a measurement
            extra["allgather_packed"] = {"ms": round(ag * 1e3, 3), "bytes_received_per_gpu": lw * 8 * (world - 1),
                                         "gb_s_per_gpu": round(lw * 8 * (world - 1) / ag / 1e9, 2), "own_slot_ok": ok, **chk, "backend": control_backend,
                                         "note": "all-gather of the packed u64 buffer (RCCL over xGMI when backend is nccl); fabric-bound, outside the timed step"}
            if on_gpu_collectives:
                # encode + concatenation end to end, one shot vs chunked overlap (8 pieces: the
                # fabric moves piece c while the GPU encodes piece c+1)
                def enc_chunk(w0, w1, dst):  # in place: the piece is encoded straight into this rank's slot of the gathered buffer
                    ctx.encode_dev(seqs[0][32 * w0:], min(n, 32 * w1) - 32 * w0, dst)

                def one_shot():
                    ctx.encode_dev(seqs[0], n, words[0])
                    return allgather_packed(words[0], group=ctl)
                def timed_e2e(fn):
                    ref_full = fn()
                    fence()
                    t = time.perf_counter()
                    for _ in range(reps):
                        full = fn()
                    fence()
                    chk2 = slots(full, nw)
                    if chk2["all_slots_ok"] is False:
                        state["rc"] = 3
                    return round((time.perf_counter() - t) / reps * 1e3, 3), bool(torch.equal(full, ref_full)) and chk2["all_slots_ok"] is not False, chk2["all_slots_ok"]
                e2e = {}
                e2e["one_shot_ms"], e2e["one_shot_ok"], e2e["one_shot_all_slots_ok"] = timed_e2e(one_shot)
                # the chunked in-place exchange rides on batched point-to-point operations: newer than the plain all-gather above, so it
                # is a SOFT block -- a thread with a bounded wait; if it never returns the line says so and the headline stands
                ov = bounded("encode_allgather_end_to_end.overlap8", lambda: timed_e2e(lambda: encode_allgather_overlapped(enc_chunk, nw, 8, words[0], group=ctl)),
                             min(args.dist_timeout, 150.0), state, torch, dev)
                if isinstance(ov, tuple):
                    e2e["overlap8_ms"], e2e["overlap8_ok"], e2e["overlap8_all_slots_ok"] = ov
                else:
                    e2e["overlap8"] = ov
                e2e["note"] = ("encode of this rank's 10^9-base shard + all-gather of the packed words; fabric-bound, so it cannot scale like the step; "
                               "overlap8 = 8 pieces exchanged in place by grouped point-to-point sends while the next piece is encoded")
                extra["encode_allgather_end_to_end"] = e2e
            else:
                extra["encode_allgather_end_to_end"] = {"skipped": "needs RCCL (backend nccl on every rank)"}
            # north_star's own split: a batch of independent sequences, whole sequences per rank, gather of unequal word counts (soft block)
            wd.arm(max(args.dist_timeout, 240.0), "ragged batch split by whole sequences")
            extra["ragged_batch_sharded"] = bounded("ragged_batch_sharded", lambda: ragged_batch_block(args, ctx, torch, dist, ctl, rank, world, dev if not rehearse else None, on_gpu_collectives, rehearse, fence, state),
                                                    min(args.dist_timeout, 200.0), state, torch, dev if not rehearse else None)
            # the fabric roofline of the gather -- the per-link rate is MEASURED on this node
            # (hipMemcpyPeerAsync from rank 0's device to every other rank's device, one link at a time and all at once)
            # while the other ranks wait at a host-side barrier; never quoted from the nominal figure alone.  Soft block, as above.
            wd.arm(max(args.dist_timeout, 240.0), "xgmi link probe")
            extra["allgather_packed"]["roofline"] = bounded("allgather_packed.roofline", lambda: xgmi_roofline(args, torch, rank, world, local_rank, rehearse, extra["allgather_packed"]),
                                                            min(args.dist_timeout, 120.0), state, torch, dev)
            dist.barrier()  # the default (gloo) group: host-side, the GPUs of the waiting ranks stay idle during the probe
        except Exception as e:  # noqa: BLE001 -- a failed collective is reported AND fails the run (rc 4): never a silent success
            extra.setdefault("allgather_packed", {})["error"] = repr(e)[:300]
            extra["collective_error"] = True
            state["rc"] = 4
        wd.disarm()

    if use_dist and not rehearse:
        # The same concatenation through the C ABI (what a C / Rust host calls): one-shot ncclAllGather and the chunked,
        # in-place overlap (bitnuc_encode_sharded_allgather_overlapped_dev).  It is an EXTRA block with its own RCCL
        # communicator: it runs in a thread with a bounded wait so that it can never cost the headline line.
        if on_gpu_collectives and not args.share_gpu and state.get("rc", 0) == 0 and not state.get("hung_thread"):
            extra["c_abi_allgather"] = c_abi_allgather_block(args, ctx, torch, dist, rank, world, seqs[0], n, state)
        else:
            extra["c_abi_allgather"] = {"skipped": "needs RCCL and one GPU per rank"}

    if not rehearse and (args.probe or (world == 1 and not args.no_extras)):
        # the step's two access shapes WITHOUT arithmetic, in the same sustained rotation as the timed loop: the
        # box's rate for exactly this traffic pattern (run before the side measurements free the rotating sets)
        try:
            extra["shape_probe_pair"] = sustained_shape_probes(args, ctx, torch, stream, seqs, words, backs, n, R)
        except Exception as e:  # noqa: BLE001
            extra["shape_probe_pair"] = {"error": repr(e)[:300]}
    if not rehearse and world == 1 and not args.no_extras:
        try:
            side_measurements(args, ctx, torch, dev, stream, seqs, words, backs, n, nw, extra)
        except Exception as e:  # noqa: BLE001 -- side measurements must never cost the headline line
            extra["extras_error"] = repr(e)[:300]
    if not rehearse and (args.probe or (world == 1 and not args.no_extras)):
        try:
            extra["stream_probe_gb_s"] = stream_probes(ctx, torch, stream, seqs, backs, n)
        except Exception as e:  # noqa: BLE001
            extra["stream_probe_gb_s"] = {"error": repr(e)[:300]}
    if state.get("side_blocks_stalled"):
        extra["side_blocks_stalled"] = state["side_blocks_stalled"]  # optional blocks that never returned: reported, the measured headline stands
    emit(make_line(extra))
    rc = state.get("rc", 0)
    if state.get("hung_thread"):
        os._exit(rc)  # the C-ABI extra block never returned: its communicator cannot be torn down in order
    if ctx:
        ctx.close()
    if use_dist:
        if probe_thread is not None and probe_thread.is_alive():
            os._exit(rc)  # an RCCL probe that never returned still holds the communicator: do not wait for it again
        wd.arm(max(args.dist_timeout, 300.0), "final barrier")
        dist.barrier()  # gloo: the ranks leave together, after rank 0 has timed the CPU baseline and printed the line
        wd.disarm()
        wd.arm(120.0, "destroy_process_group")
        dist.destroy_process_group()
        wd.disarm()
    return rc


def bounded(name, fn, seconds, state, torch, dev):
    """Run an OPTIONAL side block in a thread and wait at most `seconds` for it.  A block that never returns is reported
    ({"stalled": true} in its place, its name in the line's top-level "side_blocks_stalled") and the process later leaves
    through os._exit; the headline, measured before any of these blocks, is unaffected.  The strict rule -- a stalled
    collective fails the run -- stays with the measurements the line cannot do without (barrier, max-reduce, allgather_packed)."""
    box = {}

    def work():
        try:
            if dev is not None and getattr(dev, "type", "cpu") == "cuda":
                torch.cuda.set_device(dev)  # the current device is per-thread state
            box["value"] = fn()
        except Exception as e:  # noqa: BLE001
            box["error"] = repr(e)[:300]

    th = threading.Thread(target=work, daemon=True, name=name)
    th.start()
    th.join(seconds)
    if th.is_alive():
        state["hung_thread"] = True
        state.setdefault("side_blocks_stalled", []).append(name)
        print(f"[bench] side block '{name}' did not return within {seconds:.0f} s: reported in the line, headline unaffected", file=sys.stderr)
        return {"stalled": True, "waited_s": seconds}
    if "error" in box:
        return {"error": box["error"]}
    return box.get("value")


def ragged_batch_block(args, ctx, torch, dist, ctl, rank, world, dev, on_gpu_collectives, rehearse, fence, state):
    """north_star's multi-GPU split, measured where a node exists: "batches of independent sequences shard trivially across the 8 GPUs
    with RCCL all-gather over xGMI only for the final concatenation" (SURVEY 8e sentence 2).  A ragged batch -- 2^20 reads of 0..300 bases per
    rank, empty ones among them, one 10^7-base sequence that is longer than a fair share -- is split by WHOLE sequences
    (bitnuc_amd.dist.batch_shard_ranges), every rank plan-encodes its run straight into its slot of the output (no data-path collective) and
    the UNEQUAL word counts are gathered in place (grouped point-to-point).  Checked literally: every rank also encodes the WHOLE batch by
    itself and compares all words.  Without a GPU (--rehearse-cpu) the slots hold their global word indices: the exchange alone."""
    import numpy as np
    from bitnuc_amd.dist import allgatherv_packed_, batch_shard, batch_shard_ranges
    per_rank = 2000 if rehearse else 1 << 20
    count = world * per_rank
    rng = np.random.default_rng(SEED)  # the same batch on every rank
    lens = rng.integers(0, 301, size=count).astype(np.uint64)
    lens[::97] = 0
    lens[count // 3] = 40_000 if rehearse else 10_000_003
    off = np.zeros(count + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    seq_first, word_first = batch_shard_ranges(off, world)
    s0, s1, b0, b1, local, w0, nw = batch_shard(off, rank, world)
    total_words, total_bases = int(word_first[-1]), int(off[-1])
    res = {"workload": f"{count} independent sequences of 0..300 bases (every 97th empty) + one of {int(lens[count // 3])} bases, {total_bases} bases, split by whole sequences",
           "sequences_per_rank": [int(seq_first[r + 1] - seq_first[r]) for r in range(world)],
           "words_per_rank": [int(word_first[r + 1] - word_first[r]) for r in range(world)], "total_words": total_words}
    reps = 3
    if rehearse:
        out = torch.full((total_words,), -1, dtype=torch.int64)
        out[w0:w0 + nw] = torch.arange(w0, w0 + nw, dtype=torch.int64)
        allgatherv_packed_(out, word_first, group=ctl)
        res["all_slots_ok"] = bool(torch.equal(out, torch.arange(total_words, dtype=torch.int64)))
    else:
        import bitnuc_amd
        seq = torch.empty(b1 - b0 + 64, dtype=torch.uint8, device=dev)
        if b1 > b0:
            ctx.nucgen_dev(seq, b1 - b0, SEED + 7, first=b0)  # exactly this rank's bases of the batch's stream
        plan = bitnuc_amd.BatchPlan(ctx, torch.from_numpy(local.astype(np.int64)).to(dev), s1 - s0)
        assert plan.total_words == nw, (plan.total_words, nw)
        out = torch.full((total_words,), -1, dtype=torch.int64, device=dev)

        def encode_mine():
            if nw:
                plan.encode_dev(seq, out[w0:w0 + nw])

        def gather():
            if on_gpu_collectives:
                allgatherv_packed_(out, word_first, group=ctl)
                return out
            host = out.cpu()  # a gloo rehearsal on a GPU box: the exchange runs on host copies
            allgatherv_packed_(host, word_first, group=ctl)
            return host.to(dev)
        encode_mine()
        full = gather()
        fence()
        t = time.perf_counter()
        for _ in range(reps):
            encode_mine()
        fence()
        res["encode_ms_max_over_ranks_incl_barrier"] = round((time.perf_counter() - t) / reps * 1e3, 3)
        t = time.perf_counter()
        for _ in range(reps):
            full = gather()
        fence()
        ag = (time.perf_counter() - t) / reps
        res["gather_ms"] = round(ag * 1e3, 3)
        res["gather_gb_s_received_per_gpu"] = round((total_words - nw) * 8 / ag / 1e9, 2)
        # the check: this rank encodes the WHOLE batch by itself
        whole = torch.empty(total_bases + 64, dtype=torch.uint8, device=dev)
        ctx.nucgen_dev(whole, total_bases, SEED + 7)
        plan_all = bitnuc_amd.BatchPlan(ctx, torch.from_numpy(off.astype(np.int64)).to(dev), count)
        ref = torch.empty(plan_all.total_words, dtype=torch.int64, device=dev)
        plan_all.encode_dev(whole, ref)
        ctx.sync()
        res["all_slots_ok"] = bool(plan_all.total_words == total_words and torch.equal(full, ref))
        if not res["all_slots_ok"] and plan_all.total_words == total_words:
            w = int(torch.nonzero(full != ref)[0])
            res["first_bad_word"] = w
            res["first_bad_slot"] = int(np.searchsorted(word_first, w, side="right") - 1)
        plan.close()
        plan_all.close()
        del whole, ref
    if res["all_slots_ok"] is False:
        state["rc"] = 3
    res["note"] = ("whole sequences per rank (no rank encodes part of a sequence), each rank's actual word count gathered in place -- no padding to the largest slot; "
                   "fabric-bound like config 4's gather, outside the timed step")
    return res


def xgmi_roofline(args, torch, rank, world, local_rank, rehearse, ag):
    """roofline block of the all-gather: bound "xgmi", peak = the MEASURED aggregate rate of the links the gather uses."""
    nominal_link = 153.6  # GB/s per xGMI link (MI355X_MICROARCH.md / SURVEY section 5), 7 links per GPU
    if rehearse or world < 2:
        return {"value": None, "reason": "one rank: nothing crosses the fabric"}
    if rank != 0:
        return None  # measured and reported by rank 0
    if args.share_gpu or torch.cuda.device_count() < world:
        return {"value": None, "reason": f"{torch.cuda.device_count()} device(s) visible to rank 0 for {world} ranks: the link probe needs the peers' devices in one process"}
    try:
        from bitnuc_amd import api
        peers = [d for d in range(world) if d != local_rank]
        pr = api.peer_link_probe(local_rank, peers, nbytes=256 << 20, reps=3)
    except Exception as e:  # noqa: BLE001
        return {"value": None, "reason": "link probe failed: " + repr(e)[:200]}
    peak = pr["gb_s_all"]
    ach = ag["gb_s_per_gpu"]
    return {"bound": "xgmi", "achieved": ach, "peak": peak, "unit": "GB/s", "frac": round(ach / peak, 4) if peak else None,
            "per_link_measured": pr["gb_s_each"], "links_used": len(peers), "nominal": round(nominal_link * len(peers), 1),
            "frac_of_nominal": round(ach / (nominal_link * len(peers)), 4),
            "how": "achieved = bytes received per GPU / all-gather time; peak = hipMemcpyPeerAsync from this GPU to all peers at once (outbound; the fabric is symmetric)"}


def c_abi_allgather_block(args, ctx, torch, dist, rank, world, seq, n, state):
    """bitnuc_comm_init_rank + one-shot and chunked in-place all-gather through the C ABI, timed; bounded wait."""
    import bitnuc_amd
    res = {}

    def work():
        try:
            uid = [bitnuc_amd.Comm.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)  # default (gloo) group
            comm = bitnuc_amd.Comm(ctx, world, rank, uid[0])
            n32 = n - n % 32
            cnt = n32 // 32
            one = torch.empty(world * cnt, dtype=torch.int64, device=seq.device)
            two = torch.zeros(world * cnt, dtype=torch.int64, device=seq.device)
            reps = 3
            for name, fn, buf in (("one_shot", lambda: comm.encode_sharded_allgather_dev(seq, n32, one), one),
                                  ("overlap8", lambda: comm.encode_sharded_allgather_overlapped_dev(seq, n32, 8, two), two)):
                fn()
                ctx.sync()
                dist.barrier()
                t = time.perf_counter()
                for _ in range(reps):
                    fn()
                ctx.sync()
                dist.barrier()
                res[name + "_ms"] = round((time.perf_counter() - t) / reps * 1e3, 3)
            res["overlap_equals_one_shot"] = bool(torch.equal(one, two))
            # ... and both against the closed form of the seeded stream, every rank's slot (rank s's shard = words s * n / 32 ...)
            if n % 32 == 0:
                for name, buf in (("one_shot", one), ("overlap8", two)):
                    chk = all_slots_check(torch, buf, cnt, world, lambda s_: s_ * (n // 32), SEED)
                    res[name + "_all_slots_ok"] = chk["all_slots_ok"]
                    if not chk["all_slots_ok"]:
                        res[name + "_first_bad_slot"] = chk["first_bad_slot"]
                        state["rc"] = 3
            res["gb_s_per_gpu_one_shot"] = round(cnt * 8 * (world - 1) / (res["one_shot_ms"] * 1e-3) / 1e9, 2) if world > 1 else None
            comm.close()
            res["note"] = ("encode of this rank's shard + concatenation through the C ABI: ncclAllGather in place (one_shot) and 8 pieces moved in place by "
                           "grouped ncclSend / ncclRecv on a second stream while the next piece is encoded (overlap8); includes the host barriers around the loop")
        except Exception as e:  # noqa: BLE001
            res["error"] = repr(e)[:300]

    out = bounded("c_abi_allgather", work, min(args.dist_timeout, 240.0), state, torch, seq.device)
    if isinstance(out, dict) and out.get("stalled"):
        out["note"] = "the C-ABI extra block did not finish within its bounded wait; the headline and the torch.distributed blocks above are unaffected"
        return out
    return res


def timed_median(torch, stream, fn, reps=10):
    ms = []
    for _ in range(reps + 2):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        fn()
        b.record(stream)
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    return statistics.median(ms[2:])


def timed_sustained(torch, stream, fn, burst=8, rounds=5):
    """ms per launch in back-to-back bursts (no host sync inside a burst: how kernels run in a pipeline and how the timed step
    itself is measured); fn(i) must alternate its buffers with i so that nothing it reads was written or read by the previous
    launch out of the 256 MiB Infinity Cache."""
    ms = []
    k = 0
    for r in range(rounds + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn(k)
        k += 1
        a.record(stream)
        for _ in range(burst):
            fn(k)
            k += 1
        b.record(stream)
        torch.cuda.synchronize()
        if r:
            ms.append(a.elapsed_time(b) / burst)
    return statistics.median(ms)


def timed_queue(torch, stream, fn, n_launches=64, idle_s=0.0, every=1):
    """Durations (ms per launch) of n_launches in ONE queue (no host wait inside), one figure per group of `every` launches: what a
    kernel does when the queue never drains and the chip's power management has to settle.  idle_s > 0: the queue starts on
    an IDLE chip (host sleep after a device sync, no warm-up launch) -- how a rocprofv3 trace of the kernel alone sees it
    (profiles/r05_rocprof/kernel_stats_cfg5.csv), and the conservative reading.  An event between two launches costs a few microseconds
    of queue time that no kernel duration contains: every = 8 keeps that below 0.5 % of a 0.3 ms kernel (every = 1: 3-4 %)."""
    assert n_launches % every == 0
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n_launches // every + 1)]
    if idle_s > 0:
        torch.cuda.synchronize()
        time.sleep(idle_s)
    else:
        fn(0)
    ev[0].record(stream)
    for i in range(n_launches):
        fn(i + 1)
        if (i + 1) % every == 0:
            ev[(i + 1) // every].record(stream)
    torch.cuda.synchronize()
    return [ev[i].elapsed_time(ev[i + 1]) / every for i in range(n_launches // every)]


def hbm(alg, ms):
    gbs = alg / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
            "algorithmic_bytes_per_launch": alg}


def side_measurements(args, ctx, torch, dev, stream, seqs, words, backs, n, nw, extra):
    """BASELINE configs[2] and [4] and the SURVEY 8f rows, measured beside the headline (never inside the timed step)."""
    del backs[1:], seqs[1:], words[1:]
    torch.cuda.empty_cache()

    def timed(fn, reps=10):
        return timed_median(torch, stream, fn, reps)

    def both(fn, burst=8, rounds=5, reps=8):
        """(sustained ms per launch, isolated ms per launch): back-to-back bursts -- how the timed step itself is measured -- and
        single launches between host syncs (adds the launch ramp and tail, 2-4 us, which matters for the 40-80 us kernels)."""
        return timed_sustained(torch, stream, lambda i: fn(), burst=burst, rounds=rounds), timed_median(torch, stream, fn, reps)
    SUST = "sustained bursts of 8 launches (as the timed step is measured); isolated_ms = single launches between host syncs"
    count, k = 10**8, 31
    kseq = torch.empty(count * k, dtype=torch.uint8, device=dev)
    ctx.nucgen_dev(kseq, count * k, SEED + 100)
    kout = torch.empty(count, dtype=torch.int64, device=dev)
    ms, iso = both(lambda: ctx.as_2bit_batch_dev(kseq, k, k, count, kout), burst=6, rounds=4)
    extra["kmer_batch"] = {"workload": "BASELINE configs[2]: 10^8 dense 31-mers as_2bit -> u64", "gkmers_s": round(count / (ms * 1e-3) / 1e9, 2),
                           "ms": round(ms, 4), "isolated_ms": round(iso, 4), "timing": SUST, "roofline": hbm(count * (k + 8), ms)}
    del kseq, kout
    # every window of a sequence (`seq.windows(k)` + as_2bit, src/lib.rs:170-173): stride 1, 1 B read + 8 B written per window
    nwin = n - k + 1
    wout = torch.empty(nwin, dtype=torch.int64, device=dev)
    ms, iso = both(lambda: ctx.as_2bit_batch_dev(seqs[0], k, 1, nwin, wout), burst=4, rounds=3, reps=5)
    extra["kmer_windows"] = {"workload": "as_2bit of every 31-base window of 10^9 bases (stride 1) -> u64 per window", "ms": round(ms, 4),
                             "isolated_ms": round(iso, 4), "timing": SUST.replace("8", "4"),
                             "gwindows_s": round(nwin / (ms * 1e-3) / 1e9, 2), "roofline": hbm(9 * nwin, ms)}
    del wout
    dist_out = torch.empty(n - k + 1, dtype=torch.uint8, device=dev)
    q = 0x1B1B1B1B1B1B1B1B & ((1 << 62) - 1)
    iso = timed(lambda: ctx.kmer_hdist_scan_dev(seqs[0], n, k, q, dist_out))
    douts = (dist_out, backs[0])
    ms = timed_sustained(torch, stream, lambda i: ctx.kmer_hdist_scan_dev(seqs[0], n, k, q, douts[i & 1]))
    extra["kmer_hdist_scan"] = {"workload": "BASELINE configs[4]: sliding 31-mer pack + Hamming distance to one query over 10^9 bases",
                                "gwindows_s": round((n - k + 1) / (ms * 1e-3) / 1e9, 2), "ms": round(ms, 4), "isolated_ms": round(iso, 4),
                                "timing": "sustained bursts of 8 launches, two output buffers", "roofline": hbm(2 * (n - k + 1), ms)}
    # Three readings of the same kernel: the bursts above restart after every host sync on a busy chip; one queue of 64
    # launches on a busy chip; and -- the CONSERVATIVE one, the one `configs.cfg5_kmer_hdist_scan` carries -- a queue of 96 launches that
    # starts on an idle chip (1 s of host sleep), which includes the power controller's dip after the first launches and is what a
    # rocprofv3 trace of the scan alone measures (profiles/r05_rocprof/kernel_stats_cfg5.csv).
    # (three such queues, each after its own second of idleness; the one with the MEDIAN mean is reported, all three means beside it: a single
    # queue now and then contains a group that takes 100 us longer for reasons outside the kernel -- profiles/r05_bench_n1_outlier_group.json)
    runs = sorted((timed_queue(torch, stream, lambda i: ctx.kmer_hdist_scan_dev(seqs[0], n, k, q, douts[i & 1]), n_launches=96, idle_s=1.0, every=8) for _ in range(3)), key=sum)  # 12 groups of 8 each
    q96 = runs[1]
    extra["kmer_hdist_scan"]["from_idle_queue_of_96"] = {"mean_ms": round(sum(q96) / len(q96), 4), "first8_ms": round(q96[0], 4), "slowest_group_of_8_ms": round(max(q96), 4),
                                                         "last16_ms": round(sum(q96[-2:]) / 2, 4), "mean_frac": hbm(2 * (n - k + 1), sum(q96) / len(q96))["frac"],
                                                         "last16_frac": hbm(2 * (n - k + 1), sum(q96[-2:]) / 2)["frac"],
                                                         "slowest_group_over_settled": round(max(q96) / (sum(q96[-2:]) / 2), 3),
                                                         "groups_of_8_ms": [round(x, 4) for x in q96],
                                                         "mean_ms_of_the_three_queues": [round(sum(r) / len(r), 4) for r in runs], "reported": "the queue with the median mean"}
    qs = timed_queue(torch, stream, lambda i: ctx.kmer_hdist_scan_dev(seqs[0], n, k, q, douts[i & 1]))
    extra["kmer_hdist_scan"]["one_queue_of_64"] = {"mean_ms": round(sum(qs) / len(qs), 4), "first4_ms": round(sum(qs[:4]) / 4, 4), "slowest_ms": round(max(qs), 4),
                                                   "last16_ms": round(sum(qs[-16:]) / 16, 4), "mean_frac": hbm(2 * (n - k + 1), sum(qs) / len(qs))["frac"],
                                                   "last16_frac": hbm(2 * (n - k + 1), sum(qs[-16:]) / 16)["frac"]}
    if hasattr(ctx, "kmer_hdist_count_dev"):
        # SURVEY 8d cfg 5's optional fused output: only the COUNT of windows with d <= tau leaves the chip (1 B read per window)
        cnt1 = torch.zeros(1, dtype=torch.int64, device=dev)
        tau = 8
        ms, iso = both(lambda: ctx.kmer_hdist_count_dev(seqs[0], n, k, q, tau, cnt1))
        extra["kmer_hdist_count"] = {"workload": f"same scan, fused `d <= {tau}` count instead of the distance bytes (1 B read per window)",
                                     "gwindows_s": round((n - k + 1) / (ms * 1e-3) / 1e9, 2), "ms": round(ms, 4), "isolated_ms": round(iso, 4), "timing": SUST, "matches": int(cnt1.item()),
                                     "matches_check": int((dist_out <= tau).sum().item()),
                                     "bound": "HBM while the clock is high (bursts: 0.153-0.160 ms = 0.78-0.82 of 8 TB/s), the CU's cycles once it has fallen (segments of 32 windows x 32 shifts with THREE channels "
                                              "per base -- [b != q] is affine in an (A, C, G) one-hot with T = 0 --: per 1024 windows 3 MFMAs + ~50 vector instructions; the threshold is inside the product: 6-bit fields 32 + tau - d, three rows per "
                                              "register, v_or3 + v_bitop3 + v_bcnt per four windows, nothing on the scalar unit; trips of 4 rounds, 12 workgroups per CU; `ms` here is a burst right after the scan's queues; "
                                              "from_idle_queue_of_96 is the reading `configs.cfg5_fused_count` carries); the four-channel form 0.160 / 0.177 / 0.195 (bursts / settled / from idle), round 5's first form (v_cmp + s_bcnt1 per register) 0.177 / 0.186 / 0.202, "
                                              "round 4's bit-plane form 0.30-0.33 (profiles/r05_ab_count_ch3*.txt, r05_ab_count_emit*.txt, r05_pmc_scan_mfma_final_forms.txt)",
                                     "algorithmic_gb_s": round((n - k + 1) / (ms * 1e-3) / 1e9, 1), "roofline": hbm(n - k + 1, ms)}
        cruns = sorted((timed_queue(torch, stream, lambda i: ctx.kmer_hdist_count_dev(seqs[0], n, k, q, tau, cnt1), n_launches=96, idle_s=1.0, every=8) for _ in range(3)), key=sum)
        c96 = cruns[1]
        extra["kmer_hdist_count"]["from_idle_queue_of_96"] = {"mean_ms": round(sum(c96) / len(c96), 4), "first8_ms": round(c96[0], 4), "last16_ms": round(sum(c96[-2:]) / 2, 4), "slowest_group_of_8_ms": round(max(c96), 4),
                                                              "mean_frac": hbm(n - k + 1, sum(c96) / len(c96))["frac"], "first8_frac": hbm(n - k + 1, c96[0])["frac"], "last16_frac": hbm(n - k + 1, sum(c96[-2:]) / 2)["frac"],
                                                              "groups_of_8_ms": [round(x, 4) for x in c96],
                                                              "mean_ms_of_the_three_queues": [round(sum(r) / len(r), 4) for r in cruns], "reported": "the queue with the median mean"}
    # bulk packed-vs-packed Hamming distance (hdist, hamming/multi.rs:121-160): 16 B per 32-base word pair
    wa, wb = words[0], torch.empty(nw, dtype=torch.int64, device=dev)
    ctx.nucgen_dev(backs[0], n, SEED + 200)
    ctx.encode_dev(backs[0], n, wb)
    res = torch.zeros(1, dtype=torch.int32, device=dev)
    SUST32 = SUST.replace("bursts of 8", "bursts of 32")  # 40-80 us kernels: the two timing markers of a burst of 8 would be 1 % of it
    ms, iso = both(lambda: ctx.hdist_dev(wa, nw, wb, nw, n, res), burst=32)
    extra["hdist_bulk"] = {"workload": "hdist of two 10^9-base packed buffers (SURVEY 8f rank 1)", "ms": round(ms, 4), "isolated_ms": round(iso, 4), "timing": SUST32,
                           "gbases_s": round(n / (ms * 1e-3) / 1e9, 1), "distance": int(res.item()) & 0xFFFFFFFF, "roofline": hbm(16 * nw, ms)}
    # analysis directly on packed words
    cnt = torch.zeros(4, dtype=torch.int64, device=dev)
    # these two read only 250 MB per launch, which would fit the 256 MiB Infinity Cache:
    # alternate between two packed buffers so that every launch streams from HBM
    flip = [0]

    def alt():
        flip[0] ^= 1
        return wb if flip[0] else wa
    ms, iso = both(lambda: ctx.base_counts_dev(alt(), nw, n, cnt), burst=32)
    extra["base_counts"] = {"workload": "A/C/G/T counts of 10^9 packed bases (analysis.rs:23-39 without the decode)", "ms": round(ms, 4),
                            "isolated_ms": round(iso, 4), "timing": SUST32 + "; input alternates between two 250 MB buffers (cache-cold)",
                            "gbases_s": round(n / (ms * 1e-3) / 1e9, 1), "roofline": hbm(8 * nw, ms)}
    qd = torch.empty(nw, dtype=torch.uint8, device=dev)
    ms, iso = both(lambda: ctx.hdist_query_dev(0x1B1B1B1B1B1B1B1B, alt(), nw, 32, qd), burst=32)
    extra["hdist_query"] = {"workload": "one packed 32-mer against 3.1e7 packed 32-mers -> u8 distances", "ms": round(ms, 4),
                            "isolated_ms": round(iso, 4), "timing": SUST32 + "; input alternates between two 250 MB buffers (cache-cold)",
                            "gwords_s": round(nw / (ms * 1e-3) / 1e9, 2), "roofline": hbm(9 * nw, ms)}
    del qd
    # split_packed at an odd base in the middle (16 B per word: read once, write once)
    sidx = n // 2 + 5
    snl, snr = ctx.split_packed_sizes(nw, n, sidx, canonical=True)
    sl, sr = torch.empty(snl, dtype=torch.int64, device=dev), torch.empty(snr, dtype=torch.int64, device=dev)
    ms, iso = both(lambda: ctx.split_packed_dev(alt(), nw, n, sidx, sl, sr, canonical=True), burst=32)
    extra["split_packed"] = {"workload": "split 10^9 packed bases at base n/2+5 (functions/split.rs:15-99, funnel-shift form)", "ms": round(ms, 4),
                             "isolated_ms": round(iso, 4), "timing": SUST32,
                             "gbases_s": round(n / (ms * 1e-3) / 1e9, 1), "roofline": hbm(8 * (nw + snl + snr), ms)}
    del sl, sr
    # ragged batch of independent sequences: 150-base reads (each read pads its own last word)
    L, rcount = 150, n // 150
    roff = torch.arange(0, rcount + 1, dtype=torch.int64, device=dev) * L
    rwo = torch.empty(rcount + 1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    rtotal = ctx.batch_word_offsets_dev(roff, rcount, rwo)
    rwords = torch.empty(rtotal, dtype=torch.int64, device=dev)
    rb = L * rcount
    alg = rb + 8 * rtotal  # bases + packed words (offset tables / plan bytes not counted)

    def batch_block(ms_e, ms_d, workload, **more):
        d = {"workload": workload, "encode_ms": round(ms_e, 4), "decode_ms": round(ms_d, 4),
             "encode_gbases_s": round(rb / (ms_e * 1e-3) / 1e9, 1), "decode_gbases_s": round(rb / (ms_d * 1e-3) / 1e9, 1),
             "encode_gb_s": round(alg / (ms_e * 1e-3) / 1e9, 1), "decode_gb_s": round(alg / (ms_d * 1e-3) / 1e9, 1),
             "encode_frac": round(alg / (ms_e * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "decode_frac": round(alg / (ms_d * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
             "algorithmic_bytes_per_launch": alg}
        d.update(more)
        return d
    # (a) with a layout plan (bitnuc_batch_plan): built once per offsets table, used by every encode / decode of that layout
    import bitnuc_amd
    plan = bitnuc_amd.BatchPlan(ctx)
    tb = []
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        plan.build(roff, rcount)
        tb.append((time.perf_counter() - t0) * 1e3)
    assert plan.total_words == rtotal
    rwords2 = torch.empty(rtotal, dtype=torch.int64, device=dev)  # second set: sustained bursts alternate buffers
    back2 = torch.empty(n, dtype=torch.uint8, device=dev)
    wsets, bsets = (rwords, rwords2), (backs[0], back2)
    plan.encode_dev(seqs[0], rwords2)
    iso_e = timed(lambda: plan.encode_dev(seqs[0], rwords))
    iso_d = timed(lambda: plan.decode_dev(rwords, backs[0]))
    ms_e = timed_sustained(torch, stream, lambda i: plan.encode_dev(seqs[0], wsets[i & 1]))
    ms_d = timed_sustained(torch, stream, lambda i: plan.decode_dev(wsets[i & 1], bsets[i & 1]))
    extra["reads_batch"] = batch_block(ms_e, ms_d, f"{rcount} independent 150-base reads (each read pads its own last word), encode / decode with a layout plan",
                                       timing="sustained: back-to-back bursts of 8 launches alternating two buffer sets, as the timed step is measured",
                                       isolated_encode_ms=round(iso_e, 4), isolated_decode_ms=round(iso_d, 4),
                                       plan_build_ms=round(min(tb[1:]), 4), roundtrip_ok=bool(torch.equal(seqs[0][:rb], backs[0][:rb])),
                                       note="plan = word offsets + one byte offset per 64-word tile + one pad byte per word, from one pass over the offsets table "
                                            "(plan_build_ms: host-synchronous, includes the word-offsets scan); the same plan serves the later decode")
    # what the plan kernels actually move: + 1 pad byte per word + 8 bytes per 64-word tile (2.6-2.9 % of a tile's traffic); the
    # primary fractions above count bases + words only (SURVEY 8d's algorithmic bytes), these two count the plan's bytes too
    plan_bytes = rtotal + 8 * ((rtotal + 63) // 64)
    extra["reads_batch"].update({"plan_bytes_per_launch": plan_bytes,
                                 "encode_frac_with_plan_bytes": round((alg + plan_bytes) / (ms_e * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                 "decode_frac_with_plan_bytes": round((alg + plan_bytes) / (ms_d * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
    plan.close()
    # (b) from the two offset tables alone, nothing kept between calls: every call emits the plan (pad bytes, tile bases) into
    # context scratch with one asynchronous pass over both tables, then runs the plan kernel -- both launches are inside the time
    backs[0].zero_()
    ms_e = timed_sustained(torch, stream, lambda i: ctx.encode_batch_dev(seqs[0], roff, rwo, rcount, rtotal, wsets[i & 1]))
    ms_d = timed_sustained(torch, stream, lambda i: ctx.decode_batch_dev(wsets[i & 1], rwo, roff, rcount, rtotal, bsets[i & 1]))
    # this form's algorithmic bytes INCLUDE its two tables (every call must read offsets[] and word_offsets[], 16 B per read: +8.4 % for
    # 150-base reads): the primary fractions count them; the bases + words only figures stay beside them for comparison with (a)
    alg_t = alg + 16 * (rcount + 1)
    tb_blk = batch_block(ms_e, ms_d, "the same batch through encode_batch_dev / decode_batch_dev (offset tables only, nothing kept between calls: every call runs plan_emit_kernel + the plan kernel)",
                         timing="sustained", roundtrip_ok=bool(torch.equal(seqs[0][:rb], backs[0][:rb])), tables_bytes_per_launch=16 * (rcount + 1))
    tb_blk.update({"encode_gb_s_without_tables": tb_blk["encode_gb_s"], "decode_gb_s_without_tables": tb_blk["decode_gb_s"],
                   "encode_frac_without_tables": tb_blk["encode_frac"], "decode_frac_without_tables": tb_blk["decode_frac"],
                   "encode_gb_s": round(alg_t / (ms_e * 1e-3) / 1e9, 1), "decode_gb_s": round(alg_t / (ms_d * 1e-3) / 1e9, 1),
                   "encode_frac": round(alg_t / (ms_e * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "decode_frac": round(alg_t / (ms_d * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                   "algorithmic_bytes_per_launch": alg_t,
                   "note": "algorithmic bytes = bases + packed words + the two 8-byte-per-read tables this form reads on every call; *_without_tables = bases + words only"})
    extra["reads_batch_tables"] = tb_blk
    ms_fe = timed_sustained(torch, stream, lambda i: ctx.encode_fixed_dev(seqs[0], L, L, rcount, wsets[i & 1]))
    ms_fd = timed_sustained(torch, stream, lambda i: ctx.decode_fixed_dev(wsets[i & 1], L, L, rcount, bsets[i & 1]))
    extra["reads_fixed"] = batch_block(ms_fe, ms_fd, f"{rcount} fixed-length 150-base reads, encode_fixed / decode_fixed (no offsets tables)", timing="sustained")
    del rwords, rwords2, back2, wsets, bsets, roff, rwo
    ctx.sync()
    backs.append(dist_out)  # reused by the probes
    # host-pointer entry points: PCIe-inclusive rates (never `value`) and the latency of the reference's own bench shapes
    try:
        extra["host_path"] = host_path_block(ctx, torch)
    except Exception as e:  # noqa: BLE001
        extra["host_path"] = {"error": repr(e)[:300]}
    try:
        extra["small_call_latency"] = small_call_latency()
    except Exception as e:  # noqa: BLE001
        extra["small_call_latency"] = {"error": repr(e)[:300]}


def sustained_shape_probes(args, ctx, torch, stream, seqs, words, backs, n, R):
    """Kernels with exactly the codec's access shapes and no arithmetic (encode shape: 16 B nt-loads + 4 B nt-stores, XCD-contiguous tile order, per
    lane, 2 in flight, 128-thread workgroups; decode shape: 4 B loads + 16 B nt-stores, 256-thread workgroups), run
    back to back in the timed loop's rotation (the decode shape reads what was written R-1 steps earlier), per-kernel
    HIP events on the launch stream.  What this box sustains for this traffic pattern; the codec cannot be faster than
    it by more than noise, and the gap to 8 TB/s that remains here is the memory system's, not the arithmetic's."""
    steps = max(20, min(args.steps, 100))
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]

    def pstep(i, e=None):
        r = i % R
        d = r if args.warm_decode else (r + 1) % R
        if e:
            e[0].record(stream)
        ctx.stream_probe_dev(3, seqs[r], words[r], n)
        if e:
            e[1].record(stream)
        ctx.stream_probe_dev(4, words[d], backs[d], n)
        if e:
            e[2].record(stream)
    for i in range(5):
        pstep(i)
    torch.cuda.synchronize()
    for i in range(steps):
        pstep(i, ev[i])
    torch.cuda.synchronize()
    e_ms = sum(e[0].elapsed_time(e[1]) for e in ev) / steps
    d_ms = sum(e[1].elapsed_time(e[2]) for e in ev) / steps
    # the same pairs with NO events in between (what the timed loop's `ms_per_step` is to be compared with: per-launch event records
    # cost about 2.5 % of a step, profiles/r03_ab_event_every.txt)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for i in range(steps):
        pstep(i)
    b.record(stream)
    torch.cuda.synchronize()
    pair_free_ms = a.elapsed_time(b) / steps
    for r in range(R):  # the probes overwrote the packed words: restore them for what follows
        ctx.encode_dev(seqs[r], n, words[r])
    ctx.sync()
    alg = n * BYTES_PER_BASE
    return {"encode_shape_ms": round(e_ms, 4), "decode_shape_ms": round(d_ms, 4), "ms_per_pair": round(e_ms + d_ms, 4),
            "encode_shape_gb_s": round(alg / (e_ms * 1e-3) / 1e9, 1), "decode_shape_gb_s": round(alg / (d_ms * 1e-3) / 1e9, 1),
            "pair_gb_s": round(2 * alg / ((e_ms + d_ms) * 1e-3) / 1e9, 1), "steps": steps,
            "ms_per_pair_without_events": round(pair_free_ms, 4), "pair_gb_s_without_events": round(2 * alg / (pair_free_ms * 1e-3) / 1e9, 1),
            "note": "same bytes, same instructions shapes, same rotation as the timed step, no arithmetic between load and store"}


def stream_probes(ctx, torch, stream, seqs, backs, n):
    """The box's own streaming rates, one launch at a time (host sync between launches): pure read / copy / fill.
    Rates for orientation, not bounds (a kernel that also writes can run faster than the read probe)."""
    def rate(mode, moved):
        ms = []
        for i in range(8):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            ctx.stream_probe_dev(mode, seqs[i % len(seqs)], backs[(i + 1) % len(backs)], min(n, backs[(i + 1) % len(backs)].numel()))
            b.record(stream)
            torch.cuda.synchronize()
            ms.append(a.elapsed_time(b))
        return round(moved / (statistics.median(ms[2:]) * 1e-3) / 1e9, 1)
    def sustained(mode, moved, burst=24):
        # one burst moves burst x ~1 GB: what is still dirty in the 256 MiB Infinity Cache when its last kernel "ends" is ~1 % of it
        # (an isolated 1 GB fill ends with up to a quarter of its bytes not yet in HBM and reads 4-5 % high)
        ms = []
        for rnd in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            for i in range(burst):
                ctx.stream_probe_dev(mode, seqs[i % len(seqs)], backs[(i + 1) % len(backs)], nb)
            b.record(stream)
            torch.cuda.synchronize()
            ms.append(a.elapsed_time(b) / burst)
        return round(moved / (min(ms[1:]) * 1e-3) / 1e9, 1)
    nb = min(n, min(t.numel() for t in backs))
    return {"read": max(rate(m, nb) for m in (0 | 8, 0 | 8 | 32, 0)),
            "copy": max(rate(m, 2 * nb) for m in (1 | 8 | 16, 1 | 8, 1 | 16, 1)),
            "fill": max(rate(m, nb) for m in (2 | 16, 2)),
            "note": "isolated launches, best of a few cache-policy variants per shape",
            "sustained": {"read": sustained(0 | 8, nb), "copy": sustained(1 | 8 | 16, 2 * nb), "fill_nt": sustained(2 | 16, nb), "fill_plain": sustained(2, nb),
                          "note": "bursts of 24 launches over rotating buffers (~24 GB per burst): the two directions of this memory system; a kernel that reads r and "
                                  "writes w bytes cannot finish before r / read + w / fill (profiles/NARRATIVE_r01_r03.md section 3)"}}


def host_path_block(ctx, torch):
    """Host-pointer bulk entry points on 10^9 bases of pageable host memory (PCIe inclusive), next to the box's
    pinned hipMemcpyAsync rate."""
    import numpy as np
    n = 10**9
    out = {}
    pin = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
    devb = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    for name, (dst, src) in (("pinned_h2d_gb_s", (devb, pin)), ("pinned_d2h_gb_s", (pin, devb))):
        ts = []
        for _ in range(4):
            torch.cuda.synchronize()
            t = time.perf_counter()
            dst.copy_(src, non_blocking=True)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t)
        out[name] = round((1 << 30) / min(ts[1:]) / 1e9, 1)
    del pin, devb
    seq = np.frombuffer(np.random.default_rng(1).bytes(n // 4), dtype=np.uint8)
    seq = np.repeat(np.frombuffer(b"ACGT", dtype=np.uint8)[seq & 3], 4)[:n].copy()  # pageable, valid bases
    ctx.set_variant("force_gpu", 0)
    te, td = [], []
    words = np.zeros((n + 31) // 32, dtype=np.uint64)  # caller-owned and already touched, as in a pipeline that reuses its
    back = np.zeros(n, dtype=np.uint8)                 # buffers (a fresh output would be timed by its page faults)
    ok = True
    for it in range(4):
        if it:  # identical iterations over the same buffers would hide a stale chunk -- other data each time, every result checked
            seq[it * 1000003::7] = seq[5::7][: len(seq[it * 1000003::7])]
        words.fill(0)
        back.fill(0)
        t = time.perf_counter()
        ctx.encode_into(seq, words)
        te.append(time.perf_counter() - t)
        t = time.perf_counter()
        ctx.decode_into(words, n, back)
        td.append(time.perf_counter() - t)
        ok = ok and bool(np.array_equal(seq, back))
    te, td = te[1:], td[1:]
    # the k-mer host calls ride the same engine (configs 3 and 5 for data in host memory): 10^9 / 31 dense 31-mers and the scan of the same bytes
    import ctypes as C
    from bitnuc_amd import _lib as L
    kerr = L.BitnucErr()
    kcnt = n // 31
    kout = np.zeros(kcnt, dtype=np.uint64)
    kdist = np.zeros(n - 30, dtype=np.uint8)
    tk, ts = [], []
    for _ in range(3):
        t = time.perf_counter()
        st1 = ctx._lib.bitnuc_as_2bit_batch(ctx._h, C.c_void_p(seq.ctypes.data), 31, 31, kcnt, C.c_void_p(kout.ctypes.data), C.byref(kerr))
        tk.append(time.perf_counter() - t)
        t = time.perf_counter()
        st2 = ctx._lib.bitnuc_kmer_hdist_scan(ctx._h, C.c_void_p(seq.ctypes.data), n, 31, C.c_uint64(int(kout[12345])), C.c_void_p(kdist.ctypes.data), C.byref(kerr))
        ts.append(time.perf_counter() - t)
    out["kmer_batch_host"] = {"gkmers_s": round(kcnt / min(tk[1:]) / 1e9, 2), "ms": round(min(tk[1:]) * 1e3, 2), "status": st1,
                              "frac_of_pinned_h2d": round(31 * kcnt / min(tk[1:]) / 1e9 / out["pinned_h2d_gb_s"], 3),
                              "workload": f"{kcnt} dense 31-mers from pageable host memory -> u64 in host memory (bitnuc_as_2bit_batch)"}
    out["kmer_scan_host"] = {"gwindows_s": round((n - 30) / min(ts[1:]) / 1e9, 2), "ms": round(min(ts[1:]) * 1e3, 2), "status": st2, "self_hit_ok": bool(kdist[12345 * 31] == 0),
                             "frac_of_pinned_d2h": round((n - 30) / min(ts[1:]) / 1e9 / out["pinned_d2h_gb_s"], 3),
                             "workload": "sliding 31-mer pack + Hamming scan of 10^9 bases, host memory in and out (bitnuc_kmer_hdist_scan): a byte in and a byte out per window, both DMA engines at once"}
    del kout, kdist
    ctx.set_variant("force_gpu", 1)
    try:  # which engine ran, and how the staged engine's pools would be sized (cores visible vs the cgroup quota)
        out["pipe"] = dict(ctx.host_pipe_info(), chunk_mb_env=os.environ.get("BITNUC_PIPE_CHUNK_MB"), host_threads_env=os.environ.get("BITNUC_HOST_THREADS"))
    except Exception as e:  # noqa: BLE001
        out["pipe"] = {"error": repr(e)[:200]}
    out.update({"bases": n, "encode_gbases_s": round(n / min(te) / 1e9, 1), "decode_gbases_s": round(n / min(td) / 1e9, 1),
                "encode_gb_s_moved": round(1.25 * n / min(te) / 1e9, 1), "decode_gb_s_moved": round(1.25 * n / min(td) / 1e9, 1),
                "roundtrip_ok": ok,
                "encode_frac_of_pinned_h2d": round(n / min(te) / 1e9 / out["pinned_h2d_gb_s"], 3),
                "decode_frac_of_pinned_d2h": round(n / min(td) / 1e9 / out["pinned_d2h_gb_s"], 3),
                "note": "bitnuc_encode / bitnuc_decode on pageable host buffers, H2D / kernel / D2H overlapped chunk by chunk (pipe.direct_engine 1: pageable copies issued by the calling thread and one mover thread; 0: staged through the library's pinned buffers by copy threads); PCIe-bound, never the reported value"})
    return out


def small_call_latency():
    """The reference's criterion shapes (benches/simd_comparison.rs:19-89: as_2bit / from_2bit at 4-32 bases,
    encode / decode at 1-1024 bases, cyclic ACGT input) through the drop-in's host path (size-dispatched SWAR code
    inside libbitnuc_hip.so), timed in C by the library itself (bitnuc_selftime_small)."""
    import ctypes as C
    from bitnuc_amd import _lib as L
    lib = L.load()
    out = {}
    for name, op, sizes in (("as_2bit", 0, (4, 8, 16, 31, 32)), ("from_2bit", 1, (4, 8, 16, 31, 32)),
                            ("encode", 2, (1, 16, 32, 128, 1000, 1024)), ("decode", 3, (1, 16, 32, 128, 1000, 1024)),
                            ("hdist_scalar", 4, (32,))):
        out[name] = {str(s): round(lib.bitnuc_selftime_small(op, s, 200000), 2) for s in sizes}
    out["unit"] = "ns per call (host path, median-free mean over 2e5 calls, C loop inside the library)"
    out["gpu_launch_path_us"] = "a forced-GPU single call costs ~35 us (copy in, launch, copy out, one wait): profiles/NARRATIVE_r01_r03.md section 1"
    return out


def main():
    args = parse_args()
    if args.traffic_child:
        traffic_child(args)
        return 0
    in_rank = "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not in_rank:
        return self_launch(args, sys.argv[1:])
    # stdout carries exactly ONE line (the JSON): libraries that print banners from C (RCCL does
    # at communicator init) are sent to stderr by pointing fd 1 there and keeping the real stdout aside
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    traffic = None
    if not args.rehearse_cpu:
        if args.gpus == 1 and not in_rank and not args.no_traffic:
            from bitnuc_amd import build as bn_build
            bn_build.ensure_built()  # before any child runs under rocprofv3: a profiled process must not start the compiler
            traffic = measure_traffic_live(args.bases)  # child processes under rocprofv3; this process has not touched the GPU yet
            if "error" in traffic:
                print(f"[bench] live PMC pass unavailable: {traffic['error']}", file=sys.stderr)
                stored = stored_traffic()
                traffic = stored if "error" not in stored else {"error": traffic["error"] + " | " + stored["error"]}
        else:
            traffic = stored_traffic()
    return run_rank(args, real_stdout, traffic)


if __name__ == "__main__":
    sys.exit(main())
