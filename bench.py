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
"""
import argparse
import json
import os
import statistics
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s)
BYTES_PER_BASE = 1.25          # encode: 1 read + 0.25 write; decode: 0.25 read + 1 write
SEED = 0xB17C0DE


def cpu_baseline(n_sample, reps, all_cores):
    """Reference-algorithm restatement (oracle/bitnuc_avx2.c, the reference's AVX2 path as
    written) timed on this host.  kind = "port": the Rust reference cannot be built here."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_py
    seq = oracle_py.nucgen(n_sample, SEED)
    enc, dec = [], []
    for _ in range(reps):
        e, d = oracle_py.avx2_time_roundtrip(seq)
        enc.append(e)
        dec.append(d)
    e1, d1 = statistics.median(enc), statistics.median(dec)
    out = {"value": round(2 * n_sample / (e1 + d1) / 1e9, 4), "unit": "Gbases/s", "cores": 1, "kind": "port",
           "sample": f"{n_sample:.3g} bases of the same seeded stream, encode+decode, median of {reps}",
           "encode_gbases_s": round(n_sample / e1 / 1e9, 4), "decode_gbases_s": round(n_sample / d1 / 1e9, 4),
           "what": "C restatement of the reference's AVX2 path as written (oracle/bitnuc_avx2.c); "
                   "the reference itself is single-threaded Rust and cannot be built in this image"}
    if all_cores:
        # a one-GPU box's CPU share is 16 cores (os.cpu_count() reports the whole host)
        cores = min(len(os.sched_getaffinity(0)), 16)
        per = (n_sample // cores) // 32 * 32
        if per > 0:
            res = [None] * cores

            def work(i):
                res[i] = oracle_py.avx2_time_roundtrip(seq[i * per:(i + 1) * per])
            t0 = time.perf_counter()
            th = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
            [t.start() for t in th]
            [t.join() for t in th]
            wall = time.perf_counter() - t0
            out["all_cores"] = {"value": round(2 * per * cores / wall / 1e9, 4), "cores": cores,
                                "note": "courtesy split at 32-base boundaries over threads; the reference has no threading"}
    try:
        with open("/proc/cpuinfo") as f:
            out["cpu"] = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except Exception:
        pass
    return out


def main():
    # stdout carries exactly ONE line (the JSON): libraries that print banners from C (RCCL does
    # at communicator init) are sent to stderr by pointing fd 1 there and keeping the real stdout aside
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--bases", type=int, default=10**9, help="bases per GPU per step (BASELINE configs[1]: 1e9)")
    ap.add_argument("--rotate", type=int, default=3, help="buffer sets rotated so the 256 MiB Infinity Cache cannot serve a step (>= 2)")
    ap.add_argument("--enc-variant", type=int, default=-1)
    ap.add_argument("--dec-variant", type=int, default=-1)
    ap.add_argument("--grid-mult", type=int, default=-1)
    ap.add_argument("--cpu-sample", type=int, default=10**9, help="bases timed on the CPU (default: the whole configs[1] workload)")
    ap.add_argument("--cpu-reps", type=int, default=15, help="repetitions of the 10^9-base CPU round trip (median reported): about 12 s of single-core work")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the config-3 (k-mer batch) and config-5 (scan) side measurements")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); gloo only to rehearse the N>1 control flow")
    ap.add_argument("--force-dist", action="store_true", help="rehearsal only: initialise torch.distributed and run the collectives even at world size 1")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0 (1-GPU box)")
    ap.add_argument("--warm-decode", action="store_true", help="decode the words the same step just encoded (a literal round trip: 20 %% of decode's input then comes from the Infinity Cache)")
    ap.add_argument("--probe", action="store_true", help="also time pure read/copy/fill kernels (the box's own HBM ceiling)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import bitnuc_amd
    from bitnuc_amd import build as bn_build

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # a fresh checkout has no libbitnuc_hip.so (git-ignored): local rank 0 compiles it, the others wait for the file
    if local_rank == 0:
        bn_build.ensure_built()
    else:
        t_wait = time.time()
        while not os.path.exists(bn_build.LIB):
            if time.time() - t_wait > 600:
                raise RuntimeError(f"{bn_build.LIB} did not appear: there is no CPU fallback")
            time.sleep(1.0)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.share_gpu:
        local_rank = 0
    # a launcher that narrows each rank's visible devices (HIP_VISIBLE_DEVICES per rank) leaves fewer devices than ranks
    local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    control_backend = None
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = args.backend
        if backend == "nccl":
            try:
                dist.init_process_group("nccl", device_id=dev)  # eager: a broken RCCL setup fails here, not at the first barrier
                probe = torch.zeros(1, device=dev)
                dist.all_reduce(probe)
                torch.cuda.synchronize()
            except Exception as e:  # noqa: BLE001
                # The step has no data-path collective: RCCL only carries the barrier and the max-reduce of the timing.
                # If it cannot come up, the same control traffic goes over gloo and the measurement stays valid.
                print(f"[bench] nccl unavailable ({e!r}); using gloo for the barrier / max-reduce", file=sys.stderr)
                try:
                    dist.destroy_process_group()
                except Exception:  # noqa: BLE001
                    pass
                backend = "gloo"
                dist.init_process_group("gloo")
        else:
            dist.init_process_group(backend)
        control_backend = backend
    on_gpu_collectives = use_dist and control_backend == "nccl"

    n = args.bases
    nw = (n + 31) // 32
    stream = torch.cuda.current_stream()
    ctx = bitnuc_amd.Context(local_rank, stream=stream.cuda_stream)
    if args.enc_variant >= 0:
        ctx.set_variant("encode", args.enc_variant)
    if args.dec_variant >= 0:
        ctx.set_variant("decode", args.dec_variant)
    if args.grid_mult >= 0:
        ctx.set_variant("grid_mult", args.grid_mult)

    R = max(2, args.rotate)
    seqs = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(R)]
    words = [torch.empty(nw, dtype=torch.int64, device=dev) for _ in range(R)]
    backs = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(R)]
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

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    ctx.sync()
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, events[i])
    fence()
    t1 = time.perf_counter()
    ctx.sync()  # raises if any launch latched an InvalidBase

    # parity guard inside the bench: the last step's round trip must be the identity
    r_last = (args.steps - 1) % R if args.warm_decode else ((args.steps - 1) % R + 1) % R
    assert torch.equal(seqs[r_last], backs[r_last]), "decode(encode(x)) != x"

    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev if on_gpu_collectives else "cpu")
    if use_dist:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    sec_per_step = float(elapsed.item()) / args.steps
    enc_ms = [e[0].elapsed_time(e[1]) for e in events]
    dec_ms = [e[1].elapsed_time(e[2]) for e in events]
    enc_avg, dec_avg = sum(enc_ms) / len(enc_ms), sum(dec_ms) / len(dec_ms)

    def make_line(extra, with_cpu=True):
        total_bases = world * 2 * n  # encoded + decoded, all ranks, per step
        enc_gbs = n * BYTES_PER_BASE / (enc_avg * 1e-3) / 1e9
        dec_gbs = n * BYTES_PER_BASE / (dec_avg * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")  # written by tools/prof_summary.py from a --pmc run
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("encode_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "Gbases/s encode+decode on 10^9-base synthetic; % of HBM3E roofline",
            "value": round(total_bases / sec_per_step / 1e9, 2),
            "unit": "Gbases/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(sec_per_step * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            # the same measurement, three readings (value is the first): bases through the codec per
            # second (encoded + decoded), and the per-kernel rates from the HIP events of rank 0
            "codec_gbases_s": round(total_bases / sec_per_step / 1e9, 2),
            "roundtrip_gbases_s": round(world * n / sec_per_step / 1e9, 2),
            "encode_gbases_s": round(world * n / (enc_avg * 1e-3) / 1e9, 1),
            "decode_gbases_s": round(world * n / (dec_avg * 1e-3) / 1e9, 1),
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: bulk encode + decode of 10^9 random bases per GPU, device-resident, bit-exact vs CPU oracle",
                       "bases_per_gpu_per_step": n, "bases_counted_per_step": "encoded + decoded = 2 x bases_per_gpu_per_step x n_gpus",
                       "seed": hex(SEED), "rotating_buffer_sets": R,
                       "decode_input": "words encoded in the same step (Infinity-Cache warm)" if args.warm_decode else f"words encoded {R - 1} steps earlier (HBM resident, cache cold)",
                       "encode_variant": ctx.get("encode"), "decode_variant": ctx.get("decode"), "grid_mult": ctx.get("grid_mult"),
                       "parallelism": f"shard{world}" if world > 1 else "single",
                       "control_backend": control_backend},
            "roofline": {"kernel": "encode_kernel", "bound": "hbm", "achieved": round(enc_gbs, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(enc_gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "algorithmic_bytes_per_launch": n * BYTES_PER_BASE, "avg_launch_ms": round(enc_avg, 4),
                         "gbases_s": round(n / (enc_avg * 1e-3) / 1e9, 1)},
            "roofline_decode": {"kernel": "decode_kernel", "bound": "hbm", "achieved": round(dec_gbs, 1), "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": round(dec_gbs / HBM_PEAK_GBS, 4),
                                "algorithmic_bytes_per_launch": n * BYTES_PER_BASE, "avg_launch_ms": round(dec_avg, 4),
                                "gbases_s": round(n / (dec_avg * 1e-3) / 1e9, 1)},
        }
        line.update(extra)
        pr = extra.get("stream_probe_gb_s", {})
        if "read" in pr:  # report % of nominal AND % of the box's measured streaming peak
            best = max(pr["read"], pr["copy"], pr["fill"])
            line["roofline"]["measured_stream_peak"] = best
            line["roofline"]["frac_of_measured"] = round(enc_gbs / best, 4)
            line["roofline_decode"]["measured_stream_peak"] = best
            line["roofline_decode"]["frac_of_measured"] = round(dec_gbs / best, 4)
        if with_cpu and world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.cpu_reps, all_cores=True)
            except Exception as e:  # noqa: BLE001
                line["cpu_baseline"] = {"error": repr(e)[:300]}
        return line

    emit_lock = threading.Lock()
    emitted = []

    def emit(extra_now, with_cpu=True):
        """Print the one JSON line (rank 0), exactly once."""
        with emit_lock:
            if emitted:
                return
            emitted.append(True)
            if rank == 0:
                line = make_line(extra_now, with_cpu)
                sys.stdout.flush()
                os.write(real_stdout, (json.dumps(line) + "\n").encode())

    extra = {}
    watchdog = None
    if use_dist and on_gpu_collectives:  # config 4's concatenation, reported beside the step, never inside it
        # These side measurements are collectives: a rank that fails alone would leave the others waiting.  The headline
        # must not depend on them, so a watchdog prints it without the side numbers and ends the process if they (or the
        # process-group teardown after them) stall; it stays armed until main() returns.
        def bail():
            emit({"allgather_packed": {"error": "side measurement did not finish in 300 s; headline printed without it"}})
            os._exit(0)
        watchdog = threading.Timer(300.0, bail)
        watchdog.daemon = True
        watchdog.start()
        try:
            from bitnuc_amd.dist import allgather_packed
            allgather_packed(words[0])
            fence()
            t = time.perf_counter()
            reps = 5
            for _ in range(reps):
                full = allgather_packed(words[0])
            fence()
            ag = (time.perf_counter() - t) / reps
            del full
            extra["allgather_packed"] = {"ms": round(ag * 1e3, 3), "bytes_received_per_gpu": nw * 8 * (world - 1),
                                         "gb_s_per_gpu": round(nw * 8 * (world - 1) / ag / 1e9, 2),
                                         "note": "RCCL all-gather of the packed u64 buffer over xGMI; fabric-bound, outside the timed step"}
            # encode + concatenation end to end, one shot vs chunked overlap (8 pieces: the
            # fabric moves piece c while the GPU encodes piece c+1)
            from bitnuc_amd.dist import encode_allgather_overlapped

            def enc_chunk(w0, w1):
                ctx.encode_dev(seqs[0][32 * w0:], min(n, 32 * w1) - 32 * w0, words[0][w0:])
                return words[0][w0:w1]

            def one_shot():
                ctx.encode_dev(seqs[0], n, words[0])
                return allgather_packed(words[0])
            e2e = {}
            for name, fn in (("one_shot", one_shot), ("overlap8", lambda: encode_allgather_overlapped(enc_chunk, nw, 8, words[0]))):
                ref_full = fn()
                fence()
                t = time.perf_counter()
                for _ in range(reps):
                    full = fn()
                fence()
                e2e[name + "_ms"] = round((time.perf_counter() - t) / reps * 1e3, 3)
                e2e[name + "_ok"] = bool(torch.equal(full, ref_full))
                del full, ref_full
            e2e["note"] = "encode of this rank's 10^9-base shard + all-gather of the packed words; fabric-bound, so it cannot scale like the step"
            extra["encode_allgather_end_to_end"] = e2e
        except Exception as e:  # noqa: BLE001 -- a side measurement must never cost the headline line
            extra["allgather_packed"] = {"error": repr(e)[:300]}
    if world == 1 and not args.no_extras:
        try:
            # BASELINE configs[2] and [4], measured beside the headline (never inside the timed step)
            del backs[1:], seqs[1:], words[1:]
            torch.cuda.empty_cache()

            def timed(fn, reps=10):
                ms = []
                for _ in range(reps + 2):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(stream)
                    fn()
                    b.record(stream)
                    torch.cuda.synchronize()
                    ms.append(a.elapsed_time(b))
                return statistics.median(ms[2:])
            count, k = 10**8, 31
            kseq = torch.empty(count * k, dtype=torch.uint8, device=dev)
            ctx.nucgen_dev(kseq, count * k, SEED + 100)
            kout = torch.empty(count, dtype=torch.int64, device=dev)
            ms = timed(lambda: ctx.as_2bit_batch_dev(kseq, k, k, count, kout))
            gbs = count * (k + 8) / (ms * 1e-3) / 1e9
            extra["kmer_batch"] = {"workload": "BASELINE configs[2]: 10^8 dense 31-mers as_2bit -> u64", "gkmers_s": round(count / (ms * 1e-3) / 1e9, 2),
                                   "ms": round(ms, 4), "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                                    "frac": round(gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": count * (k + 8)}}
            del kseq, kout
            # every window of a sequence (`seq.windows(k)` + as_2bit, src/lib.rs:170-173): stride 1, 1 B read + 8 B written per window
            nwin = n - k + 1
            wout = torch.empty(nwin, dtype=torch.int64, device=dev)
            ms = timed(lambda: ctx.as_2bit_batch_dev(seqs[0], k, 1, nwin, wout), reps=6)
            gbs = 9 * nwin / (ms * 1e-3) / 1e9
            extra["kmer_windows"] = {"workload": "as_2bit of every 31-base window of 10^9 bases (stride 1) -> u64 per window", "ms": round(ms, 4),
                                     "gwindows_s": round(nwin / (ms * 1e-3) / 1e9, 2),
                                     "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                  "frac": round(gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": 9 * nwin}}
            del wout
            dist_out = torch.empty(n - k + 1, dtype=torch.uint8, device=dev)
            q = 0x1B1B1B1B1B1B1B1B & ((1 << 62) - 1)
            ms = timed(lambda: ctx.kmer_hdist_scan_dev(seqs[0], n, k, q, dist_out))
            gbs = 2 * (n - k + 1) / (ms * 1e-3) / 1e9
            extra["kmer_hdist_scan"] = {"workload": "BASELINE configs[4]: sliding 31-mer pack + Hamming distance to one query over 10^9 bases",
                                        "gwindows_s": round((n - k + 1) / (ms * 1e-3) / 1e9, 2), "ms": round(ms, 4),
                                        "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                     "frac": round(gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": 2 * (n - k + 1)}}
            # bulk packed-vs-packed Hamming distance (hdist, hamming/multi.rs:121-160): 16 B per 32-base word pair
            wa, wb = words[0], torch.empty(nw, dtype=torch.int64, device=dev)
            ctx.nucgen_dev(backs[0], n, SEED + 200)
            ctx.encode_dev(backs[0], n, wb)
            res = torch.zeros(1, dtype=torch.int32, device=dev)
            ms = timed(lambda: ctx.hdist_dev(wa, nw, wb, nw, n, res))
            gbs = 16 * nw / (ms * 1e-3) / 1e9
            extra["hdist_bulk"] = {"workload": "hdist of two 10^9-base packed buffers (SURVEY 8f rank 1)", "ms": round(ms, 4),
                                   "gbases_s": round(n / (ms * 1e-3) / 1e9, 1), "distance": int(res.item()) & 0xFFFFFFFF,
                                   "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                "frac": round(gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": 16 * nw}}
            # analysis directly on packed words
            cnt = torch.zeros(4, dtype=torch.int64, device=dev)
            # these two read only 250 MB per launch, which would fit the 256 MiB Infinity Cache:
            # alternate between two packed buffers so that every launch streams from HBM
            flip = [0]

            def alt():
                flip[0] ^= 1
                return wb if flip[0] else wa
            ms = timed(lambda: ctx.base_counts_dev(alt(), nw, n, cnt))
            gbs = 8 * nw / (ms * 1e-3) / 1e9
            extra["base_counts"] = {"workload": "A/C/G/T counts of 10^9 packed bases (analysis.rs:23-39 without the decode)", "ms": round(ms, 4),
                                    "gbases_s": round(n / (ms * 1e-3) / 1e9, 1),
                                    "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                 "frac": round(gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": 8 * nw}}
            qd = torch.empty(nw, dtype=torch.uint8, device=dev)
            ms = timed(lambda: ctx.hdist_query_dev(0x1B1B1B1B1B1B1B1B, alt(), nw, 32, qd))
            gbs = 9 * nw / (ms * 1e-3) / 1e9
            extra["hdist_query"] = {"workload": "one packed 32-mer against 3.1e7 packed 32-mers -> u8 distances", "ms": round(ms, 4),
                                    "gwords_s": round(nw / (ms * 1e-3) / 1e9, 2),
                                    "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                 "frac": round(gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": 9 * nw}}
            del qd
            # split_packed at an odd base in the middle (16 B per word: read once, write once)
            sidx = n // 2 + 5
            snl, snr = ctx.split_packed_sizes(nw, n, sidx, canonical=True)
            sl, sr = torch.empty(snl, dtype=torch.int64, device=dev), torch.empty(snr, dtype=torch.int64, device=dev)
            ms = timed(lambda: ctx.split_packed_dev(alt(), nw, n, sidx, sl, sr, canonical=True))
            gbs = 8 * (nw + snl + snr) / (ms * 1e-3) / 1e9
            extra["split_packed"] = {"workload": "split 10^9 packed bases at base n/2+5 (functions/split.rs:15-99, funnel-shift form)", "ms": round(ms, 4),
                                     "gbases_s": round(n / (ms * 1e-3) / 1e9, 1),
                                     "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                  "frac": round(gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": 8 * (nw + snl + snr)}}
            del sl, sr
            # ragged batch of independent sequences: 150-base reads (each read pads its own last word)
            L, rcount = 150, n // 150
            roff = torch.arange(0, rcount + 1, dtype=torch.int64, device=dev) * L
            rwo = torch.empty(rcount + 1, dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            rtotal = ctx.batch_word_offsets_dev(roff, rcount, rwo)
            rwords = torch.empty(rtotal, dtype=torch.int64, device=dev)
            ms_e = timed(lambda: ctx.encode_batch_dev(seqs[0], roff, rwo, rcount, rtotal, rwords))
            ms_d = timed(lambda: ctx.decode_batch_dev(rwords, rwo, roff, rcount, rtotal, backs[0]))
            rb = L * rcount
            alg = rb + 8 * rtotal  # bases + packed words (offset tables: +16 B per read, not counted)
            extra["reads_batch"] = {"workload": f"{rcount} independent 150-base reads, encode_batch / decode_batch (each read pads its own last word)",
                                    "encode_ms": round(ms_e, 4), "decode_ms": round(ms_d, 4),
                                    "encode_gbases_s": round(rb / (ms_e * 1e-3) / 1e9, 1), "decode_gbases_s": round(rb / (ms_d * 1e-3) / 1e9, 1),
                                    "encode_gb_s": round(alg / (ms_e * 1e-3) / 1e9, 1), "decode_gb_s": round(alg / (ms_d * 1e-3) / 1e9, 1),
                                    "algorithmic_bytes_per_launch": alg}
            ms_fe = timed(lambda: ctx.encode_fixed_dev(seqs[0], L, L, rcount, rwords))
            ms_fd = timed(lambda: ctx.decode_fixed_dev(rwords, L, L, rcount, backs[0]))
            extra["reads_fixed"] = {"workload": f"{rcount} fixed-length 150-base reads, encode_fixed / decode_fixed (no offsets tables)",
                                    "encode_ms": round(ms_fe, 4), "decode_ms": round(ms_fd, 4),
                                    "encode_gbases_s": round(rb / (ms_fe * 1e-3) / 1e9, 1), "decode_gbases_s": round(rb / (ms_fd * 1e-3) / 1e9, 1),
                                    "encode_gb_s": round(alg / (ms_fe * 1e-3) / 1e9, 1), "decode_gb_s": round(alg / (ms_fd * 1e-3) / 1e9, 1),
                                    "encode_frac": round(alg / (ms_fe * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "decode_frac": round(alg / (ms_fd * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                    "algorithmic_bytes_per_launch": alg}
            del rwords, roff, rwo
            ctx.sync()
            backs.append(dist_out)  # reused by the probe below
        except Exception as e:  # noqa: BLE001 -- side measurements must never cost the headline line
            extra["extras_error"] = repr(e)[:300]
    if args.probe or (world == 1 and not args.no_extras):
        # the box's own streaming ceiling, same harness: best of a few cache-policy variants per shape
        try:
            def probe_rate(mode, moved):
                ms = []
                for i in range(8):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(stream)
                    ctx.stream_probe_dev(mode, seqs[i % len(seqs)], backs[(i + 1) % len(backs)], min(n, backs[(i + 1) % len(backs)].numel()))
                    b.record(stream)
                    torch.cuda.synchronize()
                    ms.append(a.elapsed_time(b))
                return round(moved / (statistics.median(ms[2:]) * 1e-3) / 1e9, 1)
            nb = min(n, min(t.numel() for t in backs))
            probe = {"read": max(probe_rate(m, nb) for m in (0 | 8, 0 | 8 | 32, 0)),
                     "copy": max(probe_rate(m, 2 * nb) for m in (1 | 8 | 16, 1 | 8, 1 | 16, 1)),
                     "fill": max(probe_rate(m, nb) for m in (2 | 16, 2))}
            extra["stream_probe_gb_s"] = probe
        except Exception as e:  # noqa: BLE001
            extra["stream_probe_gb_s"] = {"error": repr(e)[:300]}
    emit(extra)
    ctx.close()
    if use_dist:
        dist.destroy_process_group()
    if watchdog is not None:
        watchdog.cancel()


if __name__ == "__main__":
    main()
