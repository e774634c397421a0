#!/usr/bin/env python3
"""Does tests/test_gpu_multirank_mock.py notice a broken comm.hip?  (run on the GPU box; writes nothing into the sources)

Copies bitnuc_amd/csrc to a scratch directory, applies ONE deliberate defect at a time to comm.hip's chunked all-gather,
builds a library from each copy and runs tests/c/multirank_driver.cpp against it and tests/c/mock_rccl.cpp, with the mock's
fabric fast (MOCK_RCCL_DELAY_US=0) and slow (2000).  A defect counts as noticed when at least one scenario fails.
  A  the transfer stream does not wait for the piece's encode        (hipStreamWaitEvent(xfer, piece_done[p]) removed)
  B  the context's stream does not wait for the transfer stream      (~Join's wait removed: bitnuc_ctx_sync returns early)
  C  the transfer stream does not wait for earlier work at call start (removed; REDUNDANT by construction: piece_done[p] is
     recorded later on the same stream, so this one is expected to go unnoticed -- it is listed to show the check is not
     simply failing everything)
  D  one receive lands one word too far                               (addressing)
and the same four in the single-process form (bitnuc_encode_sharded_allgather_overlapped_all, driver mode overlap_all):
  E  = A there, F = B there, G = D there
  H  the per-rank entry points do not refuse a single-process communicator: the driver's refusal check fails (and the call would
     block for the mock's patience, 60 s, before RCCL-like ncclInternalError)
and three in the gather of UNEQUAL word counts of a ragged batch split by whole sequences (round 5: bitnuc_allgatherv_words_dev / _all,
driver modes ragged / ragged_all):
  I  a peer's slot is received one word too far                         (addressing: first[s] + 1)
  J  a rank sends its slot with the NEXT rank's word count              (count mismatch: the mock, stricter than NCCL, fails the pair)
  K  a rank with an empty slot is still sent to / received from as if it held one word (the schedule: messages != non-empty x (P - 1))
"""
import os
import shutil
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bitnuc_amd import build as B

MUT = {
    "A": ("        HIPCHK(hipStreamWaitEvent(comm->xfer, comm->piece_done[(size_t)p], 0));\n", ""),
    "B": ("            if (hipEventRecord(comm->all_moved, comm->xfer) == hipSuccess) (void)hipStreamWaitEvent(c->stream, comm->all_moved, 0);", "            if (false) {}"),
    "C": ("    HIPCHK(hipStreamWaitEvent(comm->xfer, comm->all_moved, 0));\n", ""),
    "D": ("uint64_t *theirs = d_all + (size_t)s * count + w0;", "uint64_t *theirs = d_all + (size_t)s * count + w0 + (s == 1 && p == 2 ? 1 : 0);"),
    "E": ("            if (h == hipSuccess) h = hipStreamWaitEvent(comms[i]->xfer, comms[i]->piece_done[(size_t)p], 0);\n", ""),
    "F": ("            if (hipEventRecord(comms[i]->all_moved, comms[i]->xfer) == hipSuccess) (void)hipStreamWaitEvent(ctxs[i]->stream, comms[i]->all_moved, 0);", "            if (false) {}"),
    "G": ("uint64_t *theirs = d_alls[i] + (size_t)s * count + w0;", "uint64_t *theirs = d_alls[i] + (size_t)s * count + w0 + (s == 1 && p == 2 ? 1 : 0);"),
    "H": ("bool per_rank_call_would_block(const bitnuc_comm *comm) { return comm->single_process && !comm->threaded && comm->nranks > 1; }", "bool per_rank_call_would_block(const bitnuc_comm *) { return false; }"),
    "I": ("if (rc == 0 && counts[s]) rc = r.Recv(d_all + first[s], counts[s], kNcclUint64, s, comm->nccl, stream);", "if (rc == 0 && counts[s]) rc = r.Recv(d_all + first[s] + (s == 1 ? 1 : 0), counts[s], kNcclUint64, s, comm->nccl, stream);"),
    "J": ("if (counts[me]) rc = r.Send(d_all + first[me], counts[me], kNcclUint64, s, comm->nccl, stream);", "if (counts[me]) rc = r.Send(d_all + first[me], counts[(me + 1) % P], kNcclUint64, s, comm->nccl, stream);"),
    "K": ("if (rc == 0 && counts[s]) rc = r.Recv(d_all + first[s], counts[s], kNcclUint64, s, comm->nccl, stream);", "if (rc == 0) rc = r.Recv(d_all + first[s], counts[s] ? counts[s] : 1, kNcclUint64, s, comm->nccl, stream);"),
}
PER_RANK, ALL_RANKS, RAGGED = "ABCD", "EFGH", "IJK"
ODD, BIG = 32 * 100_003, 32 * 4_000_003
SCENARIOS = [(4, ODD, 6, "overlap", 5), (8, ODD, 8, "overlap", 2), (4, BIG, 8, "overlap", 3), (2, BIG, 4, "overlap", 2)]
SCENARIOS_ALL = [(4, ODD, 6, "overlap_all", 3), (8, ODD, 8, "overlap_all", 2), (4, BIG, 8, "overlap_all", 2), (2, BIG, 4, "overlap_all", 2)]
SCENARIOS_RAGGED = [(4, 3000, 11, "ragged", 2), (8, 3000, 11, "ragged", 2), (4, 3000, 11, "ragged_all", 2), (8, 5, 3, "ragged_all", 1), (2, 3000, 5, "ragged", 3), (4, 3000, 7, "ragged_threaded", 2)]


def main():
    hipcc = B.hipcc_path()
    work = tempfile.mkdtemp(prefix="bitnuc_mut_")
    src = os.path.join(work, "a", "csrc")  # runtime.h includes ../../include/bitnuc_hip.h
    shutil.copytree(B.CSRC, src)
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(work, "include"))
    flags = B.CXXFLAGS

    def cc(unit, name=None):
        obj = os.path.join(work, (name or unit) + ".o")
        subprocess.run([hipcc, *flags, "-c", os.path.join(src, (name or unit) + ".hip"), "-o", obj], check=True, cwd=src, capture_output=True)
        return obj
    orig = open(os.path.join(src, "comm.hip")).read()
    for m, (old, new) in MUT.items():
        assert old in orig, m
        open(os.path.join(src, f"comm{m}.hip"), "w").write(orig.replace(old, new))
    with ThreadPoolExecutor(max_workers=8) as ex:
        common = list(ex.map(cc, [u for u in B.UNITS if u != "comm"]))
        comms = dict(zip(["product", *MUT], ex.map(lambda n: cc("comm", n), ["comm"] + [f"comm{m}" for m in MUT])))
    libs = {}
    for name, obj in comms.items():
        d = os.path.join(work, name)
        os.makedirs(d)
        subprocess.run([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-fno-gpu-rdc", "-o", os.path.join(d, "libbitnuc_hip.so"), *common, obj, "-ldl", "-lpthread"], check=True, capture_output=True)
        libs[name] = d
    mock = os.path.join(work, "mock")
    os.makedirs(mock)
    subprocess.run([hipcc, "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-Wl,-soname,librccl.so.1", "-o", os.path.join(mock, "librccl.so.1"), os.path.join(ROOT, "tests", "c", "mock_rccl.cpp")], check=True, capture_output=True)
    exe = os.path.join(work, "multirank_driver")
    subprocess.run([hipcc, "-O1", "-g", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-o", exe, os.path.join(ROOT, "tests", "c", "multirank_driver.cpp"),
                    "-L" + libs["product"], "-lbitnuc_hip", "-ldl", "-lpthread"], check=True, capture_output=True)
    noticed, runs = {}, {}
    for name, d in libs.items():
        for delay in (0, 2000):
            for sc in (SCENARIOS_ALL if name in ALL_RANKS else SCENARIOS_RAGGED if name in RAGGED else SCENARIOS + SCENARIOS_ALL + SCENARIOS_RAGGED if name == "product" else SCENARIOS):
                if name == "H" and (delay or sc[0] != 4 or sc[1] != ODD):
                    continue  # one run shows it (every blocked operation costs the mock's patience, shortened to 0.3 s here)
                env = dict(os.environ, LD_LIBRARY_PATH=os.pathsep.join([d, mock, os.environ.get("LD_LIBRARY_PATH", "")]), MOCK_RCCL_DELAY_US=str(delay), **({"MOCK_RCCL_PATIENCE_MS": "300"} if name in "HJK" else {}))
                r = subprocess.run([exe, *map(str, sc)], capture_output=True, text=True, timeout=300, env=env)
                first = (r.stderr.strip().splitlines() or [""])[0][:110]
                print(f"{name:8s} fabric delay {delay:4d} us  P={sc[0]} shard={sc[1]:>9d} pieces={sc[2]} rounds={sc[4]}: {'ok' if r.returncode == 0 else 'FAILS  ' + first}", flush=True)
                noticed[name] = noticed.get(name, 0) + (r.returncode != 0)
                runs[name] = runs.get(name, 0) + 1
    print()
    for name, k in noticed.items():
        print(f"{name:8s}: {k} of {runs[name]} runs fail")
    shutil.rmtree(work, ignore_errors=True)
    ok = noticed["product"] == 0 and all(noticed[m] > 0 for m in "ABDEFGHIJK")
    print("verdict:", "the product passes every run; defects A, B, D (per-rank form), E, F, G, H (single-process form) and I, J, K (gather of unequal counts) are noticed" if ok else "UNEXPECTED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
