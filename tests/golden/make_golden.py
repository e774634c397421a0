#!/usr/bin/env python3
"""Writes tests/golden/golden.json.

The reference (a Rust crate) cannot be compiled or run in this image (no
rustc/cargo), so these vectors are NOT outputs of a reference run: they are the
known-answer inputs/expected values that the reference's own unit tests,
doc-tests and bench asserts hold, transcribed as data, each with the
file:line (relative to the reference root) it comes from.  Only inputs and
expected outputs are recorded -- no reference source text.

Run: python tests/golden/make_golden.py   (rewrites golden.json next to it)
"""
import json
import os

V = {"as_2bit": [], "as_2bit_err": [], "from_2bit": [], "from_2bit_err": [],
     "from_2bit_append": [], "roundtrip_strings": [], "roundtrip_prefixes": [],
     "roundtrip_lengths": {}, "hdist_scalar": [], "hdist_scalar_err": [],
     "hdist": [], "hdist_err": [], "hdist_cyclic": [], "kmer_count": []}

# --- as_2bit known answers ---------------------------------------------------
for seq, val, src in [
    ("ACGT", 0b11100100, "src/utils/packing/mod.rs:151"),
    ("AAAA", 0b00000000, "src/utils/packing/mod.rs:152"),
    ("TTTT", 0b11111111, "src/utils/packing/mod.rs:153"),
    ("GGGG", 0b10101010, "src/utils/packing/mod.rs:154"),
    ("CCCC", 0b01010101, "src/utils/packing/mod.rs:155"),
    ("ACTGACTGACTGACTG", 0b10110100101101001011010010110100, "src/utils/packing/mod.rs:165-168"),
    ("ACTGGAAAATTTTAAGG", 0b1010000011111111000000001010110100, "src/utils/packing/mod.rs:173"),
    ("ACGT", 0b11100100, "src/utils/packing/mod.rs:45-46 (doc-test); README.md:25-26"),
]:
    V["as_2bit"].append({"seq": seq, "packed": val, "src": src})
V["as_2bit_case_insensitive"] = {"lower": "acgt", "upper": "ACGT", "src": "src/utils/packing/mod.rs:181,56"}
V["as_2bit_err"] += [
    {"seq": "ACGN", "status": "InvalidBase", "byte": ord("N"), "src": "src/utils/packing/mod.rs:186-187,68-69; src/lib.rs:146-147"},
    {"seq": "A" * 33, "status": "SequenceTooLong", "value": 33, "src": "src/utils/packing/mod.rs:192-196,73-77; src/lib.rs:150-152"},
]

# --- from_2bit known answers ---------------------------------------------------
for packed, n, seq, src in [
    (0b11100100, 4, "ACGT", "src/utils/unpacking/mod.rs:191"),
    (0b00000000, 4, "AAAA", "src/utils/unpacking/mod.rs:192"),
    (0b11111111, 4, "TTTT", "src/utils/unpacking/mod.rs:193"),
    (71620941647064936, 28, "AGGCTTGAGGCCCATTCTCTGATCGTTT", "src/utils/unpacking/mod.rs:206-214"),
    (0b11100100, 2, "AC", "src/utils/mod.rs:103-105; src/utils/unpacking/mod.rs:83-84"),
    (0b11100100, 3, "ACG", "src/utils/mod.rs:107-109"),
]:
    V["from_2bit"].append({"packed": packed, "n": n, "seq": seq, "src": src})
V["from_2bit_err"].append({"packed": 0, "n": 33, "status": "InvalidLength", "value": 33,
                           "src": "src/utils/unpacking/mod.rs:95-98"})
# append semantics: two calls with n=10 on pack("ACTG"*5) give this concatenation
V["from_2bit_append"].append({"seq": "ACTGACTGACTGACTGACTG", "n": 10, "calls": 2,
                              "expected": "ACTGACTGACACTGACTGAC",
                              "src": "src/utils/unpacking/avx.rs:185-194"})
V["from_2bit_simd20"] = {"seq": "ACTGACTGACTGACTGACTG", "n": 20, "src": "src/utils/unpacking/avx.rs:161-168"}

# --- round trips -----------------------------------------------------------------
V["roundtrip_strings"] = {"cases": ["A", "C", "G", "T", "AC", "GT", "ACG", "TGC", "ACGT", "TGCA",
                                    "ACGTACGT", "AAAA", "CCCC", "GGGG", "TTTT"],
                          "src": "src/utils/mod.rs:71-96"}
V["roundtrip_prefixes"] = {"seq": "ACTGACTGACTGACTGACTGACTGACTGACTG", "lens": [1, 32],
                           "src": "src/utils/unpacking/avx.rs:172-181"}
V["roundtrip_lengths"] = {"lens": [1, 1000], "alphabet": "ACGT", "src": "src/utils/mod.rs:113-133",
                          "note": "reference uses nucgen + thread_rng (unseeded); property is input-agnostic"}

# --- hamming -----------------------------------------------------------------------
V["hdist_scalar"] += [
    {"u": 0, "v": 0, "len": 0, "d": 0, "src": "src/utils/functions/hamming/scalar.rs:58"},
    {"u": 0, "v": 0, "len": 32, "d": 0, "src": "src/utils/functions/hamming/scalar.rs:59"},
    {"u": 0, "v": 0, "len": 1, "d": 0, "src": "src/utils/functions/hamming/scalar.rs:65"},
    {"u": 0xFFFFFFFF, "v": 0xFFFFFFFF, "len": 16, "d": 0, "src": "src/utils/functions/hamming/scalar.rs:66"},
    {"u": 0xFFFFFFFFFFFFFFFF, "v": 0xFFFFFFFFFFFFFFFF, "len": 32, "d": 0, "src": "src/utils/functions/hamming/scalar.rs:67-70"},
    {"u": 0b0001, "v": 0b0010, "len": 2, "d": 1, "src": "src/utils/functions/hamming/scalar.rs:85"},
    {"u": 0b0001, "v": 0b0011, "len": 2, "d": 1, "src": "src/utils/functions/hamming/scalar.rs:86"},
    {"u": 0b0010, "v": 0b0011, "len": 2, "d": 1, "src": "src/utils/functions/hamming/scalar.rs:87"},
]
V["hdist_scalar_strings"] = {"cases": [["AAAA", "AAAA", 0], ["AAAA", "AAAT", 1], ["AAAA", "AATT", 2],
                                       ["AAAA", "ATTT", 3], ["AAAA", "TTTT", 4],
                                       ["ACTGACTG", "TGCATGCA", 8]],
                             "src": "src/utils/functions/hamming/scalar.rs:93-100"}
V["hdist_scalar_err"].append({"u": 0, "v": 0, "len": 33, "status": "InvalidLength", "value": 33,
                              "src": "src/utils/functions/hamming/scalar.rs:57"})
V["hdist_err"].append({"na": 1, "nb": 1, "n_bases": 64, "status": "InvalidLength", "value": 64,
                       "src": "src/utils/functions/hamming/multi.rs:169-172"})
V["hdist"] += [
    {"seq1": "ACTG" * 16, "seq2": "ACTG" * 16, "d": 0, "src": "src/utils/functions/hamming/multi.rs:176-179"},
    {"seq1": "A" * 128, "seq2": "T" * 128, "d": 128, "src": "src/utils/functions/hamming/multi.rs:185-189"},
]
V["hdist_A_vs_T"] = {"lens": [1, 256], "src": "src/utils/functions/hamming/multi.rs:194-206"}
# bench invariant: bases[i % 4] vs bases[i % 3]; the bench asserts the packed
# distance equals the byte-wise mismatch count (no literal number in the source)
V["hdist_cyclic"] = [{"l": 32, "mod1": 4, "mod2": 3, "src": "benches/hdist_benchmark.rs:17-37"},
                     {"l": 512, "mod1": 4, "mod2": 3, "src": "benches/hdist_benchmark.rs:52-72"}]
# k-mer counting doc example: windows(4) of ACGTACGT, count of ACGT == 2
V["kmer_count"] = {"seq": "ACGTACGT", "k": 4, "kmer": "ACGT", "count": 2, "src": "src/lib.rs:170-178"}

# --- analysis on packed sequences -------------------------------------------
V["gc_content"] = [{"seq": s_, "gc": g_, "src": "src/utils/analysis.rs:48-54"} for s_, g_ in
                   [("ACGT", 50.0), ("AAAA", 0.0), ("CCCC", 100.0), ("AACG", 50.0), ("ACGTA", 40.0)]]
V["base_counts"] = [{"seq": s_, "counts": c_, "src": "src/utils/analysis.rs:64-70"} for s_, c_ in
                    [("ACGT", [1, 1, 1, 1]), ("AAAA", [4, 0, 0, 0]), ("CCCC", [0, 4, 0, 0]), ("AACG", [2, 1, 1, 0]),
                     ("ACGTA", [2, 1, 1, 1])]]
V["empty_sequence_analysis"] = {"gc": 0.0, "counts": [0, 0, 0, 0], "src": "src/utils/analysis.rs:79-83"}

# --- PackedSequence ---------------------------------------------------------
V["packed_sequence"] = {
    "new": {"seq": "ACGT", "len": 4, "to_vec": "ACGT", "src": "src/sequence.rs:271-275"},
    "get": {"seq": "ACGT", "bases": ["A", "C", "G", "T"], "src": "src/sequence.rs:278-284"},
    "get_oob": {"seq": "ACGT", "index": 4, "length": 4, "src": "src/sequence.rs:287-296"},
    "slices": [{"seq": "ACGTACGT", "start": 1, "end": 5, "out": "CGTA", "src": "src/sequence.rs:299-302,151"},
               {"seq": "ACGTACGT", "start": 0, "end": 3, "out": "ACG", "src": "src/sequence.rs:154"},
               {"seq": "ACGTACGT", "start": 5, "end": 8, "out": "CGT", "src": "src/sequence.rs:157"},
               {"seq": "ACGTACGT", "start": 2, "end": 2, "out": "", "src": "src/sequence.rs:168"}],
    "invalid_slice": {"seq": "ACGT", "start": 3, "end": 2, "length": 4, "src": "src/sequence.rs:306-316"},
    "equality": {"same": ["ACGT", "ACGT"], "different": ["ACGT", "TGCA"], "src": "src/sequence.rs:319-338"},
    "invalid": {"seq": "ACGN", "src": "src/sequence.rs:36-37"},
    "empty": {"seq": "", "len": 0, "to_vec": "", "src": "src/sequence.rs:42-46,80,250"},
}

# crate-level integration tests (src/lib.rs:222-265) and hashability (src/sequence.rs:328-338)
V["crate_integration"] = {
    "creation_and_analysis": {"seq": "ACGTACGT", "len": 8, "is_empty": False, "to_vec": "ACGTACGT", "gc": 50.0,
                              "counts": [2, 2, 2, 2], "src": "src/lib.rs:226-240"},
    "mutations": {"seq": "ACGTACGT", "slice": [2, 6, "GTAC"], "get": [[0, "A"], [7, "T"]], "src": "src/lib.rs:242-253"},
    "error_handling": {"invalid": "ACGN", "seq": "ACGT", "get_oob": 4, "slice_oob": [2, 5], "src": "src/lib.rs:255-264"},
    "hashability": {"in_set": ["ACGT", "ACGT"], "not_in_set": "TGCA", "src": "src/sequence.rs:328-338"},
}

# --- split_packed -----------------------------------------------------------
# each case: encode(seq), split at idx -> word counts the test asserts, and the decoded halves
V["split_packed"] = [
    {"seq": "ACTGACTG", "idx": 4, "n_left": 1, "n_right": 1, "left": "ACTG", "right": "ACTG",
     "src": "src/utils/functions/split.rs:108-133"},
    {"seq": "ACTG", "idx": 0, "n_left": 0, "n_right": 1, "left": "", "right": "ACTG",
     "src": "src/utils/functions/split.rs:136-152"},
    {"seq": "ACTG", "idx": 4, "n_left": 1, "n_right": 0, "left": "ACTG", "right": "",
     "src": "src/utils/functions/split.rs:154-161"},
    {"seq": "ACTGACTGAC", "idx": 7, "n_left": 1, "n_right": 1, "left": "ACTGACT", "right": "GAC",
     "src": "src/utils/functions/split.rs:164-187"},
    {"seq": "ACTG" * 10, "idx": 32, "n_left": 2, "n_right": 1, "left": "ACTG" * 8, "right": "ACTGACTG",
     "src": "src/utils/functions/split.rs:190-213"},
]
V["split_packed_err"] = {"seq": "ACTG", "idx": 5, "status": "IndexOutOfBounds", "index": 5, "length": 4,
                         "src": "src/utils/functions/split.rs:216-224,23-28"}

if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden.json")
    with open(out, "w") as f:
        json.dump(V, f, indent=1, sort_keys=True)
    print("wrote", out)
