// C++ host-layer tests, shaped after the reference's own in-file unit tests
// (src/utils/packing/mod.rs:144-198, src/utils/unpacking/mod.rs:183-215,
//  src/utils/unpacking/avx.rs:155-196, src/utils/mod.rs:64-134,
//  src/utils/functions/hamming/scalar.rs:50-116, multi.rs:162-208).
// Needs a GPU: include/bitnuc.hpp has no CPU path.  Prints "ok <name>" per test.
#include "bitnuc.hpp"

#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>

using namespace bitnuc;

#define CHECK(cond)                                                                      \
    do {                                                                                 \
        if (!(cond)) { std::fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); std::exit(1); } \
    } while (0)

static void test_as_2bit_valid_sequence() {
    CHECK(as_2bit("ACGT").unwrap() == 0b11100100);
    CHECK(as_2bit("AAAA").unwrap() == 0b00000000);
    CHECK(as_2bit("TTTT").unwrap() == 0b11111111);
    CHECK(as_2bit("GGGG").unwrap() == 0b10101010);
    CHECK(as_2bit("CCCC").unwrap() == 0b01010101);
}
static void test_as_2bit_longer_sequence() {
    CHECK(as_2bit("ACTGACTGACTGACTG").unwrap() == 0b10110100101101001011010010110100ull);
}
static void test_as_2bit_alignments() {
    CHECK(as_2bit("ACTGGAAAATTTTAAGG").unwrap() == 0b1010000011111111000000001010110100ull);
}
static void test_as_2bit_lowercase() { CHECK(as_2bit("acgt").unwrap() == as_2bit("ACGT").unwrap()); }
static void test_as_2bit_invalid_base() {
    auto r = as_2bit("ACGN");
    CHECK(r.is_err() && r.unwrap_err() == NucleotideError::invalid_base('N'));
}
static void test_as_2bit_sequence_too_long() {
    std::vector<uint8_t> long_seq(33, 'A');
    auto r = as_2bit(long_seq);
    CHECK(r.is_err() && r.unwrap_err() == NucleotideError::sequence_too_long(33));
}
static void test_from_2bit_valid_sequence() {
    std::vector<uint8_t> unpacked;
    struct { uint64_t p; size_t n; const char *e; } tests[] = {{0b11100100, 4, "ACGT"}, {0, 4, "AAAA"}, {0xFF, 4, "TTTT"}};
    for (auto &t : tests) {
        from_2bit(t.p, t.n, unpacked).unwrap();
        CHECK(std::string(unpacked.begin(), unpacked.end()) == t.e);
        unpacked.clear();
    }
}
static void test_example_case_from_2bit() {
    auto observed = from_2bit_alloc(71620941647064936ull, 28).unwrap();
    CHECK(std::string(observed.begin(), observed.end()) == "AGGCTTGAGGCCCATTCTCTGATCGTTT");
    auto r = from_2bit_alloc(0, 33);
    CHECK(r.is_err() && r.unwrap_err() == NucleotideError::invalid_length(33));
}
static void test_various_lengths_and_append() {
    const std::string input = "ACTGACTGACTGACTGACTGACTGACTGACTG";
    for (size_t len = 1; len <= 32; ++len) {
        uint64_t packed = as_2bit(Bytes(reinterpret_cast<const uint8_t *>(input.data()), len)).unwrap();
        std::vector<uint8_t> observed;
        from_2bit(packed, len, observed).unwrap();
        CHECK(std::string(observed.begin(), observed.end()) == input.substr(0, len));
    }
    uint64_t packed = as_2bit("ACTGACTGACTGACTGACTG").unwrap();
    std::vector<uint8_t> observed;
    from_2bit(packed, 10, observed).unwrap();
    from_2bit(packed, 10, observed).unwrap();
    CHECK(std::string(observed.begin(), observed.end()) == "ACTGACTGACACTGACTGAC"); // repeated out of phase
}
static void test_partial_unpack() {
    uint64_t packed = as_2bit("ACGT").unwrap();
    std::vector<uint8_t> unpacked;
    from_2bit(packed, 2, unpacked).unwrap();
    CHECK(std::string(unpacked.begin(), unpacked.end()) == "AC");
    unpacked.clear();
    from_2bit(packed, 3, unpacked).unwrap();
    CHECK(std::string(unpacked.begin(), unpacked.end()) == "ACG");
}
static void test_large_sequence_round_trip() {
    std::mt19937_64 rng(0xB17C0DE);
    for (size_t len = 1; len <= 1000; ++len) {
        std::vector<uint8_t> seq(len);
        for (auto &b : seq) b = "ACGT"[rng() & 3];
        std::vector<uint64_t> ebuf{1, 2, 3}; // encode clears
        encode(seq, ebuf).unwrap();
        CHECK(ebuf.size() == (len + 31) / 32);
        std::vector<uint8_t> unpacked;
        decode(ebuf, len, unpacked).unwrap();
        CHECK(unpacked == seq);
    }
}
static void test_encode_error_keeps_prefix_words() {
    std::vector<uint8_t> seq(200, 'C');
    seq[77] = 'N';
    std::vector<uint64_t> ebuf;
    auto r = encode(seq, ebuf);
    CHECK(r.is_err() && r.unwrap_err() == NucleotideError::invalid_base('N') && r.unwrap_err().index == 77);
    CHECK(ebuf.size() == 2 && ebuf[0] == 0x5555555555555555ull && ebuf[1] == 0x5555555555555555ull);
    bool panicked = false;
    try { encode(std::vector<uint8_t>{}, ebuf); } catch (const std::logic_error &) { panicked = true; }
    CHECK(panicked);
    std::vector<uint8_t> d;
    auto r2 = decode(std::vector<uint64_t>{0}, 33, d);
    CHECK(r2.is_err() && r2.unwrap_err() == NucleotideError::invalid_length(33) && d.empty());
}
static void test_hdist() {
    CHECK(hdist_scalar(0, 0, 33).is_err());
    CHECK(hdist_scalar(0, 0, 0).unwrap() == 0);
    CHECK(hdist_scalar(0, 0, 32).is_ok());
    CHECK(hdist_scalar(0xFFFFFFFFull, 0xFFFFFFFFull, 16).unwrap() == 0);
    CHECK(hdist_scalar(0b0001, 0b0010, 2).unwrap() == 1);
    CHECK(hdist_scalar(0b0001, 0b0011, 2).unwrap() == 1);
    CHECK(hdist_scalar(0b0010, 0b0011, 2).unwrap() == 1);
    struct { const char *a, *b; uint32_t d; } cases[] = {{"AAAA", "AAAA", 0}, {"AAAA", "AAAT", 1}, {"AAAA", "AATT", 2},
                                                       {"AAAA", "ATTT", 3}, {"AAAA", "TTTT", 4}, {"ACTGACTG", "TGCATGCA", 8}};
    for (auto &c : cases) CHECK(hdist_scalar(as_2bit(c.a).unwrap(), as_2bit(c.b).unwrap(), std::string(c.a).size()).unwrap() == c.d);
    std::vector<uint64_t> b1(1, 0), b2(1, 0);
    CHECK(hdist(b1, b2, 64).is_err());
    for (size_t len = 1; len <= 256; ++len) {
        auto e1 = encode_alloc(std::vector<uint8_t>(len, 'A')).unwrap();
        auto e2 = encode_alloc(std::vector<uint8_t>(len, 'T')).unwrap();
        CHECK(hdist(e1, e2, len).unwrap() == len);
    }
}

// src/utils/functions/split.rs:108-224, the reference's own cases, for both modes
static void test_split_packed() {
    auto dec = [](const std::vector<uint64_t> &w, size_t n) { std::vector<uint8_t> d; decode(w, n, d).unwrap(); return std::string(d.begin(), d.end()); };
    for (bool canonical : {false, true}) {
        std::vector<uint64_t> l{9}, r{9};
        auto e = encode_alloc("ACTGACTG").unwrap();
        split_packed(e, 8, 4, l, r, canonical).unwrap();
        CHECK(l.size() == 1 && r.size() == 1 && dec(l, 4) == "ACTG" && dec(r, 4) == "ACTG");
        e = encode_alloc("ACTG").unwrap();
        split_packed(e, 4, 0, l, r, canonical).unwrap();
        CHECK(l.empty() && r.size() == 1 && dec(r, 4) == "ACTG");
        split_packed(e, 4, 4, l, r, canonical).unwrap();
        CHECK(l.size() == 1 && r.empty() && dec(l, 4) == "ACTG");
        auto bad = split_packed(e, 4, 5, l, r, canonical);
        CHECK(bad.is_err() && bad.unwrap_err().kind == NucleotideError::IndexOutOfBounds && bad.unwrap_err().oob_index == 5 &&
              bad.unwrap_err().length == 4 && l.size() == 1); // buffers untouched on error
        e = encode_alloc("ACTGACTGAC").unwrap();
        split_packed(e, 10, 7, l, r, canonical).unwrap();
        CHECK(l.size() == 1 && r.size() == 1 && dec(l, 7) == "ACTGACT" && dec(r, 3) == "GAC");
        const std::string s40 = "ACTGACTGACTGACTGACTGACTGACTGACTGACTGACTG";
        e = encode_alloc(s40).unwrap();
        split_packed(e, 40, 32, l, r, canonical).unwrap();
        CHECK(l.size() == (canonical ? 1u : 2u) && r.size() == 1 && dec(l, 32) == s40.substr(0, 32) && dec(r, 8) == s40.substr(32));
    }
}

static void test_batch_equals_loop() {
    std::mt19937_64 rng(7);
    std::vector<uint8_t> seq;
    std::vector<uint64_t> offsets{0};
    for (int i = 0; i < 500; ++i) {
        size_t len = 1 + rng() % 200;
        for (size_t j = 0; j < len; ++j) seq.push_back("ACGTacgt"[rng() & 7]);
        offsets.push_back(seq.size());
    }
    std::vector<uint64_t> word_offsets;
    auto words = default_context().encode_batch(seq, offsets, word_offsets).unwrap();
    std::vector<uint64_t> expect;
    for (size_t i = 0; i + 1 < offsets.size(); ++i) { // the reference's idiom: one encode() per sequence
        CHECK(word_offsets[i] == expect.size());
        auto w = encode_alloc(Bytes(seq.data() + offsets[i], offsets[i + 1] - offsets[i])).unwrap();
        expect.insert(expect.end(), w.begin(), w.end());
    }
    CHECK(words == expect && word_offsets.back() == expect.size());
    auto back = default_context().decode_batch(words, word_offsets, offsets).unwrap();
    CHECK(back.size() == seq.size());
    for (size_t i = 0; i < seq.size(); ++i) CHECK(back[i] == (seq[i] & 0xDF));
}

static void test_packed_sequence() { // src/sequence.rs:266-338, src/utils/analysis.rs:41-84
    auto seq = PackedSequence::new_("ACGT").unwrap();
    CHECK(seq.len() == 4);
    auto v = seq.to_vec().unwrap();
    CHECK(std::string(v.begin(), v.end()) == "ACGT");
    CHECK(seq.get(0).unwrap() == 'A' && seq.get(1).unwrap() == 'C' && seq.get(2).unwrap() == 'G' && seq.get(3).unwrap() == 'T');
    auto oob = seq.get(4);
    CHECK(oob.is_err() && oob.unwrap_err().kind == NucleotideError::IndexOutOfBounds && oob.unwrap_err().oob_index == 4 && oob.unwrap_err().length == 4);
    auto s8 = PackedSequence::new_("ACGTACGT").unwrap();
    auto sl = s8.slice(1, 5).unwrap();
    CHECK(std::string(sl.begin(), sl.end()) == "CGTA");
    CHECK(s8.slice(2, 2).unwrap().empty());
    auto bad = seq.slice(3, 2);
    CHECK(bad.is_err() && bad.unwrap_err().kind == NucleotideError::InvalidRange && bad.unwrap_err().start == 3 && bad.unwrap_err().end == 2 && bad.unwrap_err().length == 4);
    CHECK(PackedSequence::new_("ACGT").unwrap() == seq && PackedSequence::new_("TGCA").unwrap() != seq);
    CHECK(PackedSequence::new_("ACGN").is_err());
    auto empty = PackedSequence::new_("").unwrap();
    CHECK(empty.is_empty() && empty.gc_content() == 0.0 && empty.base_counts() == (std::array<size_t, 4>{0, 0, 0, 0}));
    CHECK(PackedSequence::new_("ACGTA").unwrap().gc_content() == 40.0);
    CHECK(PackedSequence::new_("AACG").unwrap().base_counts() == (std::array<size_t, 4>{2, 1, 1, 0}));
}

int main() {
#define RUN(t) do { t(); std::printf("ok %s\n", #t); } while (0)
    RUN(test_as_2bit_valid_sequence);
    RUN(test_as_2bit_longer_sequence);
    RUN(test_as_2bit_alignments);
    RUN(test_as_2bit_lowercase);
    RUN(test_as_2bit_invalid_base);
    RUN(test_as_2bit_sequence_too_long);
    RUN(test_from_2bit_valid_sequence);
    RUN(test_example_case_from_2bit);
    RUN(test_various_lengths_and_append);
    RUN(test_partial_unpack);
    RUN(test_large_sequence_round_trip);
    RUN(test_encode_error_keeps_prefix_words);
    RUN(test_hdist);
    RUN(test_split_packed);
    RUN(test_batch_equals_loop);
    RUN(test_packed_sequence);
    std::printf("ALL OK\n");
    return 0;
}
