"""PackedSequence -- the reference's owned high-level type (src/sequence.rs:5-262) over
device-encoded data, plus its GCContent / BaseCount traits (src/utils/analysis.rs:3-39).

`data` (u64 words) and `length` have the reference's meaning; `new` encodes and `slice` /
`to_vec` decode through the library (only the words the range touches; on the GPU above the
library's host cutoff), `get` is the reference's shift and mask, the trait methods count on the
packed words on the GPU.  Range / index checks are the reference's.
"""
import numpy as np

from .api import NucleotideError, default_context


class PackedSequence:
    __slots__ = ("data", "length", "_ctx")

    def __init__(self, seq, ctx=None):  # PackedSequence::new, sequence.rs:40-52
        self._ctx = ctx or default_context()
        s = np.frombuffer(bytes(seq), dtype=np.uint8) if not isinstance(seq, np.ndarray) else seq
        if s.size == 0:  # sequence.rs:42-44: skip encoding for empty sequences
            self.data = np.zeros(0, dtype=np.uint64)
        else:
            self.data = self._ctx.encode_array(s)
        self.length = int(s.size)

    new = classmethod(lambda cls, seq, ctx=None: cls(seq, ctx))

    def __len__(self):  # sequence.rs:67-69
        return self.length

    def len(self):
        return self.length

    def is_empty(self):  # sequence.rs:84-86
        return self.length == 0

    def get(self, index):  # sequence.rs:116-135
        if index < 0 or index >= self.length:
            raise NucleotideError("IndexOutOfBounds", index=index, length=self.length)
        # the reference's shift + mask + match (sequence.rs:121-134): no call into the library at all
        return b"ACGT"[(int(self.data[index // 32]) >> (2 * (index % 32))) & 3]

    def slice(self, start, end):  # sequence.rs:198-212
        if start < 0 or start > end or end > self.length:
            raise NucleotideError("InvalidRange", start=start, end=end, length=self.length)
        if start == end:
            return b""
        w0, w1 = start // 32, (end + 31) // 32
        n = min(self.length, w1 * 32) - w0 * 32
        chunk = self._ctx.decode_array(self.data[w0:w1], n)  # decode only the words the range touches
        return chunk[start - w0 * 32: end - w0 * 32].tobytes()

    def to_vec(self):  # sequence.rs:260-262
        return self.slice(0, self.length)

    # derive(PartialEq, Eq, Hash), sequence.rs:5
    def __eq__(self, other):
        return isinstance(other, PackedSequence) and self.length == other.length and np.array_equal(self.data, other.data)

    def __hash__(self):
        return hash((self.length, self.data.tobytes()))

    # traits GCContent / BaseCount (analysis.rs:3-39), evaluated on the packed words
    def gc_content(self):
        return self._ctx.gc_content(self.data, self.length)

    def base_counts(self):
        return self._ctx.base_counts(self.data, self.length)
