"""One-process-per-GPU sharding of the codec (torch.distributed; backend "nccl" is RCCL
over xGMI on ROCm, "gloo" on CPU for tests).

u64 words never share state (the reference's chunk loop has no carry,
src/utils/packing/avx.rs:138-145), so any partition at a multiple of 32 bases is
exact: rank r encodes its shard with no communication.  The only exchange the path
has is the optional final concatenation of the packed buffer (BASELINE config 4):
one all-gather of the per-rank u64 words.
"""
import torch
import torch.distributed as dist


def shard_range(n_bases, rank, world):
    """Contiguous, 32-base-aligned shard [start, stop) of n_bases for `rank` of `world`.
    Shards differ by at most one word; the last shard takes the ragged tail."""
    n_words = (n_bases + 31) // 32
    base, extra = divmod(n_words, world)
    w0 = rank * base + min(rank, extra)
    w1 = w0 + base + (1 if rank < extra else 0)
    return min(w0 * 32, n_bases), min(w1 * 32, n_bases)


def shard_word_counts(n_bases, world):
    return [(b - a + 31) // 32 for a, b in (shard_range(n_bases, r, world) for r in range(world))]


def scan_shard_range(n_bases, k, rank, world):
    """Shard of the sliding k-mer scan / window packing (BASELINE config 5, SURVEY 8e: "scan needs a 30-base halo per shard").
    The n_bases - k + 1 windows are split into `world` contiguous runs (32-window aligned like the codec's shards); rank r
    reads bases [first, first + count + k - 1) -- its windows plus a (k-1)-base halo that overlaps the next shard -- and
    produces windows [first, first + count).  Concatenating the ranks' outputs in rank order is the single-GPU result; there is
    no exchange.  Returns (first_window, n_windows, n_bases_to_read); n_windows is 0 for a rank left without work."""
    n_win = n_bases - k + 1 if n_bases >= k and k > 0 else 0
    a, b = shard_range(n_win, rank, world)  # window indices
    count = b - a
    return a, count, (count + k - 1 if count else 0)


def batch_word_prefix(offsets):
    """W[i] = packed words before sequence i of a ragged batch (sequence i = bases [offsets[i], offsets[i+1])): every sequence
    pads its OWN last word (src/utils/packing/avx.rs:147-148), so W is also the batch's global word_offsets table."""
    import numpy as np
    off = np.asarray(offsets, dtype=np.uint64)
    lens = off[1:] - off[:-1]
    if off.size > 1 and bool((off[1:] < off[:-1]).any()):
        raise ValueError("offsets must not decrease")
    W = np.zeros(off.size, dtype=np.uint64)
    np.cumsum((lens + np.uint64(31)) // np.uint64(32), out=W[1:])
    return W


def batch_shard_ranges(offsets, world):
    """The split BASELINE's north_star names -- "batches of independent sequences shard trivially across the 8 GPUs" (SURVEY 8e):
    rank r of `world` gets the run of WHOLE sequences [seq_first[r], seq_first[r+1]), balanced by word count: seq_first[r] = the
    first i with W[i] >= floor(r * W[count] / world).  A sequence longer than a fair share stays whole and the ranks it covers get
    empty runs.  Returns (seq_first, word_first), both numpy uint64[world + 1]; word_first[r] = where rank r's words start in the
    concatenation.  The same rule as bitnuc_batch_shard_ranges (include/bitnuc_hip.h); replaces nothing in the reference, whose
    caller loops over sequences one encode() at a time (src/utils/mod.rs:22-25)."""
    import numpy as np
    W = batch_word_prefix(offsets)
    total = int(W[-1])
    targets = np.array([(r * total) // world for r in range(world)], dtype=np.uint64)
    seq_first = np.empty(world + 1, dtype=np.uint64)
    seq_first[:world] = np.searchsorted(W, targets, side="left")
    seq_first[world] = W.size - 1
    return seq_first, W[seq_first.astype(np.int64)]


def batch_shard(offsets, rank, world):
    """What rank `rank` needs of a ragged batch: (seq0, seq1, base0, base1, local_offsets, word0, n_words) -- its run of sequences, the
    byte range of the batch it reads, that run's offsets rebased to 0 and its slot [word0, word0 + n_words) of the concatenation."""
    import numpy as np
    off = np.asarray(offsets, dtype=np.uint64)
    seq_first, word_first = batch_shard_ranges(off, world)
    s0, s1 = int(seq_first[rank]), int(seq_first[rank + 1])
    local = off[s0:s1 + 1] - off[s0]
    return s0, s1, int(off[s0]), int(off[s1]), local, int(word_first[rank]), int(word_first[rank + 1] - word_first[rank])


def allgatherv_packed_(out, first, group=None):
    """In-place all-gather of UNEQUAL word counts: this rank's words already sit in out[first[rank]:first[rank+1]]; every peer's
    slice arrives in place by one batch of point-to-point operations (grouped ncclSend / ncclRecv under backend "nccl": all
    world-1 xGMI links of a GPU at once, no padding to the largest count, no staging tensor).  `first` (world+1 prefix of the
    counts) must be the same on every rank; ranks with an empty slice send nothing.  C / Rust hosts: bitnuc_allgatherv_words_dev."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    first = [int(x) for x in first]
    assert len(first) == world + 1 and out.numel() >= first[-1] and out.is_contiguous()
    peers = [dist.get_global_rank(group, s) if group is not None else s for s in range(world)]
    ops = []
    for s in range(world):
        if s == rank:
            continue
        if first[rank + 1] > first[rank]:
            ops.append(dist.P2POp(dist.isend, out[first[rank]:first[rank + 1]], peers[s], group))
        if first[s + 1] > first[s]:
            ops.append(dist.P2POp(dist.irecv, out[first[s]:first[s + 1]], peers[s], group))
    if ops:
        for work in dist.batch_isend_irecv(ops):
            work.wait()
    return out


def encode_batch_sharded(encode_batch_fn, offsets, like, group=None, out=None):
    """A ragged batch encoded by `world` ranks, whole sequences each (batch_shard_ranges), and concatenated on every rank.
    encode_batch_fn(base0, base1, local_offsets, dst) encodes this rank's run -- the batch's bytes [base0, base1), sequences at
    `local_offsets` (rebased to 0) -- into `dst`, a view of the output at this rank's slot (on the GPU: a bitnuc_batch_plan built
    from local_offsets + Context.encode_batch_plan_dev; in the CPU tests: the oracle's per-sequence loop), and returns the run's
    word_offsets table (len(local_offsets) entries, starting at 0).  No data-path collective before the final gather.
    Returns (words[total], word_offsets[count+1]) -- bit-identical to one encode_batch of the whole batch on one GPU; the global
    table is each rank's own table + word_first[rank], gathered with the same exchange (and equal to batch_word_prefix)."""
    import numpy as np
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    seq_first, word_first = batch_shard_ranges(offsets, world)
    s0, s1, b0, b1, local, w0, nw = batch_shard(offsets, rank, world)
    total = int(word_first[-1])
    if out is None:
        out = torch.empty(total, dtype=like.dtype, device=like.device)
    assert out.numel() == total and out.is_contiguous()
    local_wo = encode_batch_fn(b0, b1, local, out[w0:w0 + nw])
    allgatherv_packed_(out, word_first, group)
    # the global table: rank r owns entries seq_first[r] .. seq_first[r+1]-1 (= its table without the closing entry, + its word prefix)
    count = len(offsets) - 1
    table = torch.zeros(count + 1, dtype=torch.int64, device=like.device)
    mine = torch.as_tensor(np.asarray(local_wo, dtype=np.uint64)[:s1 - s0].astype(np.int64) + w0, dtype=torch.int64, device=like.device)
    table[s0:s1].copy_(mine)
    if count:
        allgatherv_packed_(table, seq_first, group)
    table[count] = total
    return out, table


def allgather_packed(local_words, counts=None, group=None):
    """All-gather per-rank packed words (1-D int64/uint64-as-int64 tensors) into the
    concatenation every rank holds.  Equal counts are one all_gather_into_tensor straight
    into the output; ragged counts pad to the maximum and trim."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if counts is None:
        counts = [local_words.numel()] * world
    assert counts[rank] == local_words.numel()
    if len(set(counts)) == 1:
        out = torch.empty(world * counts[0], dtype=local_words.dtype, device=local_words.device)
        dist.all_gather_into_tensor(out, local_words.contiguous(), group=group)
        return out
    m = max(counts)
    padded = torch.zeros(m, dtype=local_words.dtype, device=local_words.device)
    padded[: counts[rank]].copy_(local_words)
    gathered = torch.empty(world * m, dtype=local_words.dtype, device=local_words.device)
    dist.all_gather_into_tensor(gathered, padded, group=group)
    return torch.cat([gathered[r * m: r * m + counts[r]] for r in range(world)])


def encode_sharded(encode_fn, seq_shard, n_total, group=None):
    """Encode this rank's shard with `encode_fn(seq_shard) -> 1-D int64 words` (on GPU:
    Context.encode_dev into a torch tensor) and all-gather the packed words.
    `seq_shard` must be the bytes of shard_range(n_total, rank, world)."""
    world = dist.get_world_size(group)
    counts = shard_word_counts(n_total, world)
    words = encode_fn(seq_shard)
    return allgather_packed(words, counts, group)


def encode_allgather_overlapped(encode_chunk, shard_words, n_chunks, like, group=None, out=None):
    """BASELINE config 4 end to end with chunked overlap (SURVEY 8e iii), IN PLACE: the shard is encoded in `n_chunks`
    pieces straight into this rank's slot of the output; as soon as piece c is enqueued it is exchanged with every peer
    by one batch of point-to-point operations (grouped ncclSend / ncclRecv under backend "nccl": every peer's piece lands
    directly in out[peer * shard_words + w0 ...], no staging tensor, no strided copy -- and on MI355X's full mesh of
    point-to-point xGMI links all world-1 links of a GPU carry data at once), while the GPU encodes piece c+1.

    encode_chunk(w0, w1, dst): write words [w0, w1) of this rank's shard into `dst` (a contiguous 1-D view of the output;
    on the GPU: Context.encode_dev of bases [32*w0, 32*w1) on the current stream).  A two-argument encode_chunk(w0, w1)
    that returns the words is still accepted (one device copy per piece).  Every rank must have the same `shard_words`.
    `like` gives dtype / device; `out` (world * shard_words) may be passed to reuse a buffer.
    Returns the concatenation [world * shard_words], bit-identical to allgather_packed(words).
    C / Rust hosts have the same thing as bitnuc_encode_sharded_allgather_overlapped_dev (include/bitnuc_hip.h)."""
    import inspect
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if out is None:
        out = torch.empty(world * shard_words, dtype=like.dtype, device=like.device)
    assert out.numel() == world * shard_words and out.is_contiguous()
    grid = out.view(world, shard_words)
    try:
        into = len(inspect.signature(encode_chunk).parameters) >= 3
    except (TypeError, ValueError):
        into = False
    peers = [dist.get_global_rank(group, s) if group is not None else s for s in range(world)]
    works = []
    for c in range(n_chunks):
        w0, w1 = shard_words * c // n_chunks, shard_words * (c + 1) // n_chunks
        if w1 == w0:
            continue
        dst = grid[rank, w0:w1]
        if into:
            encode_chunk(w0, w1, dst)
        else:
            local = encode_chunk(w0, w1)
            if local.data_ptr() != dst.data_ptr():
                dst.copy_(local)
        ops = []
        for s in range(world):
            if s != rank:
                ops.append(dist.P2POp(dist.isend, dst, peers[s], group))
                ops.append(dist.P2POp(dist.irecv, grid[s, w0:w1], peers[s], group))
        if ops:
            works.extend(dist.batch_isend_irecv(ops))
    for work in works:
        work.wait()  # nccl: orders the current stream after the exchange; gloo: blocks
    return out
