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
