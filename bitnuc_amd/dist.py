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


def encode_allgather_overlapped(encode_chunk, shard_words, n_chunks, like, group=None):
    """BASELINE config 4 end to end with chunked overlap (SURVEY 8e iii): the shard is encoded in
    `n_chunks` pieces; as soon as piece c is enqueued its all-gather is issued asynchronously, so
    the fabric moves piece c while the GPU encodes piece c+1 (with backend "nccl" torch runs the
    collective on its own stream, ordered after the encode by an event).

    encode_chunk(w0, w1) -> 1-D int64 tensor holding words [w0, w1) of this rank's shard (on the
    GPU: a view of the shard's word buffer after Context.encode_dev of bases [32*w0, 32*w1) on the
    current stream).  Every rank must have the same `shard_words`.  `like` gives dtype/device.
    Returns the concatenation [world * shard_words], bit-identical to allgather_packed(words)."""
    world = dist.get_world_size(group)
    bounds = [shard_words * c // n_chunks for c in range(n_chunks + 1)]
    pieces, works = [], []
    for c in range(n_chunks):
        w0, w1 = bounds[c], bounds[c + 1]
        if w1 == w0:
            continue
        local = encode_chunk(w0, w1)
        tmp = torch.empty(world * (w1 - w0), dtype=like.dtype, device=like.device)  # [world][w1-w0]
        works.append(dist.all_gather_into_tensor(tmp, local.contiguous(), group=group, async_op=True))
        pieces.append((w0, w1, tmp))
    out = torch.empty(world, shard_words, dtype=like.dtype, device=like.device)
    for work, (w0, w1, tmp) in zip(works, pieces):
        work.wait()  # nccl: orders the current stream after the collective; gloo: blocks
        out[:, w0:w1].copy_(tmp.view(world, w1 - w0))
    return out.reshape(world * shard_words)
