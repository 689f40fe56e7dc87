"""Multi-GPU driver: slices of one texture array are independent contiguous block ranges
(SliceDesc.file_ofs/file_size, basis.rs:531-552), so they shard across ranks with no exchange during
the transcode; one all-gather reassembles the array on every rank (RCCL over xGMI when the process
group is "nccl").  One process per GPU, torch.distributed only as plumbing.

The per-shard work is injected (`transcode_fn`) so the partition/gather logic is testable on CPU with
the gloo backend; the product default (`gpu_transcode_fn`) is the HIP path and needs a device.

Error contract (uastc.rs:157-165, "first failing block aborts"): every rank learns the LOWEST failing
block of the whole array before the gather (one all_reduce(MIN) of the status words), so all ranks raise
the same BasisuError and none enters the all-gather alone.
"""
import inspect

import torch
import torch.distributed as dist

_CLEAR = (1 << 63) - 1  # "no failing block" as an int64 that loses every MIN against a real status word
_RANK_FAILED = -1       # "this rank's transcode raised": wins every MIN (real status words are block << 8 | status >= 1)


def partition(n_items, world_size, rank):
    """contiguous range of slices owned by `rank`: [rank*n/P, (rank+1)*n/P)"""
    return (n_items * rank) // world_size, (n_items * (rank + 1)) // world_size


def gpu_transcode_fn(ctx, fmt):
    """per-shard function running the HIP kernels: (cuda tensor [n_blocks,16] u8, out [n_blocks,B] u8, first block
    index of the shard) -> status word (block_index << 8 | status, or _CLEAR); never raises for a block error"""
    from . import _lib

    def fn(d_in, out, base):
        n = d_in.shape[0]
        if n == 0:
            return _CLEAR
        # the rank's inputs were produced on torch's stream; the library's own streams do not wait for it
        torch.cuda.current_stream(d_in.device).synchronize()
        # bu_uastc_transcode_device_sync: one exclusive launch with tile tickets on long walks (0.77 of the roofline for a 2^25-block range), status word
        # in page-locked memory -- the deliberate sync point of the error contract ("first failing block aborts the call"), and the same code
        # bu_array_transcode_sharded runs per device
        word = ctx.transcode_device_sync(int(fmt), d_in, n, out, block_index_base=int(base))
        return _CLEAR if word == _lib.STATUS_WORD_CLEAR else word

    fn.block_bytes = _lib.BLOCK_BYTES[int(fmt)]
    fn.raise_for = lambda word: ctx.status_word_check(word)
    return fn


def transcode_array_sharded(slices, transcode_fn, group=None, gather=True, block_bytes=None):
    """slices: tensor [n_slices, blocks_per_slice, 16] u8, identical on every rank (or at least the
    rank's own range valid).  Each rank transcodes its contiguous range straight into its slot of the full
    [n_slices, blocks_per_slice, B] result; with gather=True one in-place all-gather fills the other slots on
    every rank.  `transcode_fn(d_in, out, base)` writes `out` and returns a status word (see gpu_transcode_fn);
    a function of one argument returning the output tensor is accepted too (tests)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n_slices, bps = slices.shape[0], slices.shape[1]
    bb = block_bytes or getattr(transcode_fn, "block_bytes", 16)
    lo, hi = partition(n_slices, world, rank)
    even = n_slices % world == 0
    # ragged arrays gather fixed-size slots of the largest shard; an even split gathers in place with no padding
    per = n_slices // world if even else max(partition(n_slices, world, r)[1] - partition(n_slices, world, r)[0] for r in range(world))
    want_full = gather and world > 1
    full = torch.empty((world * per if want_full else hi - lo, bps, bb), dtype=torch.uint8, device=slices.device)
    mine = full[rank * per: rank * per + (hi - lo)] if want_full else full
    d_in = slices[lo:hi].reshape(-1, 16)
    word = _CLEAR
    local_exc = None
    if hi > lo:
        # A rank whose transcode RAISES (HIP error, out of memory, a bad transcode_fn) must still take part in the all_reduce:
        # leaving before it would park every other rank in the collective until the RCCL time-out.  The exception travels as
        # a sentinel status word below every real one, so all ranks stop together.
        try:
            if len(inspect.signature(transcode_fn).parameters) >= 3:
                word = transcode_fn(d_in, mine.view(-1, bb), lo * bps)
            else:  # plain one-argument function returning the result (CPU tests)
                mine.view(-1, bb).copy_(transcode_fn(d_in).reshape(-1, bb))
        except Exception as e:  # noqa: BLE001 -- re-raised below, after the ranks have met
            local_exc = e
            word = _RANK_FAILED
    if world > 1:
        w = torch.tensor([word], dtype=torch.int64, device=slices.device)
        dist.all_reduce(w, op=dist.ReduceOp.MIN, group=group)
        word = int(w.item())
    if local_exc is not None:
        raise local_exc
    if word == _RANK_FAILED:
        raise RuntimeError("the transcode raised on another rank; no rank enters the all-gather")
    if word != _CLEAR:
        raiser = getattr(transcode_fn, "raise_for", None)
        if raiser:
            raiser(word)
        raise RuntimeError("block %d failed with status %d" % (word >> 8, word & 0xFF))
    if not want_full:
        return mine
    # in place: this rank's shard already sits in its slot of `full` (ncclAllGather with send = recv + rank*count)
    dist.all_gather_into_tensor(full.view(-1), mine_slot(full, rank, per).reshape(-1), group=group)
    if even:
        return full
    out = [full[r * per: r * per + (partition(n_slices, world, r)[1] - partition(n_slices, world, r)[0])] for r in range(world)]
    return torch.cat(out, dim=0)


def mine_slot(full, rank, per):
    """rank's fixed-size slot of the gather buffer (a view: the all-gather runs in place)"""
    return full[rank * per: (rank + 1) * per]
