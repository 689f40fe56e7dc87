"""Multi-GPU driver: slices of one texture array are independent contiguous block ranges
(SliceDesc.file_ofs/file_size, basis.rs:531-552), so they shard across ranks with no exchange during
the transcode; one all-gather reassembles the array on every rank (RCCL over xGMI when the process
group is "nccl").  One process per GPU, torch.distributed only as plumbing.

The per-shard work is injected (`transcode_fn`) so the partition/gather logic is testable on CPU with
the gloo backend; the product default (`gpu_transcode_fn`) is the HIP path and needs a device.
"""
import torch
import torch.distributed as dist


def partition(n_items, world_size, rank):
    """contiguous range of slices owned by `rank`: [rank*n/P, (rank+1)*n/P)"""
    return (n_items * rank) // world_size, (n_items * (rank + 1)) // world_size


def gpu_transcode_fn(ctx, fmt):
    """per-shard function running the HIP kernels on cuda tensors [n_blocks,16] u8 -> [n_blocks,B] u8"""
    from . import _lib

    def fn(d_in):
        n = d_in.shape[0]
        out = torch.empty((n, _lib.BLOCK_BYTES[int(fmt)]), dtype=torch.uint8, device=d_in.device)
        status = torch.empty(1, dtype=torch.int64, device=d_in.device)
        ctx.status_word_reset(status)
        ctx.transcode_device(int(fmt), d_in, n, out, d_status=status)
        # deliberate sync point: the error contract is "first failing block aborts the call"
        ctx.status_word_check(int(status.item()) & 0xFFFFFFFFFFFFFFFF)
        return out

    return fn


def transcode_array_sharded(slices, transcode_fn, group=None, gather=True):
    """slices: tensor [n_slices, blocks_per_slice, 16] u8, identical on every rank (or at least the
    rank's own range valid).  Each rank transcodes its contiguous range; with gather=True every rank
    returns the whole [n_slices, blocks_per_slice, B] result."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n_slices, bps = slices.shape[0], slices.shape[1]
    lo, hi = partition(n_slices, world, rank)
    local = transcode_fn(slices[lo:hi].reshape(-1, 16)).reshape(hi - lo, bps, -1)
    if world == 1 or not gather:
        return local
    bb = local.shape[-1]
    if n_slices % world == 0:
        full = torch.empty((n_slices, bps, bb), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(full, local.contiguous(), group=group)
        return full
    # ragged: pad every shard to the largest one
    biggest = max(partition(n_slices, world, r)[1] - partition(n_slices, world, r)[0] for r in range(world))
    padded = torch.zeros((biggest, bps, bb), dtype=local.dtype, device=local.device)
    padded[: hi - lo] = local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    out = []
    for r in range(world):
        a, b = partition(n_slices, world, r)
        out.append(parts[r][: b - a])
    return torch.cat(out, dim=0)
