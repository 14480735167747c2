"""Sample-index partitioning across GPUs (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

The path shards naturally: the RNG stream of a sample is a pure function of (px, py, sampleIndex)
(resources/shaders/raytracing/Random.hlsl:91-99), so every rank renders the FULL frame for its own sample indices into
its own float SUM target and the only exchange is one reduce of W*H*4 floats at the end (33 MB at 1080p) — no
data-path collective inside the render. The reference itself is single-GPU (no counterpart).
"""
import torch
import torch.distributed as dist


def partition_samples(total_samples, world_size, rank):
    """Contiguous block of sample indices for `rank`: returns (first_sample, num_samples).
    Blocks differ by at most one sample; their union is exactly [0, total_samples)."""
    base, rem = divmod(int(total_samples), int(world_size))
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def reduce_sum_to_root(accum, num_samples, group=None, root=0):
    """Sum the per-rank accumulation targets (tensor of W*H*4 floats, sum of samples) onto `root` and return
    (accum, total_sample_count). One reduce + one tiny all-reduce of the counts. The collective runs on torch's
    stream: call ctx.synchronize() first so every render queued on the ctx stream has landed in `accum`."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return accum, int(num_samples)
    cnt = torch.tensor([int(num_samples)], dtype=torch.int64, device=accum.device)
    dist.reduce(accum, dst=root, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=group)
    return accum, int(cnt.item())   # on non-root ranks `accum` is unspecified after the reduce


def gather_ordered_sum_to_root(accum, num_samples, group=None, root=0):
    """Bit-reproducible variant: gather every rank's target on root and add them in rank order, so the result does
    not depend on RCCL's internal reduction order (fp32 addition is not associative). ctx.synchronize() must
    precede it, as for reduce_sum_to_root."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return accum, int(num_samples)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    cnt = torch.tensor([int(num_samples)], dtype=torch.int64, device=accum.device)
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=group)
    if rank == root:
        parts = [torch.empty_like(accum) for _ in range(world)]
        dist.gather(accum, gather_list=parts, dst=root, group=group)
        total = parts[0].clone()
        for p in parts[1:]:
            total += p
        accum.copy_(total)
    else:
        dist.gather(accum, gather_list=None, dst=root, group=group)
    return accum, int(cnt.item())


def bind_torch_target(ctx, width, height, device):
    """Allocate the accumulation target as a torch tensor (so RCCL can reduce it in place) and bind it to the ctx.
    torch.empty, not zeros: pt_accum_bind zeroes the target on the ctx stream, and a fill kernel on torch's current
    stream would not be ordered against it."""
    t = torch.empty(height * width * 4, dtype=torch.float32, device=device)
    ctx.accum_bind(t.data_ptr(), width, height)
    return t


# ---------------------------------------------------------------------------------------------------------------------
# GPU-count-independent result: fp32 addition is not associative, so "the 8-GPU image equals the 1-GPU
# image" only holds bit for bit if the order of additions does not depend on the number of GPUs. The sample range is
# therefore always cut into VIRTUAL_PARTITIONS fixed blocks; a rank renders each of ITS blocks into a target of its own
# (samples added in index order), and the root adds the block sums in block order. 1, 2, 4 or 8 ranks give the same bits.
VIRTUAL_PARTITIONS = 8


def rank_partitions(world_size, rank):
    """Virtual partitions owned by `rank`: a contiguous run, so rank order == partition order."""
    if VIRTUAL_PARTITIONS % world_size != 0:
        raise ValueError(f"world size {world_size} must divide {VIRTUAL_PARTITIONS}")
    per = VIRTUAL_PARTITIONS // world_size
    return list(range(rank * per, (rank + 1) * per))


def render_virtual_partitions(ctx, consts, width, height, total_samples, world_size, rank, device, first_sample=0):
    """Render this rank's virtual partitions; returns a (len(rank_partitions), H*W*4) float32 tensor of block sums."""
    mine = rank_partitions(world_size, rank)
    out = torch.empty((len(mine), height * width * 4), dtype=torch.float32, device=device)   # accum_bind zeroes each block
    for k, v in enumerate(mine):
        first, n = partition_samples(total_samples, VIRTUAL_PARTITIONS, v)
        ctx.accum_bind(out[k].data_ptr(), width, height)      # zeroes the block target
        if n:
            ctx.render(consts, width, height, first_sample + first, n)
    ctx.synchronize()
    ctx.accum_bind(0, 0, 0)
    return out


def combine_virtual_partitions(block_sums, group=None, root=0):
    """Gather every rank's block sums on `root` and add them in partition order. Returns the (H*W*4) sum on root,
    None elsewhere. With one process the blocks are simply added in order."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        parts = [block_sums]
    else:
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        if rank == root:
            parts = [torch.empty_like(block_sums) for _ in range(world)]
            dist.gather(block_sums, gather_list=parts, dst=root, group=group)
        else:
            dist.gather(block_sums, gather_list=None, dst=root, group=group)
            return None
    return sum_blocks_in_order(parts)


def sum_blocks_in_order(parts):
    """parts: per-rank (n_blocks, N) tensors in rank order (== partition order). One fixed chain of additions."""
    total = None
    for p in parts:
        for k in range(p.shape[0]):
            total = p[k].clone() if total is None else total.add_(p[k])
    return total
