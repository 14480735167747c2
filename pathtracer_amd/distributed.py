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
    (accum, total_sample_count). One reduce + one tiny all-reduce of the counts."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return accum, int(num_samples)
    cnt = torch.tensor([int(num_samples)], dtype=torch.int64, device=accum.device)
    dist.reduce(accum, dst=root, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=group)
    return accum, int(cnt.item())   # on non-root ranks `accum` is unspecified after the reduce


def gather_ordered_sum_to_root(accum, num_samples, group=None, root=0):
    """Bit-reproducible variant: gather every rank's target on root and add them in rank order, so the result does
    not depend on RCCL's internal reduction order (fp32 addition is not associative)."""
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
    """Allocate the accumulation target as a torch tensor (so RCCL can reduce it in place) and bind it to the ctx."""
    t = torch.zeros(height * width * 4, dtype=torch.float32, device=device)
    ctx.accum_bind(t.data_ptr(), width, height)
    return t
