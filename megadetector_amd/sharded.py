"""
One-process-per-GPU helpers (launched by torch.distributed.run or by run_detector_batch.run_sharded).

The MDv5 image queue shards embarrassingly: every rank owns one GPU and
one slice of the image list; there is NO data-path collective.  torch.distributed is used only for
rendezvous-style bookkeeping: a barrier around timed regions, MAX of the elapsed time over ranks,
and an object gather of the per-image result dicts onto rank 0 (host memory, not xGMI).
Backend: "nccl" (= RCCL on ROCm) on GPUs, "gloo" in the CPU tests.
"""

import os


def rank_info():
    """(rank, local_rank, world_size) from the torchrun environment (1 process = 1 GPU)."""
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')),
            int(os.environ.get('WORLD_SIZE', '1')))


def my_shard(items, rank, world):
    """balanced `i % world` split: the slice of `items` rank `rank` is responsible for"""
    return list(items[rank::world])


def max_over_ranks(value, dist=None, device='cpu'):
    """MAX of a Python float over all ranks (the step time the slowest GPU needed)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_results(results, dist=None, expected_files=None):
    """
    Collects every rank's list of per-image result dicts on rank 0 and merges them with the
    duplicate / omission checks of the reference's manual merge (notebooks/manage_local_batch.py:
    930-964).  Returns the merged list on rank 0 and None elsewhere.
    """
    from .run_detector_batch import merge_shard_results
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return merge_shard_results([results], expected_files)
    world = dist.get_world_size()
    rank = dist.get_rank()
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(results, gathered, dst=0)
    if rank != 0:
        return None
    return merge_shard_results(gathered, expected_files)
