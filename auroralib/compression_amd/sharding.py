"""Multi-GPU sharding helpers: streams are independent, so a batch shards by stream across ranks with no
data-path collective.  The only distributed operations are a barrier and the MAX-reduce of the measured step time."""


def shard_seed(config, rank, streams_per_rank):
    """Disjoint synthetic stream indices per rank: seed = 0xA17A0000 + 1000*config + rank*streams_per_rank."""
    return 0xA17A0000 + 1000 * config + rank * streams_per_rank


def reduce_step_time(seconds, dist=None, device=None):
    """MAX over ranks of the time of the timed region."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def whole_job_value(decompressed_bytes_per_rank, world, steps, seconds):
    """decompressed GiB/s of the whole job = bytes decoded by all ranks in all steps / max time."""
    return decompressed_bytes_per_rank * world * steps / seconds / 2**30
