"""Sample sharding + timing helpers of the multi-GPU bench.

The path shards by independent samples: rank r owns `samples_per_gpu` samples with their own
camera parameters; there is NO data-path collective.  torch.distributed (RCCL on GPUs, gloo in the
CPU tests) is used only to fence the timed region and to take the max elapsed time over ranks.
"""
import os

import torch
import torch.distributed as dist


def world():
    return int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0')), \
        int(os.environ.get('LOCAL_RANK', '0'))


def init(backend, device=None):
    ws, rank, _ = world()
    if ws > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        kw = {'device_id': device} if (device is not None and backend == 'nccl') else {}
        dist.init_process_group(backend, **kw)
    return ws, rank


def shard_seed(rank, base=0):
    """Seed of rank r's synthetic samples: different rigs / augmentations / features per rank."""
    return base + 1000 * rank


def fence(device=None):
    """barrier + device sync on both sides of the timed region (bench contract)."""
    if device is not None and device.type == 'cuda':
        torch.cuda.synchronize(device)
    if dist.is_initialized():
        dist.barrier()
    if device is not None and device.type == 'cuda':
        torch.cuda.synchronize(device)


def max_over_ranks(seconds, device=None):
    if not dist.is_initialized():
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    if not dist.is_initialized():
        return int(value)
    t = torch.tensor([value], dtype=torch.int64, device=device if device is not None else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def whole_job_rate(samples_per_gpu, steps, elapsed_max, world_size):
    """value = samples ALL ranks processed / max-over-ranks time."""
    return samples_per_gpu * world_size * steps / elapsed_max


# ------------------------------------------------------------------ training step: the one collective of the path
def allreduce_gradients(params, bucket_bytes=64 << 20, async_op=False):
    """Average the gradients of `params` over all ranks: the single collective of the training step (SURVEY 8e --
    the data path itself stays collective-free).  Gradients are packed into flat fp32 buckets of <= bucket_bytes and
    each bucket is ONE all-reduce: over xGMI a ring all-reduce is bound by a single ~153 GB/s link whatever the
    message count, so few large messages beat DDP's default 25 MB buckets; the path's own parameters (BEV queries,
    encoder layer, history convs: ~4 MB) fit one bucket.  Returns the list of work handles when async_op=True (call
    `finish_allreduce` before the optimizer step) so the reduction overlaps the rest of the backward pass."""
    grads = [p.grad for p in params if p.grad is not None]
    if not dist.is_initialized() or dist.get_world_size() == 1 or not grads:
        return []
    world = dist.get_world_size()
    buckets, cur, size = [], [], 0
    for g in grads:
        n = g.numel() * g.element_size()
        if cur and size + n > bucket_bytes:
            buckets.append(cur)
            cur, size = [], 0
        cur.append(g)
        size += n
    if cur:
        buckets.append(cur)
    pending = []
    for b in buckets:
        flat = torch.cat([g.reshape(-1) for g in b])
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
        pending.append((work, flat, b, world))
    if async_op:
        return pending
    finish_allreduce(pending)
    return []


def finish_allreduce(pending):
    for work, flat, bucket, world in pending:
        work.wait()
        flat.div_(world)
        off = 0
        for g in bucket:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n
