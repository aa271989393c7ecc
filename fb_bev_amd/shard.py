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
class GradBuckets:
    """Bucketed gradient averaging launched from autograd hooks (the DDP step of tools/dist_train.sh:10-20 /
    mmdet3d/apis/train.py:229-233, rebuilt for xGMI): the single collective of the training step.

    * every `requires_grad` parameter, in REVERSE registration order (~ the order autograd produces gradients), is
      assigned a slice of a flat, pre-allocated fp32 bucket; `p.grad` IS that slice (a view), so there is no
      pack (`cat`) and no copy-back pass;
    * a post-accumulate-grad hook counts the bucket's parameters; when the last one has its gradient the bucket's
      all-reduce is launched asynchronously on the collective library's own stream, overlapping the rest of backward;
    * collectives of one process group are matched by ISSUE ORDER, so the buckets are launched strictly in index order:
      bucket i goes out from a hook only once buckets 0..i-1 have gone out, otherwise it waits for `finish()`, which
      launches whatever is left in index order (a parameter that received no gradient on this rank contributes its
      zeros).  Every rank therefore issues 0,1,2,... whatever subset of its parameters received gradients -- a block
      unused on one rank delays that rank's launches, it cannot pair bucket 1 with bucket 2; `prepare_ddp`
      additionally gives the buckets their own process group so that their issue order is independent of the
      SyncBN all-reduces interleaved with them during backward;
    * `finish()` then waits and scales by 1/world;
    * bucket size: a ring all-reduce over xGMI is bound by one ~153 GB/s link whatever the message count, so the
      buckets are few and large (default 64 MB; the 270 MB detector = 5 messages) instead of DDP's 25 MB.
    """

    def __init__(self, params, bucket_bytes=64 << 20, group=None, rebuild=True):
        self.group = group
        self.bucket_bytes = bucket_bytes
        self.params = [p for p in params if p.requires_grad]
        self._hooks = []
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._hold = False
        # first step: buckets in reverse registration order (~ the order autograd produces gradients); its finish()
        # re-lays them out in the order the gradients ACTUALLY arrived (rank 0's order, broadcast), because with
        # strictly ordered launches one late parameter in an early bucket holds back every bucket behind it
        self._arrival = [] if rebuild else None
        self._seen = set()                 # parameters whose gradient arrived since the last finish()
        self._layout(list(reversed(self.params)), keep=False)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    def _layout(self, order, keep):
        """Assign every parameter a slice of a flat fp32 bucket (order = bucket fill order); keep=True carries the
        current gradient values over into the new slices."""
        buckets, cur, size = [], [], 0
        for p in order:
            n = p.numel() * 4
            if cur and size + n > self.bucket_bytes:
                buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += n
        if cur:
            buckets.append(cur)
        self.buckets = buckets
        self._flat, self._of, self._ready, self._work = [], {}, [], []
        self._next = 0                                      # buckets [0, _next) have been launched this step
        for bi, ps in enumerate(self.buckets):
            dev = ps[0].device
            if any(p.device != dev or p.dtype != torch.float32 for p in ps):
                raise ValueError('GradBuckets: a bucket must hold fp32 parameters of one device')
            flat = torch.zeros(sum(p.numel() for p in ps), dtype=torch.float32, device=dev)
            off = 0
            for p in ps:
                view = flat[off:off + p.numel()].view_as(p)
                if keep and p.grad is not None:
                    view.copy_(p.grad)
                p.grad = view
                off += p.numel()
                self._of[p] = bi
            self._flat.append(flat)
            self._ready.append(0)
            self._work.append(None)
        self.nbytes = sum(f.numel() * 4 for f in self._flat)

    def _rebuild(self):
        """After the first step: bucket fill order = gradient arrival order of rank 0 (parameters that never arrived go
        last, in reverse registration order).  One small broadcast; every rank ends with the same layout."""
        index = {p: i for i, p in enumerate(self.params)}
        seen = set(self._arrival)
        ids = [index[p] for p in self._arrival] + [index[p] for p in reversed(self.params) if p not in seen]
        self._arrival = None
        if self.world > 1:
            t = torch.tensor(ids, dtype=torch.int64, device=self._flat[0].device)
            src = 0 if self.group is None else dist.get_global_rank(self.group, 0)
            dist.broadcast(t, src=src, group=self.group)
            ids = t.tolist()
        self._layout([self.params[i] for i in ids], keep=True)

    def _launch(self, bi):
        if self.world > 1 and self._work[bi] is None:
            self._work[bi] = dist.all_reduce(self._flat[bi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _on_grad(self, p):
        bi = self._of[p]
        if p.grad.data_ptr() < self._flat[bi].data_ptr() or \
                p.grad.data_ptr() >= self._flat[bi].data_ptr() + self._flat[bi].numel() * 4:
            raise RuntimeError('GradBuckets: a .grad was replaced (zero_grad(set_to_none=True)?); use buckets.zero_grad()')
        if self._hold:
            return                         # non-final pass of a gradient accumulation: summed into the bucket, not counted
        if p in self._seen:
            # a second backward() before finish() without no_sync(): counting the parameter twice would leave duplicate slots
            # in the re-laid-out buckets and a ready count no bucket ever reaches; and a bucket that already went
            # out must not be accumulated into while its all-reduce is in flight
            if self._work[bi] is not None:
                raise RuntimeError('GradBuckets: a gradient arrived again after its bucket was all-reduced; call finish() '
                                   'after every backward(), or accumulate under `with buckets.no_sync():`')
            return
        self._seen.add(p)
        self._ready[bi] += 1
        if self._arrival is not None:
            self._arrival.append(p)
        # strictly in index order: launch the run of complete buckets that starts at the first unlaunched one
        while self._next < len(self.buckets) and self._ready[self._next] == len(self.buckets[self._next]):
            self._launch(self._next)
            self._next += 1

    def no_sync(self):
        """Context for the non-final backward passes of gradient accumulation: gradients accumulate into the flat buckets,
        nothing is launched; the next backward() outside the context launches, or finish() does."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            self._hold = True
            try:
                yield
            finally:
                self._hold = False
        return ctx()

    def zero_grad(self):
        """Zero the flat buckets in place (the .grad views stay attached)."""
        for f in self._flat:
            f.zero_()

    def finish(self):
        """Call after backward(), before clipping / the optimizer step: every gradient is the mean over ranks."""
        for bi in range(self._next, len(self._flat)):
            self._launch(bi)
        self._next = 0
        self._seen.clear()
        for bi, w in enumerate(self._work):
            if w is not None:
                w.wait()
                self._flat[bi].div_(self.world)
            self._work[bi] = None
            self._ready[bi] = 0
        if self._arrival is not None:
            self._rebuild()

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def allreduce_gradients(params, bucket_bytes=64 << 20, async_op=False):
    """One-shot form of the same collective for code without hooks: averages the gradients of EVERY requires_grad
    parameter in a fixed order (a parameter without a gradient on this rank contributes zeros, so the bucket layout
    is identical on all ranks).  Returns pending handles when async_op=True (`finish_allreduce`)."""
    params = [p for p in params if p.requires_grad]
    if not dist.is_initialized() or dist.get_world_size() == 1 or not params:
        return []
    world = dist.get_world_size()
    buckets, cur, size = [], [], 0
    for p in params:
        n = p.numel() * 4
        if cur and size + n > bucket_bytes:
            buckets.append(cur)
            cur, size = [], 0
        cur.append(p)
        size += n
    if cur:
        buckets.append(cur)
    pending = []
    for b in buckets:
        flat = torch.zeros(sum(p.numel() for p in b), dtype=torch.float32, device=b[0].device)
        off = 0
        for p in b:
            if p.grad is not None:
                flat[off:off + p.numel()].copy_(p.grad.reshape(-1))
            off += p.numel()
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
        for p in bucket:
            n = p.numel()
            if p.grad is not None:
                p.grad.copy_(flat[off:off + n].view_as(p))
            off += n


# ------------------------------------------------------------------ cross-rank batch norm (the configs' `SyncBN`)
class _SyncBNFunction(torch.autograd.Function):
    """Batch-norm over the samples of ALL ranks: forward all-reduces (count, sum, sum of squares) per channel, backward
    all-reduces (sum dy, sum dy*xhat).  Plain torch ops + dist.all_reduce, so it runs on RCCL and on gloo (the CPU
    tests); torch.nn.SyncBatchNorm refuses CPU tensors."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, group):
        C = x.shape[1]
        dims = [0] + list(range(2, x.dim()))
        xf = x.float()
        stat = torch.empty(2 * C + 1, dtype=torch.float32, device=x.device)
        stat[:C] = xf.sum(dims)
        stat[C:2 * C] = (xf * xf).sum(dims)
        stat[2 * C] = x.numel() // C
        dist.all_reduce(stat, group=group)
        n = stat[2 * C]
        mean = stat[:C] / n
        var = (stat[C:2 * C] / n - mean * mean).clamp_(min=0)
        invstd = torch.rsqrt(var + eps)
        shape = [1, C] + [1] * (x.dim() - 2)
        xhat = (xf - mean.view(shape)) * invstd.view(shape)
        out = xhat
        if weight is not None:
            out = out * weight.float().view(shape) + bias.float().view(shape)
        ctx.save_for_backward(xhat, invstd, weight)
        ctx.group, ctx.n = group, n
        ctx.mark_non_differentiable(mean, var, n)
        return out.to(x.dtype), mean, var, n

    @staticmethod
    def backward(ctx, dy, _m, _v, _n):
        xhat, invstd, weight = ctx.saved_tensors
        C = xhat.shape[1]
        dims = [0] + list(range(2, xhat.dim()))
        shape = [1, C] + [1] * (xhat.dim() - 2)
        dyf = dy.float()
        red = torch.empty(2 * C, dtype=torch.float32, device=dy.device)
        red[:C] = dyf.sum(dims)
        red[C:] = (dyf * xhat).sum(dims)
        dw = red[C:].clone() if weight is not None else None     # parameter gradients stay per-rank (averaged later)
        db = red[:C].clone() if weight is not None else None
        dist.all_reduce(red, group=ctx.group)
        g = dyf if weight is None else dyf * weight.float().view(shape)
        w = 1.0 if weight is None else weight.float()
        mean_dy = (red[:C] * w / ctx.n).view(shape)
        mean_dyx = (red[C:] * w / ctx.n).view(shape)
        dx = (g - mean_dy - xhat * mean_dyx) * invstd.view(shape)
        return dx.to(dy.dtype), dw, db, None, None


class SyncBatchNorm(torch.nn.modules.batchnorm._BatchNorm):
    """`norm_cfg=dict(type='SyncBN')` of the FB-OCC configs (CustomResNet3D / FPN3D / OccHead): statistics over the
    global batch while training in a multi-rank job; identical to BatchNorm otherwise (same parameters and buffers)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True, process_group=None):
        super().__init__(num_features, eps, momentum, affine, track_running_stats)
        self.process_group = process_group
        self.sync = True                         # False: per-rank statistics (plain BatchNorm), no collective

    def _check_input_dim(self, x):
        if x.dim() < 2:
            raise ValueError(f'expected at least 2D input (got {x.dim()}D input)')

    def forward(self, x):
        sync = self.sync and self.training and dist.is_initialized() and dist.get_world_size(self.process_group) > 1
        if not sync:
            from .bev_encoder import _is_channels_last_3d, batch_norm_channels_last_3d
            if _is_channels_last_3d(x):                    # keep the convolution kernels' layout (no NCDHW round trip)
                return batch_norm_channels_last_3d(super().forward, x)
            return super().forward(x)
        out, mean, var, n = _SyncBNFunction.apply(x, self.weight, self.bias, self.eps, self.process_group)
        if self.track_running_stats:
            with torch.no_grad():
                self.num_batches_tracked += 1
                m = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked)
                self.running_mean.mul_(1 - m).add_(mean.to(self.running_mean.dtype), alpha=m)
                self.running_var.mul_(1 - m).add_((var * (n / (n - 1).clamp(min=1))).to(self.running_var.dtype), alpha=m)
        return out


def convert_sync_batchnorm(module, process_group=None, sync=True):
    """Replace every norm layer the config declared as `SyncBN` (marked `_fbbev_sync_bn` by build_norm) -- and every
    torch.nn.SyncBatchNorm -- by `SyncBatchNorm` above; parameters / buffers are shared, names unchanged.
    sync=False keeps the same layers on per-rank statistics (the bench's `sync_bn` off setting)."""
    def conv(m):
        if isinstance(m, torch.nn.SyncBatchNorm) or getattr(m, '_fbbev_sync_bn', False):
            if isinstance(m, SyncBatchNorm):
                m.sync = bool(sync)
                return m
            new = SyncBatchNorm(m.num_features, m.eps, m.momentum, m.affine, m.track_running_stats, process_group)
            if m.affine:
                new.weight, new.bias = m.weight, m.bias
            if m.track_running_stats:
                new.running_mean, new.running_var, new.num_batches_tracked = m.running_mean, m.running_var, m.num_batches_tracked
            new.training = m.training
            new._fbbev_sync_bn = True
            new.sync = bool(sync)
            return new
        for name, child in list(m.named_children()):
            setattr(m, name, conv(child))
        return m
    return conv(module)


def prepare_ddp(model, sync_bn=True, bucket_bytes=64 << 20, process_group=None):
    """What MMDistributedDataParallel + the config's SyncBN do for the reference (apis/train.py:229-233), for one model
    replica per rank: cross-rank batch-norm statistics for the layers the config declares `SyncBN`, and gradient
    buckets reduced from autograd hooks.  -> (model, GradBuckets); call buckets.zero_grad() / buckets.finish() around
    backward().  Parameters must already be identical on all ranks (same seed or a broadcast checkpoint)."""
    model = convert_sync_batchnorm(model, process_group, sync=sync_bn)
    grad_group = process_group
    if dist.is_initialized() and dist.get_world_size(process_group) > 1:
        # the gradient buckets get their OWN communicator: collectives are matched per group by issue order, and the
        # SyncBN backward all-reduces run on `process_group` in between the hook-launched buckets (collective call:
        # every rank of the job passes through here)
        ranks = None if process_group is None else dist.get_process_group_ranks(process_group)
        grad_group = dist.new_group(ranks=ranks)
    return model, GradBuckets([p for p in model.parameters() if p.requires_grad], bucket_bytes, grad_group)
