"""N>1 path on CPU: 2 processes over gloo exercise the sharding + fence + max-over-ranks logic that
bench.py uses with RCCL on the GPUs (the data path itself has no collective)."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank))
    from fb_bev_amd import shard, synthetic as S
    from oracle import oracle as O
    ws, r = shard.init('gloo')
    assert (ws, r) == (world, rank)
    cfg = S.CONFIGS['TINY']
    cam = S.camera_rig(cfg, 2, seed=shard.shard_seed(rank), bda_aug=True)
    ovt = O.ViewTransformerOracle(cfg.grid_config, cfg.input_size, cfg.downsample)
    rb, rd, rf, st, ln = ovt.voxel_pooling_prepare_v2(ovt.get_lidar_coor(*cam))
    shard.fence(None)
    elapsed = 0.5 + rank                      # rank 1 is the slow one
    mx = shard.max_over_ranks(elapsed)
    total_pts = shard.sum_over_ranks(rb.numel())
    shard.fence(None)
    q.put((rank, mx, total_pts, int(rb.numel()), float(cam[5].sum())))
    torch.distributed.destroy_process_group()


def test_two_rank_shards_and_max_reduce():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, mx0, tot0, n0, bda0), (r1, mx1, tot1, n1, bda1) = res
    assert mx0 == mx1 == 1.5                  # MAX over ranks, seen by both
    assert tot0 == tot1 == n0 + n1            # shards are disjoint work, summed only for reporting
    assert bda0 != bda1                       # each rank really has its own samples (own augmentation)
    from fb_bev_amd import shard
    assert shard.whole_job_rate(16, 10, 2.0, 8) == 16 * 8 * 10 / 2.0


def test_single_process_helpers_are_noops():
    from fb_bev_amd import shard
    assert shard.max_over_ranks(0.25) == 0.25 and shard.sum_over_ranks(7) == 7
    shard.fence(None)


def _grad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank))
    from fb_bev_amd import shard
    from fb_bev_amd.history_fusion import TemporalHistoryFusion
    shard.init('gloo')
    torch.manual_seed(0)                                    # same initial parameters on every rank
    m = TemporalHistoryFusion([0.8, 0.8, 0.8], [-3.6, -3.6, -0.6], single_bev_num_channels=4, history_cat_num=2)
    params = list(m.parameters())
    for i, p in enumerate(params):                          # rank-dependent gradients with a known mean
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    params[1].grad = None                                   # a parameter without gradient is skipped
    pending = shard.allreduce_gradients(params, bucket_bytes=256, async_op=True)   # tiny buckets: several messages
    assert len(pending) > 1
    shard.finish_allreduce(pending)
    ok = all(torch.allclose(p.grad, torch.full_like(p, 1.5 * (i + 1))) for i, p in enumerate(params) if p.grad is not None)
    q.put((rank, ok, params[1].grad is None))
    torch.distributed.destroy_process_group()


def test_two_rank_gradient_allreduce_of_path_parameters():
    """The training step's only collective: bucketed average of the path's parameter gradients."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok and none_kept for _, ok, none_kept in res)
