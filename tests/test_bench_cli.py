"""bench.py command line: the contract flags exist and --help renders (argparse %-formats every help string); the self-launch
builds exactly the driver's `torch.distributed.run` command line and refuses when GPUs are missing; the rank check sees the
real process group (2 gloo ranks)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_help_renders_and_names_the_contract_flags():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--help'], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-400:]
    for flag in ('--gpus', '--steps', '--warmup', '--storage', '--mode'):
        assert flag in r.stdout, flag


def _bench():
    import importlib
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    return importlib.import_module('bench')


def test_self_launch_builds_the_drivers_command_line(monkeypatch):
    """`python bench.py --gpus N` without a launcher re-executes itself as
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <flags>`
    (tools/dist_train.sh:10-20 is the reference's launcher); under a launcher, or for one GPU, it does nothing."""
    import torch
    bench = _bench()
    argv = ['--gpus', '4', '--steps', '7', '--warmup', '2', '--mode', 'train']
    monkeypatch.setattr(sys, 'argv', ['bench.py'] + argv)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.setattr(torch.cuda, 'device_count', lambda: 8)
    seen = {}

    def fake_execv(exe, cmd):
        seen['exe'], seen['cmd'] = exe, cmd
        raise SystemExit(0)
    monkeypatch.setattr(os, 'execv', fake_execv)
    args = bench.parse()
    try:
        bench.self_launch(args)
    except SystemExit as e:
        assert e.code == 0
    cmd = seen['cmd']
    assert seen['exe'] == sys.executable and cmd[0] == sys.executable
    assert cmd[1:5] == ['-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=4']
    assert cmd[5:7] == ['--master-addr', '127.0.0.1'] and cmd[7] == '--master-port' and 1024 < int(cmd[8]) < 65536
    assert cmd[9] == os.path.join(ROOT, 'bench.py') and cmd[10:] == argv
    # fewer GPUs than ranks: refused before anything is launched
    monkeypatch.setattr(torch.cuda, 'device_count', lambda: 2)
    seen.clear()
    try:
        bench.self_launch(args)
        raise AssertionError('expected a refusal')
    except SystemExit as e:
        assert 'only 2 GPU(s) visible' in str(e.code) and not seen
    # under a launcher (WORLD_SIZE set) and for --gpus 1: no-op
    monkeypatch.setenv('WORLD_SIZE', '4')
    assert bench.self_launch(args) is None and not seen
    monkeypatch.delenv('WORLD_SIZE')
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '1'])
    assert bench.self_launch(bench.parse()) is None and not seen


def _rank_check_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank))
    import types
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import bench
    dist.init_process_group('gloo')
    cpu = torch.device('cpu')
    ok = bench.check_ranks(types.SimpleNamespace(gpus=2), world, rank, cpu, index=rank)
    res = [ok]
    for gpus, index in ((4, rank), (2, 0)):          # --gpus disagrees with the group; two ranks on one device
        try:
            bench.check_ranks(types.SimpleNamespace(gpus=gpus), world, rank, cpu, index=index)
            res.append('accepted')
        except SystemExit as e:
            res.append(str(e.code))
    q.put((rank, res))
    dist.destroy_process_group()


def test_rank_check_sees_the_process_group():
    """`rccl_ranks` / `rank_devices` of the JSON line come from the process group itself (2 gloo ranks here): size ==
    --gpus, one device per rank; a mismatch or a shared device ends the run instead of reporting a wrong n_gpus."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_check_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        ok, wrong_n, shared = res[r]
        assert ok == (2, [0, 1])
        assert 'the process group has 2 rank(s)' in wrong_n
        assert 'share a device' in shared
