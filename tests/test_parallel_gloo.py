"""World-size-2 checks of the data-parallel path on CPU with the gloo backend (the GPU path uses the
same code with the nccl/RCCL backend)."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from adam_dehaze_amd.parallel import GradientSynchronizer
    torch.manual_seed(0)   # identical replicas
    params = [torch.nn.Parameter(torch.randn(s)) for s in ((64, 32, 3, 3), (64,), (7, 5), ())]
    g = torch.Generator().manual_seed(100 + rank)   # different shard -> different gradients
    for p in params:
        p.grad = torch.randn(p.shape, generator=g)
    local = [p.grad.clone() for p in params]
    sync = GradientSynchronizer(params, world, bucket_bytes=200)
    assert len(sync.buckets) >= 2
    sync.all_reduce()
    # expected: mean over ranks of the per-rank gradients
    exp = []
    for i, p in enumerate(params):
        acc = torch.zeros_like(p)
        for r in range(world):
            gg = torch.Generator().manual_seed(100 + r)
            gs = [torch.randn(q.shape, generator=gg) for q in params]
            acc += gs[i]
        exp.append(acc / world)
    ok = all(torch.allclose(p.grad, e, atol=1e-6) for p, e in zip(params, exp))
    changed = any(not torch.allclose(p.grad, l) for p, l in zip(params, local))
    # every rank ends with bit-identical gradients
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    same = all(torch.equal(gathered[0], t) for t in gathered)
    ret[rank] = bool(ok and changed and same)
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_synchronizer_world2_gloo():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 500)
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_synchronizer_is_noop_for_single_rank():
    from adam_dehaze_amd.parallel import GradientSynchronizer
    p = torch.nn.Parameter(torch.ones(3))
    p.grad = torch.full((3,), 2.0)
    GradientSynchronizer([p], 1).all_reduce()
    assert torch.equal(p.grad, torch.full((3,), 2.0))


def test_reduce_lr_on_plateau_host_logic():
    from adam_dehaze_amd.train import ReduceLROnPlateau

    class Opt:
        param_groups = [{"lr": 1.0}]
    s = ReduceLROnPlateau(Opt, factor=0.5, patience=3)
    for v in (1.0, 0.9, 0.9, 0.9, 0.9):
        s.step(v)
    assert Opt.param_groups[0]["lr"] == 1.0
    s.step(0.9)
    assert Opt.param_groups[0]["lr"] == 0.5
