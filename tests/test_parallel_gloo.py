"""World-size-2 checks of the data-parallel path on CPU with the gloo backend (the GPU path uses the
same code with the nccl/RCCL backend): flat-bucket gradient averaging, the overlapped protocol driven
through the engine's grad-ready hook, unused-parameter detection, the bucket rebuild, and the host-side
agreements (scheduler metric, empty-sub-batch skip) that keep replicas identical."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

SHAPES = ((64, 32, 3, 3), (64,), (7, 5), (), (130,))


def _params():
    torch.manual_seed(0)   # identical replicas
    return [torch.nn.Parameter(torch.randn(s)) for s in SHAPES]


def _rank_grads(rank):
    g = torch.Generator().manual_seed(100 + rank)   # different shard -> different gradients
    return [torch.randn(s, generator=g) for s in SHAPES]


def _expected(world, skip=()):
    acc = [torch.zeros(s) for s in SHAPES]
    for r in range(world):
        for i, g in enumerate(_rank_grads(r)):
            if (r, i) not in skip:
                acc[i] += g
    return [a / world for a in acc]


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _same_everywhere(params, world):
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    return all(torch.equal(gathered[0], t) for t in gathered)


def _worker_plain(rank, world, port, ret):
    _init(rank, world, port)
    from adam_dehaze_amd.parallel import GradientSynchronizer
    params = _params()
    for p, g in zip(params, _rank_grads(rank)):
        p.grad = g.clone()
    local = [p.grad.clone() for p in params]
    sync = GradientSynchronizer(params, world, bucket_bytes=200, rebuild=False)
    assert len(sync.buckets) >= 2
    sync.all_reduce()
    ok = all(torch.allclose(p.grad, e, atol=1e-6) for p, e in zip(params, _expected(world)))
    changed = any(not torch.allclose(p.grad, l) for p, l in zip(params, local))
    # p.grad IS the slice of the flat arena (no copy back), 256-byte aligned
    in_arena = all(p.grad.data_ptr() == sync.views[i].data_ptr() and sync.offset[i] % 64 == 0
                   for i, p in enumerate(params))
    ret[rank] = bool(ok and changed and in_arena and _same_everywhere(params, world))
    dist.barrier()
    dist.destroy_process_group()


def _worker_hooks(rank, world, port, ret):
    """The overlapped protocol as the engine drives it: begin_step, GRAD_SINK hands out arena slices, GRAD_READY fires
    per parameter in backward order, buckets go out as they fill, finish() waits; then a second step on the rebuilt
    (observed-order) layout; then a step in which rank 1 does not produce parameter 2 (hard routing)."""
    _init(rank, world, port)
    from adam_dehaze_amd import engine as E
    from adam_dehaze_amd.parallel import GradientSynchronizer
    params = _params()
    sync = GradientSynchronizer(params, world, bucket_bytes=300, detect_unused=True)
    sync.install()
    ok = True
    # the "model" reports gradients in this order (differs from reversed registration order -> rebuild must follow it)
    backward_order = [2, 4, 0, 1, 3]
    layouts = []
    for step in range(3):
        for p in params:
            p.grad = None
        sync.begin_step()
        grads = _rank_grads(rank)
        skip = set()
        launched_before_finish = 0
        for i in backward_order:
            if step == 2 and i == 2 and rank == 1:
                skip.add((1, 2))
                continue
            buf = E.GRAD_SINK(params[i])
            if i % 2 == 0 and buf is not None:      # engine kernels write straight into the slice ...
                buf.copy_(grads[i])
                E.GRAD_READY(params[i], buf)
            else:                                   # ... small ones arrive as separate tensors and are copied in
                E.GRAD_READY(params[i], grads[i].clone())
            launched_before_finish = max(launched_before_finish, sync._next_bucket)
        layouts.append(list(sync.order))
        sync.finish()
        if step == 2:
            skip_all = {(1, 2)}
            exp = _expected(world, skip_all)
        else:
            exp = _expected(world)
        ok &= all(p.grad is not None and torch.allclose(p.grad, e, atol=1e-6) for p, e in zip(params, exp))
        ok &= _same_everywhere(params, world)
        if step == 1:
            ok &= launched_before_finish >= 1     # at least one bucket went out while "backward" was still running
        if step == 2 and rank == 1:
            ok &= launched_before_finish == 0     # bucket 0 waits for the parameter this rank never produced: in-order
    # after the first step the layout follows the observed order
    ok &= layouts[0] == [4, 3, 2, 1, 0] and layouts[1] == backward_order and layouts[2] == backward_order
    # a parameter NO rank produced keeps grad None with detect_unused
    for p in params:
        p.grad = None
    sync.begin_step()
    grads = _rank_grads(rank)
    for i in backward_order:
        if i != 3:
            E.GRAD_READY(params[i], grads[i].clone())
    sync.finish()
    ok &= params[3].grad is None and all(params[i].grad is not None for i in (0, 1, 2, 4))
    sync.uninstall()
    ok &= E.GRAD_SINK is None and E.GRAD_READY is None
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def _worker_agreements(rank, world, port, ret):
    """a plateau must change the learning rate on every rank or on none -- the scheduler is stepped on the
    rank-averaged metric; a rank with an empty sub-batch may only skip when all ranks skip; parameters are broadcast."""
    _init(rank, world, port)
    from adam_dehaze_amd.parallel import GradientSynchronizer, all_ranks_any, all_reduce_mean_scalar
    from adam_dehaze_amd.train import ReduceLROnPlateau

    class Opt:
        param_groups = [{"lr": 1.0}]
    sched = ReduceLROnPlateau(Opt, factor=0.5, patience=1)
    # local metrics: rank 0 keeps improving, rank 1 plateaus -> un-synchronised schedulers would diverge
    local = {0: [1.0, 0.8, 0.6, 0.4, 0.2], 1: [1.0, 1.2, 1.4, 1.6, 1.8]}[rank]
    lrs = []
    for v in local:
        sched.step(all_reduce_mean_scalar(v))
        lrs.append(Opt.param_groups[0]["lr"])
    t = torch.tensor(lrs, dtype=torch.float64)
    both = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(both, t)
    ok = torch.equal(both[0], both[1]) and lrs[-1] < 1.0          # same schedule everywhere, and the plateau did fire
    ok &= all_ranks_any(rank == 1) is True and all_ranks_any(False) is False
    torch.manual_seed(rank)                                       # replicas that start different ...
    m = torch.nn.Linear(4, 3)
    m.register_buffer("running", torch.randn(3))
    GradientSynchronizer(list(m.parameters()), world).broadcast_parameters(m)
    flat = torch.cat([x.detach().reshape(-1) for x in list(m.parameters()) + list(m.buffers())])
    got = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(got, flat)
    ok &= torch.equal(got[0], got[1])                             # ... are identical after the broadcast
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def _spawn(fn, base):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = base + (os.getpid() % 400)
    mp.spawn(fn, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_gradient_synchronizer_world2_gloo():
    _spawn(_worker_plain, 29500)


def test_overlapped_protocol_rebuild_and_unused_world2_gloo():
    _spawn(_worker_hooks, 30100)


def test_rank_agreements_world2_gloo():
    _spawn(_worker_agreements, 30700)


def test_synchronizer_is_noop_for_single_rank():
    from adam_dehaze_amd.parallel import GradientSynchronizer
    p = torch.nn.Parameter(torch.ones(3))
    p.grad = torch.full((3,), 2.0)
    GradientSynchronizer([p], 1).all_reduce()
    assert torch.equal(p.grad, torch.full((3,), 2.0))


def test_single_rank_hooks_leave_gradients_in_the_arena():
    from adam_dehaze_amd import engine as E
    from adam_dehaze_amd.parallel import GradientSynchronizer
    params = _params()
    sync = GradientSynchronizer(params, 1)
    sync.install()
    try:
        sync.begin_step()
        for i, (p, g) in enumerate(zip(params, _rank_grads(0))):
            buf = E.GRAD_SINK(p)
            assert buf is not None and buf.shape == p.shape
            buf.copy_(g)
            E.GRAD_READY(p, buf)
        sync.finish()
        for p, g in zip(params, _rank_grads(0)):
            assert torch.equal(p.grad, g)
    finally:
        sync.uninstall()


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
    s2 = ReduceLROnPlateau(Opt, factor=0.5, patience=3)
    s2.load_state_dict(s.state_dict())
    assert (s2.best, s2.bad) == (s.best, s.bad)
