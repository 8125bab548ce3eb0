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
    # sync-BN hook (SURVEY 8e mode ii): installed only on request, sums the [sum | sumsq | count] vector over the ranks in
    # place, removed by uninstall
    from adam_dehaze_amd import engine as E
    plain = GradientSynchronizer(list(m.parameters()), world)
    plain.install()
    ok &= E.SYNC_BN is None
    plain.uninstall()
    sb = GradientSynchronizer(list(m.parameters()), world, sync_bn=True)
    sb.install()
    ok &= E.SYNC_BN is not None
    v = torch.tensor([1.0 + rank, 10.0 * (rank + 1), 8.0], dtype=torch.float64)
    E.SYNC_BN(v)
    ok &= torch.equal(v, torch.tensor([3.0, 30.0, 16.0], dtype=torch.float64))
    sb.uninstall()
    ok &= E.SYNC_BN is None and E.GRAD_READY is None
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def _worker_validation_and_checkpoints(rank, world, port, ret):
    """(a) a rank whose validation shard holds no image of the branch's level (or whose loader is empty) must
    still enter the all-reduce of the validation sums -- it used to return early and leave the others in the collective;
    both ranks get the same sample-weighted result.  (b) buffers are taken from rank 0 before validation / checkpoints.
    (c) checkpoints are written atomically and a barrier separates rank 0's write from the other ranks' reads."""
    _init(rank, world, port)
    import adam_dehaze_amd.train as T
    from adam_dehaze_amd.parallel import GradientSynchronizer
    # CPU stand-ins for the device metric kernels (the collective protocol is what is under test)
    T.psnr_batch = lambda a, b: -10.0 * torch.log10(((a - b) ** 2).mean(dim=(1, 2, 3)))
    T.ssim_batch = lambda a, b: torch.ones(a.shape[0])

    class Model(torch.nn.Module):
        def forward(self, x):
            return x * 0.5

    def crit(out, target):
        l = (out - target).abs().mean()
        return l, {"l1": l, "content": l * 0, "perceptual": l * 2, "total": l}
    g = torch.Generator().manual_seed(5)
    imgs = torch.rand(4, 3, 8, 8, generator=g)
    # rank 0: two images of level 1; rank 1: level 0 / 2 only -> nothing to validate there
    labels = torch.tensor([1, 0, 1, 2]) if rank == 0 else torch.tensor([0, 2, 0, 2])
    loader = [{"hazy": imgs, "clear": imgs * 0.9, "intensity": labels}]
    val = T.validate_dehazing(Model(), crit, loader, 1, "cpu")
    keep = torch.tensor([True, False, True, False])
    want_loss = float((imgs[keep] * 0.5 - imgs[keep] * 0.9).abs().mean())
    ok = val["val_samples"] == 2 and abs(val["val_loss"] - want_loss) < 1e-6 and abs(val["val_perceptual"] - 2 * want_loss) < 1e-6
    # an entirely empty loader on one rank, joint flavour: every rank still reaches the collective and sees the same result
    val2 = T.validate_dehazing(Model(), crit, loader if rank == 1 else [], 0, "cpu")
    ok &= val2["val_samples"] == 2
    both = [None, None]
    dist.all_gather_object(both, (val, val2))
    ok &= both[0] == both[1]
    # (b) buffers from rank 0
    torch.manual_seed(10 + rank)
    m = torch.nn.BatchNorm2d(3)
    m.running_mean.copy_(torch.randn(3))
    w_before = m.weight.detach().clone() + rank
    m.weight.data.copy_(w_before)
    T._sync_buffers_from_rank0(GradientSynchronizer(list(m.parameters()), world), m)
    got = [None, None]
    dist.all_gather_object(got, (m.running_mean.tolist(), m.weight.tolist()))
    ok &= got[0][0] == got[1][0] and got[0][1] != got[1][1]        # buffers agree; parameters are left alone
    # (c) atomic save + barrier: rank 1 loads what rank 0 wrote, never a partial file, and no temp file is left behind
    import tempfile
    d = os.path.join(tempfile.gettempdir(), f"adh_ckpt_test_{port}")
    path = os.path.join(d, "best_model.pth")
    if rank == 0:
        os.makedirs(d, exist_ok=True)
        T.save_checkpoint_atomic({"epoch": 3, "model_state_dict": {"w": torch.randn(1 << 16)}}, path)
    T._barrier()
    ck = torch.load(path, map_location="cpu")
    ok &= ck["epoch"] == 3 and ck["model_state_dict"]["w"].numel() == 1 << 16
    ok &= [f for f in os.listdir(d) if ".tmp." in f] == []
    T._barrier()
    if rank == 0:
        import shutil
        shutil.rmtree(d, ignore_errors=True)
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def _worker_selfcheck(rank, world, port, ret):
    """GradientSynchronizer.selfcheck_result (bench.py prints it as `ddp_selfcheck` when world > 1): a clean step passes on every
    rank; a bucket one rank corrupts after synchronisation, parameters that differ by one ulp on one rank, and a step that was
    never armed are all reported as failures, by every rank alike."""
    _init(rank, world, port)
    from adam_dehaze_amd.parallel import GradientSynchronizer
    params = _params()
    sync = GradientSynchronizer(params, world, bucket_bytes=300, rebuild=False)
    assert len(sync.buckets) >= 2

    def step(corrupt=False, arm=True):
        for p, g in zip(params, _rank_grads(rank)):
            p.grad = g.clone()
        if arm:
            sync.begin_selfcheck()
        sync.begin_step()
        sync.finish()
        if corrupt and rank == 1:
            sync.arena[3] += 1e-3
    step()
    with torch.no_grad():
        for p in params:                                  # identical "optimizer step" on identical synchronized gradients
            p -= 0.1 * p.grad
    good = sync.selfcheck_result(params)
    ok = good["ok"] and good["backend"] == "gloo" and good["max_rel"] < 1e-6 and good["params_bit_equal"] and good["buckets"] == len(sync.buckets)
    step(corrupt=True)
    bad = sync.selfcheck_result(params)
    ok &= (not bad["ok"]) and bad["max_rel_across_ranks"] > 1e-7
    step(arm=False)
    unarmed = sync.selfcheck_result(params)
    ok &= (not unarmed["ok"]) and "error" in unarmed
    step()
    if rank == 1:
        with torch.no_grad():
            params[0].view(-1)[0] = torch.nextafter(params[0].view(-1)[0], torch.tensor(10.0))
    drift = sync.selfcheck_result(params)
    ok &= (not drift["ok"]) and drift["max_rel"] < 1e-6 and not drift["params_bit_equal"]
    both = [None, None]
    dist.all_gather_object(both, (good["ok"], bad["ok"], drift["ok"], unarmed["ok"]))
    ok &= both[0] == both[1]
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


def test_selfcheck_reports_clean_and_planted_failures_world2_gloo():
    _spawn(_worker_selfcheck, 31300)


def test_rank_agreements_world2_gloo():
    _spawn(_worker_agreements, 30700)


def test_validation_collectives_buffers_and_atomic_checkpoints_world2_gloo():
    _spawn(_worker_validation_and_checkpoints, 31300)


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
