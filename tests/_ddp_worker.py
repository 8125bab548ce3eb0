"""Worker of tests/test_gpu_ddp.py: one data-parallel rank of the real training step on GPU 0 (gloo rendezvous on
127.0.0.1; every rank uses the one card of the test box).  Started as a fresh process, before anything touches the GPU.

    python tests/_ddp_worker.py RANK WORLD PORT OUTDIR
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, outdir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    backend = os.environ.get("ADH_DDP_BACKEND", "gloo")     # "nccl" = RCCL: one GPU per rank (needs >= world GPUs)
    gpu = rank if backend == "nccl" else 0
    if backend == "nccl":
        torch.cuda.set_device(gpu)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", gpu))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import adam_dehaze_amd as A
    from adam_dehaze_amd.loss import l1_loss
    from adam_dehaze_amd.optim import Adam
    from adam_dehaze_amd.parallel import GradientSynchronizer, all_reduce_mean_scalar
    from adam_dehaze_amd.train import ReduceLROnPlateau
    from oracle import ref_cpu as R          # test infrastructure: only its synthetic_batch (the shared input recipe)

    dev = torch.device("cuda", gpu)
    torch.cuda.set_device(dev)
    torch.manual_seed(100 + rank)            # replicas start DIFFERENT: the broadcast must fix that
    model = A.HighIntensityDehazeModel(base_channels=16).to(dev).train()
    params = list(model.parameters())
    sync_bn = os.environ.get("ADH_DDP_SYNC_BN", "0") == "1"
    sync = GradientSynchronizer(params, world, bucket_bytes=64 << 10, sync_bn=sync_bn)
    sync.broadcast_parameters(model)
    sync.install()
    opt = Adam(params, lr=1e-3, weight_decay=1e-4)
    sched = ReduceLROnPlateau(opt, factor=0.5, patience=0)
    hazy, clear, _ = R.synthetic_batch(2 * world, 32, 48, seed=77)
    x, y = hazy[2 * rank:2 * rank + 2].to(dev), clear[2 * rank:2 * rank + 2].to(dev)
    rec = {"sd0": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, "nbuckets": [], "early": [],
           "order": []}
    import adam_dehaze_amd.engine as E
    for step in range(3):
        opt.zero_grad()
        if step == 1:
            sync.begin_selfcheck()              # (the bucket layout is rebuilt in step 0: the step after is the one checked)
        sync.begin_step()
        if step == 0:
            E.RELU_CAPTURE = {}
            E.CBAM_CAPTURE = {}
        out = model(x)
        loss = l1_loss(out, y)
        loss.backward()
        if step == 0:   # the ReLU masks this rank's kernels used, for the kink-matched oracle of the parent test
            rec["masks"] = {k: (E.RELU_CAPTURE[id(p)] > 0).permute(0, 3, 1, 2).cpu() for k, p in model.named_parameters()
                            if id(p) in E.RELU_CAPTURE}
            E.RELU_CAPTURE = None
            # ... and the arg-max positions of its attention blocks (the network's other kinks: one flipped arg-max moves the
            # gradients behind the block by 1e-3 .. 1e-2, tests/_util.py kink_matched.cbam_indices)
            rec["cbam"] = {k: tuple(t.cpu() for t in E.CBAM_CAPTURE[id(p)]) for k, p in model.named_parameters()
                           if id(p) in E.CBAM_CAPTURE}
            E.CBAM_CAPTURE = None
        rec["early"].append(sync._next_bucket)          # buckets already in flight when backward returned
        rec["nbuckets"].append(len(sync.buckets))
        rec["order"].append(list(sync.order))
        sync.finish()
        if step == 0:
            rec["grads"] = {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters()}
            rec["in_arena"] = all(p.grad.data_ptr() == sync.views[sync.index[id(p)]].data_ptr() if sync.views[0] is not None
                                  else True for p in params)
            rec["bn"] = {k: v.detach().cpu().clone() for k, v in model.state_dict().items() if "running" in k}
            rec["loss"] = float(loss)
            rec["out"] = out.detach().cpu().clone()
        opt.step()
        if step == 1:
            rec["selfcheck"] = sync.selfcheck_result(params)
        # rank-dependent raw metric (rank 0 improves, rank 1 gets worse): only the rank-mean may drive the scheduler
        raw = (1.0 - 0.3 * step) if rank == 0 else (1.0 + 0.5 * step)
        sched.step(all_reduce_mean_scalar(raw, dev))
    torch.cuda.synchronize()
    rec["lr"] = opt.param_groups[0]["lr"]
    rec["params"] = {k: p.detach().cpu().clone() for k, p in model.named_parameters()}
    sync.uninstall()
    torch.save(rec, os.path.join(outdir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
