"""Two data-parallel ranks of the REAL training step on one MI355X: each rank runs
CORUN-Complex (base 16) in train mode on its own shard with ADH_DIST_BACKEND-style gloo rendezvous, gradients go through
GradientSynchronizer's flat buckets, launched from inside Engine.backward().  Checked against the oracle run shard by
shard: synchronized gradients = mean of the shard gradients, BatchNorm buffers = each replica's own ("replica-BN"),
replicas identical after broadcast / Adam steps / a plateau-triggered LR change, buckets in flight before backward
returns.  The ranks are fresh child processes (never a re-exec of this one)."""
import os
import subprocess
import sys

import pytest
import torch
import torch.nn.functional as F

from oracle import ref_cpu as R
from tests._util import oracle_with_masks

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_ranks(tmp_path, world, port_base, **extra_env):
    """Start `world` fresh child processes (never a re-exec of this one) of tests/_ddp_worker.py and return their records."""
    port = port_base + os.getpid() % 2000
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_ddp_worker.py"), str(r), str(world), str(port),
                               str(tmp_path)], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode(errors="replace"))
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    return [torch.load(tmp_path / f"rank{r}.pt", map_location="cpu") for r in range(world)]


BACKENDS = [pytest.param("gloo", id="gloo-one-gpu"),
            # RCCL over xGMI, one GPU per rank: runs wherever the box has two GPUs (the one-GPU test box skips it; the nccl
            # branches of parallel.py -- ReduceOp.AVG, bucket all-reduces launched from inside backward onto RCCL's stream,
            # the sync-BN all-reduce on RCCL's stream -- are otherwise only rehearsed under gloo)
            pytest.param("nccl", id="rccl-two-gpus",
                         marks=pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL)"))]


@pytest.mark.parametrize("backend", BACKENDS)
def test_two_rank_training_step_matches_shardwise_oracle(tmp_path, backend):
    world = 2
    recs = _run_ranks(tmp_path, world, 31000, ADH_WINOGRAD="0", ADH_DDP_BACKEND=backend)   # direct kernels: reference summation order
    for r in range(world):       # the self-check bench.py prints as `ddp_selfcheck` (parallel.GradientSynchronizer.selfcheck_result)
        sc = recs[r]["selfcheck"]
        assert sc["ok"] and sc["backend"] == backend and sc["params_bit_equal"] and sc["max_rel"] < 1e-5, sc
    # replicas were seeded differently and broadcast from rank 0
    for k, v in recs[0]["sd0"].items():
        assert torch.equal(v, recs[1]["sd0"][k]), k
    # oracle, shard by shard
    hazy, clear, _ = R.synthetic_batch(2 * world, 32, 48, seed=77)
    shard_grads, shard_bn = [], []
    for r in range(world):
        sd = {k: v.clone() for k, v in recs[0]["sd0"].items()}
        for k, v in sd.items():
            if v.is_floating_point() and "running" not in k:
                v.requires_grad_(True)

        def run():
            out = R.high_forward(hazy[2 * r:2 * r + 2], sd, training=True)
            loss = F.l1_loss(out, clear[2 * r:2 * r + 2])
            loss.backward()
            return loss
        loss = oracle_with_masks(run, recs[r]["masks"], recs[r]["cbam"])      # the ReLU masks / attention arg-max positions that rank's kernels used (tests/_util.py)
        assert abs(float(loss) - recs[r]["loss"]) < 1e-5
        shard_grads.append({k: sd[k].grad for k in recs[0]["grads"]})
        shard_bn.append({k: v.detach() for k, v in sd.items() if "running" in k})
    for k in recs[0]["grads"]:
        mean = sum(g[k] for g in shard_grads) / world
        assert torch.equal(recs[0]["grads"][k], recs[1]["grads"][k]), k           # bit-identical on both ranks
        scale = max(float(mean.abs().max()), 1e-8)
        if k.startswith("decoder") and k.endswith(".0.bias"):
            continue      # ConvTranspose bias feeding train-mode BN: true gradient 0, the oracle holds rounding noise
        assert float((recs[0]["grads"][k] - mean).abs().max()) < 5e-3 * scale + 2e-7, k
    for r in range(world):
        for k, v in shard_bn[r].items():
            assert float((recs[r]["bn"][k] - v).abs().max()) < 1e-5, (r, k)       # replica-BN: each rank its own shard
    assert any(not torch.equal(recs[0]["bn"][k], recs[1]["bn"][k]) for k in recs[0]["bn"])
    # after three optimizer steps and a rank-averaged plateau decision: identical replicas, identical learning rate
    assert recs[0]["lr"] == recs[1]["lr"] and recs[0]["lr"] < 1e-3
    for k, v in recs[0]["params"].items():
        assert torch.equal(v, recs[1]["params"][k]), k
        assert not torch.equal(v, recs[0]["sd0"][k]) or v.numel() == 0 or k.endswith(".0.bias"), k
    # several buckets, rebuilt in observed order after step 0, and in flight before backward returned from step 1 on
    for r in range(world):
        assert recs[r]["nbuckets"][0] >= 3 and recs[r]["in_arena"]
        assert recs[r]["order"][1] == recs[r]["order"][2] and recs[r]["order"][0] != recs[r]["order"][1]
        assert recs[r]["early"][1] >= recs[r]["nbuckets"][1] - 1 and recs[r]["early"][2] >= 1
    assert recs[0]["order"][1] == recs[1]["order"][1]


@pytest.mark.parametrize("sync_bn", ["0", "1"])
def test_one_rank_rccl_rehearsal_issues_every_collective(tmp_path, sync_bn):
    """The RCCL branches of parallel.py on real hardware with the one GPU a test box has: a process group of ONE rank on the
    `nccl` backend with ADH_DIST_FORCE=1, which makes GradientSynchronizer issue every collective it would issue at N > 1 --
    the parameter broadcast, ReduceOp.AVG bucket all-reduces launched from inside Engine.backward() onto RCCL's stream, the
    bucket-order broadcast after step 0, the self-check's all_gather / MAX / MIN, and (sync_bn=1) the per-layer BatchNorm
    all-reduces on RCCL's stream.  With one rank every reduction is the identity, so the step must equal the oracle's."""
    recs = _run_ranks(tmp_path, 1, 35000, ADH_WINOGRAD="0", ADH_DDP_BACKEND="nccl", ADH_DIST_FORCE="1", ADH_DDP_SYNC_BN=sync_bn)
    rec = recs[0]
    sc = rec["selfcheck"]
    assert sc["ok"] and sc["backend"] == "nccl" and sc["params_bit_equal"] and sc["max_rel"] < 1e-5, sc
    hazy, clear, _ = R.synthetic_batch(2, 32, 48, seed=77)
    sd = {k: v.clone() for k, v in rec["sd0"].items()}
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)

    def run():
        out = R.high_forward(hazy, sd, training=True)
        loss = F.l1_loss(out, clear)
        loss.backward()
        return loss
    loss = oracle_with_masks(run, rec["masks"], rec["cbam"])
    assert abs(float(loss) - rec["loss"]) < 1e-5
    for k, g in rec["grads"].items():
        ref = sd[k].grad
        scale = max(float(ref.abs().max()), 1e-8)
        if k.startswith("decoder") and k.endswith(".0.bias"):
            continue      # ConvTranspose bias feeding train-mode BN: true gradient 0
        assert float((g - ref).abs().max()) < 5e-3 * scale + 2e-7, k
    for k, v in sd.items():
        if "running" in k:
            assert float((rec["bn"][k] - v.detach()).abs().max()) < 1e-5, k
    assert rec["nbuckets"][0] >= 3 and rec["in_arena"]
    assert rec["order"][1] == rec["order"][2] and rec["order"][0] != rec["order"][1]      # rebuilt in observed order
    assert rec["early"][1] >= rec["nbuckets"][1] - 1 and rec["early"][2] >= 1             # in flight before backward returned


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("kernels", ["direct", "winograd"])
def test_two_rank_sync_bn_matches_the_unsharded_reference(tmp_path, kernels, backend):
    """SURVEY 8e mode (ii), VERDICT r2 item 7: with `GradientSynchronizer(sync_bn=True)` every train-mode BatchNorm layer
    all-reduces its [sum, sum of squares, count] (and [sum g, sum g*xhat] in the backward pass), so two ranks with two
    images each reproduce the SINGLE-PROCESS reference on the unsharded 4-image batch (/root/reference
    models/dehazing/base_model.py:15-16: BatchNorm2d spans the whole batch there): outputs of both shards (2e-4), BatchNorm
    buffers (1e-5, identical on both ranks), loss, and every synchronized gradient against the float64 oracle of the global
    L1 loss with the ranks' own ReLU masks and attention arg-max positions replayed (gate: 3 x the fp32 reference's own distance + 3e-4).  Run with the direct
    kernels and with the default Winograd kernels (the replica-BN test above pins ADH_WINOGRAD=0)."""
    world = 2
    recs = _run_ranks(tmp_path, world, 33000, ADH_DDP_SYNC_BN="1", ADH_DDP_BACKEND=backend,
                      **({"ADH_WINOGRAD": "0"} if kernels == "direct" else {}))
    assert all(recs[r]["selfcheck"]["ok"] for r in range(world)), [recs[r]["selfcheck"] for r in range(world)]
    hazy, clear, _ = R.synthetic_batch(2 * world, 32, 48, seed=77)
    masks = {k: torch.cat([recs[r]["masks"][k] for r in range(world)], dim=0) for k in recs[0]["masks"]}
    # the attention blocks' arg-max positions as the ranks' kernels took them (per image: the shards concatenate)
    cbam = {k: tuple(torch.cat([recs[r]["cbam"][k][i] for r in range(world)], dim=0) for i in range(2)) for k in recs[0]["cbam"]}

    def oracle(dtype):
        sd = {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in recs[0]["sd0"].items()}
        for k, v in sd.items():
            if v.is_floating_point() and "running" not in k:
                v.requires_grad_(True)

        def run():
            out = R.high_forward(hazy.to(dtype), sd, training=True)
            # each rank's loss is the L1 mean over its shard; the synchronizer averages the gradients: the global mean
            loss = F.l1_loss(out, clear.to(dtype))
            loss.backward()
            return out.detach(), loss.detach()
        out, loss = oracle_with_masks(run, masks, cbam)
        return out, float(loss), sd
    out32, loss32, sd32 = oracle(torch.float32)
    out64, loss64, sd64 = oracle(torch.float64)
    for r in range(world):
        assert float((recs[r]["out"].double() - out64[2 * r:2 * r + 2]).abs().max()) < 2e-4, r
    assert abs(0.5 * (recs[0]["loss"] + recs[1]["loss"]) - loss64) < 1e-5
    for k, v in recs[0]["bn"].items():
        assert torch.equal(v, recs[1]["bn"][k]), k                      # one set of statistics on every rank ...
        assert float((v.double() - sd64[k]).abs().max()) < 1e-5, k      # ... the global batch's
    bad = []
    for k in recs[0]["grads"]:
        assert torch.equal(recs[0]["grads"][k], recs[1]["grads"][k]), k
        if k.startswith("decoder") and k.endswith(".0.bias"):
            continue      # ConvTranspose bias feeding train-mode BN: true gradient 0
        g64 = sd64[k].grad
        scale = max(float(g64.abs().max()), 1e-8)
        err_gpu = float((recs[0]["grads"][k].double() - g64).abs().max()) / scale
        err_ref = float((sd32[k].grad.double() - g64).abs().max()) / scale
        if not err_gpu <= 3 * err_ref + 3e-4:
            bad.append((k, err_gpu, err_ref))
    assert not bad, bad[:6]
    for k, v in recs[0]["params"].items():
        assert torch.equal(v, recs[1]["params"][k]), k
