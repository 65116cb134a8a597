"""World-size-2 gloo test (CPU) of the data-parallel plumbing bench.py and the wrapper rely on: disjoint batch
shards that cover the batch, max-over-ranks timing, rank-ordered pose gather.  The per-rank 'model' here is the
oracle's pose head on a shard (the HIP path needs a GPU; its sharding logic is what is under test)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from egotap_amd import parallel as P


def test_shard_bounds_cover_and_balance():
    for total in (0, 1, 7, 256, 8192):
        for world in (1, 2, 3, 8):
            spans = [P.shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        P.shard_bounds(4, 2, 2)


def _worker(rank, world, port, total, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    r, w = P.init_from_env("gloo")
    assert (r, w) == (rank, world)
    lo, hi = P.shard_bounds(total, rank, world)
    g = torch.Generator().manual_seed(1234)
    full = torch.rand(total, 16, 3, generator=g)              # every rank can regenerate the whole batch
    mine = full[lo:hi] * 2.0 + 1.0                              # "forward" of this rank's shard
    P.barrier()
    slowest = P.max_over_ranks(0.5 + rank)                      # rank 1 is slower
    counts = [P.shard_bounds(total, k, world)[1] - P.shard_bounds(total, k, world)[0] for k in range(world)]
    allp = P.gather_poses(mine, counts)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), np.concatenate([[slowest], allp.reshape(-1).numpy()]))
    torch.distributed.destroy_process_group()


def test_two_rank_gloo(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    total = 7                                                   # ragged: shards of 4 and 3
    mp.spawn(_worker, args=(2, port, total, str(tmp_path)), nprocs=2, join=True)
    g = torch.Generator().manual_seed(1234)
    ref = (torch.rand(total, 16, 3, generator=g) * 2.0 + 1.0).reshape(-1).numpy()
    for rank in range(2):
        got = np.load(tmp_path / f"r{rank}.npy")
        assert got[0] == 1.5                                    # max over ranks
        np.testing.assert_array_equal(got[1:], ref)             # same gathered batch on every rank, rank order


def _packing_allreduce(params, bucket_bytes: int = 64 << 20):
    """Test-side reference for GradReducer (it lived in egotap_amd.parallel until both wrappers moved to the arena reducer): sum every parameter's .grad over the ranks and divide by the
    world size, bucketed so a few large collectives run instead of one per tensor (RCCL over xGMI is per-link bound:
    large messages), issued asynchronously so bucket k+1 is packed while bucket k is on the wire.  Parameters whose
    .grad is None (cls_token, pooler: never used, so never differentiated) are skipped identically on every rank.
    No-op for one rank."""
    dist = torch.distributed
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    world = dist.get_world_size()
    grads = [p.grad for p in params if p.grad is not None]
    buckets, cur, size = [], [], 0
    for g in grads:
        cur.append(g)
        size += g.numel() * g.element_size()
        if size >= bucket_bytes:
            buckets.append(cur)
            cur, size = [], 0
    if cur:
        buckets.append(cur)
    pending = []
    for b in buckets:
        flat = torch.cat([g.reshape(-1) for g in b])
        pending.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True), flat, b))
    for work, flat, b in pending:
        work.wait()
        flat.div_(world)
        off = 0
        for g in b:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n
    return len(buckets)


def _grad_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    P.init_from_env("gloo")
    g = torch.Generator().manual_seed(7)
    shapes = [(300, 200), (17,), (64, 3, 3), (5,), (1000,)]
    params = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
    for i, prm in enumerate(params):
        base = torch.rand(prm.shape, generator=g)
        prm.grad = None if i == 3 else base * (rank + 1)      # parameter 3 never gets a gradient (like cls_token / pooler)
    n = _packing_allreduce(params, bucket_bytes=100_000)   # small buckets: several collectives in flight
    assert n >= 2 and params[3].grad is None
    np.save(os.path.join(out_dir, f"g{rank}.npy"), torch.cat([q.grad.reshape(-1) for q in params if q.grad is not None]).numpy())
    torch.distributed.destroy_process_group()


def test_gradient_allreduce_two_ranks(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_grad_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g = torch.Generator().manual_seed(7)
    ref = []
    for i, shp in enumerate([(300, 200), (17,), (64, 3, 3), (5,), (1000,)]):
        base = torch.rand(shp, generator=g)
        if i != 3:
            ref.append((base * 1.5).reshape(-1))               # mean of 1x and 2x
    ref = torch.cat(ref).numpy()
    for rank in range(2):
        np.testing.assert_allclose(np.load(tmp_path / f"g{rank}.npy"), ref, rtol=1e-6)


def _reducer_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    P.init_from_env("gloo")
    g = torch.Generator().manual_seed(11)
    base = torch.rand(10_000, generator=g)
    arena = base * (rank + 1)                                   # this rank's gradients, already in arena order
    bounds = [0, 3000, 3000, 7168, 10_000]                      # four buckets, one of them empty
    red = P.GradReducer()
    assert red.active()
    red.begin(arena)
    for lo, hi in zip(bounds, bounds[1:]):                      # the training Function calls this as each bucket becomes final
        red.bucket_ready(lo, hi)
    assert len(red.pending) == 3
    n = red.finish()
    assert n == 3 and red.last_buckets == 3 and red.last_bytes == 4 * 10_000 and red.steps == 1 and red.pending == []
    assert red.read_exposed_ms() == 0.0                         # CPU tensors: nothing to time
    # the same gradients through the generic packing reducer (the test-side packing reducer above)
    prm = torch.nn.Parameter(torch.zeros(10_000))
    prm.grad = base * (rank + 1)
    _packing_allreduce([prm], bucket_bytes=8_000)
    np.save(os.path.join(out_dir, f"a{rank}.npy"), np.stack([arena.numpy(), prm.grad.numpy()]))
    torch.distributed.destroy_process_group()


def test_grad_reducer_two_ranks(tmp_path):
    """parallel.GradReducer (the in-backward bucketed all-reduce on the flat gradient arena): after finish() every rank holds the
    MEAN of the per-rank arenas, bucket by bucket in place, identical to the packing reference reducer"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_reducer_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g = torch.Generator().manual_seed(11)
    ref = (torch.rand(10_000, generator=g) * 1.5).numpy()
    for rank in range(2):
        got = np.load(tmp_path / f"a{rank}.npy")
        np.testing.assert_allclose(got[0], ref, rtol=1e-6)
        np.testing.assert_array_equal(got[0], got[1])


def test_grad_reducer_is_a_noop_without_a_group():
    red = P.GradReducer()
    assert not red.active()
    arena = torch.ones(8)
    red.begin(arena)
    red.bucket_ready(0, 8)
    assert red.finish() == 0 and red.pending == [] and torch.equal(arena, torch.ones(8))


def _stage1_worker(rank, world, port, out_dir):
    """the stage-1 estimator's gradient arena exactly as hm_training.HmTrainFn.backward drives it: arena layout and bucket bounds from
    the real module's parameter list (hm_training._grad_arena), bucket_ready in backward-completion order, finish()"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    from egotap_amd import hm_training as HT, networks
    from egotap_amd.options import preset_defaults
    opt = preset_defaults("UnrealEgo")
    opt.num_rot_heatmap = 0
    net = networks.HeatMap_UnrealEgo_Shared(opt, "resnet18", 2)            # parameters only: no kernel runs on the CPU
    Pm = dict(HT._param_items(net))
    G = HT._grad_arena(net, Pm)
    ga = net._hm_grad_arena
    g = torch.Generator().manual_seed(5)
    for k in Pm:                                                           # this rank's gradients: base * (rank + 1)
        G[k].copy_(torch.rand(G[k].shape, generator=g) * (rank + 1))
    red = HT._reducer(net)
    red.begin(ga["flat"])
    b = ga["bounds"]
    for lo, hi in zip(b, b[1:]):
        red.bucket_ready(lo, hi)
    n = red.finish()
    sums = {k: float(G[k].double().sum()) for k in Pm}
    torch.save(dict(n=n, bounds=b, bytes=red.last_bytes, sums=sums, total=ga["flat"].numel(),
                    first=list(Pm.keys())[0], order_head=[k for k, _ in sorted(ga["offs"].items(), key=lambda kv: kv[1])][:3]), os.path.join(out_dir, f"s{rank}.pt"))
    torch.distributed.destroy_process_group()


def test_stage1_estimator_arena_buckets_two_ranks(tmp_path):
    """the stage-1 wrapper averages its gradients with the in-backward arena reducer too (round 4; it packed with torch.cat before): the
    estimator's arena is laid out in backward-completion order -- decoder, layer4, the rest of the backbone -- and its three buckets
    leave every rank with the MEAN of the per-rank gradients, in place"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_stage1_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f"s{r}.pt") for r in range(2))
    assert r0["n"] == 3 and r0["bounds"][0] == 0 and r0["bounds"][-1] == r0["total"] and r0["bytes"] == 4 * r0["total"]
    assert sorted(r0["bounds"]) == r0["bounds"] and len(set(r0["bounds"])) == 4
    assert all(k.startswith("after_backbone.") for k in r0["order_head"])              # the decoder's gradients come first in the arena
    assert r0["bounds"][1] > 0.6 * r0["total"]                                         # ... and are two thirds of it
    from egotap_amd import hm_training as HT, networks
    from egotap_amd.options import preset_defaults
    opt = preset_defaults("UnrealEgo")
    opt.num_rot_heatmap = 0
    net = networks.HeatMap_UnrealEgo_Shared(opt, "resnet18", 2)
    g = torch.Generator().manual_seed(5)
    for k, prm in HT._param_items(net):
        want = float((torch.rand(prm.shape, generator=g) * 1.5).double().sum())        # mean of base and 2 * base
        for r in (r0, r1):
            np.testing.assert_allclose(r["sums"][k], want, rtol=1e-5, err_msg=k)
