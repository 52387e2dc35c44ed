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
    assert params[0].grad.data_ptr() == red.flat.data_ptr()
    assert red.nbytes >= 4 * (1600 + 4800)
    # set_to_none style replacement is folded back into the flat buffer
    red.zero_grad()
    params[0].grad = None
    (params[0].sum() * 5).backward()
    red.finish()
    assert torch.all(red.flat[:1600] == 5)
