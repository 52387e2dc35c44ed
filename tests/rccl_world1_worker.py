"""Worker of tests/test_rccl_world1.py: one process, one GPU, backend "nccl" (= RCCL on ROCm), world size 1.
Exercises exactly the calls the multi-GPU path makes: init_process_group(nccl, device_id), the reducer's
construction-time broadcast, bucketed all_reduce(async_op=True) on the side HIP stream launched from gradient hooks
during backward, finish(), the scalar exchange, barrier.  Prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29900 + os.getpid() % 500))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from sam3_lora_amd.ddp import LoRAGradReducer, allreduce_scalar_sum
    import lora_layers as L
    out = {"backend": dist.get_backend(), "world": dist.get_world_size(),
           "rccl_version": ".".join(map(str, torch.cuda.nccl.version()))}
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(256, 512), torch.nn.GELU(), torch.nn.Linear(512, 256)).to(dev)
    net[0], net[2] = L.LoRALinear(net[0], rank=16, alpha=32), L.LoRALinear(net[2], rank=16, alpha=32)
    for m in net.modules():
        if isinstance(m, torch.nn.Linear):
            m.weight.requires_grad_(False), m.bias.requires_grad_(False)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, L.LoRALayer):
                m.lora_B.normal_(0, 0.05)
    net.to(dev)
    params = [p for p in net.parameters() if p.requires_grad]
    before = [p.detach().clone() for p in params]
    red = LoRAGradReducer(params, bucket_bytes=64 << 10, run_collectives_alone=True)
    out["params_unchanged_by_broadcast"] = all(torch.equal(a, b) for a, b in zip(before, params))
    out["buckets"] = len(red.buckets)
    x = torch.randn(300, 256, device=dev)
    # reference gradients without the reducer's exchange
    red.zero_grad(arm=False)
    net(x).pow(2).mean().backward()
    want = red.flat.clone()
    launched = []
    orig = dist.all_reduce

    def spy(t, *a, **k):
        launched.append((t.numel(), torch.cuda.current_stream(dev) == red._side, bool(k.get("async_op"))))
        return orig(t, *a, **k)
    dist.all_reduce = spy
    red.zero_grad()
    net(x).pow(2).mean().backward()
    red.finish()
    dist.all_reduce = orig
    torch.cuda.synchronize()
    out["allreduce_calls"] = len(launched)
    out["all_on_side_stream_async"] = all(side and asyn for _, side, asyn in launched)
    out["elements_reduced"] = sum(n for n, _, _ in launched)
    out["flat_elements"] = red.flat.numel()          # one "used" flag per parameter (in front) + the gradient slots (ddp.py)
    out["grads_equal"] = bool(torch.equal(red.flat[red._grad0:], want[red._grad0:]))
    out["flags"] = red.flat[:len(red.params)].tolist()
    out["launch_log"] = red.launch_log
    # where the buckets' all-reduces sit relative to the END of backward on the GPU's clock: a deeper stack on
    # enough rows that the backward takes a few milliseconds; the hooks launch a bucket as soon as its last gradient is final, so every
    # bucket but the last one must START while later layers' backward kernels are still queued behind it
    deep = torch.nn.Sequential(*[torch.nn.Linear(512, 512) for _ in range(12)]).to(dev)
    for i in range(len(deep)):
        deep[i] = L.LoRALinear(deep[i], rank=16, alpha=32)
    for m in deep.modules():
        if isinstance(m, torch.nn.Linear):
            m.weight.requires_grad_(False), m.bias.requires_grad_(False)
    with torch.no_grad():
        for m in deep.modules():
            if isinstance(m, L.LoRALayer):
                m.lora_B.normal_(0, 0.05)
    deep.to(dev)                        # (the adapters are created on the CPU by the wrapper)
    dparams = [p for p in deep.parameters() if p.requires_grad]
    dred = LoRAGradReducer(dparams, bucket_bytes=128 << 10, run_collectives_alone=True)
    xd = torch.randn(40000, 512, device=dev)
    for it in range(3):                 # the first pass warms the communicator and the allocator
        dred.zero_grad()
        dred.trace = it == 2
        loss = deep(xd).pow(2).mean()
        loss.backward()
        end = torch.cuda.Event(enable_timing=True)
        end.record(torch.cuda.current_stream(dev))
        dred.finish()
        torch.cuda.synchronize()
    out["deep_buckets"] = len(dred.buckets)
    out["deep_launch_log"] = dred.launch_log
    out["deep_bucket_start_ms_after_backward_end"] = [round(end.elapsed_time(e0), 3) for (_, e0, _, _) in dred.launch_events]
    out["deep_bucket_end_ms_after_backward_end"] = [round(end.elapsed_time(e1), 3) for (_, _, e1, _) in dred.launch_events]
    nb = allreduce_scalar_sum(torch.tensor([5.0], device=dev))
    out["scalar"] = nb.item()
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
