"""
Data-parallel reduction of the LoRA gradients -- the only cross-GPU exchange of the path.

The reference's DDP variant wraps the whole model in ``torch.nn.parallel.DistributedDataParallel``
(sam3_lora/train/native_trainer.py:319-340) and lets it bucket every ``requires_grad`` gradient,
which for a LoRA run is just A and B (23.6 MB fp32 at r=16).  Here the exchange is
explicit and sized for xGMI (point-to-point links, ring collectives are per-link bound, so few
large messages beat many small ones):

  * all A/B gradients live in ONE flat fp32 buffer; every ``param.grad`` is a view into it, so the
    HIP backward accumulates straight into the buffer and no gather/scatter copy ever runs;
  * the buffer is cut into a few contiguous buckets in reverse registration order (gradients
    complete from the last ViT block to the first); when the last gradient of a bucket has been
    accumulated AND every bucket before it has been launched, that bucket is all-reduced (SUM) on a
    side HIP stream while backward continues.  Buckets are therefore issued in INDEX order on every
    rank whatever order the hooks fire in (torch DDP's ``next_bucket`` rule): ranks whose graphs
    differ -- the reference runs DDP with ``find_unused_parameters: true``, SURVEY section 3.4 --
    still issue identical collective sequences;
  * one "used this step" flag per parameter sits IN FRONT of the first-registered gradients, i.e. in
    the bucket that is always launched last, when the flags are complete: no separate message;
  * ``finish()`` launches what the hooks could not (in index order), makes the compute stream wait
    for the side stream and scales by 1/world_size (mean, as DDP does).  Parameters that never
    receive a gradient in a step are reduced as zeros, never skipped.

Works with backend "nccl" (= RCCL on ROCm) on GPUs and "gloo" on CPU (tests).
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist


class LoRAGradReducer:
    def __init__(self, params: Iterable[torch.nn.Parameter], process_group=None,
                 bucket_bytes: int = 16 << 20, average: bool = True, overlap: bool = True,
                 broadcast_parameters: bool = True, run_collectives_alone: bool = False,
                 comms_dtype: Optional[torch.dtype] = None):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("LoRAGradReducer: no trainable parameters")
        dev = self.params[0].device
        if any(p.device != dev for p in self.params):
            raise ValueError("LoRAGradReducer: parameters must share a device")
        self.device = dev
        self.group = process_group
        self.average = average
        self.world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # world size 1 normally skips the exchange; `run_collectives_alone` issues it anyway (a 1-rank RCCL all-reduce is
        # the identity) so that the whole side-stream path can be exercised on a single GPU
        self.run_alone = bool(run_collectives_alone) and dist.is_initialized()
        # the reference's optional gradient compression (native_trainer.py:329-340: torch's bf16 / fp16 compress hooks): a bucket travels
        # as `(bucket / world).to(comms_dtype)` and the summed result is written back into the fp32 buffer -- the division happens BEFORE
        # the exchange, in fp32, and finish() then does not scale again.  Halves 23.6 MB that already hide behind backward: off by default.
        if comms_dtype not in (None, torch.bfloat16, torch.float16):
            raise ValueError("LoRAGradReducer: comms_dtype must be None, torch.bfloat16 or torch.float16")
        self.comms_dtype = comms_dtype
        self.overlap = overlap and dev.type == "cuda"
        # flat buffer: [one "used this step" flag per parameter | 64-element aligned gradient slots in registration order].
        # The flags are summed by the same exchange: torch DDP leaves the gradient of a GLOBALLY unused parameter None, so
        # that the optimizer skips it (no weight decay, no moment update) -- see `finish`.  They sit in front of parameter
        # 0, i.e. inside the bucket that is launched LAST on every rank, by which time they are complete.
        self._flag0 = 0
        self._grad0 = (len(self.params) + 63) // 64 * 64
        offs, n = [], self._grad0
        for p in self.params:
            offs.append(n)
            n += (p.numel() + 63) // 64 * 64
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self._offs = offs
        for p, o in zip(self.params, offs):
            if p.dtype != torch.float32:
                raise ValueError("LoRAGradReducer: LoRA master parameters must be float32")
            p.grad = self.flat[o:o + p.numel()].view_as(p)
        # buckets: contiguous ranges of the flat buffer, built from the END (last params finish first)
        self.buckets = []          # (start, end, first_param_idx, last_param_idx)
        per = max(1, bucket_bytes // 4)
        self.skip_unused = True    # leave .grad = None on parameters no rank produced a gradient for (torch DDP's behaviour)
        self.fired = set()         # indices of the parameters whose gradient arrived in the armed backward
        # host staging of the used-flags (up: this rank's, before the last bucket goes; back: the reduced ones, for finish()); pinned,
        # each guarded by an event so that the asynchronous copies never race the next step's rewrite and finish() waits for exactly
        # the last bucket's reduction instead of synchronising the device
        self._flags_host = torch.zeros(len(self.params), dtype=torch.float32)
        self._flags_back = torch.zeros(len(self.params), dtype=torch.float32)
        self._flags_up_event = self._flags_back_event = None
        if dev.type == "cuda":
            self._flags_host = self._flags_host.pin_memory()
            self._flags_back = self._flags_back.pin_memory()
        hi = len(self.params)
        while hi > 0:
            lo = hi - 1
            end = offs[hi - 1] + (self.params[hi - 1].numel() + 63) // 64 * 64
            while lo > 0 and end - offs[lo - 1] <= per:
                lo -= 1
            self.buckets.append((offs[lo] if lo > 0 else 0, end, lo, hi))      # the last one carries the flags
            hi = lo
        self._bucket_of = {}
        for b, (_, _, lo, hi_) in enumerate(self.buckets):
            for i in range(lo, hi_):
                self._bucket_of[i] = b
        self._pending = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._next = 0             # buckets [0, _next) have been launched: the only one a hook may launch is _next
        self.launch_log = []       # (bucket, "hook" | "finish") of the last armed step, in launch order
        self.trace = False         # profiling aid: HIP events around every bucket's all-reduce on the side stream (bench.py)
        self.launch_events = []    # per launch of the last armed step: (bucket, start event, end event) when `trace`
        self._works = []
        self._side = torch.cuda.Stream(device=dev) if self.overlap else None
        self._armed = False
        self._hooks = [p.register_post_accumulate_grad_hook(self._make_hook(i)) for i, p in enumerate(self.params)]
        if broadcast_parameters:
            self.broadcast_parameters()

    def broadcast_parameters(self, src: int = 0):
        """Every rank starts from rank ``src``'s A/B -- what torch DDP does at construction
        (native_trainer.py:319-328).  The adapters are drawn per process from an unseeded generator
        (``kaiming_uniform_``), so without this the replicas would train apart on averaged gradients.
        One flat message through the gradient buffer's storage (it is all zeros at this point)."""
        if self.world_size <= 1:
            return
        with torch.no_grad():
            for p, o in zip(self.params, self._offs):
                self.flat[o:o + p.numel()].copy_(p.detach().reshape(-1))
            dist.broadcast(self.flat, src=src, group=self.group)
            for p, o in zip(self.params, self._offs):
                p.copy_(self.flat[o:o + p.numel()].view_as(p))
            self.flat.zero_()

    # ------------------------------------------------------------------ step protocol ----
    def zero_grad(self, arm: bool = True):
        """Zero the flat buffer (keeps every ``param.grad`` a view of it) and arm the hooks.  ``arm=False``: the first of
        several accumulated micro-batches -- nothing is exchanged until :meth:`arm` is called before the last backward
        (``model.no_sync()`` of native_trainer.py:985-991)."""
        self.flat.zero_()
        for p, o in zip(self.params, self._offs):
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + 4 * o:
                p.grad = self.flat[o:o + p.numel()].view_as(p)
        self.fired = set()
        self._armed = False
        if arm:
            self.arm()

    def arm(self):
        """Call before the backward of the LAST micro-batch (no_sync semantics for earlier ones)."""
        for b, (_, _, lo, hi) in enumerate(self.buckets):
            self._pending[b] = hi - lo
            self._launched[b] = False
        self._next = 0
        self.launch_log = []
        self.launch_events = []
        self._works = []
        self._flags_back_event = None      # (a step whose `skip_unused` was toggled between launch and finish() leaves none behind)
        self._armed = True

    def notify(self, param: torch.nn.Parameter):
        """Manual form of the autograd hook, for callers that accumulate into ``param.grad`` through
        the C-ABI directly (bench.py): declares this parameter's gradient final for the step."""
        if not hasattr(self, "_index"):
            self._index = {id(p): i for i, p in enumerate(self.params)}
        self._make_hook(self._index[id(param)])(param)

    def _make_hook(self, idx: int):
        def hook(param):
            self.fired.add(idx)        # also during un-armed (accumulating) micro-batches: the parameter is in use
            if not self._armed:
                return
            o = self._offs[idx]
            if param.grad is not None and param.grad.data_ptr() != self.flat.data_ptr() + 4 * o:
                # someone replaced .grad (e.g. zero_grad(set_to_none=True)): fold it back into the buffer
                self.flat[o:o + param.numel()].view_as(param).add_(param.grad)
                param.grad = self.flat[o:o + param.numel()].view_as(param)
            b = self._bucket_of[idx]
            self._pending[b] -= 1
            # index order, on every rank: a complete bucket waits until all earlier buckets have gone (a bucket another
            # rank completes early must not overtake one this rank has not completed -- the collectives would pair up
            # with different buffers)
            while self._next < len(self.buckets) and self._pending[self._next] == 0:
                self._launch(self._next, "hook")
        return hook

    def _launch(self, b: int, origin: str):
        assert b == self._next and not self._launched[b], "buckets are launched in index order, once"
        self._launched[b] = True
        self._next = b + 1
        self.launch_log.append((b, origin))
        if self.world_size == 1 and not self.run_alone:
            return
        if b == len(self.buckets) - 1:
            # the flags are complete: either every hook has fired (launched from a hook: all earlier buckets and this one
            # are complete) or backward is over (launched from finish)
            flags = self.flat[:len(self.params)]
            if len(self.fired) == len(self.params):
                flags.fill_(1.0)
            else:
                if self._flags_up_event is not None:        # the previous step's H2D copy of this buffer has been consumed
                    self._flags_up_event.synchronize()
                self._flags_host.zero_()
                if self.fired:
                    self._flags_host[sorted(self.fired)] = 1.0
                flags.copy_(self._flags_host, non_blocking=True)
                if self.device.type == "cuda":
                    self._flags_up_event = torch.cuda.Event()
                    self._flags_up_event.record(torch.cuda.current_stream(self.device))
        s, e, _, _ = self.buckets[b]
        chunk = self.flat[s:e]
        last = b == len(self.buckets) - 1
        # the reduced flags travel back to pinned host memory behind the last bucket's all-reduce, only when this rank will have to
        # look at them (it has locally unused parameters)
        want_flags = last and self.skip_unused and len(self.fired) < len(self.params) and self.device.type == "cuda"

        def flags_back(work):
            work.wait()                                     # stream-level: the stream this runs on waits for the reduction
            self._flags_back.copy_(self.flat[:len(self.params)], non_blocking=True)
            self._flags_back_event = torch.cuda.Event()
            self._flags_back_event.record(torch.cuda.current_stream(self.device))

        def exchange():
            if self.comms_dtype is None:
                self._works.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                return
            comp = (chunk / float(self.world_size) if self.average else chunk).to(self.comms_dtype)
            work = dist.all_reduce(comp, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            work.wait()                                     # stream-level on a GPU (the issuing stream waits); blocking with gloo
            chunk.copy_(comp)
            self._works.append(work)
        if self.overlap:
            self._side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self._side):
                if self.trace:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(self._side)
                exchange()
                if self.trace:
                    e1.record(self._side)
                    self.launch_events.append((b, e0, e1, origin))
                if want_flags:
                    flags_back(self._works[-1])
        else:
            exchange()
            if want_flags:
                flags_back(self._works[-1])

    def finish(self):
        """After backward: reduce whatever the hooks could not launch (buckets holding a locally unused parameter, and
        everything behind them), in index order; wait; average."""
        while self._next < len(self.buckets):
            self._launch(self._next, "finish")
        exchange = self.world_size > 1 or self.run_alone
        locally_unused = [i for i in range(len(self.params)) if i not in self.fired] if self.skip_unused else []
        flags = self.flat[:len(self.params)]
        for w in self._works:
            w.wait()
        if self.overlap:
            torch.cuda.current_stream(self.device).wait_stream(self._side)
        if self.average and self.world_size > 1 and self.comms_dtype is None:        # (compressed buckets were divided before they travelled)
            self.flat.mul_(1.0 / self.world_size)
        if locally_unused:
            # a parameter unused on EVERY rank keeps .grad = None (the optimizer then skips it, as after the reference's
            # zero_grad()); one unused only here was reduced as zeros + the other ranks' gradients.  Only a rank that has
            # locally unused parameters needs to look (a globally unused one is locally unused everywhere).
            if not exchange:
                globally = locally_unused
            elif self._flags_back_event is not None:        # wait for the last bucket's reduction + its small D2H copy, nothing else
                self._flags_back_event.synchronize()
                self._flags_back_event = None
                globally = [i for i in locally_unused if float(self._flags_back[i]) == 0.0]
            else:                                           # CPU tensors (gloo tests): the reduction has been waited for above
                globally = [i for i, v in zip(locally_unused, flags[locally_unused].tolist()) if v == 0.0]
            for i in globally:
                self.params[i].grad = None
        self._works = []
        self._armed = False

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * 4


def allreduce_scalar_sum(value: torch.Tensor, process_group=None) -> torch.Tensor:
    """``num_boxes`` style scalar exchange (sam3/train/loss/sam3_loss.py:74-76)."""
    if dist.is_initialized() and dist.get_world_size(process_group) > 1:
        dist.all_reduce(value, op=dist.ReduceOp.SUM, group=process_group)
    return value
