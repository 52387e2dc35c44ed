"""
The training loop around the adapter path -- the caller side of SURVEY section 8 row a14 (reference
``train_sam3_lora_native.py``: ``SAM3TrainerNative.__init__`` :689-793, ``train`` :795-1046, ``main``
:1049-1060).

What is restated here is the part of that script that belongs to the LoRA path and its boundary:

  * the YAML surface: exactly the keys the reference reads (ten mandatory ``lora.*`` keys,
    ``training.{data_dir,batch_size,learning_rate,weight_decay,num_epochs}``, ``output.output_dir``); a missing
    key is a ``KeyError`` as it is there, every other key is accepted and ignored as it is there;
  * adapter injection, AdamW over the ``requires_grad`` tensors only, the matcher / loss constants of
    :743-793, the per-step order forward -> back_convert -> match every step and aux output -> loss ->
    zero_grad / backward / step, the epoch loop with validation, and the artefacts: ``last_lora_weights.pt``,
    ``best_lora_weights.pt`` (copy of last when there is no validation split) and one JSON line per epoch in
    ``val_stats.json``;
  * what the reference does not have on this script (it trains on one GPU): launched under torchrun with
    WORLD_SIZE > 1 the A/B gradients go through :class:`sam3_lora_amd.ddp.LoRAGradReducer`, the loss is
    normalised over ranks ("global") and rank 0 writes the artefacts.

The model and the data come from two builders,

    model_builder(config, device) -> nn.Module       model(input_batch) -> per-stage outputs; model.back_convert(t)
    data_builder(config, split)   -> sized iterable  of batches ({"input": batch} or batch), or None for no split

which default to this library's restatement of the SAM3 image model (:func:`default_model_builder`,
``sam3_image.build_sam3_image_model``; random seeded initialisation unless ``model.checkpoint_path`` names a
``sam3.pt``) and of the COCO pipeline (:func:`default_data_builder`: ``COCOSegmentDataset`` + ``collate_fn_api`` over
``training.data_dir``, or the synthetic samples of SURVEY section 8(d) when ``data_dir`` is ``synthetic[:N]``), sharded
across ranks with DistributedSampler semantics.  ``--model-builder / --data-builder module:function`` (or
``SAM3_LORA_MODEL_BUILDER`` / ``SAM3_LORA_DATA_BUILDER``) substitute others.

Settings that exist only on this engine live under an optional ``engine:`` section of the YAML (absent = reference
behaviour) and as command-line switches: ``bf16_frozen`` (frozen tensors in bf16, A/B fp32 masters),
``act_checkpoint`` (keep | auto | on | off), ``match_once`` (skip the model-internal matching that the loop repeats),
``loader_workers`` / ``loader_prefetch`` (samples built by a thread pool and batches copied to the GPU ahead of the step;
0 = the reference's ``num_workers=0``), ``grad_accumulation_steps``, ``direct_grad``, ``fp8_frozen``.
"""
from __future__ import annotations

import importlib
import json
import os
import shutil
from pathlib import Path
from typing import Any, Callable, Dict, Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist
import yaml
from torch.optim import AdamW

from .ddp import LoRAGradReducer
from .lora_layers import LoRAConfig, apply_lora_to_model, count_parameters, save_lora_weights
from .losses import CORE_LOSS_KEY, Boxes, BinaryOneToManyMatcher, IABCEMdetr, Masks, Sam3LossWrapper
from .matcher import BinaryHungarianMatcherV2

__all__ = ["SAM3TrainerNative", "load_config", "lora_config_from", "build_criterion", "resolve_builder",
           "default_model_builder", "default_data_builder",
           "match_all_steps", "move_to_device", "DEFAULT_CONFIG"]

DEFAULT_CONFIG = "configs/full_lora_config.yaml"       # train_sam3_lora_native.py:1051
LORA_KEYS = ("rank", "alpha", "dropout", "target_modules", "apply_to_vision_encoder", "apply_to_text_encoder",
             "apply_to_geometry_encoder", "apply_to_detr_encoder", "apply_to_detr_decoder", "apply_to_mask_decoder")


def load_config(path: str) -> Dict[str, Any]:
    with open(path, "r") as f:
        return yaml.safe_load(f)


def lora_config_from(config: Dict[str, Any]) -> LoRAConfig:
    """All ten keys are mandatory (:713-725): a config without one raises KeyError naming it."""
    section = config["lora"]
    return LoRAConfig(**{k: section[k] for k in LORA_KEYS})


def build_criterion(normalization: str = "local"):
    """Matcher and loss stack with the constants the native script hard-codes (:743-793)."""
    matcher = BinaryHungarianMatcherV2(cost_class=2.0, cost_bbox=5.0, cost_giou=2.0, focal=True)
    losses = [
        Boxes(weight_dict={"loss_bbox": 5.0, "loss_giou": 2.0}),
        IABCEMdetr(pos_weight=10.0, weight_dict={"loss_ce": 20.0, "presence_loss": 20.0}, pos_focal=False,
                   alpha=0.25, gamma=2, use_presence=True, pad_n_queries=200),
        Masks(weight_dict={"loss_mask": 200.0, "loss_dice": 10.0}, focal_alpha=0.25, focal_gamma=2.0,
              compute_aux=False),
    ]
    wrapper = Sam3LossWrapper(loss_fns_find=losses, matcher=matcher,
                              o2m_matcher=BinaryOneToManyMatcher(alpha=0.3, threshold=0.4, topk=4), o2m_weight=2.0,
                              use_o2m_matcher_on_o2m_aux=False, normalization=normalization)
    return matcher, wrapper


def default_model_builder(config: Dict[str, Any], device) -> torch.nn.Module:
    """``build_sam3_image_model(device, compile=False, bpe_path="sam3/assets/...", eval_mode=False)`` of :705-711, minus
    the Hugging Face download: ``model.checkpoint_path`` (optional) names a local ``sam3.pt``, otherwise the weights are
    a seeded random initialisation (``model.seed``, default 0)."""
    from .sam3_image import build_sam3_image_model
    mcfg = config.get("model") or {}
    eng = config.get("engine") or {}
    return build_sam3_image_model(bpe_path=mcfg.get("bpe_path"), device=str(device), eval_mode=False,
                                  checkpoint_path=mcfg.get("checkpoint_path"), load_from_HF=False,
                                  match_in_forward=not eng.get("match_once", False),
                                  seed=int(mcfg.get("seed", 0)))


def default_data_builder(config: Dict[str, Any], split: str):
    """The reference's loaders (:799-846): ``COCOSegmentDataset(data_dir, split)`` batched by
    ``collate_fn_api(dict_key="input", with_seg_masks=True)``, shuffled for "train" only, ``num_workers=0`` -- here
    behind a rank-sharding loader (DistributedSampler semantics; world size 1 = the reference's DataLoader).
    ``data_dir: synthetic[:N]`` selects N (default 64; validation N/4) synthetic samples instead of files."""
    from .sam3_data import COCOSegmentDataset, ShardedLoader, SyntheticSegmentDataset, collate_fn_api
    data_dir = str(config["training"]["data_dir"])
    if data_dir.startswith("synthetic"):
        n = int(data_dir.split(":", 1)[1]) if ":" in data_dir else 64
        dataset = SyntheticSegmentDataset(n if split == "train" else max(n // 4, 1), split=split)
    else:
        dataset = COCOSegmentDataset(data_dir=data_dir, split=split)
    eng = config.get("engine") or {}
    workers = int(eng.get("loader_workers", 0))             # 0 = the reference's num_workers=0 (:831)
    device = None
    if workers > 0 and torch.cuda.is_available() and eng.get("loader_to_device", True):
        device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    return ShardedLoader(dataset, config["training"]["batch_size"],
                         lambda samples: collate_fn_api(samples, dict_key="input", with_seg_masks=True),
                         shuffle=(split == "train"), rank=int(os.environ.get("RANK", "0")),
                         world=int(os.environ.get("WORLD_SIZE", "1")),
                         seed=int((config.get("training") or {}).get("seed", 0) or 0),
                         num_workers=workers, prefetch=int(eng.get("loader_prefetch", 2)), device=device)


def resolve_builder(spec: Optional[str], env: str, what: str) -> Callable:
    spec = spec or os.environ.get(env)
    if not spec:
        return default_model_builder if what == "model" else default_data_builder
    mod, _, fn = spec.partition(":")
    if not fn:
        raise ValueError(f"{what} builder must be 'module:function', got {spec!r}")
    return getattr(importlib.import_module(mod), fn)


def move_to_device(obj, device):
    """Tensors inside lists / tuples / dicts / dataclasses (:864-879); dataclasses are updated in place."""
    if isinstance(obj, torch.Tensor):
        return obj.to(device)
    if isinstance(obj, list):
        return [move_to_device(x, device) for x in obj]
    if isinstance(obj, tuple):
        return tuple(move_to_device(x, device) for x in obj)
    if isinstance(obj, dict):
        return {k: move_to_device(v, device) for k, v in obj.items()}
    if hasattr(obj, "__dataclass_fields__"):
        for name in obj.__dataclass_fields__:
            setattr(obj, name, move_to_device(getattr(obj, name), device))
    return obj


def _steps(stage) -> List[Dict]:
    """A stage is one output dict or the list of its interactive steps (SAM3Output's ALL_STEPS_PER_STAGE view)."""
    return [stage] if isinstance(stage, dict) else list(stage)


def match_all_steps(matcher, stage_outputs: Sequence, stage_targets: Sequence[Dict]) -> None:
    """``outputs["indices"]`` for every step of every stage and each of its aux outputs (:914-927).  ``matcher`` is the
    Hungarian matcher -- or the loss wrapper, which then also prepares the one-to-many indices its ``compute_loss`` would
    otherwise match one by one (``Sam3LossWrapper.launch_matching``): one batched cost, one device->host copy.  A handle
    left by the model's forward (``Sam3Image.set_prefetch_matcher``) is collected instead of starting over."""
    whole = hasattr(matcher, "launch_matching")
    # every step of every stage, whatever view the SAM3Output currently iterates in (the reference switches it to
    # ALL_STEPS_PER_STAGE before matching, :914-918; the loss wrapper then reads every step's "indices")
    stage_outputs = getattr(stage_outputs, "output", stage_outputs)
    for stage, targets in zip(stage_outputs, stage_targets):
        for outputs in _steps(stage):
            pending = outputs.pop("_match_handle", None) if isinstance(outputs, dict) else None
            aux_outputs = list(outputs.get("aux_outputs", ()))
            mine = pending is not None and pending[0] is matcher and pending[1] is not None
            if whole:
                handle = pending[1] if mine else matcher.launch_matching(outputs, targets)
                if handle is not None:
                    matcher.collect_matching(handle, outputs)
                    continue
                hungarian, mine = matcher.matcher, False        # validity masks present: the plain per-output path
            else:
                hungarian = matcher
            if mine and pending[1]["L"] == 1 + len(aux_outputs):
                found = hungarian.collect(pending[1])           # started inside the model's forward
            else:
                found = hungarian.collect(hungarian.launch([outputs] + aux_outputs, targets))
            outputs["indices"] = found[0]
            for aux, idx in zip(aux_outputs, found[1:]):
                aux["indices"] = idx


def make_adamw(params, lr: float, weight_decay: float) -> AdamW:
    """``AdamW`` with the reference's hyper-parameters (train_sam3_lora_native.py:736-740: default betas / eps) -- on a GPU as torch's
    FUSED implementation: one multi-tensor kernel for the 128 A / B tensors instead of the ~12 ``_foreach_*`` passes of the default,
    whose HOST side (list building, dispatch) left the GPU idle for 3.6 ms of every step right before ``k_pack``
    (profiles/r05q_fullstep_idle.txt: the longest wait of the step).  Same update rule; parameters whose ``.grad`` is None are skipped
    by both."""
    params = list(params)
    fused = bool(params) and all(p.is_cuda and p.dtype == torch.float32 for p in params)
    if fused:
        try:
            return AdamW(params, lr=lr, weight_decay=weight_decay, fused=True)
        except (RuntimeError, TypeError):       # a torch build without the fused kernel for this device
            pass
    return AdamW(params, lr=lr, weight_decay=weight_decay)


class NonFiniteStepGuard:
    """fp8 frozen-W mode: an optimizer step whose A / B gradients are not all finite is SKIPPED -- what the reference's fp16 path gets from
    ``GradScaler`` (native_trainer.py:902-903, 1014, 1155: ``scaler.step`` skips a step with non-finite gradients).  Delayed scaling quantises a tensor with the range of its predecessor; about one training
    sequence in 400 of the soak (tools/fp8_soak.py, profiles/r06a*_fp8_soak_*) produced a backward pass with astronomically large or
    non-finite gradients, never in the bf16 mode.  One fused check kernel over the gradients
    (``torch._amp_foreach_non_finite_check_and_unscale_`` with scale 1) raises a device flag that torch's fused AdamW takes as ``found_inf``:
    the update, the moments and the step counts stay untouched -- no host synchronisation.  ``skipped`` counts on the device."""

    def __init__(self, optimizer, device):
        self.optimizer = optimizer
        self.found = torch.zeros((), device=device)
        self.one = torch.ones((), device=device)
        self.skipped = torch.zeros((), device=device)
        self.fused = any(g.get("fused") for g in optimizer.param_groups)

    def step(self) -> None:
        grads = [p.grad for g in self.optimizer.param_groups for p in g["params"] if p.grad is not None]
        self.found.zero_()
        if grads:
            torch._amp_foreach_non_finite_check_and_unscale_(grads, self.found, self.one)
        self.skipped += self.found
        if self.fused:
            self.optimizer.grad_scale, self.optimizer.found_inf = None, self.found
            try:
                self.optimizer.step()
            finally:
                self.optimizer.found_inf = None
        elif not bool(self.found.item()):       # the unfused optimizers have no such input: one host round trip
            self.optimizer.step()


class SAM3TrainerNative:
    def __init__(self, config_path: str, model_builder: Optional[Callable] = None,
                 data_builder: Optional[Callable] = None, bf16_frozen: Optional[bool] = None,
                 act_checkpoint: Optional[str] = None):
        self.config_path = config_path
        self.config = load_config(config_path)
        engine = self.config.get("engine") or {}
        bf16_frozen = bool(engine.get("bf16_frozen", False)) if bf16_frozen is None else bf16_frozen
        act_checkpoint = str(engine.get("act_checkpoint", "keep")) if act_checkpoint is None else act_checkpoint
        self.world_size = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if torch.cuda.is_available():
            self.device = torch.device("cuda", local_rank % torch.cuda.device_count())
            torch.cuda.set_device(self.device)
        else:
            self.device = torch.device("cpu")
        if self.world_size > 1 and not dist.is_initialized():
            backend = os.environ.get("SAM3_LORA_DIST_BACKEND") or ("nccl" if self.device.type == "cuda" else "gloo")
            dist.init_process_group(backend)

        model_builder = model_builder or resolve_builder(None, "SAM3_LORA_MODEL_BUILDER", "model")
        self.data_builder = data_builder or resolve_builder(None, "SAM3_LORA_DATA_BUILDER", "data")
        self._say("Building SAM3 model...")
        self.model = model_builder(self.config, self.device)
        self._say("Applying LoRA...")
        self.model = apply_lora_to_model(self.model, lora_config_from(self.config))
        stats = count_parameters(self.model)
        self._say(f"Trainable params: {stats['trainable_parameters']:,} ({stats['trainable_percentage']:.2f}%)")
        self.model.to(self.device)
        if bf16_frozen:         # MI355X layout: frozen tensors bf16, A/B fp32 masters (not a reference behaviour)
            from .vit import to_training_layout
            to_training_layout(self.model)
        if engine.get("fp8_frozen", False):     # fp8 frozen-W base GEMMs (BASELINE configs[4]); needs the bf16 layout
            if not bf16_frozen:
                raise ValueError("engine.fp8_frozen requires engine.bf16_frozen (fp8 operands are cut from bf16 tensors)")
            from .fp8 import enable_fp8_frozen
            enable_fp8_frozen(True)
        if act_checkpoint != "keep":   # "auto" | "on" | "off" for this library's ViT trunk (vit.set_activation_checkpointing)
            from .vit import set_activation_checkpointing
            mode = {"on": True, "off": False}.get(act_checkpoint, "auto")
            used = set_activation_checkpointing(self.model, mode, batch=int(self.config["training"]["batch_size"]))
            self._say(f"Activation checkpointing of the ViT trunk: {'on' if used else 'off'} ({act_checkpoint})")

        trainable = [p for p in self.model.parameters() if p.requires_grad]
        self.optimizer = make_adamw(trainable, lr=float(self.config["training"]["learning_rate"]),
                                    weight_decay=self.config["training"]["weight_decay"])
        self.trainable = trainable
        # engine.comms_dtype: "bf16" / "fp16" = the reference's optional gradient compression (native_trainer.py:329-340); default fp32
        comms = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp16": torch.float16, "float16": torch.float16}.get(
            str(engine.get("comms_dtype") or "").lower())
        self.reducer = LoRAGradReducer(trainable, comms_dtype=comms) if self.world_size > 1 else None
        # the adapters' backward adds straight into param.grad (the reducer's flat buffer under data parallelism)
        # instead of handing fresh gradient tensors to autograd: engine.direct_grad (default on).  The switch is scoped to
        # this trainer's own loss.backward() calls (functional.direct_grad_accumulation) -- nothing else in the process
        # sees zero placeholder gradients.
        self.direct_grad = bool(engine.get("direct_grad", True)) and self.device.type == "cuda"
        # engine.grad_accumulation_steps (default 1 = the reference CLI, which ignores training.gradient_accumulation_steps,
        # SURVEY F6): micro-batches whose gradients are summed before one exchange + one optimizer step
        # (sam3_lora/train/native_trainer.py:985-991: model.no_sync() for all but the last micro-batch)
        self.accum_steps = max(1, int(engine.get("grad_accumulation_steps", 1)))
        self._fired = set()
        if self.reducer is None:        # which parameters received a gradient this step (see _drop_unused_grads)
            self._fired_hooks = [p.register_post_accumulate_grad_hook(lambda q, i=i: self._fired.add(i))
                                 for i, p in enumerate(trainable)]
        self.matcher, self.loss_wrapper = build_criterion("global" if self.world_size > 1 else "local")
        # engine.match_once: the loop's matching starts inside the forward, right after the decoder, and its host part
        # overlaps with the mask head's device work (Sam3Image.set_prefetch_matcher); same indices either way
        if (self.config.get("engine") or {}).get("match_once", False) and hasattr(self.model, "set_prefetch_matcher"):
            self.model.set_prefetch_matcher(self.loss_wrapper)

    # ------------------------------------------------------------------------------------------------
    def _say(self, msg: str) -> None:
        if self.rank == 0:
            print(msg, flush=True)

    def _loss(self, batch) -> torch.Tensor:
        input_batch = batch["input"] if isinstance(batch, dict) and "input" in batch else batch
        input_batch = move_to_device(input_batch, self.device)
        outputs = self.model(input_batch)
        targets = [self.model.back_convert(t) for t in input_batch.find_targets]
        targets = [move_to_device(t, self.device) for t in targets]
        match_all_steps(self.loss_wrapper, outputs, targets)
        return self.loss_wrapper(outputs, targets)[CORE_LOSS_KEY]

    def _backward(self, loss: torch.Tensor) -> None:
        if self.direct_grad:
            from .functional import direct_grad_accumulation
            with direct_grad_accumulation(True):
                loss.backward()
        else:
            loss.backward()

    def _zero_grad(self) -> None:
        """``optimizer.zero_grad()`` of :937 with the tensors kept where they exist (the kernels accumulate into them); a
        gradient that is None stays None."""
        self._fired = set()
        self.optimizer.zero_grad(set_to_none=False)
        if self.direct_grad:
            for p in self.trainable:
                if p.grad is None:                  # first step, or dropped as unused last step
                    p.grad = torch.zeros_like(p)

    def _drop_unused_grads(self) -> None:
        """The reference's ``zero_grad()`` sets gradients to None, so an adapter that took no part in a step (e.g. the
        geometry encoder's when a batch has no box prompts) is skipped by AdamW -- no weight decay, no moment update.
        Here the tensors are kept and zeroed instead; to keep the same trajectory, a parameter no gradient arrived for
        gets ``.grad = None`` before the optimizer step."""
        for i, p in enumerate(self.trainable):
            if i not in self._fired:
                p.grad = None

    def train_step(self, batch) -> float:
        """One optimizer step; ``batch`` is one batch or (``engine.grad_accumulation_steps`` > 1) a list of micro-batches."""
        micro = list(batch) if isinstance(batch, (list, tuple)) else [batch]
        total = 0.0
        for k, mb in enumerate(micro):
            last = k == len(micro) - 1
            loss = self._loss(mb)
            if len(micro) > 1:
                loss = loss / len(micro)
            if self.reducer is not None:
                if k == 0:
                    self.reducer.zero_grad(arm=last)
                elif last:
                    self.reducer.arm()              # earlier micro-batches: no exchange (no_sync)
                self._backward(loss)
            else:
                if k == 0:
                    self._zero_grad()
                self._backward(loss)
            total += float(loss.item())
        if self.reducer is not None:
            self.reducer.finish()                   # leaves .grad = None on globally unused parameters
        else:
            self._drop_unused_grads()
        from . import fp8
        if fp8.fp8_enabled() and self.device.type == "cuda":
            if getattr(self, "_nonfinite_guard", None) is None or self._nonfinite_guard.optimizer is not self.optimizer:
                self._nonfinite_guard = NonFiniteStepGuard(self.optimizer, self.device)
            self._nonfinite_guard.step()            # a step with non-finite gradients is skipped (counted in .skipped)
        else:
            self.optimizer.step()
        if self.device.type == "cuda":      # A / B just changed: refresh every adapter's operand images in one batch
            from .functional import repack_adapters
            repack_adapters(self.model)
        return total

    @torch.no_grad()
    def validate(self, loader: Iterable) -> float:
        self.model.eval()
        losses = [self._loss(b).item() for b in loader]
        self.model.train()
        mean = torch.tensor([sum(losses), float(len(losses))], dtype=torch.float64, device=self.device)
        if self.world_size > 1:
            dist.all_reduce(mean)
        return (mean[0] / mean[1]).item()

    def _save(self, path: Path) -> None:
        if self.rank == 0:
            save_lora_weights(self.model, str(path))

    def train(self) -> Dict[str, Any]:
        cfg = self.config
        data_dir = cfg["training"]["data_dir"]
        self._say(f"\nLoading training data from {data_dir}...")
        train_loader = self.data_builder(cfg, "train")
        try:
            val_loader = self.data_builder(cfg, "valid")
            if val_loader is not None and len(val_loader) == 0:
                val_loader = None
        except Exception as e:                      # a missing split is not an error (:805-816)
            self._say(f"Could not load validation data: {e}")
            val_loader = None

        epochs = cfg["training"]["num_epochs"]
        out_dir = Path(cfg["output"]["output_dir"])
        if self.rank == 0:
            out_dir.mkdir(parents=True, exist_ok=True)
        self.model.train()
        best = float("inf")
        history = []
        self._say(f"Starting training for {epochs} epochs...")
        for epoch in range(epochs):
            if hasattr(train_loader, "set_epoch"):
                train_loader.set_epoch(epoch)
            if self.accum_steps > 1:
                it, losses = iter(train_loader), []
                while True:
                    group = [b for _, b in zip(range(self.accum_steps), it)]
                    if not group:
                        break
                    losses.append(self.train_step(group))
            else:
                losses = [self.train_step(b) for b in train_loader]
            avg_train = sum(losses) / len(losses) if losses else 0.0
            record = {"epoch": epoch + 1, "train_loss": avg_train}
            self._save(out_dir / "last_lora_weights.pt")
            if val_loader is not None:
                avg_val = self.validate(val_loader)
                record["val_loss"] = avg_val
                self._say(f"\nEpoch {epoch + 1}/{epochs} - Train Loss: {avg_train:.6f}, Val Loss: {avg_val:.6f}")
                if avg_val < best:
                    best = avg_val
                    self._save(out_dir / "best_lora_weights.pt")
                    self._say(f"✓ New best model saved (val_loss: {avg_val:.6f})")
                if self.rank == 0:
                    with open(out_dir / "val_stats.json", "a") as f:
                        f.write(json.dumps(record) + "\n")
            history.append(record)
        if val_loader is None and self.rank == 0 and (out_dir / "last_lora_weights.pt").exists():
            shutil.copy(out_dir / "last_lora_weights.pt", out_dir / "best_lora_weights.pt")
        if self.reducer is not None:
            dist.barrier()
        self._say(f"\nTraining complete. Models saved to {out_dir}: best_lora_weights.pt, last_lora_weights.pt")
        return {"history": history, "best_val_loss": best if val_loader is not None else None}


def main(argv: Optional[Sequence[str]] = None) -> None:
    import argparse
    parser = argparse.ArgumentParser(description="Train SAM3 with LoRA (MI355X adapter path)")
    parser.add_argument("--config", type=str, default=DEFAULT_CONFIG, help="Path to YAML configuration file")
    parser.add_argument("--model-builder", type=str, default=None, help="module:function -> nn.Module (see trainer.py)")
    parser.add_argument("--data-builder", type=str, default=None, help="module:function -> batches for a split")
    parser.add_argument("--bf16-frozen", action="store_true", default=None,
                        help="keep frozen tensors in bf16 (A/B stay fp32); default: engine.bf16_frozen of the YAML")
    parser.add_argument("--act-checkpoint", choices=["keep", "auto", "on", "off"], default=None,
                        help="per-block recompute of this library's ViT trunk; auto = off when the activations fit in HBM")
    args = parser.parse_args(argv)
    trainer = SAM3TrainerNative(
        args.config,
        model_builder=resolve_builder(args.model_builder, "SAM3_LORA_MODEL_BUILDER", "model"),
        data_builder=resolve_builder(args.data_builder, "SAM3_LORA_DATA_BUILDER", "data"),
        bf16_frozen=args.bf16_frozen, act_checkpoint=args.act_checkpoint)
    trainer.train()
