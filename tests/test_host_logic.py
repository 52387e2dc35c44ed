"""CPU: host-side mirrors of the reference API (injection rules, init, checkpoint formats, C-ABI
exports) against manifests captured from the reference (tests/golden/*.json)."""
import contextlib
import ctypes
import io
import json
import os
import re

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import lora_oracle as O

import lora_layers as root_api            # repo-root shim -> sam3_lora_amd.lora_layers
import sam3_lora                          # shim -> sam3_lora_amd.lora
from sam3_lora import lora as pkg_api
from sam3_lora_amd import _ffi, build

from make_golden_models import ToySam, ROOT_CONFIGS, PKG_CONFIGS


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


@pytest.fixture(scope="module")
def toy(golden_dir):
    return json.load(open(os.path.join(golden_dir, "toy_manifests.json")))


@pytest.fixture(scope="module")
def sam3(golden_dir):
    return json.load(open(os.path.join(golden_dir, "sam3_linears.json")))


# ------------------------------------------------------------------ public surface ------
def test_public_names_and_defaults():
    import inspect
    for n in ("LoRALayer", "LoRALinear", "LoRAConfig", "apply_lora_to_model", "get_lora_parameters",
              "count_parameters", "save_lora_weights", "load_lora_weights"):
        assert hasattr(root_api, n)
    for n in ("LoRALayer", "LinearWithLoRA", "LoRAConfig", "inject_lora_into_model", "get_lora_parameters",
              "get_lora_state_dict", "load_lora_state_dict", "merge_lora_weights", "print_trainable_parameters"):
        assert hasattr(pkg_api, n)
    assert sam3_lora.LoRALayer is pkg_api.LoRALayer
    sig = inspect.signature(root_api.LoRALayer.__init__)
    assert [p for p in sig.parameters][1:] == ["in_features", "out_features", "rank", "alpha", "dropout"]
    assert (sig.parameters["rank"].default, sig.parameters["alpha"].default) == (8, 16)
    sig = inspect.signature(pkg_api.LoRALayer.__init__)
    assert (sig.parameters["rank"].default, sig.parameters["alpha"].default) == (4, 1.0)
    sig = inspect.signature(root_api.LoRAConfig.__init__)
    want = dict(rank=8, alpha=16, dropout=0.0, target_modules=None, apply_to_vision_encoder=True,
                apply_to_text_encoder=True, apply_to_geometry_encoder=False, apply_to_detr_encoder=True,
                apply_to_detr_decoder=True, apply_to_mask_decoder=False)
    assert {k: v.default for k, v in list(sig.parameters.items())[1:]} == want
    cfg = root_api.LoRAConfig()
    assert cfg.target_modules == {"q_proj", "k_proj", "v_proj", "out_proj"}
    assert set(cfg.to_dict()) == set(want)
    assert pkg_api.LoRAConfig().target_modules == set(O.PKG_DEFAULT_TARGETS)
    assert pkg_api.LoRAConfig(target_modules=["all"]).target_modules == set(O.PKG_ALL_TARGETS)


def test_param_shapes_init_and_scaling(golden_dir):
    st = json.load(open(os.path.join(golden_dir, "init_stats.json")))
    torch.manual_seed(0)
    for key, s in st.items():
        dims, r = key.split("_r")
        fin, fout = map(int, dims.split("x"))
        r = int(r)
        a = root_api.LoRALayer(fin, fout, rank=r, alpha=2 * r)
        p = pkg_api.LoRALayer(fin, fout, rank=r, alpha=2.0 * r)
        assert list(a.lora_A.shape) == s["root_A_shape"] and list(a.lora_B.shape) == s["root_B_shape"]
        assert list(p.lora_A.shape) == s["pkg_A_shape"] and list(p.lora_B.shape) == s["pkg_B_shape"]
        # same RNG consumption as the reference under the same seed -> identical extrema
        assert float(a.lora_A.detach().abs().max()) == pytest.approx(s["root_A_absmax"], abs=0)
        assert float(p.lora_A.detach().abs().max()) == pytest.approx(s["pkg_A_absmax"], abs=0)
        assert float(a.lora_B.detach().abs().max()) == 0 and float(p.lora_B.detach().abs().max()) == 0
        assert a.scaling == s["root_scaling"] and p.scaling == s["pkg_scaling"]
    assert isinstance(root_api.LoRALayer(8, 8).dropout, nn.Identity)
    assert isinstance(root_api.LoRALayer(8, 8, dropout=0.1).dropout, nn.Dropout)


# ------------------------------------------------------------------ injection rules -----
@pytest.mark.parametrize("cfg", sorted(ROOT_CONFIGS))
def test_root_injection_on_toy_model(toy, cfg):
    want = toy["root"][cfg]
    torch.manual_seed(0)
    m = ToySam()
    quiet(root_api.apply_lora_to_model, m, root_api.LoRAConfig(rank=4, alpha=8, **ROOT_CONFIGS[cfg]))
    names = [n for n, mod in m.named_modules() if isinstance(mod, root_api.LoRALinear)]
    assert names == want["names"]
    assert root_api.count_parameters(m) == want["counts"]
    assert len(root_api.get_lora_parameters(m)) == want["n_lora_params"]
    # base frozen, only lora_* trainable
    assert not any(p.requires_grad for n, p in m.named_parameters() if "lora_" not in n)
    assert all(p.requires_grad for n, p in m.named_parameters() if "lora_" in n)


@pytest.mark.parametrize("cfg", sorted(PKG_CONFIGS))
def test_package_injection_on_toy_model(toy, cfg):
    want = toy["package"][cfg]
    torch.manual_seed(0)
    m = ToySam()
    quiet(pkg_api.inject_lora_into_model, m, pkg_api.LoRAConfig(rank=4, alpha=8.0, target_modules=PKG_CONFIGS[cfg]), False)
    names = [n for n, mod in m.named_modules() if isinstance(mod, pkg_api.LinearWithLoRA)]
    assert names == want["names"]
    ps = pkg_api.get_lora_parameters(m)
    assert len(ps) == want["n_lora_params"] and sum(p.numel() for p in ps) == want["n_lora_elems"]
    # the package injector does not freeze the base (reference behaviour)
    assert any(p.requires_grad for n, p in m.named_parameters() if "lora_" not in n) == want["any_base_trainable"]


def _skeleton(linears):
    """A module tree with the reference SAM3 model's Linear names (meta device, no memory)."""
    root = nn.Module()
    for name, fin, fout, bias in linears:
        parts = name.split(".")
        cur = root
        for p in parts[:-1]:
            if not hasattr(cur, p):
                cur.add_module(p, nn.Module())
            cur = getattr(cur, p)
        cur.add_module(parts[-1], nn.Linear(fin, fout, bias=bias, device="meta"))
    return root


def test_root_injection_on_sam3_names(sam3):
    """Every shipped YAML: same adapted module names and trainable-parameter counts as the reference
    (full -> 64 modules / 11,796,480; crack -> 64 / 5,898,240; minimal/light/base -> 0)."""
    assert sam3["total_parameters"] == 840509750
    for yaml_name, want in sam3["root"].items():
        m = _skeleton(sam3["linears"])
        quiet(root_api.apply_lora_to_model, m, root_api.LoRAConfig(**want["lora"]))
        names = [n for n, mod in m.named_modules() if isinstance(mod, root_api.LoRALinear)]
        assert names == want["names"], yaml_name
        assert root_api.count_parameters(m)["trainable_parameters"] == want["trainable_parameters"], yaml_name
        # the oracle's restatement of the rule agrees as well
        flags = {k: v for k, v in want["lora"].items() if k.startswith("apply_to")}
        sel = [n for n, *_ in sam3["linears"] if O.root_should_apply(n, want["lora"]["target_modules"], **flags)]
        assert sel == want["names"], yaml_name
    assert len(sam3["root"]["full_lora_config.yaml"]["names"]) == 64
    assert sam3["root"]["full_lora_config.yaml"]["trainable_parameters"] == 11796480


def test_package_injection_on_sam3_names(sam3):
    for key, want in sam3["package"].items():
        m = _skeleton(sam3["linears"])
        cfg = pkg_api.LoRAConfig(rank=want["rank"], alpha=2.0 * want["rank"], target_modules=want["target_modules"])
        quiet(pkg_api.inject_lora_into_model, m, cfg, False)
        names = [n for n, mod in m.named_modules() if isinstance(mod, pkg_api.LinearWithLoRA)]
        assert names == want["names"], key
        assert sum(p.numel() for p in pkg_api.get_lora_parameters(m)) == want["n_lora_elems"], key
        tg = O.package_targets(want["target_modules"])
        assert [n for n, *_ in sam3["linears"] if O.package_should_inject(n, tg)] == want["names"], key


# ------------------------------------------------------------------ checkpoint formats --
def test_checkpoint_key_formats_and_roundtrip(golden_dir, tmp_path):
    want = json.load(open(os.path.join(golden_dir, "ckpt_keys.json")))
    torch.manual_seed(0)
    m = ToySam()
    quiet(root_api.apply_lora_to_model, m, root_api.LoRAConfig(rank=4, alpha=8, **ROOT_CONFIGS["fc_only_vision"]))
    with torch.no_grad():
        for p in root_api.get_lora_parameters(m):
            p.normal_()
    path = str(tmp_path / "last_lora_weights.pt")
    quiet(root_api.save_lora_weights, m, path)
    sd = torch.load(path, weights_only=False)
    assert {k: dict(shape=list(v.shape), is_parameter=isinstance(v, nn.Parameter)) for k, v in sd.items()} == want["root"]
    m2 = ToySam()
    quiet(root_api.apply_lora_to_model, m2, root_api.LoRAConfig(rank=4, alpha=8, **ROOT_CONFIGS["fc_only_vision"]))
    quiet(root_api.load_lora_weights, m2, path)
    for (n1, p1), (n2, p2) in zip(m.named_parameters(), m2.named_parameters()):
        if "lora_" in n1:
            assert n1 == n2 and torch.equal(p1, p2)

    mp = ToySam()
    quiet(pkg_api.inject_lora_into_model, mp, pkg_api.LoRAConfig(rank=4, alpha=8.0, target_modules=["fc1", "fc2"]), False)
    psd = pkg_api.get_lora_state_dict(mp)
    assert {k: dict(shape=list(v.shape), is_parameter=isinstance(v, nn.Parameter)) for k, v in psd.items()} == want["package"]
    mp2 = ToySam()
    quiet(pkg_api.inject_lora_into_model, mp2, pkg_api.LoRAConfig(rank=4, alpha=8.0, target_modules=["fc1", "fc2"]), False)
    pkg_api.load_lora_state_dict(mp2, {k: v + 1 for k, v in psd.items()})
    for k, v in pkg_api.get_lora_state_dict(mp2).items():
        assert torch.equal(v, psd[k] + 1)


def test_linear_with_lora_exposes_linear_attrs():
    lin = nn.Linear(12, 20)
    w = pkg_api.LinearWithLoRA(lin, rank=2, alpha=4.0)
    assert w.weight is lin.weight and w.bias is lin.bias
    assert (w.in_features, w.out_features) == (12, 20)
    assert not lin.weight.requires_grad


# ------------------------------------------------------------------ C-ABI ----------------
def lib_has_packed_sizes():
    """sam3_lora_packed_bytes: four operand images per rank group of <= 32 (bf16, or fp32 for the exact path), each
    256-byte aligned; 0 (with an error text) on bad shapes.  bf16: hi + lo images -- a group of <= 16 rank indices is laid out 32 wide,
    one of 17..32 is laid out 64 wide."""
    lib = _ffi.load()
    n16 = lib.sam3_lora_packed_bytes(1024, 4736, 16, _ffi.DT_F32)          # fp32 images of a 16-wide rank tile
    assert n16 == 2 * 16 * (1024 + 4736) * 4 and n16 % 256 == 0
    n = lib.sam3_lora_packed_bytes(1024, 4736, 16, _ffi.DT_BF16)            # hi | lo: 32 bf16 columns
    assert n == 2 * 32 * (1024 + 4736) * 2 == n16 and lib.sam3_lora_packed_bytes(1024, 4736, 3, _ffi.DT_BF16) == n
    n32 = lib.sam3_lora_packed_bytes(1024, 4736, 32, _ffi.DT_BF16)
    assert n32 == 2 * n and lib.sam3_lora_packed_bytes(1024, 4736, 17, _ffi.DT_BF16) == n32 and lib.sam3_lora_packed_bytes(1024, 4736, 32, _ffi.DT_F32) == 2 * n16
    # rank 33..64: two groups (32 + the rest); rank 80 = 32 + 32 + 16
    assert lib.sam3_lora_packed_bytes(1024, 4736, 64, _ffi.DT_BF16) == 2 * n32
    assert lib.sam3_lora_packed_bytes(1024, 4736, 40, _ffi.DT_BF16) == n32 + n
    assert lib.sam3_lora_packed_bytes(1024, 4736, 80, _ffi.DT_BF16) == 2 * n32 + n
    assert lib.sam3_lora_saved_t_bytes(5184, 16, _ffi.DT_BF16) == 32 * 5184 * 2 == lib.sam3_lora_saved_t_bytes(5184, 32, _ffi.DT_BF16) // 2
    assert lib.sam3_lora_packed_bytes(1024, 4736, _ffi.MAX_RANK + 1, _ffi.DT_BF16) == 0 and "rank" in _ffi.last_error()
    assert lib.sam3_lora_packed_bytes(1024, 4736, 0, _ffi.DT_BF16) == 0
    return True


def test_cabi_library_builds_loads_and_exports_header_symbols():
    """No compute call (no GPU here): the library exists, loads, and exports every function the
    header declares."""
    path = build.build_library()
    assert os.path.exists(path)
    hdr = open(os.path.join(build.INCLUDE, "sam3_lora_amd.h")).read()
    declared = set(re.findall(r"\b(sam3_lora_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_ffi.EXPORTS), declared ^ set(_ffi.EXPORTS)
    vit_hdr = open(os.path.join(build.INCLUDE, "sam3_vit_amd.h")).read()
    vit_declared = set(re.findall(r"\b(sam3_vit_[a-z0-9_]+)\s*\(", vit_hdr))
    assert vit_declared == set(_ffi.VIT_EXPORTS), vit_declared ^ set(_ffi.VIT_EXPORTS)
    loss_hdr = open(os.path.join(build.INCLUDE, "sam3_loss_amd.h")).read()
    loss_declared = set(re.findall(r"\b(sam3_(?:loss|mask_loss|box_pair)_[a-z_]+)\s*\(", loss_hdr))
    assert loss_declared == set(_ffi.LOSS_EXPORTS), loss_declared ^ set(_ffi.LOSS_EXPORTS)
    fp8_hdr = open(os.path.join(build.INCLUDE, "sam3_fp8_amd.h")).read()
    fp8_declared = set(re.findall(r"\b(sam3_fp8_[a-z_]+)\s*\(", fp8_hdr))
    assert fp8_declared == set(_ffi.FP8_EXPORTS), fp8_declared ^ set(_ffi.FP8_EXPORTS)
    # the amax slot layout the Python side allocates is the one the header states
    consts = {k: int(v) for k, v in re.findall(r"#define (SAM3_FP8_(?:AMAX_SLOTS|AMAX_STRIDE|E4M3|E5M2))\s+(\d+)", fp8_hdr)}
    assert consts == {"SAM3_FP8_AMAX_SLOTS": _ffi.FP8_AMAX_SLOTS, "SAM3_FP8_AMAX_STRIDE": _ffi.FP8_AMAX_STRIDE,
                      "SAM3_FP8_E4M3": _ffi.FP8_E4M3, "SAM3_FP8_E5M2": _ffi.FP8_E5M2}
    assert _ffi.FP8_AMAX_FLOATS == _ffi.FP8_AMAX_SLOTS * _ffi.FP8_AMAX_STRIDE and _ffi.FP8_AMAX_STRIDE * 4 == 128
    seg_hdr = open(os.path.join(build.INCLUDE, "sam3_seg_amd.h")).read()
    seg_declared = set(re.findall(r"\b(sam3_(?:seg|gn|rpb)_[a-z_]+)\s*\(", seg_hdr))
    assert seg_declared == set(_ffi.SEG_EXPORTS), seg_declared ^ set(_ffi.SEG_EXPORTS)
    assert lib_has_packed_sizes()
    lib = ctypes.CDLL(path)
    for sym in declared | vit_declared | loss_declared | fp8_declared | seg_declared:
        assert hasattr(lib, sym), sym
    lib2 = _ffi.load()
    assert lib2.sam3_lora_abi_version() == _ffi.ABI_VERSION == 6
    # which backward calls may leave x out (the GELU' pass recomputes it): a host-side rule, no device needed
    assert lib2.sam3_lora_bwd_act_recomputes_input(16, _ffi.DT_BF16, 0.0) == 1 and lib2.sam3_lora_bwd_act_recomputes_input(3, _ffi.DT_BF16, 0.0) == 1
    assert lib2.sam3_lora_bwd_act_recomputes_input(17, _ffi.DT_BF16, 0.0) == 1 and lib2.sam3_lora_bwd_act_recomputes_input(32, _ffi.DT_BF16, 0.1) == 1
    assert lib2.sam3_lora_bwd_act_recomputes_input(33, _ffi.DT_BF16, 0.0) == 0 and lib2.sam3_lora_bwd_act_recomputes_input(16, _ffi.DT_F32, 0.0) == 0
    assert lib2.sam3_lora_bwd_act_recomputes_input(16, _ffi.DT_BF16, 0.1) == 1      # (round 6: one rank group of <= 32, with or without the mask)
    # pure host-side argument validation works without a device
    assert lib2.sam3_lora_fwd_workspace_bytes(41472, 1024, 4736, 16, 0) > 0
    assert lib2.sam3_lora_fwd_workspace_bytes(41472, 1023, 4736, 16, 0) == 0
    assert "multiples of 8" in _ffi.last_error()
    assert lib2.sam3_lora_saved_t_bytes(41472, 16, _ffi.DT_BF16) == 32 * 41472 * 2       # hi | lo
    assert lib2.sam3_lora_saved_t_bytes(41472, 16, _ffi.DT_F32) == 16 * 41472 * 4
    assert lib2.sam3_lora_saved_t_bytes(100, 17, _ffi.DT_BF16) == 64 * 128 * 2
    assert lib2.sam3_lora_saved_t_bytes(100, 64, _ffi.DT_BF16) == 2 * 64 * 128 * 2
    # workspaces are sized for one rank group: rank 64 needs what rank 32 needs
    assert (lib2.sam3_lora_bwd_workspace_bytes(41472, 1024, 4736, 64, 0)
            == lib2.sam3_lora_bwd_workspace_bytes(41472, 1024, 4736, 32, 0) > 0)


def test_forward_refuses_cpu_tensors():
    from sam3_lora_amd.functional import LoRAKernelError
    with pytest.raises(LoRAKernelError, match="no CPU fallback"):
        root_api.LoRALinear(nn.Linear(16, 16))(torch.zeros(2, 16))
    with pytest.raises(LoRAKernelError, match="no CPU fallback"):
        pkg_api.LinearWithLoRA(nn.Linear(16, 16))(torch.zeros(2, 16))


def test_nonfinite_step_guard_skips_the_update_and_counts_it():
    """fp8 frozen-W mode (trainer.NonFiniteStepGuard): a step whose gradients are not all finite leaves the parameters, the moments and
    the step counts untouched -- the reference's GradScaler behaviour (native_trainer.py:902-903) -- and the next finite step is an
    ordinary AdamW step; with the unfused optimizer the same through one host test."""
    import torch
    from sam3_lora_amd.trainer import NonFiniteStepGuard
    for fused in (True, False):
        p = [torch.nn.Parameter(torch.ones(4)), torch.nn.Parameter(torch.ones(3))]
        try:
            opt = torch.optim.AdamW(p, lr=0.1, weight_decay=0.0, fused=fused)
        except (RuntimeError, TypeError):
            continue
        guard = NonFiniteStepGuard(opt, torch.device("cpu"))
        p[0].grad, p[1].grad = torch.tensor([1.0, float("inf"), 0.0, 0.0]), torch.ones(3)
        guard.step()
        assert torch.equal(p[0].data, torch.ones(4)) and torch.equal(p[1].data, torch.ones(3)) and float(guard.skipped) == 1.0
        assert not opt.state or all(float(st["step"]) == 0.0 for st in opt.state.values())
        p[0].grad = torch.tensor([float("nan"), 0.0, 0.0, 0.0])
        guard.step()
        assert torch.equal(p[0].data, torch.ones(4)) and float(guard.skipped) == 2.0
        p[0].grad = torch.ones(4)
        guard.step()
        assert float(guard.skipped) == 2.0 and torch.allclose(p[0].data, torch.full((4,), 0.9), atol=1e-3)
        assert all(float(st["step"]) == 1.0 for st in opt.state.values())


def test_matcher_raises_on_a_nonfinite_cost_like_the_reference_unless_the_fp8_mode_is_on():
    import numpy as np
    import pytest as _pytest
    from sam3_lora_amd import fp8, matcher
    cost = np.array([[0.1, np.nan], [0.3, 0.2]])
    with _pytest.raises(ValueError):
        matcher._solve(cost, 1, True, False)
    fp8.enable_fp8_frozen(True)
    try:
        rows, cols = matcher._solve(cost, 1, True, False)
        assert sorted(rows.tolist()) == [0, 1] and sorted(cols.tolist()) == [0, 1]
    finally:
        fp8.enable_fp8_frozen(False)
