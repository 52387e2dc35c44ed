#!/usr/bin/env python3
"""
Does the hardware saturation the fp8 quantisers rely on (MODE.FP16_OVFL, set per wave by q8_begin) EVER fail?  Delayed scaling
quantises with the range of the predecessor, so finite values beyond the range are routine; they must leave as +-max, never as the
format's non-finite encoding (e4m3: NaN, e5m2: Inf).  The stand-alone quantiser and the LayerNorm producer are launched ITERS times on
a tensor most of whose elements lie beyond the range they are given; every output byte is tested on the device, one sync at the end.
    python tools/fp8_saturation_hammer.py [--iters 20000]
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from sam3_lora_amd import _ffi


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20000)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    lib = _ffi.load()
    g = torch.Generator(device=dev).manual_seed(3)
    M, C = 8192, 1024
    x = (torch.randn(M, C, device=dev, generator=g) * 100.0).bfloat16()        # |x| up to ~500
    out = {}
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    from sam3_lora_amd import vit as V
    w_ln = (1 + 0.1 * torch.randn(C, device=dev, generator=g)).bfloat16() * 50.0       # LayerNorm output ~ N(0, 50): far beyond a range of 1
    b_ln = (0.1 * torch.randn(C, device=dev, generator=g)).bfloat16()
    for fmt, name in ((_ffi.FP8_E4M3, "e4m3"), (_ffi.FP8_E5M2, "e5m2")):
        for producer in ("quantiser", "layernorm"):
            amax_in = torch.zeros(_ffi.FP8_AMAX_FLOATS, device=dev)
            amax_in[0] = 1.0                               # the range the call is given: |x| <= 1 -- almost every element is beyond it
            amax_out = torch.zeros(_ffi.FP8_AMAX_FLOATS, device=dev)
            scale = torch.ones(1, device=dev)
            img = torch.empty(M, C, dtype=torch.float8_e4m3fn if name == "e4m3" else torch.float8_e5m2, device=dev)
            bytes_ = img.view(torch.uint8)
            worst = torch.zeros((), dtype=torch.uint8, device=dev)     # largest magnitude byte seen in ANY image (0x7E / 0x7B = the format's max)
            for it in range(args.iters):
                if producer == "quantiser":
                    rc = lib.sam3_fp8_quantize(x.data_ptr(), img.data_ptr(), amax_in.data_ptr(), amax_out.data_ptr(), scale.data_ptr(), x.numel(), 0, fmt, st)
                    assert rc == 0
                else:
                    V._FrozenLayerNorm.apply(x, w_ln, b_ln, 1e-5, (img, fmt, amax_in, amax_out, scale))
                worst = torch.maximum(worst, (bytes_ & 0x7F).max())
            torch.cuda.synchronize()
            limit = 0x7E if name == "e4m3" else 0x7B
            out[f"{producer}_{name}"] = {"launches_all_tested": args.iters, "elements_per_image": M * C, "largest_magnitude_byte": hex(int(worst)),
                                         "format_max_byte": hex(limit), "non_finite_seen": int(worst) > limit}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
