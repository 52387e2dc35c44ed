#!/usr/bin/env python3
"""Names, shapes and dtypes of every state-dict entry of the REFERENCE SAM3 image model (840,509,750 parameters),
instantiated on the meta device by sam3_manifest.build_reference_model_meta() in training mode, plus the ids the
reference BPE tokenizer assigns to a list of prompt words.  Names and integers only.  Build container only.

    python tests/golden/make_state_keys_golden.py  ->  sam3_state_keys.json, ../../sam3_lora_amd/assets/prompt_tokens.json
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != os.path.abspath(os.path.join(HERE, "..", ".."))]
import sam3_manifest

# category names of the datasets the reference's docs mention + COCO's 80 + generic prompts
PROMPTS = ["crack", "object", "pothole", "person", "bicycle", "car", "motorcycle", "airplane", "bus", "train", "truck",
           "boat", "traffic light", "fire hydrant", "stop sign", "parking meter", "bench", "bird", "cat", "dog", "horse",
           "sheep", "cow", "elephant", "bear", "zebra", "giraffe", "backpack", "umbrella", "handbag", "tie", "suitcase",
           "frisbee", "skis", "snowboard", "sports ball", "kite", "baseball bat", "baseball glove", "skateboard",
           "surfboard", "tennis racket", "bottle", "wine glass", "cup", "fork", "knife", "spoon", "bowl", "banana",
           "apple", "sandwich", "orange", "broccoli", "carrot", "hot dog", "pizza", "donut", "cake", "chair", "couch",
           "potted plant", "bed", "dining table", "toilet", "tv", "laptop", "mouse", "remote", "keyboard", "cell phone",
           "microwave", "oven", "toaster", "sink", "refrigerator", "book", "clock", "vase", "scissors", "teddy bear",
           "hair drier", "toothbrush", "defect", "scratch", "dent", "corrosion", "rust", "spalling", "concrete crack",
           "road", "building", "tree", "cell", "tumor", "lesion", "leaf", "fruit", "weed", "a photo of a crack",
           "it's a dog's toy!", "yellow school bus", "damaged wall", "hole", "stain", "bolt", "weld", "pipe"]


def main():
    model = sam3_manifest.build_reference_model_meta()
    sd = model.state_dict()
    out = dict(total_parameters=int(sum(p.numel() for p in model.parameters())),
               entries=[[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()])
    json.dump(out, open(os.path.join(HERE, "sam3_state_keys.json"), "w"))
    print("state entries:", len(out["entries"]), "params:", out["total_parameters"])
    sys.modules["ftfy"].fix_text = lambda t: t          # harness stand-in: identity on well-formed text
    from sam3.model import tokenizer_ve
    tokenizer_ve.ftfy.fix_text = lambda t: t
    from sam3.model.tokenizer_ve import SimpleTokenizer
    tok = SimpleTokenizer(bpe_path=os.path.join(sam3_manifest.REF, "sam3", "assets", "bpe_simple_vocab_16e6.txt.gz"))
    table = {}
    for p in PROMPTS:
        table[tok.clean_fn(p)] = tok.encode(p)
    rows = tok(PROMPTS[:6] + ["word " * 40], context_length=32)
    assets = os.path.join(HERE, "..", "..", "sam3_lora_amd", "assets")
    os.makedirs(assets, exist_ok=True)
    json.dump({"source": "ids assigned by the reference SimpleTokenizer (tokenizer_ve.py) with OpenAI CLIP's BPE vocabulary",
               "sot": tok.sot_token_id, "eot": tok.eot_token_id, "vocab_size": tok.vocab_size, "tokens": table},
              open(os.path.join(assets, "prompt_tokens.json"), "w"), indent=0)
    json.dump({"texts": PROMPTS[:6] + ["word " * 40], "context_length": 32, "rows": rows.tolist()},
              open(os.path.join(HERE, "tokenizer_rows.json"), "w"))
    print("prompts:", len(table), "vocab:", tok.vocab_size)


if __name__ == "__main__":
    main()
