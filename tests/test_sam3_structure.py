"""Full-size structure of the SAM3 image model restatement against names / shapes recorded from the reference model
(tests/golden/make_state_keys_golden.py, sam3_manifest.py): every state-dict entry, every nn.Linear, the modules each
shipped YAML adapts -- on the meta device, no weights allocated.  Tokenizer against the reference's ids."""
import contextlib
import io
import json
import os

import pytest
import torch
import torch.nn as nn

from sam3_lora_amd.sam3_image import build_sam3_image_model
from sam3_lora_amd.sam3_text import SimpleTokenizer

HERE = os.path.dirname(os.path.abspath(__file__))
VOCAB = "/root/reference/sam3/assets/bpe_simple_vocab_16e6.txt.gz"     # build container only; tests skip without it


@pytest.fixture(scope="module")
def full_model():
    with torch.device("meta"):
        return build_sam3_image_model(device="meta", eval_mode=False, tokenizer=lambda *a, **k: None)


def test_full_size_state_dict_is_the_references(full_model):
    ref = json.load(open(os.path.join(HERE, "golden", "sam3_state_keys.json")))
    ours = {k: (list(v.shape), str(v.dtype).replace("torch.", "")) for k, v in full_model.state_dict().items()}
    theirs = {k: (shape, dt) for k, shape, dt in ref["entries"]}
    assert sorted(set(theirs) - set(ours)) == [], "missing entries"
    assert sorted(set(ours) - set(theirs)) == [], "unexpected entries"
    assert [k for k in theirs if ours[k] != theirs[k]] == []
    assert sum(p.numel() for p in full_model.parameters()) == ref["total_parameters"] == 840_509_750


def test_full_size_linears_and_injection_manifests(full_model):
    import copy
    from sam3_lora_amd import lora_layers as L
    from sam3_lora_amd.lora import lora_utils as U, lora_layer as PL
    ref = json.load(open(os.path.join(HERE, "golden", "sam3_linears.json")))
    ours = [(n, m.in_features, m.out_features, m.bias is not None)
            for n, m in full_model.named_modules() if isinstance(m, nn.Linear)]
    assert sorted(map(tuple, ref["linears"])) == sorted(ours)
    for cfg_name, rec in ref["root"].items():
        m = copy.deepcopy(full_model)
        with contextlib.redirect_stdout(io.StringIO()), torch.device("meta"):
            L.apply_lora_to_model(m, L.LoRAConfig(**rec["lora"]))
        names = [n for n, mm in m.named_modules() if isinstance(mm, L.LoRALinear)]
        assert names == rec["names"], cfg_name
        assert L.count_parameters(m)["trainable_parameters"] == rec["trainable_parameters"], cfg_name
    for key, rec in ref["package"].items():
        m = copy.deepcopy(full_model)
        with contextlib.redirect_stdout(io.StringIO()), torch.device("meta"):
            U.inject_lora_into_model(m, U.LoRAConfig(rank=rec["rank"], alpha=2.0 * rec["rank"],
                                                     target_modules=rec["target_modules"]), False)
        names = [n for n, mm in m.named_modules() if isinstance(mm, PL.LinearWithLoRA)]
        assert names == rec["names"], key
        assert sum(p.numel() for p in U.get_lora_parameters(m)) == rec["n_lora_elems"], key


def test_prompt_table_tokenizer_without_vocabulary():
    tok = SimpleTokenizer(bpe_path=None)
    table = json.load(open(os.path.join(HERE, "..", "sam3_lora_amd", "assets", "prompt_tokens.json")))
    assert (tok.sot_token_id, tok.eot_token_id, tok.vocab_size) == (table["sot"], table["eot"], table["vocab_size"])
    row = tok(["Crack", "  traffic   light "], context_length=32)
    assert row.shape == (2, 32) and row.dtype == torch.long
    assert row[0].tolist()[:2 + len(table["tokens"]["crack"])] == [table["sot"]] + table["tokens"]["crack"] + [table["eot"]]
    assert row[0, 2 + len(table["tokens"]["crack"]):].eq(0).all()
    assert row[1].tolist()[:2 + len(table["tokens"]["traffic light"])] == \
        [table["sot"]] + table["tokens"]["traffic light"] + [table["eot"]]
    with pytest.raises(FileNotFoundError):
        tok(["a prompt that is not in the table"])


@pytest.mark.skipif(not os.path.exists(VOCAB), reason="CLIP BPE vocabulary file not present")
def test_bpe_tokenizer_reproduces_reference_ids():
    tok = SimpleTokenizer(bpe_path=VOCAB)
    table = json.load(open(os.path.join(HERE, "..", "sam3_lora_amd", "assets", "prompt_tokens.json")))
    for text, ids in table["tokens"].items():
        assert tok.encode(text) == ids, text
    rows = json.load(open(os.path.join(HERE, "golden", "tokenizer_rows.json")))
    assert tok(rows["texts"], context_length=rows["context_length"]).tolist() == rows["rows"]
    assert tok.decode(tok.encode("yellow school bus")).strip() == "yellow school bus"
