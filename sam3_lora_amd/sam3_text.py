"""
Text side of the SAM3 image model: byte-pair tokenizer and the CLIP-style text tower.

Restates, for the training step of row a14, what ``sam3/model/tokenizer_ve.py:127-253`` (``SimpleTokenizer``) and
``sam3/model/text_encoder_ve.py`` (``ResidualAttentionBlock`` :13-88, ``Transformer`` :91-145, ``TextTransformer``
:165-252, ``VETextEncoder`` :255-328) compute.  Module / parameter names are the reference's, because

  * the injectors address modules by name (``language_backbone`` gates ``apply_to_text_encoder``; ``c_fc`` /
    ``c_proj`` / ``out_proj`` are adapter targets of the package API), and
  * a reference state dict must load with ``strict=True`` (``tests/golden/sam3_state_keys.json``).

Everything here runs on stock PyTorch-ROCm (north star: only the LoRA adapter path is hand-written HIP); the adapted
``nn.Linear``s inside it reach the HIP kernels through ``LoRALinear`` / ``LinearWithLoRA`` like any other.

Differences from the reference that are deliberate:
  * ``positional_embedding`` and ``text_projection`` get a seeded-init (N(0, 0.01) and N(0, width^-0.5)) instead of
    ``torch.empty`` garbage, so a randomly initialised model is finite;
  * the tokenizer loads its merge table from ``bpe_path`` when the file exists; the vocabulary file is a third-party
    data asset (OpenAI CLIP's ``bpe_simple_vocab_16e6.txt.gz``) that this repository does not redistribute -- without
    it, prompts found in the small committed table ``sam3_lora_amd/assets/prompt_tokens.json`` (token ids produced by the
    reference tokenizer) still encode, anything else raises.
"""
from __future__ import annotations

import gzip
import html
import json
import os
from collections import OrderedDict
from functools import lru_cache
from typing import Callable, Dict, List, Optional, Sequence, Tuple, Union

import torch
import torch.nn as nn
from torch.utils.checkpoint import checkpoint

__all__ = ["SimpleTokenizer", "ResidualAttentionBlock", "Transformer", "TextTransformer", "VETextEncoder",
           "LayerScale"]

SOT, EOT = "<start_of_text>", "<end_of_text>"
N_MERGES = 49152 - 256 - 2          # merge rules kept from the vocabulary file (tokenizer_ve.py:142)
_TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "prompt_tokens.json")


@lru_cache()
def _byte_alphabet() -> Dict[int, str]:
    """The reversible byte -> printable-unicode table of GPT-2 / CLIP BPE (tokenizer_ve.py:30-52): printable latin-1
    bytes map to themselves, the remaining 68 bytes to code points 256.. in byte order."""
    keep = list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256))
    table = {b: chr(b) for b in keep}
    extra = 0
    for b in range(256):
        if b not in table:
            table[b] = chr(256 + extra)
            extra += 1
    return table


def _clean_lower(text: str) -> str:
    """``clean="lower"`` (tokenizer_ve.py:84-86): unescape HTML twice, collapse whitespace, lower-case.  (The
    reference also runs ``ftfy.fix_text`` first -- a mojibake repair that is the identity on well-formed text; ftfy is
    not a dependency here.)"""
    text = html.unescape(html.unescape(text)).strip()
    return " ".join(text.split()).lower()


class SimpleTokenizer:
    """CLIP byte-pair tokenizer: ``tokenizer(texts, context_length) -> LongTensor[len(texts), context_length]``,
    ``<start_of_text> ids... <end_of_text>`` right-padded with zeros, truncated with the last id forced to EOT."""

    def __init__(self, bpe_path: Optional[Union[str, os.PathLike]] = None, context_length: Optional[int] = 77):
        import regex
        self.context_length = context_length
        self.byte_encoder = _byte_alphabet()
        self.byte_decoder = {v: k for k, v in self.byte_encoder.items()}
        self.have_vocab = bool(bpe_path) and os.path.exists(bpe_path)
        merges: List[Tuple[str, ...]] = []
        if self.have_vocab:
            with gzip.open(bpe_path, "rb") as fh:
                lines = fh.read().decode("utf-8").split("\n")
            merges = [tuple(ln.split()) for ln in lines[1:N_MERGES + 1]]
        alphabet = list(self.byte_encoder.values())
        vocab = alphabet + [c + "</w>" for c in alphabet] + ["".join(m) for m in merges]
        if not self.have_vocab:
            vocab += [None] * N_MERGES          # ids of the special tokens do not depend on the merge table
        vocab += [SOT, EOT]
        self.encoder = {tok: i for i, tok in enumerate(vocab) if tok is not None}
        self.decoder = {i: tok for tok, i in self.encoder.items()}
        self.bpe_ranks = {m: i for i, m in enumerate(merges)}
        self.vocab_size = len(vocab)
        self.sot_token_id, self.eot_token_id = self.encoder[SOT], self.encoder[EOT]
        self.all_special_ids = [self.sot_token_id, self.eot_token_id]
        self._word_cache = {SOT: SOT, EOT: EOT}
        self._split = regex.compile(
            r"<start_of_text>|<end_of_text>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+", regex.IGNORECASE)
        self._table: Dict[str, List[int]] = {}
        if os.path.exists(_TABLE):
            with open(_TABLE) as f:
                self._table = json.load(f)["tokens"]

    # -- byte-pair merging of one pre-token ------------------------------------------------------------------
    def bpe(self, token: str) -> str:
        hit = self._word_cache.get(token)
        if hit is not None:
            return hit
        parts = list(token[:-1]) + [token[-1] + "</w>"]
        unranked = float("inf")
        while len(parts) > 1:
            # the adjacent pair with the lowest merge rank goes first
            a, b = min(set(zip(parts, parts[1:])), key=lambda pair: self.bpe_ranks.get(pair, unranked))
            if (a, b) not in self.bpe_ranks:
                break
            merged, i = [], 0
            while i < len(parts):           # merge every non-overlapping occurrence of (a, b), left to right
                if i + 1 < len(parts) and parts[i] == a and parts[i + 1] == b:
                    merged.append(a + b)
                    i += 2
                else:
                    merged.append(parts[i])
                    i += 1
            parts = merged
        out = " ".join(parts)
        self._word_cache[token] = out
        return out

    def encode(self, text: str) -> List[int]:
        text = _clean_lower(text)
        if not self.have_vocab:
            if text in self._table:
                return list(self._table[text])
            raise FileNotFoundError(
                f"SimpleTokenizer: no BPE vocabulary file and {text!r} is not in the committed prompt table; pass "
                f"bpe_path=<bpe_simple_vocab_16e6.txt.gz> (model.bpe_path in the YAML / SAM3_BPE_PATH)")
        ids: List[int] = []
        for piece in self._split.findall(text):
            piece = "".join(self.byte_encoder[b] for b in piece.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self.bpe(piece).split(" "))
        return ids

    def decode(self, tokens: Sequence[int]) -> str:
        text = "".join(self.decoder[int(t)] for t in tokens)
        return bytearray(self.byte_decoder[c] for c in text).decode("utf-8", errors="replace").replace("</w>", " ")

    def __call__(self, texts: Union[str, Sequence[str]], context_length: Optional[int] = None) -> torch.Tensor:
        if isinstance(texts, str):
            texts = [texts]
        n = context_length or self.context_length
        assert n, "Please set a valid context length"
        out = torch.zeros(len(texts), n, dtype=torch.long)
        for row, text in enumerate(texts):
            ids = [self.sot_token_id] + self.encode(text) + [self.eot_token_id]
            if len(ids) > n:
                ids = ids[:n]
                ids[-1] = self.eot_token_id
            out[row, :len(ids)] = torch.tensor(ids)
        return out


# ------------------------------------------------------------------------------------------------ the tower --
class LayerScale(nn.Module):
    """Per-channel learnable gain (model_misc.py:94-105)."""

    def __init__(self, dim: int, init_values: float = 1e-5, inplace: bool = False):
        super().__init__()
        self.inplace = inplace
        self.gamma = nn.Parameter(init_values * torch.ones(dim))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return x.mul_(self.gamma) if self.inplace else x * self.gamma


class ResidualAttentionBlock(nn.Module):
    """Pre-norm block ``x += attn(ln_1(x)); x += mlp(ln_2(x))`` with ``nn.MultiheadAttention(batch_first=True)`` and
    the MLP ``c_fc -> GELU -> c_proj`` (text_encoder_ve.py:13-88)."""

    def __init__(self, d_model: int, n_head: int, mlp_ratio: float = 4.0, ls_init_value: Optional[float] = None,
                 act_layer: Callable[[], nn.Module] = nn.GELU, norm_layer: Callable[[int], nn.Module] = nn.LayerNorm):
        super().__init__()
        from .sam3_detr import MultiheadAttention      # same module, evaluated without transposed-view projections
        self.attn = MultiheadAttention(d_model, n_head, batch_first=True)
        self.ln_1 = norm_layer(d_model)
        self.ln_2 = norm_layer(d_model)
        self.ls_1 = LayerScale(d_model, ls_init_value) if ls_init_value is not None else nn.Identity()
        self.ls_2 = LayerScale(d_model, ls_init_value) if ls_init_value is not None else nn.Identity()
        hidden = int(d_model * mlp_ratio)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d_model, hidden)), ("gelu", act_layer()),
                                              ("c_proj", nn.Linear(hidden, d_model))]))

    def forward(self, q_x: torch.Tensor, k_x: Optional[torch.Tensor] = None, v_x: Optional[torch.Tensor] = None,
                attn_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        h = self.ln_1(q_x)
        if attn_mask is not None and attn_mask.dtype != torch.bool:
            attn_mask = attn_mask.to(h.dtype)
        k = h if k_x is None else k_x
        v = h if v_x is None else v_x
        x = q_x + self.ls_1(self.attn(h, k, v, need_weights=False, attn_mask=attn_mask)[0])
        return x + self.ls_2(self.mlp(self.ln_2(x)))


class Transformer(nn.Module):
    def __init__(self, width: int, layers: int, heads: int, mlp_ratio: float = 4.0,
                 ls_init_value: Optional[float] = None, act_layer: Callable = nn.GELU,
                 norm_layer: Callable = nn.LayerNorm, use_act_checkpoint: bool = False):
        super().__init__()
        self.width, self.layers = width, layers
        self.grad_checkpointing = use_act_checkpoint
        self.resblocks = nn.ModuleList([ResidualAttentionBlock(width, heads, mlp_ratio, ls_init_value, act_layer,
                                                               norm_layer) for _ in range(layers)])

    def forward(self, x: torch.Tensor, attn_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        for blk in self.resblocks:
            if self.grad_checkpointing and self.training and torch.is_grad_enabled():
                x = checkpoint(blk, x, None, None, attn_mask, use_reentrant=False)
            else:
                x = blk(x, attn_mask=attn_mask)
        return x


class TextTransformer(nn.Module):
    """Token + position embedding, causal transformer, ``ln_final``; returns ``(pooled @ text_projection, tokens)``
    with the pooled token taken at the EOT position (the arg-max id) -- text_encoder_ve.py:165-252 in the
    configuration ``VETextEncoder`` builds (``output_tokens=True``, ``pool_type="none"`` keeps all tokens)."""

    def __init__(self, context_length: int = 77, vocab_size: int = 49408, width: int = 512, heads: int = 8,
                 layers: int = 12, mlp_ratio: float = 4.0, ls_init_value: Optional[float] = None,
                 output_dim: int = 512, no_causal_mask: bool = False, pool_type: str = "none", proj_bias: bool = False,
                 act_layer: Callable = nn.GELU, norm_layer: Callable = nn.LayerNorm, output_tokens: bool = False,
                 use_ln_post: bool = True, use_act_checkpoint: bool = False):
        super().__init__()
        assert pool_type in ("first", "last", "argmax", "none")
        self.output_tokens, self.pool_type = output_tokens, pool_type
        self.num_pos = self.context_length = context_length
        self.vocab_size, self.width, self.output_dim, self.heads = vocab_size, width, output_dim, heads
        self.token_embedding = nn.Embedding(vocab_size, width)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, width))
        self.transformer = Transformer(width, layers, heads, mlp_ratio, ls_init_value, act_layer, norm_layer,
                                       use_act_checkpoint=use_act_checkpoint)
        self.ln_final = norm_layer(width) if use_ln_post else nn.Identity()
        if no_causal_mask:
            self.attn_mask = None
        else:
            causal = torch.full((context_length, context_length), float("-inf")).triu_(1)
            self.register_buffer("attn_mask", causal, persistent=False)
        if proj_bias:
            self.text_projection = nn.Linear(width, output_dim)
        else:
            self.text_projection = nn.Parameter(torch.empty(width, output_dim))
        # the reference leaves these two uninitialised (filled by the checkpoint only): SURVEY F8
        nn.init.normal_(self.positional_embedding, std=0.01)
        if isinstance(self.text_projection, nn.Parameter):
            nn.init.normal_(self.text_projection, std=width ** -0.5)

    def forward(self, text: torch.Tensor):
        n = text.shape[1]
        x = self.token_embedding(text) + self.positional_embedding[:n]
        mask = self.attn_mask[:n, :n] if self.attn_mask is not None else None
        x = self.ln_final(self.transformer(x, attn_mask=mask))
        if self.pool_type == "first":
            pooled, tokens = x[:, 0], x[:, 1:]
        elif self.pool_type == "last":
            pooled, tokens = x[:, -1], x[:, :-1]
        elif self.pool_type == "argmax":
            pooled, tokens = x[torch.arange(x.shape[0]), text.argmax(dim=-1)], x
        else:
            pooled = tokens = x
        if isinstance(self.text_projection, nn.Linear):
            pooled = self.text_projection(pooled)
        elif self.text_projection is not None:
            pooled = pooled @ self.text_projection
        return (pooled, tokens) if self.output_tokens else pooled


class VETextEncoder(nn.Module):
    """``forward(list_of_str, device=) -> (padding_mask[B, L] (True = pad), memory[L, B, d_model], embeds[L, B, width])``
    -- text_encoder_ve.py:255-328: tokenise to ``context_length`` = 32, run the tower, ``resizer`` to the DETR width."""

    def __init__(self, d_model: int, tokenizer: Callable, width: int = 1024, heads: int = 16, layers: int = 24,
                 context_length: int = 32, vocab_size: int = 49408, use_ln_post: bool = True,
                 use_act_checkpoint: bool = True):
        super().__init__()
        self.context_length, self.use_ln_post, self.tokenizer = context_length, use_ln_post, tokenizer
        self.encoder = TextTransformer(context_length=context_length, vocab_size=vocab_size, width=width, heads=heads,
                                       layers=layers, output_tokens=True, use_ln_post=use_ln_post,
                                       use_act_checkpoint=use_act_checkpoint)
        self.resizer = nn.Linear(width, d_model)

    def forward(self, text, input_boxes=None, device=None):
        if not isinstance(text[0], str):                    # (mask, memory, {"inputs_embeds": ...}) already encoded
            mask, memory, tokenized = text
            return mask, memory, tokenized["inputs_embeds"].transpose(0, 1)
        assert input_boxes is None or len(input_boxes) == 0, "not supported"
        ids = self.tokenizer(text, context_length=self.context_length).to(device)
        embeds = self.encoder.token_embedding(ids)
        _, tokens = self.encoder(ids)
        memory = self.resizer(tokens.transpose(0, 1))
        return ids == 0, memory, embeds.transpose(0, 1)
