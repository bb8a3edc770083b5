"""Optimus text VAE behind the reference's registry names (lib/model_zoo/optimus.py:16-763 there).

DECODE side -- `optimus_vae_next.decode(z)` (reference :748-763): the 768-d text latent the 0-D diffuser
produces conditions a 12-layer GPT-2 (`optimus_gpt2_connector`, reference optimus_models/optimus_gpt2.py:813-1100) in two
ways, as an embedding added to every token (`linear_emb`) and as one extra key/value "memory" slot per layer (`linear`),
and tokens are sampled one at a time (multinomial, temperature 1, <= 30 tokens).  The network runs on the HIP kernel
library: fused q/k/v GEMM, vd_attention_f16 over a growing K/V cache (the reference re-runs the whole prefix every token),
LayerNorm, the tanh-GELU MLP in the GEMM epilogue, and the tied lm_head as one fp32-output GEMM.  Parameter names, shapes
(GPT-2's transposed Conv1D layout) and buffers equal the reference's, so `vae.text.decoder.*` checkpoint tensors load.
Token sampling itself (softmax on the device, then torch.multinomial) consumes the default generator exactly like the
reference does.

ENCODE side -- `optimus_vae_next.encode(text)` (reference :729-744): WordPiece ids -> 12-layer BERT
(`optimus_bert_connector`, reference optimus_models/optimus_bert.py:1348-1439) -> pooled [CLS] -> `linear` -> the mean half
of (mu, logvar) is the text latent.  Same kernels: embedding gather, LayerNorm, fused q/k/v GEMM, vd_attention_f16 with the
padded keys simply left out (the reference adds -10000 to their scores, which is exp(.) = 0 in its fp32 softmax), dense +
residual in the GEMM epilogue, erf-GELU / tanh as vd_unary_f16.  Both tokenizers are re-implemented on the host from the
published algorithms (byte-level BPE; BERT basic + WordPiece); the vocabulary files are the published ones, looked up at
the paths the reference's config names.
"""
import json
import math
import os
import re
import unicodedata

import torch
import torch.nn as nn

from vd_hip import ops

from .common.get_model import get_model, register
from .hip_layers import PackCache, _h

symbol = "optimus"


# ---- BERT tokenizer: basic (whitespace / punctuation / CJK) splitting + greedy longest-match WordPiece ----------------------
def _is_space(ch):
    return ch in " \t\n\r" or unicodedata.category(ch) == "Zs"


def _is_ctrl(ch):
    return ch not in "\t\n\r" and unicodedata.category(ch).startswith("C")


def _is_punct(ch):
    cp = ord(ch)
    ascii_punct = 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126   # all non-alphanumeric ASCII
    return ascii_punct or unicodedata.category(ch).startswith("P")


_CJK_RANGES = ((0x4E00, 0x9FFF), (0x3400, 0x4DBF), (0x20000, 0x2A6DF), (0x2A700, 0x2B73F), (0x2B740, 0x2B81F),
               (0x2B820, 0x2CEAF), (0xF900, 0xFAFF), (0x2F800, 0x2FA1F))


def _is_cjk(ch):
    cp = ord(ch)
    return any(lo <= cp <= hi for lo, hi in _CJK_RANGES)


@register("optimus_bert_tokenizer")
class optimus_bert_tokenizer(nn.Module):
    """BERT tokenizer with the surface `optimus_vae_next.encode` uses (reference optimus_models/tokenization_bert.py:101-457,
    tokenization_utils.py:576-640): `tokenize`, `_convert_token_to_id`, `convert_tokens_to_ids`,
    `add_special_tokens_single_sentence`, `encode`, the `*_token` / `*_token_id` attributes.  The algorithm is the
    published one (Devlin et al., `tokenization.py`): drop control characters, map unicode spaces to ' ', isolate CJK
    characters, split on whitespace, (lower-case + strip accents when do_lower_case), split every punctuation character
    off, then per word the greedy longest-prefix match against the vocabulary with '##' continuation pieces; words over
    100 characters or with an unmatched remainder become [UNK].  Special tokens inside the text are kept whole.
    One deliberate difference: a whitespace-only string gives [] (the reference's added-token splitter emits one
    special token picked from a `set`, i.e. depending on the process's hash seed)."""

    def __init__(self, vocab_file=None, do_lower_case=True, do_basic_tokenize=True, never_split=None, unk_token="[UNK]",
                 sep_token="[SEP]", pad_token="[PAD]", cls_token="[CLS]", mask_token="[MASK]", tokenize_chinese_chars=True,
                 max_len=512, **kwargs):
        super().__init__()
        self.vocab = {}
        if vocab_file is not None and os.path.exists(vocab_file):
            with open(vocab_file, encoding="utf-8") as f:
                for i, line in enumerate(f):
                    self.vocab[line.rstrip("\n")] = i
        self.vocab_file = vocab_file
        self.ids_to_tokens = {i: t for t, i in self.vocab.items()}
        self.do_lower_case, self.do_basic_tokenize = do_lower_case, do_basic_tokenize
        self.never_split = list(never_split or [])
        self.tokenize_chinese_chars = tokenize_chinese_chars
        self.unk_token, self.sep_token, self.pad_token = unk_token, sep_token, pad_token
        self.cls_token, self.mask_token = cls_token, mask_token
        self.max_len = max_len
        self.max_len_single_sentence = max_len - 2
        self.max_input_chars_per_word = 100

    def _need_vocab(self):
        if not self.vocab:
            raise FileNotFoundError("optimus_bert_tokenizer: vocabulary file %r not found (the published "
                                    "bert-base-cased-vocab.txt; path is CWD-relative like in the reference)" % (self.vocab_file,))

    @property
    def all_special_tokens(self):
        return [self.unk_token, self.sep_token, self.pad_token, self.cls_token, self.mask_token]

    @property
    def vocab_size(self):
        return len(self.vocab)

    def __len__(self):
        return len(self.vocab)

    cls_token_id = property(lambda self: self._convert_token_to_id(self.cls_token))
    sep_token_id = property(lambda self: self._convert_token_to_id(self.sep_token))
    pad_token_id = property(lambda self: self._convert_token_to_id(self.pad_token))
    unk_token_id = property(lambda self: self._convert_token_to_id(self.unk_token))

    # -- basic tokenisation of one piece of text that holds no special tokens
    def _words(self, text):
        cleaned = []
        for ch in text:
            if ord(ch) in (0, 0xFFFD) or _is_ctrl(ch):
                continue
            if _is_space(ch):
                cleaned.append(" ")
            elif self.tokenize_chinese_chars and _is_cjk(ch):
                cleaned.append(" " + ch + " ")
            else:
                cleaned.append(ch)
        words = []
        for word in "".join(cleaned).split():
            if self.do_lower_case and word not in self.never_split:
                word = "".join(c for c in unicodedata.normalize("NFD", word.lower()) if unicodedata.category(c) != "Mn")
            run = ""
            for ch in word:                       # every punctuation character is a word of its own
                if _is_punct(ch):
                    if run:
                        words.append(run)
                    words.append(ch)
                    run = ""
                else:
                    run += ch
            if run:
                words.append(run)
        return words

    def _wordpiece(self, word):
        if len(word) > self.max_input_chars_per_word:
            return [self.unk_token]
        pieces, start = [], 0
        while start < len(word):
            end = len(word)
            while end > start:
                cand = ("##" if start else "") + word[start:end]
                if cand in self.vocab:
                    break
                end -= 1
            if end == start:                      # nothing matched: the whole word is unknown
                return [self.unk_token]
            pieces.append(cand)
            start = end
        return pieces

    def tokenize(self, text):
        self._need_vocab()
        specials = sorted(set(self.all_special_tokens), key=len, reverse=True)
        parts = re.split("(" + "|".join(re.escape(t) for t in specials) + ")", text)
        out = []
        for part in parts:
            if part in specials:
                out.append(part)
                continue
            part = part.strip()
            if not part:
                continue
            words = self._words(part) if self.do_basic_tokenize else part.split()
            for w in words:
                out.extend(self._wordpiece(w))
        return out

    def _convert_token_to_id(self, token):
        self._need_vocab()
        return self.vocab.get(token, self.vocab.get(self.unk_token))

    def _convert_id_to_token(self, index):
        return self.ids_to_tokens.get(index, self.unk_token)

    def convert_tokens_to_ids(self, tokens):
        if isinstance(tokens, str):
            return self._convert_token_to_id(tokens)
        return [self._convert_token_to_id(t) for t in tokens]

    def convert_tokens_to_string(self, tokens):
        return " ".join(tokens).replace(" ##", "").strip()

    def add_special_tokens_single_sentence(self, token_ids):
        return [self.cls_token_id] + list(token_ids) + [self.sep_token_id]

    def encode(self, text, add_special_tokens=False):
        ids = self.convert_tokens_to_ids(self.tokenize(text))
        return self.add_special_tokens_single_sentence(ids) if add_special_tokens else ids


# ---- BERT encoder with the latent head ---------------------------------------------------------------------------------------
class _BertEmbeddings(nn.Module):
    def __init__(self, c):
        super().__init__()
        H = _cfg(c, "hidden_size")
        self.word_embeddings = nn.Embedding(_cfg(c, "vocab_size"), H, padding_idx=0)
        self.position_embeddings = nn.Embedding(_cfg(c, "max_position_embeddings"), H)
        self.token_type_embeddings = nn.Embedding(_cfg(c, "type_vocab_size", 2), H)
        self.LayerNorm = nn.LayerNorm(H, eps=_cfg(c, "layer_norm_eps", 1e-12))


class _BertSelfAttention(nn.Module):
    def __init__(self, c):
        super().__init__()
        H = _cfg(c, "hidden_size")
        self.query, self.key, self.value = nn.Linear(H, H), nn.Linear(H, H), nn.Linear(H, H)


class _BertDenseLN(nn.Module):          # BertSelfOutput / BertOutput: dense -> LayerNorm(. + input)
    def __init__(self, n_in, H, eps):
        super().__init__()
        self.dense = nn.Linear(n_in, H)
        self.LayerNorm = nn.LayerNorm(H, eps=eps)


class _BertAttention(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.self = _BertSelfAttention(c)
        self.output = _BertDenseLN(_cfg(c, "hidden_size"), _cfg(c, "hidden_size"), _cfg(c, "layer_norm_eps", 1e-12))


class _BertIntermediate(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(_cfg(c, "hidden_size"), _cfg(c, "intermediate_size"))


class _BertLayer(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.attention = _BertAttention(c)
        self.intermediate = _BertIntermediate(c)
        self.output = _BertDenseLN(_cfg(c, "intermediate_size"), _cfg(c, "hidden_size"), _cfg(c, "layer_norm_eps", 1e-12))


class _BertEncoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.layer = nn.ModuleList([_BertLayer(c) for _ in range(_cfg(c, "num_hidden_layers"))])


class _BertPooler(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(_cfg(c, "hidden_size"), _cfg(c, "hidden_size"))


@register("optimus_bert_connector")
class optimus_bert_connector(nn.Module, PackCache):
    """BertForLatentConnector (reference optimus_models/optimus_bert.py:1348-1439): BERT + `linear` [2*latent, hidden] (no
    bias) that maps the pooled output to (mu, logvar).  `forward(input_ids, attention_mask)` returns
    `(sequence_output [B, L, H], pooled_output [B, H])` as fp16 device tensors.  Masked (padding) keys are left out of the
    attention -- every query row, padded ones included, still attends over the real tokens exactly as in the reference --
    so samples run one at a time with their own key count.  Only the hidden_act = "gelu" (erf) configuration of the VD
    checkpoint is built."""

    def __init__(self, config, latent_size):
        super().__init__()
        assert _cfg(config, "hidden_act", "gelu") == "gelu", "optimus_bert_connector: only hidden_act='gelu' is built"
        self.config = config
        self.embeddings = _BertEmbeddings(config)
        self.encoder = _BertEncoder(config)
        self.pooler = _BertPooler(config)
        self.linear = nn.Linear(_cfg(config, "hidden_size"), 2 * latent_size, bias=False)
        self.n_head = _cfg(config, "num_attention_heads")
        self.hidden = _cfg(config, "hidden_size")
        std = _cfg(config, "initializer_range", 0.02)
        for m in self.modules():
            if isinstance(m, (nn.Linear, nn.Embedding)):
                m.weight.data.normal_(mean=0.0, std=std)
            if isinstance(m, nn.Linear) and m.bias is not None:
                m.bias.data.zero_()

    def _w(self):
        params = list(self.parameters())

        def build():
            e = self.embeddings
            layers = []
            for l in self.encoder.layer:
                a = l.attention
                layers.append(dict(
                    qkv=(_h(torch.cat([a.self.query.weight, a.self.key.weight, a.self.value.weight], 0)),
                         _h(torch.cat([a.self.query.bias, a.self.key.bias, a.self.value.bias], 0))),
                    ao=(_h(a.output.dense.weight), _h(a.output.dense.bias)),
                    aln=(_h(a.output.LayerNorm.weight), _h(a.output.LayerNorm.bias), a.output.LayerNorm.eps),
                    fc=(_h(l.intermediate.dense.weight), _h(l.intermediate.dense.bias)),
                    out=(_h(l.output.dense.weight), _h(l.output.dense.bias)),
                    oln=(_h(l.output.LayerNorm.weight), _h(l.output.LayerNorm.bias), l.output.LayerNorm.eps)))
            return dict(word=_h(e.word_embeddings.weight), pos=_h(e.position_embeddings.weight),
                        type=_h(e.token_type_embeddings.weight), eln=(_h(e.LayerNorm.weight), _h(e.LayerNorm.bias), e.LayerNorm.eps),
                        layers=layers, pool=(_h(self.pooler.dense.weight), _h(self.pooler.dense.bias)), lin=_h(self.linear.weight))
        return self._packed("bert", tuple(params), build)

    def _encode_one(self, w, ids, keep, type_ids, pos_ids):
        """ids [L] long; keep [L] bool (attended keys); -> hidden states [L, H] of ALL L rows."""
        L, H, dev = ids.shape[0], self.hidden, ids.device
        order = torch.cat([keep.nonzero().view(-1), (~keep).nonzero().view(-1)])   # attended rows first (row-wise ops commute)
        nk = int(keep.sum())
        if nk == 0:
            raise ValueError("optimus_bert_connector: a sample with no attended token")
        ids, type_ids, pos_ids = ids[order], type_ids[order], pos_ids[order]
        zero = torch.zeros((L, H), dtype=torch.float16, device=dev)
        x = ops.embed_tokens(ids.view(1, L), w["word"], w["pos"][pos_ids].contiguous()).view(L, H)
        x = ops.axpby(x, ops.embed_tokens(type_ids.view(1, L), w["type"], zero).view(L, H), 1.0, 1.0)
        x = ops.layernorm(x, w["eln"][0], w["eln"][1], w["eln"][2])
        for lw in w["layers"]:
            qkv = ops.linear(x, lw["qkv"][0], lw["qkv"][1]).view(1, L, 3 * H)
            a = ops.attention(qkv[..., :H], qkv[:, :nk, H:2 * H], qkv[:, :nk, 2 * H:], self.n_head, scale=(H // self.n_head) ** -0.5)
            x = ops.layernorm(ops.linear(a.view(L, H), lw["ao"][0], lw["ao"][1], res=x), lw["aln"][0], lw["aln"][1], lw["aln"][2])
            f = ops.unary(ops.linear(x, lw["fc"][0], lw["fc"][1]), ops.UNARY_GELU_ERF)
            x = ops.layernorm(ops.linear(f, lw["out"][0], lw["out"][1], res=x), lw["oln"][0], lw["oln"][1], lw["oln"][2])
        out = torch.empty_like(x)
        out[order] = x
        return out

    @torch.no_grad()
    def forward(self, input_ids, attention_mask=None, token_type_ids=None, position_ids=None, head_mask=None):
        assert head_mask is None, "optimus_bert_connector: head_mask is not supported"
        w = self._w()
        B, L = input_ids.shape
        dev = self.linear.weight.device
        input_ids = input_ids.to(dev)
        keep = torch.ones((B, L), dtype=torch.bool, device=dev) if attention_mask is None else (attention_mask.to(dev) > 0.5)
        type_ids = torch.zeros_like(input_ids) if token_type_ids is None else token_type_ids.to(dev)
        pos_ids = torch.arange(L, device=dev).expand(B, L) if position_ids is None else position_ids.to(dev).expand(B, L)
        seq = torch.stack([self._encode_one(w, input_ids[b], keep[b], type_ids[b], pos_ids[b]) for b in range(B)])
        pooled = ops.unary(ops.linear(seq[:, 0].contiguous(), w["pool"][0], w["pool"][1]), ops.UNARY_TANH)
        return seq, pooled

    @torch.no_grad()
    def latent_stats(self, pooled):
        """(mu, logvar) = chunk(linear(pooled), 2)  (reference optimus.py:742)."""
        ml = ops.linear(pooled.contiguous(), self._w()["lin"])
        return ml.chunk(2, -1)


# ---- GPT-2 byte-level BPE tokenizer (host-side text processing) ---------------------------------------------------------
def _byte_table():
    """The reversible byte <-> printable-unicode table of GPT-2's BPE (Radford et al., `encoder.py`; reference
    optimus_models/tokenization_gpt2.py:65-86): printable latin-1 bytes map to themselves, the rest to U+0100..."""
    keep = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAD)) + list(range(0xAE, 0x100))
    table, extra = {}, 0
    for b in range(256):
        if b in keep:
            table[b] = chr(b)
        else:
            table[b] = chr(256 + extra)
            extra += 1
    return table


@register("optimus_gpt2_tokenizer")
class optimus_gpt2_tokenizer(nn.Module):
    """GPT-2 tokenizer with the reference's surface as far as decode() needs it: `add_special_tokens`, `encode`,
    `decode(ids, clean_up_tokenization_spaces)`, `convert_tokens_to_ids`, `eos_token` / `pad_token` / `bos_token`
    (reference optimus_models/tokenization_gpt2.py:96-228, tokenization_utils.py:733-815).  vocab_file / merges_file are
    the published GPT-2 files; paths are CWD-relative like every path in the reference's configs."""

    def __init__(self, vocab_file=None, merges_file=None, do_lower_case=False, max_len=1024, errors="replace", **kwargs):
        super().__init__()
        self.max_len = max_len
        self.errors = errors
        self.encoder, self.bpe_ranks = {}, {}
        if vocab_file is not None and os.path.exists(vocab_file):
            with open(vocab_file, encoding="utf-8") as f:
                self.encoder = json.load(f)
        if merges_file is not None and os.path.exists(merges_file):
            with open(merges_file, encoding="utf-8") as f:
                lines = f.read().split("\n")[1:-1]
            self.bpe_ranks = {tuple(l.split()): i for i, l in enumerate(lines)}
        self.decoder = {v: k for k, v in self.encoder.items()}
        self.byte_encoder = _byte_table()
        self.byte_decoder = {v: k for k, v in self.byte_encoder.items()}
        self.added_tokens_encoder, self.added_tokens_decoder = {}, {}
        self.bos_token = self.eos_token = self.unk_token = "<|endoftext|>"
        self.pad_token = None
        self._cache = {}

    def __len__(self):
        return len(self.encoder) + len(self.added_tokens_encoder)

    def add_special_tokens(self, mapping):
        """{'pad_token': '<PAD>', 'bos_token': '<BOS>', 'eos_token': '<EOS>'} -> new ids appended after the vocabulary in
        the dict's order (reference optimus.py:30-34: 50257 / 50258 / 50259 for the published vocabulary)."""
        added = 0
        for key, tok in mapping.items():
            setattr(self, key, tok)
            if tok not in self.encoder and tok not in self.added_tokens_encoder:
                idx = len(self)
                self.added_tokens_encoder[tok] = idx
                self.added_tokens_decoder[idx] = tok
                added += 1
        return added

    def convert_tokens_to_ids(self, tokens):
        one = isinstance(tokens, str)
        out = [self.added_tokens_encoder.get(t, self.encoder.get(t, self.encoder.get(self.unk_token))) for t in ([tokens] if one else tokens)]
        return out[0] if one else out

    def _bpe(self, token):
        """Greedy byte-pair merging: repeatedly fuse the adjacent pair with the lowest merge rank."""
        if token in self._cache:
            return self._cache[token]
        word = list(token)
        while len(word) > 1:
            best = min(zip(word, word[1:]), key=lambda pr: self.bpe_ranks.get(pr, math.inf))
            if best not in self.bpe_ranks:
                break
            merged, i = [], 0
            while i < len(word):
                if i + 1 < len(word) and (word[i], word[i + 1]) == best:
                    merged.append(word[i] + word[i + 1])
                    i += 2
                else:
                    merged.append(word[i])
                    i += 1
            word = merged
        self._cache[token] = word
        return word

    def tokenize(self, text):
        import regex
        pat = regex.compile(r"""'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+""")
        out = []
        for piece in regex.findall(pat, " " + text):   # the reference prepends one space (tokenization_gpt2.py:178)
            out.extend(self._bpe("".join(self.byte_encoder[b] for b in piece.encode("utf-8"))))
        return out

    def encode(self, text):
        """text -> ids; added (special) tokens inside the text are matched whole, the rest is BPE'd."""
        ids, rest = [], [text]
        for tok in self.added_tokens_encoder:
            nxt = []
            for seg in rest:
                if isinstance(seg, int):
                    nxt.append(seg)
                    continue
                parts = seg.split(tok)
                for i, ptxt in enumerate(parts):
                    if ptxt:
                        nxt.append(ptxt)
                    if i + 1 < len(parts):
                        nxt.append(self.added_tokens_encoder[tok])
            rest = nxt
        for seg in rest:
            if isinstance(seg, int):
                ids.append(seg)
            elif seg.strip():   # segments are stripped like the reference's split_on_token does (tokenization_utils.py)
                ids.extend(self.encoder.get(t, self.encoder.get(self.unk_token)) for t in self.tokenize(seg.strip()))
        return ids

    def decode(self, token_ids, skip_special_tokens=False, clean_up_tokenization_spaces=True):
        """ids -> text: byte-level tokens are joined and byte-decoded, added tokens are spliced in as ' <TOKEN>'
        (reference tokenization_utils.py:733-773), then the English clean-up of :809-815."""
        pieces, cur = [], []

        def flush():
            if cur:
                raw = "".join(cur)
                pieces.append(bytearray(self.byte_decoder[c] for c in raw).decode("utf-8", errors=self.errors))
                del cur[:]
        for i in token_ids:
            i = int(i)
            if i in self.added_tokens_decoder:
                if skip_special_tokens:
                    continue
                flush()
                pieces.append(" " + self.added_tokens_decoder[i])
            else:
                cur.append(self.decoder[i])
        flush()
        text = "".join(pieces)
        if clean_up_tokenization_spaces:
            for a, b in ((" .", "."), (" ?", "?"), (" !", "!"), (" ,", ","), (" ' ", "'"), (" n't", "n't"), (" 'm", "'m"),
                         (" do not", " don't"), (" 's", "'s"), (" 've", "'ve"), (" 're", "'re")):
                text = text.replace(a, b)
        return text


# ---- GPT-2 decoder conditioned on the latent ------------------------------------------------------------------------------
def _cfg(config, key, default=None):
    if isinstance(config, dict):
        return config.get(key, default)
    return getattr(config, key, default)


class Conv1D(nn.Module):
    """GPT-2's transposed dense layer: weight [nx, nf], y = x @ weight + bias (reference modeling_utils.py:408-424)."""

    def __init__(self, nf, nx):
        super().__init__()
        self.nf = nf
        self.weight = nn.Parameter(torch.empty(nx, nf).normal_(std=0.02))
        self.bias = nn.Parameter(torch.zeros(nf))


class _Attention(nn.Module):
    def __init__(self, nx, n_ctx, n_head):
        super().__init__()
        self.register_buffer("bias", torch.tril(torch.ones(n_ctx, n_ctx)).view(1, 1, n_ctx, n_ctx))  # key compatibility
        self.n_head = n_head
        self.c_attn = Conv1D(nx * 3, nx)
        self.c_proj = Conv1D(nx, nx)


class _MLP(nn.Module):
    def __init__(self, n_state, nx):
        super().__init__()
        self.c_fc = Conv1D(n_state, nx)
        self.c_proj = Conv1D(nx, n_state)


class _Block(nn.Module):
    def __init__(self, n_ctx, nx, n_head, eps):
        super().__init__()
        self.ln_1 = nn.LayerNorm(nx, eps=eps)
        self.attn = _Attention(nx, n_ctx, n_head)
        self.ln_2 = nn.LayerNorm(nx, eps=eps)
        self.mlp = _MLP(4 * nx, nx)


class _GPT2Model(nn.Module):
    def __init__(self, config):
        super().__init__()
        nx, nl = _cfg(config, "n_embd"), _cfg(config, "n_layer")
        self.wte = nn.Embedding(_cfg(config, "vocab_size"), nx)
        self.wpe = nn.Embedding(_cfg(config, "n_positions"), nx)
        self.h = nn.ModuleList([_Block(_cfg(config, "n_ctx"), nx, _cfg(config, "n_head"), _cfg(config, "layer_norm_epsilon", 1e-5))
                                for _ in range(nl)])
        self.ln_f = nn.LayerNorm(nx, eps=_cfg(config, "layer_norm_epsilon", 1e-5))
        latent = _cfg(config, "latent_size", 32)
        self.linear = nn.Linear(latent, _cfg(config, "hidden_size", nx) * nl, bias=False)   # one memory vector per layer
        self.linear_emb = nn.Linear(latent, _cfg(config, "hidden_size", nx), bias=False)    # added to every embedding
        for m in self.modules():
            if isinstance(m, (nn.Linear, nn.Embedding)):
                m.weight.data.normal_(mean=0.0, std=_cfg(config, "initializer_range", 0.02))


@register("optimus_gpt2_connector")
class optimus_gpt2_connector(nn.Module, PackCache):
    """GPT2ForLatentConnector (reference optimus_models/optimus_gpt2.py:1025-1100): `transformer` + `lm_head` tied to
    `transformer.wte`.  `logits(ids, z)` is the teacher-forced forward of the whole sequence; `generate(z, ...)` is the
    sampling loop of optimus.py:662-688 with a K/V cache."""

    def __init__(self, config, latent_size=32, latent_as_gpt_emb=True, latent_as_gpt_memory=True):
        super().__init__()
        assert latent_as_gpt_emb and latent_as_gpt_memory, "the VD checkpoints use the latent both ways"
        self.config = config
        self.transformer = _GPT2Model(config)
        self.lm_head = nn.Linear(_cfg(config, "n_embd"), _cfg(config, "vocab_size"), bias=False)
        self.lm_head.weight = self.transformer.wte.weight   # tie_weights
        self.n_head = _cfg(config, "n_head")
        self.n_embd = _cfg(config, "n_embd")
        self.n_layer = _cfg(config, "n_layer")

    # ---- kernel-layout weights (fp16, [out, in]); rebuilt when the parameters change
    def _w(self):
        t = self.transformer
        params = [t.wte.weight, t.wpe.weight, t.linear.weight, t.linear_emb.weight, t.ln_f.weight, t.ln_f.bias]
        for b in t.h:
            params += [b.ln_1.weight, b.ln_1.bias, b.attn.c_attn.weight, b.attn.c_attn.bias, b.attn.c_proj.weight,
                       b.attn.c_proj.bias, b.ln_2.weight, b.ln_2.bias, b.mlp.c_fc.weight, b.mlp.c_fc.bias,
                       b.mlp.c_proj.weight, b.mlp.c_proj.bias]

        def build():
            tr = lambda w: _h(w).t().contiguous()     # Conv1D [in, out] -> [out, in]
            blocks = [dict(ln1=(_h(b.ln_1.weight), _h(b.ln_1.bias)), qkv=(tr(b.attn.c_attn.weight), _h(b.attn.c_attn.bias)),
                           proj=(tr(b.attn.c_proj.weight), _h(b.attn.c_proj.bias)), ln2=(_h(b.ln_2.weight), _h(b.ln_2.bias)),
                           fc=(tr(b.mlp.c_fc.weight), _h(b.mlp.c_fc.bias)), out=(tr(b.mlp.c_proj.weight), _h(b.mlp.c_proj.bias)))
                      for b in t.h]
            return dict(wte=_h(t.wte.weight), wpe=_h(t.wpe.weight), lin=_h(t.linear.weight), lin_emb=_h(t.linear_emb.weight),
                        lnf=(_h(t.ln_f.weight), _h(t.ln_f.bias)), blocks=blocks)
        return self._packed("gpt2", tuple(params), build)

    def _latent(self, w, z):
        """z [latent] -> (embedding offset [1, E], per-layer memory [n_layer, E])."""
        z2 = z.to(torch.float16).reshape(1, -1).contiguous()
        return ops.linear(z2, w["lin_emb"]), ops.linear(z2, w["lin"]).view(self.n_layer, self.n_embd)

    def _embed(self, w, ids, pos0, pe):
        """token + position embeddings (positions start at 1: slot 0 is the latent memory) + latent embedding."""
        T = ids.shape[0]
        x = ops.embed_tokens(ids.view(1, T), w["wte"], w["wpe"][1 + pos0:1 + pos0 + T])
        return ops.axpby(x.view(T, -1), pe.expand(T, -1).contiguous(), 1.0, 1.0)

    def _block(self, bw, x, kv, n_new, eps1, eps2):
        """x [T, E] (the new rows); kv [1, Nk, 2E] cache whose last n_new rows are written here."""
        E, T = self.n_embd, x.shape[0]
        h = ops.layernorm(x, bw["ln1"][0], bw["ln1"][1], eps1)
        qkv = ops.linear(h, bw["qkv"][0], bw["qkv"][1])
        Nk = kv.shape[1]
        kv[0, Nk - n_new:].copy_(qkv[:, E:])
        # causal = 2: one memory slot in front of the keys, query i sees keys 0 .. (Nk - T) + i
        a = ops.attention(qkv[:, :E].reshape(1, T, E), kv[..., :E], kv[..., E:], self.n_head, scale=(E // self.n_head) ** -0.5,
                          causal=(Nk - T + 1) if T > 1 else 0)
        x = ops.linear(a.view(T, E), bw["proj"][0], bw["proj"][1], res=x)
        h = ops.layernorm(x, bw["ln2"][0], bw["ln2"][1], eps2)
        f = ops.linear(h, bw["fc"][0], bw["fc"][1], act=ops.ACT_GELU_TANH)
        return ops.linear(f, bw["out"][0], bw["out"][1], res=x)

    @torch.no_grad()
    def logits(self, ids, z):
        """Teacher-forced forward: ids [T] (long, on the device), z [latent] -> fp32 logits [T, vocab]."""
        w = self._w()
        T, E = ids.shape[0], self.n_embd
        pe, mem = self._latent(w, z)
        x = self._embed(w, ids, 0, pe)
        for l, (bw, blk) in enumerate(zip(w["blocks"], self.transformer.h)):
            kv = torch.empty((1, T + 1, 2 * E), dtype=torch.float16, device=x.device)
            kv[0, 0, :E] = mem[l]
            kv[0, 0, E:] = mem[l]
            x = self._block(bw, x, kv, T, blk.ln_1.eps, blk.ln_2.eps)
        x = ops.layernorm(x, w["lnf"][0], w["lnf"][1], self.transformer.ln_f.eps)
        return ops.gemm(x, w["wte"], out_f32=True)

    @torch.no_grad()
    def generate(self, z, context, eos_token, max_length=30, temperature=1.0, top_k=0, top_p=1.0):
        """sample_single_sequence_conditional (reference optimus.py:662-688) with a K/V cache: the reference re-runs the
        whole prefix for every new token; the values are the same, the cost is one row per token.  Returns the token ids
        including the context and the closing eos (forced at max_length like the reference does)."""
        w = self._w()
        E, dev = self.n_embd, z.device
        pe, mem = self._latent(w, z)
        caches = []
        for l in range(self.n_layer):
            kv = torch.empty((1, max_length + 1, 2 * E), dtype=torch.float16, device=dev)
            kv[0, 0, :E] = mem[l]
            kv[0, 0, E:] = mem[l]
            caches.append(kv)
        generated = [int(t) for t in context.tolist()]
        new = context.to(dev).view(-1)
        pos = 0
        while True:
            T = new.shape[0]
            x = self._embed(w, new, pos, pe)
            for l, (bw, blk) in enumerate(zip(w["blocks"], self.transformer.h)):
                x = self._block(bw, x, caches[l][:, :1 + pos + T], T, blk.ln_1.eps, blk.ln_2.eps)
            pos += T
            x = ops.layernorm(x[-1:].contiguous(), w["lnf"][0], w["lnf"][1], self.transformer.ln_f.eps)
            lg = ops.gemm(x, w["wte"], out_f32=True)                      # [1, vocab] fp32
            probs = ops.softmax_rows_f32(lg, scale=1.0 / float(temperature)).view(-1)
            probs = top_k_top_p_filtering(probs, top_k=top_k, top_p=top_p)
            nxt = torch.multinomial(probs, num_samples=1)                 # the reference's draw (optimus.py:681)
            tok = int(nxt.item())
            generated.append(tok)
            if tok == eos_token:
                break
            if len(generated) >= max_length:
                generated[-1] = eos_token
                break
            new = nxt.view(1)
        return torch.tensor(generated, dtype=torch.long)


def top_k_top_p_filtering(probs, top_k=0, top_p=1.0):
    """The reference's logit filter (optimus.py:690-721) applied to the probability vector: tokens outside the top-k /
    outside the smallest set whose cumulative probability exceeds top_p get probability 0 (their logits -inf there).
    With the decode() defaults (top_k = 0, top_p = 1.0) nothing is removed -- the reference's `cumulative > 1.0` can only
    fire on round-off in the far tail -- so the call is skipped."""
    if top_k <= 0 and top_p >= 1.0:
        return probs
    p = probs.clone()
    if top_k > 0:
        kth = torch.topk(p, min(top_k, p.numel()))[0][-1]
        p[p < kth] = 0
    if top_p < 1.0:
        sp, si = torch.sort(p, descending=True)
        cum = torch.cumsum(sp / sp.sum(), dim=-1)
        rm = cum > top_p
        rm[1:] = rm[:-1].clone()
        rm[0] = False
        p[si[rm]] = 0
    return p


@register("optimus_vae")
@register("optimus_vae_next")
class optimus_vae_next(nn.Module):
    """optimus_vae / optimus_vae_next (reference optimus.py:17-60, 724-763): holds encoder, decoder and the two
    tokenizers; `encode(text)` turns sentences into text latents (the posterior mean), `decode(z)` text latents into
    sentences.  The training-side methods of the reference class (loss, importance-weighted bounds, MI / AU statistics)
    are not part of the sampling path and are not built."""

    def __init__(self, encoder, decoder, tokenizer_encoder, tokenizer_decoder, args):
        super().__init__()
        build = lambda c, **kw: c if isinstance(c, nn.Module) else get_model()(c, **kw)
        self.encoder = build(encoder)
        self.decoder = build(decoder)
        self.tokenizer_encoder = build(tokenizer_encoder, verbose=False)
        self.tokenizer_decoder = build(tokenizer_decoder, verbose=False)
        special = {"pad_token": "<PAD>", "bos_token": "<BOS>", "eos_token": "<EOS>"}
        for tok in (self.tokenizer_encoder, self.tokenizer_decoder):
            if isinstance(tok, optimus_gpt2_tokenizer):
                tok.add_special_tokens(special)
        self.args = args
        self.nz = _cfg(args, "latent_size")
        if isinstance(self.tokenizer_decoder, optimus_gpt2_tokenizer):
            self.eos_token_id = self.tokenizer_decoder.convert_tokens_to_ids([self.tokenizer_decoder.eos_token])[0]
            self.pad_token_id = self.tokenizer_decoder.convert_tokens_to_ids([self.tokenizer_decoder.pad_token])[0]

    def get_device(self):
        return self.decoder.transformer.wte.weight.device

    @torch.no_grad()
    def encode(self, text, max_length=77):
        """sentences -> text latent z_mu [B, nz] (reference optimus.py:729-744): lower-case, WordPiece, truncate to
        max_length pieces, [CLS] .. [SEP], right-pad with 0, BERT with mask = (id > 0), pooled -> linear -> mean half."""
        tok = self.tokenizer_encoder
        rows = []
        for sentence in text:
            pieces = tok.tokenize(sentence.lower())[:max_length]
            rows.append(tok.add_special_tokens_single_sentence([tok._convert_token_to_id(p) for p in pieces]))
        L = max(len(r) for r in rows)
        ids = torch.zeros((len(rows), L), dtype=torch.long)
        for i, r in enumerate(rows):
            ids[i, :len(r)] = torch.tensor(r, dtype=torch.long)
        ids = ids.to(self.encoder.linear.weight.device)
        pooled = self.encoder(ids, attention_mask=(ids > 0).float())[1]
        z_mu, _ = self.encoder.latent_stats(pooled)
        return z_mu

    @torch.no_grad()
    def decode(self, z, temperature=1.0):
        tok = self.tokenizer_decoder
        bos, eos = tok.encode("<BOS>"), tok.encode("<EOS>")
        context = torch.LongTensor(bos).to(z.device)
        sentences = []
        for zi in z:
            out = self.decoder.generate(zi, context, eos_token=eos[0], max_length=30, temperature=temperature, top_k=0, top_p=1.0)
            text = tok.decode(out.tolist(), clean_up_tokenization_spaces=True)
            sentences.append(" ".join(text.split()[1:-1]))
        return sentences
