"""Optimus text VAE behind the reference's registry names (lib/model_zoo/optimus.py:16-763 there).

Built here: the DECODE side -- `optimus_vae_next.decode(z)` (reference :748-763): the 768-d text latent the 0-D diffuser
produces conditions a 12-layer GPT-2 (`optimus_gpt2_connector`, reference optimus_models/optimus_gpt2.py:813-1100) in two
ways, as an embedding added to every token (`linear_emb`) and as one extra key/value "memory" slot per layer (`linear`),
and tokens are sampled one at a time (multinomial, temperature 1, <= 30 tokens).  The network runs on the HIP kernel
library: fused q/k/v GEMM, vd_attention_f16 over a growing K/V cache (the reference re-runs the whole prefix every token),
LayerNorm, the tanh-GELU MLP in the GEMM epilogue, and the tied lm_head as one fp32-output GEMM.  Parameter names, shapes
(GPT-2's transposed Conv1D layout) and buffers equal the reference's, so `vae.text.decoder.*` checkpoint tensors load.
Token sampling itself (softmax on the device, then torch.multinomial) consumes the default generator exactly like the
reference does.

Not built: the ENCODE side (BERT connector, text -> latent; text-variation / text-to-text flows).  Its names stay
resolvable (`optimus_bert_connector`, `optimus_bert_tokenizer`), checkpoints load with strict=False, `encode` raises.
"""
import json
import math
import os

import torch
import torch.nn as nn

from vd_hip import ops

from .common.get_model import get_model, register
from .hip_layers import PackCache, _h

symbol = "optimus"


class _NotBuilt(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()

    def _no(self, *a, **k):
        raise NotImplementedError("%s: the Optimus ENCODE side (BERT, text -> latent) is not built in this package; "
                                  "decode (latent -> text) is" % type(self).__name__)

    encode = forward = tokenize = _no


@register("optimus_bert_connector")
class optimus_bert_connector(_NotBuilt):
    pass


@register("optimus_bert_tokenizer")
class optimus_bert_tokenizer(_NotBuilt):
    pass


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
    tokenizers; `decode(z)` turns text latents into sentences."""

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

    def encode(self, text, max_length=77):
        raise NotImplementedError("optimus_vae_next.encode: the BERT encoder (text -> latent) is not built in this package")

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
