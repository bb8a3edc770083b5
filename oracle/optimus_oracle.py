"""TEST INFRASTRUCTURE (oracle): CPU fp32 restatement of the Optimus GPT-2 decoder forward and of the decode loop --
/root/reference/lib/model_zoo/optimus_models/optimus_gpt2.py:99-246 (gelu, Attention, MLP, Block), :870-993
(GPT2Model_XX.forward with the latent as embedding AND per-layer memory), :1084-1100 (tied lm_head), and
/root/reference/lib/model_zoo/optimus.py:662-688 (sample_single_sequence_conditional).  Never imported by the product.

Pinned by tests/golden/optimus_tiny.npz: oracle/gen_golden_optimus.py runs the REFERENCE's own vendored
GPT2ForLatentConnector_XX (imported from /root/reference) on a down-sized config with seeded synthetic weights."""
import math

import torch
import torch.nn.functional as F


def gelu(x):
    """optimus_gpt2.py:99-100"""
    return 0.5 * x * (1 + torch.tanh(math.sqrt(2 / math.pi) * (x + 0.044715 * torch.pow(x, 3))))


def _conv1d(sd, p, x):
    """modeling_utils.py:420-424: x @ weight + bias, weight [in, out]"""
    return x @ sd[p + ".weight"] + sd[p + ".bias"]


def gpt2_logits(sd, p, input_ids, z, n_head, n_layer, eps=1e-5):
    """GPT2ForLatentConnector_XX.forward(input_ids [B, T], past = z [B, latent]) -> logits [B, T, V].
    sd keys under `p` (e.g. 'decoder'): transformer.{wte,wpe,h.i...,ln_f,linear,linear_emb}, lm_head tied to wte."""
    t = p + ".transformer"
    B, T = input_ids.shape
    E = sd[t + ".wte.weight"].shape[1]
    past_emb = z @ sd[t + ".linear_emb.weight"].t()                       # :876
    mem = (z @ sd[t + ".linear.weight"].t()).view(B, n_layer, E)          # :879-889: one [B, 1, E] key = value per layer
    pos = torch.arange(1, T + 1)                                          # :900 past_length = 1
    h = sd[t + ".wte.weight"][input_ids] + sd[t + ".wpe.weight"][pos][None] + past_emb[:, None]   # :941-953
    D = E // n_head
    for i in range(n_layer):
        b = "%s.h.%d" % (t, i)
        x = F.layer_norm(h, (E,), sd[b + ".ln_1.weight"], sd[b + ".ln_1.bias"], eps)
        q, k, v = _conv1d(sd, b + ".attn.c_attn", x).split(E, dim=2)       # :182-183
        k = torch.cat([mem[:, i:i + 1], k], dim=1)                          # :189-196: memory slot in front
        v = torch.cat([mem[:, i:i + 1], v], dim=1)
        sh = lambda a: a.view(B, a.shape[1], n_head, D).permute(0, 2, 1, 3)
        w = torch.matmul(sh(q), sh(k).transpose(-1, -2)) / math.sqrt(D)   # :145-147
        nd, ns = w.shape[-2], w.shape[-1]
        mask = torch.tril(torch.ones(ns, ns))[ns - nd:ns, :ns]            # :149
        w = w * mask - 1e4 * (1 - mask)
        a = torch.matmul(torch.softmax(w, dim=-1), sh(v)).permute(0, 2, 1, 3).reshape(B, T, E)
        h = h + _conv1d(sd, b + ".attn.c_proj", a)                        # :241
        x = F.layer_norm(h, (E,), sd[b + ".ln_2.weight"], sd[b + ".ln_2.bias"], eps)
        h = h + _conv1d(sd, b + ".mlp.c_proj", gelu(_conv1d(sd, b + ".mlp.c_fc", x)))   # :242-243
    h = F.layer_norm(h, (E,), sd[t + ".ln_f.weight"], sd[t + ".ln_f.bias"], eps)
    return h @ sd[t + ".wte.weight"].t()                                    # lm_head tied to wte


def sample_sequence(sd, p, z, bos, eos, n_head, n_layer, max_length=30, temperature=1.0, generator=None):
    """optimus.py:662-688 with top_k = 0, top_p = 1.0 (the decode() defaults): full-prefix forward per token,
    multinomial on softmax(logits / temperature)."""
    generated = torch.tensor([[bos]], dtype=torch.long)
    while True:
        lg = gpt2_logits(sd, p, generated, z[None], n_head, n_layer)[0, -1] / temperature
        nxt = torch.multinomial(F.softmax(lg, dim=-1), num_samples=1, generator=generator)
        generated = torch.cat([generated, nxt[None]], dim=1)
        if int(nxt) == eos:
            break
        if generated.shape[1] >= max_length:
            generated[0, -1] = eos
            break
    return generated[0]


# ---- encode side: BERT with the latent head -------------------------------------------------------------------------------
def gelu_erf(x):
    """optimus_bert.py:122-128"""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def bert_forward(sd, p, input_ids, attention_mask, n_head, n_layer, eps=1e-12):
    """BertForLatentConnector_XX.forward (optimus_bert.py:1393-1439) -> (sequence_output [B, L, H], pooled_output [B, H]).
    sd keys under `p` (e.g. 'encoder'): embeddings.*, encoder.layer.i.*, pooler.dense.*, linear.weight."""
    lin = lambda q, x: x @ sd[q + ".weight"].t() + sd[q + ".bias"]
    ln = lambda q, x: F.layer_norm(x, (x.shape[-1],), sd[q + ".weight"], sd[q + ".bias"], eps)
    B, L = input_ids.shape
    e = p + ".embeddings"
    h = sd[e + ".word_embeddings.weight"][input_ids] + sd[e + ".position_embeddings.weight"][torch.arange(L)][None] \
        + sd[e + ".token_type_embeddings.weight"][torch.zeros_like(input_ids)]          # :158-172
    h = ln(e + ".LayerNorm", h)
    ext = (1.0 - attention_mask.float())[:, None, None, :] * -10000.0                   # :1404-1412
    H = h.shape[-1]
    D = H // n_head
    sh = lambda a: a.view(B, L, n_head, D).permute(0, 2, 1, 3)
    for i in range(n_layer):
        b = "%s.encoder.layer.%d" % (p, i)
        q, k, v = (sh(lin(b + ".attention.self." + n, h)) for n in ("query", "key", "value"))   # :200-227
        w = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(D) + ext, dim=-1)
        a = torch.matmul(w, v).permute(0, 2, 1, 3).reshape(B, L, H)
        h = ln(b + ".attention.output.LayerNorm", lin(b + ".attention.output.dense", a) + h)       # :236-247
        f = gelu_erf(lin(b + ".intermediate.dense", h))                                             # :289-299
        h = ln(b + ".output.LayerNorm", lin(b + ".output.dense", f) + h)                            # :302-313
    pooled = torch.tanh(lin(p + ".pooler.dense", h[:, 0]))                                          # :364-375
    return h, pooled


def bert_latent_mu(sd, p, pooled):
    """optimus.py:742-744: z_mu = first half of encoder.linear(pooled)"""
    return (pooled @ sd[p + ".linear.weight"].t()).chunk(2, -1)[0]


def top_k_top_p_filtering(logits, top_k=0, top_p=0.0):
    """optimus.py:690-721: logits of one step -> logits with the filtered tokens at -inf (top-k first, then the nucleus on
    what is left; the first token above the threshold is kept)."""
    logits = logits.clone()
    top_k = min(top_k, logits.size(-1))
    if top_k > 0:
        logits[logits < torch.topk(logits, top_k)[0][..., -1, None]] = -float("inf")
    if top_p > 0.0:
        sl, si = torch.sort(logits, descending=True)
        rm = torch.cumsum(F.softmax(sl, dim=-1), dim=-1) > top_p
        rm[..., 1:] = rm[..., :-1].clone()
        rm[..., 0] = False
        logits[si[rm]] = -float("inf")
    return logits
