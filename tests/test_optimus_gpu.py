"""Optimus on the HIP path (lib/model_zoo/optimus.py): the decode side against the fixture the reference's vendored GPT-2
latent connector produced (tests/golden/optimus_tiny.npz) and the CPU oracle at full size (12 layers, vocab 50260); the
encode side against the fixture of the reference's BERT latent connector / optimus_vae_next.encode
(tests/golden/optimus_bert_tiny.npz) and the oracle at bert-base size."""
import json
import os

import numpy as np
import pytest
import torch

from vdtest_util import GOLD, load_gold, rel_l2

pytestmark = pytest.mark.gpu


def _build(cfg, seed, dev):
    from lib.model_zoo.optimus import optimus_gpt2_connector
    from oracle import synth
    net = optimus_gpt2_connector(cfg, latent_size=cfg["latent_size"])
    sd = synth.synth_state_dict({"decoder." + k: v for k, v in synth.shapes_of(net).items()}, seed)
    sd["decoder.lm_head.weight"] = sd["decoder.transformer.wte.weight"]
    net.load_state_dict({k[len("decoder."):]: v for k, v in sd.items()}, strict=True)
    net = net.half().to(dev)
    return net, sd


def test_tiny_gpt2_logits_vs_reference_fixture(dev):
    meta = json.load(open(os.path.join(GOLD, "optimus_tokenizer.json")))
    cfg = meta["config"]
    net, _ = _build(cfg, meta["seed"], dev)
    g = load_gold("optimus_tiny.npz")
    for b in range(2):
        lg = net.logits(torch.from_numpy(g["ids"][b]).to(dev), torch.from_numpy(g["z"][b]).to(dev))
        assert lg.dtype == torch.float32 and lg.shape == (9, cfg["vocab_size"])
        assert rel_l2(lg, g["logits"][b]) < 5e-3


def test_generate_with_kv_cache_matches_full_prefix_oracle(dev, monkeypatch):
    """Greedy decoding (torch.multinomial replaced by argmax on both sides): the K/V-cached token-by-token path must
    produce the sequence the oracle's full-prefix loop (the reference's algorithm, optimus.py:662-688) produces."""
    from oracle import optimus_oracle as OO
    meta = json.load(open(os.path.join(GOLD, "optimus_tokenizer.json")))
    cfg = meta["config"]
    net, sd = _build(cfg, meta["seed"], dev)
    g = load_gold("optimus_tiny.npz")
    monkeypatch.setattr(torch, "multinomial", lambda p, num_samples=1, **k: torch.argmax(p, dim=-1, keepdim=True))
    for b in range(2):
        z = torch.from_numpy(g["z"][b])
        ref = OO.sample_sequence(sd, "decoder", z, 5, 7, cfg["n_head"], cfg["n_layer"], max_length=14)
        out = net.generate(z.to(dev), torch.LongTensor([5]), eos_token=7, max_length=14)
        assert out.tolist() == ref.tolist()
        assert out[0] == 5 and out[-1] == 7 and len(out) <= 14


def test_full_size_gpt2_logits_vs_oracle(dev):
    """The decoder of optimus_v1 at full size (12 layers, 768 wide, 12 heads, vocabulary 50260, 124 M parameters)."""
    from lib.cfg_helper import model_cfg_bank
    from oracle import optimus_oracle as OO
    cfg = dict(model_cfg_bank()("optimus_gpt2_decoder").args.config)
    net, sd = _build(cfg, 11, dev)
    assert sum(p.numel() for p in net.parameters()) == 124439808 + 768 * 768 * 13 + 3 * 768   # GPT-2 small + 3 tokens + latent maps
    g = torch.Generator().manual_seed(12)
    ids = torch.randint(0, 50260, (10,), generator=g)
    z = torch.randn((768,), generator=g)
    with torch.no_grad():
        ref = OO.gpt2_logits(sd, "decoder", ids[None], z[None], 12, 12)[0]
    lg = net.logits(ids.to(dev), z.to(dev))
    assert rel_l2(lg, ref) < 5e-3


def test_optimus_vae_decode_end_to_end(dev, tmp_path):
    """optimus_vae_next.decode (reference optimus.py:748-763) on a byte-level toy vocabulary: sentences come back as
    strings, the BOS / EOS words are stripped, and the default generator drives the sampling (same seed, same text)."""
    from lib.cfg_helper import CfgDict
    from lib.model_zoo import get_model
    from oracle import synth
    from lib.model_zoo.optimus import _byte_table
    table = _byte_table()
    vocab = {table[b]: b for b in range(256)}
    (tmp_path / "vocab.json").write_text(json.dumps(vocab), encoding="utf-8")
    (tmp_path / "merges.txt").write_text("#version: 0.2\nĠ a\n", encoding="utf-8")   # header + one rank (never reachable: no merged token in the vocabulary)
    dcfg = dict(vocab_size=259, n_positions=40, n_ctx=40, n_embd=128, n_layer=2, n_head=2, latent_size=64, hidden_size=128,
                layer_norm_epsilon=1e-5, initializer_range=0.02)
    cfg = CfgDict(type="optimus_vae_next", args=CfgDict(
        encoder=CfgDict(type="optimus_bert_connector", args=CfgDict(config=_bert_meta()["config"], latent_size=64)),
        decoder=CfgDict(type="optimus_gpt2_connector", args=CfgDict(config=dcfg)),
        tokenizer_encoder=CfgDict(type="optimus_bert_tokenizer", args=CfgDict()),
        tokenizer_decoder=CfgDict(type="optimus_gpt2_tokenizer", args=CfgDict(vocab_file=str(tmp_path / "vocab.json"),
                                                                               merges_file=str(tmp_path / "merges.txt"))),
        args=CfgDict(latent_size=64)))
    vae = get_model()(cfg, verbose=False)
    synth.load_synth_(vae.decoder, 5, prefix="decoder.")
    vae = vae.half().to(dev)
    assert vae.eos_token_id == 258 and vae.pad_token_id == 256
    z = torch.randn((3, 64), generator=torch.Generator().manual_seed(1)).half().to(dev)
    torch.manual_seed(42)
    a = vae.decode(z)
    torch.manual_seed(42)
    b = vae.decode(z)
    assert a == b and len(a) == 3 and all(isinstance(s, str) for s in a)
    assert all("<BOS>" not in s and "<EOS>" not in s for s in a)
    with pytest.raises(FileNotFoundError):       # no BERT vocabulary configured: encode says so instead of guessing
        vae.encode(["a cat"])


# ---- encode side ------------------------------------------------------------------------------------------------------------
def _bert_meta():
    return json.load(open(os.path.join(GOLD, "optimus_bert_tokenizer.json")))


def _build_bert(cfg, seed, dev):
    from lib.model_zoo.optimus import optimus_bert_connector
    from oracle import synth
    net = optimus_bert_connector(cfg, latent_size=cfg["latent_size"])
    sd = synth.synth_state_dict({"encoder." + k: v for k, v in synth.shapes_of(net).items()}, seed)
    net.load_state_dict({k[len("encoder."):]: v for k, v in sd.items()}, strict=True)
    return net.half().to(dev), sd


def test_tiny_bert_vs_reference_fixture(dev):
    """Right-padded batch through the BERT latent connector: hidden states of ALL rows (padded query rows included),
    pooled output and the latent mean against what the reference's BertForLatentConnector_XX produced."""
    m = _bert_meta()
    net, _ = _build_bert(m["config"], m["seed"], dev)
    g = load_gold("optimus_bert_tiny.npz")
    ids = torch.from_numpy(g["ids"]).long().to(dev)
    seq, pooled = net(ids, attention_mask=(ids > 0).float())
    assert seq.shape == g["seq"].shape and seq.dtype == torch.float16
    assert rel_l2(seq, g["seq"]) < 5e-3 and rel_l2(pooled, g["pooled"]) < 5e-3
    mu, logvar = net.latent_stats(pooled)
    assert rel_l2(mu, g["mu"]) < 5e-3 and logvar.shape == mu.shape
    # a mask that is not a prefix: same rows in another order give the same per-token results
    perm = torch.tensor([4, 0, 9, 1, 2, 3, 5, 6, 7, 8, 10], device=dev)
    ids1 = ids[1:2]
    pos = torch.arange(11, device=dev)
    seq_p, _ = net(ids1[:, perm], attention_mask=(ids1[:, perm] > 0).float(), position_ids=pos[perm][None])
    assert rel_l2(seq_p[0], seq[1][perm]) < 2e-3


def test_full_size_bert_vs_oracle(dev):
    """bert-base-cased geometry of the VD checkpoint (12 layers, 768 wide, vocab 28996, latent 768), 3 sentences of
    different lengths, against the fp32 CPU oracle."""
    from oracle import optimus_oracle as OO
    cfg = dict(vocab_size=28996, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
               max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu", latent_size=768)
    net, sd = _build_bert(cfg, 77, dev)
    assert sum(p.numel() for p in net.parameters()) == 108310272 + 2 * 768 * 768    # bert-base-cased + the latent head
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(1, 28996, (3, 40), generator=g)
    ids[0, 0], ids[1, 23:], ids[2, 9:] = 101, 0, 0
    with torch.no_grad():
        seq_ref, pooled_ref = OO.bert_forward(sd, "encoder", ids, (ids > 0).float(), 12, 12)
        mu_ref = OO.bert_latent_mu(sd, "encoder", pooled_ref)
    seq, pooled = net(ids.to(dev), attention_mask=(ids > 0).float().to(dev))
    assert rel_l2(seq, seq_ref) < 5e-3 and rel_l2(pooled, pooled_ref) < 5e-3
    assert rel_l2(net.latent_stats(pooled)[0], mu_ref) < 5e-3


def test_optimus_vae_encode_vs_reference_fixture(dev, tmp_path):
    """optimus_vae_next.encode (reference optimus.py:729-744) end to end -- lower-casing, WordPiece, truncation to
    max_length pieces, [CLS] / [SEP], zero padding, mask, BERT, latent mean -- against the z the REFERENCE's encode
    returned for the same sentences (vocabulary reduced to the pieces in play, see test_oracle_golden.reduced_bert_vocab)."""
    from lib.cfg_helper import CfgDict
    from lib.model_zoo import get_model
    from oracle import synth
    from test_oracle_golden import reduced_bert_vocab
    m = _bert_meta()
    dcfg = dict(vocab_size=259, n_positions=40, n_ctx=40, n_embd=128, n_layer=2, n_head=2, latent_size=64, hidden_size=128,
                layer_norm_epsilon=1e-5, initializer_range=0.02)
    cfg = CfgDict(type="optimus_vae_next", args=CfgDict(
        encoder=CfgDict(type="optimus_bert_connector", args=CfgDict(config=m["config"], latent_size=m["config"]["latent_size"])),
        decoder=CfgDict(type="optimus_gpt2_connector", args=CfgDict(config=dcfg)),
        tokenizer_encoder=CfgDict(type="optimus_bert_tokenizer", args=CfgDict(vocab_file=reduced_bert_vocab(str(tmp_path / "v.txt")),
                                                                               do_lower_case=False, max_len=512)),
        tokenizer_decoder=CfgDict(type="optimus_gpt2_tokenizer", args=CfgDict()),
        args=CfgDict(latent_size=64)))
    vae = get_model()(cfg, verbose=False)
    synth.load_synth_(vae.encoder, m["seed"], prefix="encoder.")
    vae = vae.half().to(dev)
    z = vae.encode(m["texts"], max_length=m["max_length"])
    ref = load_gold("optimus_bert_tiny.npz")["encode_z"]
    assert z.shape == ref.shape and z.dtype == torch.float16
    assert rel_l2(z, ref) < 5e-3
