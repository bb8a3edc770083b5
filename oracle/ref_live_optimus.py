"""TEST INFRASTRUCTURE: seeded RANDOM cases through the live REFERENCE's Optimus connectors (down-sized configs of
oracle/gen_golden_optimus*.py, synthetic weights): GPT-2 latent-connector logits for random token sequences / latents and
BERT latent-connector outputs for random right-padded batches.  Writes an .npz the CPU test replays through
oracle/optimus_oracle.py.  Separate process; needs /root/reference.   usage: python oracle/ref_live_optimus.py out.npz seed n"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gen_golden_optimus as GG, gen_golden_optimus_bert as GB, refshim, synth  # noqa: E402


def main():
    out_path, seed, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    refshim.load_reference()
    rng = np.random.RandomState(seed)
    out = {"n": np.int64(n)}
    with refshim.reference_cwd():
        sys.path.insert(0, refshim.REF_ROOT)
        import lib.model_zoo.optimus as ropt
        from lib.model_zoo.optimus_models.configuration_bert import BertConfig
        from lib.model_zoo.optimus_models.configuration_gpt2 import GPT2Config
        T = GG.TINY
        cfg = GPT2Config(vocab_size_or_config_json_file=T["vocab_size"], n_positions=T["n_positions"], n_ctx=T["n_ctx"],
                         n_embd=T["n_embd"], n_layer=T["n_layer"], n_head=T["n_head"])
        cfg.latent_size = T["latent_size"]
        gpt = ropt.optimus_gpt2_connector(cfg, latent_size=T["latent_size"]).eval()
        synth.load_synth_(gpt, GG.SEED, prefix="decoder.")
        gpt.tie_weights()
        with torch.no_grad():
            gpt.transformer.wte.weight.copy_(synth.synth_tensor("decoder.transformer.wte.weight", gpt.transformer.wte.weight.shape, GG.SEED))
        for blk in gpt.transformer.h:
            m = blk.attn.bias.shape[-1]
            blk.attn.bias.copy_(torch.tril(torch.ones(m, m)).view(1, 1, m, m))
        B_ = GB.TINY
        bcfg = BertConfig(vocab_size_or_config_json_file=B_["vocab_size"], hidden_size=B_["hidden_size"],
                          num_hidden_layers=B_["num_hidden_layers"], num_attention_heads=B_["num_attention_heads"],
                          intermediate_size=B_["intermediate_size"], max_position_embeddings=B_["max_position_embeddings"],
                          layer_norm_eps=B_["layer_norm_eps"])
        bert = ropt.optimus_bert_connector(bcfg, latent_size=B_["latent_size"]).eval()
        synth.load_synth_(bert, GB.SEED, prefix="encoder.")
        for k in range(n):
            g = torch.Generator().manual_seed(seed * 1000 + k)
            Bn, Tn = int(rng.randint(1, 4)), int(rng.randint(1, 20))
            ids = torch.randint(0, T["vocab_size"], (Bn, Tn), generator=g)
            z = torch.randn((Bn, T["latent_size"]), generator=g)
            with torch.no_grad():
                out["%d_gpt_logits" % k] = gpt(input_ids=ids, past=z)[0].numpy()
            out["%d_gpt_ids" % k], out["%d_gpt_z" % k] = ids.numpy(), z.numpy()
            Bb, Lb = int(rng.randint(1, 5)), int(rng.randint(2, 30))
            bids = torch.randint(1, B_["vocab_size"], (Bb, Lb), generator=g)
            for b in range(Bb):                                  # right padding of random length (possibly none)
                keep = int(rng.randint(1, Lb + 1))
                bids[b, keep:] = 0
            with torch.no_grad():
                seq, pooled = bert(bids, attention_mask=(bids > 0).float())[:2]
                mu = bert.linear(pooled).chunk(2, -1)[0]
            out["%d_bert_ids" % k], out["%d_bert_seq" % k] = bids.numpy(), seq.numpy()
            out["%d_bert_pooled" % k], out["%d_bert_mu" % k] = pooled.numpy(), mu.numpy()
    np.savez_compressed(out_path, **out)


if __name__ == "__main__":
    main()
