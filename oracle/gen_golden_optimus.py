"""Generates tests/golden/optimus_tiny.npz by running the REFERENCE's vendored GPT-2 latent connector
(/root/reference/lib/model_zoo/optimus.py: optimus_gpt2_connector = optimus_models.optimus_gpt2.GPT2ForLatentConnector_XX)
on a down-sized config with seeded synthetic weights (oracle/synth.py; regenerated from names + seed, not shipped), and
the reference tokenizer on a few strings (ids + decoded text pin the product's own GPT-2 BPE implementation).

    python oracle/gen_golden_optimus.py        (CPU container, needs /root/reference)
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import refshim, synth  # noqa: E402

TINY = dict(vocab_size=96, n_positions=40, n_ctx=40, n_embd=128, n_layer=2, n_head=2, latent_size=64, hidden_size=128,
            layer_norm_epsilon=1e-5, initializer_range=0.02)
SEED = 4321
TEXTS = ["a photo of a cat, sitting on the mat.", "Two dogs don't play in the snow!", "it's 12:30 -- time's up", "naïve café ☕"]


def main():
    refshim.load_reference()
    out = {}
    with refshim.reference_cwd():
        sys.path.insert(0, refshim.REF_ROOT)
        try:
            import lib.model_zoo.optimus as ropt
            from lib.model_zoo.optimus_models.configuration_gpt2 import GPT2Config
            cfg = GPT2Config(vocab_size_or_config_json_file=TINY["vocab_size"], n_positions=TINY["n_positions"], n_ctx=TINY["n_ctx"],
                             n_embd=TINY["n_embd"], n_layer=TINY["n_layer"], n_head=TINY["n_head"])
            cfg.latent_size = TINY["latent_size"]
            model = ropt.optimus_gpt2_connector(cfg, latent_size=TINY["latent_size"]).eval()
            synth.load_synth_(model, SEED, prefix="decoder.")
            model.tie_weights()
            with torch.no_grad():   # lm_head shares wte's storage: pin the shared tensor to wte's synthetic draw
                model.transformer.wte.weight.copy_(synth.synth_tensor("decoder.transformer.wte.weight", model.transformer.wte.weight.shape, SEED))
            for blk in model.transformer.h:   # the causal-mask buffers are not weights: keep them lower-triangular
                n = blk.attn.bias.shape[-1]
                blk.attn.bias.copy_(torch.tril(torch.ones(n, n)).view(1, 1, n, n))
            g = torch.Generator().manual_seed(SEED)
            ids = torch.randint(0, TINY["vocab_size"], (2, 9), generator=g)
            z = torch.randn((2, TINY["latent_size"]), generator=g)
            with torch.no_grad():
                logits = model(input_ids=ids, past=z)[0]
            out.update(ids=ids.numpy(), z=z.numpy(), logits=logits.numpy())
            # the reference's sampling loop (optimus.py:662-688), seeded
            torch.manual_seed(99)
            seq = ropt.sample_single_sequence_conditional(model=model, context=torch.LongTensor([5]), past=z[0], temperature=1.0,
                                                          top_k=0, top_p=1.0, max_length=12, eos_token=7)
            out["sampled"] = seq.numpy()
            tok = ropt.optimus_gpt2_tokenizer(vocab_file="lib/model_zoo/optimus_models/vocab/gpt2-vocab.json",
                                              merges_file="lib/model_zoo/optimus_models/vocab/gpt2-merges.txt")
            tok.add_special_tokens({"pad_token": "<PAD>", "bos_token": "<BOS>", "eos_token": "<EOS>"})
            tk = {"special": [tok.encode("<BOS>"), tok.encode("<EOS>"), tok.encode("<PAD>")], "len": len(tok), "cases": []}
            for t in TEXTS:
                e = tok.encode("<BOS>") + tok.encode(t) + tok.encode("<EOS>")
                tk["cases"].append({"text": t, "ids": e, "decoded": tok.decode(e, clean_up_tokenization_spaces=True)})
        finally:
            sys.path.remove(refshim.REF_ROOT)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "optimus_tiny.npz"), **out)
    with open(os.path.join(ROOT, "tests", "golden", "optimus_tokenizer.json"), "w") as f:
        json.dump({"config": TINY, "seed": SEED, "tokenizer": tk}, f, indent=1, ensure_ascii=False)
    print({k: v.shape for k, v in out.items()}, out["sampled"])


if __name__ == "__main__":
    main()
