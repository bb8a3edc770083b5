"""Generates tests/golden/optimus_bert_tiny.npz + optimus_bert_tokenizer.json by running the REFERENCE's vendored BERT latent
connector (/root/reference/lib/model_zoo/optimus.py: optimus_bert_connector = optimus_models.optimus_bert.
BertForLatentConnector_XX), its BertTokenizer with the published bert-base-cased vocabulary, and the reference's own
`optimus_vae_next.encode` (optimus.py:729-744) on a down-sized config with seeded synthetic weights (oracle/synth.py).

    python oracle/gen_golden_optimus_bert.py        (CPU container, needs /root/reference)
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import refshim, synth  # noqa: E402

TINY = dict(vocab_size=28996, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
            max_position_embeddings=96, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu", initializer_range=0.02,
            latent_size=64)
SEED = 8765
VOCAB = "lib/model_zoo/optimus_models/vocab/bert-base-cased-vocab.txt"
TEXTS = ["A photo of a cat, sitting on the mat.", "Two dogs don't play in the snow!", "it's 12:30 -- time's up",
         "naïve café ☕ unaffable", "house in the 山水 style; (oil) painting #42 by o'keeffe",
         "supercalifragilisticexpialidocious [SEP] x", "a"]


class _Stub:   # what optimus_vae_next.encode touches of `self`
    pass


def main():
    refshim.load_reference()
    out = {}
    with refshim.reference_cwd():
        sys.path.insert(0, refshim.REF_ROOT)
        try:
            import lib.model_zoo.optimus as ropt
            from lib.model_zoo.optimus_models.configuration_bert import BertConfig
            cfg = BertConfig(vocab_size_or_config_json_file=TINY["vocab_size"], hidden_size=TINY["hidden_size"],
                             num_hidden_layers=TINY["num_hidden_layers"], num_attention_heads=TINY["num_attention_heads"],
                             intermediate_size=TINY["intermediate_size"], max_position_embeddings=TINY["max_position_embeddings"],
                             layer_norm_eps=TINY["layer_norm_eps"])
            model = ropt.optimus_bert_connector(cfg, latent_size=TINY["latent_size"]).eval()
            synth.load_synth_(model, SEED, prefix="encoder.")
            g = torch.Generator().manual_seed(SEED)
            ids = torch.randint(1, TINY["vocab_size"], (3, 11), generator=g)
            ids[1, 7:] = 0          # right padding, as encode() produces it
            ids[2, 3:] = 0
            mask = (ids > 0).float()
            with torch.no_grad():
                seq, pooled = model(ids, attention_mask=mask)[:2]
                mu = model.linear(pooled).chunk(2, -1)[0]
            out.update(ids=ids.numpy(), seq=seq.numpy(), pooled=pooled.numpy(), mu=mu.numpy())
            tok = ropt.optimus_bert_tokenizer(vocab_file=VOCAB, do_lower_case=False, max_len=512)
            cases = []
            for t in TEXTS:
                for text in (t, t.lower()):
                    pieces = tok.tokenize(text)
                    cases.append({"text": text, "pieces": pieces, "ids": [tok._convert_token_to_id(p) for p in pieces]})
            tok_lc = ropt.optimus_bert_tokenizer(vocab_file=VOCAB, do_lower_case=True, max_len=512)
            cases_lc = [{"text": t, "pieces": tok_lc.tokenize(t)} for t in TEXTS]
            stub = _Stub()
            stub.tokenizer_encoder, stub.encoder = tok, model
            stub.get_device = lambda: torch.device("cpu")
            with torch.no_grad():
                z = ropt.optimus_vae_next.encode(stub, TEXTS, max_length=12)
            out["encode_z"] = z.numpy()
            tk = {"special": [tok.cls_token_id, tok.sep_token_id, tok.pad_token_id, tok.unk_token_id], "len": len(tok),
                  "cases": cases, "cases_lower": cases_lc, "with_special": tok.add_special_tokens_single_sentence([5, 6])}
        finally:
            sys.path.remove(refshim.REF_ROOT)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "optimus_bert_tiny.npz"), **out)
    with open(os.path.join(ROOT, "tests", "golden", "optimus_bert_tokenizer.json"), "w") as f:
        json.dump({"config": TINY, "seed": SEED, "texts": TEXTS, "max_length": 12, "tokenizer": tk}, f, indent=1, ensure_ascii=False)
    print({k: v.shape for k, v in out.items()})
    print(tk["cases"][7], tk["cases_lower"][3])


if __name__ == "__main__":
    main()
