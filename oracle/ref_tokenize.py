"""TEST INFRASTRUCTURE: run the REFERENCE's vendored tokenizers (lib/model_zoo/optimus_models/tokenization_{bert,gpt2}.py)
on strings given as a JSON list on stdin; prints {"bert": [[pieces], ...], "gpt2": [[ids], ...], "gpt2_decoded": [...]}.
Separate process on purpose: the reference package is also called `lib`.  Needs /root/reference (CPU container only)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import refshim  # noqa: E402


def main():
    texts = json.load(sys.stdin)
    refshim.load_reference()
    with refshim.reference_cwd():
        sys.path.insert(0, refshim.REF_ROOT)
        import lib.model_zoo.optimus as ropt
        v = "lib/model_zoo/optimus_models/vocab/"
        bert = ropt.optimus_bert_tokenizer(vocab_file=v + "bert-base-cased-vocab.txt", do_lower_case=False, max_len=512)
        gpt2 = ropt.optimus_gpt2_tokenizer(vocab_file=v + "gpt2-vocab.json", merges_file=v + "gpt2-merges.txt")
        gpt2.add_special_tokens({"pad_token": "<PAD>", "bos_token": "<BOS>", "eos_token": "<EOS>"})
        out = {"bert": [bert.tokenize(t) for t in texts], "gpt2": [gpt2.encode(t) for t in texts]}
        out["gpt2_decoded"] = [gpt2.decode(ids, clean_up_tokenization_spaces=True) for ids in out["gpt2"]]
    json.dump(out, sys.stdout)


if __name__ == "__main__":
    main()
