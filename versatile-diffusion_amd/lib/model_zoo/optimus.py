"""Optimus text VAE names (reference lib/model_zoo/optimus.py:16-763) -- placeholders only.

The Optimus encoder/decoder serve the *text data* flow (image-to-text, text-variation), which BASELINE.json's
north-star path does not include.  The registry names are kept so
that `vd_four_flow_v1-0` resolves and reference checkpoints load with strict=False; using them raises."""
import torch.nn as nn

from .common.get_model import register


class _OutOfScope(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()

    def _no(self, *a, **k):
        raise NotImplementedError("%s: the Optimus text VAE is outside the image sampling path of this package"
                                  % type(self).__name__)

    encode = decode = forward = _no


@register("optimus_bert_connector")
class optimus_bert_connector(_OutOfScope):
    pass


@register("optimus_gpt2_connector")
class optimus_gpt2_connector(_OutOfScope):
    pass


@register("optimus_bert_tokenizer")
class optimus_bert_tokenizer(_OutOfScope):
    pass


@register("optimus_gpt2_tokenizer")
class optimus_gpt2_tokenizer(_OutOfScope):
    pass


@register("optimus_vae")
@register("optimus_vae_next")
class optimus_vae_next(_OutOfScope):
    def __init__(self, encoder=None, decoder=None, tokenizer_encoder=None, tokenizer_decoder=None, args=None):
        super().__init__()
