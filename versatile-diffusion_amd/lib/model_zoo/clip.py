"""CLIP ViT-L/14 context encoders of Versatile Diffusion on the HIP kernel library.

Surface as in the reference (lib/model_zoo/clip.py:30-149 there): registry names `clip_text_context_encoder` /
`clip_image_context_encoder`, `.encode(...)`, `.fp16`, and a `.model` sub-tree whose state-dict keys are those of
HF `transformers.CLIPModel` (text_model.*, vision_model.*, text_projection, visual_projection, logit_scale), so the
`ctx.text.model.*` / `ctx.image.model.*` tensors of the VD checkpoints load as they are.

The tower arithmetic (third-party `transformers` in the reference) is implemented here on vd_hip kernels:
LayerNorm, fused QKV GEMM (+bias), fused attention (head dim 64, causal for text), out-proj GEMM with the residual
in its epilogue, fc1 GEMM with quick-GELU in its epilogue, fc2 GEMM + residual; VD's post-processing (project ALL
tokens, divide by the norm of the projected pooled/CLS token, optional mask weighting) is one small kernel.
"""
import numpy as np
import torch
import torch.nn as nn

from vd_hip import ops, pack

from .common.get_model import register
from .hip_layers import LayerNorm, Linear, PackCache, _h

symbol = "clip"

VIT_L14 = dict(
    text=dict(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
              max_position_embeddings=77),
    vision=dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16,
                image_size=224, patch_size=14),
    projection_dim=768)

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


class _Attn(nn.Module, PackCache):
    def __init__(self, dim, heads):
        super().__init__()
        self.heads = heads
        self.k_proj, self.v_proj, self.q_proj, self.out_proj = Linear(dim, dim), Linear(dim, dim), Linear(dim, dim), Linear(dim, dim)

    def forward(self, x, res, causal):
        w, b = self._packed("qkv", (self.q_proj.weight, self.k_proj.weight, self.v_proj.weight, self.q_proj.bias,
                                    self.k_proj.bias, self.v_proj.bias),
                            lambda: (torch.cat([_h(self.q_proj.weight), _h(self.k_proj.weight), _h(self.v_proj.weight)], 0),
                                     torch.cat([_h(self.q_proj.bias), _h(self.k_proj.bias), _h(self.v_proj.bias)], 0)))
        c = x.shape[-1]
        qkv = ops.linear(x, w, b)
        a = ops.attention(qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:], self.heads, causal=causal)
        return self.out_proj(a, res=res)


class _MLP(nn.Module):
    def __init__(self, dim, inter):
        super().__init__()
        self.fc1, self.fc2 = Linear(dim, inter), Linear(inter, dim)

    def forward(self, x, res):
        return self.fc2(self.fc1(x, act=ops.ACT_QUICK_GELU), res=res)


class _Layer(nn.Module):
    def __init__(self, dim, inter, heads):
        super().__init__()
        self.self_attn = _Attn(dim, heads)
        self.layer_norm1 = LayerNorm(dim, eps=1e-5)
        self.mlp = _MLP(dim, inter)
        self.layer_norm2 = LayerNorm(dim, eps=1e-5)

    def forward(self, x, causal):
        x = self.self_attn(self.layer_norm1(x), res=x, causal=causal)
        return self.mlp(self.layer_norm2(x), res=x)


class _Encoder(nn.Module):
    def __init__(self, dim, inter, heads, layers):
        super().__init__()
        self.layers = nn.ModuleList([_Layer(dim, inter, heads) for _ in range(layers)])

    def forward(self, x, causal):
        for layer in self.layers:
            x = layer(x, causal)
        return x


class _TextEmbeddings(nn.Module, PackCache):
    def __init__(self, vocab, dim, max_pos):
        super().__init__()
        self.token_embedding = nn.Embedding(vocab, dim)
        self.position_embedding = nn.Embedding(max_pos, dim)
        self.register_buffer("position_ids", torch.arange(max_pos).expand((1, -1)), persistent=False)

    def forward(self, ids):
        tok, pos = self._packed("e", (self.token_embedding.weight, self.position_embedding.weight),
                                lambda: (_h(self.token_embedding.weight), _h(self.position_embedding.weight)))
        return ops.embed_tokens(ids, tok, pos)


class _TextModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        d = cfg["hidden_size"]
        self.embeddings = _TextEmbeddings(cfg["vocab_size"], d, cfg["max_position_embeddings"])
        self.encoder = _Encoder(d, cfg["intermediate_size"], cfg["num_attention_heads"], cfg["num_hidden_layers"])
        self.final_layer_norm = LayerNorm(d, eps=1e-5)

    def forward(self, ids):
        return self.final_layer_norm(self.encoder(self.embeddings(ids), causal=True))


class _VisionEmbeddings(nn.Module, PackCache):
    def __init__(self, dim, image_size, patch):
        super().__init__()
        self.patch_size = patch
        self.class_embedding = nn.Parameter(torch.randn(dim))
        self.patch_embedding = nn.Conv2d(3, dim, kernel_size=patch, stride=patch, bias=False)
        self.num_positions = (image_size // patch) ** 2 + 1
        self.position_embedding = nn.Embedding(self.num_positions, dim)
        self.register_buffer("position_ids", torch.arange(self.num_positions).expand((1, -1)), persistent=False)

    def forward(self, pixel_values, token_scale=None):
        wp, cls, pos = self._packed("e", (self.patch_embedding.weight, self.class_embedding, self.position_embedding.weight),
                                    lambda: (pack.pack_patch_weight(_h(self.patch_embedding.weight)), _h(self.class_embedding),
                                             _h(self.position_embedding.weight)))
        B = pixel_values.shape[0]
        a = ops.patchify(pixel_values, self.patch_size)
        pe = ops.gemm(a, wp).view(B, self.num_positions - 1, -1)
        return ops.clip_vision_embed(pe, cls, pos, token_scale)


class _VisionModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        d = cfg["hidden_size"]
        self.embeddings = _VisionEmbeddings(d, cfg["image_size"], cfg["patch_size"])
        self.pre_layrnorm = LayerNorm(d, eps=1e-5)  # (sic) HF's attribute name
        self.encoder = _Encoder(d, cfg["intermediate_size"], cfg["num_attention_heads"], cfg["num_hidden_layers"])
        self.post_layernorm = LayerNorm(d, eps=1e-5)

    def forward(self, pixel_values, token_scale=None):
        x = self.pre_layrnorm(self.embeddings(pixel_values, token_scale))
        return self.encoder(x, causal=False)


class CLIPModelHIP(nn.Module):
    """Both towers + projections, state-dict compatible with transformers.CLIPModel."""

    def __init__(self, config=None):
        super().__init__()
        config = VIT_L14 if config is None else config
        self.config = config
        self.text_model = _TextModel(config["text"])
        self.vision_model = _VisionModel(config["vision"])
        self.visual_projection = Linear(config["vision"]["hidden_size"], config["projection_dim"], bias=False)
        self.text_projection = Linear(config["text"]["hidden_size"], config["projection_dim"], bias=False)
        self.logit_scale = nn.Parameter(torch.tensor(2.6592))


def disabled_train(self, mode=True):
    return self


class AbstractEncoder(nn.Module):
    def encode(self, *args, **kwargs):
        raise NotImplementedError


def _try_tokenizer(version):
    try:
        from transformers import CLIPTokenizer
        return CLIPTokenizer.from_pretrained(version, local_files_only=True)
    except Exception:
        return None


@register("clip_text_context_encoder")
class CLIPTextContextEncoder(AbstractEncoder):
    def __init__(self, version="openai/clip-vit-large-patch14", max_length=77, fp16=False, config=None):
        super().__init__()
        self.version = version
        self.tokenizer = None  # resolved lazily: needs the vocab files on disk (no network here)
        self.model = CLIPModelHIP(config)
        self.max_length = max_length
        self.fp16 = fp16
        self.freeze()

    def get_device(self):
        return self.model.text_projection.weight.device

    def freeze(self):
        self.model = self.model.eval()
        self.train = disabled_train
        for p in self.parameters():
            p.requires_grad = False

    def tokenize(self, text):
        if self.tokenizer is None:
            self.tokenizer = _try_tokenizer(self.version)
        if self.tokenizer is None:
            raise RuntimeError("CLIP tokenizer files for '%s' are not available offline; pass token ids "
                               "(LongTensor [B, 77]) to encode() instead of strings" % self.version)
        enc = self.tokenizer(text, truncation=True, max_length=self.max_length, return_length=True,
                             return_overflowing_tokens=False, padding="max_length", return_tensors="pt")
        return enc["input_ids"]

    @torch.no_grad()
    def encode(self, text):
        """list[str] (or LongTensor token ids [B, L]) -> [B, L, 768]: text_projection of ALL hidden states divided by
        the norm of the projected pooled (EOS = largest id) state  (reference clip.py:53-62)."""
        ids = text if isinstance(text, torch.Tensor) else self.tokenize(text)
        ids_host = ids.detach().cpu()
        pool = ids_host.argmax(dim=-1).to(torch.int32)  # index bookkeeping on the host copy of the ids
        dev = self.get_device()
        h = self.model.text_model(ids_host.to(dev).contiguous())
        z = self.model.text_projection(h)
        ops.scale_by_row_norm_(z, pool_idx=pool.to(dev))
        return z if self.fp16 else z.float()


@register("clip_image_context_encoder")
class CLIPImageContextEncoder(AbstractEncoder):
    def __init__(self, version="openai/clip-vit-large-patch14", fp16=False, config=None):
        super().__init__()
        self.version = version
        self.model = CLIPModelHIP(config)
        self.fp16 = fp16
        self.freeze()

    get_device = CLIPTextContextEncoder.get_device
    freeze = CLIPTextContextEncoder.freeze

    def preprocess(self, images):
        """[B,3,H,W] in [0,1] (or a list of PIL images) -> CLIP pixel_values [B,3,224,224] fp16 on the device.
        The reference does this on the host (clip.py:88-94: ToPILImage -> CLIPProcessor = Pillow bicubic resize of the
        shortest edge, centre crop, rescale, normalise); here it is `vd_clip_preprocess_f16` on the device, bit-exact with
        that path (tests/test_clip_preprocess_*.py).  Tensors are quantised like ToPILImage (mul(255).byte()); PIL
        images go in as the uint8 they are."""
        size = self.model.config["vision"]["image_size"]
        if not isinstance(images, torch.Tensor):
            sizes = {im.size for im in images}
            assert len(sizes) == 1, "a batch of PIL images must share one size"
            images = torch.stack([torch.from_numpy(np.asarray(im.convert("RGB"), dtype=np.uint8)).permute(2, 0, 1) for im in images])
        x = images.to(device=self.get_device())
        if x.dtype not in (torch.float32, torch.float16, torch.uint8):
            x = x.float()
        return ops.clip_preprocess(x, size)

    def vtoken_mask(self, masks):
        """[B,1,H,W] mask -> per-token weights [B, 257] fp32 = [global mean | 14x14 patch means] of the mask clamped to
        [0,1] and resized to 224^2 bilinear (reference clip.py:104-122), computed on the device (vd_mask_patch_weights).
        The reference falls back to the unmasked encoder when the resized mask is all ones (a host sync on the mask
        sum, clip.py:109-110); here an all-ones mask simply yields weights of 1, which scale nothing."""
        assert isinstance(masks, torch.Tensor) and masks.dim() == 4 and masks.shape[1] == 1
        size = self.model.config["vision"]["image_size"]
        p = self.model.config["vision"]["patch_size"]
        m = masks.to(device=self.get_device())
        if m.dtype not in (torch.float32, torch.float16):
            m = m.float()
        return ops.mask_patch_weights(m, size, p)

    @torch.no_grad()
    def encode_pixels(self, pixel_values, token_scale=None):
        """pixel_values [B,3,224,224] -> [B,257,768]: post_layernorm on ALL tokens, visual_projection, divide by the
        norm of the projected CLS token; masked variant scales the embeddings before the encoder and the outputs
        after it by `token_scale` (reference clip.py:95-100, 124-142)."""
        pv = pixel_values.to(device=self.get_device(), dtype=torch.float16).contiguous()
        h = self.model.vision_model(pv, token_scale)
        z = self.model.visual_projection(self.model.vision_model.post_layernorm(h))
        ops.scale_by_row_norm_(z, row_scale=token_scale)
        return z if self.fp16 else z.float()

    def _encode(self, images):
        return self.encode_pixels(self.preprocess(images))

    @torch.no_grad()
    def _encode_wmask(self, images, masks):
        ts = self.vtoken_mask(masks)
        return self.encode_pixels(self.preprocess(images), ts)

    def encode(self, images, masks=None):
        return self._encode(images) if masks is None else self._encode_wmask(images, masks)
