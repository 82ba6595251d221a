"""Backbone factories on the path (reference: model/backbone/__init__.py, deit.py:97-109, vit.py:1092-1101).

Both factories used by training describe the SAME architecture (patch 16, dim 768, depth 12, heads 12,
mlp x4, qkv_bias, LayerNorm eps 1e-6) and differ only in which pretrained file the reference fetches;
there is no network here, so `pretrained` must be falsy or a path to a local state_dict.
`tiny_test` is the 4-layer/96-dim configuration the golden vectors were generated with.
"""
from ...engine import EncoderConfig

_REGISTRY = {
    "deit_base_patch16_224": dict(),
    "vit_base_patch16_224": dict(),
    "tiny_test": dict(embed_dim=96, depth=4, num_heads=3, head_classes=10),
}


def encoder_config(backbone: str, aux_layer=None) -> EncoderConfig:
    if backbone not in _REGISTRY:
        raise ValueError(f"backbone {backbone!r} is not on the DuPL hot path; available: {sorted(_REGISTRY)}")
    kw = dict(_REGISTRY[backbone])
    if aux_layer is not None:
        kw["aux_layer"] = aux_layer
    return EncoderConfig(**kw)


def load_pretrained_encoder(encoder_module, path: str, patch: int = 16):
    """The reference's two pretrained-weight routes, fed from a LOCAL file (no network here):

    * deit.py:101-108 (`deit_base_patch16_224`): `torch.hub.load_state_dict_from_url(...)["model"]` then a STRICT
      `model.load_state_dict` -> a `{"model": state_dict}` file;
    * vit.py:1098-1100 (`vit_base_patch16_224`, ImageNet-21k init): timm 0.4.12 `load_pretrained(model,
      num_classes=model.num_classes, in_chans=3, filter_fn=_conv_filter)`: a FLAT state_dict (the `jx_vit_base_p16_224`
      file), `_conv_filter` (vit.py:1058-1065) reshapes a manually-patchified `patch_embed.proj.weight`
      (D, 3*p*p) to the conv layout (D, 3, p, p); the encoder keeps its 1000-way `head`, so timm leaves the
      classifier in and loads strictly.

    Both are strict in the reference; a file with missing / unexpected keys (distilled DeiT, `module.`-prefixed, a
    whole-`network` checkpoint) therefore raises here too instead of silently training from random init."""
    import torch
    sd = torch.load(path, map_location="cpu")
    if isinstance(sd, dict) and "model" in sd and isinstance(sd["model"], dict):
        sd = sd["model"]
    out = {}
    for k, v in sd.items():
        if "patch_embed.proj.weight" in k and v.dim() == 2:      # _conv_filter
            v = v.reshape(v.shape[0], 3, patch, patch)
        out[k] = v
    encoder_module.load_state_dict(out, strict=True)
