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
