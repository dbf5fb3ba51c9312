"""Synthetic state_dict key/shape grammars (TEST INFRASTRUCTURE ONLY).

Key grammar follows the reference's all_moe / ufo checkpoints (SURVEY.md 8a "Param/key facts";
reference src/vilt/modules/vision_transformer.py:366-491 for the per-block tensors).
"""
from collections import OrderedDict

EXPERT_TENSORS = [  # (template with {m} = modality segment, shape fn)
    ("attn.{m}q_bias", lambda D, F: (D,)),
    ("attn.{m}v_bias", lambda D, F: (D,)),
    ("attn.{m}qkv.weight", lambda D, F: (3 * D, D)),
    ("attn.{m}proj.weight", lambda D, F: (D, D)),
    ("attn.{m}proj.bias", lambda D, F: (D,)),
    ("norm1.{m}weight", lambda D, F: (D,)),
    ("norm1.{m}bias", lambda D, F: (D,)),
    ("mlp.{m}fc1.weight", lambda D, F: (F, D)),
    ("mlp.{m}fc1.bias", lambda D, F: (F,)),
    ("mlp.{m}fc2.weight", lambda D, F: (D, F)),
    ("mlp.{m}fc2.bias", lambda D, F: (D,)),
    ("norm2.{m}weight", lambda D, F: (D,)),
    ("norm2.{m}bias", lambda D, F: (D,)),
]


def block_shapes(D, F, arch, n_layers=12, vlffn_start=10):
    """{key: (shape, 'float32')} of the transformer.blocks.* entries."""
    out = OrderedDict()
    for i in range(n_layers):
        pre = f"transformer.blocks.{i}."
        out[pre + "gamma_1"] = ((D,), "float32")
        out[pre + "gamma_2"] = ((D,), "float32")
        if arch == "ufo":
            mods = [""]
        else:
            mods = ["v.", "l."] + (["vl."] if i >= vlffn_start else [])
        for m in mods:
            for tmpl, fn in EXPERT_TENSORS:
                out[pre + tmpl.format(m=m)] = (fn(D, F), "float32")
    return out


def non_block_shapes(D, R, vocab=64, T=40, heads=12, n_layers=12):
    out = OrderedDict()
    out["relative_position_bias_table"] = ((R, heads * n_layers), "float32")
    out["logit_scale"] = ((), "float32")
    out["text_embeddings.word_embeddings.weight"] = ((vocab, D), "float32")
    out["text_embeddings.position_embeddings.weight"] = ((T, D), "float32")
    out["text_embeddings.token_type_embeddings.weight"] = ((2, D), "float32")
    out["text_embeddings.LayerNorm.weight"] = ((D,), "float32")
    out["text_embeddings.LayerNorm.bias"] = ((D,), "float32")
    out["token_type_embeddings.weight"] = ((2, D), "float32")
    out["transformer.cls_token"] = ((1, 1, D), "float32")
    out["transformer.patch_embed.proj.weight"] = ((D, 3, 16, 16), "float32")
    out["transformer.patch_embed.proj.bias"] = ((D,), "float32")
    out["transformer.norm.weight"] = ((D,), "float32")
    out["transformer.norm.bias"] = ((D,), "float32")
    return out


def state_shapes(D, F, arch, R=64, **kw):
    out = non_block_shapes(D, R, **kw)
    out.update(block_shapes(D, F, arch))
    return out


def gram_shapes(D, F, n_layers=12, modalities=("v", "l")):
    """Gram dict keys as cache_gram_matrices.py produces them."""
    out = OrderedDict()
    for i in range(n_layers):
        for m in modalities:
            out[f"transformer.blocks.{i}.attn.{m}"] = (D, D)
            out[f"transformer.blocks.{i}.attn.{m}.proj"] = (D, D)
            out[f"transformer.blocks.{i}.mlp.{m}.fc1"] = (D, D)
            out[f"transformer.blocks.{i}.mlp.{m}.fc2"] = (F, F)
    return out
