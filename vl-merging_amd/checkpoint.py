"""Checkpoint formats either side of the merge: BEiT / VLMo -> this model's keys, and
PyTorch-Lightning-compatible `.ckpt` files so merged models round-trip with the reference repo.

Host-side dictionary logic only (no kernels): tensors are re-keyed, cloned by reference and -- for the relative-position
table -- resized with torch's bicubic `interpolate`, exactly the operator the reference calls, so results are
bit-identical to it on the same torch build (pinned by tests/golden/ckpt_rekey_*.{npz,json}).
Reference: src/vilt/modules/vilt_module.py:808-972 (modify_checkpoint_beit), :974-1058 (modify_checkpoint_self),
:284-305 (how __init__ consumes a checkpoint).
"""
import torch
import torch.nn.functional as F

_PL_VERSION = "1.1.4"  # the reference's docker image (README.md:16) pins pytorch_lightning 1.1.4


def _pop_beit_tables(sd, num_layers):
    """The BEiT relative-position table(s) of a checkpoint, removed from `sd`: ([R, heads * k], shared) or (None, False).
    pt22k_ft22k checkpoints share ONE table across layers; pt22k checkpoints carry one per block (concatenated along
    the head axis, the layout of this model's [R, layers*heads] table).  vilt_module.py:817-834."""
    shared_key = "transformer.rel_pos_bias.relative_position_bias_table"
    if shared_key in sd:
        table = sd.pop(shared_key)
        sd.pop("transformer.rel_pos_bias.relative_position_index")
        return table, True
    if "transformer.blocks.0.attn.relative_position_bias_table" in sd:
        parts = []
        for i in range(num_layers):
            parts.append(sd.pop("transformer.blocks.%d.attn.relative_position_bias_table" % i))
            sd.pop("transformer.blocks.%d.attn.relative_position_index" % i)
        return torch.cat(parts, dim=-1), False
    return None, False


def beit_relpos_to_model(model, table, shared):
    """Resize a BEiT table ((2w-1)^2 image offsets + 3 cls rows) to this model's window and append the rows BEiT does
    not have (text distances + 2 text/image cross rows), which keep the model's own values.  vilt_module.py:836-880."""
    own = model.relative_position_bias_table
    if model.transformer.patch_embed.patch_shape[0] != model.transformer.patch_embed.patch_shape[1]:
        raise NotImplementedError("non-square patch grids")
    heads = table.size(1)
    n_extra = model.text_num_relative_distance + 2 + 3  # text rows, 2 cross-modal rows, 3 image cls rows
    src = int((table.size(0) - 3) ** 0.5)
    dst = int((own.size(0) - n_extra) ** 0.5)
    text_rows = own[-(n_extra - 3):, :]
    cls_rows = table[-3:, :]
    grid = table[:-3, :].transpose(0, 1).view(-1, src, src)
    grid = F.interpolate(grid.unsqueeze(0), size=(dst, dst), mode="bicubic").squeeze(0)
    body = grid.permute(1, 2, 0).contiguous().view(-1, grid.size(0))
    if shared:  # every layer starts from the same values
        repeat = own.size(1) // heads
        body = body.repeat((1, repeat))
        cls_rows = cls_rows.repeat((1, repeat))
    return torch.cat((body, cls_rows, text_rows), dim=0)


def _insert(key, what, from_end):
    parts = key.split(".")
    parts.insert(len(parts) - from_end, what)
    return ".".join(parts)


def rekey_for_experts(sd, cfg, moe_config):
    """Single-expert (BEiT) keys -> the vision expert of modality-expert modules: `mlp.fc1.weight` -> `mlp.v.fc1.weight`,
    `attn.q_bias` -> `attn.v.q_bias`, `norm1.weight` -> `norm1.v.weight`.  vilt_module.py:884-938."""
    if cfg["use_moe"]:
        out = {}
        for k, v in sd.items():
            if moe_config.in_ffn and "mlp" in k:
                k = _insert(k, "v", 2)
            elif moe_config.in_attn and "attn" in k:
                k = _insert(k, "v", 1 if ("attn.q_bias" in k or "attn.v_bias" in k) else 2)
            out[k] = v
        sd = out
    for flag, tag in (("use_custom_ln_attn", ".norm1"), ("use_custom_ln_ffn", ".norm2")):
        if cfg[flag]:
            sd = {(_insert(k, "v", 1) if tag in k else k): v for k, v in sd.items()}
    return sd


def clone_vision_experts(sd, cfg):
    """`use_vision_weights_for_other_modalities`: every `.v.` tensor also initialises the language expert and, from
    `vlffn_start_layer_index` on, the vision-language expert (same tensor objects, as in the reference :940-961)."""
    out = {}
    for k, v in sd.items():
        if ".v." in k:
            out[k.replace(".v.", ".l.")] = v
            if int(k.split(".")[2]) >= cfg["vlffn_start_layer_index"]:
                out[k.replace(".v.", ".vl.")] = v
        out[k] = v
    return out


def _final_norm_rename(sd):
    if "transformer.fc_norm.weight" in sd:  # BEiT fine-tuned checkpoints name the final LayerNorm fc_norm
        sd["transformer.norm.weight"] = sd.pop("transformer.fc_norm.weight")
        sd["transformer.norm.bias"] = sd.pop("transformer.fc_norm.bias")
    return sd


def modify_checkpoint_beit(model, ckpt):
    """vilt_module.py:808-972.  Returns the re-keyed state_dict, or None when `ckpt` has no "state_dict"."""
    if "state_dict" not in ckpt:
        return None
    cfg = model.hparams.config
    sd = ckpt["state_dict"]
    table, shared = _pop_beit_tables(sd, cfg["num_layers"])
    if table is not None:
        sd["relative_position_bias_table"] = beit_relpos_to_model(model, table, shared)
    sd = rekey_for_experts(sd, cfg, model.moe_config)
    if cfg["use_vision_weights_for_other_modalities"]:
        sd = clone_vision_experts(sd, cfg)
    return _final_norm_rename(sd)


def modify_checkpoint_self(model, sd):
    """vilt_module.py:974-1058: a bare state_dict of this model family; text position table truncated to max_text_len,
    BEiT-style tables converted, fc_norm renamed (no expert re-keying)."""
    key = "text_embeddings.position_embeddings.weight"
    if sd[key].size(0) != model.max_text_len:
        sd[key].data = sd[key].data[: model.max_text_len, :]
        sd["text_embeddings.position_ids"].data = sd["text_embeddings.position_ids"].data[:, : model.max_text_len]
    table, shared = _pop_beit_tables(sd, model.hparams.config["num_layers"])
    if table is not None:
        sd["relative_position_bias_table"] = beit_relpos_to_model(model, table, shared)
    return _final_norm_rename(sd)


# ----------------------------------------------------------------------------------------------------------------------
def save_ckpt(path, model, global_step=0, epoch=0, extra=None):
    """Write what `torch.load(path, map_location="cpu")["state_dict"]` in the reference (vilt_module.py:285-293, and
    pytorch_lightning's own resume) expects: CPU tensors under "state_dict" plus the Lightning bookkeeping keys."""
    sd = {k: v.detach().to("cpu").clone() for k, v in model.state_dict().items()}
    ckpt = {"state_dict": sd, "global_step": int(global_step), "epoch": int(epoch),
            "pytorch-lightning_version": _PL_VERSION,
            "hyper_parameters": {"config": dict(model.hparams.config)} if hasattr(model, "hparams") else {}}
    if extra:
        ckpt.update(extra)
    torch.save(ckpt, path)
    return ckpt


def load_file(path):
    """torch.load of a user-supplied LOCAL artefact of the reference repo (a Lightning .ckpt with AttributeDict
    hyper-parameters and callback class keys, or the Gram cache's defaultdict(float), cache_gram_matrices.py:349):
    these are pickles of non-tensor objects, so torch >= 2.6's weights_only default cannot read them."""
    return torch.load(path, map_location="cpu", weights_only=False)


def load_ckpt(path):
    """state_dict of a Lightning `.ckpt` (or of a bare state_dict file) on the CPU."""
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    return ckpt["state_dict"] if isinstance(ckpt, dict) and "state_dict" in ckpt else ckpt
