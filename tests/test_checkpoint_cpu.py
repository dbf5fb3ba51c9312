"""Checkpoint re-keying (BEiT / self formats) against what the reference produced on the same deterministic inputs
(tests/golden/ckpt_rekey_*, made by tests/golden/make_golden.py ckpt), and the Lightning-compatible .ckpt round trip.
CPU only: this is host dictionary logic."""
import hashlib
import importlib
import json
import os

import numpy as np
import pytest
import torch

import __graft_entry__ as ge
from oracle.detweights import det_array

HERE = os.path.dirname(os.path.abspath(__file__))
ge.import_package()
cfgmod = importlib.import_module("vl_merging_amd.vilt.config")
vm = importlib.import_module("vl_merging_amd.vilt.modules.vilt_module")
ckpt_mod = importlib.import_module("vl_merging_amd.checkpoint")


def sha(t):
    return hashlib.sha256(np.ascontiguousarray(t.detach().contiguous().numpy()).tobytes()).hexdigest()


def beit_state(D, F, heads, layers, src_window, shared_table, salt=3):
    """Same synthetic BEiT-format input as tests/golden/make_golden.py::beit_state (values keyed by parameter name)."""
    R = (2 * src_window - 1) ** 2 + 3
    n = src_window * src_window + 1
    sd = {}

    def put(k, shape):
        sd[k] = torch.from_numpy(det_array(k, shape, salt))

    put("transformer.cls_token", (1, 1, D))
    put("transformer.patch_embed.proj.weight", (D, 3, 16, 16))
    put("transformer.patch_embed.proj.bias", (D,))
    for i in range(layers):
        b = "transformer.blocks.%d." % i
        for k, shp in (("gamma_1", (D,)), ("gamma_2", (D,)), ("norm1.weight", (D,)), ("norm1.bias", (D,)),
                       ("attn.q_bias", (D,)), ("attn.v_bias", (D,)), ("attn.qkv.weight", (3 * D, D)),
                       ("attn.proj.weight", (D, D)), ("attn.proj.bias", (D,)), ("norm2.weight", (D,)),
                       ("norm2.bias", (D,)), ("mlp.fc1.weight", (F, D)), ("mlp.fc1.bias", (F,)),
                       ("mlp.fc2.weight", (D, F)), ("mlp.fc2.bias", (D,))):
            put(b + k, shp)
        if not shared_table:
            put(b + "attn.relative_position_bias_table", (R, heads))
            sd[b + "attn.relative_position_index"] = torch.arange(n * n, dtype=torch.int64).view(n, n) % R
    if shared_table:
        put("transformer.rel_pos_bias.relative_position_bias_table", (R, heads))
        sd["transformer.rel_pos_bias.relative_position_index"] = torch.arange(n * n, dtype=torch.int64).view(n, n) % R
    put("transformer.fc_norm.weight", (D,))
    put("transformer.fc_norm.bias", (D,))
    return sd


def build(arch, **over):
    cfg = cfgmod.make_config(arch, vit="vit_tiny_patch16_224", image_size=224, loss_names=cfgmod._loss_names({"irtr": 1}),
                             hidden_size=192, num_heads=3, max_text_len=40, vocab_size=64,
                             vlffn_start_layer_index=10, patch_size=16, **over)
    torch.manual_seed(0)
    model = vm.ViLTransformerSS(cfg, *cfgmod.routing_configs(cfg))
    new = {k: torch.from_numpy(det_array(k, v.shape)) for k, v in model.state_dict().items()
           if v.is_floating_point() and "index" not in k and "mask_for" not in k}
    model.load_state_dict(new, strict=False)
    return model


CASES = [
    ("beit_moe_perlayer_clone", "all_moe", False, "beit", dict(use_vision_weights_for_other_modalities=True)),
    ("beit_moe_shared", "all_moe", True, "beit", dict()),
    ("beit_ufo_perlayer", "ufo", False, "beit", dict()),
    ("self_ufo_shared", "ufo", True, "self", dict()),
]


@pytest.mark.parametrize("name,arch,shared,kind,over", CASES)
def test_rekey_matches_reference(name, arch, shared, kind, over):
    digests = json.load(open(os.path.join(HERE, "golden", "ckpt_rekey_digests.json")))[name]
    tables = np.load(os.path.join(HERE, "golden", "ckpt_rekey.npz"))
    model = build(arch, **over)
    sd = beit_state(192, 768, 3, 12, 7, shared)
    if kind == "self":
        sd["text_embeddings.position_embeddings.weight"] = torch.from_numpy(
            det_array("text_embeddings.position_embeddings.weight", (60, 192), 3))
        sd["text_embeddings.position_ids"] = torch.arange(60).view(1, 60)
        res = model.modify_checkpoint_self(dict(sd))
    else:
        res = model.modify_checkpoint_beit({"state_dict": dict(sd)})
    assert sorted(res.keys()) == sorted(digests.keys())
    want_table = torch.from_numpy(tables[name + "/relative_position_bias_table"])
    assert torch.equal(res["relative_position_bias_table"], want_table)  # same bicubic operator, same inputs
    for k, (shape, dtype, digest) in digests.items():
        assert list(res[k].shape) == shape and str(res[k].dtype) == "torch." + dtype, k
        assert sha(res[k]) == digest, k
    if over.get("use_vision_weights_for_other_modalities"):
        # clones are the SAME tensor objects in the reference (no copy); layers below vlffn_start get no vl expert
        k = "transformer.blocks.11.mlp.v.fc1.weight"
        assert res[k.replace(".v.", ".l.")] is res[k] and res[k.replace(".v.", ".vl.")] is res[k]
        assert "transformer.blocks.3.mlp.vl.fc1.weight" not in res
    # the result loads into the model (strict=False as at vilt_module.py:293) and lands in the right parameters
    missing, unexpected = model.load_state_dict(res, strict=False)
    assert not [u for u in unexpected if "position_ids" not in u], unexpected
    assert torch.equal(model.relative_position_bias_table.detach(), want_table)


def test_beit_without_state_dict_returns_none():
    model = build("ufo")
    assert model.modify_checkpoint_beit({"model": {}}) is None


def test_lightning_ckpt_round_trip(tmp_path):
    model = build("ufo")
    path = os.path.join(tmp_path, "merged.ckpt")
    written = ckpt_mod.save_ckpt(path, model, global_step=123, epoch=4)
    raw = torch.load(path, map_location="cpu", weights_only=False)
    assert set(("state_dict", "global_step", "epoch", "pytorch-lightning_version", "hyper_parameters")) <= set(raw)
    assert raw["global_step"] == 123 and raw["hyper_parameters"]["config"]["hidden_size"] == 192
    sd = ckpt_mod.load_ckpt(path)
    ref = model.state_dict()
    assert sorted(sd.keys()) == sorted(ref.keys()) == sorted(written["state_dict"].keys())
    for k in ref:
        assert torch.equal(sd[k], ref[k].cpu()), k
    other = build("ufo")
    with torch.no_grad():
        for p in other.parameters():
            p.add_(1.0)
    other.load_state_dict(sd, strict=False)
    for (k, a), (_, b) in zip(other.state_dict().items(), ref.items()):
        assert torch.equal(a, b), k
    # a bare state_dict file loads through the same entry point
    torch.save(ref, os.path.join(tmp_path, "bare.pt"))
    assert sorted(ckpt_mod.load_ckpt(os.path.join(tmp_path, "bare.pt")).keys()) == sorted(ref.keys())


def test_ckpt_written_here_is_what_the_reference_loaded(tmp_path):
    """The other direction of the .ckpt contract (run.py writes last.ckpt through checkpoint.save_ckpt): the reference
    itself loaded such a file through its `load_path=` route (tests/golden/make_golden.py ckpt_written_here ->
    ckpt_written_here.json: its missing / unexpected key lists and the sha256 of every parameter it then held).
    Re-written here from the same deterministic weights, the file must carry exactly those tensors."""
    gold = json.load(open(os.path.join(HERE, "golden", "ckpt_written_here.json")))
    cfg = cfgmod.make_config("ufo", vit="vit_tiny_patch16_224", hidden_size=192, num_heads=3, vocab_size=1024, max_text_len=40,
                             patch_size=16, vlffn_start_layer_index=10, image_size=224, max_vl_text_len=40, tasks=["vl"],
                             loss_names=cfgmod._loss_names({"itm": 1, "mlm": 1, "ifm": 1}))
    model = vm.ViLTransformerSS(cfg, *cfgmod.routing_configs(cfg))
    sd = {k: torch.from_numpy(det_array(k, v.shape, 11)) for k, v in model.state_dict().items()
          if v.is_floating_point() and "index" not in k and "mask_for" not in k}
    model.load_state_dict(sd, strict=False)
    path = str(tmp_path / "last.ckpt")
    ckpt_mod.save_ckpt(path, model, global_step=7, epoch=1)
    raw = torch.load(path, map_location="cpu", weights_only=False)
    assert sorted(raw.keys()) == gold["ckpt_top_level_keys"] and raw["global_step"] == gold["global_step"] == 7
    # every parameter of the reference model was filled from the file, bit for bit
    assert set(gold["param_sha"]) == {n for n, _ in model.named_parameters()}
    for n, h in gold["param_sha"].items():
        assert sha(raw["state_dict"][n]) == h, n
    # what the reference could not find are buffers it rebuilds itself (index tables) and torchmetrics states of its
    # Lightning module; the only key it did not want is transformers-4.x's persistent position_ids
    assert all(("index" in k) or k.startswith(("train_", "val_")) or k == "mask_for_combining_temporal" for k in gold["missing_keys"])
    assert gold["unexpected_keys"] == ["text_embeddings.position_ids"]


@pytest.mark.parametrize("name,arch,shared,kind,over", [c for c in CASES if c[0] in ("beit_moe_shared", "self_ufo_shared")])
def test_load_path_dispatches_beit_and_self_adaptation(tmp_path, name, arch, shared, kind, over):
    """The FLOW of vilt_module.py:276-295: `load_path=<file>` + `use_beit_weight` / `use_self_weight` in the config must route
    the file through modify_checkpoint_beit / _self inside __init__ and load the result (strict=False) -- the parameters
    of the constructed model carry the digests the reference produced for the direct method call."""
    digests = json.load(open(os.path.join(HERE, "golden", "ckpt_rekey_digests.json")))[name]
    sd = beit_state(192, 768, 3, 12, 7, shared)
    if kind == "self":
        sd["text_embeddings.position_embeddings.weight"] = torch.from_numpy(
            det_array("text_embeddings.position_embeddings.weight", (60, 192), 3))
        sd["text_embeddings.position_ids"] = torch.arange(60).view(1, 60)
        payload, flag = dict(sd), "use_self_weight"
    else:
        payload, flag = {"state_dict": dict(sd)}, "use_beit_weight"
    path = os.path.join(tmp_path, name + ".pth")
    torch.save(payload, path)
    cfg = cfgmod.make_config(arch, loss_names=cfgmod._loss_names({"irtr": 1}), vit="vit_tiny_patch16_224", hidden_size=192,
                             num_heads=3, max_text_len=40, vocab_size=64, vlffn_start_layer_index=10, patch_size=16,
                             load_path=path, **dict(over, **{flag: True}))
    torch.manual_seed(0)
    model = vm.ViLTransformerSS(cfg, *cfgmod.routing_configs(cfg))
    assert not [u for u in model.load_info.unexpected_keys if "position_ids" not in u], model.load_info.unexpected_keys
    loaded = dict(model.named_parameters())
    hit = 0
    for k, (shape, dtype, digest) in digests.items():
        if k == "relative_position_bias_table":
            continue  # its text / cls rows are taken from the MODEL's current table (vilt_module.py:863-881): zeros at construction
        if k in loaded:
            assert sha(loaded[k].detach()) == digest, k
            hit += 1
    assert hit >= 150, hit  # every adapted tensor that names a parameter of the target architecture landed in it
    # neither flag: the same file goes down the VLMo route and must NOT be re-keyed (all_moe keys stay absent)
    if arch == "all_moe":
        assert "transformer.blocks.0.attn.v.qkv.weight" in digests and "transformer.blocks.0.attn.v.qkv.weight" in loaded
