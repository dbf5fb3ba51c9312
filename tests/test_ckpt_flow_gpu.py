"""BASELINE config 5 flow: an all_moe pre-training checkpoint (224^2) -> ufo retrieval model with merge_weights=True,
at 224^2 and at 384^2 (relative-position table resized by modify_checkpoint_vlmo), as the reference's __init__ does
(vilt_module.py:270-295)."""
import importlib
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods(pkg):
    return (importlib.import_module("vl_merging_amd.vilt.config"),
            importlib.import_module("vl_merging_amd.vilt.modules.vilt_module"))


def tiny(cfgmod, *names, **over):
    base = dict(hidden_size=192, num_heads=3, vocab_size=2048, max_text_len=40, patch_size=16, vlffn_start_layer_index=10)
    base.update(over)
    return cfgmod.make_config(*names, **base)


def test_all_moe_ckpt_to_merged_ufo_irtr(mods, tmp_path):
    cfgmod, vm = mods
    torch.manual_seed(3)
    src_cfg = tiny(cfgmod, "all_moe", vit="vit_tiny_patch16_224", image_size=224, max_vl_text_len=40, tasks=["vl"],
                   loss_names=cfgmod._loss_names({"itm": 1, "mlm": 1, "ifm": 1}))
    src = vm.ViLTransformerSS(src_cfg, *cfgmod.routing_configs(src_cfg))
    with torch.no_grad():
        src.relative_position_bias_table.normal_(0, 0.3)
    sd = {k: v.clone() for k, v in src.state_dict().items()}
    path = os.path.join(tmp_path, "all_moe.ckpt")
    torch.save({"state_dict": sd}, path)

    for size in (224, 384):
        cfg = tiny(cfgmod, "ufo", vit="vit_tiny_patch16_%d" % size, image_size=size, load_path=path, merge_weights=True,
                   merge_ratio=0.5, loss_names=cfgmod._loss_names({"irtr": 1, "itm": 0, "mlm": 0, "ifm": 0}))
        model = vm.ViLTransformerSS(cfg, *cfgmod.routing_configs(cfg))
        info = model.load_info
        # pre-training-only entries are unexpected, rebuilt index buffers are missing -- nothing else
        assert all(k.startswith(("mlm_score", "itm_score", "ifm_vl_", "logit_vl_scale", "vl_text_")) for k in info.unexpected_keys), info.unexpected_keys
        assert all("relative_position_index" in k or "mask_for_combining_temporal" in k for k in info.missing_keys), info.missing_keys
        for i in (0, 5, 11):
            for leaf, src_t in (("attn.qkv.weight", "attn.{m}.qkv.weight"), ("mlp.fc2.bias", "mlp.{m}.fc2.bias"),
                                ("norm1.weight", "norm1.{m}.weight")):
                v = sd[f"transformer.blocks.{i}." + src_t.format(m="v")]
                l = sd[f"transformer.blocks.{i}." + src_t.format(m="l")]
                if i < 10:
                    want = (0 + 0.5 * v) + 0.5 * l
                else:
                    vl = sd[f"transformer.blocks.{i}." + src_t.format(m="vl")]
                    want = ((0 + torch.tensor((2 / 3) * 0.5, dtype=torch.float32) * v)
                            + torch.tensor((2 / 3) * 0.5, dtype=torch.float32) * l) + torch.tensor(1 / 3, dtype=torch.float32) * vl
                got = dict(model.named_parameters())[f"transformer.blocks.{i}." + leaf].detach().cpu()
                assert torch.equal(got, want), (size, i, leaf)
        tab = model.relative_position_bias_table.detach()
        g = size // 16
        assert tab.shape[0] == (2 * g - 1) ** 2 + 3 + 392 + 2
        assert torch.equal(tab[-397:], sd["relative_position_bias_table"][-397:])  # text / cls rows are carried over
        if size == 224:
            assert torch.equal(tab, sd["relative_position_bias_table"])
        # the merged model runs a retrieval training step on the engine
        model = model.cuda().train()
        model.setup_engine()
        from bench import synthetic_batch
        batch = synthetic_batch(3, size, 40, 2048, 5, "cuda")["vl"]
        loss = model.training_step(batch, 0)
        loss.backward()
        assert torch.isfinite(loss.detach())
