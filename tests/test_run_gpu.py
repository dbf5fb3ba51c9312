"""`run.py` as the reference's entry point (src/run.py:141-295): Arrow shards -> training steps with gradient
accumulation -> Lightning-layout last.ckpt -> reload / resume -> test_only retrieval evaluation.
Reference: run.py:160-163 (datamodule), :189-195 (ModelCheckpoint save_last), :218-223 + :280 (resume), :290-295."""
import importlib
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def run_mod(pkg):
    sys.path.insert(0, ROOT)
    return importlib.import_module("vl_merging_amd.run")


@pytest.fixture()
def shards(pkg, tmp_path):
    ds = importlib.import_module("vl_merging_amd.vilt.datasets")
    d = str(tmp_path / "data")
    ds.write_synthetic_shard(os.path.join(d, "synthetic_0.arrow"), 10, 3, image_hw=(48, 64), seed=1)
    ds.write_synthetic_shard(os.path.join(d, "synthetic_1.arrow"), 6, 2, image_hw=(40, 40), seed=2)
    ds.build_synthetic_tokenizer(os.path.join(d, "vocab.txt"))
    return d


PRETRAIN = ["with", "task_test_vit_tiny_mlm_itm_ifm_square_randaug_base_vl", "ufo", "vocab_size=2048", "per_gpu_batchsize=2",
            "batch_size=4", "max_steps=8", "warmup_steps=0", "learning_rate=1e-3", "vl_mlm_prob=0.3"]


def test_fit_save_reload_resume(run_mod, pkg, shards, tmp_path):
    cfgmod = importlib.import_module("vl_merging_amd.vilt.config")
    vm = importlib.import_module("vl_merging_amd.vilt.modules.vilt_module")
    ck = importlib.import_module("vl_merging_amd.checkpoint")
    log_dir = str(tmp_path / "result")
    args = PRETRAIN + ["data_root=" + shards, "log_dir=" + log_dir]
    r1 = run_mod.main(args + ["steps=3"])
    assert r1["global_step"] == 3 and r1["loss"] == r1["loss"]
    path = r1["last_ckpt"]
    assert path.endswith(os.path.join("version_0", "checkpoints", "last.ckpt")) and "_seed1_from_" in path
    ckpt = ck.load_file(path)
    # Lightning's layout (what the reference's load_path / resume_from_checkpoint read)
    for k in ("state_dict", "optimizer_states", "lr_schedulers", "global_step", "epoch", "hyper_parameters",
              "pytorch-lightning_version"):
        assert k in ckpt, k
    assert ckpt["global_step"] == 3 and ckpt["lr_schedulers"][0]["last_epoch"] == 3
    assert ckpt["hyper_parameters"]["config"]["per_gpu_batchsize"] == 2
    assert all(v.device.type == "cpu" for v in ckpt["state_dict"].values())
    # (a) reload into a FRESH model through the reference's load path: bit-identical parameters
    cfg = cfgmod.parse_cli(PRETRAIN[1:] + ["load_path=" + path])
    fresh = vm.ViLTransformerSS(cfg, *cfgmod.routing_configs(cfg))
    assert not [k for k in fresh.load_info.missing_keys if "index" not in k and "mask_for" not in k], fresh.load_info.missing_keys
    assert not fresh.load_info.unexpected_keys
    for n, p in fresh.named_parameters():
        assert torch.equal(p.detach(), ckpt["state_dict"][n]), n
    # (b) resume: picks version_0/checkpoints/last.ckpt, continues the step count, the schedule and Adam's moments
    r2 = run_mod.main(args + ["steps=2", "resume_during_pretraining=True"])
    assert r2["global_step"] == 5 and r2["last_ckpt"] == path  # save_last overwrites the run's last.ckpt
    c2 = ck.load_file(path)
    assert c2["global_step"] == 5 and c2["lr_schedulers"][0]["last_epoch"] == 5 and c2["optimizer_states"][0]["step"] == 5
    moved = [n for n in ckpt["state_dict"] if ckpt["state_dict"][n].is_floating_point()
             and not torch.equal(ckpt["state_dict"][n], c2["state_dict"][n])]
    assert "transformer.blocks.0.attn.qkv.weight" in moved
    # (c) against 5 uninterrupted steps: the resumed run continues the epoch where the first one stopped (same samples in
    # the same micro-batches; the MLM masks and the sampled negatives come from the process's RNG and differ)
    r3 = run_mod.main(PRETRAIN + ["data_root=" + shards, "log_dir=" + str(tmp_path / "straight"), "steps=5"])
    assert r1["seen_raw_index"] + r2["seen_raw_index"] == r3["seen_raw_index"] and len(r3["seen_raw_index"]) == 10
    assert abs(r2["loss"] - r3["loss"]) < 1.5


def test_test_only_irtr_recall(run_mod, pkg, shards, tmp_path):
    args = ["with", "task_finetune_irtr_coco_square_randaug_base_image384", "ufo", "vit=vit_tiny_patch16_224", "hidden_size=192",
            "num_heads=3", "image_size=224", "vocab_size=2048", "per_gpu_batchsize=3", "data_root=" + shards,
            "log_dir=" + str(tmp_path / "r"), "test_only=True"]
    res = run_mod.main(args)
    assert res["test/samples"] == 10 * 3 + 6 * 2
    assert res["test/irtr_loss"] == res["test/irtr_loss"] and res["test/irtr_loss"] > 0
    for k in ("ir_r1", "ir_r5", "ir_r10", "tr_r1", "tr_r5", "tr_r10"):
        assert 0.0 <= res["recalls/" + k] <= 1.0
    assert res["recalls/ir_r10"] >= res["recalls/ir_r1"] and res["recalls/tr_r10"] >= res["recalls/tr_r1"]
