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
    assert c2["global_step"] == 5 and c2["lr_schedulers"][0]["last_epoch"] == 5 and c2["optimizer_states"][0]["vlm_step"] == 5
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


def _reference_groups(model, vu):
    """The reference's optimizer_grouped_parameters (vilt_utils.py:272-312): four filters over named_parameters()."""
    heads = vu.head_names(model.hparams.config)
    groups = [[] for _ in range(4)]
    for n, p in model.named_parameters():
        groups[vu.param_group_of(n, heads)].append((n, p))
    return groups


def test_optimizer_states_are_torch_layout_both_ways(run_mod, pkg, shards, tmp_path):
    """`optimizer_states[0]` of a last.ckpt written here loads into a torch.optim optimizer built the reference's way (what
    Lightning's resume_from_checkpoint does with it), and a state_dict written BY such an optimizer (no vlm_* keys: a
    reference-written last.ckpt) loads here with every moment at its parameter's place."""
    cfgmod = importlib.import_module("vl_merging_amd.vilt.config")
    vm = importlib.import_module("vl_merging_amd.vilt.modules.vilt_module")
    vu = importlib.import_module("vl_merging_amd.vilt.modules.vilt_utils")
    ck = importlib.import_module("vl_merging_amd.checkpoint")
    args = PRETRAIN + ["data_root=" + shards, "log_dir=" + str(tmp_path / "result")]
    r1 = run_mod.main(args + ["steps=2"])
    ckpt = ck.load_file(r1["last_ckpt"])
    osd = ckpt["optimizer_states"][0]
    cfg = cfgmod.parse_cli(PRETRAIN[1:])
    ref_model = vm.ViLTransformerSS(cfg, *cfgmod.routing_configs(cfg))  # CPU, plain nn.Parameters
    groups = _reference_groups(ref_model, vu)
    assert [len(g["params"]) for g in osd["param_groups"]] == [len(g) for g in groups]
    assert all(isinstance(k, int) for k in osd["state"]) and osd["state"], "torch keys optimizer state by parameter index"
    t_opt = torch.optim.AdamW([{"params": [p for _, p in g]} for g in groups], lr=1e-3)
    t_opt.load_state_dict({"state": osd["state"], "param_groups": osd["param_groups"]})  # torch validates sizes and ids
    names = [n for g in groups for n, _ in g]
    params = [p for g in groups for _, p in g]
    for i, st in osd["state"].items():
        assert st["exp_avg"].shape == params[i].shape and st["step"] == 2, names[i]
        assert torch.equal(t_opt.state[params[i]]["exp_avg"], st["exp_avg"])
    assert osd["vlm_names"] == names
    i_qkv = names.index("transformer.blocks.0.attn.qkv.weight")
    assert float(osd["state"][i_qkv]["exp_avg"].abs().max()) > 0 and float(osd["state"][i_qkv]["exp_avg_sq"].abs().max()) > 0
    assert names.index("transformer.mask_token") not in osd["state"]  # never received a gradient: no Adam state (HF AdamW)

    # the other direction: a torch-written state (one step of torch.optim.AdamW on known gradients) -> this engine
    for p in params:
        p.grad = torch.full_like(p, 0.25)
    params[names.index("transformer.mask_token")].grad = None
    t2 = torch.optim.AdamW([{"params": [p for _, p in g], "lr": 1e-3, "initial_lr": 1e-3, "weight_decay": 0.01} for g in groups],
                           betas=(0.9, 0.98), eps=1e-8)
    t2.step()
    foreign = t2.state_dict()
    assert "vlm_names" not in foreign
    model = vm.ViLTransformerSS(cfg, *cfgmod.routing_configs(cfg)).cuda()
    model.setup_engine()
    (opt,), _ = vu.set_schedule(model, max_steps=8)
    opt.load_state_dict(foreign)
    assert opt.step_count == 1
    f = model._flat
    for n in ("transformer.blocks.0.attn.qkv.weight", "text_embeddings.word_embeddings.weight", "itm_score.fc.bias"):
        o, k = f.offsets[n]
        want = foreign["state"][names.index(n)]
        assert torch.equal(opt.m[o:o + k].cpu(), want["exp_avg"].reshape(-1)), n
        assert torch.equal(opt.v[o:o + k].cpu(), want["exp_avg_sq"].reshape(-1)), n
    o, k = f.offsets["transformer.mask_token"]
    assert float(opt.m[o:o + k].abs().max()) == 0.0 and "transformer.mask_token" in opt.inactive_parameters()
    # a state written for another model must not load silently
    bad = {"state": foreign["state"], "param_groups": [dict(g, params=g["params"][:-1]) for g in foreign["param_groups"]]}
    with pytest.raises(ValueError):
        opt.load_state_dict(bad)
    with pytest.raises(ValueError):
        opt.load_state_dict({"moments": {}})


def test_periodic_atomic_save_and_weights_only_resume(run_mod, pkg, shards, tmp_path, capsys, monkeypatch):
    """last.ckpt is rewritten every save interval through a temporary file (never a partial file at the resume path), and a
    checkpoint WITHOUT optimizer_states still restores step, epoch, schedule and the place in the epoch."""
    ck = importlib.import_module("vl_merging_amd.checkpoint")
    log_dir = str(tmp_path / "result")
    args = PRETRAIN + ["data_root=" + shards, "log_dir=" + log_dir]
    monkeypatch.setenv("VLM_SAVE_EVERY", "2")
    r1 = run_mod.main(args + ["steps=3"])
    out = capsys.readouterr().out
    assert "(global_step 2)" in out and "(global_step 3)" in out  # the interval save and the final one
    d = os.path.dirname(r1["last_ckpt"])
    assert os.listdir(d) == ["last.ckpt"], os.listdir(d)  # no temporary file left behind
    ckpt = ck.load_file(r1["last_ckpt"])
    assert ckpt["global_step"] == 3
    del ckpt["optimizer_states"]
    torch.save(ckpt, r1["last_ckpt"])
    monkeypatch.delenv("VLM_SAVE_EVERY")
    r2 = run_mod.main(args + ["steps=2", "resume_during_pretraining=True"])
    out = capsys.readouterr().out
    assert "holds no optimizer_states" in out
    assert r2["global_step"] == 5
    c2 = ck.load_file(r2["last_ckpt"])
    assert c2["global_step"] == 5 and c2["lr_schedulers"][0]["last_epoch"] == 5 and c2["optimizer_states"][0]["vlm_step"] == 5
    r3 = run_mod.main(PRETRAIN + ["data_root=" + shards, "log_dir=" + str(tmp_path / "straight"), "steps=5"])
    assert r1["seen_raw_index"] + r2["seen_raw_index"] == r3["seen_raw_index"]


def test_sharded_run_saves_full_optimizer_state_and_resumes(pkg, shards, tmp_path):
    """use_sharded_training (run.py:231-232) on two ranks (one device, gloo): last.ckpt carries the FULL Adam state gathered from
    both ranks' shards, and a sharded resume continues from it -- the same parameters as four uninterrupted sharded steps."""
    import subprocess
    ck = importlib.import_module("vl_merging_amd.checkpoint")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(VLM_BENCH_ONE_DEVICE="1", VLM_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    base = ["with", "task_test_vit_tiny_mlm_itm_ifm_square_randaug_base_vl", "ufo", "vocab_size=2048", "per_gpu_batchsize=2",
            "batch_size=4", "max_steps=8", "warmup_steps=0", "learning_rate=1e-3", "vl_mlm_prob=0.3", "use_sharded_training=True",
            "drop_rate=0.0"]

    def launch(log_dir, extra, port):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "vl-merging_amd", "run.py")] + base + ["log_dir=" + log_dir] + extra
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        return r.stdout

    a = str(tmp_path / "a")
    launch(a, ["steps=2"], 29611)
    path = os.path.join(a, os.listdir(a)[0], "version_0", "checkpoints", "last.ckpt")
    ckpt = ck.load_file(path)
    osd = ckpt["optimizer_states"][0]
    assert osd["vlm_step"] == 2 and len(osd["state"]) > 100
    # every active parameter's moments are there, whichever rank owned its chunk
    zero = [osd["vlm_names"][i] for i, st in osd["state"].items() if float(st["exp_avg_sq"].abs().max()) == 0.0]
    assert not zero, zero[:5]
    out = launch(a, ["steps=2", "resume_during_pretraining=True"], 29612)
    assert "resumed optimizer at global_step 2" in out
    c2 = ck.load_file(path)
    assert c2["global_step"] == 4 and c2["optimizer_states"][0]["vlm_step"] == 4


@pytest.mark.parametrize("named,arch,losskey", [
    ("task_finetune_vqa_square_randaug_base_image384_ufo", "ufo", "vqa"),
    ("task_finetune_vqa_square_randaug_base_image384", "all_moe", "vqa"),
    ("task_finetune_nlvr2_square_randaug_base", "ufo", "nlvr2"),
])
def test_downstream_named_configs_reach_their_heads_from_the_cli(run_mod, pkg, tmp_path, named, arch, losskey):
    """README.md:205-229 of the reference evaluates merged models with `run.py with task_finetune_vqa_... / nlvr2_...`: the
    named configs exist here, build the VQA / NLVR2 heads (vilt_module.py:300-337) and train on the synthetic batch's
    down-stream fields (no Arrow shards for these datasets: SURVEY.md 2, data modules out of scope)."""
    args = ["with", named, arch, "vit=vit_tiny_patch16_224", "hidden_size=192", "num_heads=3", "image_size=224", "vocab_size=2048",
            "vqav2_label_size=37", "per_gpu_batchsize=3", "batch_size=3", "max_steps=4", "warmup_steps=0", "learning_rate=1e-3",
            "log_dir=" + str(tmp_path / "result")]
    # (the VQA recipes say use_moe=False, config.py:249; `all_moe` comes after them on the command line and wins, like sacred)
    r = run_mod.main(args + ["steps=2"])
    assert r["global_step"] == 2 and r["loss"] == r["loss"] and r["loss"] > 0
    ck = importlib.import_module("vl_merging_amd.checkpoint")
    sd = ck.load_file(r["last_ckpt"])["state_dict"]
    head = "vqa_classifier" if losskey == "vqa" else "nlvr2_classifier"
    assert any(k.startswith(head + ".") for k in sd), sorted(sd)[:5]
    if losskey == "nlvr2":
        assert sd["token_type_embeddings.weight"].shape[0] == 3  # vilt_module.py:332-337: a row for the second image
