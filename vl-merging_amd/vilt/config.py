"""Configuration surface of the reference (src/vilt/config.py): same keys, same defaults, ALL of its named configs, and
sacred's `with <named_config ...> key=value ...` command-line grammar (later entries win,
config.py:611 "need to be added at the end").  sacred itself is not required.
"""
import ast
import copy


def _loss_names(d):
    ret = {"itm": 0, "ifm": 0, "mlm": 0, "vqa": 0, "nlvr2": 0, "irtr": 0, "mim": 0, "image_only_mim": 0,
           "text_only_mlm": 0, "img_cls": 0, "mnc": 0, "mld": 0}
    ret.update(d)
    return ret


def default_config():
    """config.py:25-168."""
    return dict(
        exp_name="vlmo", seed=1, datasets=["coco", "vg", "sbu", "gcc"],
        loss_names=_loss_names({"itm": 1, "ifm": 1, "mlm": 1}), batch_size=1024,
        train_transform_keys=["square_transform_randaug"], val_transform_keys=["square_transform"], image_size=224,
        max_image_len=-1, patch_size=32, draw_false_image=0, image_only=False, img_cls_label_size=1000,
        vqav2_label_size=3129, max_text_len=40, max_text_len_of_initckpt=196, tokenizer="bert-base-uncased",
        vocab_size=30522, whole_word_masking=False, mlm_prob=0.15, draw_false_text=0, vl_mlm_weight=1, ifm_weight=1,
        num_frames=1, max_vl_text_len=None, use_temporal_roll_module=False, vl_mlm_prob=0.15,
        vit="vit_base_patch16_224", hidden_size=768, num_heads=12, num_layers=12, mlp_ratio=4, drop_rate=0.1,
        vlffn_start_layer_index=-1, optim_type="adamw", beta_2=0.98, learning_rate=1e-4, weight_decay=0.01,
        weight_decay_custom_modules=0.01, decay_power=1, max_epoch=100, max_steps=200000, warmup_steps=2500, end_lr=0,
        lr_mult=1, use_cpu=False, all_mlp_mult=False, all_vl_mult=False, all_v_mult=False, all_l_mult=False,
        get_recall_metric=False, resume_from=None, fast_dev_run=False, val_check_interval=1.0, test_only=False,
        validation_only=False, use_sharded_training=False, use_webdataset=False, resume_during_pretraining=False,
        limit_val_batches=1.0, limit_train_batches=1.0, data_root="", data_roots=None, log_dir="result",
        per_gpu_batchsize=0, num_gpus=1, num_nodes=1, load_path="", num_workers=8, precision=16, compute_memory=False,
        get_middle_representation=False, get_block_representation=False, get_finegrained_representation=False,
        representation_name="tmp", use_beit_weight=False, use_self_weight=False, use_ufo=False,
        separate_inference=True, use_moe=False, self_attn_for_single_mode=False,
        use_vision_weights_for_other_modalities=False, in_attn=False, in_ffn=True, merge_weights=False, merge_ratio=0.5,
        sum_task_vectors=False, central_weight=None, sum_lambda=1, only_activate_used_experts=False, regmean=False,
        gram_matrices=None, scaling_for_non_diag=1, use_custom_ln_attn=False, use_custom_ln_ffn=False,
        discrete_vae_weight_path="", num_mask_patches=75, max_mask_patches_per_block=None,
        min_mask_patches_per_block=16, dvae_image_size=112, tasks=None, random_initialization=False,
    )


NAMED_CONFIGS = {}


def named_config(fn):
    NAMED_CONFIGS[fn.__name__] = fn
    return fn


# ---- task configs (config.py:171-609) ---------------------------------------------------------------------------------------
# Every named config of the reference, entry for entry (tests/test_host_cpu.py compares each with the reference's own sacred
# function evaluated by tests/golden/make_golden.py configs -> named_configs.json).  The site-specific `data_roots` /
# `discrete_vae_weight_path` strings are part of that contract; `data_root=<dir>` on the command line is what run.py reads.
_RANDAUG = dict(train_transform_keys=["square_transform_randaug"], val_transform_keys=["square_transform"])
_CODE224 = "/storage/linjli/data/vilt/pretrain_arrows_code224/"


def _task(exp_name, datasets, losses, vit, image_size=224, vlffn=10, **more):
    d = dict(_RANDAUG, exp_name=exp_name, datasets=datasets, loss_names=_loss_names(losses), vit=vit, image_size=image_size,
             patch_size=16, vlffn_start_layer_index=vlffn)
    d.update(more)
    return d


def _finetune(exp_name, datasets, losses, vit, **more):  # the fine-tuning recipes: steps from the epochs, 10 % warm-up
    return _task(exp_name, datasets, losses, vit, **dict(dict(max_steps=None, warmup_steps=0.1, use_sharded_training=False), **more))


_LARGE = dict(hidden_size=1024, num_heads=16, num_layers=24, vlffn=21)


@named_config
def task_mlm_itm_ifm_square_randaug_base():  # config.py:171-186
    return _task("mlm_itm_ifm_square_randaug_base", ["coco", "vg", "sbu", "gcc"], {"itm": 1, "mlm": 1, "ifm": 1},
                 "vit_base_patch16_224", batch_size=1024, max_epoch=10, max_image_len=196, max_text_len_of_initckpt=196)


def _nlvr2(tag, size, lr):  # config.py:189-226
    return _finetune("finetune_nlvr2_square_randaug_base" + tag, ["nlvr2"], {"nlvr2": 1}, "vit_base_patch16_%d" % size,
                     image_size=size, batch_size=128, max_epoch=10, draw_false_image=0, learning_rate=lr)


@named_config
def task_finetune_nlvr2_square_randaug_base():
    return _nlvr2("", 224, 1e-4)


@named_config
def task_finetune_nlvr2_square_randaug_base_image384():
    return _nlvr2("_image384", 384, 5e-5)


def _vqa(tag, lr, vit="vit_base_patch16_384", **more):  # config.py:229-249, 294-340 (image_size stays 224 in all three)
    return _finetune("finetune_vqa_square_randaug_" + tag, ["vqa"], {"vqa": 1}, vit, batch_size=512, max_epoch=10,
                     draw_false_image=0, learning_rate=lr, val_check_interval=1.0, lr_mult=10, use_moe=False, **more)


@named_config
def task_finetune_vqa_square_randaug_base_image384():
    return _vqa("base_image384", 1e-4)


@named_config
def task_finetune_vqa_square_randaug_base_image384_ufo():
    return _vqa("base_image384_ufo", 3e-5)


@named_config
def task_finetune_vqa_square_randaug_large_image384_ufo():
    return _vqa("large_image384_ufo", 3e-5, vit="vit_large_patch16_384", **_LARGE)


@named_config
def task_all_in_one_pretraining():  # config.py:252-291
    return _finetune("all_in_one_pretraining", [["imagenet"], ["bookcorpus", "wikipedia"], ["webvid", "sbu", "gcc", "coco", "vg"]],
                     {"image_only_mim": 1, "text_only_mlm": 1, "mim": 1, "itm": 1, "mlm": 1, "ifm": 1}, "vit_base_patch16_224",
                     train_transform_keys=["square_transform_randaug_mim"], val_transform_keys=["square_transform_mim"],
                     tasks=["v", "l", "vl"],
                     data_roots=[["/storage/v-yilinsung/imagenet-22k/"],
                                 ["/storage/v-yilinsung/huggingface/bookcorpus/", "/storage/v-yilinsung/huggingface/wikipedia_20200501_en/"],
                                 ["/storage/linjli/data/mtp_vlp_ray/pretrain/composite/"] + [_CODE224] * 4],
                     discrete_vae_weight_path="/storage/v-yilinsung/dall_e_tokenizer_weight", batch_size=512, max_epoch=10,
                     draw_false_image=0, learning_rate=1e-4, val_check_interval=1.0, use_moe=False, random_initialization=True,
                     max_vl_text_len=40)


def _imagenet(size, lr, mult, **more):  # config.py:343-387 (both exp_names end in _ufo, both name the 384 ViT)
    return _finetune("finetune_imagenet_square_randaug_base_image%d_ufo" % size, ["imagenet1k"], {"img_cls": 1},
                     "vit_base_patch16_384", image_size=size, batch_size=512, max_epoch=100, draw_false_image=0,
                     learning_rate=lr, val_check_interval=1.0, lr_mult=mult, use_moe=False, **more)


@named_config
def task_finetune_imagenet_square_randaug_base_image384():
    return _imagenet(384, 1e-3, 10)


@named_config
def task_finetune_imagenet_square_randaug_base_image224():
    return _imagenet(224, 3e-3, 1, warmup_steps=0.2, weight_decay=0.05)


def _irtr(tag, dataset, size, epochs, lr, losses=None, vit=None, **more):  # config.py:390-496
    return _finetune("finetune_irtr_" + tag, [dataset], losses or {"irtr": 1.0}, vit or "vit_base_patch16_%d" % size,
                     image_size=size, batch_size=1024, max_epoch=epochs, get_recall_metric=True, draw_false_text=0,
                     learning_rate=lr, **more)


@named_config
def task_finetune_irtr_f30k_square_randaug_base():
    return _irtr("f30k_square_randaug_base", "f30k", 224, 10, 5e-5)


@named_config
def task_finetune_irtr_msrvtt_frame_square_randaug_base():
    return _irtr("msrvtt_frame_square_randaug_base", "msrvtt", 224, 10, 5e-5, losses={"irtr": 1.0, "ifm": 1.0, "itm": 1.0},
                 use_moe=False)


@named_config
def task_finetune_irtr_f30k_square_randaug_base_image384():
    return _irtr("f30k_square_randaug_base_image384", "f30k", 384, 40, 5e-5)


@named_config
def task_finetune_irtr_f30k_square_randaug_large_image384():
    return _irtr("f30k_square_randaug_large_image384", "f30k", 384, 10, 5e-5, vit="vit_large_patch16_384", **_LARGE)


@named_config
def task_finetune_irtr_coco_square_randaug_base_image384():  # BASELINE configs[4]
    return _irtr("coco_square_randaug_base_image384", "coco", 384, 20, 2e-5)


def _vl_pretrain(exp_name, datasets, roots, dvae, vit="vit_base_patch16_224", **more):  # config.py:499-609
    return _finetune(exp_name, [datasets], {"itm": 1, "mlm": 1, "ifm": 1}, vit, tasks=["vl"], data_roots=[roots],
                     discrete_vae_weight_path=dvae, batch_size=512, max_epoch=10, draw_false_image=0, learning_rate=2e-4,
                     val_check_interval=1.0, max_vl_text_len=40, max_text_len=40, **more)


_TINY = dict(vit="vit_tiny_patch16_224", hidden_size=192, num_heads=3)


@named_config
def task_mlm_itm_ifm_square_randaug_base_vl():  # BASELINE configs[1] / [2]
    return _vl_pretrain("mlm_itm_ifm_square_randaug_base_vl", ["sbu", "gcc", "coco", "vg"], [_CODE224] * 4,
                        "/storage/v-yilinsung/dall_e_tokenizer_weight")


@named_config
def task_test_vit_tiny_mlm_itm_ifm_square_randaug_base_vl():
    return _vl_pretrain("vit_tiny_mlm_itm_ifm_square_randaug_base_vl", ["f30k"], ["/workspace/dataset/dataset/flickr30k/"], "",
                        **_TINY)


@named_config
def task_vit_tiny_pretraining():
    return _vl_pretrain("vit_tiny_pretraining", ["sbu", "gcc", "coco", "vg"], [_CODE224] * 4, "", **_TINY)


def _step(max_epoch, max_steps, warmup=None):
    d = dict(max_epoch=max_epoch, max_steps=max_steps)
    if warmup is not None:
        d["warmup_steps"] = warmup
    return d


@named_config
def step10k():
    return _step(100, 10000)


@named_config
def step25k():
    return _step(100, 25000)


@named_config
def step50k():
    return _step(100, 50000, 625)


@named_config
def step100k():
    return _step(100, 100000, 1250)


@named_config
def step150k():
    return _step(150, 150000, 1875)


@named_config
def step200k():
    return _step(200, 200000, 2500)


@named_config
def step400k():
    return _step(300, 400000, 5000)


@named_config
def epoch100():
    return dict(max_epoch=100, warmup_steps=10000)


@named_config
def ufo():  # config.py:664-667
    return dict(use_ufo=True, separate_inference=True)


@named_config
def ln_moe():
    return dict(use_moe=False, in_attn=False, in_ffn=False, use_custom_ln_attn=True, use_custom_ln_ffn=True,
                separate_inference=True)


@named_config
def attn_moe():
    return dict(use_moe=True, in_attn=True, in_ffn=False, use_custom_ln_attn=True, use_custom_ln_ffn=False,
                self_attn_for_single_mode=True)


@named_config
def ffn_moe():
    return dict(use_moe=True, in_attn=False, in_ffn=True, use_custom_ln_attn=False, use_custom_ln_ffn=True,
                separate_inference=True)


@named_config
def all_moe():  # config.py:703-711
    return dict(use_moe=True, in_attn=True, in_ffn=True, use_custom_ln_ffn=True, use_custom_ln_attn=True,
                self_attn_for_single_mode=True)


def _parse_value(txt):
    try:
        return ast.literal_eval(txt)
    except (ValueError, SyntaxError):
        return txt


def make_config(*updates, **overrides):
    """default config <- named configs / dicts in order <- overrides ("a.b" keys reach into nested dicts)."""
    cfg = default_config()
    for u in updates:
        if isinstance(u, str):
            if u not in NAMED_CONFIGS:
                raise KeyError("unknown named config %r" % u)
            u = NAMED_CONFIGS[u]()
        cfg.update(copy.deepcopy(u))
    for k, v in overrides.items():
        _assign(cfg, k, v)
    return cfg


def _assign(cfg, key, value):
    parts = key.split(".")
    d = cfg
    for p in parts[:-1]:
        d = d[p]
    if parts[-1] not in d:
        raise KeyError("config has no entry %r" % key)  # sacred rejects unknown keys too
    d[parts[-1]] = value


def parse_cli(argv):
    """`[with] name ... key=value ...`, applied left to right like sacred."""
    cfg = default_config()
    args = list(argv)
    if args and args[0] == "with":
        args = args[1:]
    for a in args:
        if "=" in a:
            k, v = a.split("=", 1)
            _assign(cfg, k, _parse_value(v))
        else:
            if a not in NAMED_CONFIGS:
                raise KeyError("unknown named config %r" % a)
            cfg.update(copy.deepcopy(NAMED_CONFIGS[a]()))
    return cfg


class UFOConfig:  # src/vilt/ufo/config.py
    tasks = ["vl"]
    tasks_for_shallow_layers = ["v", "l"]
    tasks_for_deep_layers = ["v", "l", "vl"]
    separate_inference = False


class MOEConfig:  # src/vilt/moe/config.py
    tasks = ["vl"]
    tasks_for_shallow_layers = ["v", "l"]
    tasks_for_deep_layers = ["v", "l", "vl"]
    in_attn = False
    in_ffn = True
    self_attn_for_single_mode = False


class LNConfig:  # src/vilt/custom_ln/config.py
    tasks = ["vl"]
    tasks_for_shallow_layers = ["v", "l"]
    tasks_for_deep_layers = ["v", "l", "vl"]
    use_custom_ln_attn = False
    use_custom_ln_ffn = False


def routing_configs(_config):
    """run.py:165-183 -> (ufo_config, ln_config, moe_config)."""
    ln_config = moe_config = ufo_config = None
    if _config["use_ufo"]:
        ufo_config = UFOConfig()
        ufo_config.separate_inference = _config["separate_inference"]
    if _config["use_custom_ln_attn"] or _config["use_custom_ln_ffn"]:
        ln_config = LNConfig()
        ln_config.use_custom_ln_attn = _config["use_custom_ln_attn"]
        ln_config.use_custom_ln_ffn = _config["use_custom_ln_ffn"]
    if _config["use_moe"]:
        moe_config = MOEConfig()
        moe_config.in_attn = _config["in_attn"]
        moe_config.in_ffn = _config["in_ffn"]
        moe_config.self_attn_for_single_mode = _config["self_attn_for_single_mode"]
        moe_config.separate_inference = _config["separate_inference"]
    return ufo_config, ln_config, moe_config
