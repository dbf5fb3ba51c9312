"""Generate the golden fixtures by running the REFERENCE (/root/reference) in this container.

Run:  python tests/golden/make_golden.py [what ...]     (what in: index merge merge_base model irtr model_base irtr_merged_base train_tiny ckpt recall batch downstream vlmo_resize schedule configs)
Outputs land next to this file.  Fixtures are DATA (inputs derive from oracle/detweights.py seeds,
expected outputs are what the reference computed); no reference source is stored.
"""
import hashlib
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

from ref_harness import import_reference, base_config, build_reference_model  # noqa: E402
from oracle.detweights import det_array, det_batch, det_gram  # noqa: E402
from oracle import synth  # noqa: E402


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ----------------------------------------------------------------------------- index buffers
def gold_index():
    out = {}
    # 480: the VQA recipe of the reference's README (:194-223, image_size=480 on the 384 ViT): a 30 x 30 window, R = 5 286
    for tag, size in (("224", 224), ("384", 384), ("480", 480)):
        cfg = base_config(vit="vit_base_patch16_%d" % min(size, 384), image_size=size, vocab_size=64,
                          max_vl_text_len=40, loss_names={"itm": 1, "mlm": 1, "ifm": 1})
        m, _ = build_reference_model(cfg, "ufo")
        for name in ("relative_position_index", "text_relative_position_index",
                     "text_imag_relative_position_index", "vl_text_imag_relative_position_index"):
            a = getattr(m, name).numpy()
            out[f"{name}_{tag}"] = a
            out[f"{name}_{tag}_sha256"] = np.array(sha(a))
            print(name, tag, a.shape, a.dtype, int(a.sum()), sha(a)[:16])
    np.savez_compressed(os.path.join(HERE, "index_buffers.npz"), **out)


# ----------------------------------------------------------------------------- merge (tiny)
def fake_self(**cfg_over):
    cfg = base_config(**cfg_over)
    return types.SimpleNamespace(hparams=types.SimpleNamespace(config=cfg))


def tiny_state(D=16, F=32, arch="all_moe", salt=0):
    shapes = synth.state_shapes(D, F, arch, R=24, vocab=32, T=8, heads=2)
    return {k: torch.from_numpy(det_array(k, s, salt)) for k, (s, dt) in shapes.items()}


MERGE_CASES = [
    dict(name="interp_r0.5", fn="merge_weights", cfg=dict(merge_ratio=0.5)),
    dict(name="interp_r0.3", fn="merge_weights", cfg=dict(merge_ratio=0.3)),
    dict(name="interp_r0.3_used_irtr", fn="merge_weights",
         cfg=dict(merge_ratio=0.3, only_activate_used_experts=True, loss_names={"irtr": 1})),
    dict(name="interp_r0.5_used_vqa", fn="merge_weights",
         cfg=dict(merge_ratio=0.5, only_activate_used_experts=True, loss_names={"vqa": 1})),
    dict(name="taskvec_l0.75", fn="sum_task_vectors", cfg=dict(sum_lambda=0.75)),
    dict(name="taskvec_l0.4_used_irtr", fn="sum_task_vectors",
         cfg=dict(sum_lambda=0.4, only_activate_used_experts=True, loss_names={"irtr": 1})),
    dict(name="regmean_a1.0", fn="regmean", cfg=dict(scaling_for_non_diag=1.0, loss_names={"irtr": 1})),
    dict(name="regmean_a0.9", fn="regmean", cfg=dict(scaling_for_non_diag=0.9, loss_names={"irtr": 1})),
    dict(name="regmean_a0.9_pretrain", fn="regmean",
         cfg=dict(scaling_for_non_diag=0.9, loss_names={"itm": 1, "mlm": 1, "ifm": 1})),
    dict(name="interp_already_ufo", fn="merge_weights", cfg=dict(merge_ratio=0.5), arch="ufo"),
]


def gold_merge():
    vm, _, _ = import_reference()
    D, F = 16, 32
    out = {}
    tmpdir = "/tmp/vlm_golden"
    os.makedirs(tmpdir, exist_ok=True)
    central = tiny_state(D, F, "ufo", salt=7)
    torch.save({"state_dict": central}, os.path.join(tmpdir, "central.ckpt"))
    # regmean grams: v and l of every layer (vl experts never get one)
    grams = {k: torch.from_numpy(det_gram(k, s[0])) for k, s in synth.gram_shapes(D, F).items()}
    torch.save(grams, os.path.join(tmpdir, "grams.pth"))
    for case in MERGE_CASES:
        sd = tiny_state(D, F, case.get("arch", "all_moe"))
        cfg = dict(case["cfg"])
        cfg["central_weight"] = os.path.join(tmpdir, "central.ckpt")
        cfg["gram_matrices"] = os.path.join(tmpdir, "grams.pth")
        me = fake_self(**cfg)
        res = getattr(vm.ViLTransformerSS, case["fn"])(me, sd)
        keys = sorted(res.keys())
        out[case["name"] + "/__keys__"] = np.array(json.dumps(keys))
        for k in keys:
            v = res[k]
            if "transformer.blocks." in k and "gamma" not in k:
                out[case["name"] + "/" + k] = v.numpy()
        print(case["name"], len(keys), "keys;", sorted({str(v.dtype) for v in res.values()}))
    np.savez_compressed(os.path.join(HERE, "merge_tiny.npz"), **out)


def gold_merge_base():
    """Full base-size all_moe -> ufo merges through the reference; only sha256 digests are kept."""
    vm, _, _ = import_reference()
    D, F = 768, 3072
    shapes = synth.block_shapes(D, F, "all_moe")
    sd = {k: torch.from_numpy(det_array(k, s)) for k, (s, dt) in shapes.items()}
    dig = {}
    for name, fn, cfg in (("interp_r0.5", "merge_weights", dict(merge_ratio=0.5)),
                          ("interp_r0.3", "merge_weights", dict(merge_ratio=0.3))):
        res = getattr(vm.ViLTransformerSS, fn)(fake_self(**cfg), sd)
        dig[name] = {k: sha(v.numpy()) for k, v in res.items() if "gamma" not in k}
        print(name, len(dig[name]))
    # task vector: central = det ufo blocks with salt 7
    cshapes = synth.block_shapes(D, F, "ufo")
    central = {k: torch.from_numpy(det_array(k, s, 7)) for k, (s, dt) in cshapes.items()}
    os.makedirs("/tmp/vlm_golden", exist_ok=True)
    torch.save(central, "/tmp/vlm_golden/central_base.ckpt")
    res = vm.ViLTransformerSS.sum_task_vectors(
        fake_self(sum_lambda=0.75, central_weight="/tmp/vlm_golden/central_base.ckpt"), sd)
    dig["taskvec_l0.75"] = {k: sha(v.numpy()) for k, v in res.items() if "gamma" not in k}
    os.remove("/tmp/vlm_golden/central_base.ckpt")
    with open(os.path.join(HERE, "merge_base_digests.json"), "w") as f:
        json.dump(dig, f, indent=0, sort_keys=True)


# ----------------------------------------------------------------------------- model
def load_det_weights(model):
    sd = model.state_dict()
    new = {}
    meta = {}
    for k, v in sd.items():
        meta[k] = (list(v.shape), str(v.dtype).replace("torch.", ""))
        if v.is_floating_point() and "index" not in k and "mask_for" not in k:
            new[k] = torch.from_numpy(det_array(k, v.shape))
    model.load_state_dict(new, strict=False)
    return meta


def to_batch(nb):
    b = {k: torch.from_numpy(v) for k, v in nb.items()}
    b["image"] = [b["image"]]
    return b


def grads_summary(model):
    out = {}
    for n, p in model.named_parameters():
        if p.grad is None:
            out[n] = None
        else:
            g = p.grad.double()
            out[n] = [float(g.norm()), float(g.sum())]
    return out


def gold_model():
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=0, world_size=1)
    vm, vit, obj = import_reference()
    for arch in ("ufo", "all_moe"):
        cfg = base_config(vit="vit_tiny_patch16_224", hidden_size=192, num_heads=3, max_vl_text_len=40,
                          max_text_len=40, vocab_size=1024, loss_names={"itm": 1, "mlm": 1, "ifm": 1},
                          tasks=["vl"], drop_rate=0.1)
        model, cfg = build_reference_model(cfg, arch)
        meta = load_det_weights(model)
        with open(os.path.join(HERE, f"keys_tiny_{arch}.json"), "w") as f:
            json.dump(meta, f, indent=0, sort_keys=True)
        model.eval()
        nb = det_batch(2, 224, 40, 1024, seed=1234)
        batch = to_batch(nb)
        out = {}
        with torch.no_grad():
            r = model.infer(batch, mask_text=False)
            for k in ("text_feats", "image_feats", "cls_feats", "raw_cls_feats"):
                out["infer/" + k] = r[k].numpy()
            r = model.infer(batch, mask_text=True)
            out["infer_mlm/text_feats"] = r["text_feats"].numpy()
            r = model.infer_image(batch)
            for k in ("image_feats", "cls_feats", "cls_vlffn_feats"):
                out["infer_image/" + k] = r[k].numpy()
            r = model.infer_text(batch)
            for k in ("text_feats", "cls_feats", "cls_vlffn_feats"):
                out["infer_text/" + k] = r[k].numpy()
            # per-op: block 0 and block 11 on a fixed hidden state
            x = torch.from_numpy(det_array("probe.x", (2, 237, 192))) * 10
            mask = torch.cat([batch["text_masks"], torch.ones(2, 197, dtype=torch.long)], 1)
            bias = model.get_rel_pos_bias(model.text_imag_relative_position_index)
            bl = torch.chunk(bias, 12, dim=0)
            for li in (0, 11):
                y, _ = model.transformer.blocks[li](x, mask=mask, type_id=2, relative_position_bias=bl[li])
                out[f"block{li}/joint"] = y.numpy()
        # one full training_step in eval mode (deterministic) + backward
        model.zero_grad()
        from vilt.modules import vilt_utils
        vilt_utils.set_task(model)
        ret = model({"vl": batch})
        total = sum(v for k, v in ret.items() if "loss" in k)
        total.backward()
        for k in ("mlm_loss", "ifm_loss", "itm_loss"):
            out["step/" + k] = np.array(float(ret[k]))
        out["step/total_loss"] = np.array(float(total))
        out["step/mlm_logits"] = ret["mlm_logits"].detach().numpy()
        out["step/itm_logits"] = ret["itm_logits"].detach().numpy()
        out["step/ifm_i2t_logits"] = ret["ifm_i2t_logits"].detach().numpy()
        gs = grads_summary(model)
        out["step/grad_summary"] = np.array(json.dumps(gs))
        named = dict(model.named_parameters())
        pick = ["relative_position_bias_table", "token_type_embeddings.weight", "transformer.cls_token",
                "transformer.blocks.0.gamma_1", "transformer.blocks.11.gamma_2", "transformer.norm.weight",
                "transformer.patch_embed.proj.bias", "logit_scale", "logit_vl_scale"]
        pick += [n for n in named if n.startswith("transformer.blocks.5.") and named[n].dim() == 1]
        for n in pick:
            if named[n].grad is not None:
                out["step/grad/" + n] = named[n].grad.numpy()
        print(arch, {k: float(out["step/" + k]) for k in ("mlm_loss", "ifm_loss", "itm_loss", "total_loss")},
              "no-grad params:", [n for n, v in gs.items() if v is None])
        np.savez_compressed(os.path.join(HERE, f"model_tiny_{arch}.npz"), **out)


def gold_irtr():
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=0, world_size=1)
    vm, vit, obj = import_reference()
    for arch in ("ufo", "all_moe"):
        cfg = base_config(vit="vit_tiny_patch16_224", hidden_size=192, num_heads=3, max_text_len=40,
                          vocab_size=1024, loss_names={"irtr": 1}, drop_rate=0.1)
        model, cfg = build_reference_model(cfg, arch)
        meta = load_det_weights(model)
        with open(os.path.join(HERE, f"keys_tiny_irtr_{arch}.json"), "w") as f:
            json.dump(meta, f, indent=0, sort_keys=True)
        model.eval()
        batch = to_batch(det_batch(3, 224, 40, 1024, seed=77))
        out = {}
        grams = {}
        if arch == "all_moe":
            # the reference's gram hook (src/cache_gram_matrices.py:246-281), re-registered here because
            # it is a closure inside main(); same module-name selection and arithmetic.
            from collections import defaultdict
            grams = defaultdict(float)
            all_keys = ["mlp.fc1", "mlp.fc1", "mlp.v.fc1", "mlp.l.fc1", "mlp.vl.fc1", "mlp.v.fc2", "mlp.l.fc2",
                        "mlp.vl.fc2", "attn", "attn.v", "attn.l", "attn.vl", "attn.proj", "attn.v.proj",
                        "attn.l.proj", "attn.vl.proj"]

            def hook_gram_input(module, input, output):
                if isinstance(input, tuple):
                    input = input[0]
                fl = input.reshape(-1, input.shape[-1]).to(torch.float64)
                grams[module.module_name] += torch.matmul(fl.T, fl).detach().cpu()

            for name, module in model.named_modules():
                if any(name.endswith(n) for n in all_keys) and ".bias" not in name:
                    module.module_name = name
                    module.register_forward_hook(hook_gram_input)
        from vilt.modules import vilt_utils
        vilt_utils.set_task(model)
        model.zero_grad()
        ret = model(batch)
        ret["irtr_loss"].backward()
        out["irtr_loss"] = np.array(float(ret["irtr_loss"]))
        out["irtr_i2t_logits"] = ret["irtr_i2t_logits"].detach().numpy()
        out["grad_summary"] = np.array(json.dumps(grads_summary(model)))
        print(arch, "irtr loss", float(ret["irtr_loss"]), "grams:", len(grams))
        if grams:
            gk = sorted(grams.keys())
            out["gram_keys"] = np.array(json.dumps(gk))
            out["gram_summary"] = np.array(json.dumps(
                {k: [list(grams[k].shape), float(grams[k].norm()), float(grams[k].sum())] for k in gk}))
            for k in ("transformer.blocks.0.attn.v", "transformer.blocks.3.attn.l.proj",
                      "transformer.blocks.7.mlp.v.fc1", "transformer.blocks.11.mlp.l.fc2"):
                out["gram/" + k] = grams[k].numpy()[:192, :192]  # fc2's [768,768]: leading block only
        np.savez_compressed(os.path.join(HERE, f"irtr_tiny_{arch}.npz"), **out)


# ----------------------------------------------------------------------------- vlmo checkpoint adaptation + schedule
def vlmo_ckpt_inputs(hidden=768, heads=12, layers=12, src_window=14, text_len=60):
    """A 224^2-pretrained VLMo checkpoint's size-dependent tensors (vilt_module.py:749-806 reads only these): the
    relative-position table of a 14x14 window with max_text_len_of_initckpt = 196, a longer text position table and the
    index buffers the reference pops."""
    n_rel = (2 * src_window - 1) ** 2 + 3
    R_src = n_rel + 2 * 196 + 2
    return {"relative_position_bias_table": det_array("ckpt224.relative_position_bias_table", (R_src, heads * layers)),
            "text_embeddings.position_embeddings.weight": det_array("ckpt224.pos", (text_len, hidden)),
            "text_embeddings.position_ids": np.arange(text_len, dtype=np.int64)[None],
            "relative_position_index": np.zeros((4, 4), dtype=np.int64),
            "text_relative_position_index": np.zeros((4, 4), dtype=np.int64),
            "text_imag_relative_position_index": np.zeros((4, 4), dtype=np.int64),
            "transformer.cls_token": det_array("ckpt224.cls", (1, 1, hidden))}


def gold_vlmo_resize():
    """modify_checkpoint_vlmo of the REFERENCE on a 224^2 checkpoint loaded into a 384^2 model: the bicubic 27x27 ->
    47x47 resize of the table body, the untouched tail rows, the truncated text positions, the popped buffers."""
    cfg = base_config(vit="vit_base_patch16_384", image_size=384, hidden_size=768, num_heads=12, max_text_len=40,
                      vocab_size=64, loss_names={"irtr": 1})
    model, cfg = build_reference_model(cfg, "ufo")
    sd = {k: torch.from_numpy(v) for k, v in vlmo_ckpt_inputs().items()}
    res = model.modify_checkpoint_vlmo({"state_dict": sd})
    out = {"__keys__": np.array(json.dumps(sorted(res.keys())))}
    for k, v in res.items():
        a = v.contiguous().numpy()
        out[k + "/sha256"] = np.array(sha(a))
        out[k + "/shape"] = np.array(a.shape)
        if k == "relative_position_bias_table":
            out[k + "/rows"] = a[::97]   # every 97th row of the resized table (the digest pins the rest)
        elif a.size <= 4096:
            out[k] = a
    print("vlmo resize:", {k: tuple(v.shape) for k, v in res.items()})
    np.savez_compressed(os.path.join(HERE, "vlmo_resize.npz"), **out)


def gold_schedule():
    """The reference's set_schedule (vilt_utils.py:225-359) on tiny models: which parameter lands in which of the four
    groups (weight decay, lr multiplier) and the learning-rate factor of the polynomial-decay-with-warm-up schedule
    at chosen steps.  AdamW = the torch.optim stub of ref_harness (only the param_groups are read)."""
    vm, vit, obj = import_reference()
    from vilt.modules import vilt_utils
    out = {}
    for tag, arch, over in (("pretrain_all_moe", "all_moe", dict(loss_names={"itm": 1, "mlm": 1, "ifm": 1}, warmup_steps=2500,
                                                                  max_steps=200000)),
                            ("vqa_ufo_mult", "ufo", dict(loss_names={"vqa": 1}, vqav2_label_size=37, lr_mult=10, warmup_steps=0.1,
                                                         max_steps=1000, all_mlp_mult=True, weight_decay_custom_modules=0.05,
                                                         learning_rate=3e-5, end_lr=1e-7, decay_power=2)),
                            ("irtr_all_moe_vlmult", "all_moe", dict(loss_names={"irtr": 1}, all_vl_mult=True, lr_mult=5,
                                                                     warmup_steps=0.1, max_steps=500))):
        cfg = base_config(vit="vit_tiny_patch16_224", hidden_size=192, num_heads=3, max_text_len=40, vocab_size=64, **over)
        model, cfg = build_reference_model(cfg, arch)
        model.trainer = types.SimpleNamespace(max_steps=cfg["max_steps"])
        (opt,), (sch,) = vilt_utils.set_schedule(model)
        ids = {id(p): n for n, p in model.named_parameters()}
        groups = []
        for g in opt.param_groups:
            groups.append({"weight_decay": g["weight_decay"], "initial_lr": g.get("initial_lr", g["lr"]),
                           "names": sorted(ids[id(p)] for p in g["params"])})
        steps = [0, 1, 10, cfg["max_steps"] // 20, cfg["max_steps"] // 10, cfg["max_steps"] // 2, cfg["max_steps"] - 1,
                 cfg["max_steps"], cfg["max_steps"] + 5]
        lam = sch["scheduler"].lr_lambdas[0]
        out[tag] = {"config": {k: cfg[k] for k in ("learning_rate", "weight_decay", "weight_decay_custom_modules", "lr_mult",
                                                   "end_lr", "decay_power", "warmup_steps", "max_steps", "all_mlp_mult",
                                                   "all_vl_mult", "all_v_mult", "all_l_mult", "beta_2")},
                    "arch": arch, "loss_names": {k: v for k, v in cfg["loss_names"].items() if v},
                    "groups": groups, "steps": steps, "lr_factor": [float(lam(s)) for s in steps]}
        print(tag, [len(g["names"]) for g in groups], out[tag]["lr_factor"][:4])
    with open(os.path.join(HERE, "schedule_groups.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)


# ----------------------------------------------------------------------------- RegMean at base size
REGMEAN_BASE_LAYERS = (0, 11)  # the reference takes ~4 s per 3072^2 inverse on this box: two layers pin the arithmetic


def regmean_base_inputs(layers=REGMEAN_BASE_LAYERS):
    """Base-size (D = 768, F = 3072) all_moe block weights and SPD Gram matrices for the listed layers; every other
    layer of the 12 the merge walks is passed through as an already-merged (ufo-shaped) tensor (the reference's
    `else: later_weight = state_dict[later_name]` branch, vilt_module.py:425-427)."""
    D, F = 768, 3072
    moe = synth.block_shapes(D, F, "all_moe")
    ufo = synth.block_shapes(D, F, "ufo")
    sd = {}
    for k, (shp, dt) in moe.items():
        if int(k.split(".")[2]) in layers:
            sd[k] = det_array(k, shp)
    for k, (shp, dt) in ufo.items():
        if int(k.split(".")[2]) not in layers and "gamma" not in k:
            sd[k] = det_array(k, shp, 5)
    for k, (shp, dt) in moe.items():
        if "gamma" in k:
            sd[k] = det_array(k, shp)
    grams = {k: det_gram(k, s[0]) for k, s in synth.gram_shapes(D, F).items() if int(k.split(".")[2]) in layers}
    return sd, grams


def gold_regmean_base():
    vm, _, _ = import_reference()
    sd, grams = regmean_base_inputs()
    tmp = "/tmp/vlm_golden"
    os.makedirs(tmp, exist_ok=True)
    torch.save({k: torch.from_numpy(v) for k, v in grams.items()}, os.path.join(tmp, "grams_base.pth"))
    me = fake_self(scaling_for_non_diag=0.9, loss_names={"irtr": 1}, gram_matrices=os.path.join(tmp, "grams_base.pth"))
    res = vm.ViLTransformerSS.regmean(me, {k: torch.from_numpy(v) for k, v in sd.items()})
    os.remove(os.path.join(tmp, "grams_base.pth"))
    out = {}
    for k, v in res.items():
        if v.dtype == torch.float64 and int(k.split(".")[2]) in REGMEAN_BASE_LAYERS:
            a = v.numpy()
            out[k + "/norm"] = np.array(np.linalg.norm(a))
            out[k + "/rows"] = a[[0, a.shape[0] // 2, a.shape[0] - 1]][:, :256]
            out[k + "/colsum"] = a.sum(0)[:256]
            print(k, a.shape, float(out[k + "/norm"]))
    np.savez_compressed(os.path.join(HERE, "regmean_base.npz"), **out)


# ----------------------------------------------------------------------------- base width (the benchmarked size)
BASE = dict(vit="vit_base_patch16_384", image_size=384, hidden_size=768, num_heads=12, max_text_len=40,
            vocab_size=1024, drop_rate=0.1)
IMG_ROWS = 16   # image_feats rows kept: every 16th token (the fixture stays small; cls row 0 included)
MLM_COLS = 8    # mlm logits kept: every 8th vocabulary column


def _dist_once():
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=0, world_size=1)


def _step_record(model, batch, out, prefix, picks):
    """One training_step of the reference (whatever mode the model is in) + backward; losses, logits, per-parameter
    gradient (norm, sum), a few whole gradients."""
    from vilt.modules import vilt_utils
    model.zero_grad()
    vilt_utils.set_task(model)
    ret = model({"vl": batch})
    total = sum(v for k, v in ret.items() if "loss" in k)
    total.backward()
    for k in ("mlm_loss", "ifm_loss", "itm_loss"):
        out[prefix + k] = np.array(float(ret[k]))
    out[prefix + "total_loss"] = np.array(float(total))
    out[prefix + "mlm_logits"] = ret["mlm_logits"].detach().numpy()[..., ::MLM_COLS]
    out[prefix + "itm_logits"] = ret["itm_logits"].detach().numpy()
    out[prefix + "ifm_i2t_logits"] = ret["ifm_i2t_logits"].detach().numpy()
    gs = grads_summary(model)
    out[prefix + "grad_summary"] = np.array(json.dumps(gs))
    named = dict(model.named_parameters())
    for n in picks:
        if named[n].grad is not None:
            g = named[n].grad.numpy()
            out[prefix + "grad/" + n] = g[::8] if n == "relative_position_bias_table" else g
    return ret, total, gs


def gold_model_base():
    """The benchmarked configuration (hidden 768, 12 heads, 384^2: N = 617, R = 2294) through the reference, eval
    mode, B = 2 (hard negatives forced): features of every pass, one training_step with backward.  Also the same step
    under the reference's own training precision (fp16 autocast, run.py precision=16) -> amp_reference_errors.json:
    what the reference's AMP path deviates from its fp32 path on exactly the quantities the GPU tests compare."""
    _dist_once()
    vm, vit, obj = import_reference()
    amp = {}
    for arch in ("ufo", "all_moe"):
        cfg = base_config(max_vl_text_len=40, loss_names={"itm": 1, "mlm": 1, "ifm": 1}, tasks=["vl"], **BASE)
        model, cfg = build_reference_model(cfg, arch)
        meta = load_det_weights(model)
        with open(os.path.join(HERE, f"keys_base_{arch}.json"), "w") as f:
            json.dump(meta, f, indent=0, sort_keys=True)
        model.eval()
        batch = to_batch(det_batch(2, 384, 40, 1024, seed=4321))
        out = {"img_rows": np.array(IMG_ROWS), "mlm_cols": np.array(MLM_COLS)}

        def feats():
            o = {}
            r = model.infer(batch, mask_text=False)
            o["infer/text_feats"] = r["text_feats"].numpy()
            o["infer/image_feats"] = r["image_feats"].numpy()[:, ::IMG_ROWS]
            o["infer/cls_feats"] = r["cls_feats"].numpy()
            r = model.infer_image(batch)
            o["infer_image/image_feats"] = r["image_feats"].numpy()[:, ::IMG_ROWS]
            for k in ("cls_feats", "cls_vlffn_feats"):
                o["infer_image/" + k] = r[k].numpy()
            r = model.infer_text(batch)
            o["infer_text/text_feats"] = r["text_feats"].numpy()
            for k in ("cls_feats", "cls_vlffn_feats"):
                o["infer_text/" + k] = r[k].numpy()
            return o

        with torch.no_grad():
            out.update(feats())
        named = dict(model.named_parameters())
        picks = ["relative_position_bias_table", "token_type_embeddings.weight", "transformer.cls_token",
                 "transformer.blocks.0.gamma_1", "transformer.blocks.11.gamma_2", "transformer.norm.weight",
                 "transformer.patch_embed.proj.bias", "logit_scale", "logit_vl_scale"]
        picks += [n for n in named if n.startswith("transformer.blocks.5.") and named[n].dim() == 1]
        ret, total, gs = _step_record(model, batch, out, "step/", picks)
        print(arch, "base", {k: float(out["step/" + k]) for k in ("mlm_loss", "ifm_loss", "itm_loss", "total_loss")})
        np.savez_compressed(os.path.join(HERE, f"model_base_{arch}.npz"), **out)
        # ---- the reference's own fp16-AMP path on the same check (errors only, nothing else is kept)
        a = {}
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.float16):
            fa = feats()
        for k, v in fa.items():
            ref = out[k]
            a["feat/" + k] = float(np.abs(v.astype(np.float32) - ref).max() / np.abs(ref).max())
        o2 = {}
        with torch.autocast("cpu", dtype=torch.float16):
            _step_record(model, batch, o2, "step/", picks)
        for k in ("mlm_loss", "ifm_loss", "itm_loss", "total_loss"):
            a["loss/" + k] = abs(float(o2["step/" + k]) - float(out["step/" + k]))
        for k in ("mlm_logits", "itm_logits", "ifm_i2t_logits"):
            a["logits/" + k] = float(np.abs(o2["step/" + k].astype(np.float32) - out["step/" + k]).max()
                                     / np.abs(out["step/" + k]).max())
        g2 = json.loads(str(o2["step/grad_summary"]))
        rel = {n: abs(g2[n][0] - v[0]) / (v[0] + 1e-12) for n, v in gs.items() if v is not None and g2[n] is not None}
        big = [r for n, r in rel.items() if gs[n][0] > 0.05]
        small = [r for n, r in rel.items() if gs[n][0] <= 0.05]
        a["grad_norm_rel/max_norm_gt_0.05"] = max(big)
        a["grad_norm_rel/median_norm_gt_0.05"] = float(np.median(big))
        a["grad_norm_rel/max_norm_le_0.05"] = max(small) if small else 0.0
        full = []
        for k in out:
            if k.startswith("step/grad/"):
                mx = float(np.abs(out[k]).max())
                full.append(float(np.abs(o2[k] - out[k]).max()) / (mx + 1e-12))
        a["grad_full_rel_to_max/max"] = max(full)
        amp[arch] = a
        print(arch, "fp16-AMP reference vs fp32 reference:", json.dumps(a, indent=0))
    with open(os.path.join(HERE, "amp_reference_errors.json"), "w") as f:
        json.dump(amp, f, indent=1, sort_keys=True)


def gold_irtr_merged_base():
    """configs[4] at base size: two-expert all_moe weights -> the reference's merge_weights (ratio 0.5) -> ufo model
    with the irtr objective at 384^2, one step with backward (B = 3).  The test redoes the merge with the HIP kernel."""
    _dist_once()
    vm, vit, obj = import_reference()
    from vilt.modules import vilt_utils
    cfg = base_config(loss_names={"irtr": 1}, **BASE)
    moe, _ = build_reference_model(cfg, "all_moe")
    meta_moe = load_det_weights(moe)
    with open(os.path.join(HERE, "keys_base_irtr_all_moe.json"), "w") as f:
        json.dump(meta_moe, f, indent=0, sort_keys=True)
    merged = vm.ViLTransformerSS.merge_weights(fake_self(merge_ratio=0.5), {k: v.clone() for k, v in moe.state_dict().items()})
    ufo, _ = build_reference_model(cfg, "ufo")
    res = ufo.load_state_dict(merged, strict=False)
    print("merged -> ufo: missing", [k for k in res.missing_keys][:8], "unexpected", len(res.unexpected_keys))
    ufo.eval()
    vilt_utils.set_task(ufo)
    batch = to_batch(det_batch(3, 384, 40, 1024, seed=99))
    ufo.zero_grad()
    ret = ufo(batch)
    ret["irtr_loss"].backward()
    out = {"irtr_loss": np.array(float(ret["irtr_loss"])), "irtr_i2t_logits": ret["irtr_i2t_logits"].detach().numpy(),
           "grad_summary": np.array(json.dumps(grads_summary(ufo))),
           "merged_sha/transformer.blocks.3.mlp.fc1.weight": np.array(sha(merged["transformer.blocks.3.mlp.fc1.weight"].numpy())),
           "merged_sha/transformer.blocks.11.attn.qkv.weight": np.array(sha(merged["transformer.blocks.11.attn.qkv.weight"].numpy()))}
    with torch.no_grad():
        out["img_cls_feats"] = ufo.infer_image_ft(batch)["cls_feats"].numpy()
        out["txt_cls_feats"] = ufo.infer_text_ft(batch)["cls_feats"].numpy()
    print("irtr merged base: loss", float(ret["irtr_loss"]))
    np.savez_compressed(os.path.join(HERE, "irtr_merged_base.npz"), **out)


GRAM_BASE_PICKS = ("transformer.blocks.0.attn.v", "transformer.blocks.0.attn.l.proj", "transformer.blocks.0.mlp.v.fc1",
                   "transformer.blocks.0.mlp.l.fc2", "transformer.blocks.11.attn.l", "transformer.blocks.11.attn.v.proj",
                   "transformer.blocks.11.mlp.l.fc1", "transformer.blocks.11.mlp.v.fc2")


def gold_gram_base():
    """configs[3]'s capture leg at BASE width (hidden 768, F 3072, 384^2): the reference's gram hook
    (src/cache_gram_matrices.py:246-281; a closure inside its main(), so the same module selection and arithmetic are
    re-registered here) on the all_moe irtr model, one eval forward over B = 3.  Kept: key list, (shape, norm, sum) of all
    96 matrices, the leading 256 x 256 block of a v / l attn, proj, fc1, fc2 Gram at layers 0 and 11 (float32)."""
    _dist_once()
    vm, vit, obj = import_reference()
    from collections import defaultdict
    from vilt.modules import vilt_utils
    cfg = base_config(loss_names={"irtr": 1}, **BASE)
    model, _ = build_reference_model(cfg, "all_moe")
    load_det_weights(model)  # same deterministic weights as keys_base_irtr_all_moe.json
    model.eval()
    grams = defaultdict(float)
    all_keys = ["mlp.fc1", "mlp.fc1", "mlp.v.fc1", "mlp.l.fc1", "mlp.vl.fc1", "mlp.v.fc2", "mlp.l.fc2", "mlp.vl.fc2", "attn",
                "attn.v", "attn.l", "attn.vl", "attn.proj", "attn.v.proj", "attn.l.proj", "attn.vl.proj"]

    def hook_gram_input(module, input, output):
        if isinstance(input, tuple):
            input = input[0]
        fl = input.reshape(-1, input.shape[-1]).to(torch.float64)
        grams[module.module_name] += torch.matmul(fl.T, fl).detach().cpu()

    for name, module in model.named_modules():
        if any(name.endswith(n) for n in all_keys) and ".bias" not in name:
            module.module_name = name
            module.register_forward_hook(hook_gram_input)
    vilt_utils.set_task(model)
    batch = to_batch(det_batch(3, 384, 40, 1024, seed=99))
    with torch.no_grad():
        model(batch)
    gk = sorted(grams.keys())
    out = {"gram_keys": np.array(json.dumps(gk)),
           "gram_summary": np.array(json.dumps({k: [list(grams[k].shape), float(grams[k].norm()), float(grams[k].sum())]
                                                for k in gk}))}
    for k in GRAM_BASE_PICKS:
        out["gram/" + k] = grams[k].numpy()[:256, :256].astype(np.float32)
    print("gram base:", len(gk), "matrices;", {k: list(grams[k].shape) for k in GRAM_BASE_PICKS[2:4]})
    np.savez_compressed(os.path.join(HERE, "gram_base.npz"), **out)


def gold_train_tiny():
    """TRAIN mode (DropPath + text-embedding dropout live) with injected masks (ref_harness.inject_train_masks):
    one training_step of the reference at tiny width, B = 2."""
    _dist_once()
    vm, vit, obj = import_reference()
    from ref_harness import inject_train_masks, INJECT
    for arch in ("ufo", "all_moe"):
        cfg = base_config(vit="vit_tiny_patch16_224", hidden_size=192, num_heads=3, max_vl_text_len=40,
                          max_text_len=40, vocab_size=1024, loss_names={"itm": 1, "mlm": 1, "ifm": 1},
                          tasks=["vl"], drop_rate=0.1)
        model, cfg = build_reference_model(cfg, arch)
        load_det_weights(model)  # same weights as model_tiny_<arch>.npz / keys_tiny_<arch>.json
        model.train()
        st = inject_train_masks(model)
        batch = to_batch(det_batch(2, 224, 40, 1024, seed=1234))
        out = {"mlm_cols": np.array(MLM_COLS),
               "drop_path_probs": np.array([float(getattr(b.drop_path, "drop_prob", 0.0)) for b in model.transformer.blocks])}
        named = dict(model.named_parameters())
        picks = ["transformer.blocks.0.gamma_1", "transformer.blocks.11.gamma_2", "transformer.norm.weight",
                 "token_type_embeddings.weight"]
        _step_record(model, batch, out, "step/", picks)
        INJECT["keep"] = None
        print(arch, "train-mode", {k: float(out["step/" + k]) for k in ("mlm_loss", "ifm_loss", "itm_loss", "total_loss")},
              "infer calls", st["n_infer"])
        np.savez_compressed(os.path.join(HERE, f"train_tiny_{arch}.npz"), **out)


# ----------------------------------------------------------------------------- checkpoint re-keying
def beit_state(D, F, heads, layers, src_window, shared_table, salt=3):
    """A BEiT-format state_dict as the reference expects it at vilt_module.py:808 (keys already under "transformer."):
    per-layer (pt22k) or shared (pt22k_ft22k) relative-position tables at `src_window`, fc_norm instead of norm."""
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


CKPT_CASES = [
    dict(name="beit_moe_perlayer_clone", arch="all_moe", shared=False, fn="modify_checkpoint_beit",
         cfg=dict(use_vision_weights_for_other_modalities=True)),
    dict(name="beit_moe_shared", arch="all_moe", shared=True, fn="modify_checkpoint_beit", cfg=dict()),
    dict(name="beit_ufo_perlayer", arch="ufo", shared=False, fn="modify_checkpoint_beit", cfg=dict()),
    dict(name="self_ufo_shared", arch="ufo", shared=True, fn="modify_checkpoint_self", cfg=dict()),
]


def gold_ckpt():
    out = {}
    meta = {}
    for case in CKPT_CASES:
        cfg = base_config(vit="vit_tiny_patch16_224", hidden_size=192, num_heads=3, max_text_len=40, vocab_size=64,
                          loss_names={"irtr": 1}, **case["cfg"])
        model, cfg = build_reference_model(cfg, case["arch"])
        load_det_weights(model)
        sd = beit_state(192, 768, 3, 12, 7, case["shared"])
        if case["fn"] == "modify_checkpoint_self":
            # the reference reads these two keys unconditionally (vilt_module.py:979): a 60-position text table to truncate
            sd["text_embeddings.position_embeddings.weight"] = torch.from_numpy(
                det_array("text_embeddings.position_embeddings.weight", (60, 192), 3))
            sd["text_embeddings.position_ids"] = torch.arange(60).view(1, 60)
            res = model.modify_checkpoint_self(dict(sd))
        else:
            res = model.modify_checkpoint_beit({"state_dict": dict(sd)})
        digest = {k: [list(v.shape), str(v.dtype).replace("torch.", ""), sha(v.detach().contiguous().numpy())]
                  for k, v in res.items()}
        meta[case["name"]] = digest
        out[case["name"] + "/relative_position_bias_table"] = res["relative_position_bias_table"].detach().numpy()
        print(case["name"], len(sd), "->", len(res), "keys; table", tuple(res["relative_position_bias_table"].shape))
    np.savez_compressed(os.path.join(HERE, "ckpt_rekey.npz"), **out)
    with open(os.path.join(HERE, "ckpt_rekey_digests.json"), "w") as f:
        json.dump(meta, f, indent=0, sort_keys=True)


def written_here_state(arch="ufo"):
    """What tests/test_checkpoint_cpu.py writes with checkpoint.save_ckpt: the build's own model (CPU construction, no
    engine), deterministic weights keyed by parameter name."""
    import importlib
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.import_package()
    cfgmod = importlib.import_module("vl_merging_amd.vilt.config")
    vmod = importlib.import_module("vl_merging_amd.vilt.modules.vilt_module")
    cfg = cfgmod.make_config(arch, vit="vit_tiny_patch16_224", hidden_size=192, num_heads=3, vocab_size=1024, max_text_len=40,
                             patch_size=16, vlffn_start_layer_index=10, image_size=224, max_vl_text_len=40, tasks=["vl"],
                             loss_names=cfgmod._loss_names({"itm": 1, "mlm": 1, "ifm": 1}))
    model = vmod.ViLTransformerSS(cfg, *cfgmod.routing_configs(cfg))
    sd = {k: torch.from_numpy(det_array(k, v.shape, 11)) for k, v in model.state_dict().items()
          if v.is_floating_point() and "index" not in k and "mask_for" not in k}
    model.load_state_dict(sd, strict=False)
    return model


def gold_ckpt_written_here():
    """The other direction of the .ckpt contract: a `last.ckpt` written by the build's checkpoint.save_ckpt must load in the
    REFERENCE through its own `load_path=` route (torch.load -> modify_checkpoint_vlmo -> load_state_dict(strict=False),
    vilt_module.py:270-295).  Kept: the reference's missing / unexpected key lists and the sha256 of every parameter the
    reference model holds after the load."""
    import importlib
    _dist_once()
    ours = written_here_state("ufo")
    ck = importlib.import_module("vl_merging_amd.checkpoint")
    tmp = os.path.join(HERE, "_written_here.ckpt")
    ck.save_ckpt(tmp, ours, global_step=7, epoch=1)
    try:
        cfg = base_config(vit="vit_tiny_patch16_224", hidden_size=192, num_heads=3, max_text_len=40, vocab_size=1024,
                          max_vl_text_len=40, tasks=["vl"], loss_names={"itm": 1, "mlm": 1, "ifm": 1}, load_path=tmp)
        ref, _ = build_reference_model(cfg, "ufo")  # the reference's __init__ loads the file
        raw = torch.load(tmp, map_location="cpu", weights_only=False)
    finally:
        os.remove(tmp)
    import vilt.modules.vilt_module as vm_ref  # noqa
    # the reference logs these itself (vilt_module.py:293-295); recompute them the same way for the fixture
    info = ref.load_state_dict(ref.modify_checkpoint_vlmo({"state_dict": {k: v.clone() for k, v in raw["state_dict"].items()}}),
                               strict=False)
    out = {"missing_keys": sorted(info.missing_keys), "unexpected_keys": sorted(info.unexpected_keys),
           "ckpt_top_level_keys": sorted(raw.keys()), "global_step": int(raw["global_step"]),
           "param_sha": {n: sha(p.detach().contiguous().numpy()) for n, p in ref.named_parameters()}}
    print("written-here ckpt in the reference: missing", out["missing_keys"], "unexpected", out["unexpected_keys"],
          len(out["param_sha"]), "parameters")
    with open(os.path.join(HERE, "ckpt_written_here.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)


# ----------------------------------------------------------------------------- retrieval recall
class _RecallDset(torch.utils.data.Dataset):
    """Stand-in for the datamodule's no-false test dataset: items are what BaseDataset.collate would have produced."""

    def __init__(self, items):
        self.items = items

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]

    @staticmethod
    def collate(batch, mlm_collator=None):
        out = {"img_index": [b["img_index"] for b in batch]}
        for k in batch[0]:
            if k == "img_index":
                continue
            if k == "image":
                out[k] = [torch.stack([b[k] for b in batch])]
            else:
                out[k] = torch.stack([b[k] for b in batch])
        return out


def recall_inputs(n_img=10, caps=3, size=224, T=40, vocab=1024):
    ib = det_batch(n_img, size, T, vocab, seed=91)
    tb = det_batch(n_img * caps, size, T, vocab, seed=92)
    texts = [{"text_ids": torch.from_numpy(tb["text_ids"][j]), "text_masks": torch.from_numpy(tb["text_masks"][j]),
              "text_labels": torch.from_numpy(tb["text_labels"][j]), "img_index": j // caps}
             for j in range(n_img * caps)]
    images = [{"image": torch.from_numpy(ib["image"][i]), "img_index": i} for i in range(n_img)]
    return texts, images


def gold_recall():
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=0, world_size=1)
    vm, vit, obj = import_reference()
    out = {}
    for arch in ("ufo", "all_moe"):
        cfg = base_config(vit="vit_tiny_patch16_224", hidden_size=192, num_heads=3, max_text_len=40,
                          vocab_size=1024, loss_names={"irtr": 1}, drop_rate=0.1)
        model, cfg = build_reference_model(cfg, arch)
        load_det_weights(model)  # same weights as irtr_tiny_<arch>.npz / keys_tiny_irtr_<arch>.json
        model.eval()
        texts, images = recall_inputs()

        class DM:
            tokenizer = None
            mlm_collator = None

            def make_no_false_test_dset(self, image_only=False):
                return _RecallDset(images if image_only else texts)

        model.trainer = types.SimpleNamespace(datamodule=types.SimpleNamespace(dms=[DM()]))
        r = obj.compute_irtr_recall(model, split="test")
        out[arch + "/recalls"] = np.array([float(x) for x in r])  # ir_r1, ir_r5, ir_r10, tr_r1, tr_r5, tr_r10
        with torch.no_grad():
            tf = model.infer_text_ft({"text_ids": torch.stack([t["text_ids"] for t in texts]),
                                      "text_masks": torch.stack([t["text_masks"] for t in texts]),
                                      "text_labels": torch.stack([t["text_labels"] for t in texts])})["cls_feats"]
            imf = model.infer_image_ft({"image": [torch.stack([i["image"] for i in images])],
                                        "text_masks": texts[0]["text_masks"][None]})["cls_feats"]
        out[arch + "/txt_cls_feats"] = tf.numpy()
        out[arch + "/img_cls_feats"] = imf.numpy()
        out[arch + "/tiids"] = np.array([t["img_index"] for t in texts])
        out[arch + "/iids"] = np.array([i["img_index"] for i in images])
        print(arch, "recalls", out[arch + "/recalls"])
    np.savez_compressed(os.path.join(HERE, "irtr_recall_tiny.npz"), **out)


# ----------------------------------------------------------------------------- batch contract
def gold_batch():
    """The reference's BaseDataset (index mapping, get_suite, collate) over a synthetic Arrow shard written by
    vl_merging_amd.vilt.datasets.write_synthetic_shard (seeded data, regenerated by the test), with the synthetic
    tokenizer and transformers' DataCollatorForLanguageModeling under fixed seeds.
    Run with PYTHONHASHSEED=1: the reference's collate() walks a Python SET of keys, so which text key reaches the MLM
    collator first is hash-seed dependent; seed 1 gives "text" before "false_text_0" (ArrowDataset's fixed order)."""
    import importlib
    import random
    import tempfile
    import __graft_entry__ as ge
    ge.import_package()
    ds = importlib.import_module("vl_merging_amd.vilt.datasets")
    import_reference()
    from vilt.datasets.base_dataset import BaseDataset
    from transformers import DataCollatorForLanguageModeling
    d = tempfile.mkdtemp()
    ds.write_synthetic_shard(os.path.join(d, "coco_caption_karpathy_train.arrow"), 6, 3, image_hw=(48, 64), seed=1)
    ds.write_synthetic_shard(os.path.join(d, "vg.arrow"), 4, 2, image_hw=(40, 40), seed=2)
    tok = ds.build_synthetic_tokenizer(os.path.join(d, "vocab.txt"))
    ref = BaseDataset(d, ["square_transform"], 32, ["coco_caption_karpathy_train", "vg"], patch_size=16,
                      num_mask_patches=0, max_mask_patches_per_block=None, min_mask_patches_per_block=0,
                      dvae_image_size=16, text_column_name="caption", remove_duplicate=False, max_text_len=12,
                      draw_false_image=1, draw_false_text=1)
    ref.tokenizer = tok
    random.seed(7)
    items = [ref.get_suite(i) for i in (0, 4, 5, 17, 19, 25)]
    torch.manual_seed(11)
    batch = ref.collate(items, DataCollatorForLanguageModeling(tok, mlm=True, mlm_probability=0.4))
    out = {"n_samples": np.array(len(ref)), "index_mapper": np.array([[i, j] for i, j in ref.index_mapper.values()]),
           "table_names": np.array(json.dumps(ref.table_names))}
    lists = {}
    for k, v in batch.items():
        if isinstance(v, torch.Tensor):
            out["batch/" + k] = v.numpy()
        elif isinstance(v, list) and len(v) == 1 and isinstance(v[0], torch.Tensor):
            out["batch_img/" + k] = v[0].numpy()
        else:
            lists[k] = v
    out["batch_lists"] = np.array(json.dumps(lists, sort_keys=True))
    np.savez_compressed(os.path.join(HERE, "batch_contract.npz"), **out)
    print("batch keys:", sorted(batch.keys()))


# ----------------------------------------------------------------------------- VQA / NLVR2 heads
def downstream_batches():
    nb = det_batch(3, 224, 40, 1024, seed=55)
    nb2 = det_batch(3, 224, 40, 1024, seed=56)
    vqa = dict(nb)
    vqa_labels = [[3, 17], [0], [5, 6, 30]]
    vqa_scores = [[1.0, 0.3], [0.6], [0.9, 0.3, 0.3]]
    nlvr = {k: v for k, v in nb.items() if k != "image"}
    return nb, nb2, vqa_labels, vqa_scores, [1, 0, 1]


def gold_downstream():
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=0, world_size=1)
    vm, vit, obj = import_reference()
    from vilt.modules import vilt_utils
    nb, nb2, vqa_labels, vqa_scores, answers = downstream_batches()
    out = {}
    for task in ("vqa", "nlvr2"):
        cfg = base_config(vit="vit_tiny_patch16_224", hidden_size=192, num_heads=3, max_text_len=40, vocab_size=1024,
                          vqav2_label_size=37, loss_names={task: 1}, drop_rate=0.1)
        model, cfg = build_reference_model(cfg, "ufo")
        meta = load_det_weights(model)
        with open(os.path.join(HERE, f"keys_tiny_{task}_ufo.json"), "w") as f:
            json.dump(meta, f, indent=0, sort_keys=True)
        model.eval()
        vilt_utils.set_task(model)
        model.zero_grad()
        if task == "vqa":
            batch = to_batch(nb)
            batch["vqa_labels"], batch["vqa_scores"] = vqa_labels, vqa_scores
        else:
            batch = {k: torch.from_numpy(v) for k, v in nb.items() if k != "image"}
            batch["image_0"] = [torch.from_numpy(nb["image"])]
            batch["image_1"] = [torch.from_numpy(nb2["image"])]
            batch["answers"] = answers
            batch["table_name"] = ["nlvr2_dev"] * 3
        ret = model(batch)
        loss = ret[task + "_loss"]
        loss.backward()
        out[task + "/loss"] = np.array(float(loss))
        out[task + "/logits"] = ret[task + "_logits"].detach().numpy()
        out[task + "/grad_summary"] = np.array(json.dumps(grads_summary(model)))
        print(task, "loss", float(loss), tuple(ret[task + "_logits"].shape))
    np.savez_compressed(os.path.join(HERE, "downstream_tiny_ufo.npz"), **out)


# ----------------------------------------------------------------------------- sacred configuration surface
def gold_configs():
    """Every @ex.config / @ex.named_config function of the reference's src/vilt/config.py evaluated the way sacred does
    (the function's locals are the entries; names with a leading underscore are not), captured as JSON: the default
    config and each named config's own entries.  sacred is not installed: the stub Experiment only collects the functions."""
    import importlib.util

    class Experiment:
        def __init__(self, name):
            self.default, self.named = None, {}

        def config(self, fn):
            self.default = fn
            return fn

        def named_config(self, fn):
            self.named[fn.__name__] = fn
            return fn

    saved = sys.modules.get("sacred")
    sys.modules["sacred"] = types.SimpleNamespace(Experiment=Experiment)
    try:
        spec = importlib.util.spec_from_file_location("_ref_vilt_config", "/root/reference/src/vilt/config.py")
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        if saved is None:
            del sys.modules["sacred"]
        else:
            sys.modules["sacred"] = saved

    def entries(fn):
        got = {}

        def prof(frame, event, arg):
            if event == "return" and frame.f_code is fn.__code__:
                got.update(frame.f_locals)
        sys.setprofile(prof)
        try:
            fn()
        finally:
            sys.setprofile(None)
        return {k: v for k, v in got.items() if not k.startswith("_")}

    out = {"default": entries(mod.ex.default), "named": {n: entries(f) for n, f in mod.ex.named.items()}}
    with open(os.path.join(HERE, "named_configs.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("configs: default %d keys, %d named configs" % (len(out["default"]), len(out["named"])))


if __name__ == "__main__":
    what = sys.argv[1:] or ["index", "merge", "merge_base", "model", "irtr"]
    torch.manual_seed(0)
    for w in what:
        {"index": gold_index, "merge": gold_merge, "merge_base": gold_merge_base, "model": gold_model,
         "irtr": gold_irtr, "model_base": gold_model_base, "irtr_merged_base": gold_irtr_merged_base,
         "train_tiny": gold_train_tiny, "regmean_base": gold_regmean_base, "vlmo_resize": gold_vlmo_resize, "schedule": gold_schedule,
         "gram_base": gold_gram_base, "ckpt_written_here": gold_ckpt_written_here, "ckpt": gold_ckpt, "recall": gold_recall, "batch": gold_batch, "downstream": gold_downstream, "configs": gold_configs}[w]()
