"""Import the reference (/root/reference) in THIS container so that golden vectors can be captured.

TEST INFRASTRUCTURE ONLY; never imported by tests at run time (the reference does not exist on the
GPU box).  The reference needs pytorch_lightning / timm / torchvision / fairscale / torchmetrics /
sacred, none of which are installed; the stubs below provide exactly the names the reference
touches at import and in the code paths we drive.  Fidelity patches applied, each a
reference-*environment* issue, not an algorithm change:

  (1) torch.Tensor.get_device -> t.device     objectives.py:320,414 crash on CPU otherwise
  (2) BertEmbeddings.forward -> transformers-4.x semantics for position_embedding_type="rel_pos"
      (word + bert-token-type(0) -> LayerNorm -> dropout; NO absolute position embedding)
  (3) transformers.optimization.AdamW stub     (removed upstream; only imported by vilt_utils.py:4)
"""
import os
import sys
import types
import copy

import torch
import torch.nn as nn

REF_SRC = "/root/reference/src"
# train-mode goldens: the DropPath stub below asks INJECT["keep"](B, keep_prob) for its 0/1 draw when it is set
INJECT = {"keep": None}


def _mod(name, **attrs):
    import importlib.machinery
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)  # importlib.util.find_spec() must not choke on stubs
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    # transformers first (before a fake torchvision exists)
    import transformers
    import transformers.models.bert.modeling_bert as mb
    import transformers.optimization as topt

    if not hasattr(topt, "AdamW"):
        topt.AdamW = torch.optim.AdamW

    def bert_embeddings_forward_4x(self, input_ids=None, token_type_ids=None, position_ids=None,
                                   inputs_embeds=None, past_key_values_length=0):
        emb = self.word_embeddings(input_ids)
        tt = self.token_type_embeddings(torch.zeros_like(input_ids))
        return self.dropout(self.LayerNorm(emb + tt))

    mb.BertEmbeddings.forward = bert_embeddings_forward_4x
    torch.Tensor.get_device = lambda t: t.device

    # ---- pytorch_lightning -------------------------------------------------------------------
    class LightningModule(nn.Module):
        def __init__(self):
            super().__init__()
            self.hparams = types.SimpleNamespace()

        def save_hyperparameters(self):
            import inspect
            frame = inspect.currentframe().f_back
            loc = frame.f_locals
            for k in ("config", "ufo_config", "ln_config", "moe_config"):
                if k in loc:
                    setattr(self.hparams, k, copy.deepcopy(loc[k]) if k == "config" else loc[k])

        def log(self, *a, **k):
            pass

        @property
        def device(self):
            return next(self.parameters()).device

    class LightningDataModule:
        pass

    pl = _mod("pytorch_lightning", LightningModule=LightningModule, LightningDataModule=LightningDataModule)
    _mod("pytorch_lightning.utilities")
    _mod("pytorch_lightning.utilities.distributed", rank_zero_info=lambda *a, **k: None)

    # ---- timm --------------------------------------------------------------------------------
    class DropPath(nn.Module):
        def __init__(self, drop_prob=0.0):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            if self.drop_prob == 0.0 or not self.training:
                return x
            keep = 1 - self.drop_prob
            shape = (x.shape[0],) + (1,) * (x.ndim - 1)
            if INJECT["keep"] is not None:  # fixed keep masks (train-mode goldens): see inject_train_masks()
                rnd = INJECT["keep"](x.shape[0], keep).to(x.dtype).view(shape)
                return x * rnd / keep
            rnd = x.new_empty(shape).bernoulli_(keep)
            return x * rnd.div_(keep)

    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

    def trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0):
        return nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b)

    _mod("timm")
    _mod("timm.data", IMAGENET_DEFAULT_MEAN=(0.485, 0.456, 0.406), IMAGENET_DEFAULT_STD=(0.229, 0.224, 0.225))
    _mod("timm.models")
    _mod("timm.models.helpers", load_pretrained=lambda *a, **k: None)
    _mod("timm.models.layers", StdConv2dSame=nn.Conv2d, DropPath=DropPath, to_2tuple=to_2tuple,
         trunc_normal_=trunc_normal_)
    _mod("timm.models.resnet", resnet26d=None, resnet50d=None)
    _mod("timm.models.resnetv2", ResNetV2=None)
    _mod("timm.models.registry", register_model=lambda f: f)

    # ---- torchvision / fairscale / torchmetrics ----------------------------------------------
    # torchvision.transforms: the three operators the reference's `square_transform` composes
    # (transforms/square_transform.py:12-19, transforms/utils.py:48-50), restated from torchvision's published
    # semantics for PIL inputs: Resize = Image.resize((w, h), interpolation); ToTensor = uint8 HWC -> float32 CHW / 255;
    # Normalize = (x - mean) / std per channel.  Everything else the transforms package imports is a named placeholder
    # (the augmenting pipelines are never run by the fixtures).
    import numpy as _np

    class Compose:
        def __init__(self, ts):
            self.ts = self.transforms = list(ts)

        def __call__(self, x):
            for t in self.transforms:
                x = t(x)
            return x

    class Resize:
        def __init__(self, size, interpolation=None):
            self.size, self.interpolation = size, interpolation

        def __call__(self, img):
            h, w = self.size
            return img.resize((w, h), self.interpolation)

    class ToTensor:
        def __call__(self, img):
            a = _np.asarray(img, dtype=_np.uint8)
            return torch.from_numpy(a.transpose(2, 0, 1).copy()).to(torch.float32).div(255)

    class Normalize:
        def __init__(self, mean, std):
            self.mean = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
            self.std = torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)

        def __call__(self, t):
            return (t - self.mean) / self.std

    class _Placeholder:
        def __init__(self, *a, **k):
            pass

    tv = _mod("torchvision")
    tv.transforms = _mod("torchvision.transforms", Compose=Compose, Resize=Resize, ToTensor=ToTensor,
                         Normalize=Normalize, RandomResizedCrop=_Placeholder, RandomHorizontalFlip=_Placeholder,
                         ColorJitter=_Placeholder, RandomCrop=_Placeholder, CenterCrop=_Placeholder)
    _mod("torchvision.transforms.functional")
    _mod("dall_e", load_model=lambda *a, **k: None)
    _mod("cv2")  # transforms/randaugment.py imports it at module scope; the augmenting pipelines are never run
    _mod("dall_e.utils", map_pixels=lambda x: x)
    _mod("fairscale")
    _mod("fairscale.nn", checkpoint_wrapper=lambda m, **k: m)

    class Metric(nn.Module):
        def __init__(self, dist_sync_on_step=False):
            super().__init__()

        def add_state(self, name, default, dist_reduce_fx=None):
            self.register_buffer(name, default.clone())

        def forward(self, *a, **k):
            self.update(*a, **k)
            return self.compute()

    _mod("torchmetrics", Metric=Metric)

    # the dVAE tokenizer is only reached for mim losses (never enabled here)
    _mod("vilt_dvae_stub")
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)


def import_reference():
    install_stubs()
    import vilt.modules.vilt_module as vm  # noqa
    import vilt.modules.vision_transformer as vit  # noqa
    import vilt.modules.objectives as obj  # noqa
    return vm, vit, obj


def base_config(**over):
    """The reference's @ex.config defaults (src/vilt/config.py:25-168) as a plain dict."""
    loss_names = {"itm": 0, "ifm": 0, "mlm": 0, "vqa": 0, "nlvr2": 0, "irtr": 0, "mim": 0,
                  "image_only_mim": 0, "text_only_mlm": 0, "img_cls": 0, "mnc": 0, "mld": 0}
    cfg = dict(
        exp_name="vlmo", seed=1, loss_names=loss_names, batch_size=1024, image_size=224, max_image_len=-1,
        patch_size=16, draw_false_image=0, image_only=False, img_cls_label_size=1000, vqav2_label_size=3129,
        max_text_len=40, max_text_len_of_initckpt=196, tokenizer="bert-base-uncased", vocab_size=30522,
        whole_word_masking=False, mlm_prob=0.15, draw_false_text=0, vl_mlm_weight=1, ifm_weight=1, num_frames=1,
        max_vl_text_len=None, use_temporal_roll_module=False, vl_mlm_prob=0.15, vit="vit_base_patch16_224",
        hidden_size=768, num_heads=12, num_layers=12, mlp_ratio=4, drop_rate=0.1, vlffn_start_layer_index=10,
        optim_type="adamw", beta_2=0.98, learning_rate=1e-4, weight_decay=0.01, weight_decay_custom_modules=0.01,
        decay_power=1, max_epoch=100, max_steps=200000, warmup_steps=2500, end_lr=0, lr_mult=1, use_cpu=True,
        all_mlp_mult=False, all_vl_mult=False, all_v_mult=False, all_l_mult=False, get_recall_metric=False,
        test_only=False, validation_only=False, load_path="", precision=32, use_beit_weight=False,
        use_self_weight=False, use_ufo=False, separate_inference=True, use_moe=False,
        self_attn_for_single_mode=False, use_vision_weights_for_other_modalities=False, in_attn=False, in_ffn=True,
        merge_weights=False, merge_ratio=0.5, sum_task_vectors=False, central_weight=None, sum_lambda=1,
        only_activate_used_experts=False, regmean=False, gram_matrices=None, scaling_for_non_diag=1,
        use_custom_ln_attn=False, use_custom_ln_ffn=False, discrete_vae_weight_path="", tasks=None,
        random_initialization=True, log_dir="result", per_gpu_batchsize=2, num_gpus=0, num_nodes=1,
    )
    for k, v in over.items():
        if k == "loss_names":
            cfg["loss_names"] = dict(loss_names, **v)
        else:
            cfg[k] = v
    return cfg


def build_reference_model(cfg, arch):
    """arch in {"ufo", "all_moe"}; mirrors run.py:165-185."""
    vm, vit, obj = import_reference()
    from vilt.ufo.config import UFOConfig
    from vilt.moe.config import MOEConfig
    from vilt.custom_ln.config import LNConfig
    cfg = dict(cfg)
    ufo = ln = moe = None
    if arch == "ufo":
        cfg.update(use_ufo=True, separate_inference=True)
        ufo = UFOConfig()
        ufo.separate_inference = True
    elif arch == "all_moe":
        cfg.update(use_moe=True, in_attn=True, in_ffn=True, use_custom_ln_ffn=True, use_custom_ln_attn=True,
                   self_attn_for_single_mode=True)
        ln = LNConfig()
        ln.use_custom_ln_attn = True
        ln.use_custom_ln_ffn = True
        moe = MOEConfig()
        moe.in_attn = True
        moe.in_ffn = True
        moe.self_attn_for_single_mode = True
        moe.separate_inference = cfg["separate_inference"]
    else:
        raise ValueError(arch)
    model = vm.ViLTransformerSS(cfg, ufo, ln, moe)
    return model, cfg


def inject_train_masks(model):
    """Make a train-mode step of the reference deterministic: every DropPath call and the text-embedding dropout use
    oracle.detweights.det_keep / det_dropout_mask, keyed by (pass tag, site or sample).  Pass tags follow the order
    in which forward() runs its passes for mlm + ifm + itm (vilt_module.py:1493-1510, objectives.py:146-245):
    infer -> mlm, pos, negimg, negtxt; infer_image -> img; infer_text -> txt.  Only the wrapping is added here; the
    reference's arithmetic (x * mask / keep) is untouched."""
    from oracle.detweights import det_keep, det_dropout_mask
    state = {"tag": None, "site": 0, "n_infer": 0}

    def keep(B, keep_prob):
        k = det_keep(state["tag"], state["site"], B, keep_prob)
        state["site"] += 1
        return torch.from_numpy(k)

    INJECT["keep"] = keep

    def wrap(name, tagger):
        fn = getattr(model, name)

        def wrapped(*a, **k):
            state["tag"], state["site"] = tagger(), 0
            return fn(*a, **k)

        setattr(model, name, wrapped)

    def infer_tag():
        t = ("mlm", "pos", "negimg", "negtxt")[state["n_infer"] % 4]
        state["n_infer"] += 1
        return t

    wrap("infer", infer_tag)
    wrap("infer_image", lambda: "img")
    wrap("infer_text", lambda: "txt")
    p = model.text_embeddings.dropout.p

    class DetDropout(nn.Module):
        def forward(self, x):
            if not self.training or p == 0.0:
                return x
            B, T, D = x.shape
            m = torch.stack([torch.from_numpy(det_dropout_mask(state["tag"], b, T, D, 1.0 - p)) for b in range(B)])
            return x * m / (1.0 - p)

    model.text_embeddings.dropout = DetDropout()
    return state
