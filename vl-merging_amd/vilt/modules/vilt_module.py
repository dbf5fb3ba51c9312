"""ViLTransformerSS: the reference's LightningModule surface (src/vilt/modules/vilt_module.py) on the MI355X engine.

Same constructor signature, parameter / buffer names and shapes (reference checkpoints load with strict=False exactly
as at vilt_module.py:293), same `infer*` / `forward` / `training_step` / merge method contracts.  pytorch_lightning is
not required: the class is a plain nn.Module exposing the hooks the reference's Trainer calls.
"""
import math
import types

import numpy as np
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import _lib as L
from ... import engine
from ... import merge as merge_ops
from ... import ops
from ... import regmean as regmean_ops
from . import heads, objectives, vilt_utils
from . import vision_transformer as vit


class BertEmbeddings(nn.Module):
    """Parameter layout of HF BertEmbeddings; forward = transformers-4.x semantics for
    position_embedding_type="rel_pos" (vilt_module.py:51-63): word + bert-token-type(0) -> LayerNorm(1e-12) ->
    dropout.  position_embeddings / position_ids are carried for checkpoint compatibility only."""

    def __init__(self, vocab_size, hidden_size, max_position_embeddings, dropout, eps=1e-12):
        super().__init__()
        self.word_embeddings = nn.Embedding(vocab_size, hidden_size, padding_idx=0)
        self.position_embeddings = nn.Embedding(max_position_embeddings, hidden_size)
        self.token_type_embeddings = nn.Embedding(2, hidden_size)
        self.LayerNorm = nn.LayerNorm(hidden_size, eps=eps)
        self.dropout = nn.Dropout(dropout)
        self.register_buffer("position_ids", torch.arange(max_position_embeddings).expand((1, -1)).clone())

    def forward(self, input_ids):
        we = self.word_embeddings
        emb = engine.embedding(input_ids, we.weight, we.padding_idx) + self.token_type_embeddings.weight[0]
        emb = engine.layer_norm(emb, self.LayerNorm.weight, self.LayerNorm.bias, self.LayerNorm.eps, out_f32=True)
        return self.dropout(emb)


def build_relative_position_indices(window, max_text_len, max_text_len_of_initckpt, max_vl_text_len=None):
    """The integer index tables of vilt_module.py:123-206, bit-exact (int64 / float32 dtypes as in the reference)."""
    gh, gw = window
    num_rel = (2 * gh - 1) * (2 * gw - 1) + 3
    text_num_rel = 2 * max_text_len_of_initckpt
    all_num_rel = num_rel + text_num_rel + 2
    coords = torch.stack(torch.meshgrid([torch.arange(gh), torch.arange(gw)], indexing="ij"))
    flat = torch.flatten(coords, 1)
    rel = (flat[:, :, None] - flat[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += gh - 1
    rel[:, :, 1] += gw - 1
    rel[:, :, 0] *= 2 * gw - 1
    n_img = gh * gw + 1
    rpi = torch.zeros((n_img, n_img), dtype=rel.dtype)
    rpi[1:, 1:] = rel.sum(-1)
    rpi[0, 0:] = num_rel - 3
    rpi[0:, 0] = num_rel - 2
    rpi[0, 0] = num_rel - 1
    ids = torch.arange(max_text_len - 1)
    trel = ids.unsqueeze(-2) - ids.unsqueeze(-1)
    trel = trel - int(2 - max_text_len_of_initckpt) + (num_rel + 2)
    tpi = torch.zeros((max_text_len, max_text_len), dtype=rel.dtype)
    tpi[1:, 1:] = trel
    tpi[0, 0:] = all_num_rel - 3
    tpi[0:, 0] = all_num_rel - 2
    tpi[0, 0] = all_num_rel - 1
    t2i = torch.ones(max_text_len, n_img) * num_rel
    i2t = torch.ones(n_img, max_text_len) * (num_rel + 1)
    joint = torch.cat((torch.cat((tpi, t2i), 1), torch.cat((i2t, rpi), 1)), 0)
    out = {"relative_position_index": rpi, "text_relative_position_index": tpi,
           "text_imag_relative_position_index": joint}
    if max_vl_text_len is not None:
        v = max_vl_text_len
        out["vl_text_imag_relative_position_index"] = torch.cat(
            (torch.cat((tpi[:v, :v], t2i[:v]), 1), torch.cat((i2t[:, :v], rpi), 1)), 0)
    return out, num_rel, text_num_rel, all_num_rel


def _index16(index, n0):
    """int16 index in kernel coordinates: text positions [0,n0), image positions start at pos1 = roundup(n0,8);
    leading dimension padded to a multiple of 4; values are 4 x index (byte offsets into the fp32 table column, see
    include/vlm_hip.h).  Returns (index, transpose)."""
    index = index.long() * 4
    if int(index.max()) > 32767:
        raise L.VlmError("relative-position table too large for int16 byte offsets (R <= 8191)")
    n = index.shape[0]
    n1 = n - n0
    pos1 = (n0 + 7) // 8 * 8
    NP = pos1 + n1
    ld = (NP + 3) // 4 * 4
    pos = torch.cat([torch.arange(n0), pos1 + torch.arange(n1)]).to(index.device)
    m = torch.zeros(NP, ld, dtype=torch.int16, device=index.device)
    mt = torch.zeros(NP, ld, dtype=torch.int16, device=index.device)
    m[pos[:, None], pos[None, :]] = index.to(torch.int16)
    mt[pos[:, None], pos[None, :]] = index.t().to(torch.int16)
    return m.contiguous(), mt.contiguous()


class ViLTransformerSS(nn.Module):
    def __init__(self, config, ufo_config=None, ln_config=None, moe_config=None):
        super().__init__()
        import copy
        self.hparams = types.SimpleNamespace(config=copy.deepcopy(config), ufo_config=ufo_config, ln_config=ln_config,
                                             moe_config=moe_config)
        config = self.hparams.config
        hs = config["hidden_size"]
        self.text_embeddings = BertEmbeddings(config["vocab_size"], hs, config["max_text_len"], config["drop_rate"])
        self.text_embeddings.apply(objectives.init_weights)
        self.token_type_embeddings = nn.Embedding(2, hs)
        self.token_type_embeddings.apply(objectives.init_weights)
        self.moe_config = moe_config
        self.transformer = vit.create_vit(config["vit"], config=config, ufo_config=ufo_config, ln_config=ln_config,
                                          moe_config=moe_config)
        self.pooler = heads.Pooler(hs)
        self.pooler.apply(objectives.init_weights)
        ln_ = config["loss_names"]
        if ln_["mlm"] > 0 or ln_["text_only_mlm"] > 0:
            self.mlm_score = heads.MLMHead(hs, config["vocab_size"])
            self.mlm_score.apply(objectives.init_weights)
        if ln_["itm"] > 0:
            self.itm_score = heads.ITMHead(hs)
            self.itm_score.apply(objectives.init_weights)
        if ln_["ifm"] > 0:
            for nm in ("ifm_text_proj", "ifm_image_proj", "ifm_vl_text_proj", "ifm_vl_image_proj"):
                setattr(self, nm, heads.IFMHead(hs))
                getattr(self, nm).apply(objectives.init_weights)
            self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
            self.logit_vl_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
        if ln_["irtr"] > 0:
            for nm in ("ifm_text_proj", "ifm_image_proj"):
                setattr(self, nm, heads.IFMHead(hs))
                getattr(self, nm).apply(objectives.init_weights)
            self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
        if ln_.get("vqa", 0) > 0:  # vilt_module.py:300-308
            self.vqa_classifier = heads.MLPClassifier(hs, hs * 2, config["vqav2_label_size"])
            self.vqa_classifier.apply(objectives.init_weights)
        if ln_.get("nlvr2", 0) > 0:  # vilt_module.py:323-337: pair classifier + a third token-type row (second image)
            self.nlvr2_classifier = heads.MLPClassifier(hs * 2, hs * 2, 2)
            self.nlvr2_classifier.apply(objectives.init_weights)
            emb_data = self.token_type_embeddings.weight.data
            self.token_type_embeddings = nn.Embedding(3, hs)
            self.token_type_embeddings.apply(objectives.init_weights)
            self.token_type_embeddings.weight.data[0, :] = emb_data[0, :]
            self.token_type_embeddings.weight.data[1, :] = emb_data[1, :]
            self.token_type_embeddings.weight.data[2, :] = emb_data[1, :]
        for k in ("mim", "image_only_mim", "img_cls"):
            if ln_.get(k, 0) > 0:
                raise NotImplementedError("loss %r is outside the MI355X hot path (SURVEY.md 2.1 #3/#13)" % k)

        g = int(config["image_size"] / config["patch_size"])
        self.window_size = (g, g)
        self.max_text_len = config["max_text_len"]
        self.max_vl_text_len = config["max_vl_text_len"]
        self.max_imag_len = g * g + 1
        idx, self.num_relative_distance, self.text_num_relative_distance, self.all_num_relative_distance = \
            build_relative_position_indices(self.window_size, self.max_text_len, config["max_text_len_of_initckpt"],
                                            self.max_vl_text_len)
        self.relative_position_bias_table = nn.Parameter(
            torch.zeros(self.all_num_relative_distance, config["num_heads"] * config["num_layers"]))
        for k, v in idx.items():
            self.register_buffer(k, v)
        # "temporal" leftovers the reference keeps because its checkpoints contain them (vilt_module.py:189-265)
        self.num_frames = config["num_frames"]
        if self.num_frames >= 1:
            nf, ni, T = self.num_frames, self.max_imag_len, self.max_text_len
            t2i = torch.ones(T, ni * nf) * self.num_relative_distance
            i2t = torch.ones(ni * nf, T) * (self.num_relative_distance + 1)
            video = self.relative_position_index.repeat(nf, nf)
            self.register_buffer("video_relative_position_index", video)
            self.register_buffer("text_video_relative_position_index", torch.cat(
                (torch.cat((self.text_relative_position_index, t2i), 1), torch.cat((i2t, video), 1)), 0))
            self.temporal_relative_position_bias_table = nn.Parameter(
                torch.zeros(2 * nf, config["num_heads"] * config["num_layers"]))
            tid = torch.arange(nf)
            trel = tid.unsqueeze(-2) - tid.unsqueeze(-1)
            self.register_buffer("temporal_relative_position_index", (trel - trel.min()).repeat(ni, ni))
            m = torch.eye(nf).repeat_interleave(ni, dim=1).repeat_interleave(ni, dim=0).unsqueeze(0)
            self.register_buffer("mask_for_combining_temporal", m)
            if self.max_vl_text_len is not None:
                v = self.max_vl_text_len
                self.register_buffer("vl_text_video_relative_position_index", torch.cat(
                    (torch.cat((self.text_relative_position_index[:v, :v], t2i[:v]), 1),
                     torch.cat((i2t[:, :v], video), 1)), 0))

        self.vlffn_start_layer_index = config["vlffn_start_layer_index"]
        self.num_layers = config["num_layers"]
        self.current_tasks = []
        self.fuse_joint_passes = True  # engine option, not a reference config key
        # engine option: image-only + text-only pass of a batch as one block-diagonal pass (infer_unimodal_pair)
        self.fuse_unimodal_passes = os.environ.get("VLM_FUSE_UNIMODAL", "1") != "0"
        self._flat = None
        self._idx_cache = {}
        self._ones_cache = {}
        self._grad_hook = None
        self._gram = None
        self.trainer = None

        # ---- checkpoint load / merge (vilt_module.py:270-295 and :345-364) -------------------------------------------
        if config["load_path"] != "":
            from ... import checkpoint
            ckpt = checkpoint.load_file(config["load_path"])
            eval_only = config["test_only"] or config["validation_only"]
            if config["use_beit_weight"]:
                state_dict = self.modify_checkpoint_beit(ckpt)
            elif config["use_self_weight"]:
                state_dict = self.modify_checkpoint_self(ckpt)
            else:
                state_dict = ckpt["state_dict"] if eval_only else self.modify_checkpoint_vlmo(ckpt)
            if config["merge_weights"]:
                state_dict = self.merge_weights(state_dict)
            elif config["sum_task_vectors"]:
                state_dict = self.sum_task_vectors(state_dict)
            elif config["regmean"] and not eval_only:
                state_dict = self.regmean(state_dict)
            self.load_info = self.load_state_dict(state_dict, strict=False)

    # ---- engine plumbing -----------------------------------------------------------------------------------------------
    @property
    def device(self):
        return self.relative_position_bias_table.device

    def setup_engine(self, grad_hook=None):
        """Flatten parameters into the engine's fp32 / grad / bf16 buffers (GPU only) and build the bf16 shadows."""
        if self.device.type != "cuda":
            raise L.VlmError("ViLTransformerSS runs on the GPU only: call .cuda() before setup_engine()")
        L.get_lib()
        self._flat = engine.FlatParams(self, order_key=vilt_utils.flat_order_key)
        if os.environ.get("VLM_TRANSPOSED_SHADOWS", "1") != "0":  # A/B switch for measurements
            self._flat.enable_transposed(lambda n: n.startswith("transformer.blocks.") and n.endswith(".weight"))
        if os.environ.get("VLM_FOLD_LAYERSCALE", "1") != "0":  # A/B switch for measurements
            # gamma_1 / gamma_2 folded into attn.proj / mlp.fc2 of every expert (vision_transformer.py:489-491, :586, :603)
            entries = []
            for blk in self.transformer.blocks:
                for mod, gamma in ((blk.attn, blk.gamma_1), (blk.mlp, blk.gamma_2)):
                    for m in (mod.values() if isinstance(mod, nn.ModuleDict) else [mod]):
                        lin = m.proj if hasattr(m, "proj") else m.fc2
                        entries.append((blk.layer_number, lin.weight, lin.bias, gamma))
            self._flat.enable_layerscale_fold(entries)
        self._flat.refresh_shadow()
        self._grad_hook = grad_hook
        return self._flat

    def _ensure_engine(self):
        if self._flat is None or self._flat.flat_p.device != self.device:
            self.setup_engine(self._grad_hook)
        elif self._flat.dirty:
            self._flat.refresh_shadow()

    def load_state_dict(self, state_dict, strict=True):
        sd = {k: (v.float() if torch.is_tensor(v) and v.dtype == torch.float64 else v) for k, v in state_dict.items()}
        res = super().load_state_dict(sd, strict=strict)
        if self._flat is not None:
            self._flat.dirty = True
        return res

    def zero_grad(self, set_to_none=False):
        if self._flat is not None:
            self._flat.zero_grad()
        else:
            super().zero_grad(set_to_none=set_to_none)

    def mark_weights_changed(self):
        if self._flat is not None:
            self._flat.dirty = True

    # ---- merging: same method contracts as the reference -----------------------------------------------------------------
    def merge_weights(self, state_dict):
        return merge_ops.merge_weights(state_dict, self.hparams.config, device=self._merge_device())

    def sum_task_vectors(self, state_dict):
        return merge_ops.sum_task_vectors(state_dict, self.hparams.config, device=self._merge_device())

    def regmean(self, state_dict):
        return regmean_ops.regmean(state_dict, self.hparams.config, device=self._merge_device())

    def _merge_device(self):
        return self.device if self.device.type == "cuda" else torch.device("cuda", torch.cuda.current_device())

    def modify_checkpoint_vlmo(self, ckpt):
        """vilt_module.py:749-806: text position truncation, index-buffer pruning, bicubic resize of the image part
        of the relative-position table (27x27 -> 47x47 for 224 -> 384)."""
        state_dict = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
        key = "text_embeddings.position_embeddings.weight"
        if state_dict[key].size(0) != self.max_text_len:
            state_dict[key] = state_dict[key][: self.max_text_len, :]
            if "text_embeddings.position_ids" in state_dict:
                state_dict["text_embeddings.position_ids"] = state_dict["text_embeddings.position_ids"][:, : self.max_text_len]
            for k in ("relative_position_index", "text_relative_position_index", "text_imag_relative_position_index"):
                state_dict.pop(k)
        rel = state_dict["relative_position_bias_table"]
        src_num_pos = rel.size(0)
        dst_num_pos = self.relative_position_bias_table.size(0)
        non_image = self.text_num_relative_distance + 2 + 3
        src_size = int((src_num_pos - non_image) ** 0.5)
        dst_size = int((dst_num_pos - non_image) ** 0.5)
        for k in ("relative_position_index", "text_relative_position_index", "text_imag_relative_position_index",
                  "video_relative_position_index", "text_video_relative_position_index",
                  "temporal_relative_position_index", "mask_for_combining_temporal"):
            state_dict.pop(k, None)
        if src_size != dst_size:
            extra = rel[-non_image:, :]
            body = rel[:-non_image, :]
            embed = body.transpose(0, 1).reshape(-1, src_size, src_size)
            embed = F.interpolate(embed.unsqueeze(0), size=(dst_size, dst_size), mode="bicubic")
            embed = embed.squeeze(0).permute((1, 2, 0)).contiguous().view(-1, embed.size(1))
            state_dict["relative_position_bias_table"] = torch.cat((embed, extra), dim=0)
        return state_dict

    def modify_checkpoint_beit(self, ckpt):
        """vilt_module.py:808-972 (BEiT checkpoints -> this model's keys), see vl_merging_amd/checkpoint.py."""
        from ... import checkpoint
        return checkpoint.modify_checkpoint_beit(self, ckpt)

    def modify_checkpoint_self(self, ckpt):
        """vilt_module.py:974-1058."""
        from ... import checkpoint
        return checkpoint.modify_checkpoint_self(self, ckpt)

    # ---- relative position bias --------------------------------------------------------------------------------------------
    def get_rel_pos_bias(self, relative_position_index, n_text=None):
        """Reference :1061-1064 returns a dense [H*L, N, N] tensor; here: an engine.RelPos handle for the attention
        kernel (the table's transpose + int16 index).  n_text = number of leading text positions of the index."""
        name = None
        for k in ("relative_position_index", "text_relative_position_index", "text_imag_relative_position_index",
                  "vl_text_imag_relative_position_index"):
            if hasattr(self, k) and getattr(self, k) is relative_position_index:
                name = k
        n = relative_position_index.shape[0]
        if n_text is None:
            n_text = {"relative_position_index": 0, "text_relative_position_index": n}.get(name, self.max_text_len)
        key = (name, n, n_text, str(relative_position_index.device))
        if name is None or key not in self._idx_cache:
            pair = _index16(relative_position_index, n_text)
            if name is None:
                return engine.make_relpos(self.relative_position_bias_table, *pair)
            self._idx_cache[key] = pair
        return engine.make_relpos(self.relative_position_bias_table, *self._idx_cache[key], index_tag=key)

    # ---- embeddings ----------------------------------------------------------------------------------------------------
    def _text_rows(self, text_ids, text_masks):
        e = self.text_embeddings(text_ids)
        # token_type_embeddings(zeros_like(text_masks)) of the reference (:1111-1113) = row 0 for every token: as a broadcast
        # add its backward is one column sum instead of a sorted scatter (embedding_dense_backward: 0.26 ms per step)
        e = e + self.token_type_embeddings.weight[0]
        return e.reshape(-1, e.shape[-1])

    def _text_spec(self, text_ids):
        """engine.TextSpec for the fused front end (the text rows made in place by ONE launch inside engine.pass_rows), or None
        when the stock path has to run: a replaced dropout module, parameters outside the flat buffers, a width the kernel
        does not take.  Train-mode dropout: a uniform draw per element, kept where u >= p (nn.Dropout's Bernoulli(1 - p) keep,
        scale 1 / (1 - p)); `text_embeddings.dropout_source(B, T, D) -> keep mask` injects the draw (parity tests)."""
        te = self.text_embeddings
        w = te.word_embeddings.weight
        D = w.shape[1]
        if (type(te.dropout) is not nn.Dropout or getattr(w, "_vlm_flat", None) is None or not w.is_cuda or D % 4 or D > 1024
                or text_ids.dtype != torch.int64 or not text_ids.is_cuda):
            return None
        u, p, scale = None, 0.0, 1.0
        if self.training and te.dropout.p > 0.0:
            n = text_ids.numel()
            scale = 1.0 / (1.0 - te.dropout.p)
            src = getattr(te, "dropout_source", None)
            if src is not None:
                u, p = src(text_ids.shape[0], text_ids.shape[1], D).to(device=w.device, dtype=torch.float32).reshape(n, D).contiguous(), 0.5
            else:
                u, p = torch.rand(n, D, device=w.device, dtype=torch.float32), te.dropout.p
        ln = te.LayerNorm
        return engine.TextSpec(text_ids.contiguous(), w, te.word_embeddings.padding_idx, te.token_type_embeddings.weight, ln.weight, ln.bias,
                               ln.eps, u, p, scale)

    def _text_rows_any(self, text_ids, text_masks, spec):
        """The text rows as a tensor (paths that concatenate them themselves, text-only passes)."""
        if spec is None:
            return self._text_rows(text_ids, text_masks)
        tr = self.transformer
        pe = tr.patch_embed
        return engine.pass_rows(None, None, pe.proj.weight, pe.proj.bias, tr.cls_token, self.token_type_embeddings.weight, 0,
                                pe.patch_size[0], text=spec)

    def _image_rows(self, img, image_token_type_idx=1, mask_image=False, bool_masked_pos=None):
        x, x_mask, _, _ = self.transformer.visual_embed(img, max_image_len=self.hparams.config["max_image_len"],
                                                        mask_it=mask_image, bool_masked_pos=bool_masked_pos)
        x = x + self.token_type_embeddings.weight[image_token_type_idx]
        return x.reshape(-1, x.shape[-1]), x_mask, x.shape[1]

    def _pass_rows(self, trows, img, image_token_type_idx=1, text=None, mask_like=None):
        """x = [text rows ; image rows] of a pass with the image side of visual_embed fused into the patch-embed GEMM and, given a
        TextSpec, the text side made in place (engine.pass_rows).  -> (x, image mask of ones [B, I] in mask_like's dtype, I)."""
        tr = self.transformer
        pe = tr.patch_embed
        x = engine.pass_rows(trows, img, pe.proj.weight, pe.proj.bias, tr.cls_token, self.token_type_embeddings.weight,
                             image_token_type_idx, pe.patch_size[0], text=text)
        I = 1 + pe.num_patches if img.shape[-1] == pe.img_size[1] and img.shape[-2] == pe.img_size[0] \
            else 1 + (img.shape[-2] // pe.patch_size[0]) * (img.shape[-1] // pe.patch_size[1])
        dt = mask_like.dtype if mask_like is not None else torch.float32
        key = (img.shape[0], I, dt, str(img.device))
        ones = self._ones_cache.get(key)
        if ones is None:  # a constant: one fill per geometry instead of one (plus a cast) per pass
            if len(self._ones_cache) > 16:
                self._ones_cache.clear()
            ones = self._ones_cache[key] = torch.ones(img.shape[0], I, device=img.device, dtype=dt)
        return x, ones, I

    def _final_norm(self, x):
        n = self.transformer.norm
        return engine.layer_norm(x, n.weight, n.bias, n.eps)

    def _hook(self):
        return self._grad_hook

    # ---- Gram cache (src/cache_gram_matrices.py) ---------------------------------------------------------------
    def start_gram_capture(self):
        """Record G += X^T X (float64, on device) for the input X of every linear the reference hooks."""
        self._gram = engine.GramCapture()
        return self._gram

    def stop_gram_capture(self):
        g, self._gram = self._gram, None
        return g

    def _pass_ctx(self, *a, **k):
        pc = engine.PassCtx(*a, **k)
        pc.gram = getattr(self, "_gram", None)
        pc.uniform_source = getattr(self, "droppath_uniform_source", None)
        return pc

    def _drop_sites(self, with_vlffn):
        """DropPath probabilities of a pass in call order: two sites per block evaluation, the vlffn branch re-runs the
        last layers."""
        p = [b.drop_path_prob for b in self.transformer.blocks]
        sites = [q for q in p for _ in (0, 1)]
        if with_vlffn:
            sites += [q for q in p[self.vlffn_start_layer_index:] for _ in (0, 1)]
        return sites

    # ---- passes ----------------------------------------------------------------------------------------------------------
    def infer(self, batch, mask_text=False, mask_image=False, bool_masked_pos=None, image_token_type_idx=1,
              image_embeds=None, image_masks=None):
        """Joint text+image pass, reference :1071-1156."""
        self._ensure_engine()
        imgkey = f"image_{image_token_type_idx - 1}" if f"image_{image_token_type_idx - 1}" in batch else "image"
        do_mlm = "_mlm" if mask_text else ""
        text_ids = batch[f"text_ids{do_mlm}"]
        text_labels = batch[f"text_labels{do_mlm}"]
        text_masks = batch["text_masks"]
        B, T = text_ids.shape
        spec = self._text_spec(text_ids)
        trows = None if spec is not None else self._text_rows(text_ids, text_masks)
        keep1 = None
        if (image_embeds is not None or image_masks is not None or mask_image) and trows is None:
            trows = self._text_rows_any(text_ids, text_masks, spec)
        if image_embeds is not None or image_masks is not None:
            # precomputed visual_embed output (reference :1092-1108; the reference itself then fails at its result dict,
            # `"image": img` with img unbound -- here the entry is None).  Both must be given, as there (image_masks.type_as).
            if image_embeds is None or image_masks is None:
                raise ValueError("infer: image_embeds and image_masks go together")
            img = None
            xi = image_embeds + self.token_type_embeddings.weight[image_token_type_idx]
            irows, I = xi.reshape(-1, xi.shape[-1]), xi.shape[1]
            keep1 = image_masks.to(torch.uint8).contiguous()
            x = torch.cat([trows, irows], 0)
        elif mask_image:
            img = batch[imgkey][0]
            irows, image_masks, I = self._image_rows(img, image_token_type_idx, mask_image, bool_masked_pos)
            x = torch.cat([trows, irows], 0)
        else:
            img = batch[imgkey][0]
            x, image_masks, I = self._pass_rows(trows, img, image_token_type_idx, spec, text_masks)
        image_masks = image_masks.type_as(text_masks)
        index = self.vl_text_imag_relative_position_index if self.max_vl_text_len is not None \
            else self.text_imag_relative_position_index
        pc = self._pass_ctx(ops.Seq(B, T, I), self.hparams.config["num_heads"], self.get_rel_pos_bias(index, T),
                            keep0=text_masks.to(torch.uint8).contiguous(), keep1=keep1)
        pc.plan_drop_path(self._drop_sites(False))
        for blk in self.transformer.blocks:
            x = blk.run(x, pc, 2, self._hook())
        x = self._final_norm(x)
        text_feats, image_feats, text_cls, _ = engine.feature_views(x, B, T, I)
        cls_feats = self.pooler(text_feats, cls_rows=text_cls)
        return {"text_feats": text_feats, "image_feats": image_feats, "cls_feats": cls_feats,
                "raw_cls_feats": text_cls, "image_labels": None, "image_masks": image_masks, "image": img,
                "text_labels": text_labels, "text_ids": text_ids, "text_masks": text_masks, "patch_index": None}

    def _unimodal(self, x, pc, type_id, with_vlffn):
        pc.plan_drop_path(self._drop_sites(with_vlffn))
        hs = None
        for i, blk in enumerate(self.transformer.blocks):
            x = blk.run(x, pc, type_id, self._hook())
            if i == self.vlffn_start_layer_index - 1:
                hs = x
        v = None
        if with_vlffn:
            v = hs
            for i in range(self.vlffn_start_layer_index, self.num_layers):
                v = self.transformer.blocks[i].run(v, pc, 2, self._hook())
            v = self._final_norm(v)
        return self._final_norm(x), v

    @staticmethod
    def _l2(x):
        return engine.l2_normalize(x)

    def _infer_text(self, batch, mask_text, with_vlffn):
        self._ensure_engine()
        do_mlm = "_mlm" if mask_text else ""
        text_ids = batch[f"text_ids{do_mlm}"]
        text_labels = batch[f"text_labels{do_mlm}"]
        text_masks = batch["text_masks"]
        B, T = text_ids.shape
        x = self._text_rows_any(text_ids, text_masks, self._text_spec(text_ids))
        index = self.text_relative_position_index
        if self.max_vl_text_len is not None and T != index.shape[0]:
            index = index[:T, :T]
        pc = self._pass_ctx(ops.Seq(B, T, 0), self.hparams.config["num_heads"], self.get_rel_pos_bias(index, T),
                            keep0=text_masks.to(torch.uint8).contiguous())
        l, v = self._unimodal(x, pc, 1, with_vlffn)
        D = l.shape[-1]
        return self._text_result(l.view(B, T, D), v.view(B, T, D) if v is not None else None, text_labels, text_ids,
                                 text_masks)

    def infer_unimodal_pair(self, batch, with_vlffn=True, mask_text=False, mask_image=False, image_token_type_idx=1,
                            bool_masked_pos=None):
        """(infer_image*, infer_text*) of one batch as ONE pass: text rows and image rows share every launch, block-
        diagonal attention (SEPARATE mode on the joint index, whose diagonal blocks ARE the unimodal indices) keeps
        them apart.  Each row computes exactly what its own unimodal pass computes (reference :1159-1464), the 880-row
        text pass no longer costs ~150 tiny launches per step.  Returns (image result dict, text result dict)."""
        self._ensure_engine()
        do_mlm = "_mlm" if mask_text else ""
        text_ids, text_labels, text_masks = batch[f"text_ids{do_mlm}"], batch[f"text_labels{do_mlm}"], batch["text_masks"]
        imgkey = f"image_{image_token_type_idx - 1}" if f"image_{image_token_type_idx - 1}" in batch else "image"
        img = batch[imgkey][0]
        B, T = text_ids.shape
        spec = self._text_spec(text_ids)
        if mask_image:
            trows = self._text_rows_any(text_ids, text_masks, spec)
            irows, image_masks, I = self._image_rows(img, image_token_type_idx, mask_image, bool_masked_pos)
            x = torch.cat([trows, irows], 0)
        else:
            trows = None if spec is not None else self._text_rows(text_ids, text_masks)
            x, image_masks, I = self._pass_rows(trows, img, image_token_type_idx, spec, text_masks)
        image_masks = image_masks.type_as(text_masks)
        index = self.vl_text_imag_relative_position_index if self.max_vl_text_len is not None \
            else self.text_imag_relative_position_index
        pc = self._pass_ctx(ops.Seq(B, T, I), self.hparams.config["num_heads"], self.get_rel_pos_bias(index, T),
                            keep0=text_masks.to(torch.uint8).contiguous())
        pc.independent_segments = True
        pc.plan_drop_path(self._drop_sites(with_vlffn))
        hs = None
        for i, blk in enumerate(self.transformer.blocks):
            x = blk.run(x, pc, 3, self._hook())
            if i == self.vlffn_start_layer_index - 1:
                hs = x
        v = None
        if with_vlffn:
            v = hs
            for i in range(self.vlffn_start_layer_index, self.num_layers):
                v = self.transformer.blocks[i].run(v, pc, 4, self._hook())
            v = self._final_norm(v)
        x = self._final_norm(x)
        # the views the two result dicts expose, one autograd node per feature matrix (engine.feature_views)
        lt, li, lt_cls, li_cls = engine.feature_views(x, B, T, I)
        vt, vi, vt_cls, vi_cls = engine.feature_views(v, B, T, I) if v is not None else (None, None, None, None)
        text = self._text_result(lt, vt, text_labels, text_ids, text_masks, lt_cls, vt_cls)
        image = self._image_result(li, vi, image_masks, text_masks, li_cls, vi_cls)
        return image, text

    def _text_result(self, l, v, text_labels, text_ids, text_masks, l_cls=None, v_cls=None):
        l_cls = l[:, 0] if l_cls is None else l_cls
        v_cls = (v[:, 0] if v_cls is None else v_cls) if v is not None else None
        cls = self._l2(self.ifm_text_proj(l_cls)) if getattr(self, "ifm_text_proj", None) is not None else None
        cls_v = self._l2(self.ifm_vl_text_proj(v_cls)) if v is not None else None
        return {"text_feats": l, "image_feats": None, "cls_feats": cls, "cls_vlffn_feats": cls_v,
                "raw_cls_feats": l_cls, "image_labels": None, "image_masks": None, "text_labels": text_labels,
                "text_ids": text_ids, "text_masks": text_masks, "patch_index": None}

    def _image_result(self, vf, v, image_masks, text_masks, vf_cls=None, v_cls=None):
        vf_cls = vf[:, 0] if vf_cls is None else vf_cls
        v_cls = (v[:, 0] if v_cls is None else v_cls) if v is not None else None
        if getattr(self, "ifm_image_proj", None) is not None:
            cls = self._l2(self.ifm_image_proj(vf_cls))
        else:
            cls = self.pooler(vf, cls_rows=vf_cls)
        cls_v = self._l2(self.ifm_vl_image_proj(v_cls)) if v is not None else None
        return {"text_feats": None, "image_feats": vf, "cls_feats": cls, "cls_vlffn_feats": cls_v,
                "raw_cls_feats": vf_cls, "image_labels": None, "image_masks": image_masks, "text_labels": None,
                "text_ids": None, "text_masks": text_masks, "patch_index": None}

    def infer_text(self, batch, mask_text=False):  # :1159-1223
        return self._infer_text(batch, mask_text, True)

    def infer_text_ft(self, batch, mask_text=False):  # :1226-1285
        return self._infer_text(batch, mask_text, False)

    def _infer_image(self, batch, mask_image, image_token_type_idx, bool_masked_pos, with_vlffn):
        self._ensure_engine()
        imgkey = f"image_{image_token_type_idx - 1}" if f"image_{image_token_type_idx - 1}" in batch else "image"
        text_masks = batch["text_masks"]
        img = batch[imgkey][0]
        B = img.shape[0]
        if mask_image:
            x, image_masks, I = self._image_rows(img, image_token_type_idx, mask_image, bool_masked_pos)
        else:
            x, image_masks, I = self._pass_rows(None, img, image_token_type_idx, None, text_masks)
        image_masks = image_masks.type_as(text_masks)
        pc = self._pass_ctx(ops.Seq(B, 0, I), self.hparams.config["num_heads"],
                            self.get_rel_pos_bias(self.relative_position_index, 0))
        vf, v = self._unimodal(x, pc, 0, with_vlffn)
        D = vf.shape[-1]
        return self._image_result(vf.view(B, I, D), v.view(B, I, D) if v is not None else None, image_masks, text_masks)

    def infer_image(self, batch, mask_image=False, image_token_type_idx=1, image_embeds=None, image_masks=None,
                    bool_masked_pos=None):  # :1287-1375
        return self._infer_image(batch, mask_image, image_token_type_idx, bool_masked_pos, True)

    def infer_image_ft(self, batch, mask_image=False, image_token_type_idx=1, image_embeds=None, image_masks=None,
                       bool_masked_pos=None):  # :1378-1464
        return self._infer_image(batch, mask_image, image_token_type_idx, bool_masked_pos, False)

    # ---- task dispatch (vilt_module.py:1467-1530) ----------------------------------------------------------------------
    def forward(self, batch):
        ret = dict()
        if len(self.current_tasks) == 0:
            ret.update(self.infer(batch))
            return ret
        if self.hparams.config["tasks"] is not None:
            if "vl" in batch:
                batch = batch["vl"]
            else:
                return ret
        fuse = self.fuse_joint_passes and all(t in self.current_tasks for t in ("mlm", "itm", "ifm"))
        if fuse:
            objectives.prefetch_negative_candidates(batch)  # multi-rank: candidate all-gathers fly under the ifm pass
        if "mlm" in self.current_tasks and not fuse:
            ret.update(objectives.compute_mlm(self, batch))
        if "ifm" in self.current_tasks:
            ret.update(objectives.compute_ifm(self, batch))
        if "irtr" in self.current_tasks:
            ret.update(objectives.compute_irtr(self, batch))
        if "vqa" in self.current_tasks:
            ret.update(objectives.compute_vqa(self, batch))
        if "nlvr2" in self.current_tasks:
            ret.update(objectives.compute_nlvr2(self, batch))
        if fuse:
            # the four joint passes of mlm + itm as one 4B-sample pass (same losses, see compute_mlm_itm_fused)
            ret.update(objectives.compute_mlm_itm_fused(self, batch, ret["ifm_i2t_logits"], ret["ifm_t2i_logits"]))
        elif "itm" in self.current_tasks:
            ret.update(objectives.compute_itm_hardneg(self, batch, ret["ifm_i2t_logits"], ret["ifm_t2i_logits"]))
        return ret

    def training_step(self, batch, batch_idx=0):
        vilt_utils.set_task(self)
        output = self(batch)
        return engine.weighted_sum([v for k, v in output.items() if "loss" in k])

    def validation_step(self, batch, batch_idx=0):
        vilt_utils.set_task(self)
        return self(batch)

    def log(self, *a, **k):  # LightningModule.log: observability, not on the hot path
        pass

    def configure_optimizers(self):
        return vilt_utils.set_schedule(self)
