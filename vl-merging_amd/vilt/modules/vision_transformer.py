"""Backbone modules with the reference's names, parameter shapes and routing semantics
(reference src/vilt/modules/vision_transformer.py: Mlp :272, Attention :299, Block :366, PatchEmbed :694,
VisionTransformer :758-1008), executing on the HIP kernels through vl_merging_amd.engine.

The modules only HOLD parameters and describe routing; the arithmetic of a block evaluation is one fused
autograd function (engine._BlockFn).  Activations are [rows, D] matrices in the segment-major token layout.
"""
import math

import torch
import torch.nn as nn

from ... import _lib as L
from ... import engine
from ... import ops


def trunc_normal_(t, std=0.02):
    return nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2.0, b=2.0)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None):
        super().__init__()
        self.num_heads = num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        if dim // num_heads != 64:
            raise L.VlmError("the fused attention kernel is built for head_dim 64 (got %d)" % (dim // num_heads))
        self.qkv = nn.Linear(dim, dim * 3, bias=False)
        if qkv_bias:
            self.q_bias = nn.Parameter(torch.zeros(dim))
            self.v_bias = nn.Parameter(torch.zeros(dim))
        else:
            self.q_bias = None
            self.v_bias = None
        self.proj = nn.Linear(dim, dim)


class Block(nn.Module):
    """Parameter container + routing of one transformer layer (reference Block.__init__ :366-493)."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop_path=0.0,
                 layer_number=0, vlffn_start_layer_index=-1, max_text_len=40, ufo_config=None, ln_config=None,
                 moe_config=None, eps=1e-6):
        super().__init__()
        self.use_custom_ln = ln_config is not None
        self.use_ufo = ufo_config is not None
        self.use_moe = moe_config is not None
        self.separate_inference = False
        deep = layer_number >= vlffn_start_layer_index
        pick = lambda c: list(c.tasks_for_deep_layers if deep else c.tasks_for_shallow_layers)  # noqa: E731
        self.tasks = None
        # the reference mutates the shared config objects per layer (:394-413); we keep the per-layer value here
        if self.use_custom_ln:
            self.ln_tasks = pick(ln_config)
        if self.use_moe:
            self.moe_tasks = pick(moe_config)
            self.in_attn, self.in_ffn = moe_config.in_attn, moe_config.in_ffn
            self.self_attn_for_single_mode = moe_config.self_attn_for_single_mode
            self.separate_inference = getattr(moe_config, "separate_inference", False)
        if self.use_ufo:
            self.ufo_tasks = pick(ufo_config)
            self.separate_inference = ufo_config.separate_inference
        if self.use_custom_ln:
            self.tasks = self.ln_tasks
        elif self.use_moe:
            self.tasks = self.moe_tasks
        elif self.use_ufo:
            self.tasks = self.ufo_tasks

        def norm():
            return nn.LayerNorm(dim, eps=eps)

        if self.use_moe and self.in_attn:
            self.attn = nn.ModuleDict({t: Attention(dim, num_heads, qkv_bias, qk_scale) for t in self.tasks})
            self.norm1 = nn.ModuleDict({t: norm() for t in self.tasks})
        else:
            self.attn = Attention(dim, num_heads, qkv_bias, qk_scale)
            self.norm1 = norm()
        hidden = int(dim * mlp_ratio)
        if self.use_moe and self.in_ffn:
            self.mlp = nn.ModuleDict({t: Mlp(dim, hidden) for t in self.tasks})
        else:
            self.mlp = Mlp(dim, hidden)
        self.norm2 = norm()
        if self.use_custom_ln:
            if ln_config.use_custom_ln_attn:
                self.norm1 = nn.ModuleDict({t: norm() for t in self.ln_tasks})
            if ln_config.use_custom_ln_ffn:
                self.norm2 = nn.ModuleDict({t: norm() for t in self.ln_tasks})
        self.has_vl_moe = deep
        self.layer_number = layer_number
        self.drop_path_prob = float(drop_path)
        self.gamma_1 = nn.Parameter(0.1 * torch.ones(dim))
        self.gamma_2 = nn.Parameter(0.1 * torch.ones(dim))
        self.max_text_len = max_text_len
        self.eps = eps
        self.num_heads = num_heads
        self._plans = {}

    # ---- routing ------------------------------------------------------------------------------------------------
    def _expert(self, key):
        """ExpertWeights for modality key in {'v','l','vl'} following apply_ln / moe_forward's module selection."""
        def sel(mod):
            return mod[key] if isinstance(mod, nn.ModuleDict) else mod
        a, m, n1, n2 = sel(self.attn), sel(self.mlp), sel(self.norm1), sel(self.norm2)
        e = engine.ExpertWeights()
        e.n1w, e.n1b, e.n2w, e.n2b = n1.weight, n1.bias, n2.weight, n2.bias
        e.qkvw, e.qb, e.vb = a.qkv.weight, a.q_bias, a.v_bias
        e.projw, e.projb = a.proj.weight, a.proj.bias
        e.fc1w, e.fc1b, e.fc2w, e.fc2b = m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias
        # module names the reference's Gram hook keys on (cache_gram_matrices.py:264-281): with MoE modules the
        # Attention module itself (input of qkv), attn.<m>.proj, mlp.<m>.fc1, mlp.<m>.fc2; shared (ufo) modules
        # are hooked at mlp.fc1, mlp.fc2 and attn.proj only
        pre = "transformer.blocks.%d." % self.layer_number
        an = pre + "attn" + ("." + key if isinstance(self.attn, nn.ModuleDict) else "")
        mn = pre + "mlp" + ("." + key if isinstance(self.mlp, nn.ModuleDict) else "")
        e.gram_names = {"proj": an + ".proj", "fc1": mn + ".fc1", "fc2": mn + ".fc2"}
        if self.use_moe:
            e.gram_names["qkv"] = an
        return e

    def plan(self, type_id, seq):
        """Row ranges -> experts and the attention mode for (type_id, token layout).
        type_id: 0 image, 1 text, 2 vision-language (reference Block.forward :683-691)."""
        k = (type_id, seq.B, seq.n0, seq.n1, seq.base0, seq.base1)
        if k in self._plans:
            return self._plans[k]
        rows0 = (seq.base0, seq.base0 + seq.B * seq.n0)
        rows1 = (seq.base1, seq.base1 + seq.B * seq.n1)
        if seq.n0 and seq.n1 and rows0[1] != rows1[0]:
            raise L.VlmError("segment-major layout expects the image segment right after the text segment")
        lo = rows0[0] if seq.n0 else rows1[0]
        hi = rows1[1] if seq.n1 else rows0[1]
        any_dict = any(isinstance(m, nn.ModuleDict) for m in (self.attn, self.mlp, self.norm1, self.norm2))
        mode = L.ATTN_JOINT
        if type_id in (3, 4):
            # The image-only and the text-only pass of one batch run as ONE pass: block-diagonal attention keeps the two
            # modalities apart, every other op is per row, so each row sees exactly what its own unimodal pass computes
            # (type 3: modality experts l / v, reference type_id 1 / 0; type 4: the vl expert on both, the vlffn branch).
            mode = L.ATTN_SEPARATE
            if type_id == 4 or not any_dict:
                ranges = [(lo, hi, self._expert("vl" if type_id == 4 else "v"))]
            else:
                ranges = []
                if seq.n0:
                    ranges.append((rows0[0], rows0[1], self._expert("l")))
                if seq.n1:
                    ranges.append((rows1[0], rows1[1], self._expert("v")))
        elif type_id == 0:
            ranges = [(lo, hi, self._expert("v"))]
        elif type_id == 1:
            ranges = [(lo, hi, self._expert("l"))]
        elif self.tasks is not None and "vl" in self.tasks:
            ranges = [(lo, hi, self._expert("vl"))]
        elif not any_dict and not self.separate_inference:
            ranges = [(lo, hi, self._expert("vl"))]  # plain_forward: one shared expert, joint attention
        else:
            # shallow layer of a type-2 pass: text rows -> "l", image rows -> "v"
            if isinstance(self.attn, nn.ModuleDict):
                if not self.self_attn_for_single_mode:
                    raise NotImplementedError("in_attn MoE with self_attn_for_single_mode=False (both experts over "
                                              "all tokens, vision_transformer.py:641-651) is not on the hot path")
                mode = L.ATTN_SEPARATE
            else:
                mode = L.ATTN_SEPARATE if self.separate_inference else L.ATTN_JOINT
            if any_dict:
                ranges = []
                if seq.n0:
                    ranges.append((rows0[0], rows0[1], self._expert("l")))
                if seq.n1:
                    ranges.append((rows1[0], rows1[1], self._expert("v")))
            else:
                ranges = [(lo, hi, self._expert("v"))]
        p = engine.BlockPlan(ranges, mode, self.gamma_1, self.gamma_2, self.layer_number, self.drop_path_prob, self.eps)
        self._plans[k] = p
        return p

    def run(self, x, pc, type_id, hook=None):
        """x: fp32 [rows, D] (segment-major) -> fp32 [rows, D]."""
        return engine.run_block(x, self.plan(type_id, pc.seq), pc, self.training, hook)

    def forward(self, x, mask=None, type_id=None, relative_position_bias=None):
        """Reference-shaped entry (Block.forward(x[B,N,D], mask[B,N], type_id, relative_position_bias)) -> (x, None).
        `relative_position_bias` is an engine.RelPos handle (see ViLTransformerSS.get_rel_pos_bias), not a dense
        [H,N,N] tensor; tokens [0, max_text_len) are text when type_id == 2.  The attention matrix is not returned
        (every reference caller discards it)."""
        B, N, D = x.shape
        if type_id == 0:
            n0, n1 = 0, N
        elif type_id == 1:
            n0, n1 = N, 0
        else:
            n0, n1 = self.max_text_len, N - self.max_text_len
        seq = ops.Seq(B, n0, n1)
        rows = torch.cat([x[:, :n0].reshape(B * n0, D), x[:, n0:].reshape(B * n1, D)], 0).float()
        keep0 = keep1 = None
        if mask is not None:
            mk = mask.to(torch.uint8)
            keep0 = mk[:, :n0].contiguous() if n0 else None
            keep1 = mk[:, n0:].contiguous() if n1 else None
        pc = engine.PassCtx(seq, self.num_heads, relative_position_bias, keep0, keep1)
        y = self.run(rows, pc, 2 if type_id is None else type_id)
        out = torch.cat([y[:B * n0].view(B, n0, D), y[B * n0:].view(B, n1, D)], 1)
        return out, None


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.patch_shape = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.patch_shape[0] * self.patch_shape[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)  # parameter holder

    def forward(self, x):
        """-> [B, 1 + patches, D] fp32; row 0 of every image is a placeholder for the cls token."""
        return engine.patch_embed(x, self.proj.weight, self.proj.bias, self.patch_size[0])


class VisionTransformer(nn.Module):
    """reference VisionTransformer.__init__ :776-895 (rel-pos / abs-pos variants it never enables are dropped)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0,
                 qkv_bias=True, qk_scale=None, drop_path_rate=0.0, config=None, ufo_config=None, ln_config=None,
                 moe_config=None):
        super().__init__()
        drop_path_rate = drop_path_rate if config is None else config["drop_rate"]
        self.num_features = self.embed_dim = embed_dim
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        self.patch_size = patch_size
        self.patch_dim = img_size // patch_size
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.mask_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = nn.ModuleList([
            Block(embed_dim, num_heads, mlp_ratio, qkv_bias, qk_scale, drop_path=dpr[i], layer_number=i,
                  vlffn_start_layer_index=config["vlffn_start_layer_index"], max_text_len=config["max_text_len"],
                  ufo_config=ufo_config, ln_config=ln_config, moe_config=moe_config)
            for i in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        trunc_normal_(self.cls_token)
        trunc_normal_(self.mask_token)
        self.apply(self._init_weights)
        self._rescale()

    def _rescale(self):  # :897-903
        for i, block in enumerate(self.blocks):
            for n, p in block.named_parameters():
                if ("attn" in n and "proj" in n and "bias" not in n) or ("mlp" in n and "fc" in n and "bias" not in n):
                    p.data /= (2 * (i + 1)) ** 0.5

    @staticmethod
    def _init_weights(m):  # :905-912
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def visual_embed(self, _x, max_image_len=200, mask_it=False, bool_masked_pos=None):
        """:952-991 -> (x [B, 1+patches, D] fp32, x_mask [B, 1+patches] ones, None, None)."""
        x = self.patch_embed(_x)
        B = x.shape[0]
        if mask_it:
            w = bool_masked_pos.unsqueeze(-1).type_as(x)
            x = torch.cat([x[:, :1], x[:, 1:] * (1 - w) + self.mask_token.expand(B, x.shape[1] - 1, -1) * w], 1)
        x = torch.cat((self.cls_token.expand(B, -1, -1), x[:, 1:]), dim=1)
        x_mask = torch.ones(x.shape[0], x.shape[1], device=x.device)
        return x, x_mask, None, None


_VIT_ARCH = {
    "vit_base_patch16_224": dict(img_size=224, patch_size=16, embed_dim=768, depth=12, num_heads=12),
    "vit_base_patch16_384": dict(img_size=384, patch_size=16, embed_dim=768, depth=12, num_heads=12),
    "vit_tiny_patch16_224": dict(img_size=224, patch_size=16, embed_dim=192, depth=12, num_heads=3),
    "vit_tiny_patch16_384": dict(img_size=384, patch_size=16, embed_dim=192, depth=12, num_heads=3),
    "vit_large_patch16_224": dict(img_size=224, patch_size=16, embed_dim=1024, depth=24, num_heads=16),
    "vit_large_patch16_384": dict(img_size=384, patch_size=16, embed_dim=1024, depth=24, num_heads=16),
}


def create_vit(name, config=None, ufo_config=None, ln_config=None, moe_config=None):
    """The reference's factory functions (:1261-1372) for the variants its configs name."""
    if name not in _VIT_ARCH:
        raise KeyError("unsupported vit variant %r (hot path covers %s)" % (name, sorted(_VIT_ARCH)))
    kw = dict(_VIT_ARCH[name])
    if name == "vit_base_patch16_384" and config is not None:
        kw["mlp_ratio"] = config["mlp_ratio"]  # :1307
    return VisionTransformer(config=config, ufo_config=ufo_config, ln_config=ln_config, moe_config=moe_config, **kw)
