"""CPU restatement (plain fp32 PyTorch, functional over a state_dict) of the reference's transformer hot path.
TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.

Follows /root/reference/src/vilt/modules:
  vision_transformer.py  Attention.forward :329-363, Mlp.forward :290-296, Block.apply_ln :495-523,
                         plain_forward :525-530, separate_plain_forward :560-605, moe_forward :607-681,
                         PatchEmbed :714-728, visual_embed :952-991
  vilt_module.py         get_rel_pos_bias :1061-1064, infer :1071-1156, infer_text :1159-1223,
                         infer_image :1287-1375, *_ft :1226-1285/:1378-1464
  objectives.py          compute_mlm :88, compute_ifm :248, compute_itm_hardneg :146, compute_irtr :372
  heads.py               Pooler :8, ITMHead :21, IFMHead :30, MLMHead :40
BertEmbeddings uses the transformers-4.x semantics the reference was written for (no absolute position embedding
for position_embedding_type="rel_pos").  Dropout / DropPath are identities unless a TrainMasks object is
passed: then timm's DropPath (x * keep_b / keep_prob, two sites per block evaluation, vision_transformer.py:527-528)
and the text-embedding dropout (BertEmbeddings, p = drop_rate) use the INJECTED masks of oracle/detweights.py, the
same ones tests/golden/ref_harness.py::inject_train_masks feeds the reference (train-mode parity).

parity: PINNED against tests/golden/model_tiny_{ufo,all_moe}.npz and irtr_tiny_*.npz, model_base_*.npz (the benchmarked
width) and train_tiny_*.npz (train mode, injected masks): outputs of the reference itself on the same deterministic
weights and batch; tests/test_oracle_model.py.
"""
import torch
import torch.nn.functional as F


class Arch:
    def __init__(self, arch, hidden=768, heads=12, layers=12, vlffn_start=10, max_text_len=40, patch=16):
        assert arch in ("ufo", "all_moe")
        self.arch, self.D, self.H, self.L = arch, hidden, heads, layers
        self.S, self.T, self.P = vlffn_start, max_text_len, patch


class TrainMasks:
    """Train-mode randomness as data.  drop_path_rate / dropout p = config["drop_rate"]; per-layer DropPath
    probabilities = linspace(0, rate, L) (vision_transformer.py:855).  begin(tag) starts a pass."""

    def __init__(self, drop_rate, layers=12):
        from oracle.detweights import det_keep, det_dropout_mask
        self._keep, self._mask = det_keep, det_dropout_mask
        self.p = float(drop_rate)
        self.dpr = [float(x) for x in torch.linspace(0, drop_rate, layers)]
        self.tag, self.site = None, 0

    def begin(self, tag):
        self.tag, self.site = tag, 0

    def drop_path(self, i, x):
        prob = self.dpr[i]
        if prob == 0.0:
            return x
        keep = 1.0 - prob
        k = torch.from_numpy(self._keep(self.tag, self.site, x.shape[0], keep)).to(x.dtype)
        self.site += 1
        return x * k.view(-1, 1, 1) / keep

    def dropout(self, x):
        if self.p == 0.0:
            return x
        B, T, D = x.shape
        m = torch.stack([torch.from_numpy(self._mask(self.tag, b, T, D, 1.0 - self.p)) for b in range(B)])
        return x * m / (1.0 - self.p)


def _dp(tm, i, x):
    return x if tm is None else tm.drop_path(i, x)


def _k(i, mod, leaf, m):
    """state_dict key of block i: mod in {attn, mlp, norm1, norm2}, m = modality segment ('' for ufo)."""
    mm = (m + ".") if m else ""
    return f"transformer.blocks.{i}.{mod}.{mm}{leaf}"


def attention(sd, a: Arch, i, m, x, mask, bias):
    B, N, C = x.shape
    qb, vb = sd[_k(i, "attn", "q_bias", m)], sd[_k(i, "attn", "v_bias", m)]
    qkv_bias = torch.cat((qb, torch.zeros_like(vb), vb))
    qkv = F.linear(x, sd[_k(i, "attn", "qkv.weight", m)], qkv_bias)
    qkv = qkv.reshape(B, N, 3, a.H, -1).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (C // a.H) ** -0.5, qkv[1], qkv[2]
    attn = q.float() @ k.float().transpose(-2, -1)
    if bias is not None:
        attn = attn + bias.unsqueeze(0)
    if mask is not None:
        attn = attn.masked_fill(~mask.bool()[:, None, None, :], float("-inf"))
    attn = attn.softmax(dim=-1)
    x = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(x, sd[_k(i, "attn", "proj.weight", m)], sd[_k(i, "attn", "proj.bias", m)])


def mlp(sd, i, m, x):
    x = F.linear(x, sd[_k(i, "mlp", "fc1.weight", m)], sd[_k(i, "mlp", "fc1.bias", m)])
    x = F.gelu(x)
    return F.linear(x, sd[_k(i, "mlp", "fc2.weight", m)], sd[_k(i, "mlp", "fc2.bias", m)])


def ln(sd, i, which, m, x):
    return F.layer_norm(x, (x.shape[-1],), sd[_k(i, which, "weight", m)], sd[_k(i, which, "bias", m)], 1e-6)


def block(sd, a: Arch, i, x, mask, type_id, bias, tm=None):
    """type_id 0 image / 1 text / 2 vl.  ufo: shared weights, text/image attention separated below layer S in vl
    passes; all_moe: per-modality LN/attn/MLP, 'vl' expert from layer S on."""
    g1, g2 = sd[f"transformer.blocks.{i}.gamma_1"], sd[f"transformer.blocks.{i}.gamma_2"]
    moe = a.arch == "all_moe"
    deep = i >= a.S
    T = a.T
    if type_id in (0, 1) or deep:
        m = "" if not moe else ("v" if type_id == 0 else "l" if type_id == 1 else "vl")
        x = x + _dp(tm, i, g1 * attention(sd, a, i, m, ln(sd, i, "norm1", m, x), mask, bias))
        x = x + _dp(tm, i, g2 * mlp(sd, i, m, ln(sd, i, "norm2", m, x)))
        return x
    ml, mv = ("l", "v") if moe else ("", "")
    xt = ln(sd, i, "norm1", ml, x[:, :T])
    xi = ln(sd, i, "norm1", mv, x[:, T:])
    at = attention(sd, a, i, ml, xt, mask[:, :T], bias[:, :T, :T])
    ai = attention(sd, a, i, mv, xi, mask[:, T:], bias[:, T:, T:])
    x = x + _dp(tm, i, g1 * torch.cat([at, ai], 1))
    xt = mlp(sd, i, ml, ln(sd, i, "norm2", ml, x[:, :T]))
    xi = mlp(sd, i, mv, ln(sd, i, "norm2", mv, x[:, T:]))
    return x + _dp(tm, i, g2 * torch.cat([xt, xi], 1))


def rel_pos_bias(sd, index, a: Arch):
    b = F.embedding(index.long(), sd["relative_position_bias_table"]).permute(2, 0, 1).contiguous()
    return torch.chunk(b, a.L, dim=0)


def text_embed(sd, ids, masks, tm=None):
    e = F.embedding(ids, sd["text_embeddings.word_embeddings.weight"]) + sd["text_embeddings.token_type_embeddings.weight"][0]
    e = F.layer_norm(e, (e.shape[-1],), sd["text_embeddings.LayerNorm.weight"], sd["text_embeddings.LayerNorm.bias"], 1e-12)
    if tm is not None:
        e = tm.dropout(e)
    return e + F.embedding(torch.zeros_like(masks), sd["token_type_embeddings.weight"])


def image_embed(sd, a: Arch, img, type_idx=1):
    x = F.conv2d(img, sd["transformer.patch_embed.proj.weight"], sd["transformer.patch_embed.proj.bias"], stride=a.P)
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat((sd["transformer.cls_token"].expand(x.shape[0], -1, -1), x), dim=1)
    return x + sd["token_type_embeddings.weight"][type_idx]


def final_norm(sd, x):
    return F.layer_norm(x, (x.shape[-1],), sd["transformer.norm.weight"], sd["transformer.norm.bias"], 1e-6)


def infer(sd, a: Arch, idx, text_ids, text_masks, img, tm=None, tag=None):
    if tm is not None:
        tm.begin(tag)
    te = text_embed(sd, text_ids, text_masks, tm)
    ie = image_embed(sd, a, img)
    x = torch.cat([te, ie], 1)
    mask = torch.cat([text_masks, torch.ones(ie.shape[0], ie.shape[1], dtype=text_masks.dtype)], 1)
    bl = rel_pos_bias(sd, idx["text_imag_relative_position_index"], a)
    for i in range(a.L):
        x = block(sd, a, i, x, mask, 2, bl[i], tm)
    x = final_norm(sd, x)
    T = te.shape[1]
    cls = torch.tanh(F.linear(x[:, 0], sd["pooler.dense.weight"], sd["pooler.dense.bias"]))
    return {"text_feats": x[:, :T], "image_feats": x[:, T:], "cls_feats": cls, "raw_cls_feats": x[:, 0]}


def _unimodal(sd, a, x, mask, type_id, bl, vlffn, tm=None):
    hs = None
    for i in range(a.L):
        x = block(sd, a, i, x, mask, type_id, bl[i], tm)
        if i == a.S - 1:
            hs = x
    v = None
    if vlffn:
        v = hs
        for i in range(a.S, a.L):
            v = block(sd, a, i, v, mask, 2, bl[i], tm)
        v = final_norm(sd, v)
    return final_norm(sd, x), v


def _l2(x):
    return x / x.norm(dim=-1, keepdim=True)


def infer_text(sd, a, idx, text_ids, text_masks, vlffn=True, tm=None):
    if tm is not None:
        tm.begin("txt")
    x = text_embed(sd, text_ids, text_masks, tm)
    bl = rel_pos_bias(sd, idx["text_relative_position_index"], a)
    l, v = _unimodal(sd, a, x, text_masks, 1, bl, vlffn, tm)
    out = {"text_feats": l, "cls_feats": _l2(F.linear(l[:, 0], sd["ifm_text_proj.fc.weight"]))}
    if vlffn:
        out["cls_vlffn_feats"] = _l2(F.linear(v[:, 0], sd["ifm_vl_text_proj.fc.weight"]))
    return out


def infer_image(sd, a, idx, img, vlffn=True, tm=None):
    if tm is not None:
        tm.begin("img")
    x = image_embed(sd, a, img)
    mask = torch.ones(x.shape[0], x.shape[1], dtype=torch.long)
    bl = rel_pos_bias(sd, idx["relative_position_index"], a)
    vf, v = _unimodal(sd, a, x, mask, 0, bl, vlffn, tm)
    out = {"image_feats": vf, "cls_feats": _l2(F.linear(vf[:, 0], sd["ifm_image_proj.fc.weight"]))}
    if vlffn:
        out["cls_vlffn_feats"] = _l2(F.linear(v[:, 0], sd["ifm_vl_image_proj.fc.weight"]))
    return out


def mlm_head(sd, x):
    h = F.gelu(F.linear(x, sd["mlm_score.transform.dense.weight"], sd["mlm_score.transform.dense.bias"]))
    h = F.layer_norm(h, (h.shape[-1],), sd["mlm_score.transform.LayerNorm.weight"], sd["mlm_score.transform.LayerNorm.bias"], 1e-12)
    return F.linear(h, sd["mlm_score.decoder.weight"]) + sd["mlm_score.bias"]


def _sym_ce(li):
    gt = torch.arange(len(li))
    return (F.cross_entropy(li, gt) + F.cross_entropy(li.t(), gt)) / 2


def pretrain_step(sd, a: Arch, idx, batch, neg_img=None, neg_txt=None, tm=None):
    """One training_step (mlm + ifm + itm, single process).  neg_img / neg_txt: indices of the hard negatives; the
    reference samples them with torch.multinomial, for B == 2 the draw is forced (the other sample)."""
    out = {}
    r = infer(sd, a, idx, batch["text_ids_mlm"], batch["text_masks"], batch["image"], tm, "mlm")
    logits = mlm_head(sd, r["text_feats"])
    out["mlm_logits"] = logits
    out["mlm_loss"] = F.cross_entropy(logits.view(-1, logits.shape[-1]), batch["text_labels_mlm"].view(-1), ignore_index=-100)
    im = infer_image(sd, a, idx, batch["image"], tm=tm)
    tx = infer_text(sd, a, idx, batch["text_ids"], batch["text_masks"], tm=tm)
    li = sd["logit_scale"].exp() * im["cls_feats"] @ tx["cls_feats"].t()
    lv = sd["logit_vl_scale"].exp() * im["cls_vlffn_feats"] @ tx["cls_vlffn_feats"].t()
    out["ifm_i2t_logits"] = li
    out["ifm_loss"] = (_sym_ce(li) + _sym_ce(lv)) * 0.5
    B = batch["text_ids"].shape[0]
    with torch.no_grad():
        wi = F.softmax(li[:B], 1).clone()
        wt = F.softmax(li.t()[:B], 1).clone()
        wi.fill_diagonal_(0)
        wt.fill_diagonal_(0)
        if neg_img is None:
            neg_img = torch.multinomial(wt, 1).squeeze(1)
        if neg_txt is None:
            neg_txt = torch.multinomial(wi, 1).squeeze(1)
    pos = infer(sd, a, idx, batch["text_ids"], batch["text_masks"], batch["image"], tm, "pos")
    ni = infer(sd, a, idx, batch["text_ids"], batch["text_masks"], batch["image"][neg_img], tm, "negimg")
    nt = infer(sd, a, idx, batch["text_ids"][neg_txt], batch["text_masks"][neg_txt], batch["image"], tm, "negtxt")
    cls = torch.cat([pos["cls_feats"], ni["cls_feats"], nt["cls_feats"]], 0)
    itm_logits = F.linear(cls, sd["itm_score.fc.weight"], sd["itm_score.fc.bias"])
    labels = torch.cat([torch.ones(B), torch.zeros(B), torch.zeros(B)]).long()
    out["itm_logits"] = itm_logits
    out["itm_loss"] = F.cross_entropy(itm_logits, labels)
    out["total_loss"] = out["mlm_loss"] + out["ifm_loss"] + out["itm_loss"]
    return out


def irtr_step(sd, a: Arch, idx, batch):
    im = infer_image(sd, a, idx, batch["image"], vlffn=False)
    tx = infer_text(sd, a, idx, batch["text_ids"], batch["text_masks"], vlffn=False)
    li = sd["logit_scale"].exp() * im["cls_feats"] @ tx["cls_feats"].t()
    return {"irtr_i2t_logits": li, "irtr_loss": _sym_ce(li)}


def gram_inputs(sd, a: Arch, idx, batch):
    """Gram matrices X^T X (fp64) of the inputs of every hooked linear during an irtr forward of an all_moe model:
    restates cache_gram_matrices.py:246-281 (keys = module names; qkv's key is the Attention module)."""
    grams = {}

    def add(name, x):
        f = x.reshape(-1, x.shape[-1]).to(torch.float64)
        grams[name] = grams.get(name, 0) + f.T @ f

    for m, x, mask, index in (("v", image_embed(sd, a, batch["image"]), None, "relative_position_index"),
                              ("l", text_embed(sd, batch["text_ids"], batch["text_masks"]), batch["text_masks"],
                               "text_relative_position_index")):
        bl = rel_pos_bias(sd, idx[index], a)
        if mask is None:
            mask = torch.ones(x.shape[0], x.shape[1], dtype=torch.long)
        for i in range(a.L):
            g1, g2 = sd[f"transformer.blocks.{i}.gamma_1"], sd[f"transformer.blocks.{i}.gamma_2"]
            h = ln(sd, i, "norm1", m, x)
            add(f"transformer.blocks.{i}.attn.{m}", h)
            B, N, C = h.shape
            qb, vb = sd[_k(i, "attn", "q_bias", m)], sd[_k(i, "attn", "v_bias", m)]
            qkv = F.linear(h, sd[_k(i, "attn", "qkv.weight", m)], torch.cat((qb, torch.zeros_like(vb), vb)))
            qkv = qkv.reshape(B, N, 3, a.H, -1).permute(2, 0, 3, 1, 4)
            att = (qkv[0] * (C // a.H) ** -0.5) @ qkv[1].transpose(-2, -1) + bl[i].unsqueeze(0)
            att = att.masked_fill(~mask.bool()[:, None, None, :], float("-inf")).softmax(-1)
            o = (att @ qkv[2]).transpose(1, 2).reshape(B, N, C)
            add(f"transformer.blocks.{i}.attn.{m}.proj", o)
            x = x + g1 * F.linear(o, sd[_k(i, "attn", "proj.weight", m)], sd[_k(i, "attn", "proj.bias", m)])
            h2 = ln(sd, i, "norm2", m, x)
            add(f"transformer.blocks.{i}.mlp.{m}.fc1", h2)
            a1 = F.gelu(F.linear(h2, sd[_k(i, "mlp", "fc1.weight", m)], sd[_k(i, "mlp", "fc1.bias", m)]))
            add(f"transformer.blocks.{i}.mlp.{m}.fc2", a1)
            x = x + g2 * F.linear(a1, sd[_k(i, "mlp", "fc2.weight", m)], sd[_k(i, "mlp", "fc2.bias", m)])
    return grams


def recall_ref(img_feats, txt_feats, iids, tiids, ks=(1, 5, 10)):
    """TEST INFRASTRUCTURE.  numpy restatement of the recall arithmetic of compute_irtr_recall, reference
    src/vilt/modules/objectives.py:679-710: scores = img @ txt^T; a query is a hit at k when any of its k best
    candidates (largest score first, ties by lower index as torch.topk on CPU resolves them) carries its id.
    Returns (ir_r1, ir_r5, ir_r10, tr_r1, tr_r5, tr_r10) as float32 means, like the reference's .float().mean()."""
    import numpy as np
    scores = np.asarray(img_feats, dtype=np.float32) @ np.asarray(txt_feats, dtype=np.float32).T
    iids, tiids = np.asarray(iids), np.asarray(tiids)
    tr, ir = [], []
    for k in ks:
        top_txt = np.argsort(-scores, axis=1, kind="stable")[:, :k]
        tr.append(np.float32((iids[:, None] == tiids[top_txt]).any(axis=1).astype(np.float32).mean()))
        top_img = np.argsort(-scores, axis=0, kind="stable")[:k, :]
        ir.append(np.float32((tiids[None, :] == iids[top_img]).any(axis=0).astype(np.float32).mean()))
    return tuple(ir) + tuple(tr)
