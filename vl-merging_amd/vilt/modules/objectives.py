"""Losses of the hot path (reference src/vilt/modules/objectives.py): compute_mlm :88, compute_ifm :248,
compute_itm_hardneg :146, compute_irtr :372, init_weights :713.  The transformer passes and every GEMM run on the
HIP kernels; the scalar loss algebra on [B, *] logits stays in torch.
"""
import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from ... import engine, ops


_LOCAL_ONLY = [False]


class local_only:
    """with local_only(): the losses see a world of one -- no feature / candidate gathers.  For sweeps whose ranks run
    DIFFERENT numbers of forward passes and do not need the loss (the Gram cache deals batches round-robin and only wants
    the hooks to fire: cache_gram_matrices.py:339 drives them with trainer.validate, whose sampler pads instead)."""

    def __enter__(self):
        self.prev = _LOCAL_ONLY[0]
        _LOCAL_ONLY[0] = True
        return self

    def __exit__(self, *a):
        _LOCAL_ONLY[0] = self.prev
        return False


def _world():
    if not _LOCAL_ONLY[0] and dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def _gather_first_own(t):
    """all_gather without autograd, own slice re-inserted at index 0 (objectives.py:269-286): gradients flow only
    through the local features."""
    world, rank = _world()
    if world == 1:
        return t
    gathered = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(gathered, t.detach().contiguous())
    return torch.cat([t] + gathered[:rank] + gathered[rank + 1:])


def init_weights(module):
    if isinstance(module, (nn.Linear, nn.Embedding)):
        module.weight.data.normal_(mean=0.0, std=0.02)
    elif isinstance(module, nn.LayerNorm):
        module.bias.data.zero_()
        module.weight.data.fill_(1.0)
    if isinstance(module, nn.Linear) and module.bias is not None:
        module.bias.data.zero_()


def compute_mlm(pl_module, batch):
    infer = pl_module.infer(batch, mask_text=True, mask_image=False)
    mlm_logits = pl_module.mlm_score(infer["text_feats"])
    mlm_labels = infer["text_labels"]
    mlm_loss = engine.cross_entropy(mlm_logits.view(-1, pl_module.hparams.config["vocab_size"]), mlm_labels.view(-1), ignore_index=-100)
    w = pl_module.hparams.config["vl_mlm_weight"]
    return {"mlm_loss": mlm_loss if w == 1 else engine.weighted_sum([mlm_loss], [w]), "mlm_logits": mlm_logits,
            "mlm_labels": mlm_labels, "mlm_ids": infer["text_ids"]}


def _others(t):
    """The other ranks' rows of a feature matrix (no autograd), in rank order without this rank's: what _gather_first_own appends."""
    world, rank = _world()
    if world == 1:
        return None
    gathered = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(gathered, t.detach().contiguous())
    return torch.cat(gathered[:rank] + gathered[rank + 1:])


def _contrastive_pair(image_features, text_features, log_scale_param):
    """(loss, logits_per_image, logits_per_text, exp(scale)) of the symmetric contrastive cross-entropy (objectives.py:274-300):
    own features first, the gathered ones behind them without gradient.  GPU: engine.contrastive_loss (two launches, forward
    and gradient); otherwise the reference's torch formulation."""
    if image_features.is_cuda and engine._FUSED_LOSS and log_scale_param.numel() == 1:
        loss, li, scale = engine.contrastive_loss(image_features.float(), text_features.float(), log_scale_param,
                                                  _others(image_features.float()), _others(text_features.float()))
        return loss, li, li.t(), scale
    logit_scale = log_scale_param.exp().mean()
    all_img = _gather_first_own(image_features)
    all_txt = _gather_first_own(text_features)
    li = logit_scale * all_img @ all_txt.t()
    gt = torch.arange(len(li), device=li.device)
    return (F.cross_entropy(li.float(), gt) + F.cross_entropy(li.t().float(), gt)) / 2, li, li.t(), logit_scale


def _itm_labels(bsz, device):
    """(float, int64) labels of the ITM head: B positives then 2B negatives (objectives.py:187-190); built once per batch size."""
    key = ("itm", bsz, str(device))
    t = _ARANGE.get(key)
    if t is None:
        f = torch.cat([torch.ones(bsz, device=device), torch.zeros(2 * bsz, device=device)])
        t = _ARANGE[key] = (f, f.long())
    return t


def _labels_arange(n, device):
    key = (n, str(device))
    t = _ARANGE.get(key)
    if t is None:
        t = _ARANGE[key] = torch.arange(n, device=device)
    return t


_ARANGE = {}


def compute_ifm(pl_module, batch, aggregate=True):
    if getattr(pl_module, "fuse_unimodal_passes", False):
        infer_imag, infer_text = pl_module.infer_unimodal_pair(batch, with_vlffn=True)
    else:
        infer_imag = pl_module.infer_image(batch, mask_image=False)
        infer_text = pl_module.infer_text(batch, mask_text=False)
    ifm_loss, li, lt, logit_scale = _contrastive_pair(infer_imag["cls_feats"], infer_text["cls_feats"], pl_module.logit_scale)
    ifm_vlffn_loss, _, _, logit_vl_scale = _contrastive_pair(infer_imag["cls_vlffn_feats"], infer_text["cls_vlffn_feats"],
                                                             pl_module.logit_vl_scale)
    total = engine.weighted_sum([ifm_loss, ifm_vlffn_loss], [0.5 * pl_module.hparams.config["ifm_weight"], 0.5])  # (w a + b) * 0.5
    return {"ifm_loss": total, "ifm_i2t_logits": li, "ifm_t2i_logits": lt, "ifm_labels": _labels_arange(len(li), li.device),
            "ifm_logit_scale": logit_scale, "ifm_logit_vl_scale": logit_vl_scale}


def compute_irtr(pl_module, batch, aggregate=True):
    if getattr(pl_module, "fuse_unimodal_passes", False):
        infer_imag, infer_text = pl_module.infer_unimodal_pair(batch, with_vlffn=False)
    else:
        infer_imag = pl_module.infer_image_ft(batch, mask_image=False)
        infer_text = pl_module.infer_text_ft(batch, mask_text=False)
    loss, li, lt, logit_scale = _contrastive_pair(infer_imag["cls_feats"], infer_text["cls_feats"], pl_module.logit_scale)
    return {"irtr_loss": loss, "irtr_i2t_logits": li, "irtr_t2i_logits": lt, "irtr_labels": _labels_arange(len(li), li.device),
            "irtr_logit_scale": logit_scale}


def _gather_images(img):
    """_gather_cat for the fp32 image batch (the candidates of hard-negative sampling, objectives.py:176-178): over RCCL
    the images travel as bf16 and come back as fp32.  The patch embedding rounds its input to bf16 anyway (im2col feeds
    the MFMA GEMM), so the negatives' features are bit-identical to gathering fp32, at half the xGMI bytes
    (39 MB -> 19 MB per rank and step at 384^2, B = 22)."""
    world, _ = _world()
    if world == 1 or not img.is_cuda or img.dtype != torch.float32:
        return _gather_cat(img)
    return _gather_cat(img.to(torch.bfloat16)).float()


class _CandidatePrefetch:
    """The hard-negative candidates (text ids, masks, images of every rank) depend only on the input batch: their
    all-gathers are issued asynchronously at the top of the step and consumed after the contrastive pass, so the
    156 MB image gather (8 ranks, bf16) overlaps ~12 ms of compute instead of sitting between two passes."""

    def __init__(self, batch):
        world, rank = _world()
        self.rank = rank
        img = batch["image"][0]
        self.as_bf16 = img.is_cuda and img.dtype == torch.float32
        self.src = [batch["text_ids"].contiguous(), batch["text_masks"].contiguous(),
                    (img.to(torch.bfloat16) if self.as_bf16 else img).contiguous()]
        self.bufs, self.works = [], []
        for t in self.src:
            g = [torch.zeros_like(t) for _ in range(world)]
            self.bufs.append(g)
            self.works.append(dist.all_gather(g, t, async_op=True))

    def result(self):
        out = []
        for t, g, w in zip(self.src, self.bufs, self.works):
            w.wait()
            out.append(torch.cat([t] + g[:self.rank] + g[self.rank + 1:]))
        if self.as_bf16:
            out[2] = out[2].float()
        return out


def prefetch_negative_candidates(batch):
    """Start the candidate all-gathers of compute_itm_hardneg / compute_mlm_itm_fused early (no-op on one rank)."""
    world, _ = _world()
    if world > 1 and "_neg_candidates" not in batch:
        batch["_neg_candidates"] = _CandidatePrefetch(batch)


def _gather_cat(t):
    world, rank = _world()
    if world == 1:
        return t
    gathered = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(gathered, t.contiguous())
    return torch.cat([t] + gathered[:rank] + gathered[rank + 1:])


_TORCH_MULTINOMIAL = torch.multinomial


def _draw_negatives(sim_i2t, sim_t2i, bsz):
    """(neg_img, neg_txt) int64 [bsz]: one draw per row from softmax(sim_t2i[i]) / softmax(sim_i2t[i]) with the own pair removed
    (:197-215: F.softmax, fill_diagonal_(0), torch.multinomial(., 1)).  On the GPU: ONE launch from 2 * bsz uniforms
    (csrc/frontops.hip vlm_sample_negatives, inverse CDF: the same distribution) instead of torch's softmax / fill / multinomial
    chains.  A test that replaces torch.multinomial (a deterministic stand-in) gets the torch form."""
    a, b = sim_t2i[:bsz], sim_i2t[:bsz]
    if (torch.multinomial is _TORCH_MULTINOMIAL and a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32
            and a.dim() == 2 and a.shape == b.shape):
        idx = ops.sample_negatives(a, b, bsz, torch.rand(2, bsz, device=a.device, dtype=torch.float32))
        return idx[0], idx[1]
    weights_i2t = F.softmax(b.float(), dim=1)
    weights_t2i = F.softmax(a.float(), dim=1)
    weights_i2t.fill_diagonal_(0)
    weights_t2i.fill_diagonal_(0)
    return torch.multinomial(weights_t2i, 1).squeeze(1), torch.multinomial(weights_i2t, 1).squeeze(1)


def _sample_hard_negatives(pl_module, batch, sim_i2t, sim_t2i):
    """ONE batched multinomial per direction over the (gathered) candidates, device-side gathers (:176-229)."""
    bsz = batch["text_ids"].size(0)
    with torch.no_grad():
        pre = batch.pop("_neg_candidates", None)
        if pre is not None:
            all_text_ids, all_text_masks, all_image = pre.result()
        else:
            all_text_ids = _gather_cat(batch["text_ids"])
            all_text_masks = _gather_cat(batch["text_masks"])
            all_image = _gather_images(batch["image"][0])
        neg_img, neg_txt = _draw_negatives(sim_i2t, sim_t2i, bsz)
        return all_image[neg_img], all_text_ids[neg_txt], all_text_masks[neg_txt]


def compute_mlm_itm_fused(pl_module, batch, sim_i2t, sim_t2i):
    """compute_mlm (:88) + compute_itm_hardneg (:146) with their FOUR joint passes (masked text, positive pairs,
    negative images, negative texts) run as ONE pass over 4B samples: the passes share weights and have no
    cross-sample op, so losses and gradients are those of the four separate passes, with 4x larger GEMMs and a
    quarter of the launches (MI355X: 288 GB of HBM easily holds the 4B activations)."""
    bsz = batch["text_ids"].size(0)
    images_neg, text_ids_neg, text_masks_neg = _sample_hard_negatives(pl_module, batch, sim_i2t, sim_t2i)
    img = batch["image"][0]
    big = {
        "text_ids": torch.cat([batch["text_ids_mlm"], batch["text_ids"], batch["text_ids"], text_ids_neg]),
        "text_masks": torch.cat([batch["text_masks"], batch["text_masks"], batch["text_masks"], text_masks_neg]),
        "text_labels": torch.cat([batch["text_labels_mlm"]] + [batch["text_labels"]] * 3),
        "image": [torch.cat([img, img, images_neg, img])],
    }
    infer = pl_module.infer(big, mask_text=False, mask_image=False)
    mlm_logits = pl_module.mlm_score(engine.row_range(infer["text_feats"], 0, bsz))
    mlm_labels = batch["text_labels_mlm"]
    mlm_loss = engine.cross_entropy(mlm_logits.view(-1, pl_module.hparams.config["vocab_size"]), mlm_labels.view(-1), ignore_index=-100)
    itm_labels, itm_long = _itm_labels(bsz, img.device)
    itm_logits = pl_module.itm_score(engine.row_range(infer["cls_feats"], bsz, None))
    itm_loss = engine.small_cross_entropy(itm_logits, itm_long)
    w = pl_module.hparams.config["vl_mlm_weight"]
    return {"mlm_loss": mlm_loss if w == 1 else engine.weighted_sum([mlm_loss], [w]), "mlm_logits": mlm_logits,
            "mlm_labels": mlm_labels, "mlm_ids": batch["text_ids_mlm"], "itm_loss": itm_loss,
            "itm_logits": itm_logits, "itm_labels": itm_labels}


def compute_itm_hardneg(pl_module, batch, sim_i2t, sim_t2i):
    """Hard-negative ITM (:146-245).  The reference draws each negative with a per-row torch.multinomial(...).item()
    (2B host syncs); here ONE batched multinomial per direction and device-side gathers: same distribution, no
    host round trip."""
    bsz = batch["text_ids_mlm"].size(0)
    dev = batch["text_ids"].device
    itm_labels, itm_long = _itm_labels(bsz, dev)
    infer_pos = pl_module.infer(batch, mask_text=False, mask_image=False)
    with torch.no_grad():
        all_text_ids = _gather_cat(infer_pos["text_ids"])
        all_text_masks = _gather_cat(infer_pos["text_masks"])
        all_image = _gather_images(infer_pos["image"])
        neg_img, neg_txt = _draw_negatives(sim_i2t, sim_t2i, bsz)
        images_neg = all_image[neg_img]
        text_ids_neg = all_text_ids[neg_txt]
        text_masks_neg = all_text_masks[neg_txt]
    batch_imgs_neg = {"image": [images_neg], "text_ids": batch["text_ids"], "text_labels": batch["text_labels"],
                      "text_masks": batch["text_masks"]}
    infer_imags_neg = pl_module.infer(batch_imgs_neg, mask_text=False, mask_image=False)
    batch_text_neg = {"image": batch["image"], "text_ids": text_ids_neg, "text_labels": batch["text_labels"],
                      "text_masks": text_masks_neg}
    infer_text_neg = pl_module.infer(batch_text_neg, mask_text=False, mask_image=False)
    all_cls_feats = torch.cat([infer_pos["cls_feats"], infer_imags_neg["cls_feats"], infer_text_neg["cls_feats"]], dim=0)
    itm_logits = pl_module.itm_score(all_cls_feats)
    itm_loss = engine.small_cross_entropy(itm_logits, itm_long)
    return {"itm_loss": itm_loss, "itm_logits": itm_logits, "itm_labels": itm_labels}


# ---------------------------------------------------------------------------------------------------- VQA / NLVR2
def compute_vqa(pl_module, batch):
    """objectives.py:446-486: soft-target BCE over the answer vocabulary, scaled by its size (ban-vqa convention)."""
    infer = pl_module.infer(batch, mask_text=False, mask_image=False)
    logits = pl_module.vqa_classifier(infer["cls_feats"]).float()
    targets = torch.zeros(len(logits), pl_module.hparams.config["vqav2_label_size"], device=logits.device)
    rows = [i for i, labels in enumerate(batch["vqa_labels"]) for _ in labels]
    if rows:  # one scatter instead of the reference's per-answer python loop (:458-460)
        cols = [int(l) for labels in batch["vqa_labels"] for l in labels]
        vals = [float(s) for scores in batch["vqa_scores"] for s in scores]
        targets[torch.tensor(rows, device=logits.device), torch.tensor(cols, device=logits.device)] = \
            torch.tensor(vals, device=logits.device)
    loss = F.binary_cross_entropy_with_logits(logits, targets) * targets.shape[1]
    return {"vqa_loss": loss, "vqa_logits": logits, "vqa_targets": targets, "vqa_labels": batch["vqa_labels"],
            "vqa_scores": batch["vqa_scores"]}


def compute_nlvr2(pl_module, batch):
    """objectives.py:512-532: the sentence with each of the two images (token types 1 and 2, keys image_0 / image_1),
    the two pooled features concatenated into the pair classifier."""
    infer1 = pl_module.infer(batch, mask_text=False, mask_image=False, image_token_type_idx=1)
    infer2 = pl_module.infer(batch, mask_text=False, mask_image=False, image_token_type_idx=2)
    logits = pl_module.nlvr2_classifier(torch.cat([infer1["cls_feats"], infer2["cls_feats"]], dim=-1)).float()
    labels = torch.as_tensor(batch["answers"], device=logits.device).long()
    return {"nlvr2_loss": F.cross_entropy(logits, labels), "nlvr2_logits": logits, "nlvr2_labels": labels}


# ---------------------------------------------------------------------------------------------------- retrieval recall
def recall_at_k(scores, iids, tiids, ks=(1, 5, 10)):
    """(ir_r1, ir_r5, ir_r10, tr_r1, tr_r5, tr_r10) from the image x text score matrix, objectives.py:683-710:
    text retrieval (tr): an image is a hit at k when one of its top-k captions carries the image's id; image retrieval
    (ir): a caption is a hit when one of its top-k images is its own.  `iids` [n_img], `tiids` [n_txt] integer ids."""
    iids = torch.as_tensor(iids, device=scores.device)
    tiids = torch.as_tensor(tiids, device=scores.device)
    tr, ir = [], []
    for k in ks:
        top_txt = tiids[scores.topk(k, dim=1).indices]               # [n_img, k]
        tr.append((iids.unsqueeze(1) == top_txt).float().max(dim=1)[0].mean())
        top_img = iids[scores.topk(k, dim=0).indices]                # [k, n_txt]
        ir.append((tiids.unsqueeze(0) == top_img).float().max(dim=0)[0].mean())
    return tuple(ir) + tuple(tr)


def _gather_ragged_rows(t):
    """all_gather of per-rank feature blocks with different row counts, returned in rank order."""
    world, rank = _world()
    if world == 1:
        return [t]
    n = torch.tensor([t.shape[0]], device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s) for s in sizes]
    pad = torch.zeros(max(sizes), t.shape[1], device=t.device, dtype=t.dtype)
    pad[: t.shape[0]] = t
    out = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return [o[:s] for o, s in zip(out, sizes)]


@torch.no_grad()
def compute_irtr_recall(pl_module, text_preload, image_preload):
    """Retrieval evaluation, objectives.py:572-710, over the batch lists the reference prefetches from its datamodule
    (`text_preload`: dicts with text_ids / text_masks / text_labels / img_index; `image_preload`: dicts with
    image=[tensor] / img_index / text_masks).  The reference repeats both feature sweeps on every rank; here rank r
    encodes batches r, r+W, ... and the [n, D] features are all-gathered (the sweeps are the cost, the score GEMM and
    top-k are O(n_img * n_txt))."""
    world, rank = _world()
    dev = pl_module.device

    def sweep(batches, encode):
        mine = [encode(b) for b in batches[rank::world]]
        D = pl_module.hparams.config["hidden_size"]
        local = torch.cat(mine) if mine else torch.zeros(0, D, device=dev)
        per_rank = _gather_ragged_rows(local.float())
        # undo the round-robin: batch j sits in rank j % W's block, in order
        counts = [b_count(b) for b in batches]
        cursor = [0] * world
        parts = []
        for j, n in enumerate(counts):
            r = j % world
            parts.append(per_rank[r][cursor[r]: cursor[r] + n])
            cursor[r] += n
        return torch.cat(parts) if parts else local

    def b_count(b):
        return len(b["img_index"])

    def enc_text(b):
        return pl_module.infer_text_ft({"text_ids": b["text_ids"].to(dev), "text_masks": b["text_masks"].to(dev),
                                        "text_labels": b["text_labels"].to(dev)})["cls_feats"]

    def enc_image(b):
        return pl_module.infer_image_ft({"image": [b["image"][0].to(dev)],
                                         "text_masks": b["text_masks"].to(dev)})["cls_feats"]

    txt = sweep(text_preload, enc_text)
    img = sweep(image_preload, enc_image)
    tiids = torch.tensor([i for b in text_preload for i in b["img_index"]], device=dev)
    iids = torch.tensor([i for b in image_preload for i in b["img_index"]], device=dev)
    scores = img @ txt.t()
    return recall_at_k(scores, iids, tiids) + ({"txt_cls_feats": txt, "img_cls_feats": img, "scores": scores},)
