"""Synthetic image + text batches of the benchmarked shape: what `bench.py`, `run.py` without Arrow shards and
`cache_gram_matrices.py` without a dataset feed the model.  Seeded per rank; built on the CPU generator, uploaded once."""
import torch


def synthetic_batch(B, image_size, T, vocab, seed, device, mlm_prob=0.25, loss_names=None, vqav2_label_size=3129):
    """image ~ U(-1,1); ids: [CLS]=101, length ~ U{8..T}, ids ~ U{1000..vocab-1}, [SEP]=102, pad 0; `mlm_prob` of the
    non-special positions -> [MASK]=103 with the original id as label (at least one per sample).  Returned as {"vl": batch},
    the wrapping the reference's training_step unpacks (vilt_module.py:1485).  With `loss_names` the fields of the down-stream
    tasks come along: vqa -> `vqa_labels` / `vqa_scores` (1..3 answers per question with soft scores, what the collate of
    datasets/vqav2_dataset.py hands over), nlvr2 -> `image_0`, `image_1` and `answers` (datasets/nlvr2_dataset.py)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    image = torch.rand(B, 3, image_size, image_size, generator=g) * 2 - 1
    ids = torch.zeros(B, T, dtype=torch.long)
    masks = torch.zeros(B, T, dtype=torch.long)
    ids_mlm = torch.zeros(B, T, dtype=torch.long)
    labels_mlm = torch.full((B, T), -100, dtype=torch.long)
    for b in range(B):
        ln = int(torch.randint(8, T + 1, (1,), generator=g))
        ids[b, 0] = 101
        ids[b, 1:ln - 1] = torch.randint(1000, vocab, (ln - 2,), generator=g)
        ids[b, ln - 1] = 102
        masks[b, :ln] = 1
        ids_mlm[b] = ids[b]
        pick = torch.rand(ln - 2, generator=g) < mlm_prob
        if not pick.any():
            pick[0] = True
        pos = pick.nonzero().squeeze(1) + 1
        labels_mlm[b, pos] = ids[b, pos]
        ids_mlm[b, pos] = 103
    batch = {"image": [image.to(device)], "text_ids": ids.to(device), "text_masks": masks.to(device),
             "text_labels": torch.full((B, T), -100, dtype=torch.long, device=device),
             "text_ids_mlm": ids_mlm.to(device), "text_labels_mlm": labels_mlm.to(device)}
    ln_ = loss_names or {}
    if ln_.get("vqa", 0) > 0:
        labels, scores = [], []
        for b in range(B):
            n = int(torch.randint(1, 4, (1,), generator=g))
            labels.append([int(v) for v in torch.randperm(vqav2_label_size, generator=g)[:n]])
            scores.append([float(v) for v in (torch.randint(1, 4, (n,), generator=g).float() * 0.3).clamp(max=1.0)])
        batch["vqa_labels"], batch["vqa_scores"] = labels, scores
    if ln_.get("nlvr2", 0) > 0:
        second = torch.rand(B, 3, image_size, image_size, generator=g) * 2 - 1
        batch["image_0"], batch["image_1"] = batch["image"], [second.to(device)]
        batch["answers"] = [int(v) for v in torch.randint(0, 2, (B,), generator=g)]
    return {"vl": batch}
