"""The data side of `run.py` for the hot path: Arrow shards -> per-rank batch streams.

Mirrors what the reference's datamodules do around its datasets, nothing more:
  * shard names per dataset and split (src/vilt/datasets/{coco_caption_karpathy,f30k_caption_karpathy,vg_caption,
    sbu_caption,conceptual_caption}_dataset.py); a directory without any of those names (synthetic shards) serves
    every `*.arrow` file it holds to every split;
  * tokenizer + masked-language-model collator (datamodules/datamodule_base.py:61-74; `vl_mlm_prob` for the "vl" task of
    the multi-task pre-training, multi_multitask_datamodule.py:21-29);
  * a DistributedSampler per split: shuffled per epoch for training, `drop_last=True`
    (multitask_datamodule.py:58-65); evaluation batches are dealt round-robin, the last one may be ragged.
Decoding and collation run in the calling process (the reference's DataLoader workers are throughput plumbing).
"""
import os

import torch

from .datasets import ArrowDataset, build_synthetic_tokenizer

SPLIT_NAMES = {
    "coco": {"train": ["coco_caption_karpathy_train"], "val": ["coco_caption_karpathy_val"],
             "test": ["coco_caption_karpathy_test"]},
    "f30k": {"train": ["f30k_caption_karpathy_train", "f30k_caption_karpathy_val"], "val": ["f30k_caption_karpathy_test"],
             "test": ["f30k_caption_karpathy_test"]},
    "vg": {"train": ["code224_vg"], "val": [], "test": []},
    "sbu": {"train": ["code224_sbu_%d" % i for i in range(9)], "val": [], "test": []},
    "gcc": {"train": ["code224_conceptual_caption_train_%d" % i for i in range(30)], "val": [], "test": []},
}


def _flatten(datasets):
    out = []
    for d in datasets:
        out.extend(_flatten(d) if isinstance(d, (list, tuple)) else [d])
    return out


def shard_names(cfg, split):
    root = cfg["data_root"]
    present = sorted(f[:-6] for f in os.listdir(root) if f.endswith(".arrow"))
    want = []
    for d in _flatten(cfg["datasets"]):
        want.extend(SPLIT_NAMES.get(d, {}).get(split, []))
    names = [n for n in want if n in present]
    if names:
        return names
    if any(n in present for d in SPLIT_NAMES.values() for s in d.values() for n in s):
        return []  # a real dataset directory that has nothing for this split
    return present  # synthetic shards: one pool for every split


def load_tokenizer(cfg):
    """bert-base-uncased in the reference (downloaded, datamodule_base.py:13-24); no network here: `vocab.txt` next to the
    shards -- a full WordPiece vocabulary goes through BertTokenizer, a small synthetic one through the synthetic builder."""
    vocab = os.path.join(cfg["data_root"], "vocab.txt")
    if not os.path.isfile(vocab):
        raise FileNotFoundError("%s: the tokenizer vocabulary must sit next to the Arrow shards" % vocab)
    if os.path.getsize(vocab) < 65536:
        return build_synthetic_tokenizer(vocab)
    from transformers import BertTokenizer
    return BertTokenizer(vocab, do_lower_case=True)


class ArrowBatches:
    """Per-rank batch streams of one split."""

    def __init__(self, cfg, split, rank=0, world=1, image_only=False, tokenizer=None):
        from transformers import DataCollatorForLanguageModeling
        self.cfg, self.split, self.rank, self.world = cfg, split, rank, world
        self.tokenizer = tokenizer or load_tokenizer(cfg)
        names = shard_names(cfg, split)
        keys = cfg["val_transform_keys"] if split != "train" else ["square_transform"]
        keys = [k for k in keys if k == "square_transform"] or ["square_transform"]  # RandAugment: a recipe, not the contract
        self.data = ArrowDataset(cfg["data_root"], keys, cfg["image_size"], names, text_column_name="caption",
                                 max_text_len=cfg["max_text_len"], max_vl_text_len=cfg["max_vl_text_len"],
                                 draw_false_image=cfg["draw_false_image"], draw_false_text=cfg["draw_false_text"],
                                 image_only=image_only, tokenizer=self.tokenizer)
        prob = cfg["vl_mlm_prob"] if cfg["tasks"] is not None else cfg["mlm_prob"]
        self.collator = DataCollatorForLanguageModeling(tokenizer=self.tokenizer, mlm=True, mlm_probability=prob)
        self.B = cfg["per_gpu_batchsize"] or 2

    def __len__(self):
        return len(self.data)

    def steps_per_epoch(self):
        return (len(self.data) // self.world) // self.B  # DistributedSampler(drop_last) then DataLoader(drop_last)

    def _collate(self, idxs, device):
        b = self.data.collate([self.data[i] for i in idxs], self.collator)
        out = {}
        for k, v in b.items():
            if torch.is_tensor(v):
                out[k] = v.to(device)
            elif isinstance(v, list) and v and torch.is_tensor(v[0]):
                out[k] = [t.to(device) for t in v]
            else:
                out[k] = v
        return out

    def train_epoch(self, epoch, device, skip=0):
        """DistributedSampler(shuffle=True, seed=0).set_epoch(epoch) semantics: one permutation per epoch shared by the
        ranks, rank r takes elements r, r + W, ...; incomplete batches dropped.  `skip`: batches of this epoch a resumed
        run has already consumed (passed over without decoding)."""
        g = torch.Generator().manual_seed(int(self.cfg["seed"]) + epoch)
        perm = torch.randperm(len(self.data), generator=g).tolist()
        per_rank = len(perm) // self.world
        mine = perm[self.rank: per_rank * self.world: self.world]
        for lo in range(skip * self.B, len(mine) - self.B + 1, self.B):
            yield self._collate(mine[lo: lo + self.B], device)

    def eval_batches(self, device, all_ranks=False):
        """In order; batch j belongs to rank j % world unless `all_ranks` (retrieval preloads shard later by themselves)."""
        starts = list(range(0, len(self.data), self.B))
        for j, lo in enumerate(starts):
            if not all_ranks and j % self.world != self.rank:
                continue
            yield self._collate(range(lo, min(lo + self.B, len(self.data))), device)
