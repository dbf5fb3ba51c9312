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

    def _host_batch(self, idxs, pin=False):
        """Decode + tokenise + collate on the host (the expensive part with real JPEG shards); `pin`: page-locked tensors,
        so that the upload is an asynchronous DMA."""
        b = self.data.collate([self.data[i] for i in idxs], self.collator)
        if pin:
            pinned = lambda t: t.pin_memory() if t.numel() else t  # noqa: E731
            for k, v in list(b.items()):
                if torch.is_tensor(v):
                    b[k] = pinned(v)
                elif isinstance(v, list) and v and torch.is_tensor(v[0]):
                    b[k] = [pinned(t) for t in v]
        return b

    @staticmethod
    def _upload(b, device, non_blocking=False):
        out = {}
        for k, v in b.items():
            if torch.is_tensor(v):
                out[k] = v.to(device, non_blocking=non_blocking)
            elif isinstance(v, list) and v and torch.is_tensor(v[0]):
                out[k] = [t.to(device, non_blocking=non_blocking) for t in v]
            else:
                out[k] = v
        return out

    def _collate(self, idxs, device):
        return self._upload(self._host_batch(idxs), device)

    def _prefetched(self, index_lists, device, depth):
        """Batches of `index_lists` with decode + collate of batch t + 1 (and t + 2) running on a worker thread under step t,
        into pinned host memory; the consumer only issues the asynchronous upload.  Same batches in the same order as the
        synchronous path.  A consumer that stops early (steps= cap, exception) releases the worker."""
        import queue
        import threading
        q, stop = queue.Queue(maxsize=depth), threading.Event()
        pin = torch.cuda.is_available() and str(device) != "cpu"

        def put(item):
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.2)
                    return True
                except queue.Full:
                    continue
            return False

        def worker():
            try:
                for idxs in index_lists:
                    if not put(("batch", self._host_batch(idxs, pin=pin))):
                        return
                put(("end", None))
            except BaseException as e:  # surfaces in the training thread, at the batch it belongs to
                put(("error", e))

        th = threading.Thread(target=worker, name="vlm-batch-prefetch", daemon=True)
        th.start()
        try:
            while True:
                kind, item = q.get()
                if kind == "end":
                    return
                if kind == "error":
                    raise item
                yield self._upload(item, device, non_blocking=pin)
        finally:
            stop.set()

    def train_epoch(self, epoch, device, skip=0):
        """DistributedSampler(shuffle=True, seed=0).set_epoch(epoch) semantics: one permutation per epoch shared by the
        ranks, rank r takes elements r, r + W, ...; incomplete batches dropped (the reference's sampler PADS the
        permutation to a multiple of the world size instead of dropping its tail, and seeds with 0: here the permutation is
        seeded with cfg["seed"] + epoch -- a different but equally valid shuffle; what is pinned is the per-sample batch
        contract, tests/test_batch_contract_cpu.py).  `skip`: batches of this epoch a resumed run has already consumed (passed over
        without decoding).  Decode + collate of the next batches run on a worker thread (VLM_PREFETCH_BATCHES, default 2 ahead;
        0: synchronous)."""
        g = torch.Generator().manual_seed(int(self.cfg["seed"]) + epoch)
        perm = torch.randperm(len(self.data), generator=g).tolist()
        per_rank = len(perm) // self.world
        mine = perm[self.rank: per_rank * self.world: self.world]
        lists = [mine[lo: lo + self.B] for lo in range(skip * self.B, len(mine) - self.B + 1, self.B)]
        depth = int(os.environ.get("VLM_PREFETCH_BATCHES", "2"))
        if depth <= 0:
            for idxs in lists:
                yield self._collate(idxs, device)
        else:
            yield from self._prefetched(lists, device, depth)

    def eval_batches(self, device, all_ranks=False):
        """In order; batch j belongs to rank j % world unless `all_ranks` (retrieval preloads shard later by themselves)."""
        starts = list(range(0, len(self.data), self.B))
        for j, lo in enumerate(starts):
            if not all_ranks and j % self.world != self.rank:
                continue
            yield self._collate(range(lo, min(lo + self.B, len(self.data))), device)
