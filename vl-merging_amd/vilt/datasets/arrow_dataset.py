"""The step BEFORE the hot path: Arrow shards -> samples -> the batch dict the model
consumes.  Host-side only.

Mirrors the reference's BaseDataset contract (src/vilt/datasets/base_dataset.py:12-253): one Arrow IPC file per shard
with columns `image` (encoded bytes), a list<string> text column, `image_id`, `split`
(src/vilt/utils/write_coco_karpathy.py:53-63); a sample is addressed by (image row, caption number); a batch is
    image            [Tensor[B,3,H,W]]            (a ONE-element list, consumed as batch["image"][0], vilt_module.py:1092)
    text             list[str]
    text_ids / text_masks / text_labels(=-100) / text_ids_mlm / text_labels_mlm    int64 [B, max_text_len]
    img_index / cap_index / raw_index / replica                                     python lists
plus false_image_<i> / false_text_<i>* when negatives are drawn.
Differences, all deliberate: only the non-augmenting `square_transform` is implemented (RandAugment is a training
recipe, not the batch contract); duplicate captions are removed in first-seen order (the reference's list(set(...))
order changes with PYTHONHASHSEED); masked-image-modelling outputs are out of scope (dVAE path).
"""
import io
import os
import random

import numpy as np
import pyarrow as pa
import torch
from PIL import Image


def square_transform(size=224):
    """Resize((size,size), BICUBIC) -> ToTensor -> Normalize(0.5, 0.5): transforms/square_transform.py:12-19 and
    transforms/utils.py:48-50, on PIL + torch (torchvision's PIL Resize is Image.resize; ToTensor is /255, CHW)."""
    mean = torch.full((3, 1, 1), 0.5)
    std = torch.full((3, 1, 1), 0.5)

    def apply(img):
        img = img.resize((size, size), Image.BICUBIC)
        t = torch.from_numpy(np.asarray(img, dtype=np.uint8).transpose(2, 0, 1).copy()).to(torch.float32).div(255)
        return (t - mean) / std

    return apply


_TRANSFORMS = {"square_transform": square_transform}


class ArrowDataset(torch.utils.data.Dataset):
    def __init__(self, data_dir, transform_keys, image_size, names, text_column_name="", remove_duplicate=True,
                 max_text_len=40, max_vl_text_len=None, draw_false_image=0, draw_false_text=0, image_only=False,
                 tokenizer=None):
        if not transform_keys:
            raise ValueError("at least one transform key")
        unknown = [k for k in transform_keys if k not in _TRANSFORMS]
        if unknown:
            raise NotImplementedError("transforms %s: only %s are part of the batch contract here" % (unknown, sorted(_TRANSFORMS)))
        self.transforms = [_TRANSFORMS[k](size=image_size) for k in transform_keys]
        self.names, self.data_dir = list(names), data_dir
        self.text_column_name = text_column_name
        self.max_text_len, self.max_vl_text_len = max_text_len, max_vl_text_len
        self.draw_false_image, self.draw_false_text = draw_false_image, draw_false_text
        self.image_only = image_only
        self.tokenizer = tokenizer
        tables, self.table_names = [], []
        for name in self.names:
            path = os.path.join(data_dir, name + ".arrow")
            if os.path.isfile(path):
                t = pa.ipc.open_file(pa.memory_map(path, "r")).read_all()
                tables.append(t)
                self.table_names += [name] * len(t)
        self.table = pa.concat_tables(tables, promote_options="default") if tables else None
        self.all_texts = []
        if self.table is not None and text_column_name:
            self.all_texts = [list(t) for t in self.table[text_column_name].to_pylist()]
            if remove_duplicate:
                self.all_texts = [list(dict.fromkeys(t)) for t in self.all_texts]
        # sample id -> (image row, caption number or None)
        if text_column_name and not image_only:
            self.index_mapper = [(i, j) for i, texts in enumerate(self.all_texts) for j in range(len(texts))]
        else:
            self.index_mapper = [(i, None) for i in range(len(self.table) if self.table is not None else 0)]

    @property
    def corpus(self):
        return [t for texts in self.all_texts for t in texts]

    def __len__(self):
        return len(self.index_mapper)

    # ---- single items ---------------------------------------------------------------------------------------------
    def get_raw_image(self, index, image_key="image"):
        row, _ = self.index_mapper[index]
        return Image.open(io.BytesIO(self.table[image_key][row].as_py())).convert("RGB")

    def get_image(self, index, image_key="image"):
        img = self.get_raw_image(index, image_key)
        row, cap = self.index_mapper[index]
        return {"image": [tr(img) for tr in self.transforms], "img_index": row, "cap_index": cap, "raw_index": index}

    def get_false_image(self, rep, image_key="image"):
        img = self.get_raw_image(random.randint(0, len(self.index_mapper) - 1), image_key)
        return {"false_image_%d" % rep: [tr(img) for tr in self.transforms]}

    def _encode(self, text, pad):
        kw = dict(truncation=True, return_special_tokens_mask=True,
                  max_length=self.max_text_len if self.max_vl_text_len is None else self.max_vl_text_len)
        if pad:
            kw["padding"] = "max_length"
        return self.tokenizer(text, **kw)

    def get_text(self, raw_index):
        row, cap = self.index_mapper[raw_index]
        text = self.all_texts[row][cap]
        return {"text": (text, self._encode(text, pad=True)), "img_index": row, "cap_index": cap, "raw_index": raw_index}

    def get_false_text(self, rep):
        row, cap = self.index_mapper[random.randint(0, len(self.index_mapper) - 1)]
        text = self.all_texts[row][cap]
        return {"false_text_%d" % rep: (text, self._encode(text, pad=False))}

    def get_suite(self, index):
        while True:
            try:
                ret = dict(self.get_image(index))
                if not self.image_only:
                    txt = self.get_text(index)
                    ret["replica"] = bool(txt["cap_index"] > 0)
                    ret.update(txt)
                for i in range(self.draw_false_image):
                    ret.update(self.get_false_image(i))
                for i in range(self.draw_false_text):
                    ret.update(self.get_false_text(i))
                return ret
            except Exception as e:  # unreadable sample: draw another one, as the reference does (:198-201)
                print("Error while reading sample %d of %s -> %s" % (index, self.names[0] if self.names else "?", e))
                index = random.randint(0, len(self.index_mapper) - 1)

    def __getitem__(self, index):
        return self.get_suite(index)

    # ---- batch contract -------------------------------------------------------------------------------------------
    def collate(self, batch, mlm_collator):
        """List of get_suite() dicts -> the batch dict (base_dataset.py:204-253).  ONE call of `mlm_collator` over the
        encodings of every text key (key-major), whose rows are then handed back to their keys."""
        n = len(batch)
        keys = []
        for b in batch:
            for k in b:
                if k not in keys:
                    keys.append(k)
        out = {k: [b.get(k) for b in batch] for k in keys}
        for k in [k for k in keys if "image" in k]:
            out[k] = [torch.stack([views[0] for views in out[k]], dim=0)]
        txt_keys = [k for k in keys if "text" in k]
        if txt_keys:
            flat = [pair[1] for k in txt_keys for pair in out[k]]
            mlm = mlm_collator(flat)
            for i, k in enumerate(txt_keys):
                pairs = out[k]
                ids_mlm = mlm["input_ids"][n * i: n * (i + 1)]
                labels_mlm = mlm["labels"][n * i: n * (i + 1)]
                ids = torch.zeros_like(ids_mlm)
                masks = torch.zeros_like(ids_mlm)
                for r, (_, enc) in enumerate(pairs):
                    a = torch.tensor(enc["input_ids"])
                    m = torch.tensor(enc["attention_mask"])
                    ids[r, : len(a)] = a
                    masks[r, : len(m)] = m
                out[k] = [p[0] for p in pairs]
                out[k + "_ids"] = ids
                out[k + "_labels"] = torch.full_like(ids, -100)
                out[k + "_ids_mlm"] = ids_mlm
                out[k + "_labels_mlm"] = labels_mlm
                out[k + "_masks"] = masks
        return out


# ---- synthetic shards (BASELINE configs[0]: "synthetic arrow shards") ----------------------------------------------------
_WORDS = ("a an the man woman dog cat child bird horse rides holds eats watches stands sits runs red blue green small "
          "large old young wooden metal on under near behind table street beach field kitchen park with two three and "
          "ball bike plate bench kite umbrella phone cake train bus").split()


def write_synthetic_shard(path, n_images, captions_per_image=5, image_hw=(48, 64), seed=0, split="train"):
    """One Arrow IPC file in the reference's schema (write_coco_karpathy.py:53-63): PNG-encoded noise images and
    captions drawn from a fixed word list; deterministic in `seed`."""
    g = np.random.default_rng(seed)
    images, captions, ids, splits = [], [], [], []
    for i in range(n_images):
        arr = g.integers(0, 256, size=(image_hw[0], image_hw[1], 3), dtype=np.uint8)
        buf = io.BytesIO()
        Image.fromarray(arr, "RGB").save(buf, format="PNG")
        images.append(buf.getvalue())
        caps = []
        for _ in range(captions_per_image):
            ln = int(g.integers(4, 12))
            caps.append(" ".join(_WORDS[int(j)] for j in g.integers(0, len(_WORDS), size=ln)))
        captions.append(caps)
        ids.append("img_%06d" % i)
        splits.append(split)
    table = pa.table({"image": pa.array(images, type=pa.binary()), "caption": pa.array(captions, type=pa.list_(pa.string())),
                      "image_id": pa.array(ids), "split": pa.array(splits)})
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with pa.OSFile(path, "wb") as sink:
        with pa.RecordBatchFileWriter(sink, table.schema) as w:
            w.write_table(table)
    return table


def build_synthetic_tokenizer(vocab_path, words=_WORDS):
    """A BertTokenizerFast over a synthetic vocabulary with bert-base-uncased's special-token ids ([PAD]=0, [UNK]=100,
    [CLS]=101, [SEP]=102, [MASK]=103).  The reference downloads bert-base-uncased (datamodule_base.py:13-24); there is
    no network here, and the batch contract only depends on the special ids and the padding/truncation rules."""
    from tokenizers import Tokenizer, models, normalizers, pre_tokenizers, processors
    from transformers import PreTrainedTokenizerFast
    vocab = ["[PAD]"] + ["[unused%d]" % i for i in range(1, 100)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    vocab += ["[unused%d]" % i for i in range(100, 996)]  # words start at 1000, like bert-base-uncased's
    vocab += sorted(set(w.lower() for w in words))
    os.makedirs(os.path.dirname(os.path.abspath(vocab_path)), exist_ok=True)
    with open(vocab_path, "w") as f:
        f.write("\n".join(vocab) + "\n")
    tok = Tokenizer(models.WordPiece({w: i for i, w in enumerate(vocab)}, unk_token="[UNK]"))
    tok.normalizer = normalizers.BertNormalizer(lowercase=True)
    tok.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
    tok.post_processor = processors.TemplateProcessing(single="[CLS] $A [SEP]", pair="[CLS] $A [SEP] $B:1 [SEP]:1",
                                                       special_tokens=[("[CLS]", 101), ("[SEP]", 102)])
    return PreTrainedTokenizerFast(tokenizer_object=tok, unk_token="[UNK]", pad_token="[PAD]", cls_token="[CLS]",
                                   sep_token="[SEP]", mask_token="[MASK]")
