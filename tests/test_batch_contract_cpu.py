"""Batch / collate contract: ArrowDataset over synthetic Arrow shards against what the
reference's BaseDataset produced on the same shards, tokenizer and seeds (tests/golden/batch_contract.npz, made by
tests/golden/make_golden.py batch).  The reference iterates a Python set of keys in collate(), so WHICH text key the
collator sees first depends on its PYTHONHASHSEED; the fixture was generated with "text" before "false_text_0", the
order ArrowDataset always uses."""
import importlib
import json
import os
import random

import numpy as np
import pytest
import torch

import __graft_entry__ as ge

HERE = os.path.dirname(os.path.abspath(__file__))
ge.import_package()
ds = importlib.import_module("vl_merging_amd.vilt.datasets")


def make(tmp_path, **kw):
    d = str(tmp_path)
    ds.write_synthetic_shard(os.path.join(d, "coco_caption_karpathy_train.arrow"), 6, 3, image_hw=(48, 64), seed=1)
    ds.write_synthetic_shard(os.path.join(d, "vg.arrow"), 4, 2, image_hw=(40, 40), seed=2)
    tok = ds.build_synthetic_tokenizer(os.path.join(d, "vocab.txt"))
    args = dict(text_column_name="caption", remove_duplicate=False, max_text_len=12, draw_false_image=1,
                draw_false_text=1, tokenizer=tok)
    args.update(kw)
    return ds.ArrowDataset(d, ["square_transform"], 32, ["coco_caption_karpathy_train", "vg", "missing_shard"], **args), tok


def test_batch_matches_reference(tmp_path):
    from transformers import DataCollatorForLanguageModeling
    gold = np.load(os.path.join(HERE, "golden", "batch_contract.npz"))
    dset, tok = make(tmp_path)
    assert len(dset) == int(gold["n_samples"]) == 6 * 3 + 4 * 2
    assert np.array_equal(np.array(dset.index_mapper), gold["index_mapper"])
    assert dset.table_names == json.loads(str(gold["table_names"]))
    random.seed(7)
    items = [dset.get_suite(i) for i in (0, 4, 5, 17, 19, 25)]
    torch.manual_seed(11)
    batch = dset.collate(items, DataCollatorForLanguageModeling(tok, mlm=True, mlm_probability=0.4))
    want_keys = {k.split("/", 1)[1] for k in gold.files if k.startswith(("batch/", "batch_img/"))}
    lists = json.loads(str(gold["batch_lists"]))
    assert set(batch.keys()) == want_keys | set(lists.keys())
    for k in gold.files:
        if k.startswith("batch/"):
            name = k[len("batch/"):]
            assert batch[name].dtype == torch.int64
            assert np.array_equal(batch[name].numpy(), gold[k]), name
        elif k.startswith("batch_img/"):
            name = k[len("batch_img/"):]
            assert isinstance(batch[name], list) and len(batch[name]) == 1  # the one-element list of vilt_module.py:1092
            assert np.array_equal(batch[name][0].numpy(), gold[k]), name  # same PIL resize, same float ops
    for k, v in lists.items():
        assert batch[k] == v, k
    assert batch["image"][0].shape == (6, 3, 32, 32) and batch["text_ids"].shape == (6, 12)
    assert int(batch["text_labels"].min()) == int(batch["text_labels"].max()) == -100
    # special ids as bert-base-uncased: [CLS] first, [SEP] last real token, zero padding
    ids, masks = batch["text_ids"], batch["text_masks"]
    assert bool((ids[:, 0] == 101).all())
    last = masks.sum(1) - 1
    assert bool((ids[torch.arange(6), last] == 102).all()) and bool((ids * (1 - masks) == 0).all())


def test_square_transform_matches_numpy_restatement():
    from PIL import Image
    g = np.random.default_rng(3)
    img = Image.fromarray(g.integers(0, 256, size=(37, 53, 3), dtype=np.uint8), "RGB")
    t = ds.square_transform(24)(img)
    a = np.asarray(img.resize((24, 24), Image.BICUBIC), dtype=np.float32).transpose(2, 0, 1)
    want = ((a / np.float32(255)) - np.float32(0.5)) / np.float32(0.5)
    assert t.shape == (3, 24, 24) and t.dtype == torch.float32
    assert np.array_equal(t.numpy(), want)
    assert float(t.min()) >= -1.0 and float(t.max()) <= 1.0


def test_dedup_image_only_and_unknown_transform(tmp_path):
    dset, tok = make(tmp_path, remove_duplicate=True)
    assert sorted(dset.corpus) == sorted(set(dset.corpus)) or len(dset.corpus) <= 26  # duplicates dropped, order stable
    img_only, _ = make(tmp_path, image_only=True, draw_false_text=0)
    assert len(img_only) == 10 and img_only.index_mapper[3] == (3, None)
    s = img_only[3]
    assert "text" not in s and s["cap_index"] is None
    with pytest.raises(NotImplementedError):
        ds.ArrowDataset(str(tmp_path), ["square_transform_randaug"], 32, ["vg"], text_column_name="caption", tokenizer=tok)


def test_shard_schema_and_model_consumes_batch_keys(tmp_path):
    import pyarrow as pa
    make(tmp_path)
    t = pa.ipc.open_file(pa.memory_map(os.path.join(str(tmp_path), "vg.arrow"), "r")).read_all()
    assert t.schema.names == ["image", "caption", "image_id", "split"]
    assert t.schema.field("image").type == pa.binary() and t.schema.field("caption").type == pa.list_(pa.string())


def test_prefetch_thread_yields_the_same_batches_in_order(pkg, tmp_path, monkeypatch):
    """ArrowBatches.train_epoch with the decode + collate of the next batches on a worker thread (default) against the
    synchronous path: same samples in the same batches, a resumed epoch skips the same ones, an abandoned generator releases its
    worker, a decoding error surfaces in the consumer."""
    import importlib
    import threading
    import time
    ds = importlib.import_module("vl_merging_amd.vilt.datasets")
    dm = importlib.import_module("vl_merging_amd.vilt.datamodules")
    cfgmod = importlib.import_module("vl_merging_amd.vilt.config")
    d = str(tmp_path / "data")
    ds.write_synthetic_shard(os.path.join(d, "synthetic_0.arrow"), 9, 2, image_hw=(40, 48), seed=1)
    ds.build_synthetic_tokenizer(os.path.join(d, "vocab.txt"))
    cfg = cfgmod.parse_cli(["task_test_vit_tiny_mlm_itm_ifm_square_randaug_base_vl", "ufo", "vocab_size=2048", "per_gpu_batchsize=2",
                            "data_root=" + d])
    data = dm.ArrowBatches(cfg, "train")

    def order(depth, skip=0, stop_after=None):
        monkeypatch.setenv("VLM_PREFETCH_BATCHES", str(depth))
        out = []
        for b in data.train_epoch(0, "cpu", skip=skip):
            out.append(list(b["raw_index"]))
            if stop_after and len(out) == stop_after:
                break
        return out

    sync = order(0)
    assert len(sync) == data.steps_per_epoch() and len(sync) >= 4
    assert order(2) == sync and order(1) == sync
    assert order(2, skip=2) == sync[2:]
    n0 = threading.active_count()
    assert order(2, stop_after=1) == sync[:1]
    for _ in range(50):  # the abandoned generator's worker ends by itself
        if threading.active_count() <= n0:
            break
        time.sleep(0.1)
    assert threading.active_count() <= n0
    monkeypatch.setattr(data.data, "collate", lambda *a, **k: (_ for _ in ()).throw(RuntimeError("bad shard")))
    import pytest as _pt
    with _pt.raises(RuntimeError, match="bad shard"):
        order(2)
