"""Host-side throughput of the batch contract (ArrowDataset.get_suite + collate) on synthetic shards: samples/s per core."""
import importlib
import os
import sys
import tempfile
import time

sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge

ge.import_package()
ds = importlib.import_module("vl_merging_amd.vilt.datasets")


def main():
    from transformers import DataCollatorForLanguageModeling
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 384
    d = tempfile.mkdtemp()
    ds.write_synthetic_shard(os.path.join(d, "coco_caption_karpathy_train.arrow"), 64, 5, image_hw=(480, 640), seed=1)
    tok = ds.build_synthetic_tokenizer(os.path.join(d, "vocab.txt"))
    dset = ds.ArrowDataset(d, ["square_transform"], size, ["coco_caption_karpathy_train"], text_column_name="caption",
                           max_text_len=40, tokenizer=tok)
    coll = DataCollatorForLanguageModeling(tok, mlm=True, mlm_probability=0.25)
    dset.collate([dset[i] for i in range(22)], coll)
    t0 = time.perf_counter()
    n = 0
    for lo in range(0, 220, 22):
        dset.collate([dset[i] for i in range(lo, lo + 22)], coll)
        n += 22
    dt = time.perf_counter() - t0
    print("ArrowDataset + collate, 640x480 PNG -> %d^2, batch 22: %.1f samples/s on one core" % (size, n / dt))


if __name__ == "__main__":
    main()
