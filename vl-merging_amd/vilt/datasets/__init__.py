from .arrow_dataset import ArrowDataset, square_transform, write_synthetic_shard, build_synthetic_tokenizer  # noqa: F401
