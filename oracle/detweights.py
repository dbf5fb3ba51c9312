"""Deterministic, torch-RNG-independent tensor generator (TEST INFRASTRUCTURE ONLY).

The golden fixtures under tests/golden/ were produced by overwriting every floating-point
entry of the *reference* model's state_dict with det_tensor(key, shape) and recording the
reference's outputs.  Tests regenerate the same weights from the key names alone, so no weight
blob has to be committed.  numpy's PCG64 stream is platform independent.

Nothing under vl-merging_amd/ may import this file.
"""
import zlib
import numpy as np


def _rng(key: str, salt: int = 0) -> np.random.Generator:
    return np.random.default_rng([zlib.crc32(key.encode()), salt])


def det_array(key: str, shape, salt: int = 0) -> np.ndarray:
    """fp32 values whose distribution depends on the parameter's role (decided by its name)."""
    shape = tuple(int(s) for s in shape)
    r = _rng(key, salt).standard_normal(shape, dtype=np.float32)
    leaf = key.split(".")[-1]
    if "gamma_" in key:
        out = 0.1 + 0.02 * r
    elif "relative_position_bias_table" in key:
        out = 0.3 * r
    elif "logit" in key and "scale" in key:
        out = np.float32(np.log(1 / 0.07)) + 0.05 * r
    elif leaf == "weight" and ("norm" in key.lower()) and len(shape) == 1:
        out = 1.0 + 0.1 * r
    else:
        out = 0.04 * r
    return np.asarray(out, dtype=np.float32).reshape(shape)


def det_state_dict(shapes: dict, salt: int = 0) -> dict:
    """shapes: {key: (shape, dtype_str)} -> {key: np.ndarray} for float entries only."""
    return {k: det_array(k, s, salt) for k, (s, dt) in shapes.items() if dt.startswith("float")}


def det_batch(B: int, image_size: int, T: int, vocab: int, seed: int = 1234, mlm_prob: float = 0.25):
    """Synthetic batch per SURVEY.md 8(d): image ~ U(-1,1); ids with [CLS]=101, [SEP]=102, pad 0."""
    g = np.random.default_rng(seed)
    image = g.uniform(-1.0, 1.0, size=(B, 3, image_size, image_size)).astype(np.float32)
    ids = np.zeros((B, T), dtype=np.int64)
    masks = np.zeros((B, T), dtype=np.int64)
    ids_mlm = np.zeros((B, T), dtype=np.int64)
    labels_mlm = np.full((B, T), -100, dtype=np.int64)
    for b in range(B):
        ln = int(g.integers(min(8, T), T + 1))
        ids[b, 0] = 101
        ids[b, 1:ln - 1] = g.integers(1000, vocab, size=ln - 2)
        ids[b, ln - 1] = 102
        masks[b, :ln] = 1
        ids_mlm[b] = ids[b]
        pick = g.uniform(size=ln - 2) < mlm_prob
        if not pick.any():
            pick[0] = True
        pos = np.nonzero(pick)[0] + 1
        labels_mlm[b, pos] = ids[b, pos]
        ids_mlm[b, pos] = 103
    return {
        "image": image,
        "text_ids": ids,
        "text_masks": masks,
        "text_labels": np.full((B, T), -100, dtype=np.int64),
        "text_ids_mlm": ids_mlm,
        "text_labels_mlm": labels_mlm,
    }


def det_gram(key: str, D: int, salt: int = 0) -> np.ndarray:
    """SPD fp64 gram X^T X with X ~ N(0,1) [D+64, D]."""
    X = _rng("gram:" + key, salt).standard_normal((D + 64, D))
    return X.T @ X


def det_keep(tag: str, site: int, B: int, keep_prob: float) -> np.ndarray:
    """Injected DropPath draw of one site of one pass: float32 [B] of 0/1 (1 = the sample's branch is kept).
    `tag` names the reference pass (mlm / img / txt / pos / negimg / negtxt), `site` counts the pass's DropPath calls
    (two per block evaluation, the vlffn re-run of the last layers continues the count)."""
    u = _rng("droppath:" + tag, site).uniform(size=64)[:B]
    return (u < keep_prob).astype(np.float32)


def det_dropout_mask(tag: str, b: int, T: int, D: int, keep_prob: float) -> np.ndarray:
    """Injected nn.Dropout mask of the text embeddings of sample b of pass `tag`: float32 [T, D] of 0/1."""
    u = _rng("dropout:" + tag, b).uniform(size=(T, D))
    return (u < keep_prob).astype(np.float32)
