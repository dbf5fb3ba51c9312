"""Retrieval recall arithmetic (compute_irtr_recall) against the reference's own result on the
reference's own features (tests/golden/irtr_recall_tiny.npz), for the oracle restatement and the host function; and the
rank-sharded sweep + ragged all-gather over gloo (world size 2)."""
import importlib
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import __graft_entry__ as ge
from oracle import vlmo_ref as R

HERE = os.path.dirname(os.path.abspath(__file__))
ge.import_package()
obj = importlib.import_module("vl_merging_amd.vilt.modules.objectives")
G = np.load(os.path.join(HERE, "golden", "irtr_recall_tiny.npz"))


@pytest.mark.parametrize("arch", ["ufo", "all_moe"])
def test_recall_arithmetic_matches_reference(arch):
    img, txt = G[arch + "/img_cls_feats"], G[arch + "/txt_cls_feats"]
    iids, tiids, want = G[arch + "/iids"], G[arch + "/tiids"], G[arch + "/recalls"]
    got_oracle = np.array(R.recall_ref(img, txt, iids, tiids))
    assert np.array_equal(got_oracle.astype(np.float64), want), (got_oracle, want)
    scores = torch.from_numpy(img) @ torch.from_numpy(txt).t()
    got = np.array([float(x) for x in obj.recall_at_k(scores, iids, tiids)])
    assert np.array_equal(got, want), (got, want)


def test_recall_known_answers():
    # 3 images, 2 captions each; caption 2j, 2j+1 belong to image j.  Scores rigged: image 0 ranks its captions first,
    # image 1 ranks them 2nd/3rd, image 2 last.
    scores = torch.tensor([[9., 8., 1., 0., 2., 3.],
                           [7., 1., 6., 5., 0., 2.],
                           [5., 6., 7., 8., 1., 0.]])
    iids, tiids = [0, 1, 2], [0, 0, 1, 1, 2, 2]
    ir1, ir5, ir10, tr1, tr5, tr10 = [float(x) for x in obj.recall_at_k(scores, iids, tiids, ks=(1, 2, 3))]
    assert tr1 == pytest.approx(1 / 3) and tr5 == pytest.approx(2 / 3) and tr10 == pytest.approx(2 / 3)
    # per caption best image: c0->img0 (hit), c1->img0 (hit), c2->img2 (miss), c3->img2 (miss), c4->img0 (miss), c5->img0 (miss)
    assert ir1 == pytest.approx(2 / 6)
    assert np.allclose(R.recall_ref(np.eye(3), np.eye(3), [0, 1, 2], [0, 1, 2], ks=(1,)), (1.0, 1.0))


class _Stub:
    """Deterministic encoder standing in for the model: features are functions of the inputs only."""

    def __init__(self):
        self.device = torch.device("cpu")
        self.hparams = type("H", (), {"config": {"hidden_size": 4}})()

    def infer_text_ft(self, b):
        x = b["text_ids"].float()
        return {"cls_feats": torch.stack([x.sum(1), x[:, 0], x[:, 1] * 2, x.mean(1)], 1)}

    def infer_image_ft(self, b):
        x = b["image"][0].flatten(1)
        return {"cls_feats": torch.stack([x.sum(1), x[:, 0], x[:, 1], x.max(1)[0]], 1)}


def _batches():
    g = torch.Generator().manual_seed(5)
    texts, images = [], []
    tid = 0
    for n in (5, 4, 6):  # ragged batch sizes, odd number of batches
        texts.append({"text_ids": torch.randint(0, 9, (n, 6), generator=g), "text_masks": torch.ones(n, 6),
                      "text_labels": torch.zeros(n, 6), "img_index": list(range(tid, tid + n))})
        tid += n
    iid = 0
    for n in (6, 5):
        images.append({"image": [torch.rand(n, 3, 2, 2, generator=g)], "img_index": list(range(iid, iid + n)),
                       "text_masks": torch.ones(1, 6)})
        iid += n
    return texts, images


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    texts, images = _batches()
    out = obj.compute_irtr_recall(_Stub(), texts, images)
    q.put((rank, [float(x) for x in out[:6]], out[6]["scores"].numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_sweep_matches_single_process():
    texts, images = _batches()
    single = obj.compute_irtr_recall(_Stub(), texts, images)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, recalls, scores in res:
        assert np.array_equal(scores, single[6]["scores"].numpy()), rank
        assert recalls == [float(x) for x in single[:6]]
