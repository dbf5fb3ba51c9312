"""N > 1 path on CPU: two gloo processes drive the flat-gradient reducer (bucket plan, use-count learning,
per-block launches, tail buckets) exactly as bench.py drives it over RCCL."""
import importlib
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Tiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.transformer = nn.Module()
        self.transformer.blocks = nn.ModuleList([nn.Linear(8, 8) for _ in range(3)])
        self.text_embeddings = nn.Embedding(10, 8)
        self.head = nn.Linear(8, 2)
        self._flat = None
        self._grad_hook = None


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        import __graft_entry__ as ge
        ge.import_package()
        engine = importlib.import_module("vl_merging_amd.engine")
        ddp = importlib.import_module("vl_merging_amd.ddp")
        vu = importlib.import_module("vl_merging_amd.vilt.modules.vilt_utils")
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.manual_seed(0)
        m = Tiny()
        m._flat = engine.FlatParams(m, order_key=vu.flat_order_key)
        red = ddp.FlatGradReducer(m)
        assert red.world == 2 and abs(red.grad_scale - 0.5) < 1e-12
        assert sorted(red.block_slices) == [0, 1, 2] and len(red.tail_slices) == 1 and len(red.late_slices) == 1
        plan = red.bucket_plan()  # what bench.py reports as grad_comm.buckets: issue order, MB on the wire
        assert plan["count"] == 5 and [b["what"] for b in plan["in_issue_order"]] == ["block 2", "heads", "block 1", "block 0", "embeddings"]
        assert abs(plan["total_MB"] - sum(hi - lo for lo, hi in red.buckets()) * 4 / 1e6) < 0.06
        for step in range(3):
            red.begin_step()
            for p in m.parameters():
                p.grad.fill_(float(rank + 1 + step))
            # backward order: block 2 is used twice per step (two passes), blocks 1, 0 once
            for layer in (2, 2, 1, 0):
                m._grad_hook(layer)
            red.defer_tail = step == 2  # last step: the embeddings' all-reduce stays in flight until wait_tail()
            red.finish_backward()
            if red.defer_tail:
                lo, hi = red.tail_range()
                assert (lo, hi) == red.tail_slices[0] and len(red._tail_handles) == 1
                red.wait_tail()
                assert red._tail_handles == []
            want = sum(r + 1 + step for r in range(world))
            for n, p in m.named_parameters():
                assert torch.all(p.grad == want), (step, n, p.grad.flatten()[:3], want)
            assert red.expected == {0: 1, 1: 1, 2: 2}
        # ---- sharded mode (ddp_sharded): reduce-scatter semantics, the own chunks partition every bucket, parameter
        # all-gather from the owners
        m2 = Tiny()
        m2._flat = engine.FlatParams(m2, order_key=vu.flat_order_key)
        red2 = ddp.FlatGradReducer(m2, sharded=True)
        own = red2.own_ranges()
        assert sum(hi - lo for lo, hi in own) * world == sum(hi - lo for lo, hi in red2.buckets())
        for step in range(2):
            red2.begin_step()
            for p in m2.parameters():
                p.grad.fill_(float(rank + 1 + step))
            for layer in (2, 2, 1, 0):
                m2._grad_hook(layer)
            red2.finish_backward()
            want = sum(r + 1 + step for r in range(world))
            real = torch.zeros(m2._flat.numel, dtype=torch.bool)   # elements that belong to a parameter (not padding)
            for n in m2._flat.names:
                o, k = m2._flat.offsets[n]
                real[o:o + k] = True
            for lo, hi in own:
                assert torch.all(m2._flat.flat_g[lo:hi][real[lo:hi]] == want)
        for (lo, hi), (clo, chi) in zip(red2.buckets(), own):
            m2._flat.flat_p[lo:hi] = -1.0
            m2._flat.flat_p[clo:chi] = float(rank + 1)   # "the optimizer updated the own chunk"
        red2.gather_params()
        for lo, hi in red2.buckets():
            c = (hi - lo) // world
            for r in range(world):
                assert torch.all(m2._flat.flat_p[lo + r * c: lo + (r + 1) * c] == float(r + 1))
        # ---- bf16 wire buffer: the bucket is cast, reduced in bf16 and widened back (comm_dtype option) -------------------
        m3 = Tiny()
        m3._flat = engine.FlatParams(m3, order_key=vu.flat_order_key)
        red3 = ddp.FlatGradReducer(m3, comm_dtype=torch.bfloat16, collective="rs_ag")  # rs_ag needs nccl: falls back to all-reduce here
        assert red3.comm_dtype is torch.bfloat16
        vals = (1.2345, 2.3456)
        for step in range(2):
            red3.begin_step()
            for p in m3.parameters():
                p.grad.fill_(vals[rank])
            for layer in (2, 2, 1, 0):
                m3._grad_hook(layer)
            red3.finish_backward()
            want16 = (torch.tensor(vals[0]).to(torch.bfloat16) + torch.tensor(vals[1]).to(torch.bfloat16)).float()
            for n, p in m3.named_parameters():
                assert p.grad.dtype == torch.float32 and torch.all(p.grad == want16), (n, p.grad.flatten()[:2], want16)
        assert red3._wire is not None and red3._wire.dtype == torch.bfloat16
        # a changed use count must be loud
        red.begin_step()
        m._grad_hook(2)
        try:
            red.finish_backward()
            ok = False
        except RuntimeError:
            ok = True
        assert ok
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL " + repr(e) + traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_flat_grad_reducer_two_ranks_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(r[1] == "ok" for r in res), res


def test_flat_params_views_and_order():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.import_package()
    engine = importlib.import_module("vl_merging_amd.engine")
    vu = importlib.import_module("vl_merging_amd.vilt.modules.vilt_utils")
    m = Tiny()
    w0 = m.transformer.blocks[1].weight.detach().clone()
    flat = engine.FlatParams(m, order_key=vu.flat_order_key)
    assert torch.equal(m.transformer.blocks[1].weight.detach(), w0)
    lo, hi = flat.slice_of(["transformer.blocks.1.weight", "transformer.blocks.1.bias"])
    assert hi - lo == 128  # 64 + 8 elements, each padded to 64
    m.transformer.blocks[1].weight.data.fill_(3.0)
    o, k = flat.offsets["transformer.blocks.1.weight"]
    assert torch.all(flat.flat_p[o:o + k] == 3.0)
    # embeddings first, blocks in order, heads last
    names = flat.names
    assert names[0].startswith("text_embeddings") and names[-1].startswith("head")
    with pytest.raises(Exception):
        flat.refresh_shadow()  # bf16 shadows exist on the GPU only
