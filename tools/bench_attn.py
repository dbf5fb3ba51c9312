"""Micro-benchmark of the attention kernels at the bench shape (B=22, T=40, I=577, H=12)."""
import importlib
import sys

import torch

sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge

ge.import_package()
ops = importlib.import_module("vl_merging_amd.ops")
L = importlib.import_module("vl_merging_amd._lib")
vm = importlib.import_module("vl_merging_amd.vilt.modules.vilt_module")


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    B, T, I, H = int(sys.argv[1]) if len(sys.argv) > 1 else 22, 40, 577, 12
    D = H * 64
    idx, nrel, _, allrel = vm.build_relative_position_indices((24, 24), 40, 196, 40)
    for name, n0, n1, index in (("joint", T, I, idx["text_imag_relative_position_index"]),
                                ("image", 0, I, idx["relative_position_index"])):
        m, mt = vm._index16(index.cuda(), n0)
        seq = ops.Seq(B, n0, n1)
        rows = seq.rows
        qkv = torch.randn(rows, 3 * D, device="cuda").to(torch.bfloat16)
        dout = torch.randn(rows, D, device="cuda").to(torch.bfloat16)
        out = torch.empty(rows, D, device="cuda", dtype=torch.bfloat16)
        lse = torch.empty(H, rows, device="cuda")
        dqkv = torch.empty(rows, 3 * D, device="cuda", dtype=torch.bfloat16)
        table = torch.randn(allrel, 144, device="cuda") * 0.3
        bias_t = table.t().contiguous()
        dbias = torch.zeros_like(bias_t)
        delta = torch.empty(H, rows, device="cuda")
        flops = 4.0 * B * H * 64 * (n0 + n1) ** 2
        for mode, mname in ((L.ATTN_JOINT, "joint"), (L.ATTN_SEPARATE, "sep")):
            if n0 == 0 and mode == L.ATTN_SEPARATE:
                continue
            dense = ops.bias_dense(bias_t, m, seq, mode)
            for wb in ("dense", False):
                kw = dict(bias_t=bias_t if wb else None, head_row0=12, rel_index=m if wb else None,
                          rel_index_t=mt if wb else None, mode=mode, bias_dense=dense if wb == "dense" else None)
                f = timeit(lambda: ops.attention_fwd(qkv, out, lse, seq, H, **kw))
                b = timeit(lambda: ops.attention_bwd(qkv, out, dout, lse, dqkv, seq, H,
                                                     dbias_t=dbias if wb else None, delta_ws=delta, **kw))
                fl = flops if mode == L.ATTN_JOINT else 4.0 * B * H * 64 * (n0 * n0 + n1 * n1)
                print("%s/%s bias=%s: fwd %.1f us (%.0f TF)  bwd %.1f us (%.0f TF of 2.5x fwd flops)" %
                      (name, mname, wb, f, fl / f / 1e6, b, 2.5 * fl / b / 1e6))


if __name__ == "__main__":
    main()
