"""Error of the forward attention kernel against the fp32 restatement: mean / max |err| of out, max |err| of lse (GPU)."""
import importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
g.import_package()
sys.path.insert(0, os.path.join(os.path.dirname(g.__file__), "tests"))
import test_attention_gpu as T
ops = importlib.import_module("vl_merging_amd.ops"); L = importlib.import_module("vl_merging_amd._lib")
for ci in (0, 4):
    for sep in (0, 1):
        c = T.build_case(seed=ci * 10 + sep, with_bias=True, **T.CASES[ci])
        seq = ops.Seq(c["B"], c["n0"], c["n1"])
        out = torch.full((seq.rows, c["D"]), float("nan"), device="cuda", dtype=torch.bfloat16)
        lse = torch.empty(c["H"], seq.rows, device="cuda")
        ops.attention_fwd(c["qkv"], out, lse, seq, c["H"], bias_t=c["table"].t().contiguous(), head_row0=c["H"], rel_index=c["idx"] * 4,
                          rel_index_t=T.make_idx_t(c), keep0=c["keep0"], mode=L.ATTN_SEPARATE if sep else L.ATTN_JOINT)
        torch.cuda.synchronize()
        ref, s, _ = T.reference(c, 1, sep)
        refb = ref.to(torch.bfloat16).float()  # what a perfect kernel would store
        err = (out.float() - ref).abs()
        ref_lse = T.from_seq(torch.logsumexp(s, -1).permute(0, 2, 1), c).t() * 1.4426950408889634
        print("case %d sep %d: mean|err| %.3e  max|err| %.3e  rms %.3e  (bf16 rounding of the exact result alone: mean %.3e)  lse max|err| %.3e mean %.3e"
              % (ci, sep, float(err.mean()), float(err.max()), float((err ** 2).mean().sqrt()), float((refb - ref).abs().mean()),
                 float((lse - ref_lse).abs().max()), float((lse - ref_lse).mean())))
