"""Generates attention_dq2_body.inc: the hand-placed instruction stream of attn_bwd_dq2_kernel (attention_bwd_dq2.hip).

    python vl-merging_amd/csrc/gen/attn_dq2_gen.py            # rewrite the .inc
    python vl-merging_amd/csrc/gen/attn_dq2_gen.py --check    # exit 1 if the committed .inc is stale

One wave = 32 query positions of one (sample, head); keys stream in 32-key BLOCKS through a ring of four 8-KiB stages
[K image | V image]; ONE K image serves the row reads of the score chain and the transposed reads of the dQ products (the swizzle
of attention_bwd_dkvb.h: chunk ^ f(row), conflict-free for the LDS-DMA writes, ds_read_b128 and ds_read_b64_tr_b16).  Per block n (a UNIT) a wave runs two phases, so that the two waves of a
SIMD (two workgroups per CU) can sit in opposite phases -- one in its MFMA chain while the other does its exponentials:
  M(n): 16 MFMAs   E  = bias (2 selection MFMAs, f16) + K (c1 Q)^T (4) + statistics step (-lse in three bf16 terms, key mask)
                   dP = V dO^T (4) + statistics step (-delta in three bf16 terms)
                   dQ^T += K^T(n-1) dS^T(n-1) (4)                      || LDS-DMA of block n + 2, bias operands of block n + 1
  barrier          block n + 1 has landed for every wave; block n - 1's stage is free
  V(n): vector     dS = exp2(E) * dP -> bf16 (16 v_exp, 16 v_mul, 8 v_cvt_pk) || fragment reads: K^T(n), K / V rows of block n + 1
The loop body is four units (ring positions 0..3), so every LDS address is a per-lane base + an immediate.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from asmgen import Op, mfma, rr, regs, interleave, place_waits, pad_hazards, emit  # noqa: E402

NSTAGE = 4
STAGE = 8192                  # per stage: K | V, 4 KiB each (32 keys x 128 B)

# ---- register map (must match attention_bwd_dq2.hip); v0..v31 are the compiler's ---------------------------------------
V_ADDR = 32      # v32..35 row-fragment addresses (stage 0, K image; V = +4096), v36..39 transposed-fragment addresses [db][lo / hi rows],
                 # v40 mask word (tile 0), v41 "no mask" word, v42 DMA voffset, v43 block row of the DMA piece, v44 bias voffset,
                 # v45 mask word (tile 1), v46 = 0xFFFFFFF0, v47 current mask word address
V_SEL = 48       # sel0, sel1
V_Q = 56         # c1 Q fragments (4 x 4)
V_DO = 72        # dO fragments (4 x 4)
V_E = 88         # E (16)
V_DP = 104       # dP (16)
V_DS = 120       # dS as bf16 (8)
V_KF = 128       # K row fragments (4 x 4)
V_VF = 144       # V row fragments (4 x 4)
V_KT = 160       # K^T fragments, (s2, db) -> 160 + 4 (2 s2 + db)
V_BW = 176       # bias operands of the current block (2 x 4)
V_KONE = 184     # key side of both statistics steps: dword 0 = (1, 1), dword 1 = (mask, 1) in the lower half-wave
V_QL = 188       # query side for E:  (-lse hi, -lse mid), (1, -lse lo)
V_QD = 192       # query side for dP: (-delta hi, -delta mid), (0, -delta lo)
V_T = 196        # temporaries
V_DOFF = 200     # current (range-masked) DMA voffset
V_KMC = 47
A_O = 0          # dQ^T accumulators a[16 db : +15]

S_RKV, S_RB = 40, 44
S_SOFF = 48      # s48 K, s49 V: byte offsets of the DMA block's first row
S_NU = 50        # number of units (32-key blocks, even)
S_REM = 51       # valid rows from the DMA block's first position on (may be <= 0)
S_W1K = 52       # LDS base + wave * 1024
S_STEP = 53      # 32 * ld * 2     (s54, s55: spare inputs)
S_U = 57         # unit counter
S_B1, S_B2 = 58, 59  # bias scalar offsets of the pass's second tile and of the next pass's first
S_BOFF = 60      # bias scalar offset of the pass's first 64-key tile (tile * 4096)
S_TMP = 61
S_LO = 62        # s[62:63] = lower half-wave

PRIO = os.environ.get("VLM_GEN_PRIO", "slot")
MF_BF = "v_mfma_f32_32x32x16_bf16"
MF_F16 = "v_mfma_f32_32x32x16_f16"

E = rr("v", V_E, 16)
DP = rr("v", V_DP, 16)
KONE = rr("v", V_KONE, 4)
SEL = [rr("v", V_SEL, 4), rr("v", V_SEL + 4, 4)]


def Er(i): return "v%d" % (V_E + i)
def Dr(i): return "v%d" % (V_DP + i)
def DS(half): return rr("v", V_DS + 4 * half, 4)
def DSr(d): return "v%d" % (V_DS + d)
def KF(ss): return rr("v", V_KF + 4 * ss, 4)
def VF(ss): return rr("v", V_VF + 4 * ss, 4)
def KT(s2, db): return rr("v", V_KT + 4 * (2 * s2 + db), 4)
def BW(j): return rr("v", V_BW + 4 * j, 4)
def QF(ss): return rr("v", V_Q + 4 * ss, 4)
def DOF(ss): return rr("v", V_DO + 4 * ss, 4)
def O(db): return rr("a", A_O + 16 * db, 16)


def valu(text, reads, writes, **kw):
    return Op(text, "valu", reads, writes, **kw)


def salu(text, writes=(), reads=()):
    return Op(text, "salu", reads, writes)


def chains(n):
    """E and dP of block n: two independent accumulate chains, interleaved."""
    e = [mfma(MF_F16, E, SEL[0], BW(0), "0"), mfma(MF_F16, E, SEL[1], BW(1), E)]
    e[0].needs = ["bias%d" % (n & 1)]
    for ss in range(4):
        m = mfma(MF_BF, E, KF(ss), QF(ss), E)
        if ss == 0:
            m.needs = ["kr%d_3" % (n & 3)]
        e.append(m)
    m = mfma(MF_BF, E, KONE, rr("v", V_QL, 4), E)
    m.needs = ["kone%d" % (n & 3)]
    e.append(m)
    d = []
    for ss in range(4):
        m = mfma(MF_BF, DP, VF(ss), DOF(ss), DP if ss else "0")
        if ss == 0:
            m.needs = ["vr%d_3" % (n & 3)]
        d.append(m)
    m = mfma(MF_BF, DP, KONE, rr("v", V_QD, 4), DP)
    m.needs = ["kone%d" % (n & 3)]
    d.append(m)
    out = []
    while e or d:  # E, dP, E, dP, ... (independent accumulators: neither waits for the other)
        if e:
            out.append(e.pop(0))
        if d:
            out.append(d.pop(0))
    return out


def dq_products(n):
    """dQ^T += K^T(n) dS^T(n)."""
    ms = []
    for s2 in range(2):
        for db in range(2):
            m = mfma(MF_BF, O(db), KT(s2, db), DS(s2), O(db))
            if db == 0:
                m.needs = ["kt%d_%d1b" % (n & 3, s2)]
            ms.append(m)
    return ms


def row_reads(n):
    """K and V row fragments + the mask word of block n (ring position n % 4)."""
    st, kb = n & 3, n & 1
    out = []
    for ss in range(4):
        out.append(Op("ds_read_b128 %s, v%d offset:%d" % (KF(ss), V_ADDR + ss, st * STAGE), "lds",
                      ["v%d" % (V_ADDR + ss)], regs("v", V_KF + 4 * ss, 4), tag="kr%d_%d" % (st, ss)))
    for ss in range(4):
        out.append(Op("ds_read_b128 %s, v%d offset:%d" % (VF(ss), V_ADDR + ss, st * STAGE + 4096), "lds",
                      ["v%d" % (V_ADDR + ss)], regs("v", V_VF + 4 * ss, 4), tag="vr%d_%d" % (st, ss)))
    out.append(Op("ds_read_b32 v%d, v%d offset:%d" % (V_KONE + 1, V_KMC, kb * 128), "lds",
                  ["v%d" % V_KMC], ["v%d" % (V_KONE + 1)], cost=4, tag="kone%d" % st))
    return out


def kt_reads(n):
    st = n & 3
    out = []
    for s2 in range(2):
        for db in range(2):
            base = V_KT + 4 * (2 * s2 + db)
            off = st * STAGE + s2 * 2048
            out.append(Op("ds_read_b64_tr_b16 %s, v%d offset:%d" % (rr("v", base, 2), V_ADDR + 4 + 2 * db, off), "lds",
                          ["v%d" % (V_ADDR + 4 + 2 * db)], regs("v", base, 2), cost=6, tag="kt%d_%d%da" % (st, s2, db)))
            out.append(Op("ds_read_b64_tr_b16 %s, v%d offset:%d" % (rr("v", base + 2, 2), V_ADDR + 5 + 2 * db, off), "lds",
                          ["v%d" % (V_ADDR + 5 + 2 * db)], regs("v", base + 2, 2), cost=6, tag="kt%d_%d%db" % (st, s2, db)))
    return out


def ds_valu():
    """dS = exp2(E) * dP as bf16: every consumer at least two instructions behind the transcendental that feeds it."""
    def ex(i): return Op("v_exp_f32_e32 %s, %s" % (Er(i), Er(i)), "trans", [Er(i)], [Er(i)])
    def mu(i): return valu("v_mul_f32_e32 %s, %s, %s" % (Dr(i), Er(i), Dr(i)), [Er(i), Dr(i)], [Dr(i)])
    def cv(d): return valu("v_cvt_pk_bf16_f32 %s, %s, %s" % (DSr(d), Dr(2 * d), Dr(2 * d + 1)), [Dr(2 * d), Dr(2 * d + 1)], [DSr(d)])
    # all sixteen exponentials first, then the products, then the conversions: every instruction's inputs were written eight or
    # more instructions earlier (the first arrangement -- exp, exp, mul, exp, mul, cvt ... -- issued each consumer one to three
    # instructions behind its producer: SQ_WAIT_INST_ANY was 57 % of the wave cycles)
    return [ex(i) for i in range(16)] + [mu(i) for i in range(16)] + [cv(d) for d in range(8)]


PIECES = [(0, 0), (4096, 1)]  # (LDS offset inside the stage, K or V)


def dma_piece(j, st):
    lds, kv = PIECES[j]
    return [salu("s_add_u32 m0, s%d, 0x%x" % (S_W1K, st * STAGE + lds), ["m0"], ["s%d" % S_W1K]),
            Op("buffer_load_dwordx4 v%d, s[%d:%d], s%d offen lds" % (V_DOFF, S_RKV, S_RKV + 3, S_SOFF + kv),
               "dma", ["v%d" % V_DOFF, "m0"], [], tag="dma%d_%d" % (st, j))]


def dma_offsets():
    """voffset of the DMA block with every row IN RANGE (a row at or past the valid row count reads the last valid row: finite
    data whose scores the dense bias switches off).  An instruction whose lanes are ALL out of range retires at once and out of
    order, and the counted s_waitcnt vmcnt(n) then pass with an older, real piece still in flight: attn_fwd2_gen.py."""
    t = "v%d" % (V_DOFF + 1)
    return [valu("v_subrev_u32_e32 %s, s%d, v%d" % (t, S_REM, V_ADDR + 11), ["v%d" % (V_ADDR + 11)], [t]),   # row - rem
            valu("v_add_u32_e32 %s, 1, %s" % (t, t), [t], [t]),
            valu("v_max_i32_e32 %s, 0, %s" % (t, t), [t], [t]),                                              # rows to step back
            valu("v_mul_u32_u24_e32 %s, s%d, %s" % (t, S_STEP, t), [t], [t]),                                 # x 32 rows of bytes
            valu("v_lshrrev_b32_e32 %s, 5, %s" % (t, t), [t], [t]),
            valu("v_sub_u32_e32 v%d, v%d, %s" % (V_DOFF, V_ADDR + 10, t), [t, "v%d" % (V_ADDR + 10)], ["v%d" % V_DOFF])]


def dma_advance():
    out = [salu("s_add_u32 s%d, s%d, s%d" % (S_SOFF + i, S_SOFF + i, S_STEP), ["s%d" % (S_SOFF + i)]) for i in range(2)]
    out.append(salu("s_sub_i32 s%d, s%d, 32" % (S_REM, S_REM), ["s%d" % S_REM]))
    return out + dma_offsets()


def bias_request(m):
    """Operands of block m, counted from the first block of the running loop pass (m = 1 in the preamble, 2..5 in the body:
    blocks 4 and 5 are the next pass's first tile)."""
    out = []
    soff = [S_BOFF, S_B1, S_B2][m >> 1]
    for j in range(2):
        out.append(Op("buffer_load_dwordx4 %s, v%d, s[%d:%d], s%d offen offset:%d" % (BW(j), V_ADDR + 12, S_RB, S_RB + 3, soff, (m & 1) * 2048 + j * 1024),
                      "vmem", ["v%d" % (V_ADDR + 12)], regs("v", V_BW + 4 * j, 4), tag="bias%d" % (m & 1)))
    return out


def spread(fill, extra, at, gap=2):
    for k, e in enumerate(extra):
        fill.insert(min(at + gap * k, len(fill)), e)
    return fill


def unit(n, first=False):
    """barrier; V(n); M(n + 1) for ring position n % 4 (n = 0..3 inside the unrolled body)."""
    body = []
    # ---- block n + 1 has landed (its pieces were issued in M(n - 1)); every wave is past V(n - 1): stage of block n - 1 is free
    body.append(Op("s_nop 0", "salu", needs=["dma%d_1" % ((n + 1) & 3)]))
    body.append(Op("s_barrier", "salu"))
    if (n & 1) == 1:
        # the rows read below belong to the NEXT 64-key tile: its mask words (tile 1, then "no mask")
        body += [salu("s_cmp_eq_u32 s%d, 1" % S_U, ["scc"]),
                 salu("s_cselect_b64 vcc, -1, 0", ["vcc"], ["scc"]),
                 valu("v_cndmask_b32_e32 v%d, v%d, v%d, vcc" % (V_KMC, V_ADDR + 9, V_ADDR + 13), ["vcc"], ["v%d" % V_KMC])]
    # ---- V(n): fragment reads first (they fly under the exponentials), then dS
    v = kt_reads(n) + row_reads(n + 1) + ds_valu()
    body += v
    # ---- M(n + 1): dQ(n) + chains(n + 1); fillers: DMA of block n + 3, bias operands of block n + 2, bookkeeping
    ms = dq_products(n)[:2] + chains(n + 1) + dq_products(n)[2:]
    fill = []
    st = (n + 3) & 3
    for j in range(2):
        d = dma_piece(j, st)
        d[0].after = d[1].after = 1 + 6 * j
        fill += d
    adv = dma_advance()
    for o in adv:
        o.after = 11
    req = bias_request(n + 2)
    for o in req:
        o.after = 5  # both selection MFMAs of block n + 1 have been issued (they are MFMAs 3 and 5 of the phase)
    bump = []
    if (n & 3) == 1:
        bump = []
    fill = fill[:2] + req + fill[2:] + adv + [salu("s_add_u32 s%d, s%d, 1" % (S_U, S_U), ["s%d" % S_U])]
    fill.sort(key=lambda o: o.after)
    body += interleave(ms, fill)
    return body


def build():
    pre = []
    for a in range(32):
        pre.append(valu("v_accvgpr_write_b32 a%d, 0" % a, [], ["a%d" % a]))
    pre += [salu("s_mov_b32 s%d, 0" % S_U, ["s%d" % S_U]), salu("s_mov_b32 s%d, 0" % S_BOFF, ["s%d" % S_BOFF]),
            salu("s_mov_b32 s%d, 0x1000" % S_B1, ["s%d" % S_B1]), salu("s_mov_b32 s%d, 0x2000" % S_B2, ["s%d" % S_B2]),
            salu("s_mov_b32 s%d, -1" % S_LO, ["s%d" % S_LO]), salu("s_mov_b32 s%d, 0" % (S_LO + 1), ["s%d" % (S_LO + 1)])]
    # key side of the statistics steps: k-slots 0, 1 (and 3, with the mask word) = 1 in the lower half-wave
    for v in range(V_KONE, V_KONE + 4):
        pre.append(valu("v_mov_b32_e32 v%d, 0" % v, [], ["v%d" % v]))
    t0 = "v%d" % V_T
    pre.append(valu("v_mov_b32_e32 %s, 0x3f803f80" % t0, [], [t0]))
    pre.append(valu("v_cndmask_b32_e64 v%d, 0, %s, s[%d:%d]" % (V_KONE, t0, S_LO, S_LO + 1), [t0], ["v%d" % V_KONE]))
    # Two waves share a SIMD (two workgroups per CU) and run the same program: left alone they fall into lockstep and take
    # turns MFMA by MFMA (both stretch their matrix phases, both then sit in their vector phases together).  The wave in the ODD
    # hardware slot of its SIMD runs at priority 1 for the whole stream -- one static s_setprio, no flips
    # (MI355X_MICROARCH.md, "Two waves per SIMD", items 4 and 9) -- so that one wave's matrix chain runs through while its
    # partner does its vector work.  PRIO_EXPERIMENT (env-less switch of the generator): "none" leaves it out.
    if PRIO == "slot":
        pre += [Op("s_getreg_b32 s%d, hwreg(HW_REG_HW_ID, 0, 4)" % S_TMP, "salu", [], ["s%d" % S_TMP]),
                Op("s_and_b32 s%d, s%d, 1" % (S_TMP, S_TMP), "salu", ["s%d" % S_TMP], ["s%d" % S_TMP]),
                Op("s_cmp_eq_u32 s%d, 1" % S_TMP, "salu", ["s%d" % S_TMP], ["scc"]),
                Op("s_cbranch_scc0 L_noprio_%=", "salu"),
                Op("s_setprio 1", "salu"),
                Op("L_noprio_%=:", "raw")]
    pre += dma_offsets()
    pre.append(valu("v_mov_b32_e32 v%d, v%d" % (V_KMC, V_ADDR + 8), [], ["v%d" % V_KMC]))
    # M(0) alone: the chains of block 0 (no dQ yet) on the bias operands the prologue loaded; block 1's operands came with the
    # prologue too (in the first K^T fragment registers, which nothing uses before V(0)) and move into place behind the chains
    pre += row_reads(0)
    pre += chains(0)
    for i in range(8):
        pre.append(valu("v_mov_b32_e32 v%d, v%d" % (V_BW + i, V_KT + i), ["v%d" % (V_KT + i)], ["v%d" % (V_BW + i)]))

    body = [Op("L_loop_%=:", "raw")]
    for n in range(4):
        body += unit(n)
        if n == 1:
            body += [salu("s_cmp_lt_u32 s%d, s%d" % (S_U, S_NU), ["scc"]), Op("s_cbranch_scc0 L_exit_%=", "salu")]
        if n == 3:
            body += [salu("s_add_u32 s%d, s%d, 0x2000" % (S_BOFF, S_BOFF), ["s%d" % S_BOFF], ["s%d" % S_BOFF]),
                     salu("s_add_u32 s%d, s%d, 0x2000" % (S_B1, S_B1), ["s%d" % S_B1], ["s%d" % S_B1]),
                     salu("s_add_u32 s%d, s%d, 0x2000" % (S_B2, S_B2), ["s%d" % S_B2], ["s%d" % S_B2]),
                     salu("s_cmp_lt_u32 s%d, s%d" % (S_U, S_NU), ["scc"]), Op("s_cbranch_scc1 L_loop_%=", "salu")]

    pre_w, body_w = place_waits(pre, body)
    pre_h = pad_hazards(pre_w)
    body_h1 = pad_hazards(body_w, history=pre_h)
    body_h = pad_hazards(body_w, history=body_h1)
    if [o.text for o in body_h] != [o.text for o in body_h1]:
        merged, i, j = [], 0, 0
        while i < len(body_h1) or j < len(body_h):
            a = body_h1[i] if i < len(body_h1) else None
            b = body_h[j] if j < len(body_h) else None
            if a is not None and b is not None and a.text == b.text:
                merged.append(a); i += 1; j += 1
            elif a is not None and a.text.startswith("s_nop"):
                merged.append(a); i += 1
            elif b is not None and b.text.startswith("s_nop"):
                merged.append(b); j += 1
            else:
                raise AssertionError("streams diverge: %r / %r" % (a and a.text, b and b.text))
        body_h = merged
    tail = ["L_exit_%=:", "s_setprio 0", "s_nop 15", "s_nop 15"]
    text = emit(pre_h) + emit(body_h) + "".join(("" if l.endswith(":") else "  ") + l + "\n" for l in tail)
    return text, pre_h, body_h


HEADER = """// GENERATED by gen/attn_dq2_gen.py -- do not edit; `python vl-merging_amd/csrc/gen/attn_dq2_gen.py` rewrites it.
// The instruction stream of attn_bwd_dq2_kernel's block loop (register map: the generator / attention_bwd_dq2.hip).
"""


def render():
    text, pre, body = build()
    lines = [HEADER]
    for l in text.splitlines():
        lines.append('"%s\\n"\n' % l.replace('"', '\\"'))
    return "".join(lines), pre, body


if __name__ == "__main__":
    out_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "attention_dq2_body.inc")
    txt, pre, body = render()
    if "--check" in sys.argv:
        sys.exit(0 if os.path.exists(out_path) and open(out_path).read() == txt else 1)
    open(out_path, "w").write(txt)
    from collections import Counter
    c = Counter(o.kind for o in body)
    print("loop body (four 32-key units): %d instructions, %s" % (len(body), dict(c)))
