"""Generates attention_fwd2_body.inc: the hand-placed instruction stream of attn_fwd2_kernel (attention_fwd2.hip).

    python vl-merging_amd/csrc/gen/attn_fwd2_gen.py            # rewrite the .inc
    python vl-merging_amd/csrc/gen/attn_fwd2_gen.py --check    # exit 1 if the committed .inc is stale

The kernel's work decomposition and LDS layout are documented in attention_fwd2.hip; this file owns the register map.
One wave = 32 query positions of TWO samples (the relative-position bias of a (query block, key block) pair is loaded once
and added to both samples' scores); 256 registers per wave, two workgroups (waves) per CU (SIMD).

Keys are streamed in 32-key BLOCKS through a ring of four 16-KiB stages [K s0 | K s1 | V s0 | V s1]; a TRIP is one 64-key
tile = two blocks = four UNITS u = (kb, s):
  A(u): 7 MFMAs  S[s]  = bias(kb) (2 selection MFMAs, f16) + K(kb, s) (c1 Q_s)^T (4) + statistics step (-m_s, key mask)
  B(u): 4 MFMAs  O_s^T += V(kb, s)^T P^T                     (P = bf16(exp2(S[s])); the row sums are f32 adds)
software-pipelined as  A(u+1) || exp / row sums / cvt(u), V reads(u)   then   B(u) || max(u+1), rescale decision(u+1).
The loop body is two trips (ring positions 0,1 and 2,3), so every LDS address is a per-lane base + an immediate.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from asmgen import Op, mfma, rr, regs, interleave, place_waits, pad_hazards, emit  # noqa: E402

NSTAGE = 4
STAGE = 16384                 # per stage: K s0 | K s1 | V s0 | V s1, 4 KiB each (32 keys x 128 B)
KM_BASE = NSTAGE * STAGE      # key-mask words of tiles 0, 1: [tile][sample][64]; then 1 KiB of zeros

# ---- register map (must match attention_fwd2.hip) -----------------------------------------------------------------------
V_ADDR = 32      # v32..35 K fragment addresses (stage 0, sample 0), v36..37 V^T addresses, v38 mask word (tile 0), v39 zero word,
                 # v40 K DMA voffset, v41 V DMA voffset, v42 block row of this lane's DMA piece, v43 bias voffset,
                 # v44 mask word (tile 1), v45 = 0xFFFFFFF0 (out-of-range offset), v46 current mask word address, v47 = bf16 (1, 1)
V_SEL = 48       # v48..51 sel0, v52..55 sel1
V_Q = 56         # v[56 + 16 s + 4 ss : +3] = c1 Q_s fragment of k-step ss
V_S = 88         # S[s] = v[88 + 16 s : +15]
V_P = 120        # P = v[120:127]: one buffer (a unit's P is consumed before the next unit's conversions start)
V_KF = 128       # K fragments, 4 x 4
V_VF = 144       # V^T fragments, (s2, db) -> 144 + 4 (2 s2 + db)
V_BW = 160       # bias operands of a trip's FIRST key block, w[j] -> 160 + 4 j; the SECOND block's set is V_BW1.  A set is
                 # re-requested for the NEXT trip's block as soon as both samples' selection MFMAs have read it: a whole trip of
                 # flight (one set re-requested for the very next block gave the load ~500 cycles against an L2 round trip of ~700)
V_BW1 = 24       # v24..v31 (below the stream's v32..v191: handed in and out through the statement's operands)
V_KONE = 168     # key side of the statistics step: dword 0 = (1, 1) in the lower half-wave, dword 1 = mask word
V_QM = 172       # query side: v[172 + 4 s : +3]; dword 0 = (-m hi, -m lo), dword 1 = (1, 0)
V_M = 180        # running reference point m_s
V_L = 182        # row sums: v[182 + 2 s], v[183 + 2 s] (two partial sums per sample)
V_T = 186        # temporaries v186..v189
V_DOFF = 190     # current (range-masked) DMA voffsets: 190 K, 191 V
V_KMC = 46       # current mask word address (one of v38 / v44 / v39)
A_O = 0          # o[s][db] = a[32 s + 16 db : +15]
# v0..v31 are the compiler's (values that live across the asm statement)

S_RKV, S_RB = 40, 44
S_SOFF = 48      # s48 K sample 0, s49 K sample 1, s50 V sample 0, s51 V sample 1 (byte offsets of the DMA block's first row)
S_NT = 52        # number of trips (64-key tiles)
S_REM = 53       # valid rows from the DMA block's first position on (may be <= 0)
S_W1K = 54       # wave * 1024: this wave's piece inside a 4-KiB operand image
S_STEP = 55      # 32 * ld * 2
S_T = 57
S_NL = 58        # s[58:59]: all ones while a next trip exists
S_BOFF = 60      # bias scalar offset of the current trip's SECOND key block (trip * 4096 + 2048)
S_PF = 56        # bias prefetch base: the NEXT trip's first key block (S_BOFF + 2048) while a next trip exists, else this
                 # trip's own (S_BOFF - 2048): the last trip's requests stay inside the table (see dma_offsets)
S_TMP = 61
S_LO = 62        # s[62:63] = lower half-wave

PRIO = os.environ.get("VLM_GEN_PRIO", "slot")
KO = set(x for x in os.environ.get("VLM_GEN_KO", "").split(",") if x)  # timing-only knock-outs (wrong results): docs/experiments.md
MF_BF = "v_mfma_f32_32x32x16_bf16"
MF_F16 = "v_mfma_f32_32x32x16_f16"


def S(s): return rr("v", V_S + 16 * s, 16)
def Sr(s, i): return "v%d" % (V_S + 16 * s + i)
def P(s, half): return rr("v", V_P + 4 * half, 4)
def Pr(s, d): return "v%d" % (V_P + d)
def KF(ss): return rr("v", V_KF + 4 * ss, 4)
def VF(s2, db): return rr("v", V_VF + 4 * (2 * s2 + db), 4)
def BW(kb, j): return rr("v", (V_BW1 if kb else V_BW) + 4 * j, 4)
def QF(s, ss): return rr("v", V_Q + 16 * s + 4 * ss, 4)
def QM(s): return rr("v", V_QM + 4 * s, 4)
def O(s, db): return rr("a", A_O + 32 * s + 16 * db, 16)
def Lr(s, i): return "v%d" % (V_L + 2 * s + i)


KONE = rr("v", V_KONE, 4)
SEL = [rr("v", V_SEL, 4), rr("v", V_SEL + 4, 4)]


def valu(text, reads, writes, **kw):
    return Op(text, "valu", reads, writes, **kw)


def salu(text, writes=(), reads=()):
    return Op(text, "salu", reads, writes)


# ---- pieces of a unit; `st` = ring position (0..3) of the unit's key block ------------------------------------------------
def chain(u):
    """A(u): the score chain of unit u = (kb, s) into S[s]."""
    kb, s = u >> 1, u & 1
    ms = [mfma(MF_F16, S(s), SEL[0], BW(kb, 0), "0"), mfma(MF_F16, S(s), SEL[1], BW(kb, 1), S(s))]
    ms[0].needs = ["bias%d" % kb]
    for ss in range(4):
        m = mfma(MF_BF, S(s), KF(ss), QF(s, ss), S(s))
        if ss == 0:
            m.needs = ["k%d_3" % u]  # LDS results return in order: the last fragment's wait covers all four
        ms.append(m)
    m = mfma(MF_BF, S(s), KONE, QM(s), S(s))
    m.needs = ["kone%d" % u]
    ms.append(m)
    return ms


def k_reads(u, st):
    kb, s = u >> 1, u & 1
    out = []
    for ss in range(4):
        out.append(Op("ds_read_b128 %s, v%d offset:%d" % (KF(ss), V_ADDR + ss, st * STAGE + s * 4096), "lds",
                      ["v%d" % (V_ADDR + ss)], regs("v", V_KF + 4 * ss, 4), tag="k%d_%d" % (u, ss)))
    out.append(Op("ds_read_b32 v%d, v%d offset:%d" % (V_KONE + 1, V_KMC, s * 256 + kb * 128), "lds",
                  ["v%d" % V_KMC], ["v%d" % (V_KONE + 1)], cost=4, tag="kone%d" % u))
    return out


def v_reads(u, st, after=0):
    kb, s = u >> 1, u & 1
    out = []
    for s2 in range(2):
        for db in range(2):
            base = V_VF + 4 * (2 * s2 + db)
            off = st * STAGE + 8192 + s * 4096 + s2 * 2048
            out.append(Op("ds_read_b64_tr_b16 %s, v%d offset:%d" % (rr("v", base, 2), V_ADDR + 4 + db, off), "lds",
                          ["v%d" % (V_ADDR + 4 + db)], regs("v", base, 2), cost=6, tag="v%d_%d%da" % (u, s2, db), after=after))
            out.append(Op("ds_read_b64_tr_b16 %s, v%d offset:%d" % (rr("v", base + 2, 2), V_ADDR + 4 + db, off + 1024), "lds",
                          ["v%d" % (V_ADDR + 4 + db)], regs("v", base + 2, 2), cost=6, tag="v%d_%d%db" % (u, s2, db), after=after))
    return out


def exp_cvt(u):
    """P[s] = bf16(exp2(S[s])), l_s += sum of the ROUNDED weights (v_dot2c_f32_bf16 against (1, 1): the normaliser is the sum of
    exactly the values that enter P V, every output row an exact convex combination -- f32 adds of the unrounded exponentials
    measured 4-5 % more output error): 16 v_exp_f32 in place, 8 v_cvt_pk_bf16_f32, 8 v_dot2c into two partial sums; every
    consumer sits at least two instructions behind the transcendental that feeds it."""
    s = u & 1
    out = []
    ones = "v%d" % (V_ADDR + 15)

    def ex(i):
        if "exp" in KO:
            return valu("v_mov_b32_e32 %s, %s" % (Sr(s, i), Sr(s, i)), [Sr(s, i)], [Sr(s, i)])
        return Op("v_exp_f32_e32 %s, %s" % (Sr(s, i), Sr(s, i)), "trans", [Sr(s, i)], [Sr(s, i)])
    def cv(d): return valu("v_cvt_pk_bf16_f32 %s, %s, %s" % (Pr(s, d), Sr(s, 2 * d), Sr(s, 2 * d + 1)), [Sr(s, 2 * d), Sr(s, 2 * d + 1)], [Pr(s, d)])
    def dt(d): return valu("v_dot2c_f32_bf16_e32 %s, %s, %s" % (Lr(s, d & 1), Pr(s, d), ones), [Lr(s, d & 1), Pr(s, d), ones], [Lr(s, d & 1)], cost=8)
    # sixteen exponentials, then the conversions, then the sums: every input was written eight or more instructions earlier
    out += [ex(i) for i in range(16)] + [cv(d) for d in range(8)] + ([] if "dot" in KO else [dt(d) for d in range(8)])
    return out


def pv(u):
    """B(u): O^T += V^T P^T."""
    kb, s = u >> 1, u & 1
    ms = []
    for s2 in range(2):
        for db in range(2):
            m = mfma(MF_BF, O(s, db), VF(s2, db), P(s, s2), O(s, db))
            if db == 0:
                m.needs = ["v%d_%d1b" % (u, s2)]  # in-order returns: the last read of this k-step covers its four
            ms.append(m)
    return ms


def max_decide(u, site, guard_last=False, after=1):
    """Row maximum of S[s] (both half-waves), then the rare branch that moves the reference point."""
    s = u & 1
    a, b, t = "v%d" % V_T, "v%d" % (V_T + 1), "v%d" % (V_T + 2)
    out = []

    def m3(d, x, y, z):
        return valu("v_max3_f32 %s, %s, %s, %s" % (d, x, y, z), [x, y, z], [d], after=after)
    out.append(m3(a, Sr(s, 0), Sr(s, 1), Sr(s, 2)))
    out.append(m3(b, Sr(s, 3), Sr(s, 4), Sr(s, 5)))
    out.append(m3(a, a, Sr(s, 6), Sr(s, 7)))
    out.append(m3(b, b, Sr(s, 8), Sr(s, 9)))
    out.append(m3(a, a, Sr(s, 10), Sr(s, 11)))
    out.append(m3(b, b, Sr(s, 12), Sr(s, 13)))
    out.append(m3(a, a, Sr(s, 14), Sr(s, 15)))
    out.append(valu("v_max_f32_e32 %s, %s, %s" % (a, a, b), [a, b], [a], after=after))
    out.append(valu("v_mov_b32_e32 %s, %s" % (t, a), [a], [t], after=after))
    out.append(valu("v_mov_b32_e32 %s, %s" % (b, a), [a], [b], after=after))  # (a wait state between the write and the swap)
    out.append(Op("v_permlane32_swap_b32_e32 %s, %s" % (t, a), "perm", [t, a], [t, a], after=after))
    out.append(valu("v_max_f32_e32 %s, %s, %s" % (a, a, t), [a, t], [a], after=after))
    out.append(valu("v_cmp_lt_f32_e32 vcc, 0x40c00000, %s" % a, [a], ["vcc"], after=after))
    if guard_last:
        out.append(Op("s_and_b64 vcc, vcc, s[%d:%d]" % (S_NL, S_NL + 1), "salu", ["vcc"], ["vcc"], after=after))
    out.append(Op("s_cbranch_vccnz L_rare_%s_%%=" % site, "salu", ["vcc"], [], after=after))
    out.append(Op("L_join_%s_%%=:" % site, "raw", after=after))
    return out


def rare_block(site, s):
    """Reference-point move of sample s (rows whose exponents exceed 2^6): m, (-m hi, -m lo), S -= delta, l and O *= 2^-delta."""
    mx, mn, dl, al = ("v%d" % (V_T + i) for i in range(4))
    mreg = "v%d" % (V_M + s)
    t = []
    t.append("L_rare_%s_%%=:" % site)
    t += ["s_nop 15", "s_nop 15"]
    t.append("v_add_f32_e32 %s, %s, %s" % (mn, mreg, mx))
    t.append("v_cvt_f16_f32_e32 %s, %s" % (mn, mn))
    t.append("v_cvt_f32_f16_e32 %s, %s" % (mn, mn))
    t.append("v_cmp_lt_f32_e32 vcc, 0, %s" % mx)
    t.append("v_cndmask_b32_e32 %s, %s, %s, vcc" % (mn, mreg, mn))      # m_new
    t.append("v_sub_f32_e32 %s, %s, %s" % (dl, mn, mreg))               # delta (exact: both fp16 values)
    t.append("v_mov_b32_e32 %s, %s" % (mreg, mn))
    t.append("v_exp_f32_e64 %s, -%s" % (al, dl))                        # alpha
    # (-m hi, -m lo) into dword 0 of the query-side operand, lower half-wave only
    t.append("v_cvt_pk_bf16_f32 %s, -%s, 0" % (mx, mn))
    t.append("v_lshlrev_b32_e32 %s, 16, %s" % (mx, mx))
    t.append("v_sub_f32_e64 %s, -%s, %s" % (mx, mn, mx))
    t.append("v_cvt_pk_bf16_f32 %s, -%s, %s" % (mx, mn, mx))
    t.append("v_cndmask_b32_e64 v%d, v%d, %s, s[%d:%d]" % (V_QM + 4 * s, V_QM + 4 * s, mx, S_LO, S_LO + 1))
    for i in range(16):
        t.append("v_sub_f32_e32 %s, %s, %s" % (Sr(s, i), Sr(s, i), dl))
    for i in range(2):
        t.append("v_mul_f32_e32 %s, %s, %s" % (Lr(s, i), Lr(s, i), al))
    for a in range(A_O + 32 * s, A_O + 32 * s + 32):
        t.append("v_accvgpr_read_b32 %s, a%d" % (mx, a))
        t.append("s_nop 0")
        t.append("v_mul_f32_e32 %s, %s, %s" % (mx, mx, al))
        t.append("v_accvgpr_write_b32 a%d, %s" % (a, mx))
    t += ["s_nop 7", "s_branch L_join_%s_%%=" % site]
    return t


PIECES = [(0, 0), (0, 1), (1, 0), (1, 1)]  # (sample, K or V): this wave's four pieces of a block


def dma_piece(j, st, name):
    s, kv = PIECES[j]
    imm = st * STAGE + kv * 8192 + s * 4096
    return [salu("s_add_u32 m0, s%d, 0x%x" % (S_W1K, imm), ["m0"], ["s%d" % S_W1K]),
            Op("buffer_load_dwordx4 v%d, s[%d:%d], s%d offen lds" % (V_DOFF + kv, S_RKV, S_RKV + 3, S_SOFF + 2 * kv + s),
               "dma", ["v%d" % (V_DOFF + kv), "m0"], [], tag="dma%s%d" % (name, j))]


def dma_offsets():
    """voffsets of the DMA block with every row IN RANGE: a row at or past the block's valid row count (S_REM, may be <= 0 for a
    block past the end) reads the LAST VALID row instead -- finite data whose scores the dense bias switches off (ATT_NEG_BIG
    for keys outside the range).  An out-of-range offset (hardware zero fill) would do for the arithmetic, but an instruction
    whose lanes are ALL out of range retires at once and out of order: the counted s_waitcnt vmcnt(n) in front of the barriers
    then pass with an older, real piece still in flight (seen as wrong rows of sample 1 -- the youngest pieces -- at two
    workgroups per CU)."""
    t = "v%d" % (V_T + 3)
    return [valu("v_subrev_u32_e32 %s, s%d, v%d" % (t, S_REM, V_ADDR + 10), ["v%d" % (V_ADDR + 10)], [t]),   # row - rem
            valu("v_add_u32_e32 %s, 1, %s" % (t, t), [t], [t]),
            valu("v_max_i32_e32 %s, 0, %s" % (t, t), [t], [t]),                                              # rows to step back
            valu("v_mul_u32_u24_e32 %s, s%d, %s" % (t, S_STEP, t), [t], [t]),                                 # x 32 rows of bytes
            valu("v_lshrrev_b32_e32 %s, 5, %s" % (t, t), [t], [t]),
            valu("v_sub_u32_e32 v%d, v%d, %s" % (V_DOFF, V_ADDR + 8, t), [t, "v%d" % (V_ADDR + 8)], ["v%d" % V_DOFF]),
            valu("v_sub_u32_e32 v%d, v%d, %s" % (V_DOFF + 1, V_ADDR + 9, t), [t, "v%d" % (V_ADDR + 9)], ["v%d" % (V_DOFF + 1)])]


def dma_advance():
    """The DMA block moves on by one (32 rows)."""
    out = []
    for i in range(4):
        out.append(salu("s_add_u32 s%d, s%d, s%d" % (S_SOFF + i, S_SOFF + i, S_STEP), ["s%d" % (S_SOFF + i)]))
    out.append(salu("s_sub_i32 s%d, s%d, 32" % (S_REM, S_REM), ["s%d" % S_REM]))
    return out + dma_offsets()


def bias_reload(kb, after):
    """Request the NEXT trip's key block kb into set kb (its current contents have been read by both samples' selection MFMAs)."""
    out = []
    for j in range(2):
        soff, imm = S_PF, kb * 2048 + j * 1024
        out.append(Op("buffer_load_dwordx4 %s, v%d, s[%d:%d], s%d offen offset:%d" % (BW(kb, j), V_ADDR + 11, S_RB, S_RB + 3, soff, imm),
                      "vmem", ["v%d" % (V_ADDR + 11)], regs("v", (V_BW1 if kb else V_BW) + 4 * j, 4), tag="bias%d" % kb, after=after))
    return out


def spread(fill, extra, at):
    """Insert the ops of `extra` into `fill` starting at index `at`, two positions apart."""
    for n, e in enumerate(extra):
        fill.insert(min(at + 2 * n, len(fill)), e)
    return fill


def trip(x, half):
    """One 64-key trip whose blocks sit at ring positions x, x + 1; the next trip's first block at (x + 2) % 4.
    DMA: phases 1-2 finish the block for ring position x + 2, phases 3-6 load x + 3, phases 7-8 start x + 4 = x."""
    st = [x, x, x + 1, x + 1]          # ring position of unit u's block
    nx = (x + 2) % NSTAGE
    body = []
    body += [salu("s_add_u32 s%d, s%d, 1" % (S_TMP, S_T), ["s%d" % S_TMP]),
             salu("s_cmp_lt_u32 s%d, s%d" % (S_TMP, S_NT), ["scc"]),
             salu("s_cselect_b64 s[%d:%d], -1, 0" % (S_NL, S_NL + 1), ["s%d" % S_NL, "s%d" % (S_NL + 1)], ["scc"]),
             salu("s_cselect_b32 s%d, 0x1000, 0" % S_PF, ["s%d" % S_PF], ["scc"]),           # (one literal per instruction)
             salu("s_add_u32 s%d, s%d, s%d" % (S_PF, S_PF, S_BOFF), ["s%d" % S_PF], ["s%d" % S_PF, "s%d" % S_BOFF]),
             salu("s_sub_u32 s%d, s%d, 0x800" % (S_PF, S_PF), ["s%d" % S_PF], ["s%d" % S_PF])]
    tag = "ab"[half]
    for u in (1, 2, 3):
        fill = v_reads(u - 1, st[u - 1], after=1) + exp_cvt(u - 1)
        if u & 1:
            fill += bias_reload(u >> 1, after=2)   # unit u = (kb, 1) is the second and last reader of set kb in this trip
        if u == 1:
            d = dma_piece(2, nx, tag + "A")
        else:
            d = dma_piece(2 * (u - 2), (x + 3) % NSTAGE, tag + "B")
        d[0].after = d[1].after = 1
        spread(fill, d, 9)
        # (the K fragments of unit u were requested at the head of the O phase before this one: an LDS round trip is ~150 cycles,
        # and as the lead of this phase it stood between the two bias-selection MFMAs and the first K MFMA of every chain)
        body += interleave(chain(u), fill)
        if u == 1:
            d = dma_piece(3, nx, tag + "A")
            adv = dma_advance()
        elif u == 2:
            d = dma_piece(1, (x + 3) % NSTAGE, tag + "B")
            adv = []
        else:
            d = dma_piece(3, (x + 3) % NSTAGE, tag + "B")
            adv = dma_advance()
        for o in adv:
            o.after = 3
        # head of the O phase: chain(u) has issued, the fragment registers are free -> request unit u + 1's K fragments
        if u == 1:
            # block x + 1 has landed for every wave (its pieces were issued in the previous trip's phases 3-6)
            lead = [Op("s_nop 0", "salu", needs=["dma%sB3" % "ba"[half]]), Op("s_nop 0" if "barrier" in KO else "s_barrier", "salu")] + k_reads(2, st[2])
        elif u == 2:
            lead = k_reads(3, st[3])
        else:
            # trip boundary: the next trip's first block has landed (pieces 0, 1 in the previous trip's phases 7-8, 2, 3 above)
            lead = [Op("s_nop 0", "salu", needs=["dma%sA3" % tag]), Op("s_nop 0" if "barrier" in KO else "s_barrier", "salu"),
                    salu("s_cmp_eq_u32 s%d, 0" % S_T, ["scc"]),
                    salu("s_cselect_b64 vcc, -1, 0", ["vcc"], ["scc"]),
                    valu("v_cndmask_b32_e32 v%d, v%d, v%d, vcc" % (V_KMC, V_ADDR + 7, V_ADDR + 12), ["vcc"], ["v%d" % V_KMC])] + k_reads(0, nx)
        body += interleave(pv(u - 1), d + max_decide(u, "%s%d" % (tag, u), after=1) + adv, lead=lead)
    # phase 7: A(0) of the next trip || exp/cvt(3), V reads of unit 3
    fill = v_reads(3, st[3], after=1) + exp_cvt(3)
    d = dma_piece(0, x, tag + "C")
    d[0].after = d[1].after = 1
    spread(fill, d, 9)
    body += interleave(chain(0), fill)
    d = dma_piece(1, x, tag + "C")
    body += interleave(pv(3), d + [salu("s_add_u32 s%d, s%d, 0x1000" % (S_BOFF, S_BOFF), ["s%d" % S_BOFF], ["s%d" % S_BOFF]),
                                   salu("s_add_u32 s%d, s%d, 1" % (S_T, S_T), ["s%d" % S_T])] + max_decide(0, tag + "0", guard_last=True, after=1),
                       lead=k_reads(1, (x + 2) % NSTAGE))
    return body


def build():
    # ---------------------------------------------------------------- once, before the loop
    pre = []
    for a in range(64):
        pre.append(valu("v_accvgpr_write_b32 a%d, 0" % a, [], ["a%d" % a]))
    pre += [salu("s_mov_b32 s%d, -1" % S_LO, ["s%d" % S_LO]), salu("s_mov_b32 s%d, 0" % (S_LO + 1), ["s%d" % (S_LO + 1)]),
            salu("s_mov_b32 s%d, 0" % S_T, ["s%d" % S_T]), salu("s_mov_b32 s%d, 0x800" % S_BOFF, ["s%d" % S_BOFF])]
    # statistics-step operands: key side (1, 1) in k-slots 0, 1 of the lower half-wave (dword 1 = the mask word, read per unit);
    # query side k-slots 0, 1 = -m = 0, k-slot 2 = 1 (times the key's mask word); m = 0; row sums = 0
    for v in range(V_KONE, V_L + 4):
        pre.append(valu("v_mov_b32_e32 v%d, 0" % v, [], ["v%d" % v]))
    t0 = "v%d" % V_T
    pre.append(valu("v_mov_b32_e32 %s, 0x3f803f80" % t0, [], [t0]))
    pre.append(valu("v_cndmask_b32_e64 v%d, 0, %s, s[%d:%d]" % (V_KONE, t0, S_LO, S_LO + 1), [t0], ["v%d" % V_KONE]))
    pre.append(valu("v_mov_b32_e32 %s, 0x3f80" % t0, [], [t0]))
    for s in range(2):
        pre.append(valu("v_cndmask_b32_e64 v%d, 0, %s, s[%d:%d]" % (V_QM + 4 * s + 1, t0, S_LO, S_LO + 1), [t0], ["v%d" % (V_QM + 4 * s + 1)]))
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
    pre.append(valu("v_mov_b32_e32 v%d, v%d" % (V_KMC, V_ADDR + 6), [], ["v%d" % V_KMC]))
    # A(0) of trip 0 alone, then its maximum / decision
    pre += k_reads(0, 0)
    pre += chain(0)
    pre += k_reads(1, 0)   # unit 1's fragments: the loop requests them one phase ahead, at the head of the O phase
    pre += max_decide(0, "pre", after=0)

    # ---------------------------------------------------------------- the loop: two trips per pass
    body = [Op("L_loop_%=:", "raw")]
    body += trip(0, 0)
    body += [salu("s_cmp_lt_u32 s%d, s%d" % (S_T, S_NT), ["scc"]), Op("s_cbranch_scc0 L_exit_%=", "salu")]
    body += trip(2, 1)
    body += [salu("s_cmp_lt_u32 s%d, s%d" % (S_T, S_NT), ["scc"]), Op("s_cbranch_scc1 L_loop_%=", "salu")]

    pre_w, body_w = place_waits(pre, body)
    pre_h = pad_hazards(pre_w)
    body_h1 = pad_hazards(body_w, history=pre_h)
    body_h = pad_hazards(body_w, history=body_h1)
    if [o.text for o in body_h] != [o.text for o in body_h1]:
        merged = []
        i = j = 0
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
    tail = ["L_exit_%=:", "s_branch L_done_%="]
    sites = [("pre", 0)]
    for tag in "ab":
        sites += [(tag + "1", 1), (tag + "2", 0), (tag + "3", 1), (tag + "0", 0)]
    for site, s in sites:
        tail += rare_block(site, s)
    # the last trip has requested bias operands (and DMA pieces) for a trip that never runs: they must have landed before the
    # statement ends -- the compiler takes the operand registers back and knows nothing of a load still in flight into them
    tail += ["L_done_%=:", "s_setprio 0", "s_waitcnt vmcnt(0)", "s_nop 15", "s_nop 15"]
    text = emit(pre_h) + emit(body_h) + "".join(("" if l.endswith(":") else "  ") + l + "\n" for l in tail)
    return text, pre_h, body_h


def stats(stream):
    from collections import Counter
    c = Counter(o.kind for o in stream)
    cost = sum(o.cost for o in stream)
    return dict(c), cost


HEADER = """// GENERATED by gen/attn_fwd2_gen.py -- do not edit; `python vl-merging_amd/csrc/gen/attn_fwd2_gen.py` rewrites it.
// The instruction stream of attn_fwd2_kernel's tile loop (register map: attention_fwd2.hip / the generator).
"""


def render():
    text, pre, body = build()
    lines = [HEADER]
    for l in text.splitlines():
        lines.append('"%s\\n"\n' % l.replace('"', '\\"'))
    return "".join(lines), pre, body


if __name__ == "__main__":
    out_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "attention_fwd2_body.inc")
    txt, pre, body = render()
    if "--check" in sys.argv:
        sys.exit(0 if os.path.exists(out_path) and open(out_path).read() == txt else 1)
    open(out_path, "w").write(txt)
    k, c = stats(body)
    nm = k.get("mfma", 0)
    print("loop body (two 64-key trips): %d instructions, %s; issue-cost estimate %d cycles (%d MFMAs = %d matrix cycles)"
          % (len(body), k, c, nm, 32 * nm))
