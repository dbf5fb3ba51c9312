"""Generates attention_fwd2_body.inc: the hand-placed instruction stream of attn_fwd2_kernel (attention_fwd2.hip).

    python vl-merging_amd/csrc/gen/attn_fwd2_gen.py            # rewrite the .inc
    python vl-merging_amd/csrc/gen/attn_fwd2_gen.py --check    # exit 1 if the committed .inc is stale

The kernel's work decomposition, LDS layout and register map are documented in attention_fwd2.hip; this file only
knows the map below.  One wave owns 32 query positions of TWO samples (the relative-position bias of a (query block,
key block) pair is loaded once and added to both samples' scores) and the whole 512-entry register file of its SIMD.

Per 64-key tile a wave runs four UNITS u = (key block kb, sample s), each
  A(u): 7 MFMAs  S[s]  = bias(kb) (2 selection MFMAs, f16) + K(kb, s) (c1 Q_s)^T (4) + statistics step (-m_s, key mask)
  B(u): 6 MFMAs  l_s  += 1^T P,  O_s^T += V(kb, s)^T P^T          (P = bf16(exp2(S[s])))
software-pipelined as  A(u+1) || exp/cvt(u), K/V fragment reads   then   B(u) || max(u+1), rescale decision(u+1),
so that the exponentials of one unit issue in the gaps of the other sample's score chain.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from asmgen import Op, mfma, rr, regs, interleave, place_waits, pad_hazards, emit  # noqa: E402

NSTAGE = 4
STAGE = 32768                 # per stage: K0 | K1 | V0 | V1, 8 KiB each
KM_BASE = NSTAGE * STAGE      # key-mask words of tiles 0, 1: [tile][sample][64]; then 1 KiB of zeros

# ---- register map (must match attention_fwd2.hip) -----------------------------------------------------------------------
V_ADDR = 32      # v32..35 K fragment addresses (stage 0), v36..37 V^T addresses, v38 mask word (tile 0), v39 zero word,
                 # v40..41 K DMA voffsets, v42..43 V DMA voffsets, v44..45 tile row of DMA piece u, v46 bias voffset,
                 # v47 mask word (tile 1)
V_CONST = 48     # v48..51 sel0, v52..55 sel1, v56..59 ones, v60 = 0xFFFFFFF0 (out-of-range offset), v61..63 spare
V_Q = 64         # v[64 + 16 s + 4 ss : +3] = c1 Q_s fragment of k-step ss
V_S = 96         # S[s] = v[96 + 16 s : +15]
V_P = 128        # P[s] = v[128 + 8 s : +7]
V_KF = 144       # K fragments, 4 x 4
V_VF = 160       # V^T fragments, (s2, db) -> 160 + 4 (2 s2 + db)
V_BW = 176       # bias operands w[kb][j] -> 176 + 8 kb + 4 j
V_KONE = 192     # key side of the statistics step: dword 0 = (1, 1) in the lower half-wave, dword 1 = mask word
V_QM = 196       # query side: v[196 + 4 s : +3]; dword 0 = (-m hi, -m lo), dword 1 = (1, 0)
V_M = 204        # running reference point m_s
V_T = 206        # temporaries v206..v209
V_DOFF = 210     # current (range-masked) DMA voffsets: 210, 211 K u = 0, 1; 212, 213 V
V_KC = 214       # current-stage K fragment addresses (4), 218..219 V^T addresses, 220 current mask word address
A_O = 0          # o[s][db] = a[32 s + 16 db : +15]
A_L = 64         # l[s] = a[64 + 16 s : +15]

S_RKV, S_RB = 40, 44
S_SOFF = 48      # s48 K sample 0, s49 K sample 1, s50 V sample 0, s51 V sample 1 (byte offsets of the DMA tile's first row)
S_NT = 52
S_REM = 53       # valid rows from the DMA tile's first position on (may be <= 0)
S_LDSW = 54      # LDS byte address of this wave's first piece in the DMA tile's stage
S_STEP = 55      # 64 * ld * 2
S_RD = 56        # LDS byte address of the stage being read
S_T = 57
S_NL = 58        # s[58:59]: all ones while a next tile exists
S_BOFF = 60      # bias scalar offset of the NEXT tile
S_TMP = 61
S_LO = 62        # s[62:63] = lower half-wave
S_LDS0 = 64      # LDS base (stage 0), s65 = end of the ring

MF_BF = "v_mfma_f32_32x32x16_bf16"
MF_F16 = "v_mfma_f32_32x32x16_f16"


def S(s): return rr("v", V_S + 16 * s, 16)
def Sr(s, i): return "v%d" % (V_S + 16 * s + i)
def P(s, half): return rr("v", V_P + 8 * s + 4 * half, 4)
def Pr(s, d): return "v%d" % (V_P + 8 * s + d)
def KF(ss): return rr("v", V_KF + 4 * ss, 4)
def VF(s2, db): return rr("v", V_VF + 4 * (2 * s2 + db), 4)
def BW(kb, j): return rr("v", V_BW + 8 * kb + 4 * j, 4)
def QF(s, ss): return rr("v", V_Q + 16 * s + 4 * ss, 4)
def QM(s): return rr("v", V_QM + 4 * s, 4)
def O(s, db): return rr("a", A_O + 32 * s + 16 * db, 16)
def L(s): return rr("a", A_L + 16 * s, 16)


KONE = rr("v", V_KONE, 4)
SEL = [rr("v", V_CONST, 4), rr("v", V_CONST + 4, 4)]
ONES = rr("v", V_CONST + 8, 4)


def valu(text, reads, writes, **kw):
    return Op(text, "valu", reads, writes, **kw)


def salu(text, writes=(), reads=()):
    return Op(text, "salu", reads, writes)


# ---- pieces of a unit ---------------------------------------------------------------------------------------------------
def chain(u, tile_tag):
    """A(u): the score chain of unit u = (kb, s) into S[s]."""
    kb, s = u >> 1, u & 1
    ms = [mfma(MF_F16, S(s), SEL[0], BW(kb, 0), "0"), mfma(MF_F16, S(s), SEL[1], BW(kb, 1), S(s))]
    ms[0].needs = ["bias%d" % kb]
    for ss in range(4):
        m = mfma(MF_BF, S(s), KF(ss), QF(s, ss), S(s))
        m.needs = ["k%d_%d" % (u, ss)]
        ms.append(m)
    m = mfma(MF_BF, S(s), KONE, QM(s), S(s))
    m.needs = ["kone%d" % u]
    ms.append(m)
    return ms


def k_reads(u):
    kb, s = u >> 1, u & 1
    out = []
    for ss in range(4):
        out.append(Op("ds_read_b128 %s, v%d offset:%d" % (KF(ss), V_KC + ss, s * 8192 + kb * 4096), "lds",
                      ["v%d" % (V_KC + ss)], regs("v", V_KF + 4 * ss, 4), tag="k%d_%d" % (u, ss)))
    out.append(Op("ds_read_b32 v%d, v%d offset:%d" % (V_KONE + 1, V_KC + 6, s * 256 + kb * 128), "lds",
                  ["v%d" % (V_KC + 6)], ["v%d" % (V_KONE + 1)], cost=4, tag="kone%d" % u))
    return out


def v_reads(u, after=0):
    kb, s = u >> 1, u & 1
    out = []
    for s2 in range(2):
        for db in range(2):
            base = V_VF + 4 * (2 * s2 + db)
            off = 16384 + s * 8192 + kb * 4096 + s2 * 2048
            out.append(Op("ds_read_b64_tr_b16 %s, v%d offset:%d" % (rr("v", base, 2), V_KC + 4 + db, off), "lds",
                          ["v%d" % (V_KC + 4 + db)], regs("v", base, 2), cost=6, tag="v%d_%d%da" % (u, s2, db), after=after))
            out.append(Op("ds_read_b64_tr_b16 %s, v%d offset:%d" % (rr("v", base + 2, 2), V_KC + 4 + db, off + 1024), "lds",
                          ["v%d" % (V_KC + 4 + db)], regs("v", base + 2, 2), cost=6, tag="v%d_%d%db" % (u, s2, db), after=after))
    return out


def exp_cvt(u):
    """P[s] = bf16(exp2(S[s])): 16 v_exp_f32 in place, 8 v_cvt_pk_bf16_f32 (each at least two instructions behind its inputs)."""
    s = u & 1
    out = []
    order = list(range(16))
    pending = []
    for n, i in enumerate(order):
        out.append(Op("v_exp_f32_e32 %s, %s" % (Sr(s, i), Sr(s, i)), "trans", [Sr(s, i)], [Sr(s, i)]))
        if i & 1:
            pending.append(i >> 1)
        if len(pending) >= 2 or (n == 15):
            # convert the OLDER finished pair(s): its exponentials are >= 2 instructions back
            while pending and (len(pending) >= 2 or n == 15):
                d = pending.pop(0)
                out.append(valu("v_cvt_pk_bf16_f32 %s, %s, %s" % (Pr(s, d), Sr(s, 2 * d), Sr(s, 2 * d + 1)),
                                [Sr(s, 2 * d), Sr(s, 2 * d + 1)], [Pr(s, d)]))
    return out


def pv(u):
    """B(u): row sums and O^T += V^T P^T."""
    kb, s = u >> 1, u & 1
    ms = []
    for s2 in range(2):
        ms.append(mfma(MF_BF, L(s), ONES, P(s, s2), L(s)))
        for db in range(2):
            m = mfma(MF_BF, O(s, db), VF(s2, db), P(s, s2), O(s, db))
            m.needs = ["v%d_%d%da" % (u, s2, db), "v%d_%d%db" % (u, s2, db)]
            ms.append(m)
    return ms


def max_decide(u, site, guard_last=False, after=2):
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
    accs = list(range(A_L + 16 * s, A_L + 16 * s + 16)) + list(range(A_O + 32 * s, A_O + 32 * s + 32))
    for a in accs:
        t.append("v_accvgpr_read_b32 %s, a%d" % (mx, a))
        t.append("s_nop 0")
        t.append("v_mul_f32_e32 %s, %s, %s" % (mx, mx, al))
        t.append("v_accvgpr_write_b32 a%d, %s" % (a, mx))
    t += ["s_nop 7", "s_branch L_join_%s_%%=" % site]
    return t


def dma_piece(j):
    """Piece j of this wave's eight per tile: (sample, K or V, u)."""
    order = [(0, 0, 0), (0, 0, 1), (0, 1, 0), (0, 1, 1), (1, 0, 0), (1, 0, 1), (1, 1, 0), (1, 1, 1)]
    s, kv, u = order[j]
    imm = kv * 16384 + s * 8192 + u * 1024
    return [salu("s_add_u32 m0, s%d, 0x%x" % (S_LDSW, imm), ["m0"], ["s%d" % S_LDSW]),
            Op("buffer_load_dwordx4 v%d, s[%d:%d], s%d offen lds" % (V_DOFF + 2 * kv + u, S_RKV, S_RKV + 3, S_SOFF + 2 * kv + s),
               "dma", ["v%d" % (V_DOFF + 2 * kv + u), "m0"], [], tag="dma%d" % j)]


def dma_offsets():
    """Range-masked voffsets of the DMA tile: rows >= the tile's valid row count point out of range (zero fill)."""
    t = "s%d" % S_TMP
    out = [salu("s_min_i32 %s, s%d, 64" % (t, S_REM), [t]), salu("s_max_i32 %s, %s, 0" % (t, t), [t])]
    for u in range(2):
        out.append(valu("v_cmp_gt_u32_e32 vcc, %s, v%d" % (t, V_ADDR + 12 + u), [t], ["vcc"]))
        out.append(valu("v_cndmask_b32_e32 v%d, v%d, v%d, vcc" % (V_DOFF + u, V_CONST + 12, V_ADDR + 8 + u), ["vcc"], ["v%d" % (V_DOFF + u)]))
        out.append(valu("v_cndmask_b32_e32 v%d, v%d, v%d, vcc" % (V_DOFF + 2 + u, V_CONST + 12, V_ADDR + 10 + u), ["vcc"], ["v%d" % (V_DOFF + 2 + u)]))
    return out


def dma_advance():
    """The DMA tile moves on by one: scalar row offsets, remaining rows, ring slot."""
    out = []
    for i in range(4):
        out.append(salu("s_add_u32 s%d, s%d, s%d" % (S_SOFF + i, S_SOFF + i, S_STEP), ["s%d" % (S_SOFF + i)]))
    out.append(salu("s_sub_i32 s%d, s%d, 64" % (S_REM, S_REM), ["s%d" % S_REM]))
    out.append(salu("s_add_u32 s%d, s%d, 0x%x" % (S_LDSW, S_LDSW, STAGE), ["s%d" % S_LDSW]))
    out.append(salu("s_cmp_ge_u32 s%d, s%d" % (S_LDSW, S_LDS0 + 1), ["scc"]))
    out.append(salu("s_cselect_b32 s%d, 0x%x, 0" % (S_TMP, NSTAGE * STAGE), ["s%d" % S_TMP], ["scc"]))
    out.append(salu("s_sub_u32 s%d, s%d, s%d" % (S_LDSW, S_LDSW, S_TMP), ["s%d" % S_LDSW]))
    return out + dma_offsets()


def read_addresses(k_only=None):
    """Current-stage fragment addresses from the stage-0 addresses + the read stage's byte offset (s56 is RELATIVE to stage 0)."""
    out = []
    if k_only in (None, True):
        for ss in range(4):
            out.append(valu("v_add_u32_e32 v%d, s%d, v%d" % (V_KC + ss, S_RD, V_ADDR + ss), ["s%d" % S_RD], ["v%d" % (V_KC + ss)]))
    if k_only in (None, False):
        for db in range(2):
            out.append(valu("v_add_u32_e32 v%d, s%d, v%d" % (V_KC + 4 + db, S_RD, V_ADDR + 4 + db), ["s%d" % S_RD], ["v%d" % (V_KC + 4 + db)]))
    return out


def bias_reload(kb, after):
    out = []
    for j in range(2):
        out.append(Op("buffer_load_dwordx4 %s, v%d, s[%d:%d], s%d offen offset:%d" % (BW(kb, j), V_ADDR + 14, S_RB, S_RB + 3, S_BOFF, (2 * kb + j) * 1024),
                      "vmem", ["v%d" % (V_ADDR + 14)], regs("v", V_BW + 8 * kb + 4 * j, 4), tag="bias%d" % kb, after=after))
    return out


def build():
    # ---------------------------------------------------------------- once, before the loop
    pre = []
    for a in range(96):
        pre.append(valu("v_accvgpr_write_b32 a%d, 0" % a, [], ["a%d" % a]))
    pre += [salu("s_mov_b32 s%d, -1" % S_LO, ["s%d" % S_LO]), salu("s_mov_b32 s%d, 0" % (S_LO + 1), ["s%d" % (S_LO + 1)]),
            salu("s_mov_b32 s%d, 0" % S_T, ["s%d" % S_T]), salu("s_mov_b32 s%d, 0" % S_RD, ["s%d" % S_RD]),
            salu("s_mov_b32 s%d, 0x1000" % S_BOFF, ["s%d" % S_BOFF]),
            salu("s_add_u32 s%d, s%d, 0x%x" % (S_LDS0 + 1, S_LDS0, NSTAGE * STAGE), ["s%d" % (S_LDS0 + 1)])]
    # statistics-step operands: key side (1, 1) in k-slots 0, 1 of the lower half-wave (dword 1 = the mask word, read per unit);
    # query side k-slots 0, 1 = -m = 0, k-slot 2 = 1 (times the key's mask word); m = 0
    for v in range(V_KONE, V_M + 2):
        pre.append(valu("v_mov_b32_e32 v%d, 0" % v, [], ["v%d" % v]))
    t0 = "v%d" % V_T
    pre.append(valu("v_mov_b32_e32 %s, 0x3f803f80" % t0, [], [t0]))
    pre.append(valu("v_cndmask_b32_e64 v%d, 0, %s, s[%d:%d]" % (V_KONE, t0, S_LO, S_LO + 1), [t0], ["v%d" % V_KONE]))
    pre.append(valu("v_mov_b32_e32 %s, 0x3f80" % t0, [], [t0]))
    for s in range(2):
        pre.append(valu("v_cndmask_b32_e64 v%d, 0, %s, s[%d:%d]" % (V_QM + 4 * s + 1, t0, S_LO, S_LO + 1), [t0], ["v%d" % (V_QM + 4 * s + 1)]))
    pre += dma_offsets()
    pre += read_addresses()
    pre.append(valu("v_mov_b32_e32 v%d, v%d" % (V_KC + 6, V_ADDR + 6), [], ["v%d" % (V_KC + 6)]))
    # A(0) of tile 0 alone, then its maximum / decision
    pre += k_reads(0)
    pre += chain(0, 0)
    pre += max_decide(0, "pre", after=0)

    # ---------------------------------------------------------------- one trip = one 64-key tile
    body = [Op("L_loop_%=:", "raw")]
    # "a next tile exists" mask for the decision taken at the end of this trip
    body += [salu("s_add_u32 s%d, s%d, 1" % (S_TMP, S_T), ["s%d" % S_TMP]),
             salu("s_cmp_lt_u32 s%d, s%d" % (S_TMP, S_NT), ["scc"]),
             salu("s_cselect_b64 s[%d:%d], -1, 0" % (S_NL, S_NL + 1), ["s%d" % S_NL, "s%d" % (S_NL + 1)], ["scc"])]
    # phases 1..6: A(u) || exp/cvt(u-1) + reads, B(u-1) || max(u)
    for u in (1, 2, 3):
        fill = v_reads(u - 1, after=1) + exp_cvt(u - 1)
        if u & 1:
            fill += bias_reload(u >> 1, after=2)
        d = dma_piece(2 * (u - 1) + 2)
        d[0].after = d[1].after = 1
        fill.insert(8, d[0]); fill.insert(9, d[1])
        body += interleave(chain(u, None), fill, lead=k_reads(u))
        d = dma_piece(2 * (u - 1) + 3)
        body += interleave(pv(u - 1), d + max_decide(u, "u%d" % u, after=2))
    # tile boundary: the next tile's pieces (all but the two youngest DMA groups) have landed; publish; move on
    wait = Op("s_nop 0", "salu", needs=["^dma7"])
    body.append(wait)
    body.append(Op("s_barrier", "salu"))
    body += [salu("s_add_u32 s%d, s%d, 0x%x" % (S_RD, S_RD, STAGE), ["s%d" % S_RD]),
             salu("s_cmp_ge_u32 s%d, 0x%x" % (S_RD, NSTAGE * STAGE), ["scc"]),
             salu("s_cselect_b32 s%d, 0, s%d" % (S_RD, S_RD), ["s%d" % S_RD], ["scc"]),
             salu("s_cmp_eq_u32 s%d, 0" % S_T, ["scc"]),
             salu("s_cselect_b64 vcc, -1, 0", ["vcc"], ["scc"])]
    body += read_addresses(k_only=True)
    # mask words: tile 1 -> v47, tiles >= 2 -> the zero region
    body.append(valu("v_cndmask_b32_e32 v%d, v%d, v%d, vcc" % (V_KC + 6, V_ADDR + 7, V_ADDR + 15), ["vcc"], ["v%d" % (V_KC + 6)]))
    body += dma_advance()
    # phase 7: A(0) of the next tile || exp/cvt(3) + reads of V(3) (still the old stage addresses)
    fill = v_reads(3, after=1) + exp_cvt(3)
    d = dma_piece(0)
    d[0].after = d[1].after = 1
    fill.insert(8, d[0]); fill.insert(9, d[1])
    body += interleave(chain(0, None), fill, lead=k_reads(0))
    body += read_addresses(k_only=False)
    d = dma_piece(1)
    body += interleave(pv(3), d + [salu("s_add_u32 s%d, s%d, 0x1000" % (S_BOFF, S_BOFF), ["s%d" % S_BOFF]),
                                   salu("s_add_u32 s%d, s%d, 1" % (S_T, S_T), ["s%d" % S_T])] + max_decide(0, "u0", guard_last=True, after=2))
    body += [salu("s_cmp_lt_u32 s%d, s%d" % (S_T, S_NT), ["scc"]), Op("s_cbranch_scc1 L_loop_%=", "salu")]

    pre_w, body_w = place_waits(pre, body)
    pre_h = pad_hazards(pre_w)
    body_h1 = pad_hazards(body_w, history=pre_h)
    body_h = pad_hazards(body_w, history=body_h1)  # steady state; must not need more than the first trip
    # the first trip follows `pre`, later ones the loop's own tail: use the union (pad where either needs it)
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
    tail = ["s_branch L_done_%="]
    for site, s in (("pre", 0), ("u1", 1), ("u2", 0), ("u3", 1), ("u0", 0)):
        tail += rare_block(site, s)
    tail += ["L_done_%=:", "s_nop 15", "s_nop 15"]
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
    print("loop body: %d instructions, %s; issue-cost estimate %d cycles per tile (%d MFMAs = %d matrix cycles)"
          % (len(body), k, c, nm, 32 * nm))
