"""Small instruction-stream builder for hand-placed gfx950 inline-asm bodies (attention kernels).

What it does for the author of a stream:
  * phases = a backbone of MFMAs with "filler" instructions dealt into the gaps between them by issue cost;
  * counted `s_waitcnt lgkmcnt(n)` / `vmcnt(n)` in front of the first consumer of every LDS / vector-memory result
    (results are named by tags; counts follow from the issue order, loop-carried ones from a two-trip simulation);
  * software wait states (`s_nop`) for the hazards hipcc does not pad inside an asm statement
    (cdna_hip_programming.md 5.7 item 2): MFMA result -> vector / memory instruction, vector write -> MFMA operand,
    transcendental -> vector read, vector write -> v_permlane, M0 write -> LDS-DMA.
Registers are physical and named by the stream's author ("v96", "a3", "s48", "vcc", "m0").
"""
import re


def rr(prefix, lo, n=1):
    """Register-range text: rr('v', 96, 16) -> 'v[96:111]', rr('v', 5) -> 'v5'."""
    return "%s%d" % (prefix, lo) if n == 1 else "%s[%d:%d]" % (prefix, lo, lo + n - 1)


def regs(prefix, lo, n=1):
    return ["%s%d" % (prefix, lo + i) for i in range(n)]


class Op:
    __slots__ = ("text", "kind", "reads", "writes", "cost", "tag", "needs", "after", "srcc")

    def __init__(self, text, kind, reads=(), writes=(), cost=None, tag=None, needs=(), after=0, srcc=()):
        self.text = text
        self.kind = kind  # mfma valu trans lds vmem dma salu perm raw
        self.reads = list(reads)
        self.writes = list(writes)
        self.cost = cost if cost is not None else {"mfma": 8, "valu": 4, "trans": 8, "lds": 8, "vmem": 16, "dma": 56,
                                                  "salu": 4, "perm": 4, "raw": 0}[kind]
        self.tag = tag          # name of the memory result this op produces (lds / vmem / dma)
        self.needs = list(needs)  # tags that must have landed before this op issues
        self.after = after      # filler: not before this many MFMAs of its phase have been issued
        self.srcc = list(srcc)  # MFMA: registers read as the C operand (accumulate chains need no wait states)


def expand(txt):
    """'v[96:111]' / 'a[0:3]' / 'v5' / 's[40:43]' -> list of single registers."""
    m = re.fullmatch(r"([vas])\[(\d+):(\d+)\]", txt)
    if m:
        return ["%s%d" % (m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)]
    return [txt]


def mfma(op, dst, a, b, c):
    rd = expand(a) + expand(b)
    cc = [] if c == "0" else expand(c)
    return Op("%s %s, %s, %s, %s" % (op, dst, a, b, c), "mfma", rd + cc, expand(dst), srcc=cc)


def interleave(mfmas, fillers, lead=()):
    """Deal `fillers` (program order kept) into the gaps after each MFMA, by cost, honouring Op.after."""
    out = list(lead)
    rest = list(fillers)
    n = len(mfmas)
    for i, m in enumerate(mfmas):
        out.append(m)
        gaps_left = n - i
        budget = sum(f.cost for f in rest) / float(gaps_left)
        used = 0.0
        while rest and rest[0].after <= i + 1 and (used < budget - 1e-6 or gaps_left == 1):
            f = rest.pop(0)
            out.append(f)
            used += f.cost
    assert not rest, "fillers left over (after-constraint beyond the phase's MFMAs?)"
    return out


# ---- wait counts -------------------------------------------------------------------------------------------------------
LGKM_KINDS = ("lds",)
VM_KINDS = ("vmem", "dma")


def place_waits(pre, body, vm_history=0):
    """Insert counted waits.  `pre` runs once, `body` is the loop (its ops may need tags produced in the previous trip).
    Returns (pre_out, body_out).  LDS tags must be produced and consumed in the same trip (asserted)."""

    lds_tags = set(o.tag for o in list(pre) + list(body) if o.kind in LGKM_KINDS and o.tag)
    vm_tags = set(o.tag for o in list(pre) + list(body) if o.kind in VM_KINDS and o.tag)

    def last_index(q, t, skip):
        idx = [i for i, x in enumerate(q) if x == t]
        if len(idx) <= skip:
            return None
        return idx[-1 - skip]

    def run(stream, lgkm_q, vm_q, out, lenient):
        for op in stream:
            if op.needs:
                wl = wv = None
                for t0 in op.needs:
                    skip = 1 if t0.startswith("^") else 0  # "^tag": the occurrence BEFORE the most recent one
                    t = t0.lstrip("^")
                    if t in lds_tags:
                        i = last_index(lgkm_q, t, skip)
                        if i is None:
                            continue  # an earlier wait of this trip already covered it (LDS results return in order)
                        k = len(lgkm_q) - 1 - i
                        wl = k if wl is None else min(wl, k)
                    elif t in vm_tags:
                        i = last_index(vm_q, t, skip)
                        if i is None:
                            continue  # landed long ago (a wait since then covered it) or issued by the caller and drained
                        k = len(vm_q) - 1 - i
                        wv = k if wv is None else min(wv, k)
                    else:
                        assert lenient, "tag %s is produced nowhere (%s)" % (t, op.text)
                parts = []
                if wv is not None:
                    assert wv <= 63
                    parts.append("vmcnt(%d)" % wv)
                    del vm_q[: len(vm_q) - wv]  # everything older has landed
                if wl is not None:
                    wl = min(wl, 15)
                    parts.append("lgkmcnt(%d)" % wl)
                    del lgkm_q[: len(lgkm_q) - wl]
                if parts:
                    out.append(Op("s_waitcnt " + " ".join(parts), "salu", cost=4))
            out.append(op)
            if op.kind in LGKM_KINDS:
                lgkm_q.append(op.tag or "_")
            elif op.kind in VM_KINDS:
                vm_q.append(op.tag or "_")

    pre_out = []
    run(pre, [], [], pre_out, True)
    # steady state: simulate one trip to build the vector-memory queue a later trip sees, then emit such a trip
    vm1 = []
    run(body, [], vm1, [], True)
    vm1b = list(vm1)
    run(body, [], vm1b, [], True)
    body_out = []
    run(body, [], vm1b, body_out, True)
    return pre_out, body_out


# ---- hazards -----------------------------------------------------------------------------------------------------------
MFMA_TO_OTHER = 13  # 8-pass XDL result -> VALU / memory read or write of the register (guide: 12 states; one spare)
VALU_TO_MFMA = 2
TRANS_TO_VALU = 1
VALU_TO_PERM = 2
M0_TO_DMA = 1


def states_of(op):
    m = re.match(r"s_nop (\d+)", op.text)
    if m:
        return int(m.group(1)) + 1
    return 0 if op.kind == "raw" else 1


def pad_hazards(stream, history=()):
    """Insert s_nop so that every (producer, consumer) pair has its wait states.  `history` = ops issued before the stream."""
    out = []
    past = list(history)[-24:]

    def need_for(op):
        need = 0
        dist = 0
        for prev in reversed(past):
            st = states_of(prev)
            if prev.kind != "raw":
                req = 0
                w = set(prev.writes)
                if w:
                    touched = w & (set(op.reads) | set(op.writes))
                    if prev.kind == "mfma" and touched:
                        if op.kind == "mfma":
                            # accumulate chain (same registers as C, whole tuple): free; as A/B operand: full wait
                            if not (touched <= set(op.srcc)) or (set(op.writes) & w and not (w <= set(op.srcc))):
                                req = MFMA_TO_OTHER
                        else:
                            req = MFMA_TO_OTHER
                    elif prev.kind in ("valu", "trans", "perm") and w & set(op.reads) and op.kind == "mfma":
                        req = VALU_TO_MFMA
                    elif prev.kind == "trans" and w & set(op.reads) and op.kind in ("valu", "perm"):
                        req = TRANS_TO_VALU
                    if prev.kind in ("valu", "trans") and op.kind == "perm" and w & (set(op.reads) | set(op.writes)):
                        req = max(req, VALU_TO_PERM)
                    if prev.kind == "salu" and "m0" in w and op.kind == "dma":
                        req = max(req, M0_TO_DMA)
                need = max(need, req - dist)
            dist += st
            if dist > 16:
                break
        return need

    for op in stream:
        n = need_for(op)
        while n > 0:
            k = min(n, 16)
            nop = Op("s_nop %d" % (k - 1), "salu", cost=4 * k)
            out.append(nop)
            past.append(nop)
            n -= k
        out.append(op)
        past.append(op)
        past = past[-24:]
    return out


def emit(stream, indent="  "):
    return "".join("%s%s\n" % ("" if o.text.endswith(":") else indent, o.text) for o in stream)
