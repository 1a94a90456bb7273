#!/usr/bin/env python3
"""Generates llamarec_amd/csrc/llama_attn256_body.inc: the hand-placed instruction stream of ONE key block of
attn_mfma256_kernel (llama_attn256.hip) -- 64 v_mfma_f32_32x32x16_bf16 with every other instruction assigned to one of
the 64 MFMA gaps by the tables below (cdna_hip_programming.md, 'Fused attention prefill', 4-wave structure: <= 5
single-issue fillers per gap, at most one v_exp_f32).

A wave owns 64 query rows = two 32-row halves A (0) and B (1); a key block is 64 keys = two 32-key tiles kt.
Asm-owned accumulator registers (named literally; the compiler never sees them):
    O[half][dt]  a[(4 half + dt) 16 .. +15]      O^T tile: rows d = 32 dt .., column = query row (lane & 31)
    Q[half][ks]  a[128 + (8 half + ks) 4 .. +3]  B operand of S^T = K Q^T, k-step ks (16 of the 128 dims)
    K[kt][ks]    a[192 + (8 kt + ks) 4 .. +3]    A operand
Compiler-allocated VGPRs: S[half][kt] (f32 x 16), Pf[half][s] (4 words of packed bf16 pairs: the B operand of k-step s),
Vf[dt][s] (the V^T A operand of k-step s: two transposed 64-bit reads).

Gap schedule of block kb (the two halves run half a block apart; B's softmax wraps into the next block's gaps):
    g  0..15  QK(A, kb)       | fill: E-phase of B (block kb-1), DMA issue, then A's row maximum as S[A][0] completes
    g 16..31  PV(B, kb-1)     | fill: A maximum + bookkeeping, E-phase of A, V^T fragment reads of block kb
    g 32..47  QK(B, kb)       | fill: E-phase of A, V^T reads, B maximum
    g 48..63  PV(A, kb)       | fill: B maximum + bookkeeping, E-phase of B, K fragment reads of block kb + 1
E-phase of score i: fma (scale, subtract the row's reference maximum) at gap e0+i, v_exp_f32 at e0+i+1, row-sum add at
e0+i+2, v_cvt_pk_bf16_f32 of a pair behind its second add: one exp per gap, every dependent pair a gap apart.
"""
import os
import sys

OUT = os.environ.get("A2_OUT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "llamarec_amd", "csrc", "llama_attn256_body.inc")
# timing-only ablation builds (tools/gpu_attn256_abl.sh; results are WRONG): a comma list of
#   nodma notr nokr noe nomax nobk nowait mfmaonly
ABL = set(x for x in os.environ.get("A2_ABL", "").split(",") if x)


def ablated(ins):
    t = ins.text or ""
    if "mfmaonly" in ABL:
        return True
    if "nodma" in ABL and "buffer_load" in t:
        return True
    if "notr" in ABL and "ds_read_b64_tr" in t:
        return True
    if "nokr" in ABL and "ds_read_b128" in t:
        return True
    if "noe" in ABL and (t.startswith("v_fmamk") or t.startswith("v_exp_f32 {0}, {1}") or t.startswith("v_add_f32 {0}, {0}") or t.startswith("v_cvt_pk")):
        return True
    if "nomax" in ABL and (t.startswith("v_max3") or t.startswith("v_max_f32 {0}, {0}")):
        return True
    if "nobk" in ABL and ins.kind == "bk":
        return True
    if "nowait" in ABL and ins.kind == "wait":
        return True
    return False

NG = 64


def areg(lo, n):
    return "a[%d:%d]" % (lo, lo + n - 1)


def clob(lo, n):
    return "A2_ALLA"   # every statement names the WHOLE accumulator file as clobbered: hipcc then cannot park a value of its
                       # own in an accumulator register across any of them (it does, given the chance: llama_attn256.hip)


# gaps of the block's 8 LDS-DMA pieces (K, V alternating). A2_DMA_GAPS="g0,g1,..,g7" overrides (tools/build_attn256_dma.sh). A piece
# costs the stream ~48 cycles where eight follow each other (gaps 1..8: the first placement); every other gap behind the transposed V
# reads they are nearly free: 4 x 8 192 tokens 2 183 -> 2 095 us, 16 x 1 000 tokens 276 -> 266 us, same box (round 5). hipBLASLt's
# 256 x 256 GEMM spreads its 16 pieces over the K tile the same way.
DMA_GAPS = [int(x) for x in os.environ.get("A2_DMA_GAPS", "41,43,45,47,49,51,53,55").split(",")]
assert len(DMA_GAPS) == 8 and all(0 <= g <= 62 for g in DMA_GAPS)


def O_(h, dt):
    return (4 * h + dt) * 16


def Q_(h, ks):
    return 128 + (8 * h + ks) * 4


def K_(kt, ks):
    return 192 + (8 * kt + ks) * 4


class Ins:
    """One instruction of a gap: text with {0} {1} .. placeholders and its operands (mode, C expression).
    mode: 'o' written, 'i' read, 'io' both (VGPR); 's' SGPR read; 'n' integer constant."""
    def __init__(self, gap, text, operands=(), cond="", kind="valu", prio=5, raw=None):
        self.gap, self.text, self.operands, self.cond, self.kind, self.prio, self.raw = gap, text, list(operands), cond, kind, prio, raw


def build():
    mf = {}      # gap -> Ins (the MFMA)
    ops = []     # fillers

    def add(gap, text, operands=(), cond="", kind="valu", prio=5, raw=None):
        ops.append(Ins(gap, text, operands, cond, kind, prio, raw))

    # ---------------- MFMAs
    for h, g0 in ((0, 0), (1, 32)):
        for kt in range(2):
            for ks in range(8):
                g = g0 + 8 * kt + ks
                S = "S[%d][%d]" % (h, kt)
                if ks == 0:
                    mf[g] = Ins(g, "v_mfma_f32_32x32x16_bf16 {0}, %s, %s, 0" % (areg(K_(kt, ks), 4), areg(Q_(h, ks), 4)), [("o", S)])
                    mf[g].diag_text = "v_mfma_f32_32x32x16_bf16 {0}, %s, %s, {0}" % (areg(K_(kt, ks), 4), areg(Q_(h, ks), 4))
                else:
                    mf[g] = Ins(g, "v_mfma_f32_32x32x16_bf16 {0}, %s, %s, {0}" % (areg(K_(kt, ks), 4), areg(Q_(h, ks), 4)), [("io", S)])
    for h, g0 in ((1, 16), (0, 48)):
        for s in range(4):
            for dt in range(4):
                g = g0 + 4 * s + dt
                o = areg(O_(h, dt), 16)
                opr = [("i", "Vf[%d][%d]" % (dt, s)), ("i", "Pf[%d][%d]" % (h, s))]
                m = Ins(g, "v_mfma_f32_32x32x16_bf16 %s, {0}, {1}, %s" % (o, o), opr, cond="!FIRST" if h == 1 else "")
                if h == 0 and s == 0:   # O[A] of a tile starts at 0: its first PV takes C = 0
                    m.first_text = "v_mfma_f32_32x32x16_bf16 %s, {0}, {1}, 0" % o
                mf[g] = m

    # ---------------- softmax of half h; gb = gap of the first QK MFMA of that half (0 / 32)
    for h, gb in ((0, 0), (1, 32)):
        # causal mask (DIAG blocks only): the accumulator tile STARTS at 0 / -inf (key c of the block > thr[h]: masked) and the
        # first QK MFMA accumulates onto it. (Masking the finished scores in place made hipcc copy accumulator elements
        # around the statement -- compiler code that reads an asm MFMA's result with no wait states.) The tile's registers are
        # free by then: S[h][kt] was last read by the previous block's E-phase.
        for kt in range(2):
            g0 = (-1, 0, 24, 32)[2 * h + kt]      # -1: in front of the block's first MFMA
            for rr in range(16):
                c = 32 * kt + (rr & 3) + 8 * (rr >> 2)
                add(g0 + (rr >> 1) if g0 >= 0 else -1, "v_cmp_gt_i32 vcc, %d, {1}\n\tv_cndmask_b32 {0}, 0, {2}, vcc" % c,
                    [("o", "S[%d][%d][%d]" % (h, kt, rr)), ("i", "thr[%d]" % h), ("i", "ninf")], "DIAG", kind="mask", prio=1)
        # row maximum (this lane's 32 keys of the row): one chain over both tiles, 8 v_max3 each; S[h][0] is complete after gap
        # gb+7, S[h][1] after gb+15, and an asm MFMA's result may be read two gaps later at the earliest (nothing pads it)
        H = h
        mx = "mx[%d]" % H
        v0, v1 = "S[%d][0]" % H, "S[%d][1]" % H
        seq = [("v_max3_f32 {0}, {1}, {2}, {3}", [("o", mx), ("i", v0 + "[0]"), ("i", v0 + "[1]"), ("i", v0 + "[2]")])]
        for j in range(1, 7):
            seq.append(("v_max3_f32 {0}, {0}, {1}, {2}", [("io", mx), ("i", "%s[%d]" % (v0, 2 * j + 1)), ("i", "%s[%d]" % (v0, 2 * j + 2))]))
        seq.append(("v_max_f32 {0}, {0}, {1}", [("io", mx), ("i", v0 + "[15]")]))
        for n, (t, o) in enumerate(seq):
            add(gb + 9 + n, t, o, prio=2)
        for j in range(8):
            add(gb + 17 + j // 2, "v_max3_f32 {0}, {0}, {1}, {2}", [("io", mx), ("i", "%s[%d]" % (v1, 2 * j)), ("i", "%s[%d]" % (v1, 2 * j + 1))], prio=2)
        # Deferred maximum. Common path: does this lane's maximum exceed the row's cached raw threshold (m + 2^THR) / scale? The two
        # lanes of a row are combined on the scalar side; nothing else happens while no row of the half moves. Rare path (always in
        # a tile's first block): the row maximum across the lane halves, m' = need ? max * scale : m, alpha = 2^(m - m'), -m', the
        # new threshold, the row sum and O[half] (not in a first block: O starts there) times alpha. alpha = 1 exactly for the rows
        # that keep their m, so the rows of a tile do not couple.
        g = gb + 21
        rare = ["v_cmp_gt_f32 vcc, {0}, {1}", "s_or_b32 vcc_lo, vcc_lo, vcc_hi", "s_mov_b32 vcc_hi, vcc_lo",
                "s_mov_b64 vcc, vcc",            # gfx9: a write of one half of VCC leaves VCCZ stale
                "s_cbranch_vccz .La2_keep%d_%%=" % H,
                "v_mov_b32 {2}, {0}", "v_mov_b32 {3}, {0}", "s_nop 1", "v_permlane32_swap_b32 {2}, {3}", "s_nop 1", "v_max_f32 {2}, {2}, {3}",
                "v_mul_f32 {2}, 0x3e0293ee, {2}",            # scaled row maximum
                "v_cndmask_b32 {2}, {4}, {2}, vcc",           # m'
                "v_sub_f32 {3}, {4}, {2}", "v_exp_f32 {5}, {3}", "v_mov_b32 {4}, {2}", "v_xor_b32 {6}, 0x80000000, {2}",
                "v_add_f32 {3}, 0x41000000, {2}", "v_mul_f32 {1}, 0x40fae54b, {3}",   # (m' + 8) / scale: 1 / scale = 7.84316
                "s_nop 0", "v_mul_f32 {7}, {7}, {5}"]
        opr = [("i", mx), ("io", "mthr[%d]" % H), ("o", "bt_[0]"), ("o", "bt_[1]"), ("io", "m_run[%d]" % H), ("o", "alpha[%d]" % H),
               ("io", "negm[%d]" % H), ("io", "lsum[%d]" % H)]
        resc = []
        for c in range(8):
            lo = O_(H, 0) + 8 * c
            resc += ["v_accvgpr_read_b32 {%d}, a%d" % (8 + i, lo + i) for i in range(8)]
            resc += ["v_mul_f32 {%d}, {5}, {%d}" % (8 + i, 8 + i) for i in range(8)]
            resc += ["v_accvgpr_write_b32 a%d, {%d}" % (lo + i, 8 + i) for i in range(8)]
        ropr = [("o", "rt_[%d]" % i) for i in range(8)]
        tail = [".La2_keep%d_%%=:" % H]
        bk = Ins(g, "\n\t".join(rare + resc + tail), opr + ropr, kind="bk", prio=3)
        bk.first_text = "\n\t".join(rare + tail)
        bk.first_operands = opr
        ops.append(bk)
        # E-phase (temporaries rotate: t lives one gap, p at most three)
        e0 = gb + 23
        for i in range(32):
            kt, rr = i >> 4, i & 15
            t, pp, pm = "t_[%d][%d]" % (h, i % 2), "p_[%d][%d]" % (h, i % 4), "p_[%d][%d]" % (h, (i - 1) % 4)
            add(e0 + i, "v_fmamk_f32 {0}, {1}, 0x3e0293ee, {2}", [("o", t), ("i", "S[%d][%d][%d]" % (h, kt, rr)), ("i", "negm[%d]" % h)], prio=6)
            add(e0 + i + 1, "v_exp_f32 {0}, {1}", [("o", pp), ("i", t)], kind="exp", prio=4)
            add(e0 + i + 2, "v_add_f32 {0}, {0}, {1}", [("io", "lsum[%d]" % h), ("i", pp)], prio=7)
            if i & 1:
                add(e0 + i + 2, "v_cvt_pk_bf16_f32 {0}, {1}, {2}",
                    [("o", "Pf[%d][%d][%d]" % (h, i >> 3, (i >> 1) & 3)), ("i", pm), ("i", pp)], prio=8)

    # ---------------- V^T fragment reads of block kb (fragment f = 4 s + dt is free after PV(B, kb-1)'s MFMA at gap 16 + f)
    for n in range(32):
        f, jj = n >> 1, n & 1
        s, dt = f >> 2, f & 3
        g = 23 + (n * 18) // 32
        assert g >= 16 + f + 2 and g <= 46
        add(g, "ds_read_b64_tr_b16 {0}, {1} offset:{2}", [("o", "Vf[%d][%d][%d]" % (dt, s, jj)), ("i", "vaddr_c"), ("n", "%d" % (4096 * dt + 1024 * s + 512 * jj))],
            kind="lds", prio=5)
    # every V^T fragment has landed before PV(A) starts (the K reads of gaps 56.. come later)
    add(47, "s_waitcnt lgkmcnt(0)", [], kind="wait", prio=9)
    # ---------------- K fragment reads of block kb + 1 (K[kt][ks] is free after QK(B)'s MFMA at gap 32 + 8 kt + ks)
    for n in range(16):
        kt, ks = n >> 3, n & 7
        g = 56 + n // 2
        assert g >= 32 + 8 * kt + ks + 2
        add(g, "ds_read_b128 %s, {0} offset:{1}" % areg(K_(kt, ks), 4), [("i", "kaddr_n"), ("n", "%d" % (kt * 8192 + ks * 512))], kind="lds", prio=5)
    # ---------------- LDS-DMA of K(kb + 2) and V(kb + 1): early in the block, in the lightest gaps
    dma = "s_mov_b32 m0, {0}\n\ts_nop 0\n\tbuffer_load_dwordx4 {1}, {2}, {3} offen lds"
    for i in range(4):
        add(DMA_GAPS[2 * i], dma, [("s", "dst_k + %d" % (1024 * i)), ("i", "koff_x"), ("s", "rs_k"), ("s", "soff[%d]" % i)], kind="dma", prio=9)
        add(DMA_GAPS[2 * i + 1], dma, [("s", "dst_v + %d" % (4096 * i)), ("i", "voff_x"), ("s", "rs_v"), ("s", "soff[%d]" % i)], kind="dma", prio=9)
    if os.environ.get("A2_STAMPS"):   # diagnostic build (tools/build_attn256_abl.sh stamps): where a block's cycles go
        for k in range(8):
            add(8 * k, "s_memtime {0}", [("so", "st_[%d]" % k)], kind="stamp", prio=0)
        if os.environ.get("A2_STAMPS") == "kwait":   # a first block's K(1) fragment reads, waited for in front of the last stamp
            add(63, "s_waitcnt lgkmcnt(0)", [], "FIRST", kind="stamp", prio=98)
        add(63, "s_memtime {0}", [("so", "st_[8]")], kind="stamp", prio=99)
    # O[B] of a tile starts at zero: written in the FIRST block, whose PV(B) gaps carry no MFMA
    for dt in range(4):
        o = O_(1, dt)
        add(16 + dt, "\n\t".join("v_accvgpr_write_b32 a%d, 0" % (o + i) for i in range(16)), [], "FIRST", kind="init", prio=5)
    return mf, ops


def merged_statement(inss):
    """ONE asm volatile statement for a list of instructions (hipcc pads every asm statement with an s_nop: a statement per
    instruction doubled the stream). Operands are unified by expression; an output is early-clobber unless it is also read."""
    exprs, modes, first = [], {}, {}
    for ins in inss:
        # reads of an instruction happen before its writes: order an instruction's operands reads first
        for mode, e in sorted(ins.operands, key=lambda t: 0 if t[0] in ("i", "io", "s", "n") else 1):
            if mode == "so":
                first[e] = "o"
            if e not in modes:
                exprs.append(e)
                modes[e] = set()
                first[e] = mode
            modes[e].add(mode)
    outs = [e for e in exprs if modes[e] & {"o", "io", "so"}]
    ins_ = [e for e in exprs if not (modes[e] & {"o", "io", "so"})]
    order = outs + ins_
    idx = {e: n for n, e in enumerate(order)}
    lines = []
    for ins in inss:
        t = ins.text
        for n, (mode, e) in reversed(list(enumerate(ins.operands))):
            t = t.replace("{%d}" % n, "%%%d" % idx[e])
        lines.append(t)
    text = "\\n\\t".join(l.replace("\n\t", "\\n\\t") for l in lines)
    text = text.replace(":\\n\\t", ":\\n\\t")
    def con(e):
        m = modes[e]
        if ("io" in m or "o" in m) and first[e] == "o":
            return '"=&v"(%s)' % e        # written before it is read: the statement does not need the incoming value
        if "io" in m or ("o" in m and "i" in m):
            return '"+&v"(%s)' % e       # early-clobber too: a plain "+v" may share its register with another INPUT that holds
                                         # the same value (seen: ninf and a reference maximum still at -inf), which the statement
                                         # then overwrites before its later instructions read that input
        if "o" in m:
            return '"=&v"(%s)' % e
        if "so" in m:
            return '"=&s"(%s)' % e
        if "s" in m:
            return '"s"(%s)' % e
        if "n" in m:
            return '"i"(%s)' % e
        return '"v"(%s)' % e
    clobbers = ['"memory"'] if any(i.kind in ("wait",) for i in inss) else []
    if any("vcc" in i.text for i in inss):
        clobbers.append('"vcc"')
    if any("s_or_b32" in i.text for i in inss):
        clobbers.append('"scc"')          # hipcc keeps loop conditions in SCC across a statement that does not name it (seen: a hang)
    clobbers.append("A2_ALLA")
    return 'asm volatile("%s" : %s : %s : %s);' % (text, ", ".join(con(e) for e in outs), ", ".join(con(e) for e in ins_), ", ".join(clobbers))


import re
TUPLES = ("S", "Pf", "Vf")
MAXGAPS = int(os.environ.get("A2_MAXGAPS", "4"))   # gaps per asm statement (register allocation degrades with very long ones)


def tuple_use(e):
    """(tuple name, 'whole' / 'elem') of an operand expression, or None."""
    m = re.match(r"^(S|Pf|Vf)\[(\d)\]\[(\d)\](\[\d+\])?$", e)
    if not m:
        return None
    return ("%s[%s][%s]" % (m.group(1), m.group(2), m.group(3)), "elem" if m.group(4) else "whole")


def emit(mf, ops, lo, hi, first, diag, drain, f):
    """gaps lo..hi-1 of one variant. drain: what a wave still owes after its last block (B's wrapped ops + PV(B)).
    Consecutive gaps are merged into ONE asm statement for as long as no register tuple would be named both whole (an MFMA
    operand) and by element (a vector instruction's operand) in it -- the compiler would see two unrelated values."""
    by_gap = {}
    for o in ops:
        g = o.gap
        wrapped = g >= NG
        if wrapped:
            g -= NG
        if drain != wrapped and drain:
            continue
        if drain and g < 0:
            continue
        if not drain and wrapped and first:
            continue                      # nothing wraps into a tile's first block
        if o.cond == "DIAG" and not diag:
            continue
        if o.cond == "FIRST" and not first:
            continue
        if o.cond == "!FIRST" and first:
            continue
        if ABL and ablated(o):
            continue
        if o.kind == "dma" and first:
            continue     # a tile's first block: its K(2) / V(1) were requested behind the PREVIOUS tile's last barrier (the kernel's
                         # pre_issue), in front of that tile's output stores, so that this block's wait need not cover those stores
        if o.kind == "stamp" and (diag or drain):
            continue     # only the steady-state variant waits for the stamps (an s_memtime landing in a register hipcc has reused
                         # since -- the outputs are dead in the other variants -- overwrote an address: a memory fault)
        if first and hasattr(o, "first_text"):
            o = Ins(o.gap, o.first_text, o.first_operands, kind=o.kind, prio=o.prio)
        by_gap.setdefault(g, []).append(o)
    group, uses, ngaps = [], {}, 0

    def flush():
        nonlocal group, uses, ngaps
        if group:
            f.write(merged_statement(group) + "\n")
        group, uses, ngaps = [], {}, 0

    def push(ins):
        nonlocal ngaps
        for mode, e in ins.operands:
            tu = tuple_use(e)
            if tu and uses.get(tu[0], tu[1]) != tu[1]:
                flush()
                break
        for mode, e in ins.operands:
            tu = tuple_use(e)
            if tu:
                uses[tu[0]] = tu[1]
        group.append(ins)

    for o in sorted(by_gap.get(-1, []), key=lambda t: t.prio):
        push(o)
    for g in range(lo, hi):
        if g in mf:
            m = mf[g]
            take = (m.cond == "!FIRST") if drain else not (m.cond == "!FIRST" and first)
            if take:
                if first and hasattr(m, "first_text"):
                    m = Ins(g, m.first_text, m.operands)
                if diag and hasattr(m, "diag_text"):
                    m = Ins(g, m.diag_text, [("io", m.operands[0][1])])
                m2 = Ins(g, "; ---- gap %d\n\t" % g + m.text, m.operands)
                if ngaps >= MAXGAPS:
                    flush()
                ngaps += 1
                push(m2)
        for o in sorted(by_gap.get(g, []), key=lambda t: t.prio):
            if o.raw is not None:           # a C++ statement: closes the running asm statement
                flush()
                f.write(o.raw + "\n")
            else:
                push(o)
    flush()


def report(ops):
    load = [[0, 0, 0] for _ in range(NG)]   # fillers, exps, diag-only
    for o in ops:
        g = o.gap % NG
        if o.cond == "DIAG":
            load[g][2] += 2
            continue
        if o.cond == "FIRST":
            continue
        load[g][0] += 1
        if o.kind == "exp":
            load[g][1] += 1
    sys.stderr.write("gap: fillers (exp) [+diag]\n")
    for g in range(NG):
        sys.stderr.write("%2d: %d (%d) [+%d]\n" % (g, load[g][0], load[g][1], load[g][2]))
    sys.stderr.write("total fillers %d = %.2f per gap\n" % (sum(l[0] for l in load), sum(l[0] for l in load) / NG))


def emit_static(f):
    # Q fragments of a tile: global memory -> a[128:191] (lane: row 32 half + (l & 31), 16 bytes at 32 ks + 16 (l >> 5))
    f.write("#ifdef A2_EMIT_QGLOAD\n")
    for h in range(2):
        for ks in range(8):
            f.write('asm volatile("global_load_dwordx4 %s, %%0, %%1 offset:%d" :: "v"(qvo[%d]), "s"(qkv) : "memory", %s);\n'
                    % (areg(Q_(h, ks), 4), ks * 32, h, clob(Q_(h, ks), 4)))
    f.write("#endif\n#ifdef A2_EMIT_KLOAD\n")   # K fragments of the block in the ring slot kaddr_n points at
    for kt in range(2):
        for ks in range(8):
            f.write('asm volatile("ds_read_b128 %s, %%0 offset:%d" :: "v"(kaddr_n) : %s);\n'
                    % (areg(K_(kt, ks), 4), kt * 8192 + ks * 512, clob(K_(kt, ks), 4)))
    f.write("#endif\n")
    # epilogue: one O^T tile (16 registers) at a time into ov[], A2_OSTORE(half, dt) consumes it; one section per 32-row half
    for h in range(2):
        f.write("#ifdef A2_EMIT_OREAD_%d\n" % h)
        for dt in range(4):
            o = O_(h, dt)
            rd = "\\n\\t".join("v_accvgpr_read_b32 %%%d, a%d" % (i, o + i) for i in range(16))
            outs = ", ".join('"=v"(ov[%d])' % i for i in range(16))
            f.write('{ asm volatile("%s" : %s :: A2_ALLA); A2_OSTORE(%d, %d) }\n' % (rd, outs, h, dt))
        f.write("#endif\n")


def main():
    mf, ops = build()
    last_wrapped = max(o.gap for o in ops) - NG
    with open(OUT, "w") as f:
        f.write("// GENERATED by tools/gen_attn256.py -- do not edit. One key block of attn_mfma256_kernel.\n")
        for first in (0, 1):
            for diag in (0, 1):
                f.write("#ifdef A2_EMIT_BODY_%d%d\n" % (first, diag))
                emit(mf, ops, 0, NG, bool(first), bool(diag), False, f)
                f.write("#endif\n")
        f.write("#ifdef A2_EMIT_DRAIN\n")
        emit(mf, ops, 0, max(32, last_wrapped + 1), False, False, True, f)
        f.write("#endif\n")
        emit_static(f)
    if "-v" in sys.argv:
        report(ops)


if __name__ == "__main__":
    main()
