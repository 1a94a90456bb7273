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

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "llamarec_amd", "csrc", "llama_attn256_body.inc")

NG = 64


def areg(lo, n):
    return "a[%d:%d]" % (lo, lo + n - 1)


def clob(lo, n):
    return "A2_ALLA"   # every statement names the WHOLE accumulator file as clobbered: hipcc then cannot park a value of its
                       # own in an accumulator register across any of them (it does, given the chance: llama_attn256.hip)


def O_(h, dt):
    return (4 * h + dt) * 16


def Q_(h, ks):
    return 128 + (8 * h + ks) * 4


def K_(kt, ks):
    return 192 + (8 * kt + ks) * 4


class Op:
    def __init__(self, gap, code, cond="", kind="valu", prio=5):
        self.gap, self.code, self.cond, self.kind, self.prio = gap, code, cond, kind, prio


def build():
    mf = {}      # gap -> (code, cond)
    ops = []     # fillers

    def add(gap, code, cond="", kind="valu", prio=5):
        ops.append(Op(gap, code, cond, kind, prio))

    # ---------------- MFMAs
    for h, g0 in ((0, 0), (1, 32)):
        for kt in range(2):
            for ks in range(8):
                g = g0 + 8 * kt + ks
                c = "0" if ks == 0 else "%0"
                con = '"=v"' if ks == 0 else '"+v"'
                mf[g] = ('asm volatile("v_mfma_f32_32x32x16_bf16 %%0, %s, %s, %s" : %s(S[%d][%d]));'
                         % (areg(K_(kt, ks), 4), areg(Q_(h, ks), 4), c, con, h, kt), "")
                mf[g] = (mf[g][0].replace("));", ") :: A2_ALLA);"), "")
    for h, g0 in ((1, 16), (0, 48)):
        for s in range(4):
            for dt in range(4):
                g = g0 + 4 * s + dt
                o = O_(h, dt)
                acc = ('asm volatile("v_mfma_f32_32x32x16_bf16 %s, %%0, %%1, %s" :: "v"(Vf[%d][%d]), "v"(Pf[%d][%d]) : %s);'
                       % (areg(o, 16), areg(o, 16), dt, s, h, s, clob(o, 16)))
                ini = ('asm volatile("v_mfma_f32_32x32x16_bf16 %s, %%0, %%1, 0" :: "v"(Vf[%d][%d]), "v"(Pf[%d][%d]) : %s);'
                       % (areg(o, 16), dt, s, h, s, clob(o, 16)))
                if h == 1:
                    mf[g] = (acc, "!FIRST")            # PV(B, kb-1): nothing to do in a tile's first block
                elif s == 0:
                    mf[g] = ("if constexpr (FIRST) { %s } else { %s }" % (ini, acc), "")   # O[A] starts at 0
                else:
                    mf[g] = (acc, "")

    # ---------------- softmax of half h; gb = gap of the first QK MFMA of that half (0 / 32). Every instruction is an
    # asm volatile statement: the order below IS the order in the kernel (hipcc allocates registers, nothing else)
    for h, gb in ((0, 0), (1, 32)):
        H = str(h)
        # causal mask (DIAG blocks only), right in front of the maxima that read the tile: key c of the block > thr[h]
        for kt in range(2):
            gm = gb + 9 + 8 * kt
            for rr in range(16):
                c = 32 * kt + (rr & 3) + 8 * (rr >> 2)
                add(gm + (rr >> 3), 'asm volatile("v_cmp_gt_i32 vcc, %d, %%1\\n\\tv_cndmask_b32 %%0, %%0, %%2, vcc" '
                    ': "+v"(S[%s][%d][%d]) : "v"(thr[%s]), "v"(ninf) : "vcc", A2_ALLA);' % (c, H, kt, rr, H), "DIAG", prio=1)
        # row maximum: 8 v_max3 / v_max per 16-value tile; S[h][0] is complete after gap gb+7, S[h][1] after gb+15 and an
        # asm MFMA's result may be read two gaps later at the earliest (the compiler pads nothing for an asm statement)
        for kt, g_first, per_gap in ((0, gb + 9, 1), (1, gb + 17, 2)):
            v = "S[%s][%d]" % (H, kt)
            mx = "mx%d[%s]" % (kt, H)
            seq = ['asm volatile("v_max3_f32 %%0, %%1, %%2, %%3" : "=v"(%s) : "v"(%s[0]), "v"(%s[1]), "v"(%s[2]) : A2_ALLA);' % (mx, v, v, v)]
            for j in range(1, 7):
                seq.append('asm volatile("v_max3_f32 %%0, %%0, %%1, %%2" : "+v"(%s) : "v"(%s[%d]), "v"(%s[%d]) : A2_ALLA);' % (mx, v, 2 * j + 1, v, 2 * j + 2))
            seq.append('asm volatile("v_max_f32 %%0, %%0, %%1" : "+v"(%s) : "v"(%s[15]) : A2_ALLA);' % (mx, v))
            for n, code in enumerate(seq):
                add(g_first + n // per_gap, code, prio=2)
        g = gb + 21
        add(g, "A2_BK0(%s)" % H, prio=2)        # max of the two tiles, across the lane halves (permlane32 swap)
        add(g + 1, "A2_BK1(%s)" % H, prio=2)    # scaled maximum, deferred-maximum decision, alpha, -m, row-sum start
        add(g + 1, "if constexpr (!FIRST) { A2_RESCALE(%s) }" % H, prio=3)
        # E-phase
        e0 = gb + 23
        for i in range(32):
            kt, rr = i >> 4, i & 15
            add(e0 + i, 'asm volatile("v_fmamk_f32 %%0, %%1, 0x3e0293ee, %%2" : "=v"(t_[%s][%d]) : "v"(S[%s][%d][%d]), "v"(negm[%s]) : A2_ALLA);'
                % (H, i, H, kt, rr, H), prio=6)
            add(e0 + i + 1, 'asm volatile("v_exp_f32 %%0, %%1" : "=v"(p_[%s][%d]) : "v"(t_[%s][%d]) : A2_ALLA);' % (H, i, H, i), kind="exp", prio=4)
            add(e0 + i + 2, 'asm volatile("v_add_f32 %%0, %%0, %%1" : "+v"(lsum[%s]) : "v"(p_[%s][%d]) : A2_ALLA);' % (H, H, i), prio=7)
            if i & 1:
                add(e0 + i + 2, 'asm volatile("v_cvt_pk_bf16_f32 %%0, %%1, %%2" : "=v"(Pf[%s][%d][%d]) : "v"(p_[%s][%d]), "v"(p_[%s][%d]) : A2_ALLA);'
                    % (H, i >> 3, (i >> 1) & 3, H, i - 1, H, i), prio=8)
        add(e0 + 34, "l_run[%s] = lsum[%s];" % (H, H), prio=9)

    # ---------------- V^T fragment reads of block kb (fragment f = 4 s + dt is free after PV(B, kb-1)'s MFMA at gap 16 + f)
    for n in range(32):
        f, jj = n >> 1, n & 1
        s, dt = f >> 2, f & 3
        g = 23 + (n * 18) // 32
        assert g >= 16 + f + 2 and g <= 46
        add(g, 'asm volatile("ds_read_b64_tr_b16 %%0, %%1 offset:%%2" : "=v"(Vf[%d][%d][%d]) : "v"(vb[%d][%d]), "i"(VCUR + %d) : A2_ALLA);'
            % (dt, s, jj, dt, jj, 4096 * s), kind="lds", prio=5)
    # every V^T fragment has landed before PV(A) starts (the K reads of gaps 56.. come later)
    add(47, 'asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory", A2_ALLA);', kind="wait", prio=9)
    # ---------------- K fragment reads of block kb + 1 (K[kt][ks] is free after QK(B)'s MFMA at gap 32 + 8 kt + ks)
    for n in range(16):
        kt, ks = n >> 3, n & 7
        g = 56 + n // 2
        assert g >= 32 + 8 * kt + ks + 2
        add(g, 'asm volatile("ds_read_b128 %s, %%0 offset:%%1" :: "v"(kaddr[%d]), "i"(KNEXT + %d) : %s);'
            % (areg(K_(kt, ks), 4), ks, kt * 8192, clob(K_(kt, ks), 4)), kind="lds", prio=5)
    # ---------------- LDS-DMA of K(kb + 2) and V(kb + 1): early in the block, in the lightest gaps
    for i in range(4):
        add(1 + 2 * i, "A2_DMA_K(%d)" % i, kind="dma", prio=5)
        add(2 + 2 * i, "A2_DMA_V(%d)" % i, kind="dma", prio=5)
    # O[B] of a tile starts at zero: written in the FIRST block, whose PV(B) gaps carry no MFMA
    for dt in range(4):
        o = O_(1, dt)
        code = 'asm volatile("' + "\\n\\t".join("v_accvgpr_write_b32 a%d, 0" % (o + i) for i in range(16)) + '" ::: %s);' % clob(o, 16)
        add(16 + dt, code, "FIRST", prio=5)
    return mf, ops


def emit(mf, ops, lo, hi, wrap_only, f):
    """gaps lo..hi-1. wrap_only: the drain stream (what a wave still owes after its last block)."""
    by_gap = {}
    for o in ops:
        g, cond = o.gap, o.cond
        wrapped = g >= NG
        if wrapped:
            g -= NG
        if wrap_only and not wrapped:
            continue
        by_gap.setdefault(g, []).append((o, wrapped))
    for g in range(lo, hi):
        f.write("// ---- gap %d\n" % g)
        if g in mf:
            code, cond = mf[g]
            if wrap_only:
                if cond == "!FIRST":      # PV(B, last block)
                    f.write(code + "\n")
            elif cond:
                f.write("if constexpr (%s) { %s }\n" % (cond, code))
            else:
                f.write(code + "\n")
        lst = sorted(by_gap.get(g, []), key=lambda t: t[0].prio)
        for o, wrapped in lst:
            conds = []
            if wrapped and not wrap_only:
                conds.append("!FIRST")
            if o.cond:
                conds.append(o.cond)
            if conds:
                f.write("if constexpr (%s) { %s }\n" % (" && ".join(conds), o.code))
            else:
                f.write(o.code + "\n")
        f.write("__builtin_amdgcn_sched_barrier(0);\n")


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
    # Q fragments of a tile: LDS (the wave's own 64 x 256 B region) -> a[128:191]
    f.write("#ifdef A2_EMIT_QLOAD\n")
    for h in range(2):
        for ks in range(8):
            f.write('asm volatile("ds_read_b128 %s, %%0 offset:%d" :: "v"(kaddr[%d] + qoff) : %s);\n'
                    % (areg(Q_(h, ks), 4), h * 8192, ks, clob(Q_(h, ks), 4)))
    f.write("#endif\n#ifdef A2_EMIT_KLOAD\n")   # K fragments of a tile's block 0 (slot 0)
    for kt in range(2):
        for ks in range(8):
            f.write('asm volatile("ds_read_b128 %s, %%0 offset:%d" :: "v"(kaddr[%d]) : %s);\n'
                    % (areg(K_(kt, ks), 4), kt * 8192, ks, clob(K_(kt, ks), 4)))
    f.write("#endif\n")
    # O[half] *= alpha (a row's reference maximum moved): accumulator file -> VGPR -> multiply -> back, 8 at a time
    for h in range(2):
        f.write("#ifdef A2_EMIT_RESCALE_%d\n" % h)
        for c in range(8):
            lo = O_(h, 0) + 8 * c
            rd = "\\n\\t".join("v_accvgpr_read_b32 %%%d, a%d" % (i, lo + i) for i in range(8))
            ml = "\\n\\t".join("v_mul_f32 %%%d, %%8, %%%d" % (i, i) for i in range(8))
            wr = "\\n\\t".join("v_accvgpr_write_b32 a%d, %%%d" % (lo + i, i) for i in range(8))
            outs = ", ".join('"=&v"(rt_[%d])' % i for i in range(8))
            f.write('asm volatile("%s\\n\\t%s\\n\\t%s" : %s : "v"(alpha[%d]) : %s);\n' % (rd, ml, wr, outs, h, clob(lo, 8)))
        f.write("#endif\n")
    # epilogue: one O^T tile (16 registers) at a time into ov[], A2_OSTORE(half, dt) consumes it
    f.write("#ifdef A2_EMIT_OREAD\n")
    for h in range(2):
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
        f.write("#ifdef A2_EMIT_BODY\n")
        emit(mf, ops, 0, NG, False, f)
        f.write("#endif\n#ifdef A2_EMIT_DRAIN\n")
        emit(mf, ops, 0, max(32, last_wrapped + 1), True, f)
        f.write("#endif\n")
        emit_static(f)
    if "-v" in sys.argv:
        report(ops)


if __name__ == "__main__":
    main()
