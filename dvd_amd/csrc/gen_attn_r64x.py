#!/usr/bin/env python3
"""Generates dvd_amd/csrc/attn_r64x_body.inc: the key-tile loop of flash_attn_r64x_kernel (attention.hip) - the
16x16x32-MFMA sibling of flash_attn_r64m_kernel (gen_attn_r64m.py; read that file first: same pipeline across key tiles,
same ownership of v[28:255] / a[0:255] by the statements, same six glue-free tile variants, same LDS rings and DMA).

Why the other MFMA shape: every MFMA kernel of the step runs against the board's power cap, and the chip holds a higher
clock on v_mfma_f32_16x16x32_f16 than on 32x32x16 for the same FLOPs (benchmarks/lab/shape_lab.hip: +17 % in bare loops;
priced inside r64m by an ablation that issues two 16x16x32 for every 32x32x16: 1.90 instead of 1.50 GHz, but 57.0 M
instead of 45.8 M cycles, because one wave per SIMD has to ISSUE 128 instead of 64 MFMAs per tile beside ~1600 issue-cycles
of other instructions).  So this kernel also cuts the other instructions: no threshold registers, no side sums - but NOT by
packed-f32 or DOT instructions, which serialise with the matrix pipe (see pk_arg, sum_word).

Fragment maps (v_mfma_f32_16x16x32_f16; lane l: c = l & 15, g = l >> 4):
  A / B operand  row (A) or column (B) c of the 16-wide tile, k = 8 g .. 8 g + 7        (4 VGPRs = 8 halves)
  C / D          column c, rows 4 g + r, r = 0..3                                          (4 registers)
  S^T = K . Q^T  tile (kb2, qb): 16 keys x 16 queries; lane (c, g): query 16 qb + c, A-rows 4 g + r of key block kb2
  O^T = V^T . P  tile (db, qb):  16 dims x 16 queries; contraction over the tile's 32 keys in ONE k-step:
                 lane (c, g) of the B operand P holds, for query 16 qb + c, the keys k' = 8 g + j, j = 0..7
  => A-row i of key block kb2 must be the natural key 8 (i >> 2) + 4 kb2 + (i & 3): then a lane's S^T registers
     (kb2, r) ARE its P keys j = 4 kb2 + r.  K tiles are therefore stored in LDS in A-row order (row 16 kb2 + i), which the
     LDS-DMA does for free (a piece = two consecutive natural keys; the four pieces of a wave start at natural keys
     16 (w & 1) + 4 (w >> 1) + {0, 2, 8, 10}), in r64m's piece layout (pitch 1056, odd row's chunks XOR 1): conflict-free
     for ds_read_b128's lane groups {0-3, 12-15, 20-27} ... and ONE per-lane base + immediates.
     V^T tiles stay [dim][key] with 64-byte rows; the chunk swizzle becomes g ^ ((-(row >> 2)) & 3) (those lane groups).

Register plan:
  AGPR  a[0:255]   O^T: tile (db, qb) at a[64 qb + 4 db ...]
  VGPR  v[0:27]    the compiler's (amdgpu_num_vgpr(28))
        v[28:31]   running maximum m of the lane's query in query block 0..3
        v[32:47]   packed P fragments, query block qb at 32 + 4 qb (word j = keys 2 j, 2 j + 1)
        v[48:63]   fragment ring (four slots)
        v[64:95]   S^T buffer A: element 8 qb + 4 kb2 + r  (= P key order: unit u = element u)
        v[96:127]  S^T buffer B
        v[128:255] Q fragments: 128 + 4 (8 qb + ks)
  SGPR  s[80:91]   loop scalars (clobbers): DMA source pairs, tile counter, scratch, return variant, (c, c) pair

Schedule of tile t (step = one K / V^T fragment + its FOUR MFMAs, 64 matrix cycles; one VALU instruction per MFMA gap):
  phase 1   16 steps  S^T(t+1) += K(t+1) frag (ks, kb2) . Q^T   | exp units 16..31 of tile t; the 16 packs of P(t) and
                                                                  their row-sum dot2; K(t+3) LDS-DMA pieces (f = 3, 7, 11, 15)
  vmcnt(4) + s_barrier
  phase 2   16 steps  O^T += V^T(t) frag db . P(t)              | lane-local maxima of S^T(t+1) and the rescale test
                                                                  (steps 0..5; rare block), then s * c - m of ALL of tile
                                                                  t + 1 (16 v_pk_fma) and its exp units 0..15;
                                                                  V^T(t+2) pieces (g = 1..4)
The rare block runs INSIDE phase 2, after some of its MFMAs: it scales O^T, l and the packed P(t) by alpha, so the tiles
of O^T that already took P(t) V^T(t) and those that will take it stay consistent.
"""
import os
import sys

MREG = 28
P0 = 32
FR0 = 48
SBUF = (64, 96)
Q0 = 128
KPIECE = 1056
KBYTES, VBYTES = 16 * KPIECE, 16384
S_KG, S_VG, S_TC, S_TMP, S_SEL, S_C = 80, 82, 84, 85, 86, 90
SGPR_CLOBBERS = [f"s{i}" for i in range(80, 92)]
THR_BITS = "0x41200000"          # 10.0f (log2 units), as in the other attention kernels
ONES_F16X2 = "0x3c003c00"
COMPILER_VGPRS = MREG

MF = "v_mfma_f32_16x16x32_f16"
ABL = set()        # timing ablations (lab builds only; garbage results): "valu", "dma", "read", "wait", "bar"


def vr(lo, n=1):
    return f"v{lo}" if n == 1 else f"v[{lo}:{lo + n - 1}]"


def frag(slot):
    return vr(FR0 + 4 * (slot & 3), 4)


def stile(buf, kb2, qb):
    return vr(SBUF[buf] + 8 * qb + 4 * kb2, 4)


def oreg(qb, db):
    return f"a[{64 * qb + 4 * db}:{64 * qb + 4 * db + 3}]"


def qreg(qb, ks):
    return vr(Q0 + 4 * (8 * qb + ks), 4)


def pfrag(qb):
    return vr(P0 + 4 * qb, 4)


class Stmt:
    def __init__(self):
        self.lines = []

    def add(self, s):
        if "read" in ABL and s.startswith("ds_read"):
            return
        if "wait" in ABL and s.startswith("s_waitcnt lgkmcnt"):
            return
        if "bar" in ABL and s.startswith("s_barrier"):
            return
        self.lines.append(s)

    def label(self, name):
        self.lines.append(name + ":")

    def text(self):
        return "\n".join(f'      "{ln}\\n\\t"' for ln in self.lines)


# ---- VALU items (strings; "valu" ablation drops them all, the test then never fires) ----
def pk_arg(buf, k):
    """s * c - m for elements 2k, 2k + 1 of the tile in `buf` (query block k >> 2), in place -> list of instructions.
    Two v_fma_f32, NOT one v_pk_fma_f32: packed-f32 and DOT instructions do not overlap with the matrix pipe on gfx950
    (benchmarks/lab/opsel_lab.hip: MFMA 16x16x32 + v_fma_f32 = 17 cycles per pair, + v_pk_fma_f32 / v_pk_add_f32 /
    v_dot2c_f32_f16 = 34), so a packed instruction costs three times what the two scalar ones cost beside MFMAs."""
    x, q = SBUF[buf] + 2 * k, k >> 2
    if PK_ARGS:       # experiment switch (even query blocks only: the odd ones, m in the high register of its pair, came out wrong)
        if not q & 1:
            return [f"v_pk_fma_f32 {vr(x, 2)}, {vr(x, 2)}, s[{S_C}:{S_C + 1}], {vr(MREG + q, 2)} op_sel_hi:[1,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]"]
    return [f"v_fma_f32 v{x + i}, v{x + i}, s{S_C}, -v{MREG + q}" for i in range(2)]


def exp_unit(buf, u):
    return f"v_exp_f32_e32 v{SBUF[buf] + u}, v{SBUF[buf] + u}"


def cvt_word(buf, w):
    return f"v_cvt_pk_f16_f32 v{P0 + w}, v{SBUF[buf] + 2 * w}, v{SBUF[buf] + 2 * w + 1}"


SUM_BY_DOT2 = os.environ.get("R64X_SUM_BY_DOT2") == "1"    # experiment switches of the generator (not product options)
PK_ARGS = os.environ.get("R64X_PK_ARGS") == "1"
WAIT_EVERY_STEP = os.environ.get("R64X_WAIT_EVERY_STEP") == "1"


def sum_word(w, buf):
    """row sum of the two exponentials of word w.  Two v_add_f32 on the f32 values: one v_dot2c_f32_f16 on the packed word
    measured 2711 instead of 2460 cycles per tile here, and the same 300 cycles in r64m when its row sums were moved to dot2
    (round 4) - DOT instructions wait for the matrix pipe (see pk_arg)."""
    if SUM_BY_DOT2:
        return f"v_dot2c_f32_f16 %[l{w >> 2}], {ONES_F16X2}, v{P0 + w}"
    return (f"v_add_f32_e32 %[l{w >> 2}], %[l{w >> 2}], v{SBUF[buf] + 2 * w}\\n\\t"
            f"v_add_f32_e32 %[l{w >> 2}], %[l{w >> 2}], v{SBUF[buf] + 2 * w + 1}")


def max_chain(buf):
    """lane-local maximum of each query block's 8 scores -> a0..a3 (four interleaved chains), then the test:
    vcc = some lane's max(a_q c - m_q) > THR"""
    items = []
    x = lambda q, j: f"v{SBUF[buf] + 8 * q + j}"
    for q in range(4):
        items.append(f"v_max3_f32 %[a{q}], {x(q, 0)}, {x(q, 1)}, {x(q, 2)}")
    for j in (3, 5):
        for q in range(4):
            items.append(f"v_max3_f32 %[a{q}], %[a{q}], {x(q, j)}, {x(q, j + 1)}")
    for q in range(4):
        items.append(f"v_max_f32_e32 %[a{q}], %[a{q}], {x(q, 7)}")
    for q in range(4):
        items.append(f"v_fma_f32 %[t{q + 1}], %[a{q}], s{S_C}, -v{MREG + q}")
    items.append("v_max3_f32 %[t1], %[t1], %[t2], %[t3]")
    items.append("v_max_f32_e32 %[t1], %[t1], %[t4]")
    items.append(f"v_cmp_lt_f32_e32 vcc, {THR_BITS}, %[t1]")
    return items


def read_for_step(n, slot):
    """(address operand, immediate) of the fragment that step n of tile t consumes (slot = t % 3); n >= 32: the next tile's"""
    if n < 16:      # K(t+1), fragment (ks, kb2) = (n >> 1, n & 1)
        return "kaddr", ((slot + 1) % 3) * KBYTES + (n & 1) * 8 * KPIECE + (n >> 1) * 64
    if n < 32:      # V^T(t), dims 16 (n - 16) ...
        return "vrel", slot * VBYTES + (n - 16) * 1024
    n -= 32         # K(t+2)
    return "kaddr", ((slot + 2) % 3) * KBYTES + (n & 1) * 8 * KPIECE + (n >> 1) * 64


def dma_m0(s, which, slot, i):
    if "dma" in ABL:
        return
    imm = slot * (KBYTES if which == "k" else VBYTES) + i * (KPIECE if which == "k" else 1024)
    s.add(f"s_add_i32 m0, %[{which}dst], {imm}")


def dma(s, which, i):
    if "dma" in ABL:
        return
    sg = S_KG if which == "k" else S_VG
    s.add(f"global_load_lds_dwordx4 %[{which}off{i}], s[{sg}:{sg + 1}]")


def advance(s, which):
    if "dma" in ABL:
        return
    sg = S_KG if which == "k" else S_VG
    s.add(f"s_cmp_lt_i32 s{S_TC}, %[{which}lim]")
    s.add(f"s_cselect_b32 s{S_TMP}, %[{which}step], 0")
    s.add(f"s_add_u32 s{sg}, s{sg}, s{S_TMP}")
    s.add(f"s_addc_u32 s{sg + 1}, s{sg + 1}, 0")


def ring_wait(s, n):
    """before step n uses ring slot n & 3.  Fragment reads are issued three steps ahead and return in order: at an EVEN step
    `lgkmcnt(1)` (all but the read issued last, fragment n + 2) covers fragments n AND n + 1, so the odd steps need no wait -
    16 instead of 32 s_waitcnt per tile, in a loop where every instruction beside the 128 MFMA issues costs its issue time."""
    if WAIT_EVERY_STEP:
        s.add("s_waitcnt lgkmcnt(2)")
    elif not n & 1:
        s.add("s_waitcnt lgkmcnt(1)")


def emit_gap(s, items):
    for it in items:
        if it.startswith(".L") or it.startswith("s_cbranch"):
            if it.endswith(":"):
                s.label(it[:-1])
            elif "valu" not in ABL:                      # without the test there is nothing to branch on
                s.add(it)
        elif "valu" not in ABL:
            s.add(it)


def tile(s, var):
    par, slot = var & 1, var % 3
    cur, nxt = par, 1 - par
    # ---------------- phase 1: S^T(t+1); exp units 16..31 of tile t, packs and row sums of P(t)
    gaps = [[] for _ in range(64)]
    for f in range(16):
        gaps[4 * f + 0].append(exp_unit(cur, 16 + f))
        gaps[4 * f + 1].append(cvt_word(cur, f))          # words 8..15 follow their second exponential by >= one MFMA
        adds = sum_word(f, cur).split("\\n\\t")
        if len(adds) == 2:                                # one VALU instruction per MFMA gap
            gaps[4 * f + 2].append(adds[0])
            gaps[4 * f + 3].append(adds[1])
        else:
            gaps[4 * f + 3].append(adds[0])
    for f in range(16):
        n, ks, kb2 = f, f >> 1, f & 1
        ring_wait(s, n)
        for qb in range(4):
            d = stile(nxt, kb2, qb)
            s.add(f"{MF} {d}, {frag(n)}, {qreg(qb, ks)}, {'0' if ks == 0 else d}")
            if qb == 0:
                a, off = read_for_step(n + 3, slot)
                s.add(f"ds_read_b128 {frag(n + 3)}, %[{a}] offset:{off}")
                if f in (3, 7, 11, 15):
                    dma_m0(s, "k", slot, f >> 2)              # K(t+3) -> K slot t % 3
            if qb == 2 and f in (3, 7, 11, 15):
                dma(s, "k", f >> 2)
                if f == 15:
                    advance(s, "k")
            emit_gap(s, gaps[4 * f + qb])
    s.add("s_waitcnt vmcnt(4)")
    s.add("s_barrier")
    # ---------------- phase 2: PV(t); maxima + test of tile t+1, its softmax argument, its exp units 0..15
    seq = max_chain(nxt)
    seq.append(f"s_cbranch_vccnz .Lr64x_stub{var}_%=")
    seq.append(f".Lr64x_back{var}_%=:")
    order = [("pk", 0), ("pk", 1), ("e", 0), ("e", 1), ("pk", 2), ("e", 2), ("e", 3), ("pk", 3), ("e", 4), ("e", 5), ("pk", 4),
             ("e", 6), ("e", 7), ("pk", 5), ("e", 8), ("e", 9), ("pk", 6), ("e", 10), ("e", 11), ("pk", 7), ("e", 12), ("e", 13),
             ("e", 14), ("e", 15)] + [("pk", k) for k in range(8, 16)]
    for kind, i in order:          # one instruction per gap; the arguments of units 16..31 (not needed before the next tile's
        if kind == "e":            # phase 1) go two to a gap: 23 + 16 + 16 + 8 = the 63 gaps of the phase
            seq.append(exp_unit(nxt, i))
        elif i < 8:
            seq.extend(pk_arg(nxt, i))
        else:
            seq.append("\\n\\t".join(pk_arg(nxt, i)))
    gaps = [[] for _ in range(64)]
    pos = 1                                                # gap 0 stays empty: the last S^T MFMA must have written its tile
    for it in seq:
        gaps[pos].append(it)
        if not (it.startswith("s_cbranch") or it.endswith(":")):
            pos += 1
    assert pos <= 64, pos
    for g in range(16):
        n = 16 + g
        ring_wait(s, n)
        for qb in range(4):
            s.add(f"{MF} {oreg(qb, g)}, {frag(n)}, {pfrag(qb)}, {oreg(qb, g)}")
            if qb == 0:
                a, off = read_for_step(n + 3, slot)
                s.add(f"ds_read_b128 {frag(n + 3)}, %[{a}] offset:{off}")
                if 1 <= g <= 4:
                    dma_m0(s, "v", (slot + 2) % 3, g - 1)     # V^T(t+2) -> V slot (t + 2) % 3
            if qb == 2 and 1 <= g <= 4:
                dma(s, "v", g - 1)
                if g == 4:
                    advance(s, "v")
            if qb == 3 and g == 5:
                s.add(f"s_add_i32 s{S_TC}, s{S_TC}, 1")
            emit_gap(s, gaps[4 * g + qb])


def rare_block(s):
    """out of line, shared by the six variants (s[S_SEL] = the variant to return to): new reference maxima; O^T, l and the
    packed P(t) of every query block scaled by alpha = 2^(m_old - m_new)"""
    s.label(".Lr64x_rare_%=")
    s.add("s_nop 15")                                     # the PV MFMAs issued so far must have written O^T
    s.add("s_nop 7")
    t0, t1 = "%[t0]", "%[t1]"
    for q in range(4):
        s.add(f"v_mul_f32_e32 {t0}, s{S_C}, %[a{q}]")
        s.add(f"ds_swizzle_b32 {t1}, {t0} offset:swizzle(SWAP,16)")     # the query's other keys: lanes ^ 16 and ^ 32
        s.add("s_waitcnt lgkmcnt(0)")
        s.add(f"v_max_f32_e32 {t0}, {t0}, {t1}")
        s.add(f"v_mov_b32_e32 {t1}, {t0}")
        s.add("s_nop 1")
        s.add(f"v_permlane32_swap_b32 {t0}, {t1}")
        s.add("s_nop 1")
        s.add(f"v_max_f32_e32 {t0}, {t0}, {t1}")
        s.add(f"v_max_f32_e32 {t1}, v{MREG + q}, {t0}")    # m_new
        s.add(f"v_sub_f32_e32 {t0}, v{MREG + q}, {t1}")
        s.add(f"v_exp_f32_e32 {t0}, {t0}")                # alpha
        s.add(f"v_mov_b32_e32 v{MREG + q}, {t1}")
        s.add("s_nop 0")
        s.add(f"v_mul_f32_e32 %[l{q}], %[l{q}], {t0}")
        s.add(f"v_cvt_pk_f16_f32 {t1}, {t0}, {t0}")
        for j in range(4):
            s.add(f"v_pk_mul_f16 v{P0 + 4 * q + j}, v{P0 + 4 * q + j}, {t1}")
        for a0 in range(64 * q, 64 * q + 64, 4):
            for i in range(4):
                s.add(f"v_accvgpr_read_b32 %[t{1 + i}], a{a0 + i}")
            for i in range(4):
                s.add(f"v_mul_f32_e32 %[t{1 + i}], {t0}, %[t{1 + i}]")
            for i in range(4):
                s.add(f"v_accvgpr_write_b32 a{a0 + i}, %[t{1 + i}]")
    s.add("s_nop 1")
    for var in range(5):
        s.add(f"s_cmp_eq_u32 s{S_SEL}, {var}")
        s.add(f"s_cbranch_scc1 .Lr64x_back{var}_%=")
    s.add("s_branch .Lr64x_back5_%=")


def loop_stmt():
    s = Stmt()
    s.add(f"s_mov_b64 s[{S_KG}:{S_KG + 1}], %[kg]")
    s.add(f"s_mov_b64 s[{S_VG}:{S_VG + 1}], %[vg]")
    s.add(f"s_mov_b32 s{S_TC}, 0")
    s.add(f"s_mov_b32 s{S_C}, %[c]")
    s.add(f"s_mov_b32 s{S_C + 1}, %[c]")
    s.label(".Lr64x_loop_%=")
    for var in range(6):
        tile(s, var)
        if var in (1, 3):                                 # the tile count is even
            s.add(f"s_cmp_ge_i32 s{S_TC}, %[nt]")
            s.add("s_cbranch_scc1 .Lr64x_end_%=")
    s.add(f"s_cmp_lt_i32 s{S_TC}, %[nt]")
    s.add("s_cbranch_scc1 .Lr64x_loop_%=")
    s.add("s_branch .Lr64x_end_%=")
    for var in range(6):
        s.label(f".Lr64x_stub{var}_%=")
        s.add(f"s_mov_b32 s{S_SEL}, {var}")
        s.add("s_branch .Lr64x_rare_%=")
    rare_block(s)
    s.label(".Lr64x_end_%=")
    s.add("s_waitcnt vmcnt(0) lgkmcnt(0)")                # no LDS-DMA may land after the workgroup has ended
    s.add("s_nop 15")                                     # the last PV MFMAs must have written O^T before it is read out
    s.add("s_nop 7")
    return s


def prologue_s0():
    """S^T(0) into buffer 0 from K slot 0 (un-pipelined), then the lane-local maxima of the four query blocks"""
    s = Stmt()
    for f in range(3):
        s.add(f"ds_read_b128 {frag(f)}, %[kaddr] offset:{(f & 1) * 8 * KPIECE + (f >> 1) * 64}")
    for f in range(16):
        if f + 3 < 16:
            n = f + 3
            s.add(f"ds_read_b128 {frag(n)}, %[kaddr] offset:{(n & 1) * 8 * KPIECE + (n >> 1) * 64}")
            s.add("s_waitcnt lgkmcnt(3)")
        else:
            s.add(f"s_waitcnt lgkmcnt({15 - f})")
        for qb in range(4):
            d = stile(0, f & 1, qb)
            s.add(f"{MF} {d}, {frag(f)}, {qreg(qb, f >> 1)}, {'0' if f < 2 else d}")
    s.add("s_nop 15")
    s.add("s_nop 7")
    for it in max_chain(0)[:16]:
        s.add(it)
    return s


def prologue_units():
    """m -> v[28:31]; s * c - m for all of tile 0 and its exp units 0..15 (what phase 2 of a tile does for the next one);
    the fragment ring primed with K(1) fragments 0..2"""
    s = Stmt()
    for f in range(3):
        s.add(f"ds_read_b128 {frag(f)}, %[kaddr] offset:{KBYTES + (f & 1) * 8 * KPIECE + (f >> 1) * 64}")
    for q in range(4):
        s.add(f"v_mov_b32_e32 v{MREG + q}, %[m{q}]")
    for u in range(32):
        s.add(f"v_fma_f32 v{SBUF[0] + u}, v{SBUF[0] + u}, %[c], -%[m{u >> 3}]")
    for u in range(16):
        s.add(exp_unit(0, u))
    return s


VARIANTS = [("", ()), ("novalu", ("valu",)), ("nobar", ("bar",)), ("mfmaonly", ("valu", "dma", "read", "wait"))]


def emit_loop(w, sfx):
    w(f"// ---- the key-tile loop{sfx}: six tile variants, the rare rescale block, the drain")
    w(f"__device__ __forceinline__ void r64x_loop{sfx}(float& l0, float& l1, float& l2, float& l3, const char* kg, const char* vg, int nt,")
    w("    unsigned kaddr, unsigned vrel, const unsigned (&koff)[4], const unsigned (&voff)[4], float c, unsigned kdst, unsigned vdst,")
    w("    unsigned kstep, unsigned vstep, int klim, int vlim) {")
    w("  float a0, a1, a2, a3, t0, t1, t2, t3, t4;")
    w("  asm volatile(")
    w(loop_stmt().text())
    w('      : [l0] "+v"(l0), [l1] "+v"(l1), [l2] "+v"(l2), [l3] "+v"(l3), [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3),')
    w('        [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4)')
    w('      : [kg] "s"(kg), [vg] "s"(vg), [nt] "s"(nt), [kaddr] "v"(kaddr), [vrel] "v"(vrel), [koff0] "v"(koff[0]), [koff1] "v"(koff[1]),')
    w('        [koff2] "v"(koff[2]), [koff3] "v"(koff[3]), [voff0] "v"(voff[0]), [voff1] "v"(voff[1]), [voff2] "v"(voff[2]), [voff3] "v"(voff[3]),')
    w('        [c] "s"(c), [kdst] "s"(kdst), [vdst] "s"(vdst), [kstep] "s"(kstep), [vstep] "s"(vstep), [klim] "s"(klim), [vlim] "s"(vlim)')
    w('      : "memory", "scc", "vcc", ' + ", ".join(f'"{r}"' for r in SGPR_CLOBBERS) + ");")
    w("}")
    w("")


def emit():
    out, lab = [], []
    lab.append("// GENERATED by dvd_amd/csrc/gen_attn_r64x.py --lab - do not edit.  TIMING ABLATIONS of the r64x loop (lab builds only:")
    lab.append("// they compute garbage).")
    lab.append("// clang-format off")
    w = out.append
    w("// GENERATED by gen_attn_r64x.py - do not edit; see that file for the fragment maps, the register plan and the schedule.")
    w("// clang-format off")
    w(f"#define R64X_COMPILER_VGPRS {COMPILER_VGPRS}   // the kernel carries __attribute__((amdgpu_num_vgpr(R64X_COMPILER_VGPRS)))")
    w("")
    w("// Q: 4 query blocks x 8 slabs of 32 dims; lane (c, g) holds query 16 qb + c, dims 32 ks + 8 g .. + 7")
    w("// (one wave-uniform base + a 32-bit byte offset per query block: 4 VGPRs of the compiler's 28 instead of 8)")
    w("__device__ __forceinline__ void r64x_load_q(const _Float16* base, unsigned q0, unsigned q1, unsigned q2, unsigned q3) {")
    w("  asm volatile(")
    for qb in range(4):
        for ks in range(8):
            w(f'      "global_load_dwordx4 {qreg(qb, ks)}, %[q{qb}], %[base] offset:{64 * ks}\\n\\t"')
    w('      "s_waitcnt vmcnt(0)"')
    w('      : : [base] "s"(base), [q0] "v"(q0), [q1] "v"(q1), [q2] "v"(q2), [q3] "v"(q3) : "memory", "v255", "a255");   // the clobbers: 256 + 256 registers')
    w("}")
    w("")
    w("__device__ __forceinline__ void r64x_prologue_s0(unsigned kaddr, float& a0, float& a1, float& a2, float& a3) {")
    w("  asm volatile(")
    w(prologue_s0().text())
    w('      : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3)')
    w('      : [kaddr] "v"(kaddr)')
    w('      : "memory");')
    w("}")
    w("")
    w("__device__ __forceinline__ void r64x_prologue_units(unsigned kaddr, float c, float m0, float m1, float m2, float m3) {")
    w("  asm volatile(")
    w(prologue_units().text())
    w('      :')
    w('      : [kaddr] "v"(kaddr), [c] "s"(c), [m0] "v"(m0), [m1] "v"(m1), [m2] "v"(m2), [m3] "v"(m3)')
    w('      : "memory");')
    w("}")
    w("")
    for abl_name, abl in VARIANTS:
        ABL.clear()
        ABL.update(abl)
        emit_loop(out.append if not abl_name else lab.append, "" if not abl_name else "_" + abl_name)
    ABL.clear()
    w("// clang-format on")
    lab.append("// clang-format on")
    return "\n".join(out) + "\n", "\n".join(lab) + "\n"


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    prod, lab = emit()
    ppath = os.path.join(here, "attn_r64x_body.inc")
    lpath = os.path.normpath(os.path.join(here, "..", "..", "benchmarks", "lab", "csrc", "attn_r64x_abl.inc"))
    arg = sys.argv[1] if len(sys.argv) > 1 else ""
    if arg == "--check":
        sys.exit(0 if os.path.exists(ppath) and open(ppath).read() == prod else 1)
    path, text = (lpath, lab) if arg == "--lab" else (ppath, prod)
    if not (os.path.exists(path) and open(path).read() == text):      # identical content keeps its mtime (make)
        open(path, "w").write(text)
    print(f"wrote {path}: {text.count(chr(10))} lines")
