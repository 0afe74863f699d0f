#!/usr/bin/env python3
"""Generates dvd_amd/csrc/attn_h64x_body.inc: the key-tile loop of flash_attn_h64x_kernel (attention.hip) - head_dim 64 on
v_mfma_f32_16x16x32_f16, i.e. gen_attn_h64x.py's kernel (read that file first: fragment maps, keys in P order, K tiles in
A-row order, statement-owned registers) at the geometry of gen_attn_h64m.py (64 query rows per wave, TWO waves per SIMD at
160 VGPRs + 64 AGPRs, 32-key tiles of 8 steps, one LDS-DMA piece per wave, stream and tile, barrier after phase 1).

Why: priced by an ablation of h64m that issues two 16x16x32 MFMAs for every 32x32x16 one (garbage math, same FLOPs): 10.01 vs
10.53 ms - the chip holds 1.98 instead of 1.68 GHz on this shape under its power cap, more than the extra cycles cost.

Per tile and wave: S^T = 2 key blocks x 4 query blocks x 2 slabs of 32 dims = 16 MFMAs (steps 0..3, fragment (ks, kb2) =
(f >> 1, f & 1)), PV = 4 dim blocks x 4 query blocks = 16 MFMAs (steps 4..7); 32 exp units, i.e. ~17 VALU instructions per
step, shared out over the four MFMA gaps of a step (two waves per SIMD issue them concurrently).
K image: rows in A-row order (row 16 kb2 + i = natural key 8 (i >> 2) + 4 kb2 + (i & 3)), 128-byte rows, chunks XOR-swizzled
by (row >> 1) & 7 (conflict-free for ds_read_b128's lane groups); the XOR depends on ks: one fragment base per ks.
V^T image: [dim][key], 64-byte rows, chunk ^ ((-(row >> 2)) & 3), as in r64x.

Register plan (per wave):
  AGPR  a[0:63]    O^T: tile (db, qb) at a[16 qb + 4 db ...]
  VGPR  v[0:31]    the compiler's (amdgpu_num_vgpr(32))
        v[32:47]   packed P fragments (query block qb at 32 + 4 qb); v[48:63] fragment ring
        v[64:95]   S^T buffer A (element 8 qb + 4 kb2 + r), v[96:127] buffer B
        v[128:159] Q fragments: 128 + 4 (2 qb + ks), PRE-SCALED by c = scale * log2(e) at load (fp32 multiply, one f16 rounding)
        v[160:175] -m of query block q, four copies at 160 + 4 q: the C operand of the first MFMA of each S^T chain, so the
                   accumulators come out as s * c - m and feed v_exp directly (as flash_attn_glds_kernel<64> does since round
                   2): 32 v_fma per tile gone from a loop in which VALU work does not hide.  The rare block adds its delta to
                   the tuples and subtracts it from the S^T(t+1) already formed against the old m (one block per buffer parity).
"""
import os
import sys

P0 = 32
FR0 = 48
SBUF = (64, 96)
Q0 = 128
KBYTES, VBYTES = 4096, 4096
S_KG, S_VG, S_TC, S_TMP, S_SEL = 80, 82, 84, 85, 86
SGPR_CLOBBERS = [f"s{i}" for i in range(80, 88)]
THR_BITS = "0x41200000"          # 10.0f (log2 units), as in the other attention kernels
ONES_F16X2 = "0x3c003c00"
COMPILER_VGPRS = 32
NEGM0 = 160                       # tuple of query block q at 160 + 4 q: four copies of -m_q (the C operand of a chain's first MFMA)

MF = "v_mfma_f32_16x16x32_f16"
# LM (round 6): the row sums on the MATRIX pipe.  V^T gets a fifth "dim block" of ones: O^T tile (db = 4, qb) =
# ones[16 x 32] . P[32 x 16] holds sum_k P[k][q] in every row - the row sum l of the lane's own query, of the SAME f16-rounded P that
# O^T takes - in a[64 + 4 qb ...]; the constant A fragment is one 4-register operand of the statement (0x3c003c00 x 4) that takes
# the place of the four l operands.  4 more MFMAs per tile (36 instead of 32), 32 v_add_f32 fewer (of ~100 VALU instructions per
# tile and wave in a loop whose VALU port is busier than its matrix pipe: 77.6 % vs 60.6 %, profiles/r5_pmc_*); 176 VGPRs + 80 AGPRs
# = the 256 registers of a wave at two per SIMD.  The rare block scales the l tiles like the O^T tiles; after the loop l is read
# from a[64 + 4 qb], complete (no cross-lane reduction).  Generated as a second body (prefix h64l) beside the round-4 one.
LM = False
L_SPREAD = os.environ.get("H64X_L_SPREAD") == "1"     # experiment switch of the generator (not a product option): one row-sum
#                                                        MFMA behind each PV group instead of four at the end of phase 2 - measured
#                                                        equal (8.52-8.56 vs 8.57 ms at the bench shape, two rounds), not adopted
PFX = "h64x"
ABL = set()        # timing ablations (lab builds only; garbage results): "valu", "dma", "read", "wait", "bar"


def vr(lo, n=1):
    return f"v{lo}" if n == 1 else f"v[{lo}:{lo + n - 1}]"


def frag(slot):
    return vr(FR0 + 4 * (slot & 3), 4)


def stile(buf, kb2, qb):
    return vr(SBUF[buf] + 8 * qb + 4 * kb2, 4)


def oreg(qb, db):
    return f"a[{16 * qb + 4 * db}:{16 * qb + 4 * db + 3}]"


def qreg(qb, ks):
    return vr(Q0 + 4 * (2 * qb + ks), 4)


def pfrag(qb):
    return vr(P0 + 4 * qb, 4)


def lreg(qb):
    return f"a[{64 + 4 * qb}:{64 + 4 * qb + 3}]"


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
def exp_unit(buf, u):
    return f"v_exp_f32_e32 v{SBUF[buf] + u}, v{SBUF[buf] + u}"


def cvt_word(buf, w):
    return f"v_cvt_pk_f16_f32 v{P0 + w}, v{SBUF[buf] + 2 * w}, v{SBUF[buf] + 2 * w + 1}"


SUM_BY_DOT2 = os.environ.get("R64X_SUM_BY_DOT2") == "1"    # experiment switches of the generator (not product options)
WAIT_EVERY_STEP = os.environ.get("R64X_WAIT_EVERY_STEP") == "1"


def sum_word(w, buf):
    """row sum of the two exponentials of word w.  Two v_add_f32 on the f32 values: one v_dot2c_f32_f16 on the packed word
    measured 2711 instead of 2460 cycles per tile here, and the same 300 cycles in r64m when its row sums were moved to dot2
    (round 4) - DOT instructions wait for the matrix pipe (gen_attn_r64x.py, pk_arg)."""
    if SUM_BY_DOT2:
        return f"v_dot2c_f32_f16 %[l{w >> 2}], {ONES_F16X2}, v{P0 + w}"
    return (f"v_add_f32_e32 %[l{w >> 2}], %[l{w >> 2}], v{SBUF[buf] + 2 * w}\\n\\t"
            f"v_add_f32_e32 %[l{w >> 2}], %[l{w >> 2}], v{SBUF[buf] + 2 * w + 1}")


def max_chain(buf):
    """lane-local maximum of each query block's 8 scores -> a0..a3 (four interleaved chains), then the test:
    vcc = some lane's maximum > THR"""
    items = []
    x = lambda q, j: f"v{SBUF[buf] + 8 * q + j}"
    for q in range(4):
        items.append(f"v_max3_f32 %[a{q}], {x(q, 0)}, {x(q, 1)}, {x(q, 2)}")
    for j in (3, 5):
        for q in range(4):
            items.append(f"v_max3_f32 %[a{q}], %[a{q}], {x(q, j)}, {x(q, j + 1)}")
    for q in range(4):
        items.append(f"v_max_f32_e32 %[a{q}], %[a{q}], {x(q, 7)}")
    items.append("v_max3_f32 %[t1], %[a0], %[a1], %[a2]")      # the scores already are s * c - m
    items.append("v_max_f32_e32 %[t1], %[t1], %[a3]")
    items.append(f"v_cmp_lt_f32_e32 vcc, {THR_BITS}, %[t1]")
    return items


def kfrag_addr(n, slot_off):
    """K fragment n = (ks, kb2) = (n >> 1, n & 1) of the tile in K slot `slot_off`"""
    return f"kf{n >> 1}", slot_off * KBYTES + (n & 1) * 2048


def read_for_step(n, slot):
    """(address operand, immediate) of the fragment that step n of tile t consumes (slot = t % 3); n >= 8: the next tile's"""
    if n < 4:
        return kfrag_addr(n, (slot + 1) % 3)                           # K(t+1)
    if n < 8:
        return "vrel", slot * VBYTES + (n - 4) * 1024                  # V^T(t), dims 16 (n - 4) ...
    return kfrag_addr(n - 8, (slot + 2) % 3)                           # K(t+2)


def dma_m0(s, which, slot):
    if "dma" in ABL:
        return
    s.add(f"s_add_i32 m0, %[{which}dst], {slot * (KBYTES if which == 'k' else VBYTES)}")


def dma(s, which):
    if "dma" in ABL:
        return
    sg = S_KG if which == "k" else S_VG
    s.add(f"global_load_lds_dwordx4 %[{which}off], s[{sg}:{sg + 1}]")


def advance(s, which):
    if "dma" in ABL:
        return
    sg = S_KG if which == "k" else S_VG
    s.add(f"s_cmp_lt_i32 s{S_TC}, %[{which}lim]")
    s.add(f"s_cselect_b32 s{S_TMP}, %[{which}step], 0")
    s.add(f"s_add_u32 s{sg}, s{sg}, s{S_TMP}")
    s.add(f"s_addc_u32 s{sg + 1}, s{sg + 1}, 0")


def ring_wait(s, n):
    """before step n uses ring slot n & 3 (as in r64x: `lgkmcnt(1)` at the even steps covers two fragments)"""
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


def spread(seq, ngaps):
    """seq (control items attach to the following instruction) -> ngaps lists of near-equal size, in order"""
    n = sum(1 for it in seq if not (it.startswith("s_cbranch") or it.endswith(":")))
    sizes = [n // ngaps + (1 if g < n % ngaps else 0) for g in range(ngaps)]
    gaps, g, k = [[] for _ in range(ngaps)], 0, 0
    for it in seq:
        while g < ngaps - 1 and k >= sizes[g]:
            g, k = g + 1, 0
        gaps[g].append(it)
        if not (it.startswith("s_cbranch") or it.endswith(":")):
            k += 1
    return gaps


def tile(s, var):
    par, slot = var & 1, var % 3
    cur, nxt = par, 1 - par
    # ---------------- phase 1 (steps 0..3): S^T(t+1); exp units 16..31 of tile t, the 16 packs of P(t) and their row sums
    seq = []
    for j in range(8):
        seq += [exp_unit(cur, 16 + 2 * j), exp_unit(cur, 17 + 2 * j), cvt_word(cur, j)]
        seq += [] if LM else sum_word(j, cur).split("\\n\\t")
    for j in range(8, 16):
        seq += [cvt_word(cur, j)] + ([] if LM else sum_word(j, cur).split("\\n\\t"))
    gaps = spread(seq, 16)
    for f in range(4):
        n, ks, kb2 = f, f >> 1, f & 1
        ring_wait(s, n)
        for qb in range(4):
            d = stile(nxt, kb2, qb)
            s.add(f"{MF} {d}, {frag(n)}, {qreg(qb, ks)}, {vr(NEGM0 + 4 * qb, 4) if ks == 0 else d}")
            if qb == 0:
                a, off = read_for_step(n + 3, slot)
                s.add(f"ds_read_b128 {frag(n + 3)}, %[{a}] offset:{off}")
                if f == 1:
                    dma_m0(s, "k", slot)                      # K(t+3) -> K slot t % 3
            if qb == 2 and f == 1:
                dma(s, "k")
                advance(s, "k")
            emit_gap(s, gaps[4 * f + qb])
    s.add("s_waitcnt vmcnt(1)")
    s.add("s_barrier")
    # ---------------- phase 2 (steps 4..7): PV(t); maxima + test of tile t+1, its softmax argument, its exp units 0..15
    seq = max_chain(nxt)
    seq.append(f"s_cbranch_vccnz .Lh64x_stub{var}_%=")            # -> the rare block of this variant's buffer parity
    seq.append(f".Lh64x_back{var}_%=:")
    order = [("pk", 0), ("pk", 1), ("e", 0), ("e", 1), ("pk", 2), ("e", 2), ("e", 3), ("pk", 3), ("e", 4), ("e", 5), ("pk", 4),
             ("e", 6), ("e", 7), ("pk", 5), ("e", 8), ("e", 9), ("pk", 6), ("e", 10), ("e", 11), ("pk", 7), ("e", 12), ("e", 13),
             ("e", 14), ("e", 15)] + [("pk", k) for k in range(8, 16)]
    for kind, i in order:
        if kind == "e":
            seq.append(exp_unit(nxt, i))
    gaps = spread(seq, 20 if LM else 16)
    for g in range(4):
        n = 4 + g
        ring_wait(s, n)
        for qb in range(4):
            s.add(f"{MF} {oreg(qb, g)}, {frag(n)}, {pfrag(qb)}, {oreg(qb, g)}")
            if qb == 0:
                a, off = read_for_step(n + 3, slot)
                s.add(f"ds_read_b128 {frag(n + 3)}, %[{a}] offset:{off}")
                if g == 1:
                    dma_m0(s, "v", (slot + 2) % 3)            # V^T(t+2) -> V slot (t + 2) % 3
            if qb == 2 and g == 1:
                dma(s, "v")
                advance(s, "v")
            if qb == 3 and g == 2:
                s.add(f"s_add_i32 s{S_TC}, s{S_TC}, 1")
            emit_gap(s, gaps[(5 * g + qb) if (LM and L_SPREAD) else (4 * g + qb)])
        if LM and L_SPREAD:                                   # experiment (H64X_L_SPREAD=1): one row-sum MFMA behind each PV group
            s.add(f"{MF} {lreg(g)}, %[ones], {pfrag(g)}, {lreg(g)}")
            emit_gap(s, gaps[5 * g + 4])
    if LM and not L_SPREAD:                                   # the row sums: O^T's fifth dim block, A = ones (no fragment read)
        for qb in range(4):
            s.add(f"{MF} {lreg(qb)}, %[ones], {pfrag(qb)}, {lreg(qb)}")
            emit_gap(s, gaps[16 + qb])


def rare_block(s, par):
    """out of line, one per buffer parity, shared by the three variants of that parity (s[S_SEL] = the variant to return to):
    delta = max(row maximum of S^T(t+1), 0) per query; m += delta (the -m tuples), S^T(t+1) -= delta; O^T, l and the packed
    P(t) of the query block scaled by alpha = 2^-delta"""
    nxt = 1 - par
    s.label(f".Lh64x_rare{par}_%=")
    s.add("s_nop 15")                                     # the PV MFMAs issued so far must have written O^T
    s.add("s_nop 7")
    t0, t1 = "%[t0]", "%[t1]"
    for q in range(4):
        s.add(f"ds_swizzle_b32 {t1}, %[a{q}] offset:swizzle(SWAP,16)")     # the query's other keys: lanes ^ 16 and ^ 32
        s.add("s_waitcnt lgkmcnt(0)")
        s.add(f"v_max_f32_e32 {t0}, %[a{q}], {t1}")
        s.add(f"v_mov_b32_e32 {t1}, {t0}")
        s.add("s_nop 1")
        s.add(f"v_permlane32_swap_b32 {t0}, {t1}")
        s.add("s_nop 1")
        s.add(f"v_max_f32_e32 {t0}, {t0}, {t1}")
        s.add(f"v_max_f32_e32 {t0}, 0, {t0}")             # delta
        s.add(f"v_exp_f32_e64 {t1}, -{t0}")               # alpha = 2^-delta
        for i in range(4):
            s.add(f"v_sub_f32_e32 v{NEGM0 + 4 * q + i}, v{NEGM0 + 4 * q + i}, {t0}")
        for i in range(8):
            s.add(f"v_sub_f32_e32 v{SBUF[nxt] + 8 * q + i}, v{SBUF[nxt] + 8 * q + i}, {t0}")
        if not LM:
            s.add(f"v_mul_f32_e32 %[l{q}], %[l{q}], {t1}")
        s.add(f"v_cvt_pk_f16_f32 {t0}, {t1}, {t1}")
        for j in range(4):
            s.add(f"v_pk_mul_f16 v{P0 + 4 * q + j}, v{P0 + 4 * q + j}, {t0}")
        tmp = ("%[t0]", "%[t2]", "%[t3]", "%[t4]")        # t1 = alpha; t0 (the packed alpha) is free again after the P words
        for a0 in list(range(16 * q, 16 * q + 16, 4)) + ([64 + 4 * q] if LM else []):     # LM: the l tile is O^T's fifth dim block
            for i in range(4):
                s.add(f"v_accvgpr_read_b32 {tmp[i]}, a{a0 + i}")
            for i in range(4):
                s.add(f"v_mul_f32_e32 {tmp[i]}, {t1}, {tmp[i]}")
            for i in range(4):
                s.add(f"v_accvgpr_write_b32 a{a0 + i}, {tmp[i]}")
    s.add("s_nop 1")
    mine = [v for v in range(6) if v & 1 == par]
    for var in mine[:-1]:
        s.add(f"s_cmp_eq_u32 s{S_SEL}, {var}")
        s.add(f"s_cbranch_scc1 .Lh64x_back{var}_%=")
    s.add(f"s_branch .Lh64x_back{mine[-1]}_%=")


def loop_stmt():
    s = Stmt()
    s.add(f"s_mov_b64 s[{S_KG}:{S_KG + 1}], %[kg]")
    s.add(f"s_mov_b64 s[{S_VG}:{S_VG + 1}], %[vg]")
    s.add(f"s_mov_b32 s{S_TC}, 0")
    s.label(".Lh64x_loop_%=")
    for var in range(6):
        tile(s, var)
        if var in (1, 3):                                 # the tile count is even
            s.add(f"s_cmp_ge_i32 s{S_TC}, %[nt]")
            s.add("s_cbranch_scc1 .Lh64x_end_%=")
    s.add(f"s_cmp_lt_i32 s{S_TC}, %[nt]")
    s.add("s_cbranch_scc1 .Lh64x_loop_%=")
    s.add("s_branch .Lh64x_end_%=")
    for var in range(6):
        s.label(f".Lh64x_stub{var}_%=")
        s.add(f"s_mov_b32 s{S_SEL}, {var}")
        s.add(f"s_branch .Lh64x_rare{var & 1}_%=")
    rare_block(s, 0)
    rare_block(s, 1)
    s.label(".Lh64x_end_%=")
    s.add("s_waitcnt vmcnt(0) lgkmcnt(0)")                # no LDS-DMA may land after the workgroup has ended
    s.add("s_nop 15")                                     # the last PV MFMAs must have written O^T before it is read out
    s.add("s_nop 7")
    return s


def prologue_s0():
    """S^T(0) into buffer 0 from K slot 0 (un-pipelined), then the lane-local maxima of the four query blocks"""
    s = Stmt()
    for f in range(4):
        a, off = kfrag_addr(f, 0)
        s.add(f"ds_read_b128 {frag(f)}, %[{a}] offset:{off}")
    for f in range(4):
        s.add(f"s_waitcnt lgkmcnt({3 - f})")
        for qb in range(4):
            d = stile(0, f & 1, qb)
            s.add(f"{MF} {d}, {frag(f)}, {qreg(qb, f >> 1)}, {'0' if f < 2 else d}")
    s.add("s_nop 15")
    s.add("s_nop 7")
    for it in max_chain(0)[:16]:
        s.add(it)
    return s


def prologue_units():
    """-m -> the C tuples; tile 0's scores (formed with C = 0) minus m; its exp units 0..15 (what phase 2 of a tile does for the
    next one); the fragment ring primed with K(1) fragments 0..2"""
    s = Stmt()
    for f in range(3):
        a, off = kfrag_addr(f, 1)
        s.add(f"ds_read_b128 {frag(f)}, %[{a}] offset:{off}")
    for q in range(4):
        for i in range(4):
            s.add(f"v_sub_f32_e32 v{NEGM0 + 4 * q + i}, 0, %[m{q}]")
    for u in range(32):
        s.add(f"v_sub_f32_e32 v{SBUF[0] + u}, v{SBUF[0] + u}, %[m{u >> 3}]")
    for u in range(16):
        s.add(exp_unit(0, u))
    return s


VARIANTS = [("", ()), ("novalu", ("valu",)), ("nobar", ("bar",)), ("mfmaonly", ("valu", "dma", "read", "wait"))]


def emit_loop(w, sfx):
    w(f"// ---- the key-tile loop{sfx}: six tile variants, the rare rescale block, the drain")
    if LM:
        w(f"__device__ __forceinline__ void {PFX}_loop{sfx}(uintx4 ones, const char* kg, const char* vg, int nt,")
    else:
        w(f"__device__ __forceinline__ void {PFX}_loop{sfx}(float& l0, float& l1, float& l2, float& l3, const char* kg, const char* vg, int nt,")
    w("    unsigned kf0, unsigned kf1, unsigned vrel, unsigned koff, unsigned voff, unsigned kdst, unsigned vdst,")
    w("    unsigned kstep, unsigned vstep, int klim, int vlim) {")
    w("  float a0, a1, a2, a3, t0, t1, t2, t3, t4;")
    w("  asm volatile(")
    w(loop_stmt().text())
    ls = "" if LM else '[l0] "+v"(l0), [l1] "+v"(l1), [l2] "+v"(l2), [l3] "+v"(l3), '
    w(f'      : {ls}[a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3),')
    w('        [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4)')
    w('      : [kg] "s"(kg), [vg] "s"(vg), [nt] "s"(nt), [kf0] "v"(kf0), [kf1] "v"(kf1), [vrel] "v"(vrel), [koff] "v"(koff), [voff] "v"(voff),')
    w('        [kdst] "s"(kdst), [vdst] "s"(vdst), [kstep] "s"(kstep), [vstep] "s"(vstep), [klim] "s"(klim), [vlim] "s"(vlim)'
      + (', [ones] "v"(ones)' if LM else ""))
    w('      : "memory", "scc", "vcc", ' + ", ".join(f'"{r}"' for r in SGPR_CLOBBERS) + ");")
    w("}")
    w("")


def emit(lm=False):
    global LM, PFX
    LM, PFX = lm, ("h64l" if lm else "h64x")
    nacc = 80 if LM else 64
    U = PFX.upper()
    out, lab = [], []
    lab.append(f"// GENERATED by dvd_amd/csrc/gen_attn_h64x.py --lab - do not edit.  TIMING ABLATIONS of the {PFX} loop (lab builds only:")
    lab.append("// they compute garbage).")
    lab.append("// clang-format off")
    w = out.append
    w("// GENERATED by gen_attn_h64x.py - do not edit; see that file for the fragment maps, the register plan and the schedule.")
    if LM:
        w("// This is the LM body: the row sums l on the matrix pipe (O^T's fifth dim block, A = ones), a[64:79].")
    w("// clang-format off")
    w(f"#define {U}_COMPILER_VGPRS {COMPILER_VGPRS}   // the kernel carries __attribute__((amdgpu_num_vgpr({U}_COMPILER_VGPRS)))")
    w("")
    w("// Q: 4 query blocks x 2 slabs of 32 dims; lane (c, g) holds query 16 qb + c, dims 32 ks + 8 g .. + 7; scaled by c in fp32")
    w("// (one wave-uniform base + a 32-bit byte offset per query block)")
    w(f"__device__ __forceinline__ void {PFX}_load_q(const _Float16* base, unsigned q0, unsigned q1, unsigned q2, unsigned q3, float c) {{")
    w("  float t0, t1;")
    w("  asm volatile(")
    for qb in range(4):
        for ks in range(2):
            w(f'      "global_load_dwordx4 {qreg(qb, ks)}, %[q{qb}], %[base] offset:{64 * ks}\\n\\t"')
    w('      "s_waitcnt vmcnt(0)\\n\\t"')
    for r in range(Q0, Q0 + 32):
        w(f'      "v_cvt_f32_f16_e32 %[t0], v{r}\\n\\tv_cvt_f32_f16_sdwa %[t1], v{r} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\\n\\t"')
        w(f'      "v_mul_f32_e32 %[t0], %[c], %[t0]\\n\\tv_mul_f32_e32 %[t1], %[c], %[t1]\\n\\tv_cvt_pk_f16_f32 v{r}, %[t0], %[t1]\\n\\t"')
    w('      "s_nop 0"')
    w('      : [t0] "=&v"(t0), [t1] "=&v"(t1)')
    w('      : [base] "s"(base), [q0] "v"(q0), [q1] "v"(q1), [q2] "v"(q2), [q3] "v"(q3), [c] "s"(c)')
    w(f'      : "memory", "v175", "a{nacc - 1}");   // the clobbers: 176 VGPRs + {nacc} AGPRs per wave')
    w("}")
    w("")
    w(f"__device__ __forceinline__ void {PFX}_zero_o() {{")
    w("  asm volatile(")
    for i in range(nacc):
        w(f'      "v_accvgpr_write_b32 a{i}, 0\\n\\t"')
    w('      "s_nop 1" ::: "memory");')
    w("}")
    w("")
    if LM:
        w("// the row sum of query block QB after the loop: any register of its l tile (all 16 rows of the tile are equal)")
        w("template <int QB> __device__ __forceinline__ float h64l_read_l() {")
        w("  float x;")
        for q in range(4):
            w(f'  {"if" if q == 0 else "else if"} constexpr (QB == {q}) asm volatile("v_accvgpr_read_b32 %0, a{64 + 4 * q}" : "=v"(x) : : "memory");')
        w("  return x;")
        w("}")
        w("")
    w(f"__device__ __forceinline__ void {PFX}_prologue_s0(unsigned kf0, unsigned kf1, float& a0, float& a1, float& a2, float& a3) {{")
    w("  asm volatile(")
    w(prologue_s0().text())
    w('      : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3)')
    w('      : [kf0] "v"(kf0), [kf1] "v"(kf1)')
    w('      : "memory");')
    w("}")
    w("")
    w(f"__device__ __forceinline__ void {PFX}_prologue_units(unsigned kf0, unsigned kf1, float m0, float m1, float m2, float m3) {{")
    w("  asm volatile(")
    w(prologue_units().text())
    w('      :')
    w('      : [kf0] "v"(kf0), [kf1] "v"(kf1), [m0] "v"(m0), [m1] "v"(m1), [m2] "v"(m2), [m3] "v"(m3)')
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
    labdir = os.path.normpath(os.path.join(here, "..", "..", "benchmarks", "lab", "csrc"))
    arg = sys.argv[1] if len(sys.argv) > 1 else ""
    ok = True
    for lm, stem in ((False, "attn_h64x"), (True, "attn_h64l")):
        prod, lab = emit(lm)
        ppath, lpath = os.path.join(here, stem + "_body.inc"), os.path.join(labdir, stem + "_abl.inc")
        if arg == "--check":
            ok = ok and os.path.exists(ppath) and open(ppath).read() == prod
            continue
        path, text = (lpath, lab) if arg == "--lab" else (ppath, prod)
        if not (os.path.exists(path) and open(path).read() == text):      # identical content keeps its mtime (make)
            open(path, "w").write(text)
        print(f"wrote {path}: {text.count(chr(10))} lines")
    if arg == "--check":
        sys.exit(0 if ok else 1)
