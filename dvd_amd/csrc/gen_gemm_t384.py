#!/usr/bin/env python3
"""Generates dvd_amd/csrc/gemm_t384_body.inc: the K loop of gemm_nt_t384_kernel (gemm.hip) as ONE asm statement with
hand-allocated registers - the recipe that moved the decoder attention in round 4 (gen_attn_r64x.py), applied to the 256-wide
per-step GEMMs.  What is different from gemm_nt_big_kernel (256 x 256 x 64 tiles, two 64-KiB LDS stages, the
next slab's 64 KiB issued at the top of a slab and waited for with vmcnt(0) + barrier at its bottom, compiler-scheduled):

  * a 384 x 256 tile per workgroup: 640 operand rows feed 384 x 256 outputs, 17 % fewer L2 -> LDS bytes per MFMA than 256 x 256
    (26.7 instead of 32 B/clk per CU at full matrix rate, against the ~33 B/clk the path delivers: benchmarks/lab/l2path_lab.hip);
    8 waves as 4 (M) x 2 (N), a wave owns 96 x 128 = 3 x 4 accumulators of v_mfma_f32_32x32x16_f16 = 192 AGPRs, and reads
    7 fragments per 12 MFMAs (0.58 ds_read_b128 per MFMA; the 256 x 256 kernel: 0.75);
  * K in HALF slabs of 32 (40 KiB: A 384 rows x 64 B | B 256 rows x 64 B), a ring of FOUR slots = all 160 KiB of LDS: a half
    slab's LDS-DMA is issued two and a half iterations before its first read, `s_waitcnt vmcnt(5)` (never 0) in front of the ONE
    barrier per iteration leaves the youngest group in flight;
  * the schedule is this file: per MFMA gap at most one ds_read_b128 and one LDS-DMA piece, counted lgkmcnt computed by the
    generator from the in-order LDS queue, every LDS address base + immediate.

Iteration j (half slab j in slot j % 4), one wave:
  phase 1   12 MFMAs of k-step 0 (fragments AX, Bq read one phase earlier)     | reads of k-step 1 of slot j % 4 -> AY, Bq
  s_waitcnt vmcnt(5) ; s_barrier            half slab j + 1 has landed for every wave; every wave is done with slot (j - 1) % 4
  phase 2   12 MFMAs of k-step 1                                               | reads of k-step 0 of slot (j + 1) % 4 -> AX, Bq;
                                                                                 the wave's 5 pieces of half slab j + 3 -> slot (j - 1) % 4
MFMA i of a k-step is (n, m) = (i // 3, i % 3): a B fragment serves three consecutive MFMAs and its register quad is re-loaded
right behind them (ring of four quads, v[48:63]); the three A fragments of a k-step live in one of two sets (v[24:35] / v[36:47]).
Per accumulator the MFMAs come in ascending k, 16 per instruction, like in gemm_nt_big_kernel: the two kernels give the same bits.

LDS image of a half slab (gemm_nt_big2_kernel's): 64-byte rows, 16-byte chunk c of row r at position c ^ ((r >> 2) & 3) -
conflict-free for ds_read_b128's lane groups {0-3, 12-15, 20-27}; a 1-KiB LDS-DMA piece = 16 rows, the swizzle applied to the
per-lane SOURCE address.  Wave w loads A pieces 3 w .. 3 w + 2 and B pieces 2 w, 2 w + 1.

Register plan (two waves per SIMD: 256 registers per lane, which hipcc splits 128 VGPRs + 128 AGPRs):
  the 12 accumulators are OUTPUT operands of the statement (floatx16 each), allocated by the compiler - rows m = 1, 2 "=&a"
  (128 AGPRs), row m = 0 "=&v" (64 VGPRs, the first the epilogue frees; an MFMA takes C/D in either file) - so the epilogue is ordinary compiled code
  that knows where they are, and no unwritten contract about registers exists
  VGPR  v[0:87]    the compiler's: 64 accumulator registers + the statement's operands (5 DMA source offsets, 4 read bases)
        v[88:99]   A fragments, set X        v[100:111]  set Y        v[112:127]  B fragment ring
  SGPR  s[80:85]   DMA source pointers (A, B) and the block counter (clobbers)
The loop statement declares v[88:127] as clobbers.
"""
import os
import sys

HALF, BOFF, RING = 40960, 24576, 4
SHIFT = 2 * HALF                   # the read bases reach two slots by immediates; the other two after a shift
AX, AY, BQ = 88, 100, 112
S_A, S_B, S_CNT = 80, 82, 84
SGPR_CLOBBERS = [f"s{i}" for i in range(80, 86)]
MF = "v_mfma_f32_32x32x16_f16"
ABL = set()        # timing ablations (lab builds only; garbage results): "dma", "read", "bar"
# phase-2 MFMA gaps that carry an LDS-DMA piece (T384_PIECES: experiment switch of the generator, not a product option)
PIECES_AFTER = tuple(int(x) for x in os.environ.get("T384_PIECES", "1,3,5,7,9").split(","))


def vq(lo):
    return f"v[{lo}:{lo + 3}]"


def acc(m, n):
    return f"%[c{m}{n}]"


# "m16" (round 6, PRICING ablation, lab only): every v_mfma_f32_32x32x16_f16 replaced by TWO v_mfma_f32_16x16x32_f16 on quads of the
# same accumulator (garbage math: the fragments keep their 32x32x16 layout; same FLOPs, the register-file traffic, MFMA issue
# count and LDS reads a real 16x16x32 loop would have).  The accumulators are then PINNED to physical registers
# ("={a[0:15]}" ...: rows m = 1, 2 in a[0:127], row m = 0 in v[24:87]) so that the statement can name their quads.
ACC_V0 = 24


def acc_quad(m, n, q):
    if m == 0:
        lo = ACC_V0 + 16 * n + 4 * q
        return f"v[{lo}:{lo + 3}]"
    lo = 16 * (4 * (m - 1) + n) + 4 * q
    return f"a[{lo}:{lo + 3}]"


def acc_pin(m, n):
    if m == 0:
        return f"{{v[{ACC_V0 + 16 * n}:{ACC_V0 + 16 * n + 15}]}}"
    lo = 16 * (4 * (m - 1) + n)
    return f"{{a[{lo}:{lo + 15}]}}"


def mfma(e, m, n, a_quad, b_quad, first, phase):
    if "m16" in ABL:
        for q in (2 * phase, 2 * phase + 1):
            d = acc_quad(m, n, q)
            e.add(f"v_mfma_f32_16x16x32_f16 {d}, {vq(a_quad)}, {vq(b_quad)}, {'0' if first else d}")
    else:
        e.add(f"{MF} {acc(m, n)}, {vq(a_quad)}, {vq(b_quad)}, {'0' if first else acc(m, n)}")


class Emit:
    """instruction list + the in-order LDS queue (counted lgkmcnt) + the shift state of the four read bases"""

    def __init__(self):
        self.lines = []
        self.issued = 0            # LDS reads issued so far
        self.done = -1             # highest read sequence number known to have returned
        self.holder = {}           # fragment register quad -> sequence number of the read that fills it
        self.shift = {"fa0": 0, "fa1": 0, "fb0": 0, "fb1": 0}

    def add(self, s):
        self.lines.append(s)

    def label(self, s):
        self.lines.append(s + ":")

    def read(self, quad, base, slot, off):
        """ds_read_b128 of a fragment of the half slab in `slot`: base register (shifted by 2 slots when needed) + immediate"""
        want = slot >> 1
        if self.shift[base] != want:
            self.add(f"v_add_u32_e32 %[{base}], {'0x%x' % (SHIFT if want else (1 << 32) - SHIFT)}, %[{base}]")
            self.shift[base] = want
        self.holder[quad] = self.issued
        self.issued += 1
        if "read" not in ABL:
            self.add(f"ds_read_b128 {vq(quad)}, %[{base}] offset:{(slot & 1) * HALF + off}")

    def need(self, quad):
        q = self.holder[quad]
        if q > self.done:
            if "read" not in ABL:
                self.add(f"s_waitcnt lgkmcnt({self.issued - q - 1})")
            self.done = q

    def text(self):
        return "\n".join(f'      "{ln}\\n\\t"' for ln in self.lines)


def piece(e, i, slot):
    if "dma" in ABL:
        return
    if i < 3:
        e.add(f"s_add_u32 m0, %[pda], {slot * HALF + i * 1024}")
        e.add("s_nop 0")
        e.add(f"global_load_lds_dwordx4 %[va{i}], s[{S_A}:{S_A + 1}]")
    else:
        e.add(f"s_add_u32 m0, %[pdb], {slot * HALF + (i - 3) * 1024}")
        e.add("s_nop 0")
        e.add(f"global_load_lds_dwordx4 %[vb{i - 3}], s[{S_B}:{S_B + 1}]")


def prime(e):
    """the k-step-0 fragments of slot 0 in the order a phase 2 issues them (the loop's counted waits expect exactly this queue)"""
    order = [("a", 0), ("b", 0), ("a", 1), ("a", 2), ("b", 1), ("b", 2), ("b", 3)]
    for kind, i in order:
        if kind == "a":
            e.read(AX + 4 * i, "fa0", 0, 2048 * i)
        else:
            e.read(BQ + 4 * i, "fb0", 0, 2048 * i)


def iteration(e, slot, first=False, dma=True, vm=5, last=False, barrier=None):
    """one half slab in `slot`; first: the accumulators start from the constant 0 (no zeroing pass); dma: issue the pieces of
    the half slab three ahead; vm: the vmcnt in front of the barrier (None: nothing left to wait for); last: no next half slab
    to read fragments of; barrier: default = not last (the cross-tile loop keeps it: its last iteration still issues pieces)"""
    if barrier is None:
        barrier = not last
    nslot = (slot + 1) % RING
    # ---- phase 1: k-step 0; reads of k-step 1 of this slot
    rd1 = {0: [("a", 0)], 1: [("a", 1)], 2: [("b", 0)], 3: [("a", 2)], 5: [("b", 1)], 8: [("b", 2)], 11: [("b", 3)]}
    for i in range(12):
        n, m = divmod(i, 3)
        e.need(AX + 4 * m)
        e.need(BQ + 4 * n)
        mfma(e, m, n, AX + 4 * m, BQ + 4 * n, first, 0)
        for kind, k in rd1.get(i, ()):
            if kind == "a":
                e.read(AY + 4 * k, "fa1", slot, 2048 * k)
            else:
                e.read(BQ + 4 * k, "fb1", slot, 2048 * k)
    if vm is not None and "dma" not in ABL:
        e.add(f"s_waitcnt vmcnt({vm})")
    if barrier and "bar" not in ABL:
        e.add("s_barrier")
    # ---- phase 2: k-step 1; reads of k-step 0 of the next slot; the pieces of half slab j + 3 -> slot (j - 1) % 4
    rd2 = {0: [("a", 0)], 2: [("b", 0)], 3: [("a", 1)], 4: [("a", 2)], 5: [("b", 1)], 8: [("b", 2)], 11: [("b", 3)]}
    pc = 0
    for i in range(12):
        n, m = divmod(i, 3)
        e.need(AY + 4 * m)
        e.need(BQ + 4 * n)
        mfma(e, m, n, AY + 4 * m, BQ + 4 * n, first and "m16" in ABL, 1)
        if not last:
            for kind, k in rd2.get(i, ()):
                if kind == "a":
                    e.read(AX + 4 * k, "fa0", nslot, 2048 * k)
                else:
                    e.read(BQ + 4 * k, "fb0", nslot, 2048 * k)
        if dma and i in PIECES_AFTER:
            piece(e, pc, (slot + RING - 1) % RING)
            pc += 1
    if dma and "dma" not in ABL:
        e.add(f"s_add_u32 s{S_A}, s{S_A}, 64")
        e.add(f"s_addc_u32 s{S_A + 1}, s{S_A + 1}, 0")
        e.add(f"s_add_u32 s{S_B}, s{S_B}, 64")
        e.add(f"s_addc_u32 s{S_B + 1}, s{S_B + 1}, 0")


# ================================================================================================
# Round 6: the same tile, ring and DMA schedule on v_mfma_f32_16x16x32_f16 ("x16": functions t384x_loop / t384x_loop_xt).
# Priced first (ablation "m16" above: two 16x16x32 per 32x32x16, garbage math): +5-7 % wall on the f16 flavour, +4-6 % on the
# f32 one under the board's power cap (profiles/r6_gemm_m16_pricing.txt) - the shape that moved the decoder attention in round 4.
#
# A half slab is ONE k-step of 32: 6 A fragments (16 rows each) x 8 B fragments (16 columns each) = 48 MFMAs per wave, each
# 16 x 16 x 32; fragment = 16 rows x 64 bytes, lane l reads the 16-byte chunk (l >> 4) of row (l & 15) - the same bytes per MFMA-FLOP
# from LDS as before (14 ds_read_b128 per half slab).  MFMA order: a outer (A fragment), n inner: A[a] is used by 8 consecutive
# MFMAs and dead after them - a ring of TWO quads, A[a + 1] read one block ahead - while the 8 B fragments stay resident for the
# whole half slab (32 registers) and are re-loaded in place for the next one, each right behind its last MFMA of block a = 5
# (8 MFMAs = 128 issue cycles ahead of its next use).  2 + 8 quads = the 40 fragment registers v[88:127] of the old plan.
# LDS image: chunk c of row r at position c ^ ((-(r >> 2)) & 3): conflict-free for this read's lane groups ({0-3,12-15,20-27}, ...:
# per row residue the four lanes of a group hold positions {s0, 1^s1, 1^s2, s3} with s = (0,3,2,1) = all different); the old
# image ((r >> 2) & 3) is 2-way conflicted for it.  The swizzle is applied to the LDS-DMA's per-lane SOURCE offsets (kernel).
# Accumulators: 12 floatx16 PINNED to physical registers (rows m = 1, 2: a[0:127]; row m = 0: v[24:87]) so the statement can
# name their quads; quad q = 2 mi + ni of accumulator (m, nb) is the 16 x 16 tile at rows 32 m + 16 mi, columns 32 nb + 16 ni:
# lane (c = l & 15, g = l >> 4), register j = row 4 g + j, column c.  The epilogues turn two quads into rows of 32 columns with
# v_permlane16_swap_b32 (kernel).  Per accumulator element the products are summed in ascending k, 32 per MFMA: NOT the bits of
# the 32x32x16 kernels (16 per MFMA) - parity is the oracle's tolerance, and the kernel choice never depends on M.
# ================================================================================================
AR, BR = 88, 96                    # A ring: 2 quads v[88:95]; B fragments: 8 quads v[96:127]
MF16 = "v_mfma_f32_16x16x32_f16"


def xquad(a, n):
    m, mi, nb, ni = a >> 1, a & 1, n >> 1, n & 1
    return acc_quad(m, nb, 2 * mi + ni)


def x_prime(e):
    """slot 0's fragments in the order block 5 of an iteration issues the next half slab's: B0, A0, B1 .. B7"""
    e.read(BR, "fb0", 0, 0)
    e.read(AR, "fa0", 0, 0)
    for n in range(1, 8):
        e.read(BR + 4 * n, "fb0", 0, 1024 * n)


def x_iteration(e, slot, first=False, dma=True, vm=5, last=False, barrier=None):
    if barrier is None:
        barrier = not last
    nslot = (slot + 1) % RING
    pc = 0
    # (block, gap) of the wave's 5 pieces of half slab j + 3, and the block in front of which the barrier sits: experiment
    # switches of the generator (T384X_PIECES, T384X_BAR), not product options.  Measured (profiles/r6_gemm_t384x_sched.txt,
    # f16 flavour, four shapes, two rounds): barrier before block 3 + a piece every 2nd gap 1177 / 1145 / 1194 TF/s (c1 / fc / c2);
    # before block 2 + a piece every 4th gap (this default) 1186-1194 / 1162-1174 / 1208-1215; before block 1 within 1 % of it.
    pieces_at = {tuple(int(v) for v in it.split(".")) for it in os.environ.get("T384X_PIECES", "2.1,2.5,3.1,3.5,4.1").split(",")}
    bar_at = int(os.environ.get("T384X_BAR", "2"))
    assert all(b >= bar_at for b, _ in pieces_at) and len(pieces_at) == 5
    for a in range(6):
        if a == bar_at:
            if vm is not None and "dma" not in ABL:
                e.add(f"s_waitcnt vmcnt({vm})")
            if barrier and "bar" not in ABL:
                e.add("s_barrier")
        for n in range(8):
            e.need(AR + 4 * (a & 1))
            e.need(BR + 4 * n)
            d = xquad(a, n)
            e.add(f"{MF16} {d}, {vq(AR + 4 * (a & 1))}, {vq(BR + 4 * n)}, {'0' if first else d}")
            if a < 5 and n == 0:
                e.read(AR + 4 * ((a + 1) & 1), "fa0", slot, 1024 * (a + 1))       # A[a + 1] of this half slab
            if a == 5 and not last:
                e.read(BR + 4 * n, "fb0", nslot, 1024 * n)                         # B[n] of the next half slab, in place
                if n == 0:
                    e.read(AR, "fa0", nslot, 0)                                    # ... and its A[0]
            if dma and (a, n) in pieces_at:
                piece(e, pc, (slot + RING - 1) % RING)
                pc += 1
    if dma and "dma" not in ABL:
        e.add(f"s_add_u32 s{S_A}, s{S_A}, 64")
        e.add(f"s_addc_u32 s{S_A + 1}, s{S_A + 1}, 0")
        e.add(f"s_add_u32 s{S_B}, s{S_B}, 64")
        e.add(f"s_addc_u32 s{S_B + 1}, s{S_B + 1}, 0")


def x_loop_stmt(xt=False):
    """the structure of loop_stmt (first block | steady block x nloop | final block; K = 128 (2 + nloop)) on x_iteration"""
    e = Emit()
    e.add(f"s_mov_b64 s[{S_A}:{S_A + 1}], %[asrc]")
    e.add(f"s_mov_b64 s[{S_B}:{S_B + 1}], %[bsrc]")
    e.add(f"s_mov_b32 s{S_CNT}, %[nloop]")
    x_prime(e)
    for s in range(4):
        x_iteration(e, s, first=(s == 0))
    e.add(f"s_cmp_eq_u32 s{S_CNT}, 0")
    e.add("s_cbranch_scc1 .Lt384x_final_%=")
    e.label(".Lt384x_loop_%=")
    entry = dict(e.shift)
    e.done = -1
    for s in range(4):
        x_iteration(e, s)
    e.add(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    e.add(f"s_cmp_lg_u32 s{S_CNT}, 0")
    e.add("s_cbranch_scc1 .Lt384x_loop_%=")
    assert e.shift == entry, (e.shift, entry)
    e.label(".Lt384x_final_%=")
    e.done = -1
    x_iteration(e, 0)
    if xt:
        if "dma" not in ABL:
            e.add(f"s_mov_b64 s[{S_A}:{S_A + 1}], %[anext]")
            e.add(f"s_mov_b64 s[{S_B}:{S_B + 1}], %[bnext]")
        x_iteration(e, 1)
        x_iteration(e, 2)
        x_iteration(e, 3, last=True, barrier=True)
        if "dma" not in ABL:
            e.add("s_waitcnt vmcnt(10)")
        if "bar" not in ABL:
            e.add("s_barrier")
    else:
        x_iteration(e, 1, dma=False)
        x_iteration(e, 2, dma=False, vm=0)
        x_iteration(e, 3, dma=False, vm=None, last=True)
    e.add("s_nop 15")
    e.add("s_nop 7")
    return e


def emit_x_loop(w, sfx, xt=False):
    name = f"t384x_loop{'_xt' if xt else ''}{sfx}"
    w(f"// ---- the 16x16x32 K loop {name}: acc[4 m + nb] = the 32 x 32 block (m, nb) as four 16 x 16 quads (2 mi + ni), written from 0")
    w(f"__device__ __forceinline__ void {name}(floatx16 (&acc)[12], const char* asrc, const char* bsrc, int nloop, unsigned pda,")
    w("    unsigned pdb, unsigned va0, unsigned va1, unsigned va2, unsigned vb0, unsigned vb1, unsigned fa0, unsigned fb0"
      + (", const char* anext, const char* bnext" if xt else "") + ") {")
    w("  asm volatile(")
    w(x_loop_stmt(xt).text())
    w("      : " + ", ".join(f'[c{m}{n}] "=&{acc_pin(m, n)}"(acc[{4 * m + n}])' for m in range(3) for n in range(4)) + ",")
    w('        [fa0] "+v"(fa0), [fb0] "+v"(fb0)')
    w('      : [asrc] "s"(asrc), [bsrc] "s"(bsrc), [nloop] "s"(nloop), [pda] "s"(pda), [pdb] "s"(pdb), [va0] "v"(va0), [va1] "v"(va1),')
    w('        [va2] "v"(va2), [vb0] "v"(vb0), [vb1] "v"(vb1)' + (', [anext] "s"(anext), [bnext] "s"(bnext)' if xt else ""))
    clob = ['"memory"', '"scc"'] + [f'"{r}"' for r in SGPR_CLOBBERS] + [f'"v{i}"' for i in range(AX, 128)]
    rows = [", ".join(clob[i:i + 16]) for i in range(0, len(clob), 16)]
    w("      : " + ",\n        ".join(rows) + ");")
    w("}")
    w("")


def loop_stmt(xt=False):
    """first block (half slabs 0-3) | steady block x nloop | final block.  K = 128 (2 + nloop).
    xt = False: the tile's own prologue has put half slabs 0-2 into slots 0-2; the final block issues no pieces after its first
    iteration and its waits shrink with the queue.
    xt = True (cross-tile): the ring never drains - the final block keeps the steady pattern with the DMA sources switched to the
    NEXT tile of this workgroup, whose half slabs 0, 1, 2 land in slots 0, 1, 2 while this tile's epilogue runs (LDS-free
    epilogues only); the statement ends with half slab 0 of the next tile landed for every wave (vmcnt(10) + barrier), so the
    next statement can prime its fragments at once."""
    e = Emit()
    e.add(f"s_mov_b64 s[{S_A}:{S_A + 1}], %[asrc]")
    e.add(f"s_mov_b64 s[{S_B}:{S_B + 1}], %[bsrc]")
    e.add(f"s_mov_b32 s{S_CNT}, %[nloop]")
    prime(e)
    for s in range(4):
        iteration(e, s, first=(s == 0))
    e.add(f"s_cmp_eq_u32 s{S_CNT}, 0")
    e.add("s_cbranch_scc1 .Lt384_final_%=")
    e.label(".Lt384_loop_%=")
    entry = dict(e.shift)                         # the back edge and the skip must arrive with the same base shifts
    e.done = -1                                   # back edge: nothing is known about the queue's head
    for s in range(4):
        iteration(e, s)
    e.add(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    e.add(f"s_cmp_lg_u32 s{S_CNT}, 0")
    e.add("s_cbranch_scc1 .Lt384_loop_%=")
    assert e.shift == entry, (e.shift, entry)
    e.label(".Lt384_final_%=")
    e.done = -1
    iteration(e, 0)                               # issues the pieces of the last half slab
    if xt:
        if "dma" not in ABL:
            e.add(f"s_mov_b64 s[{S_A}:{S_A + 1}], %[anext]")
            e.add(f"s_mov_b64 s[{S_B}:{S_B + 1}], %[bnext]")
        iteration(e, 1)                           # ... of the NEXT tile's half slab 0 -> slot 0
        iteration(e, 2)                           # ... 1 -> slot 1
        iteration(e, 3, last=True, barrier=True)  # ... 2 -> slot 2 (after the barrier: every wave has left slot 2)
        if "dma" not in ABL:
            e.add("s_waitcnt vmcnt(10)")          # the next tile's half slab 0 has landed (its groups 1 and 2 stay in flight)
        if "bar" not in ABL:
            e.add("s_barrier")
    else:
        iteration(e, 1, dma=False)                # outstanding: the two last groups -> vmcnt(5) still right
        iteration(e, 2, dma=False, vm=0)          # outstanding: the last group
        iteration(e, 3, dma=False, vm=None, last=True)
    e.add("s_nop 15")                             # the last MFMAs must have written their accumulators before the read-out
    e.add("s_nop 7")
    return e


VARIANTS = [("", ()), ("nodma", ("dma",)), ("noread", ("read",)), ("nobar", ("bar",)), ("mfmaonly", ("dma", "read", "bar")),
            ("m16", ("m16",))]


def emit_loop(w, sfx, xt=False):
    name = f"t384_loop{'_xt' if xt else ''}{sfx}"
    w(f"// ---- the K loop {name}: prime, first block, steady block, final block; acc[4 m + n] = accumulator (m, n), written from 0")
    w(f"__device__ __forceinline__ void {name}(floatx16 (&acc)[12], const char* asrc, const char* bsrc, int nloop, unsigned pda,")
    w("    unsigned pdb, unsigned va0, unsigned va1, unsigned va2, unsigned vb0, unsigned vb1, unsigned fa0, unsigned fa1, unsigned fb0,")
    w("    unsigned fb1" + (", const char* anext, const char* bnext" if xt else "") + ") {")
    w("  asm volatile(")
    w(loop_stmt(xt).text())
    if "m16" in ABL:
        w("      : " + ", ".join(f'[c{m}{n}] "=&{acc_pin(m, n)}"(acc[{4 * m + n}])' for m in range(3) for n in range(4)) + ",")
    else:
        w("      : " + ", ".join(f'[c{m}{n}] "=&{"v" if m == 0 else "a"}"(acc[{4 * m + n}])' for m in range(3) for n in range(4)) + ",")
    w('        [fa0] "+v"(fa0), [fa1] "+v"(fa1), [fb0] "+v"(fb0), [fb1] "+v"(fb1)')
    w('      : [asrc] "s"(asrc), [bsrc] "s"(bsrc), [nloop] "s"(nloop), [pda] "s"(pda), [pdb] "s"(pdb), [va0] "v"(va0), [va1] "v"(va1),')
    w('        [va2] "v"(va2), [vb0] "v"(vb0), [vb1] "v"(vb1)' + (', [anext] "s"(anext), [bnext] "s"(bnext)' if xt else ""))
    clob = ['"memory"', '"scc"'] + [f'"{r}"' for r in SGPR_CLOBBERS] + [f'"v{i}"' for i in range(AX, 128)]
    rows = [", ".join(clob[i:i + 16]) for i in range(0, len(clob), 16)]
    w("      : " + ",\n        ".join(rows) + ");")
    w("}")
    w("")


def emit():
    out, lab = [], []
    w = out.append
    w("// GENERATED by gen_gemm_t384.py - do not edit; see that file for the tile, the ring, the register plan and the schedule.")
    w("// clang-format off")
    w(f"#define T384_COMPILER_VGPRS {AX}   // the loop statement clobbers v[{AX}:127]; its 12 accumulators are \"=&a\" outputs")
    w("")
    for name, abl in VARIANTS:
        ABL.clear()
        ABL.update(abl)
        emit_loop(out.append if not name else lab.append, "" if not name else "_" + name)
        emit_loop(out.append if not name else lab.append, "" if not name else "_" + name, xt=True)
    ABL.clear()
    emit_x_loop(out.append, "")
    emit_x_loop(out.append, "", xt=True)
    for name, abl in VARIANTS[1:5]:
        ABL.clear()
        ABL.update(abl)
        emit_x_loop(lab.append, "_" + name)
        emit_x_loop(lab.append, "_" + name, xt=True)
    ABL.clear()
    w("// clang-format on")
    head = ["// GENERATED by dvd_amd/csrc/gen_gemm_t384.py --lab - do not edit.  TIMING ABLATIONS of the t384 K loop (lab builds only:",
            "// they compute garbage).", "// clang-format off"]
    return "\n".join(out) + "\n", "\n".join(head + lab + ["// clang-format on"]) + "\n"


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    prod, lab = emit()
    ppath = os.path.join(here, "gemm_t384_body.inc")
    lpath = os.path.normpath(os.path.join(here, "..", "..", "benchmarks", "lab", "csrc", "gemm_t384_abl.inc"))
    arg = sys.argv[1] if len(sys.argv) > 1 else ""
    if arg == "--check":
        sys.exit(0 if os.path.exists(ppath) and open(ppath).read() == prod else 1)
    path, text = (lpath, lab) if arg == "--lab" else (ppath, prod)
    if not (os.path.exists(path) and open(path).read() == text):      # identical content keeps its mtime (make)
        open(path, "w").write(text)
    print(f"wrote {path}: {text.count(chr(10))} lines")
