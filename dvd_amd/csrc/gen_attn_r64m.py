#!/usr/bin/env python3
"""Generates dvd_amd/csrc/attn_r64m_body.inc: flash_attn_r64m_kernel's (attention.hip) whole KEY-TILE LOOP as ONE
`asm volatile` statement with HAND-ALLOCATED registers, plus the small statements around it (Q load, prologue, accumulator
read-out).

Why a generator: the loop is ~2300 instructions whose order, registers and wait counts follow from a schedule table (below);
written by hand it would be unreadable and unmaintainable.  The product build does not run this script - the generated
file is committed - `make gen` (or `python3 gen_attn_r64m.py`) regenerates it, and tests/test_abi.py checks that the
committed file is what the script produces.  `--lab` writes the timing ablations (benchmarks/lab/csrc/attn_r64m_abl.inc,
not committed: `make lab` generates it).

Register plan (one wave per SIMD: 256 architectural VGPRs + 256 AGPRs):
  AGPR  a[0:255]   O^T, 16 accumulator tiles: a[16 * (8 rb + dt) ...]
  VGPR  v[0:31]    the compiler's: the kernel carries amdgpu_num_vgpr(32), so hipcc's allocator is confined to v[0:31] and
                   the statements own v[32:255] and every AGPR outright (named literally in the text, never operands).
                   tests/test_abi.py greps the kernel's ISA: outside the asm statements no instruction names a VGPR above
                   v31 or an AGPR.  (v1 passed every hand-allocated register as a physical-register operand; with six tile
                   variants in a loop hipcc failed to coalesce the copies around the back-edge and spilled the state.)
        v[32:47]   packed P fragments      p00 p10 p01 p11       (row block, chunk) = (0,0) (1,0) (0,1) (1,1)
        v[48:63]   fragment ring, four slots of one K / V^T fragment (ds_read_b128)
        v[64:95]   S^T buffer A            rb 0: 64..79, rb 1: 80..95
        v[96:127]  S^T buffer B
        v[128:255] Q fragments             128 + 4 * (16 rb + ks)
  SGPR  s[80:89]   the loop's own scalars (clobbers of the statement): K / V^T DMA source pairs, tile counter, a scratch
                   value, the rescale mask, the variant to return to from the rare block
An exponential OVERWRITES the S^T element it consumes and a packed word is written straight into its P fragment register,
so the softmax needs no register of its own - which is what lets a tile's first eight exp units run one tile early (in
the previous tile's phase 2b) and puts exactly ONE unit beside every pair of MFMAs.

Schedule of tile t (S^T(t) in buffer t & 1, S^T(t+1) into the other; K / V^T ring slot t % 3):
  phase 1   16 fragment steps  S^T(t+1) += K(t+1) frag f . Q^T       | exp unit 8 + f of tile t
                               second gap: K(t+3) LDS-DMA piece (f = 3, 7, 11, 15), packs p00 (f = 9), p10 (10);
                               after the last piece the K source pair advances (not past the last tile: it is re-loaded)
  phase 2a   8 fragment steps  O^T += V^T(t) frag (chunk 0, g) . P00/P10 | unit 24 + g of tile t
                               second gap: lane-local maximum of S^T(t+1) (g = 0..3), test (4), pack p01 (5)
  tail       row-sum add of unit 31, pack p11; branch to the rare block if some lane's row maximum grew by more than THR
  vmcnt(4) + s_barrier
  phase 2b   8 fragment steps  O^T += V^T(t) frag (chunk 1, g) . P01/P11 | unit g of tile t + 1 (in the other buffer)
                               second gap: V^T(t+2) LDS-DMA piece (g = 1..4); V^T source pair advances, tile counter
Fragment ring: step n (0..31 of a tile) uses slot n & 3 after `s_waitcnt lgkmcnt(2)` and reads the fragment of step n + 3
into slot (n + 3) & 3 in its first gap; the last three steps of a tile read the first three K fragments of the next.

No glue (v2, v3): the tile comes in SIX variants, i = t % 6, so every LDS address is a loop-invariant base register + an
immediate and every LDS-DMA destination is `s_add_i32 m0, base, imm`; the DMA sources are one SGPR pair per stream + four
loop-invariant VGPR offsets.  v1's MFMA-only ablation ran 2412 cycles per tile for 64 MFMAs (2048): ~55 scalar address
instructions per tile, issued one per ~6 cycles by the only wave of the SIMD, were the largest single loss of the kernel.
v2 (two statements per tile, the compiler's `l += rs`, pointer advance and rare branch between them) ran 2196; its
MFMA-only ablation 2151, of which ~70 were still the statement boundaries.  v3 is one statement for the whole loop: row
sums accumulate straight into l (the early units of the NEXT tile into a side pair that the next tile's first steps add,
so the phantom units after the last tile are never counted), the pointers advance in SALU instructions in MFMA shadows,
and the rare rescale is an out-of-line block at the end of the statement shared by the six variants.
"""
import os
import sys

P00, P10, P01, P11 = 32, 36, 40, 44
FR0 = 48
SBUF = (64, 96)
Q0 = 128
KPIECE = 1056
KBYTES, VBYTES = 16 * KPIECE, 16384
S_KG, S_VG, S_TC, S_TMP, S_MASK, S_SEL = 80, 82, 84, 85, 86, 88
SGPR_CLOBBERS = [f"s{i}" for i in range(80, 90)]
RESCALE_THR_BITS = "0x41200000"      # 10.0f (log2 units), as in the other attention kernels


def vr(lo, n=1):
    return f"v{lo}" if n == 1 else f"v[{lo}:{lo + n - 1}]"


def urb(u):
    return (u & 1) if u < 16 else (0 if u < 24 else 1)


def uel(u):
    return (u >> 1) if u < 16 else (8 + u - 16 if u < 24 else 8 + u - 24)


def sreg(buf, u):
    """register of exp unit u's S^T element (and, afterwards, of its exponential)"""
    return SBUF[buf] + 16 * urb(u) + uel(u)


def frag(slot):
    return vr(FR0 + 4 * (slot & 3), 4)


def oreg(rb, dt):
    """O^T accumulator tile (row block rb, 32 output dims dt)"""
    return f"a[{16 * (8 * rb + dt)}:{16 * (8 * rb + dt) + 15}]"


def qreg(rb, ks):
    return vr(Q0 + 4 * (16 * rb + ks), 4)


MF = "v_mfma_f32_32x32x16_f16"

ABL = set()        # timing ablations (lab builds only; results are garbage): "eu", "pack", "dma", "read", "wait", "max", "bar"


class Stmt:
    def __init__(self):
        self.lines = []

    def add(self, s):
        if "m16" in ABL and s.startswith(MF):
            # POWER ablation: the same FLOPs from two 16x16x32 MFMAs on the first 8 accumulator registers (garbage math)
            d, a, b, c = [x.strip() for x in s[len(MF):].split(",")]
            lo = int(d[2:].split(":")[0])
            for h in range(2):
                dd = f"{d[0]}[{lo + 4 * h}:{lo + 4 * h + 3}]"
                self.lines.append(f"v_mfma_f32_16x16x32_f16 {dd}, {a}, {b}, {dd if c != '0' else '0'}")
            return
        if "read" in ABL and s.startswith("ds_read"):
            return
        if "wait" in ABL and s.startswith("s_waitcnt lgkmcnt"):
            return
        if "bar" in ABL and s.startswith("s_barrier"):
            return
        if "max" in ABL and (s.startswith("v_max") or s.startswith("v_cmp")):
            if s.startswith("v_cmp"):
                self.lines.append(f"s_mov_b64 s[{S_MASK}:{S_MASK + 1}], 0")
            return
        self.lines.append(s)

    def label(self, name):
        self.lines.append(name + ":")

    def eu(self, buf, u, add_from=None, acc="l", first=False):
        """exp unit u of the tile in `buf`, in place; then the row-sum add of unit add_from (same buffer) if given"""
        if "eu" in ABL:
            return
        x = vr(sreg(buf, u))
        self.add(f"v_fma_f32 {x}, {x}, %[c], -%[m{urb(u)}]")
        self.add(f"v_exp_f32_e32 {x}, {x}")
        if add_from is not None:
            a = f"%[{acc}{urb(add_from)}]"
            if first:
                self.add(f"v_mov_b32_e32 {a}, {vr(sreg(buf, add_from))}")
            else:
                self.add(f"v_add_f32_e32 {a}, {a}, {vr(sreg(buf, add_from))}")

    def pack(self, dst, buf, units):
        """four packed words of one P fragment: word j <- (units[2j], units[2j + 1])"""
        if "pack" in ABL:
            return
        for j in range(4):
            self.add(f"v_cvt_pk_f16_f32 {vr(dst + j)}, {vr(sreg(buf, units[2 * j]))}, {vr(sreg(buf, units[2 * j + 1]))}")

    def dma_m0(self, which, slot, i):
        """first gap: the LDS destination of piece i (the MFMA that follows separates the M0 write from its use)"""
        if "dma" in ABL:
            return
        imm = slot * (KBYTES if which == "k" else VBYTES) + i * (KPIECE if which == "k" else 1024)
        self.add(f"s_add_i32 m0, %[{which}dst], {imm}")

    def dma(self, which, i):
        if "dma" in ABL:
            return
        sg = S_KG if which == "k" else S_VG
        self.add(f"global_load_lds_dwordx4 %[{which}off{i}], s[{sg}:{sg + 1}]")

    def advance(self, which):
        """source pair += one tile, unless the stream has reached its last tile (which is then re-loaded)"""
        sg = S_KG if which == "k" else S_VG
        self.add(f"s_cmp_lt_i32 s{S_TC}, %[{which}lim]")
        self.add(f"s_cselect_b32 s{S_TMP}, %[{which}step], 0")
        self.add(f"s_add_u32 s{sg}, s{sg}, s{S_TMP}")
        self.add(f"s_addc_u32 s{sg + 1}, s{sg + 1}, 0")

    def text(self):
        return "\n".join(f'      "{ln}\\n\\t"' for ln in self.lines)


def read_for_step(n, slot):
    """(address operand, immediate) of the fragment that step n of tile t consumes (slot = t % 3); n >= 32: the next tile's"""
    if n < 16:
        return "kaddr", ((slot + 1) % 3) * KBYTES + n * 32            # K(t+1)
    if n < 24:
        return "vrel0", slot * VBYTES + (n - 16) * 2048                # V^T(t), chunk 0
    if n < 32:
        return "vrel1", slot * VBYTES + (n - 24) * 2048                # V^T(t), chunk 1
    return "kaddr", ((slot + 2) % 3) * KBYTES + (n - 32) * 32         # K(t+2)


def tile(s, var):
    """one key tile, variant var = t % 6"""
    par, slot = var & 1, var % 3
    sc, sn = par, 1 - par
    sn0, sn1 = vr(SBUF[sn], 16), vr(SBUF[sn] + 16, 16)
    for f in range(16):                                   # ---- phase 1
        n = f
        s.add("s_waitcnt lgkmcnt(2)")
        c_in = "0" if f == 0 else sn0
        s.add(f"{MF} {sn0}, {frag(n)}, {qreg(0, f)}, {c_in}")
        a, off = read_for_step(n + 3, slot)
        s.add(f"ds_read_b128 {frag(n + 3)}, %[{a}] offset:{off}")
        if f in (3, 7, 11, 15):
            s.dma_m0("k", slot, f >> 2)                   # K(t+3) -> K slot t % 3
        s.eu(sc, 8 + f, add_from=7 + f)                   # f = 0 adds unit 7, the last of the early units, straight to l
        if f in (0, 1) and "eu" not in ABL:               # the early units' side sums (rb f) join l
            s.add(f"v_add_f32_e32 %[l{f}], %[l{f}], %[e{f}]")
        c_in = "0" if f == 0 else sn1
        s.add(f"{MF} {sn1}, {frag(n)}, {qreg(1, f)}, {c_in}")
        if f in (3, 7, 11, 15):
            s.dma("k", f >> 2)
        if f == 15 and "dma" not in ABL:
            s.advance("k")
        if f == 9:
            s.pack(P00, sc, [0, 2, 4, 6, 8, 10, 12, 14])
        if f == 10:
            s.pack(P10, sc, [1, 3, 5, 7, 9, 11, 13, 15])
    S0, S1 = SBUF[sn], SBUF[sn] + 16
    for g in range(8):                                    # ---- phase 2a
        n = 16 + g
        s.add("s_waitcnt lgkmcnt(2)")
        s.add(f"{MF} {oreg(0, g)}, {frag(n)}, {vr(P00, 4)}, {oreg(0, g)}")
        a, off = read_for_step(n + 3, slot)
        s.add(f"ds_read_b128 {frag(n + 3)}, %[{a}] offset:{off}")
        s.eu(sc, 24 + g, add_from=23 + g)
        s.add(f"{MF} {oreg(1, g)}, {frag(n)}, {vr(P10, 4)}, {oreg(1, g)}")
        if g == 0:      # four chains: a = elements 0..6, b = 7..13 of each row block
            s.add(f"v_max3_f32 %[a0], v{S0 + 0}, v{S0 + 1}, v{S0 + 2}")
            s.add(f"v_max3_f32 %[b0], v{S0 + 7}, v{S0 + 8}, v{S0 + 9}")
            s.add(f"v_max3_f32 %[a1], v{S1 + 0}, v{S1 + 1}, v{S1 + 2}")
            s.add(f"v_max3_f32 %[b1], v{S1 + 7}, v{S1 + 8}, v{S1 + 9}")
        if g in (1, 2):
            k = 2 * g + 1
            s.add(f"v_max3_f32 %[a0], %[a0], v{S0 + k}, v{S0 + k + 1}")
            s.add(f"v_max3_f32 %[b0], %[b0], v{S0 + 7 + k}, v{S0 + 8 + k}")
            s.add(f"v_max3_f32 %[a1], %[a1], v{S1 + k}, v{S1 + k + 1}")
            s.add(f"v_max3_f32 %[b1], %[b1], v{S1 + 7 + k}, v{S1 + 8 + k}")
        if g == 3:
            s.add(f"v_max3_f32 %[a0], %[a0], %[b0], v{S0 + 14}")
            s.add(f"v_max3_f32 %[a1], %[a1], %[b1], v{S1 + 14}")
            s.add(f"v_max_f32_e32 %[a0], %[a0], v{S0 + 15}")
            s.add(f"v_max_f32_e32 %[a1], %[a1], v{S1 + 15}")
        if g == 4:      # d = max(a0 c - thr0, a1 c - thr1) > 0 in some lane <=> a row's maximum grew by more than THR
            s.add("v_fma_f32 %[b0], %[a0], %[c], -%[thr0]")
            s.add("v_fma_f32 %[b1], %[a1], %[c], -%[thr1]")
            s.add("v_max_f32_e32 %[b0], %[b0], %[b1]")
            s.add(f"v_cmp_lt_f32_e64 s[{S_MASK}:{S_MASK + 1}], 0, %[b0]")
        if g == 5:
            s.pack(P01, sc, [16, 17, 18, 19, 20, 21, 22, 23])
    if "eu" not in ABL:
        s.add(f"v_add_f32_e32 %[l1], %[l1], {vr(sreg(sc, 31))}")
    s.pack(P11, sc, [24, 25, 26, 27, 28, 29, 30, 31])
    # deferred rescale (rare): some lane saw its row's maximum over its 16 keys of tile t + 1 exceed m + THR.  O^T holds the
    # tiles up to t's chunk 0 and l the row sums up to t, both at the old reference - and so does the packed chunk 1 of P(t),
    # which enters O^T in phase 2b: the rare block scales all three.
    s.add(f"s_cmp_lg_u64 s[{S_MASK}:{S_MASK + 1}], 0")
    s.add(f"s_cbranch_scc1 .Lr64m_stub{var}_%=")
    s.label(f".Lr64m_back{var}_%=")
    s.add("s_waitcnt vmcnt(4)")
    s.add("s_barrier")
    for g in range(8):                                    # ---- phase 2b
        n = 24 + g
        s.add("s_waitcnt lgkmcnt(2)")
        s.add(f"{MF} {oreg(0, g)}, {frag(n)}, {vr(P01, 4)}, {oreg(0, g)}")
        a, off = read_for_step(n + 3, slot)
        s.add(f"ds_read_b128 {frag(n + 3)}, %[{a}] offset:{off}")
        if 1 <= g <= 4:
            s.dma_m0("v", (slot + 2) % 3, g - 1)          # V^T(t+2) -> V slot (t + 2) % 3
        # units 0..7 of tile t + 1: their sums go to the side pair e0 / e1 (unit 0 -> rb 0 and unit 1 -> rb 1 START them)
        s.eu(sn, g, add_from=(g - 1) if g else None, acc="e", first=g in (1, 2))
        s.add(f"{MF} {oreg(1, g)}, {frag(n)}, {vr(P11, 4)}, {oreg(1, g)}")
        if 1 <= g <= 4:
            s.dma("v", g - 1)
        if g == 4 and "dma" not in ABL:
            s.advance("v")
        if g == 5:
            s.add(f"s_add_i32 s{S_TC}, s{S_TC}, 1")


def rare_block(s):
    """out of line, shared by the six variants (s[S_SEL] = the variant to return to)"""
    s.label(".Lr64m_rare_%=")
    s.add("s_nop 15")                                     # the last PV MFMAs must have written O^T
    s.add("s_nop 7")
    for rb in range(2):
        t0, t1 = "%[t0]", "%[t1]"
        s.add(f"v_mul_f32_e32 {t0}, %[c], %[a{rb}]")
        s.add(f"v_mov_b32_e32 {t1}, {t0}")
        s.add("s_nop 1")
        s.add(f"v_permlane32_swap_b32 {t0}, {t1}")      # the row's other 16 keys live in lane ^ 32
        s.add("s_nop 1")
        s.add(f"v_max_f32_e32 {t0}, {t0}, {t1}")
        s.add(f"v_max_f32_e32 {t1}, %[m{rb}], {t0}")    # m_new
        s.add(f"v_sub_f32_e32 {t0}, %[m{rb}], {t1}")
        s.add(f"v_exp_f32_e32 {t0}, {t0}")              # alpha = 2^(m - m_new)
        s.add(f"v_mov_b32_e32 %[m{rb}], {t1}")
        s.add(f"v_add_f32_e32 %[thr{rb}], {RESCALE_THR_BITS}, {t1}")
        s.add("s_nop 0")
        s.add(f"v_mul_f32_e32 %[l{rb}], %[l{rb}], {t0}")
        s.add(f"v_cvt_pk_f16_f32 {t1}, {t0}, {t0}")
        base = P01 if rb == 0 else P11
        for j in range(4):
            s.add(f"v_pk_mul_f16 v{base + j}, v{base + j}, {t1}")
        for a0 in range(128 * rb, 128 * rb + 128, 4):
            for i in range(4):
                s.add(f"v_accvgpr_read_b32 %[t{1 + i}], a{a0 + i}")
            for i in range(4):
                s.add(f"v_mul_f32_e32 %[t{1 + i}], {t0}, %[t{1 + i}]")
            for i in range(4):
                s.add(f"v_accvgpr_write_b32 a{a0 + i}, %[t{1 + i}]")
    s.add("s_nop 1")
    for var in range(5):
        s.add(f"s_cmp_eq_u32 s{S_SEL}, {var}")
        s.add(f"s_cbranch_scc1 .Lr64m_back{var}_%=")
    s.add("s_branch .Lr64m_back5_%=")


def loop_stmt():
    s = Stmt()
    s.add(f"s_mov_b64 s[{S_KG}:{S_KG + 1}], %[kg]")
    s.add(f"s_mov_b64 s[{S_VG}:{S_VG + 1}], %[vg]")
    s.add(f"s_mov_b32 s{S_TC}, 0")
    s.label(".Lr64m_loop_%=")
    for var in range(6):
        tile(s, var)
        if var in (1, 3):                                 # the tile count is even
            s.add(f"s_cmp_ge_i32 s{S_TC}, %[nt]")
            s.add("s_cbranch_scc1 .Lr64m_end_%=")
    s.add(f"s_cmp_lt_i32 s{S_TC}, %[nt]")
    s.add("s_cbranch_scc1 .Lr64m_loop_%=")
    s.add("s_branch .Lr64m_end_%=")
    for var in range(6):
        s.label(f".Lr64m_stub{var}_%=")
        s.add(f"s_mov_b32 s{S_SEL}, {var}")
        s.add("s_branch .Lr64m_rare_%=")
    rare_block(s)
    s.label(".Lr64m_end_%=")
    # drain the LDS-DMA and the fragment reads still in flight (the last tiles re-load the last K / V^T tile and pre-read a
    # tile that does not exist): LDS must not be written after the workgroup has ended; and the last PV MFMAs must have
    # written O^T before it is read out
    s.add("s_waitcnt vmcnt(0) lgkmcnt(0)")
    s.add("s_nop 15")
    s.add("s_nop 7")
    return s


def prologue_s0():
    """S^T(0) into buffer 0 from K slot 0 (un-pipelined), then the lane-local maxima of both row blocks"""
    s = Stmt()
    s0, s1 = vr(SBUF[0], 16), vr(SBUF[0] + 16, 16)
    for f in range(3):
        s.add(f"ds_read_b128 {frag(f)}, %[kaddr] offset:{f * 32}")
    for f in range(16):
        if f + 3 < 16:
            s.add(f"ds_read_b128 {frag(f + 3)}, %[kaddr] offset:{(f + 3) * 32}")
            s.add("s_waitcnt lgkmcnt(3)")
        else:
            s.add(f"s_waitcnt lgkmcnt({15 - f})")
        s.add(f"{MF} {s0}, {frag(f)}, {qreg(0, f)}, {'0' if f == 0 else s0}")
        s.add(f"{MF} {s1}, {frag(f)}, {qreg(1, f)}, {'0' if f == 0 else s1}")
    s.add("s_nop 15")
    s.add("s_nop 7")
    for rb, name in ((0, "a0"), (1, "a1")):
        b = SBUF[0] + 16 * rb
        s.add(f"v_max3_f32 %[{name}], v{b}, v{b + 1}, v{b + 2}")
        for k in range(3, 15, 2):
            s.add(f"v_max3_f32 %[{name}], %[{name}], v{b + k}, v{b + k + 1}")
        s.add(f"v_max_f32_e32 %[{name}], %[{name}], v{b + 15}")
    return s


def prologue_units():
    """units 0..7 of tile 0 in place (buffer 0), the side sums of units 0..6, and the ring primed with K(1) frags 0..2"""
    s = Stmt()
    for f in range(3):
        s.add(f"ds_read_b128 {frag(f)}, %[kaddr] offset:{KBYTES + f * 32}")
    for u in range(8):
        s.eu(0, u, add_from=(u - 1) if u else None, acc="e", first=u in (1, 2))
    return s


VARIANTS = [("", ()), ("m16", ("m16",)), ("noeu", ("eu", "pack", "max")), ("nodma", ("dma",)), ("noread", ("read", "wait")), ("nobar", ("bar",)), ("mfmaonly_nobar", ("eu", "pack", "dma", "read", "wait", "max", "bar")),
            ("mfmaonly", ("eu", "pack", "dma", "read", "wait", "max"))]


def emit_loop(w, sfx):
    w(f"// ---- the key-tile loop{sfx}: six tile variants, the rare rescale block, the drain")
    w(f"__device__ __forceinline__ void r64m_loop{sfx}(float& l0, float& l1, float& m0, float& m1, float& thr0, float& thr1, float e0, float e1,")
    w("    const char* kg, const char* vg, int nt, unsigned kaddr, unsigned vrel0, unsigned vrel1, const unsigned (&koff)[4],")
    w("    const unsigned (&voff)[4], float c, unsigned kdst, unsigned vdst, unsigned kstep, unsigned vstep, int klim, int vlim) {")
    w("  float a0, a1, b0, b1, t0, t1, t2, t3, t4;")
    w("  asm volatile(")
    w(loop_stmt().text())
    w('      : [l0] "+v"(l0), [l1] "+v"(l1), [m0] "+v"(m0), [m1] "+v"(m1), [thr0] "+v"(thr0), [thr1] "+v"(thr1), [e0] "+v"(e0), [e1] "+v"(e1),')
    w('        [a0] "=&v"(a0), [a1] "=&v"(a1), [b0] "=&v"(b0), [b1] "=&v"(b1), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3),')
    w('        [t4] "=&v"(t4)')
    w('      : [kg] "s"(kg), [vg] "s"(vg), [nt] "s"(nt), [kaddr] "v"(kaddr), [vrel0] "v"(vrel0), [vrel1] "v"(vrel1), [koff0] "v"(koff[0]),')
    w('        [koff1] "v"(koff[1]), [koff2] "v"(koff[2]), [koff3] "v"(koff[3]), [voff0] "v"(voff[0]), [voff1] "v"(voff[1]), [voff2] "v"(voff[2]),')
    w('        [voff3] "v"(voff[3]), [c] "s"(c), [kdst] "s"(kdst), [vdst] "s"(vdst), [kstep] "s"(kstep), [vstep] "s"(vstep), [klim] "s"(klim),')
    w('        [vlim] "s"(vlim)')
    w('      : "memory", "scc", ' + ", ".join(f'"{r}"' for r in SGPR_CLOBBERS) + ");")
    w("}")
    w("")


def emit():
    """-> (product file text, lab file text: the timing ablations)"""
    out, lab = [], []
    lab.append("// GENERATED by dvd_amd/csrc/gen_attn_r64m.py --lab - do not edit.  TIMING ABLATIONS of the r64m loop (lab builds only:")
    lab.append("// they compute garbage) - which part of a tile costs what: see attention.hip (DVD_ATTN_R64M_ABL).")
    lab.append("// clang-format off")
    w = out.append
    w("// GENERATED by gen_attn_r64m.py - do not edit; see that file for the register plan and the schedule.")
    w("// clang-format off")
    w(f"#define R64M_COMPILER_VGPRS {P00}   // the kernel carries __attribute__((amdgpu_num_vgpr(R64M_COMPILER_VGPRS)))")
    w("")
    # Q load: 32 x 16 bytes per lane, two row pointers
    w("__device__ __forceinline__ void r64m_load_q(const _Float16* q0, const _Float16* q1) {")
    w("  asm volatile(")
    for rb in range(2):
        for ks in range(16):
            w(f'      "global_load_dwordx4 {qreg(rb, ks)}, %[q{rb}], off offset:{32 * ks}\\n\\t"')
    w('      "s_waitcnt vmcnt(0)"')
    w('      : : [q0] "v"(q0), [q1] "v"(q1) : "memory", "v255", "a255");   // the clobbers make the kernel descriptor allocate 256 + 256')
    w("}")
    w("")
    w("__device__ __forceinline__ void r64m_prologue_s0(unsigned kaddr, float& a0, float& a1) {")
    w("  asm volatile(")
    w(prologue_s0().text())
    w('      : [a0] "=&v"(a0), [a1] "=&v"(a1)')
    w('      : [kaddr] "v"(kaddr)')
    w('      : "memory");')
    w("}")
    w("")
    w("__device__ __forceinline__ void r64m_prologue_units(unsigned kaddr, float c, float m0, float m1, float& e0, float& e1) {")
    w("  asm volatile(")
    w(prologue_units().text())
    w('      : [e0] "=&v"(e0), [e1] "=&v"(e1)')
    w('      : [kaddr] "v"(kaddr), [c] "s"(c), [m0] "v"(m0), [m1] "v"(m1)')
    w('      : "memory");')
    w("}")
    w("")
    for abl_name, abl in VARIANTS:
        ABL.clear()
        ABL.update(abl)
        emit_loop(out.append if not abl_name else lab.append, "" if not abl_name else "_" + abl_name)
    ABL.clear()
    w("__device__ __forceinline__ void r64m_zero_o() {")
    w("  asm volatile(")
    for i in range(256):
        w(f'      "v_accvgpr_write_b32 a{i}, 0\\n\\t"')
    w('      "s_nop 1" ::: "memory");')
    w("}")
    w("")
    w("// one accumulator tile of O^T for the compiler (the epilogue): 16 registers of its own; BASE = 16 * (8 * rb + dt)")
    w("template <int BASE>")
    w("__device__ __forceinline__ floatx16 r64m_read_o() {")
    w("  floatx16 x;")
    w("  asm volatile(")
    for i in range(16):
        w(f'      "v_accvgpr_read_b32 %{i}, a[%{16 + i}]\\n\\t"')
    w('      "s_nop 0"')
    w("      : " + ", ".join(f'"=&v"(x[{i}])' for i in range(16)))
    w("      : " + ", ".join(f'"n"(BASE + {i})' for i in range(16)) + " : \"memory\");")
    w("  return x;")
    w("}")
    w("// clang-format on")
    lab.append("// clang-format on")
    return "\n".join(out) + "\n", "\n".join(lab) + "\n"


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    prod, lab = emit()
    ppath = os.path.join(here, "attn_r64m_body.inc")
    lpath = os.path.normpath(os.path.join(here, "..", "..", "benchmarks", "lab", "csrc", "attn_r64m_abl.inc"))
    arg = sys.argv[1] if len(sys.argv) > 1 else ""
    if arg == "--check":
        sys.exit(0 if os.path.exists(ppath) and open(ppath).read() == prod else 1)
    path, text = (lpath, lab) if arg == "--lab" else (ppath, prod)
    if not (os.path.exists(path) and open(path).read() == text):      # identical content keeps its mtime (make)
        open(path, "w").write(text)
    print(f"wrote {path}: {text.count(chr(10))} lines")
