#!/usr/bin/env python3
"""Generates dvd_amd/csrc/attn_h64m_body.inc: the key-tile loop of flash_attn_h64m_kernel (attention.hip) - head_dim 64
on the decoder kernel's recipe (gen_attn_r64m.py; read that file first): 64 query rows per wave, the product S^T(t+1) and
the softmax of tile t overlapped by pipelining across 32-key tiles, hand-allocated registers, the whole loop ONE asm
statement in six glue-free tile variants.

What head_dim 64 changes:
  * a tile is 16 MFMAs (32x32x16) instead of 64 for the same 32 exponential units: FOUR units beside every pair of MFMAs.
    One wave cannot issue that beside its MFMAs (~100 issue cycles per 64-cycle step), so the kernel runs TWO waves per
    SIMD (two 256-row workgroups per CU): a wave needs 160 VGPRs + 64 AGPRs, and two waves issue VALU work concurrently
    (benchmarks/lab/valu_lab.hip);
  * steps per tile: 4 (S^T: K fragments ks = 0..3) + 2 (PV chunk 0: dt = 0, 1) + 2 (PV chunk 1) = 8; the fragment ring still
    reads three steps ahead, so the barrier sits between phase 1 and phase 2a (the pre-reads of the next tile's K follow it);
  * one 1-KiB LDS-DMA piece per wave and stream and tile (K tile 32 keys x 128 B, V^T tile 64 dims x 64 B);
  * K image: natural key rows of 128 B, 16-byte chunks XOR-swizzled by (row >> 1) & 7 (conflict-free for ds_read_b128's lane
    groups with rows taken in kappa order); the XOR depends on ks, so a lane keeps one K fragment base per ks (4 VGPRs).

Register plan (per wave; two waves per SIMD):
  AGPR  a[0:63]    O^T: tile (rb, dt) at a[16 (2 rb + dt) ...]
  VGPR  v[0:31]    the compiler's (amdgpu_num_vgpr(32))
        v[32:47]   packed P fragments p00 p10 p01 p11
        v[48:63]   fragment ring (four slots)
        v[64:95]   S^T buffer A (rb 0: 64..79, rb 1: 80..95), v[96:127] buffer B
        v[128:159] Q fragments: 128 + 4 (4 rb + ks)
  SGPR  s[80:89]   loop scalars (clobbers)

Schedule of tile t (exp unit numbering as in r64m: 0..15 chunk 0 of both row blocks interleaved, 16..23 / 24..31 chunk 1):
  phase 1   steps 0..3   S^T(t+1) += K(t+1) frag ks . Q^T        | units 8 + 4 f .. + 3 of tile t; K(t+3) piece (f = 1);
                                                                   packs p00 (f = 2), p10 (f = 3)
  vmcnt(1) + s_barrier
  phase 2a  steps 4, 5   O^T += V^T(t) frag (chunk 0, dt) . P00/P10 | units 24 + 4 g .. + 3; lane-local maxima of S^T(t+1) and the
                                                                   test; packs p01 (g = 0), p11 (tail); rare block
  phase 2b  steps 6, 7   O^T += V^T(t) frag (chunk 1, dt) . P01/P11 | units 4 g .. + 3 of tile t + 1; V^T(t+2) piece (g = 0)
"""
import os
import sys

P00, P10, P01, P11 = 32, 36, 40, 44
FR0 = 48
SBUF = (64, 96)
Q0 = 128
KS, DT = 4, 2
KBYTES, VBYTES = 4096, 4096
S_KG, S_VG, S_TC, S_TMP, S_MASK, S_SEL = 80, 82, 84, 85, 86, 88
SGPR_CLOBBERS = [f"s{i}" for i in range(80, 90)]
RESCALE_THR_BITS = "0x41200000"
MF = "v_mfma_f32_32x32x16_f16"
ABL = set()        # timing ablations (lab builds only; garbage results): "eu", "pack", "dma", "read", "wait", "max", "bar"


def vr(lo, n=1):
    return f"v{lo}" if n == 1 else f"v[{lo}:{lo + n - 1}]"


def urb(u):
    return (u & 1) if u < 16 else (0 if u < 24 else 1)


def uel(u):
    return (u >> 1) if u < 16 else (8 + u - 16 if u < 24 else 8 + u - 24)


def sreg(buf, u):
    return SBUF[buf] + 16 * urb(u) + uel(u)


def frag(slot):
    return vr(FR0 + 4 * (slot & 3), 4)


def oreg(rb, dt):
    return f"a[{16 * (2 * rb + dt)}:{16 * (2 * rb + dt) + 15}]"


def qreg(rb, ks):
    return vr(Q0 + 4 * (KS * rb + ks), 4)


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

    def eu2(self, buf, u, acc="l", first=()):
        """exp units u and u + 1 interleaved (a unit's v_exp does not follow its own v_fma back to back: two waves per SIMD
        hide much, but 32 dependent pairs per tile showed in the cycles), then the row-sum adds of units u - 1 and u"""
        if "eu" in ABL:
            return
        xs = [vr(sreg(buf, u + i)) for i in range(2)]
        for i in range(2):
            self.add(f"v_fma_f32 {xs[i]}, {xs[i]}, %[c], -%[m{urb(u + i)}]")
        for i in range(2):
            self.add(f"v_exp_f32_e32 {xs[i]}, {xs[i]}")
        for w in (u - 1, u):
            if w < 0:
                continue
            a = f"%[{acc}{urb(w)}]"
            if w + 1 in first:
                self.add(f"v_mov_b32_e32 {a}, {vr(sreg(buf, w))}")
            else:
                self.add(f"v_add_f32_e32 {a}, {a}, {vr(sreg(buf, w))}")

    def pack(self, dst, buf, units):
        if "pack" in ABL:
            return
        for j in range(4):
            self.add(f"v_cvt_pk_f16_f32 {vr(dst + j)}, {vr(sreg(buf, units[2 * j]))}, {vr(sreg(buf, units[2 * j + 1]))}")

    def dma_m0(self, which, slot):
        if "dma" in ABL:
            return
        self.add(f"s_add_i32 m0, %[{which}dst], {slot * (KBYTES if which == 'k' else VBYTES)}")

    def dma(self, which):
        if "dma" in ABL:
            return
        sg = S_KG if which == "k" else S_VG
        self.add(f"global_load_lds_dwordx4 %[{which}off], s[{sg}:{sg + 1}]")

    def advance(self, which):
        if "dma" in ABL:
            return
        sg = S_KG if which == "k" else S_VG
        self.add(f"s_cmp_lt_i32 s{S_TC}, %[{which}lim]")
        self.add(f"s_cselect_b32 s{S_TMP}, %[{which}step], 0")
        self.add(f"s_add_u32 s{sg}, s{sg}, s{S_TMP}")
        self.add(f"s_addc_u32 s{sg + 1}, s{sg + 1}, 0")

    def text(self):
        return "\n".join(f'      "{ln}\\n\\t"' for ln in self.lines)


def read_for_step(n, slot):
    """(address operand, immediate) of the fragment that step n of tile t consumes (slot = t % 3); n >= 8: the next tile's"""
    if n < 4:
        return f"kf{n}", ((slot + 1) % 3) * KBYTES                    # K(t+1), ks = n
    if n < 6:
        return "vrel0", slot * VBYTES + (n - 4) * 2048                 # V^T(t), chunk 0, dt = n - 4
    if n < 8:
        return "vrel1", slot * VBYTES + (n - 6) * 2048                 # V^T(t), chunk 1
    return f"kf{n - 8}", ((slot + 2) % 3) * KBYTES                    # K(t+2)


def tile(s, var):
    par, slot = var & 1, var % 3
    sc, sn = par, 1 - par
    sn0, sn1 = vr(SBUF[sn], 16), vr(SBUF[sn] + 16, 16)
    for f in range(4):                                    # ---- phase 1
        n = f
        s.add("s_waitcnt lgkmcnt(2)")
        s.add(f"{MF} {sn0}, {frag(n)}, {qreg(0, f)}, {'0' if f == 0 else sn0}")
        a, off = read_for_step(n + 3, slot)
        s.add(f"ds_read_b128 {frag(n + 3)}, %[{a}] offset:{off}")
        if f == 1:
            s.dma_m0("k", slot)                           # K(t+3) -> K slot t % 3
        s.eu2(sc, 8 + 4 * f)                              # u = 8 adds unit 7, the last early unit, straight to l
        if f == 0 and "eu" not in ABL:                    # the early units' side sums join l
            s.add("v_add_f32_e32 %[l0], %[l0], %[e0]")
            s.add("v_add_f32_e32 %[l1], %[l1], %[e1]")
        s.add(f"{MF} {sn1}, {frag(n)}, {qreg(1, f)}, {'0' if f == 0 else sn1}")
        if f == 1:
            s.dma("k")
            s.advance("k")
        s.eu2(sc, 10 + 4 * f)
        if f == 2:
            s.pack(P00, sc, [0, 2, 4, 6, 8, 10, 12, 14])
        if f == 3:
            s.pack(P10, sc, [1, 3, 5, 7, 9, 11, 13, 15])
    s.add("s_waitcnt vmcnt(1)")
    s.add("s_barrier")
    S0, S1 = SBUF[sn], SBUF[sn] + 16
    mx = [f"v_max3_f32 %[a0], v{S0 + 0}, v{S0 + 1}, v{S0 + 2}", f"v_max3_f32 %[b0], v{S0 + 7}, v{S0 + 8}, v{S0 + 9}",
          f"v_max3_f32 %[a1], v{S1 + 0}, v{S1 + 1}, v{S1 + 2}", f"v_max3_f32 %[b1], v{S1 + 7}, v{S1 + 8}, v{S1 + 9}"]
    for k in (3, 5):
        mx += [f"v_max3_f32 %[a0], %[a0], v{S0 + k}, v{S0 + k + 1}", f"v_max3_f32 %[b0], %[b0], v{S0 + 7 + k}, v{S0 + 8 + k}",
               f"v_max3_f32 %[a1], %[a1], v{S1 + k}, v{S1 + k + 1}", f"v_max3_f32 %[b1], %[b1], v{S1 + 7 + k}, v{S1 + 8 + k}"]
    mx += [f"v_max3_f32 %[a0], %[a0], %[b0], v{S0 + 14}", f"v_max3_f32 %[a1], %[a1], %[b1], v{S1 + 14}",
           f"v_max_f32_e32 %[a0], %[a0], v{S0 + 15}", f"v_max_f32_e32 %[a1], %[a1], v{S1 + 15}",
           "v_fma_f32 %[b0], %[a0], %[c], -%[thr0]", "v_fma_f32 %[b1], %[a1], %[c], -%[thr1]", "v_max_f32_e32 %[b0], %[b0], %[b1]",
           f"v_cmp_lt_f32_e64 s[{S_MASK}:{S_MASK + 1}], 0, %[b0]"]
    for g in range(2):                                    # ---- phase 2a
        n = 4 + g
        s.add("s_waitcnt lgkmcnt(2)")
        s.add(f"{MF} {oreg(0, g)}, {frag(n)}, {vr(P00, 4)}, {oreg(0, g)}")
        a, off = read_for_step(n + 3, slot)
        s.add(f"ds_read_b128 {frag(n + 3)}, %[{a}] offset:{off}")
        s.eu2(sc, 24 + 4 * g)
        for it in mx[10 * g:10 * g + 5]:
            s.add(it)
        s.add(f"{MF} {oreg(1, g)}, {frag(n)}, {vr(P10, 4)}, {oreg(1, g)}")
        s.eu2(sc, 26 + 4 * g)
        for it in mx[10 * g + 5:10 * g + 10]:
            s.add(it)
        if g == 0:
            s.pack(P01, sc, [16, 17, 18, 19, 20, 21, 22, 23])
    if "eu" not in ABL:
        s.add(f"v_add_f32_e32 %[l1], %[l1], {vr(sreg(sc, 31))}")
    s.pack(P11, sc, [24, 25, 26, 27, 28, 29, 30, 31])
    s.add(f"s_cmp_lg_u64 s[{S_MASK}:{S_MASK + 1}], 0")
    s.add(f"s_cbranch_scc1 .Lh64m_stub{var}_%=")
    s.label(f".Lh64m_back{var}_%=")
    for g in range(2):                                    # ---- phase 2b
        n = 6 + g
        s.add("s_waitcnt lgkmcnt(2)")
        s.add(f"{MF} {oreg(0, g)}, {frag(n)}, {vr(P01, 4)}, {oreg(0, g)}")
        a, off = read_for_step(n + 3, slot)
        s.add(f"ds_read_b128 {frag(n + 3)}, %[{a}] offset:{off}")
        if g == 0:
            s.dma_m0("v", (slot + 2) % 3)                 # V^T(t+2) -> V slot (t + 2) % 3
        s.eu2(sn, 4 * g, acc="e", first=(1, 2))           # early units of tile t + 1: side sums e0 / e1
        s.add(f"{MF} {oreg(1, g)}, {frag(n)}, {vr(P11, 4)}, {oreg(1, g)}")
        if g == 0:
            s.dma("v")
            s.advance("v")
        s.eu2(sn, 4 * g + 2, acc="e", first=(1, 2))
        if g == 1:
            s.add(f"s_add_i32 s{S_TC}, s{S_TC}, 1")


def rare_block(s):
    s.label(".Lh64m_rare_%=")
    s.add("s_nop 15")
    s.add("s_nop 7")
    for rb in range(2):
        t0, t1 = "%[t0]", "%[t1]"
        s.add(f"v_mul_f32_e32 {t0}, %[c], %[a{rb}]")
        s.add(f"v_mov_b32_e32 {t1}, {t0}")
        s.add("s_nop 1")
        s.add(f"v_permlane32_swap_b32 {t0}, {t1}")
        s.add("s_nop 1")
        s.add(f"v_max_f32_e32 {t0}, {t0}, {t1}")
        s.add(f"v_max_f32_e32 {t1}, %[m{rb}], {t0}")
        s.add(f"v_sub_f32_e32 {t0}, %[m{rb}], {t1}")
        s.add(f"v_exp_f32_e32 {t0}, {t0}")
        s.add(f"v_mov_b32_e32 %[m{rb}], {t1}")
        s.add(f"v_add_f32_e32 %[thr{rb}], {RESCALE_THR_BITS}, {t1}")
        s.add("s_nop 0")
        s.add(f"v_mul_f32_e32 %[l{rb}], %[l{rb}], {t0}")
        s.add(f"v_cvt_pk_f16_f32 {t1}, {t0}, {t0}")
        base = P01 if rb == 0 else P11
        for j in range(4):
            s.add(f"v_pk_mul_f16 v{base + j}, v{base + j}, {t1}")
        for a0 in range(32 * rb, 32 * rb + 32, 4):
            for i in range(4):
                s.add(f"v_accvgpr_read_b32 %[t{1 + i}], a{a0 + i}")
            for i in range(4):
                s.add(f"v_mul_f32_e32 %[t{1 + i}], {t0}, %[t{1 + i}]")
            for i in range(4):
                s.add(f"v_accvgpr_write_b32 a{a0 + i}, %[t{1 + i}]")
    s.add("s_nop 1")
    for var in range(5):
        s.add(f"s_cmp_eq_u32 s{S_SEL}, {var}")
        s.add(f"s_cbranch_scc1 .Lh64m_back{var}_%=")
    s.add("s_branch .Lh64m_back5_%=")


def loop_stmt():
    s = Stmt()
    s.add(f"s_mov_b64 s[{S_KG}:{S_KG + 1}], %[kg]")
    s.add(f"s_mov_b64 s[{S_VG}:{S_VG + 1}], %[vg]")
    s.add(f"s_mov_b32 s{S_TC}, 0")
    s.label(".Lh64m_loop_%=")
    for var in range(6):
        tile(s, var)
        if var in (1, 3):
            s.add(f"s_cmp_ge_i32 s{S_TC}, %[nt]")
            s.add("s_cbranch_scc1 .Lh64m_end_%=")
    s.add(f"s_cmp_lt_i32 s{S_TC}, %[nt]")
    s.add("s_cbranch_scc1 .Lh64m_loop_%=")
    s.add("s_branch .Lh64m_end_%=")
    for var in range(6):
        s.label(f".Lh64m_stub{var}_%=")
        s.add(f"s_mov_b32 s{S_SEL}, {var}")
        s.add("s_branch .Lh64m_rare_%=")
    rare_block(s)
    s.label(".Lh64m_end_%=")
    s.add("s_waitcnt vmcnt(0) lgkmcnt(0)")
    s.add("s_nop 15")
    s.add("s_nop 7")
    return s


def prologue_s0():
    s = Stmt()
    s0, s1 = vr(SBUF[0], 16), vr(SBUF[0] + 16, 16)
    for f in range(3):
        s.add(f"ds_read_b128 {frag(f)}, %[kf{f}]")
    s.add(f"ds_read_b128 {frag(3)}, %[kf3]")
    for f in range(4):
        s.add(f"s_waitcnt lgkmcnt({3 - f})")
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
    s = Stmt()
    for f in range(3):
        s.add(f"ds_read_b128 {frag(f)}, %[kf{f}] offset:{KBYTES}")
    for u in range(8):
        s.eu(0, u, add_from=(u - 1) if u else None, acc="e", first=u in (1, 2))
    return s


VARIANTS = [("", ()), ("noeu", ("eu", "pack", "max")), ("m16", ("m16",)), ("mfmaonly", ("eu", "pack", "dma", "read", "wait", "max"))]
KF = ", ".join(f'[kf{i}] "v"(kf[{i}])' for i in range(4))


def emit_loop(w, sfx):
    w(f"// ---- the key-tile loop{sfx}: six tile variants, the rare rescale block, the drain")
    w(f"__device__ __forceinline__ void h64m_loop{sfx}(float& l0, float& l1, float& m0, float& m1, float& thr0, float& thr1, float e0, float e1,")
    w("    const char* kg, const char* vg, int nt, const unsigned (&kf)[4], unsigned vrel0, unsigned vrel1, unsigned koff, unsigned voff,")
    w("    float c, unsigned kdst, unsigned vdst, unsigned kstep, unsigned vstep, int klim, int vlim) {")
    w("  float a0, a1, b0, b1, t0, t1, t2, t3, t4;")
    w("  asm volatile(")
    w(loop_stmt().text())
    w('      : [l0] "+v"(l0), [l1] "+v"(l1), [m0] "+v"(m0), [m1] "+v"(m1), [thr0] "+v"(thr0), [thr1] "+v"(thr1), [e0] "+v"(e0), [e1] "+v"(e1),')
    w('        [a0] "=&v"(a0), [a1] "=&v"(a1), [b0] "=&v"(b0), [b1] "=&v"(b1), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3),')
    w('        [t4] "=&v"(t4)')
    w(f'      : [kg] "s"(kg), [vg] "s"(vg), [nt] "s"(nt), {KF}, [vrel0] "v"(vrel0), [vrel1] "v"(vrel1),')
    w('        [koff] "v"(koff), [voff] "v"(voff), [c] "s"(c), [kdst] "s"(kdst), [vdst] "s"(vdst), [kstep] "s"(kstep), [vstep] "s"(vstep),')
    w('        [klim] "s"(klim), [vlim] "s"(vlim)')
    w('      : "memory", "scc", ' + ", ".join(f'"{r}"' for r in SGPR_CLOBBERS) + ");")
    w("}")
    w("")


def emit():
    out, lab = [], []
    lab.append("// GENERATED by dvd_amd/csrc/gen_attn_h64m.py --lab - do not edit.  TIMING ABLATIONS of the h64m loop (lab builds only).")
    lab.append("// clang-format off")
    w = out.append
    w("// GENERATED by gen_attn_h64m.py - do not edit; see that file for the register plan and the schedule.")
    w("// clang-format off")
    w(f"#define H64M_COMPILER_VGPRS {P00}   // the kernel carries __attribute__((amdgpu_num_vgpr(H64M_COMPILER_VGPRS)))")
    w("")
    w("__device__ __forceinline__ void h64m_load_q(const _Float16* q0, const _Float16* q1) {")
    w("  asm volatile(")
    for rb in range(2):
        for ks in range(KS):
            w(f'      "global_load_dwordx4 {qreg(rb, ks)}, %[q{rb}], off offset:{32 * ks}\\n\\t"')
    w('      "s_waitcnt vmcnt(0)"')
    w('      : : [q0] "v"(q0), [q1] "v"(q1) : "memory", "v159", "a63");   // the clobbers: 160 VGPRs + 64 AGPRs per wave')
    w("}")
    w("")
    w("__device__ __forceinline__ void h64m_zero_o() {")
    w("  asm volatile(")
    for i in range(64):
        w(f'      "v_accvgpr_write_b32 a{i}, 0\\n\\t"')
    w('      "s_nop 1" ::: "memory");')
    w("}")
    w("")
    w("__device__ __forceinline__ void h64m_prologue_s0(const unsigned (&kf)[4], float& a0, float& a1) {")
    w("  asm volatile(")
    w(prologue_s0().text())
    w('      : [a0] "=&v"(a0), [a1] "=&v"(a1)')
    w(f'      : {KF}')
    w('      : "memory");')
    w("}")
    w("")
    w("__device__ __forceinline__ void h64m_prologue_units(const unsigned (&kf)[4], float c, float m0, float m1, float& e0, float& e1) {")
    w("  asm volatile(")
    w(prologue_units().text())
    w('      : [e0] "=&v"(e0), [e1] "=&v"(e1)')
    w(f'      : {KF}, [c] "s"(c), [m0] "v"(m0), [m1] "v"(m1)')
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
    ppath = os.path.join(here, "attn_h64m_body.inc")
    lpath = os.path.normpath(os.path.join(here, "..", "..", "benchmarks", "lab", "csrc", "attn_h64m_abl.inc"))
    arg = sys.argv[1] if len(sys.argv) > 1 else ""
    if arg == "--check":
        sys.exit(0 if os.path.exists(ppath) and open(ppath).read() == prod else 1)
    path, text = (lpath, lab) if arg == "--lab" else (ppath, prod)
    if not (os.path.exists(path) and open(path).read() == text):      # identical content keeps its mtime (make)
        open(path, "w").write(text)
    print(f"wrote {path}: {text.count(chr(10))} lines")
