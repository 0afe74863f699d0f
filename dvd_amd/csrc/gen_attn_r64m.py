#!/usr/bin/env python3
"""Generates dvd_amd/csrc/attn_r64m_body.inc: the tile loop of flash_attn_r64m_kernel (attention.hip) as four
`asm volatile` statements with HAND-ALLOCATED registers.

Why a generator: the loop body is ~800 instructions whose order, registers and wait counts follow from a schedule table
(below); written by hand once per parity it would be unreadable and unmaintainable.  The product build does not run this
script - the generated file is committed - `make gen` (or `python3 gen_attn_r64m.py`) regenerates it, and
tests/test_abi.py checks that the committed file is what the script produces.

Register plan (one wave per SIMD: 256 architectural VGPRs + 256 AGPRs):
  AGPR  a[0:255]   O^T, 16 accumulator tiles                     - allocated by the compiler ("+a" operands)
  VGPR  v[0:31]    everything the compiler keeps across statements (addresses, running max / sum, loop state)
        v[32:47]   packed P fragments      p00 p10 p01 p11       (row block, chunk) = (0,0) (1,0) (0,1) (1,1)
        v[48:63]   fragment ring, four slots of one K / V^T fragment (ds_read_b128)
        v[64:95]   S^T buffer A            rb 0: 64..79, rb 1: 80..95
        v[96:127]  S^T buffer B
        v[128:255] Q fragments             128 + 4 * (16 rb + ks)
  Every hand-allocated register is also passed as an operand with a physical-register constraint, so the compiler knows
  the state lives there between statements (see `pinned`).
An exponential OVERWRITES the S^T element it consumes and a packed word is written straight into its P fragment register,
so the softmax needs no register of its own - which is what lets a tile's first eight exp units run one iteration early
(in the previous iteration's phase 2b) and puts exactly ONE unit beside every pair of MFMAs (see attention.hip).

Schedule of iteration t (parity p = t & 1: S^T(t) in buffer p, S^T(t+1) into buffer 1 - p):
  statement A   phase 1   16 fragment steps  S^T(t+1) += K(t+1) frag f . Q^T       | unit 8 + f of tile t
                                             second gap: K(t+3) LDS-DMA piece (f = 3, 7, 11, 15), packs p00 (f = 9), p10 (10)
                phase 2a   8 fragment steps  O^T += V^T(t) frag (chunk 0, g) . P00/P10 | unit 24 + g of tile t
                                             second gap: lane-local maximum of S^T(t+1) (g = 0..3), test (4), pack p01 (5)
                tail       row-sum add of unit 31, pack p11
  (compiler)    l += row sums; rare: rescale O^T, l, p01 / p11, new reference
  statement B   vmcnt(4) + s_barrier
                phase 2b   8 fragment steps  O^T += V^T(t) frag (chunk 1, g) . P01/P11 | unit g of tile t + 1 (in buffer 1 - p)
                                             second gap: V^T(t+2) LDS-DMA piece (g = 1..4)
Fragment ring: step n (0..31 over A and B) uses slot n & 3 after `s_waitcnt lgkmcnt(2)` and reads the fragment of step n + 3
into slot (n + 3) & 3 in its first gap; the last three steps of B read the first three K fragments of the next iteration.
"""
import os
import sys

P00, P10, P01, P11 = 32, 36, 40, 44
FR0 = 48
SBUF = (64, 96)
Q0 = 128
KPIECE = 1056


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


def qreg(rb, ks):
    return vr(Q0 + 4 * (16 * rb + ks), 4)


MF = "v_mfma_f32_32x32x16_f16"


class Stmt:
    def __init__(self):
        self.lines = []

    def add(self, s):
        self.lines.append(s)

    def eu(self, buf, u, add_from=None):
        """exp unit u of the tile in `buf`, in place; then the row-sum add of unit add_from (same buffer) if given"""
        x = vr(sreg(buf, u))
        self.add(f"v_fma_f32 {x}, {x}, %[c], -%[m{urb(u)}]")
        self.add(f"v_exp_f32_e32 {x}, {x}")
        if add_from is not None:
            self.add(f"v_add_f32_e32 %[rs{urb(add_from)}], %[rs{urb(add_from)}], {vr(sreg(buf, add_from))}")

    def pack(self, dst, buf, units):
        """four packed words of one P fragment: word j <- (units[2j], units[2j + 1])"""
        for j in range(4):
            self.add(f"v_cvt_pk_f16_f32 {vr(dst + j)}, {vr(sreg(buf, units[2 * j]))}, {vr(sreg(buf, units[2 * j + 1]))}")

    def dma(self, which, i):
        self.add(f"s_mov_b32 m0, %[lds{which}{i}]")
        self.add("s_nop 0")
        self.add(f"global_load_lds_dwordx4 %[{which}off], %[gb{which}{i}]")

    def text(self):
        return "\n".join(f'      "{ln}\\n\\t"' for ln in self.lines)


def read_for_step(n):
    """(address operand, immediate) of the fragment that step n of an iteration consumes; n >= 32: next iteration's K"""
    if n < 16:
        return "kcur", n * 32
    if n < 24:
        return "vrd0", (n - 16) * 2048
    if n < 32:
        return "vrd1", (n - 24) * 2048
    return "knext", (n - 32) * 32


def stmt_a(par):
    sc, sn = par, 1 - par
    s = Stmt()
    sn0, sn1 = vr(SBUF[sn], 16), vr(SBUF[sn] + 16, 16)
    for f in range(16):                                   # ---- phase 1
        n = f
        s.add("s_waitcnt lgkmcnt(2)")
        c_in = "0" if f == 0 else sn0
        s.add(f"{MF} {sn0}, {frag(n)}, {qreg(0, f)}, {c_in}")
        a, off = read_for_step(n + 3)
        s.add(f"ds_read_b128 {frag(n + 3)}, %[{a}] offset:{off}")
        s.eu(sc, 8 + f, add_from=7 + f)
        c_in = "0" if f == 0 else sn1
        s.add(f"{MF} {sn1}, {frag(n)}, {qreg(1, f)}, {c_in}")
        if f in (3, 7, 11, 15):
            s.dma("k", f >> 2)
        if f == 9:
            s.pack(P00, sc, [0, 2, 4, 6, 8, 10, 12, 14])
        if f == 10:
            s.pack(P10, sc, [1, 3, 5, 7, 9, 11, 13, 15])
    S0, S1 = SBUF[sn], SBUF[sn] + 16
    for g in range(8):                                    # ---- phase 2a
        n = 16 + g
        s.add("s_waitcnt lgkmcnt(2)")
        s.add(f"{MF} %[o0_{g}], {frag(n)}, {vr(P00, 4)}, %[o0_{g}]")
        a, off = read_for_step(n + 3)
        s.add(f"ds_read_b128 {frag(n + 3)}, %[{a}] offset:{off}")
        s.eu(sc, 24 + g, add_from=23 + g)
        s.add(f"{MF} %[o1_{g}], {frag(n)}, {vr(P10, 4)}, %[o1_{g}]")
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
            s.add("v_cmp_lt_f32_e64 %[mask], 0, %[b0]")
        if g == 5:
            s.pack(P01, sc, [16, 17, 18, 19, 20, 21, 22, 23])
    s.add(f"v_add_f32_e32 %[rs1], %[rs1], {vr(sreg(sc, 31))}")
    s.pack(P11, sc, [24, 25, 26, 27, 28, 29, 30, 31])
    return s


def stmt_b(par):
    sn = 1 - par
    s = Stmt()
    s.add("s_waitcnt vmcnt(4)")
    s.add("s_barrier")
    for g in range(8):                                    # ---- phase 2b
        n = 24 + g
        s.add("s_waitcnt lgkmcnt(2)")
        s.add(f"{MF} %[o0_{g}], {frag(n)}, {vr(P01, 4)}, %[o0_{g}]")
        a, off = read_for_step(n + 3)
        s.add(f"ds_read_b128 {frag(n + 3)}, %[{a}] offset:{off}")
        s.eu(sn, g, add_from=(g - 1) if g else None)
        s.add(f"{MF} %[o1_{g}], {frag(n)}, {vr(P11, 4)}, %[o1_{g}]")
        if 1 <= g <= 4:
            s.dma("v", g - 1)
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
    """units 0..7 of tile 0 in place (buffer 0), the row sums of units 0..6, and the ring primed with K(1) frags 0..2"""
    s = Stmt()
    for f in range(3):
        s.add(f"ds_read_b128 {frag(f)}, %[kcur] offset:{f * 32}")
    for u in range(8):
        s.eu(0, u, add_from=(u - 1) if u else None)
    return s


def drain_text():
    return None


# Every hand-allocated register is ALSO an operand with a physical-register constraint ("{v[64:79]}"): the compiler then
# knows these values live there between the statements (it may not park a temporary of its own in them) and would copy
# rather than corrupt if it ever disagreed - correctness never rests on the compiler leaving registers alone.
def pinned(io):
    """operand list of the pinned state; io = '+' (in/out) or '' (input only, Q)"""
    ops = []
    for rb in range(2):
        ops.append(f'"+{{v[{SBUF[0] + 16 * rb}:{SBUF[0] + 16 * rb + 15}]}}"(st.sa[{rb}])')
    for rb in range(2):
        ops.append(f'"+{{v[{SBUF[1] + 16 * rb}:{SBUF[1] + 16 * rb + 15}]}}"(st.sb[{rb}])')
    for i in range(4):
        ops.append(f'"+{{v[{P00 + 4 * i}:{P00 + 4 * i + 3}]}}"(st.p[{i}])')
    for i in range(4):
        ops.append(f'"+{{v[{FR0 + 4 * i}:{FR0 + 4 * i + 3}]}}"(st.fr[{i}])')
    return ops


def pinned_q():
    return [f'"{{v[{Q0 + 4 * (16 * rb + ks)}:{Q0 + 4 * (16 * rb + ks) + 3}]}}"(st.q[{rb}][{ks}])' for rb in range(2)
            for ks in range(16)]


O_OPS = ", ".join(f'[o{rb}_{dt}] "+a"(o[{rb}][{dt}])' for rb in range(2) for dt in range(8))


def wrap(items, indent="        ", width=150):
    lines, cur = [], indent
    for it in items:
        if len(cur) + len(it) + 2 > width and cur.strip():
            lines.append(cur.rstrip())
            cur = indent
        cur += it + ", "
    lines.append(cur.rstrip().rstrip(","))
    return "\n".join(lines)


def emit():
    out = []
    w = out.append
    w("// GENERATED by gen_attn_r64m.py - do not edit; see that file for the register plan and the schedule.")
    w("// clang-format off")
    w("struct R64mState {        // the hand-allocated registers, as the compiler sees them (pinned operands)")
    w("  floatx16 sa[2], sb[2];  // S^T buffers A / B             v[64:95] / v[96:127]")
    w("  u32x4 p[4];             // packed P: p00 p10 p01 p11    v[32:47]")
    w("  half8 fr[4];            // fragment ring                v[48:63]")
    w("  half8 q[2][16];         // Q fragments                  v[128:255]")
    w("};")
    w("")
    # Q load: 32 x 16 bytes per lane, two row pointers
    w("__device__ __forceinline__ void r64m_load_q(R64mState& st, const _Float16* q0, const _Float16* q1) {")
    w("  asm volatile(")
    for rb in range(2):
        for ks in range(16):
            w(f'      "global_load_dwordx4 {qreg(rb, ks)}, %[q{rb}], off offset:{32 * ks}\\n\\t"')
    w('      "s_waitcnt vmcnt(0)"')
    w("      : " + wrap([x.replace('"{', '"={') for x in pinned_q()], "        ").lstrip())
    w('      : [q0] "v"(q0), [q1] "v"(q1) : "memory");')
    w("}")
    w("")
    w("__device__ __forceinline__ void r64m_prologue_s0(R64mState& st, unsigned kaddr, float& a0, float& a1) {")
    w("  asm volatile(")
    w(prologue_s0().text())
    w('      : [a0] "=&v"(a0), [a1] "=&v"(a1),')
    w(wrap([x.replace('"+{', '"=&{') for x in pinned("+")[:2]] + [x.replace('"+{', '"=&{') for x in pinned("+")[8:]]))
    w('      : [kaddr] "v"(kaddr),')
    w(wrap(pinned_q()))
    w('      : "memory");')
    w("}")
    w("")
    w("__device__ __forceinline__ void r64m_prologue_units(R64mState& st, unsigned kcur, float c, float m0, float m1, float& rs0,")
    w("                                                    float& rs1) {")
    w("  asm volatile(")
    w(prologue_units().text())
    w('      : [rs0] "+v"(rs0), [rs1] "+v"(rs1),')
    w(wrap(pinned("+")[:2] + [x.replace('"+{', '"=&{') for x in pinned("+")[8:]]))
    w('      : [kcur] "v"(kcur), [c] "s"(c), [m0] "v"(m0), [m1] "v"(m1)')
    w('      : "memory");')
    w("}")
    w("")
    for par in range(2):
        w(f"// ---- statement A, parity {par}: phase 1 (S^T(t+1) into buffer {1 - par}, units 8..23 of buffer {par}) and phase 2a")
        w(f"__device__ __forceinline__ void r64m_A{par}(R64mState& st, floatx16 (&o)[2][8], float& rs0, float& rs1, float& a0, float& a1,")
        w("    unsigned long long& mask, unsigned kcur, unsigned vrd0, unsigned vrd1, unsigned koff, float c, float m0, float m1,")
        w("    float thr0, float thr1, const char* gbk0, const char* gbk1, const char* gbk2, const char* gbk3, unsigned ldsk0,")
        w("    unsigned ldsk1, unsigned ldsk2, unsigned ldsk3) {")
        w("  float b0, b1;")
        w("  asm volatile(")
        w(stmt_a(par).text())
        w(f"      : {O_OPS},")
        w('        [rs0] "+v"(rs0), [rs1] "+v"(rs1), [a0] "=&v"(a0), [a1] "=&v"(a1), [b0] "=&v"(b0), [b1] "=&v"(b1), [mask] "=&s"(mask),')
        w(wrap(pinned("+")))
        w('      : [kcur] "v"(kcur), [vrd0] "v"(vrd0), [vrd1] "v"(vrd1), [koff] "v"(koff), [c] "s"(c), [m0] "v"(m0), [m1] "v"(m1),')
        w('        [thr0] "v"(thr0), [thr1] "v"(thr1), [gbk0] "s"(gbk0), [gbk1] "s"(gbk1), [gbk2] "s"(gbk2), [gbk3] "s"(gbk3),')
        w('        [ldsk0] "s"(ldsk0), [ldsk1] "s"(ldsk1), [ldsk2] "s"(ldsk2), [ldsk3] "s"(ldsk3),')
        w(wrap(pinned_q()))
        w('      : "memory");')
        w("}")
        w("")
        w(f"// ---- statement B, parity {par}: barrier, phase 2b (chunk 1 of tile t; units 0..7 of tile t + 1 in buffer {1 - par})")
        w(f"__device__ __forceinline__ void r64m_B{par}(R64mState& st, floatx16 (&o)[2][8], float& rs0, float& rs1, unsigned vrd1,")
        w("    unsigned knext, unsigned voff, float c, float m0, float m1, const char* gbv0, const char* gbv1, const char* gbv2,")
        w("    const char* gbv3, unsigned ldsv0, unsigned ldsv1, unsigned ldsv2, unsigned ldsv3) {")
        w("  asm volatile(")
        w(stmt_b(par).text())
        w(f"      : {O_OPS},")
        w('        [rs0] "+v"(rs0), [rs1] "+v"(rs1),')
        w(wrap(pinned("+")))
        w('      : [vrd1] "v"(vrd1), [knext] "v"(knext), [voff] "v"(voff), [c] "s"(c), [m0] "v"(m0), [m1] "v"(m1),')
        w('        [gbv0] "s"(gbv0), [gbv1] "s"(gbv1), [gbv2] "s"(gbv2), [gbv3] "s"(gbv3),')
        w('        [ldsv0] "s"(ldsv0), [ldsv1] "s"(ldsv1), [ldsv2] "s"(ldsv2), [ldsv3] "s"(ldsv3),')
        w(wrap(pinned_q()))
        w('      : "memory");')
        w("}")
        w("")
    # rare branch: scale the packed chunk 1 of one row block
    for rb, idx, base in ((0, 2, P01), (1, 3, P11)):
        w(f"__device__ __forceinline__ void r64m_scale_p{rb}(R64mState& st, unsigned a2) {{")
        w("  asm volatile(")
        for j in range(4):
            w(f'      "v_pk_mul_f16 v{base + j}, v{base + j}, %[a2]\\n\\t"')
        w('      "s_nop 1"')
        w(f'      : "+{{v[{base}:{base + 3}]}}"(st.p[{idx}]) : [a2] "v"(a2));')
        w("}")
    w("// clang-format on")
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    text = emit()
    path = os.path.join(here, "attn_r64m_body.inc")
    if len(sys.argv) > 1 and sys.argv[1] == "--check":
        sys.exit(0 if os.path.exists(path) and open(path).read() == text else 1)
    open(path, "w").write(text)
    print(f"wrote {path}: {text.count(chr(10))} lines")
