#!/usr/bin/env python3
"""Generates gypsum_amd/csrc/fft32_gen.hpp: fully unrolled 32-point complex DFT codelets for the wavefront transforms
(corr_core.hpp), as straight-line float code.

Decimation in time, radix 2, with the multiply-add folding the hand-rolled radix-2 loops of r01/r02 did not have: a butterfly
with a general twiddle w = c + i s is
    out1 = a + w b :  re = fma(b.re, c, fma(-b.im, s, a.re)),  im = fma(b.im, c, fma(b.re, s, a.im))      4 FMAs
    out2 = a - w b  =  2 a - out1 :  re = fma(2, a.re, -out1.re),  im likewise                              2 FMAs
six instructions instead of eight (twiddle: 2 mul + 2 fma, then 4 add/sub); butterflies with w = 1 or -+i stay four adds.
388 instructions per 32-point transform instead of 456.  Where a sample sits on entry (`in_slot`) and where a bin is left on
exit (`out_slot`) are free parameters of straight-line code, so the same network serves "natural in, bit-reversed out" (the
forward passes, and the inverse's second) and "bit-reversed in, natural out" (the inverse's first): no reordering pass exists.

    python tools/gen_fft32.py            # rewrites the header
    python tools/gen_fft32.py --check    # evaluates every generated codelet in float64 against numpy.fft (also run by the tests)
"""
from __future__ import annotations

import math
import sys
from pathlib import Path

OUT = Path(__file__).resolve().parents[1] / "gypsum_amd" / "csrc" / "fft32_gen.hpp"


def bitrev5(v: int) -> int:
    return int(f"{v:05b}"[::-1], 2)


class Emitter:
    def __init__(self):
        self.lines = []          # (name, op, args) with op in {"add","sub","fma","fnma2"}; executed by check(), printed by cpp()
        self.n = 0

    def tmp(self):
        self.n += 1
        return f"t{self.n}"

    def op(self, kind, *args):
        t = self.tmp()
        self.lines.append((t, kind, args))
        return t


def _f32(v: float) -> float:
    import numpy as np
    return float(np.float32(v))


def _ulp_up(v: float) -> float:
    import numpy as np
    return float(np.nextafter(np.float32(v), np.float32(np.inf), dtype=np.float32))


# The float32 twiddles the codelets are emitted with (r06).  Rounding cos and sin of 2 pi k / 32 to the nearest float32 each gives
# |w| < 1 for EVERY one of the seven constants (-2.9e-8, -2.9e-8, -7.5e-9, -1.7e-8, ...): a codelet then has a mean gain of 1 - 2.3e-8
# over its bins, a 1024-point transform 1 - 4.7e-8 and a forward + inverse pair 1 - 1e-7 -- measured on the device as a systematic
# -1.0 .. -1.6e-7 on every correlation peak (tools/gain_bias_probe.py).  A systematic gain on the prompt peak is a systematic change of
# the Costas loop's gain, which an unlocked loop amplifies (profiles/r06_experiments.txt item 6).  Moving two of the
# seven constants by one unit in the last place -- sin(2 pi 2 / 32) and sin(2 pi 4 / 32), and with them cos(2 pi 6 / 32) -- brings the
# codelet's mean gain error to 1e-11 (largest single bin 1.3e-8), at twiddle-angle errors (<= 4.2e-8 rad) of the size rounding to nearest
# leaves anyway (an exhaustive search over the +-1-ulp neighbours of the four independent pairs; `--check` re-derives the figure).
def quantised_twiddle(idx: int):
    """(cos, sin) of 2 pi idx / 32 as the float32 pair the codelets use, 0 < idx < 16, idx != 8."""
    def first_octant(k):          # k = 1..7
        c, s = _f32(math.cos(2.0 * math.pi * k / 32)), _f32(math.sin(2.0 * math.pi * k / 32))
        if k > 4:
            s2, c2 = first_octant(8 - k)
            return c2, s2
        if k in (2, 4):
            s = _ulp_up(s)
        return c, s
    if idx < 8:
        return first_octant(idx)
    c, s = first_octant(idx - 8)  # + pi / 2:  cos -> -sin, sin -> cos
    return -s, c


def gen(sign: int, in_slot, out_slot, exact: bool = True):
    """Returns (Emitter, outputs) for X[k] = sum_n x[n] exp(sign * 2 pi i n k / 32).  exact: float64 twiddles (for --check against
    numpy.fft); otherwise the float32 pairs of quantised_twiddle, as emitted."""
    e = Emitter()
    v = [(f"x[{in_slot[bitrev5(i)]}].x", f"x[{in_slot[bitrev5(i)]}].y") for i in range(32)]   # v[i] = time sample bitrev5(i)
    for s in range(5):
        half = 1 << s
        for g in range(0, 32, 2 * half):
            for j in range(half):
                (ar, ai), (br, bi) = v[g + j], v[g + j + half]
                num, den = j, 2 * half                                  # w = exp(sign * 2 pi i * j / (2 half))
                if num == 0:
                    o1 = (e.op("add", ar, br), e.op("add", ai, bi))
                    o2 = (e.op("sub", ar, br), e.op("sub", ai, bi))
                elif 4 * num == den:                                    # w = sign * i :  w b = sign * (-b.im, b.re)
                    if sign < 0:                                        # w = -i: w b = (b.im, -b.re)
                        o1 = (e.op("add", ar, bi), e.op("sub", ai, br))
                        o2 = (e.op("sub", ar, bi), e.op("add", ai, br))
                    else:                                               # w = +i: w b = (-b.im, b.re)
                        o1 = (e.op("sub", ar, bi), e.op("add", ai, br))
                        o2 = (e.op("add", ar, bi), e.op("sub", ai, br))
                else:
                    ang = sign * 2.0 * math.pi * num / den
                    c, sn = math.cos(ang), math.sin(ang)
                    if not exact:
                        c, sn = quantised_twiddle(num * (32 // den))
                        sn *= sign
                    # out1 = a + (c + i sn)(br + i bi) = (ar + br c - bi sn) + i (ai + bi c + br sn)
                    r1 = e.op("fma", bi, -sn, ar)
                    r1 = e.op("fma", br, c, r1)
                    i1 = e.op("fma", br, sn, ai)
                    i1 = e.op("fma", bi, c, i1)
                    o1 = (r1, i1)
                    o2 = (e.op("fnma2", ar, r1), e.op("fnma2", ai, i1))   # 2 a - out1
                v[g + j], v[g + j + half] = o1, o2
    return e, {out_slot[k]: v[k] for k in range(32)}


def gen_split_radix(sign: int):
    """The same transform as a split-radix (2/4) network with every multiplication folded into an FMA, for the count only (VERDICT
    r03 item 1d asked whether split-radix codelets would be shorter): X[k], X[k+N/2] = U[k] +- (w^k Z[k] + w^3k Z'[k]),
    X[k+N/4], X[k+3N/4] = U[k+N/4] -+ sign*i (w^k Z[k] - w^3k Z'[k]), each output pair as one FMA chain + one `2a - out`.
    A general-position L-butterfly (4 outputs, two radix-2 stages' worth) costs 20 instructions against 24 for four folded radix-2
    butterflies, but the radix-2 network has more multiplier-free butterflies (w = 1, -+i): 424 against 388 for N = 32 as generated
    here (404 if the k = N/8 butterflies were special-cased at 16 instructions each).  On a
    machine where a multiply-add costs what an add costs, radix 2 with `a - w b = 2a - (a + w b)` is already the short form."""
    e = Emitter()

    def rec(idx):
        n = len(idx)
        if n == 1:
            return [(f"x[{idx[0]}].x", f"x[{idx[0]}].y")]
        if n == 2:
            (ar, ai), (br, bi) = (f"x[{idx[0]}].x", f"x[{idx[0]}].y"), (f"x[{idx[1]}].x", f"x[{idx[1]}].y")
            return [(e.op("add", ar, br), e.op("add", ai, bi)), (e.op("sub", ar, br), e.op("sub", ai, bi))]
        u, z, zp = rec(idx[0::2]), rec(idx[1::4]), rec(idx[3::4])
        out = [None] * n
        q = n // 4
        for k in range(q):
            (ur, ui), (vr, vi) = u[k], u[k + q]
            (zr, zi), (yr, yi) = z[k], zp[k]
            a1, a3 = sign * 2.0 * math.pi * k / n, sign * 2.0 * math.pi * 3 * k / n
            c1, s1, c3, s3 = math.cos(a1), math.sin(a1), math.cos(a3), math.sin(a3)
            if k == 0:
                sr, si, dr, di = e.op("add", zr, yr), e.op("add", zi, yi), e.op("sub", zr, yr), e.op("sub", zi, yi)
                out[k] = (e.op("add", ur, sr), e.op("add", ui, si))
                out[k + 2 * q] = (e.op("sub", ur, sr), e.op("sub", ui, si))
                # sign*i*D = sign * (-di, dr)
                if sign < 0:
                    out[k + q] = (e.op("add", vr, di), e.op("sub", vi, dr))
                    out[k + 3 * q] = (e.op("sub", vr, di), e.op("add", vi, dr))
                else:
                    out[k + q] = (e.op("sub", vr, di), e.op("add", vi, dr))
                    out[k + 3 * q] = (e.op("add", vr, di), e.op("sub", vi, dr))
                continue
            # X[k] = U + w1 Z + w3 Z'
            r0 = e.op("fma", zi, -s1, ur); r0 = e.op("fma", zr, c1, r0); r0 = e.op("fma", yi, -s3, r0); r0 = e.op("fma", yr, c3, r0)
            i0 = e.op("fma", zr, s1, ui); i0 = e.op("fma", zi, c1, i0); i0 = e.op("fma", yr, s3, i0); i0 = e.op("fma", yi, c3, i0)
            out[k] = (r0, i0)
            out[k + 2 * q] = (e.op("fnma2", ur, r0), e.op("fnma2", ui, i0))
            # X[k+q] = V + sign*i*(w1 Z - w3 Z'):  A = w1 Z = (zr c1 - zi s1, zi c1 + zr s1),  B = w3 Z' likewise;  i*(A-B) = (-(Ai-Bi), Ar-Br)
            g = float(sign)
            r1 = e.op("fma", zi, -g * c1, vr); r1 = e.op("fma", zr, -g * s1, r1); r1 = e.op("fma", yi, g * c3, r1); r1 = e.op("fma", yr, g * s3, r1)
            i1 = e.op("fma", zr, g * c1, vi); i1 = e.op("fma", zi, -g * s1, i1); i1 = e.op("fma", yr, -g * c3, i1); i1 = e.op("fma", yi, g * s3, i1)
            out[k + q] = (r1, i1)
            out[k + 3 * q] = (e.op("fnma2", vr, r1), e.op("fnma2", vi, i1))
        return out

    v = rec(list(range(32)))
    return e, {k: v[k] for k in range(32)}


def cpp(name: str, comment: str, e: Emitter, outs) -> str:
    rows = [f"// {comment}", f"__device__ __forceinline__ void {name}(cf (&x)[32]) {{"]
    for t, kind, a in e.lines:
        if kind == "add":
            rows.append(f"    const float {t} = {a[0]} + {a[1]};")
        elif kind == "sub":
            rows.append(f"    const float {t} = {a[0]} - {a[1]};")
        elif kind == "fma":
            rows.append(f"    const float {t} = fmaf({a[0]}, {a[1]!r}f, {a[2]});")
        else:
            rows.append(f"    const float {t} = fmaf(2.0f, {a[0]}, -{a[1]});")
    for slot in range(32):
        r, i = outs[slot]
        rows.append(f"    x[{slot}] = make_float2({r}, {i});")
    rows.append("}")
    return "\n".join(rows)


def evaluate(e: Emitter, outs, x):
    import numpy as np
    env = {}
    def val(a):
        if isinstance(a, float):
            return a
        if a.startswith("x["):
            k = int(a[2:a.index("]")])
            return x[k].real if a.endswith(".x") else x[k].imag
        return env[a]
    for t, kind, a in e.lines:
        if kind == "add":
            env[t] = val(a[0]) + val(a[1])
        elif kind == "sub":
            env[t] = val(a[0]) - val(a[1])
        elif kind == "fma":
            env[t] = val(a[0]) * a[1] + val(a[2])
        else:
            env[t] = 2.0 * val(a[0]) - val(a[1])
    return np.array([complex(env[outs[s][0]], env[outs[s][1]]) for s in range(32)])


VARIANTS = [
    # name, sign, in_slot(n), out_slot(k), comment
    ("fft32_fwd_nat_br", -1, lambda n: n, bitrev5, "forward DFT32 (exp(-2 pi i n k / 32)): natural-order input, X[k] left in x[bitrev5(k)]"),
    ("fft32_inv_br_nat", +1, bitrev5, lambda k: k, "inverse DFT32 (exp(+2 pi i n k / 32), unnormalised): element n expected in x[bitrev5(n)], natural-order output"),
    ("fft32_inv_nat_br", +1, lambda n: n, bitrev5, "inverse DFT32 (unnormalised): natural-order input, X[k] left in x[bitrev5(k)]"),
]


def build(exact: bool = False):
    out = []
    for name, sign, ins, outs_, comment in VARIANTS:
        e, outs = gen(sign, [ins(n) for n in range(32)], [outs_(k) for k in range(32)], exact=exact)
        out.append((name, sign, ins, outs_, comment, e, outs))
    return out


def check():
    import numpy as np
    rng = np.random.default_rng(1)
    # the emitted (float32-twiddle) codelets: still the transform to float32 rounding, and their MEAN gain over matched tones is 1 to 1e-9
    for name, sign, ins, outs_, comment, e, outs in build(exact=False):
        gains = []
        worst = 0.0
        for k in range(32):
            t = np.exp(-sign * 2j * np.pi * k * np.arange(32) / 32)           # the tone bin k answers to
            x = np.zeros(32, dtype=complex)
            for n in range(32):
                x[ins(n)] = t[n]
            got = evaluate(e, outs, x)
            gains.append(abs(got[outs_(k)]) / 32.0 - 1.0)
            ref = np.zeros(32, dtype=complex)
            ref[outs_(k)] = 32.0
            worst = max(worst, float(np.abs(got - ref).max()))
        assert worst < 3e-6 and abs(float(np.mean(gains))) < 1e-9 and max(abs(g) for g in gains) < 2e-8, (name, worst, np.mean(gains))
        print(f"{name} as emitted (float32 twiddles): mean gain error over matched tones {np.mean(gains):+.1e}, largest bin {max(gains, key=abs):+.1e}, leakage {worst:.1e}")
    for name, sign, ins, outs_, comment, e, outs in build(exact=True):
        t = rng.standard_normal(32) + 1j * rng.standard_normal(32)          # t[n]: time samples
        x = np.zeros(32, dtype=complex)
        for n in range(32):
            x[ins(n)] = t[n]
        got = evaluate(e, outs, x)
        ref = np.fft.fft(t) if sign < 0 else np.fft.ifft(t) * 32
        want = np.zeros(32, dtype=complex)
        for k in range(32):
            want[outs_(k)] = ref[k]
        err = np.abs(got - want).max()
        assert err < 1e-12, (name, err)
        print(f"{name}: {len(e.lines)} float instructions, max |error| {err:.1e} (float64 evaluation vs numpy.fft)")
    for sign in (-1, +1):   # the split-radix alternative: correct, and longer
        e, outs = gen_split_radix(sign)
        t = rng.standard_normal(32) + 1j * rng.standard_normal(32)
        got = evaluate(e, outs, t)
        ref = np.fft.fft(t) if sign < 0 else np.fft.ifft(t) * 32
        err = np.abs(got - ref).max()
        assert err < 1e-12, ("split radix", sign, err)
        print(f"split-radix FMA network, sign {sign:+d}: {len(e.lines)} float instructions (not emitted: the radix-2 codelets above are shorter), "
              f"max |error| {err:.1e}")


def main():
    if "--check" in sys.argv:
        check()
        return
    parts = ["// fft32_gen.hpp -- GENERATED by tools/gen_fft32.py; do not edit.  32-point DFT codelets of the wavefront transforms:",
             "// radix-2 decimation in time, multiply-add folded butterflies (6 instructions per general twiddle), fully unrolled.",
             "#pragma once", "#include <hip/hip_runtime.h>", "", "namespace gyp {", "typedef float2 cf;", ""]
    for name, sign, ins, outs_, comment, e, outs in build():
        parts.append(cpp(name, f"{comment}.  {len(e.lines)} float instructions.", e, outs))
        parts.append("")
    parts.append("}  // namespace gyp")
    OUT.write_text("\n".join(parts) + "\n")
    print(f"wrote {OUT}")


if __name__ == "__main__":
    main()
