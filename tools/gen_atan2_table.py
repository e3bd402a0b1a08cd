"""Writes blom_amd/csrc/atan2_libm_table.h: the table cij[241][7] of glibc's double precision atan / atan2 (the IBM Accurate Mathematical
Library routines, sysdeps/ieee754/dbl-64/e_atan2.c with atnat2.h / uatan.tbl): for i = 0 .. 240 a sample point x_i near (i + 16) / 256
chosen by its authors so that atan(x_i) is unusually close to a double, then atan(x_i) and the Taylor coefficients of atan at x_i,
    cij[i] = { x_i, atan(x_i), 1/(1+x^2), ~ -x/(1+x^2)^2, ~ (3x^2-1)/(3(1+x^2)^3), ~ x(1-x^2)/(1+x^2)^4, ~ (5x^4-10x^2+1)/(5(1+x^2)^5) }
(atan(x_i) is stored to 2e-18 relative -- eight bits better than a double's half ulp, which is what the choice of x_i buys --, the first
derivative correctly rounded, the higher coefficients fitted over the interval rather than pure Taylor values: within 1e-11, 2e-10, 3e-4,
3e-4 of them).
The x_i are the result of a search and cannot be regenerated from first principles, so the table is READ from the libm of this machine
(located by its first sample point; the shared object's data) -- glibc's numbers, third-party constants like exp's and pow's tables -- and
every entry is checked against the relations above in exact rational arithmetic (atan itself against a 60-digit series) before it is
written.  `--check` compares the committed header with the libm of this machine."""
import os
import struct
import sys
from fractions import Fraction

N = 241


def bits(d):
    return struct.unpack("<Q", struct.pack("<d", d))[0]


def libm_table(path="/lib/x86_64-linux-gnu/libm.so.6"):
    f = open(path, "rb").read()
    key = struct.pack("<Q", 0x3FB0400665E0244E)
    o = f.find(key)
    if o < 0:
        return None
    return [tuple(struct.unpack_from("<d", f, o + 56 * i + 8 * j)[0] for j in range(7)) for i in range(N)]


def atan_fr(x, terms=400):
    """atan of the rational 0 < x <= 1 to ~1e-60: two halvings atan(x) = 2 atan(x / (1 + sqrt(1 + x^2))) with a 200-bit rational
    square root, then the alternating series"""
    def sqrt_fr(a):
        n = (a.numerator << 800) // a.denominator
        r = int(n ** 0.5) if n < (1 << 1000) else 1 << ((n.bit_length() + 1) // 2)
        for _ in range(12):
            r = (r + n // r) // 2
        return Fraction(r, 1 << 400)
    k = 0
    while x > Fraction(1, 4):
        x = x / (1 + sqrt_fr(1 + x * x))
        k += 1
    s, t, x2 = Fraction(0), x, x * x
    for n in range(terms):
        s += t / (2 * n + 1) if n % 2 == 0 else -t / (2 * n + 1)
        t *= x2
        if t < Fraction(1, 1 << 260):
            break
    return s * (1 << k)


def check_entry(e):
    x = Fraction(e[0])
    q = 1 + x * x
    want = [atan_fr(x), 1 / q, -x / q ** 2, (3 * x * x - 1) / (3 * q ** 3), x * (1 - x * x) / q ** 4, (5 * x ** 4 - 10 * x * x + 1) / (5 * q ** 5)]
    for got, w, tol in zip(e[1:], want, (1e-17, 1.2e-16, 1e-10, 1e-8, 1e-3, None)):
        if tol is None:                       # (the last coefficient passes through zero near x = 0.325: absolute there)
            if abs(Fraction(got) - w) > Fraction(1, 10000):
                return False
        elif abs(Fraction(got) - w) > abs(w) * Fraction(tol):
            return False
    return True


if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "blom_amd", "csrc", "atan2_libm_table.h")
    ref = libm_table()
    if ref is None:
        sys.exit("the table was not found in this machine's libm")
    if "--check" in sys.argv:
        txt = open(path).read()
        vals = [int(t.rstrip("ul,\\"), 16) for t in txt.split() if t.startswith("0x")]
        same = vals == [bits(v) for e in ref for v in e]
        print("committed header == libm's table:", same)
        sys.exit(0 if same else 1)
    bad = [i for i, e in enumerate(ref) if not check_entry(e)]
    near = [i for i, e in enumerate(ref) if abs(e[0] - (i + 16) / 256.0) > 1.5 / 256]
    if bad or near:
        sys.exit(f"entries that fail the relations: {bad}; sample points off their grid: {near}")
    with open(path, "w") as f:
        f.write("// written by tools/gen_atan2_table.py -- do not edit.  glibc's cij[241][7] (atnat2.h): sample point x_i, atan(x_i) and five Taylor\n"
                "// coefficients of atan at x_i, read from libm.so.6 (GLIBC 2.35) and checked entry by entry against those relations\n"
                "#define ATAN2_LIBM_TABLE \\\n")
        for i, e in enumerate(ref):
            f.write("  " + ", ".join(f"0x{bits(v):016x}ull" for v in e) + (", \\\n" if i < N - 1 else "\n"))
    print("wrote", os.path.normpath(path), "--", N, "entries, each checked against atan and its derivatives at its sample point")
