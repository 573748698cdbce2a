#!/usr/bin/env python3
"""Generates the table of csrc/mfpa_fastlog.h: for each of the 128 sub-intervals of z in [0.6875, 1.375) (equal steps of the offset
bit pattern, so 1.0 is an interval boundary) a centre c, invc = double(1 / c) (1 for the two intervals that touch 1.0) and -log(invc) as an
unevaluated pair of doubles; ln 2 as Ln2hi (42 significant bits, k * Ln2hi is exact for every double exponent k) + Ln2lo.
Exact decimal arithmetic at 80 digits; prints the C initialiser."""
import struct
from decimal import Decimal, getcontext

getcontext().prec = 80
OFF = 0x3FE6000000000000


def as_double(bits):
    return struct.unpack("<d", struct.pack("<Q", bits))[0]


def main():
    ln2 = Decimal(2).ln()
    ln2hi = float(int(ln2 * (Decimal(2) ** 42))) / 2.0 ** 42
    ln2lo = float(ln2 - Decimal(ln2hi))
    print("static MFPA_LOG_CONSTEXPR double MFPA_LN2HI = %s, MFPA_LN2LO = %s;" % (ln2hi.hex(), ln2lo.hex()))
    print("MFPA_LOG_TABLE_QUAL double mfpa_log_tab[128][3] = {")
    for i in range(128):
        zlo, zhi = as_double(OFF + (i << 45)), as_double(OFF + ((i + 1) << 45))
        invc = 1.0 if i in (79, 80) else float(Decimal(1) / ((Decimal(zlo) + Decimal(zhi)) / 2))
        logc = -(Decimal(invc).ln())
        hi = float(logc)
        lo = float(logc - Decimal(hi))
        print("  {%s, %s, %s}," % (invc.hex(), hi.hex(), lo.hex()))
    print("};")


if __name__ == "__main__":
    main()
