#!/usr/bin/env python
"""Prints the table rh_logit_link reads (rainier_amd/csrc/device/rh_prelude.hip.h, rh_lk_tab): for j = 0..256 the pair
   rc_j = 1 / (1 + j/256) rounded to binary64 (rc_0 = 1),   L_j = -log(rc_j) rounded to binary64  (log of the ROUNDED reciprocal,
so that log(1 + u) = L_j + log1p((1 + u) rc_j - 1) holds exactly for the tabulated pair).  Correctly rounded: Fraction -> float
and a 60-digit decimal logarithm.  With --exp: rh_ex_tab instead, 2^(j/128) for j = 0..127, correctly rounded (60-digit decimal
exponential: hi = the nearest double, lo = the remainder).  usage: python tools/gen_lk_table.py [--exp] > /tmp/tab.inc"""
import math
import sys
from decimal import Decimal, getcontext
from fractions import Fraction

getcontext().prec = 60
NB = 256
if "--exp" in sys.argv:
    getcontext().prec = 70
    NB = 128
    vals = []
    for j in range(NB):
        v = (Decimal(2).ln() * Decimal(j) / Decimal(NB)).exp()
        f = float(v)
        hi = min((math.nextafter(f, 0), f, math.nextafter(f, 4)), key=lambda c: abs(Decimal(c) - v))
        vals.append("%s, %s" % (hi.hex(), float(v - Decimal(hi)).hex()))      # (hi, lo = 2^(j/256) - hi)
    for i in range(0, NB, 3):
        print("  " + ", ".join(vals[i:i + 3]) + ("," if i + 3 < NB else ""))
    sys.exit(0)
out = []
for j in range(NB + 1):
    rc = float(Fraction(NB, NB + j))
    L = float(-(Decimal(Fraction(rc).numerator) / Decimal(Fraction(rc).denominator)).ln()) if j else 0.0
    out.append("%s, %s" % (rc.hex(), L.hex()))
for i in range(0, len(out), 3):
    print("  " + ", ".join(out[i:i + 3]) + ("," if i + 3 < len(out) else ""))
