#!/usr/bin/env python3
"""The tables of include/rp_libm_glibc.h, recomputed.

expf: T[i] = bits(RN(2^(i/32))) - (i << 47), 2^(i/32) to 60 digits with decimal, rounded to nearest double by float().
logf / powf: the sixteen 1/c are the authors' choice (doubles near the centres of the sixteenths of [OFF, 2 OFF), picked so that
log c rounds well) and are data; the second column is COMPUTED from them: log c = RN(-ln(1/c)) for logf, log2 c = RN(-log2(1/c)) for
powf — both recomputed here to 70 digits and compared with the header bit for bit.

  python scripts/glibc_tables.py          prints the exp2 table and checks both tables against the header
"""
import re
import struct
import sys
from decimal import Decimal, getcontext
from pathlib import Path

getcontext().prec = 70


def bits(x: float) -> int:
    return struct.unpack("<Q", struct.pack("<d", x))[0]


def exp2_table():
    ln2 = Decimal(2).ln()
    out = []
    for i in range(32):
        v = (ln2 * Decimal(i) / Decimal(32)).exp()
        out.append((bits(float(v)) - (i << 47)) & (2**64 - 1))  # float(Decimal) rounds to nearest, ties to even
    return out


def main() -> int:
    text = (Path(__file__).resolve().parent.parent / "include" / "rp_libm_glibc.h").read_text()
    tab = exp2_table()
    for i in range(0, 32, 4):
        print(", ".join(f"0x{v:016x}ull" for v in tab[i : i + 4]) + ",")
    body = text[text.index("#define RP_GLIBC_EXP2F_TAB_INIT") : text.index("return T[i")]  # the initialiser macro (shared with the LDS copy)
    have = [int(h, 16) for h in re.findall(r"0x([0-9a-f]{16})ull", body)]
    ok = have == tab
    print("exp2 table in the header:", "matches" if ok else "DIFFERS")
    ln2 = Decimal(2).ln()
    for fn, end, log in (("rp_glibc_logf", "const double A0 = -0x1.00ea", lambda v: -Decimal(v).ln()),
                         ("rp_glibc_powf", "const double A0 = 0x1.2761", lambda v: -Decimal(v).ln() / ln2)):
        at = text.index("RP_HD float " + fn)
        if fn == "rp_glibc_logf":  # its table is the macro in front of the function (shared with the kernels' LDS copy)
            body = text[text.index("#define RP_GLIBC_LOGF_TAB_INIT") : at]
        else:
            body = text[text.index("LT[16][2]", at) : text.index(end, at)]
        pairs = re.findall(r"\{(-?0x[0-9a-fp.+-]+), (-?0x[0-9a-fp.+-]+)\}", body)
        assert len(pairs) == 16, (fn, len(pairs))
        same = all(float(log(float.fromhex(invc))) + 0.0 == float.fromhex(logc) for invc, logc in pairs)
        print(f"{fn}: second column recomputed from 1/c:", "matches" if same else "DIFFERS")
        ok = ok and same
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
