#!/usr/bin/env python3
"""Generates tests/golden/decimal_kats.json: decimal literals and the float bits that the REFERENCE's own OBJ parser
(oracle/_ref/tinyobj_dump, the vendored header compiled unmodified) reads from them.  Random literals of every shape the
grammar allows plus the ones on which a correctly rounded strtod gives a different float (those are kept preferentially:
they are what distinguishes the reference's digit-by-digit accumulation).  TEST INFRASTRUCTURE; run where
/root/reference exists:  python oracle/gen_decimal_kats.py"""
import json
import os
import random
import struct
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
DUMP = os.path.join(HERE, "_ref", "tinyobj_dump")


def literals(n, seed=7):
    rnd = random.Random(seed)
    out = ["0", "-0", "+3.1417e+2", "-0.0E-3", "1.0324", "-1.41", "11e2", ".5", "-.5234", ".7e+2", "1e-3", "278.0", "0.1", "0.3",
           "548.799988", "559.200012", "1e10", "1E-10", "123456789", "0.000001", "16777217", "4.2e1", "5.", "-7.e2"]
    while len(out) < n:
        kind = rnd.random()
        ip = str(rnd.randint(0, 10 ** rnd.randint(0, 9)))
        fp = "".join(rnd.choice("0123456789") for _ in range(rnd.randint(0, 12)))
        s = rnd.choice(["", "", "-", "+"]) + (ip if kind > 0.1 or not fp else "") + ("." + fp if fp or kind < 0.2 else "")
        if rnd.random() < 0.5:
            s += rnd.choice("eE") + rnd.choice(["", "+", "-"]) + str(rnd.randint(0, 30))
        if s in ("", "+", "-", ".", "+.", "-."):
            continue
        out.append(s)
    return out


def reference_bits(lits):
    """float bits the reference parser reads for each literal (three per `v` line)."""
    lits = list(lits)
    while len(lits) % 3:
        lits.append("0")
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "k.obj"), "w") as f:
            for i in range(0, len(lits), 3):
                f.write("v %s %s %s\n" % tuple(lits[i:i + 3]))
        out = subprocess.run([DUMP, os.path.join(d, "k.obj"), d + "/"], capture_output=True, text=True, check=True).stdout.split()
    return [int(x, 16) for x in out]


def strtod_bits(s):
    try:
        return struct.unpack("<I", struct.pack("<f", float(s)))[0]
    except (ValueError, OverflowError):
        return None


def main():
    cand = literals(600000)
    bits = reference_bits(cand)
    differs = [(s, b) for s, b in zip(cand, bits) if strtod_bits(s) is not None and strtod_bits(s) != b]
    same = [(s, b) for s, b in zip(cand, bits) if strtod_bits(s) == b]
    keep = differs[:400] + same[:600]
    path = os.path.join(os.path.dirname(HERE), "tests", "golden", "decimal_kats.json")
    with open(path, "w") as f:
        json.dump({"_comment": "literal -> float bits read by the reference's vendored OBJ parser (oracle/gen_decimal_kats.py); "
                               "the first entries are literals on which a correctly rounded conversion gives another float",
                   "differs_from_strtod": len(differs[:400]), "kats": [[s, b] for s, b in keep]}, f)
    print(f"{len(cand)} literals, {len(differs)} differ from strtod; kept {len(keep)} -> {path}")


if __name__ == "__main__":
    main()
