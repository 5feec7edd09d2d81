#!/usr/bin/env python3
"""Extract the reference's hard-coded known-answer data for the field arithmetic into
tests/golden/field_kats.json.

Run in the build container only (it reads /root/reference, which does not exist on the GPU
box); the JSON it writes is the committed fixture.  Only DATA is extracted -- operands and
expected values of the reference's own unit tests and its constant conversion tables:

  * tower mul KATs          crates/field/src/binary_field.rs  test_bin{2,4,8,16,64}b_mul
  * multiplicative generators  crates/field/src/binary_field.rs:740-747
  * BINARY_TO_POLYVAL_TRANSFORMATION / POLYVAL_TO_BINARY_TRANSFORMATION (128 constants each)
                            crates/field/src/polyval.rs:516-646, 654-784
  * POLYVAL mul / square KAT   crates/field/src/polyval.rs test_mul / test_sqr
"""
import json
import os
import re
import sys

REF = os.environ.get("BINIUS_REFERENCE", "/root/reference")


def main():
    bf = open(os.path.join(REF, "crates/field/src/binary_field.rs")).read()
    pv = open(os.path.join(REF, "crates/field/src/polyval.rs")).read()

    out = {"source": "IrreducibleOSS/binius @ 2025-09-19, crates/field/src/{binary_field,polyval}.rs"}

    # --- tower mul KATs: BFn::new(a) * BFn::new(b), BFn::new(c)
    kat_re = re.compile(
        r"BF(\d+)::(?:new|from)\((0x[0-9a-fA-F]+)\)\s*\*\s*BF\1::(?:new|from)\((0x[0-9a-fA-F]+)\),\s*"
        r"BF\1::(?:new|from)\((0x[0-9a-fA-F]+)\)",
        re.S,
    )
    kats = {}
    for m in kat_re.finditer(bf):
        bits = int(m.group(1))
        kats.setdefault(str(bits), []).append([m.group(2), m.group(3), m.group(4)])
    assert set(kats) >= {"2", "4", "8", "16", "64"}, sorted(kats)
    out["tower_mul_kats"] = kats

    # --- multiplicative generators
    gens = {}
    for m in re.finditer(r"binary_field!\(pub BinaryField(\d+)b\(\w+\),\s*(?:U\d+::new\()?(0x[0-9a-fA-F]+)", bf):
        gens[m.group(1)] = m.group(2)
    assert set(gens) == {"1", "2", "4", "8", "16", "32", "64", "128"}, gens
    out["multiplicative_generators"] = gens

    # --- conversion tables
    def table(name, elem):
        start = pv.index("pub const " + name)
        end = pv.index("]);", start)
        vals = re.findall(elem + r"\((0x[0-9a-fA-F]+)\)", pv[start:end])
        assert len(vals) == 128, (name, len(vals))
        return vals

    out["binary_to_polyval"] = table("BINARY_TO_POLYVAL_TRANSFORMATION", "BinaryField128bPolyval")
    out["polyval_to_binary"] = table("POLYVAL_TO_BINARY_TRANSFORMATION", "BinaryField128b")

    # --- POLYVAL KATs (operands are given in the non-Montgomery domain: ::new(x))
    m = re.search(
        r"fn test_mul\(\).*?new\((0x[0-9a-f]+)\)\s*\*\s*BinaryField128bPolyval::new\((0x[0-9a-f]+)\),\s*"
        r"BinaryField128bPolyval::new\((0x[0-9a-f]+)\)",
        pv,
        re.S,
    )
    out["polyval_mul_kat"] = [m.group(1), m.group(2), m.group(3)]
    m = re.search(
        r"fn test_sqr\(\).*?new\((0x[0-9a-f]+)\)\),\s*BinaryField128bPolyval::new\((0x[0-9a-f]+)\)", pv, re.S
    )
    out["polyval_sqr_kat"] = [m.group(1), m.group(2)]
    m = re.search(r"const ONE: Self = Self\((0x[0-9a-f]+)\)", pv)
    out["polyval_one_montgomery"] = m.group(1)
    m = re.search(r"fn to_montgomery.*?Self\((0x[0-9a-f]+)\)", pv, re.S)
    out["polyval_to_montgomery_factor"] = m.group(1)

    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "field_kats.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", dst, {k: (len(v) if hasattr(v, "__len__") else v) for k, v in out.items()})


if __name__ == "__main__":
    sys.exit(main())
