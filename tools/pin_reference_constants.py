#!/usr/bin/env python3
"""Read -- do not retype -- the parameter constants of the hot path out of the reference's Rust sources and write them to
tests/golden/ref_constants.json.  BUILD CONTAINER ONLY (it reads /root/reference; the GPU box has no such tree): the JSON
it writes is the committed fixture, data extracted mechanically from

  arkworks/curves/bls12_377/src/fields/fr.rs    FrParameters   (TWO_ADICITY, TWO_ADIC_ROOT_OF_UNITY, MODULUS, MODULUS_BITS,
                                                REPR_SHAVE_BITS, R, R2, INV, GENERATOR, MODULUS_MINUS_ONE_DIV_TWO, T, T_MINUS_ONE_DIV_TWO)
  arkworks/curves/bls12_377/src/fields/fq.rs    FqParameters   (same set)
  arkworks/curves/bls12_377/src/fields/fq2.rs   NONRESIDUE, FROBENIUS_COEFF_FP2_C1
  arkworks/curves/bls12_377/src/curves/g1.rs    COEFF_A/B (by name), COFACTOR, COFACTOR_INV, G1_GENERATOR_X/Y
  arkworks/curves/bls12_377/src/curves/g2.rs    COEFF_B, COFACTOR, COFACTOR_INV, G2_GENERATOR_{X,Y}_C{0,1}
  arkworks/curves/bls12_377/src/curves/mod.rs   X, X_IS_NEGATIVE, TWIST_TYPE
  arkworks/curves/mnt4_753/src/fields/fq.rs     the SHE ciphertext modulus' FqParameters
  arkworks/algebra/serialize/src/flags.rs       SWFlags::u8_bitmask bit positions, BIT_SIZE
  arkworks/algebra/poly/src/domain/radix2/mod.rs, ff get_root_of_unity: nothing numeric (derived from the above)

Every value carries the file and line it was read from.  Consumers: tests/test_ref_constants.py (CPU: oracle/zkref.py,
oracle/zkref_consts.h, zk-mpc_amd/csrc/gen_consts.py, consts.cuh; GPU: the library's constants through the C ABI).

  python tools/pin_reference_constants.py            # rewrite the fixture
  python tools/pin_reference_constants.py --check    # exit 1 if the committed fixture differs from the reference tree
"""
import json
import os
import re
import sys

REF = "/root/reference/arkworks"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "ref_constants.json")


def _int(tok: str) -> int:
    tok = tok.strip().replace("_", "")
    tok = re.sub(r"(u64|u32|usize|u8)$", "", tok)
    return int(tok, 16) if tok.lower().startswith("0x") else int(tok)


class Src:
    def __init__(self, rel):
        self.rel = rel
        self.text = open(os.path.join(REF, rel)).read()

    def line_of(self, pos):
        return self.text.count("\n", 0, pos) + 1

    def where(self, pos):
        return "arkworks/%s:%d" % (self.rel, self.line_of(pos))

    def bigint(self, name):
        """const NAME: BigInteger = BigInteger([ l0, l1, ... ]);  -> little-endian u64 limbs"""
        m = re.search(r"const %s:\s*(?:Option<)?BigInteger>?\s*=\s*(?:Some\()?BigInteger\(\[(.*?)\]\)" % name, self.text, re.S)
        assert m, (self.rel, name)
        body = re.sub(r"//[^\n]*", "", m.group(1))
        limbs = [_int(t) for t in body.split(",") if t.strip()]
        return {"limbs": ["0x%016x" % l for l in limbs], "value": str(sum(l << (64 * i) for i, l in enumerate(limbs))),
                "at": self.where(m.start())}

    def scalar(self, name, ty):
        m = re.search(r"const %s:\s*%s\s*=\s*([^;]+);" % (name, ty), self.text)
        assert m, (self.rel, name)
        return {"value": str(_int(m.group(1))), "at": self.where(m.start())}

    def field_new(self, name, field):
        """[pub] const NAME: F = field_new!(F, "decimal");"""
        m = re.search(r"const %s:\s*%s\s*=\s*field_new!\(\s*%s\s*,\s*\"(-?\d+)\"\s*\)" % (name, field, field), self.text)
        assert m, (self.rel, name)
        return {"value": m.group(1), "at": self.where(m.start())}

    def u64_slice(self, name):
        m = re.search(r"const %s:\s*&'static \[u64\]\s*=\s*&\[(.*?)\];" % name, self.text, re.S)
        assert m, (self.rel, name)
        limbs = [_int(t) for t in re.sub(r"//[^\n]*", "", m.group(1)).split(",") if t.strip()]
        return {"limbs": ["0x%016x" % l for l in limbs], "value": str(sum(l << (64 * i) for i, l in enumerate(limbs))),
                "at": self.where(m.start())}


def fp_params(rel):
    s = Src(rel)
    out = {k: s.bigint(k) for k in ("TWO_ADIC_ROOT_OF_UNITY", "MODULUS", "R", "R2", "GENERATOR", "MODULUS_MINUS_ONE_DIV_TWO", "T",
                                    "T_MINUS_ONE_DIV_TWO")}
    out["TWO_ADICITY"] = s.scalar("TWO_ADICITY", "u32")
    out["MODULUS_BITS"] = s.scalar("MODULUS_BITS", "u32")
    out["REPR_SHAVE_BITS"] = s.scalar("REPR_SHAVE_BITS", "u32")
    out["INV"] = s.scalar("INV", "u64")
    return out


def collect():
    c = {"_generated_by": "tools/pin_reference_constants.py (parsed from /root/reference, not retyped)",
         "_reference": "Yoii-Inc/zk-mpc, vendored arkworks tree"}
    c["bls12_377_fr"] = fp_params("curves/bls12_377/src/fields/fr.rs")
    c["bls12_377_fq"] = fp_params("curves/bls12_377/src/fields/fq.rs")
    c["mnt4_753_fq"] = fp_params("curves/mnt4_753/src/fields/fq.rs")
    s = Src("curves/bls12_377/src/fields/fq2.rs")
    m = re.search(r"const NONRESIDUE:\s*Fq\s*=\s*field_new!\(Fq,\s*\"(-?\d+)\"\)", s.text)
    fr1 = re.search(r"FROBENIUS_COEFF_FP2_C1.*?\[(.*?)\];", s.text, re.S)
    c["bls12_377_fq2"] = {"NONRESIDUE": {"value": m.group(1), "at": s.where(m.start())},
                          "FROBENIUS_COEFF_FP2_C1": {"value": [a or b for a, b in re.findall(r"(FQ_ONE)|field_new!\(Fq,\s*\"(-?\d+)\"\)", fr1.group(1))],
                                                     "at": s.where(fr1.start())}}
    g1 = Src("curves/bls12_377/src/curves/g1.rs")
    c["bls12_377_g1"] = {
        "COEFF_A": {"value": re.search(r"const COEFF_A:\s*Fq\s*=\s*(\w+);", g1.text).group(1), "at": g1.where(g1.text.index("const COEFF_A"))},
        "COEFF_B": {"value": re.search(r"const COEFF_B:\s*Fq\s*=\s*(\w+);", g1.text).group(1), "at": g1.where(g1.text.index("const COEFF_B"))},
        "COFACTOR": g1.u64_slice("COFACTOR"), "COFACTOR_INV": g1.field_new("COFACTOR_INV", "Fr"),
        "G1_GENERATOR_X": g1.field_new("G1_GENERATOR_X", "Fq"), "G1_GENERATOR_Y": g1.field_new("G1_GENERATOR_Y", "Fq")}
    g2 = Src("curves/bls12_377/src/curves/g2.rs")
    mb = re.search(r"const COEFF_B:\s*Fq2\s*=\s*field_new!\(Fq2,\s*(\w+),\s*field_new!\(Fq,\s*\"(\d+)\"\),?\s*\)", g2.text, re.S)
    c["bls12_377_g2"] = {
        "COEFF_B": {"value": [mb.group(1), mb.group(2)], "at": g2.where(mb.start())},
        "COFACTOR": g2.u64_slice("COFACTOR"), "COFACTOR_INV": g2.field_new("COFACTOR_INV", "Fr"),
        **{k: g2.field_new(k, "Fq") for k in ("G2_GENERATOR_X_C0", "G2_GENERATOR_X_C1", "G2_GENERATOR_Y_C0", "G2_GENERATOR_Y_C1")}}
    cm = Src("curves/bls12_377/src/curves/mod.rs")
    neg = re.search(r"const X_IS_NEGATIVE:\s*bool\s*=\s*(\w+);", cm.text)
    tw = re.search(r"const TWIST_TYPE:\s*TwistType\s*=\s*TwistType::(\w+);", cm.text)
    c["bls12_377"] = {"X": cm.u64_slice("X"), "X_IS_NEGATIVE": {"value": neg.group(1), "at": cm.where(neg.start())},
                      "TWIST_TYPE": {"value": tw.group(1), "at": cm.where(tw.start())}}
    fl = Src("algebra/serialize/src/flags.rs")
    sw = fl.text[fl.text.index("impl Flags for SWFlags"):]
    inf = re.search(r"SWFlags::Infinity\s*=>\s*mask\s*\|=\s*1\s*<<\s*(\d+)", sw)
    pos = re.search(r"SWFlags::PositiveY\s*=>\s*mask\s*\|=\s*1\s*<<\s*(\d+)", sw)
    bs = re.search(r"const BIT_SIZE:\s*usize\s*=\s*(\d+);", sw)
    off = fl.text.index("impl Flags for SWFlags")
    c["serialize_sw_flags"] = {"INFINITY_BIT": {"value": inf.group(1), "at": fl.where(off + inf.start())},
                               "POSITIVE_Y_BIT": {"value": pos.group(1), "at": fl.where(off + pos.start())},
                               "BIT_SIZE": {"value": bs.group(1), "at": fl.where(off + bs.start())}}
    # known answers the reference's OWN tests hold for long arithmetic chains (not parameters: expected outputs)
    ft = Src("curves/bls12_377/src/fields/tests.rs")
    m = re.search(r"fn test_fq_root_of_unity\(\).*?multiplicative_generator\(\)\s*\.pow\(\[(.*?)\]\)", ft.text, re.S)
    assert m
    limbs = [_int(t) for t in re.sub(r"//[^\n]*", "", m.group(1)).split(",") if t.strip()]
    ct = Src("curves/bls12_377/src/curves/tests.rs")
    g = re.search(r"fn test_g1_generator_raw\(\).*?assert_eq!\(i,\s*(\d+)\)", ct.text, re.S)
    assert g
    c["bls12_377_tests"] = {
        "FQ_ROOT_OF_UNITY_EXPONENT": {"limbs": ["0x%016x" % l for l in limbs], "value": str(sum(l << (64 * i) for i, l in enumerate(limbs))),
                                      "at": ft.where(m.start(1)), "claim": "Fq::multiplicative_generator().pow(this) == Fq::two_adic_root_of_unity()"},
        "G1_GENERATOR_RAW_X": {"value": g.group(1), "at": ct.where(g.start(1)),
                               "claim": "the point with this x and the smaller y (y < -y), scaled by the cofactor, is G1Affine::prime_subgroup_generator(); "
                                        "every smaller x gives a point that the cofactor sends to zero"}}
    return c


def main():
    c = collect()
    text = json.dumps(c, indent=1, sort_keys=True) + "\n"
    if "--check" in sys.argv:
        ok = os.path.exists(OUT) and open(OUT).read() == text
        print("tests/golden/ref_constants.json %s the reference tree" % ("matches" if ok else "DIFFERS from"))
        sys.exit(0 if ok else 1)
    open(OUT, "w").write(text)
    print("wrote %s (%d bytes)" % (OUT, len(text)))


if __name__ == "__main__":
    main()
