"""GPU: the committed fixtures under tests/golden/ (oracle outputs frozen in round 1 by oracle/gen_golden.py) against the
library through the C ABI -- field products, the four transforms, MSMs with host-slice bases, point bytes, proving-key
elements, witness map and the 192 proof bytes.  Frozen files catch a drift of oracle AND product together (a change that moved
both would still move away from the files)."""
import numpy as np
import pytest

import zk_mpc_amd.convert as cv
from helpers import csr, g1_from_json, g2_from_json, golden, ih, mont1, r1cs_from_json, td_mont, trapdoor_from_json

pytestmark = pytest.mark.gpu


def test_golden_field_and_transforms(ctx):
    g = golden("field_ntt.json")
    a, b = [ih(x) for x in g["fr_a"]], [ih(x) for x in g["fr_b"]]
    am, bm = cv.fr_to_mont(a), cv.fr_to_mont(b)
    assert cv._limbs_to_ints(am) == [ih(x) for x in g["fr_mont_a"]]
    prod = am.copy()
    ctx.batch_product_in_place(prod, bm)
    assert cv.fr_from_mont(prod) == [ih(x) for x in g["fr_mul"]]
    for key, fn in (("fft8", ctx.fft_in_place), ("ifft8", ctx.ifft_in_place), ("coset_fft8", ctx.coset_fft_in_place),
                    ("coset_ifft8", ctx.coset_ifft_in_place)):
        assert cv.fr_from_mont(fn(am.copy(), 3)) == [ih(x) for x in g[key]]


def test_golden_msm_and_point_bytes(ctx):
    g = golden("msm.json")
    g1b = [g1_from_json(p) for p in g["g1_bases"]]
    g2b = [g2_from_json(p) for p in g["g2_bases"]]
    sc = [ih(x) for x in g["scalars"]]
    ks = ctx.upload(cv.fr_to_mont([ih(k) for k in g["base_scalars"]]))
    assert cv.g1_array_to_affine(ctx.fixed_base(ks.ptr, len(g1b), 1, mont1(1)).download()) == g1b
    assert cv.g1_projective_to_affine(ctx.multi_scalar_mul_g1(cv.g1_affine_to_array(g1b), cv.fr_to_mont(sc))) == g1_from_json(g["msm_g1"])
    assert cv.g2_projective_to_affine(ctx.multi_scalar_mul_g2(cv.g2_affine_to_array(g2b), cv.fr_to_mont(sc[:4]))) == g2_from_json(g["msm_g2"])
    b1 = ctx.bases_upload(cv.g1_affine_to_array(g1b + [None]), 1)
    b2 = ctx.bases_upload(cv.g2_affine_to_array(g2b + [None]), 2)
    assert [b1.serialize(True, offset=i, n=1).hex() for i in range(len(g1b) + 1)] == g["g1_compressed"]
    assert [b2.serialize(True, offset=i, n=1).hex() for i in range(len(g2b) + 1)] == g["g2_compressed"]


@pytest.mark.parametrize("name", ["my_simple_circuit", "mul_chain_5"])
def test_golden_groth16(ctx, name):
    j = golden("groth16.json")[name]
    r1cs = r1cs_from_json(j)
    z = [ih(v) for v in j["z"]]
    td = trapdoor_from_json(j["trapdoor"])
    dr = ctx.r1cs_upload(r1cs.num_instance, r1cs.num_witness, csr(r1cs.a), csr(r1cs.b), csr(r1cs.c))
    tdm = td_mont(td)
    pk = ctx.groth16_setup(dr, *[tdm[i] for i in range(7)])
    assert cv.g1_array_to_affine(pk.download("a_query")) == [g1_from_json(p) for p in j["pk"]["a_query"]]
    assert cv.g2_array_to_affine(pk.download("b_g2_query")) == [g2_from_json(p) for p in j["pk"]["b_g2_query"]]
    assert cv.g1_array_to_affine(pk.download("h_query")) == [g1_from_json(p) for p in j["pk"]["h_query"]]
    zd = ctx.upload(cv.fr_to_mont(z))
    D = 1 << dr.domain_log
    h = ctx.alloc(D * 32)
    ctx.witness_map_dev(dr, zd.ptr, h.ptr)
    assert cv.fr_from_mont(ctx.download(h, (D, 4)))[:len(j["h"])] == [ih(v) for v in j["h"]]
    proof = ctx.create_proof(pk, dr, cv.fr_to_mont(z), mont1(ih(j["r"])), mont1(ih(j["s"])))
    assert proof.hex() == j["proof"] and len(proof) == 192
    pk.free()
