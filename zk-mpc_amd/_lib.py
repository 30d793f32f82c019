"""ctypes binding of libzkmpc_hip.so (the C ABI declared in include/zkmpc_hip.h).

There is no CPU fallback: if the shared library is missing or no HIP device is present the
import of the library / creation of a context raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libzkmpc_hip.so")

ZK_OK = 0
OP_MUL, OP_ADD, OP_SUB, OP_NEG = 0, 1, 2, 3


class ZkError(RuntimeError):
    pass


class Fr(C.Structure):
    _fields_ = [("l", C.c_uint64 * 4)]


class Fq(C.Structure):
    _fields_ = [("l", C.c_uint64 * 6)]


class MarlinMatrixEvals(C.Structure):
    _fields_ = [("row", C.c_void_p), ("col", C.c_void_p), ("val", C.c_void_p), ("row_col", C.c_void_p)]


class PolyRef(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("n", C.c_size_t)]


class MarlinIndex(C.Structure):
    _fields_ = [("num_constraints", C.c_size_t), ("num_variables", C.c_size_t), ("num_non_zero", C.c_size_t), ("num_instance", C.c_size_t),
                ("r1cs", C.c_void_p), ("r1cs_t", C.c_void_p), ("index_polys", PolyRef * 12),
                ("on_k", MarlinMatrixEvals * 3), ("on_b", MarlinMatrixEvals * 3),
                ("w_idx", C.c_void_p), ("x_idx", C.c_void_p), ("ivk_bytes", C.c_char_p), ("ivk_len", C.c_size_t)]


class Fq753(C.Structure):
    _fields_ = [("l", C.c_uint64 * 12)]


class G1Affine(C.Structure):
    _fields_ = [("x", Fq), ("y", Fq)]


class G1Projective(C.Structure):
    _fields_ = [("x", Fq), ("y", Fq), ("z", Fq)]


class G2Affine(C.Structure):
    _fields_ = [("x", Fq * 2), ("y", Fq * 2)]


class G2Projective(C.Structure):
    _fields_ = [("x", Fq * 2), ("y", Fq * 2), ("z", Fq * 2)]


class R1csHost(C.Structure):
    _fields_ = [("num_constraints", C.c_size_t), ("num_instance", C.c_size_t), ("num_witness", C.c_size_t),
                ("a_row_ptr", C.c_void_p), ("a_col", C.c_void_p), ("a_coeff", C.c_void_p),
                ("b_row_ptr", C.c_void_p), ("b_col", C.c_void_p), ("b_coeff", C.c_void_p),
                ("c_row_ptr", C.c_void_p), ("c_col", C.c_void_p), ("c_coeff", C.c_void_p)]


class PkHost(C.Structure):
    _fields_ = [("alpha_g1", G1Affine), ("beta_g1", G1Affine), ("delta_g1", G1Affine),
                ("beta_g2", G2Affine), ("delta_g2", G2Affine),
                ("a_query", C.c_void_p), ("a_len", C.c_size_t),
                ("b_g1_query", C.c_void_p), ("b_g1_len", C.c_size_t),
                ("b_g2_query", C.c_void_p), ("b_g2_len", C.c_size_t),
                ("h_query", C.c_void_p), ("h_len", C.c_size_t),
                ("l_query", C.c_void_p), ("l_len", C.c_size_t)]


_P = C.c_void_p
_SZ = C.c_size_t
_I = C.c_int
_U32 = C.c_uint32

# name -> (restype, argtypes).  Every symbol include/zkmpc_hip.h declares appears here.
PROTOTYPES = {
    "zk_ctx_create": (_I, [_I, _I, _I, C.POINTER(_P)]),
    "zk_ctx_destroy": (_I, [_P]),
    "zk_last_error": (C.c_char_p, [_P]),
    "zk_ctx_sync": (_I, [_P]),
    "zk_ctx_stream": (_P, [_P]),
    "zk_version": (_I, []),
    "zk_groth16_prove_shared": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "zk_groth16_prove_shared_spdz": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "zk_selftest_exception_barrier": (_I, [_I]),
    "zk_dev_alloc": (_I, [_P, _SZ, C.POINTER(_P)]),
    "zk_dev_free": (_I, [_P, _P]),
    "zk_memcpy_h2d": (_I, [_P, _P, _P, _SZ]),
    "zk_memcpy_d2h": (_I, [_P, _P, _P, _SZ]),
    "zk_fr_vec_op_dev": (_I, [_P, _I, _P, _P, _P, _SZ]),
    "zk_fr_vec_scale_dev": (_I, [_P, _P, _P, _P, _SZ]),
    "zk_fr_batch_product_in_place": (_I, [_P, _P, _P, _SZ]),
    "zk_fr_ntt_dev": (_I, [_P, _P, _U32, _I, _I]),
    "zk_fr_fft_in_place": (_I, [_P, _P, _SZ, _U32, _I, _I]),
    "zk_fr_divide_by_vanishing_on_coset_dev": (_I, [_P, _P, _U32]),
    "zk_msm_g1": (_I, [_P, _P, _SZ, _P, _SZ, _P]),
    "zk_msm_g2": (_I, [_P, _P, _SZ, _P, _SZ, _P]),
    "zk_msm_g1_strided": (_I, [_P, _P, _SZ, _P, _P, _SZ, _P]),
    "zk_msm_g2_strided": (_I, [_P, _P, _SZ, _P, _P, _SZ, _P]),
    "zk_bases_cache_config": (_I, [_P, _SZ, _I]),
    "zk_bases_cache_trust": (_I, [_P, _I]),
    "zk_msm_speculate": (_I, [_P, _I]),
    "zk_msm_speculate_stats": (_I, [_P, _P]),
    "zk_bases_cache_drop": (_I, [_P]),
    "zk_bases_cache_sync": (_I, [_P]),
    "zk_bases_cache_stats": (_I, [_P, _P]),
    "zk_bases_cache_stats2": (_I, [_P, _P]),
    "zk_mpc_fft_in_place": (_I, [_P, _P, _SZ, _P, _U32, _I, _I]),
    "zk_mpc_divide_by_vanishing_on_coset_in_place": (_I, [_P, _P, _P, _U32]),
    "zk_mpc_batch_product_in_place": (_I, [_P, _P, _P, _SZ, _P, _P, _P, _P]),
    "zk_mpc_msm_g1": (_I, [_P, _P, _SZ, _P, _P, _SZ, _P, _P, _P]),
    "zk_mpc_msm_g2": (_I, [_P, _P, _SZ, _P, _P, _SZ, _P, _P, _P]),
    "zk_fr_divide_by_vanishing_on_coset_in_place": (_I, [_P, _P, _U32]),
    "zk_bases_upload_g1": (_I, [_P, _P, _SZ, C.POINTER(_P)]),
    "zk_bases_upload_g2": (_I, [_P, _P, _SZ, C.POINTER(_P)]),
    "zk_bases_free": (_I, [_P, _P]),
    "zk_bases_precompute": (_I, [_P, _P]),
    "zk_bases_precompute_as": (_I, [_P, _P, _I]),
    "zk_bases_precompute_note": (C.c_char_p, [_P]),
    "zk_bases_len": (_SZ, [_P]),
    "zk_bases_window_bits": (_U32, [_P]),
    "zk_msm_g1_dev": (_I, [_P, _P, _SZ, _P, _SZ, _P]),
    "zk_msm_g2_dev": (_I, [_P, _P, _SZ, _P, _SZ, _P]),
    "zk_fixed_base_g1_dev": (_I, [_P, _P, _P, _SZ, C.POINTER(_P)]),
    "zk_fixed_base_g2_dev": (_I, [_P, _P, _P, _SZ, C.POINTER(_P)]),
    "zk_bases_download_g1": (_I, [_P, _P, _SZ, _SZ, _P]),
    "zk_bases_download_g2": (_I, [_P, _P, _SZ, _SZ, _P]),
    "zk_g1_add": (_I, [_P, _P, _P]),
    "zk_g2_add": (_I, [_P, _P, _P]),
    "zk_g1_neg": (_I, [_P, _P]),
    "zk_g2_neg": (_I, [_P, _P]),
    "zk_g1_mul": (_I, [_P, _P, _P]),
    "zk_g2_mul": (_I, [_P, _P, _P]),
    "zk_g1_from_affine": (_I, [_P, _P]),
    "zk_g2_from_affine": (_I, [_P, _P]),
    "zk_g1_serialize": (_I, [_P, _P]),
    "zk_g2_serialize": (_I, [_P, _P]),
    "zk_fr_add": (_I, [_P, _P, _P]),
    "zk_fr_sub": (_I, [_P, _P, _P]),
    "zk_fr_mul": (_I, [_P, _P, _P]),
    "zk_fq_add": (_I, [_P, _P, _P]),
    "zk_fq_sub": (_I, [_P, _P, _P]),
    "zk_fq_mul": (_I, [_P, _P, _P]),
    "zk_fq_mul2": (_I, [_P, _P, _P, _P, _P]),
    "zk_fq_neg5_almost_raw": (_I, [_P, _P]),
    "zk_fq_lazy_raw": (_I, [_I, _P, _P]),
    "zk_fr_lazy_raw": (_I, [_I, _P, _P]),
    "zk_fr_from_canonical": (_I, [_P, _P]),
    "zk_fr_to_canonical": (_I, [_P, _P]),
    "zk_r1cs_upload": (_I, [_P, _P, C.POINTER(_P)]),
    "zk_r1cs_free": (_I, [_P, _P]),
    "zk_r1cs_mul_chain": (_I, [_P, _SZ, C.POINTER(_P)]),
    "zk_mul_chain_assignment_dev": (_I, [_P, _SZ, _P, _P, _P]),
    "zk_pk_upload": (_I, [_P, _P, C.POINTER(_P)]),
    "zk_pk_free": (_I, [_P, _P]),
    "zk_groth16_setup": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, C.POINTER(_P)]),
    "zk_pk_query_len": (_SZ, [_P, _I]),
    "zk_pk_query_bases": (_P, [_P, _I]),
    "zk_pk_download_g1": (_I, [_P, _P, _I, _SZ, _SZ, _P]),
    "zk_pk_download_g2": (_I, [_P, _P, _I, _SZ, _SZ, _P]),
    "zk_pk_vk_g1": (_I, [_P, _I, _P]),
    "zk_pk_vk_g2": (_I, [_P, _I, _P]),
    "zk_groth16_witness_map_dev": (_I, [_P, _P, _P, _P]),
    "zk_r1cs_domain_log": (_U32, [_P]),
    "zk_groth16_witness_map_pre_dev": (_I, [_P, _P, _P, _I, _P, _P, _P]),
    "zk_groth16_witness_map_post_dev": (_I, [_P, _P, _P, _P]),
    "zk_groth16_msms_dev": (_I, [_P, _P, _P, _P, _P, _P, _P]),
    "zk_groth16_msms_presort_dev": (_I, [_P, _P, _P, _P]),
    "zk_groth16_msms_begin_dev": (_I, [_P, _P, _P, _P]),
    "zk_groth16_hint_next_dev": (_I, [_P, _P]),
    "zk_groth16_chain_fronts": (_I, [_P, _I]),
    "zk_groth16_prove_dev": (_I, [_P, _P, _P, _P, _P, _P, _P]),
    "zk_groth16_prove_multi": (_I, [_P, _P, _P, _I, _P, _P, _P, _P]),
    "zk_groth16_multi_plan": (_I, [_P, _P, _I, C.c_char_p, _SZ, C.POINTER(_SZ)]),
    "zk_groth16_prove": (_I, [_P, _P, _P, _P, _P, _P, _P]),
    "zk_groth16_prove_queued": (_I, [_P, _P, _P, _P, _P, _P, _P, _P]),
    "zk_host_alloc": (_I, [_P, _SZ, C.POINTER(_P)]),
    "zk_host_free": (_I, [_P, _P]),
    "zk_fr_powers_dev": (_I, [_P, _P, _P, _SZ, _P]),
    "zk_fr_batch_inverse_dev": (_I, [_P, _P, _SZ]),
    "zk_poly_evaluate_dev": (_I, [_P, _P, _SZ, _P, _P]),
    "zk_poly_evaluate_batch_dev": (_I, [_P, _P, _P, _SZ, _P]),
    "zk_poly_divide_by_linear_dev": (_I, [_P, _P, _SZ, _P, _P, _P]),
    "zk_poly_divide_by_vanishing_dev": (_I, [_P, _P, _SZ, _U32, _P, _P]),
    "zk_poly_mul_dev": (_I, [_P, _P, _SZ, _P, _SZ, _P]),
    "zk_kzg_commit_dev": (_I, [_P, _P, _P, _SZ, _P, _P, _SZ, _P]),
    "zk_kzg_open_dev": (_I, [_P, _P, _P, _SZ, _P, _P, _P, _SZ, _P, _P]),
    "zk_point_serialized_size": (_SZ, [_I, _I]),
    "zk_bases_serialize": (_I, [_P, _P, _SZ, _SZ, _I, _P]),
    "zk_bases_deserialize_uncompressed": (_I, [_P, _I, _P, _SZ, C.POINTER(_P)]),
    "zk_bases_deserialize_compressed": (_I, [_P, _I, _P, _SZ, C.POINTER(_P)]),
    "zk_msm_batch_dev": (_I, [_P, _SZ, _P, _P, _P, _P, _P]),
    "zk_vk_serialized_size": (_SZ, [_P, _I]),
    "zk_pk_serialized_size": (_SZ, [_P, _I]),
    "zk_vk_serialize": (_I, [_P, _P, _I, _P, _SZ]),
    "zk_pk_serialize": (_I, [_P, _P, _I, _P, _SZ]),
    "zk_pk_deserialize": (_I, [_P, _P, _SZ, _I, C.POINTER(_P)]),
    "zk_kzg_srs_serialized_size": (_SZ, [_SZ, _SZ, _I]),
    "zk_kzg_srs_serialize": (_I, [_P, _P, _P, _P, _P, _I, _P, _SZ]),
    "zk_kzg_srs_deserialize": (_I, [_P, _P, _SZ, _I, C.POINTER(_P), C.POINTER(_P), _P, _P]),
    "zk_comm_unique_id": (_I, [_P]),
    "zk_comm_init": (_I, [_P, _P, _I, _I]),
    "zk_comm_destroy": (_I, [_P]),
    "zk_comm_info": (_I, [_P, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.c_char_p, _SZ]),
    "zk_comm_set_open_pattern": (_I, [_P, _I]),
    "zk_open_sum_fr_dev": (_I, [_P, _P, _SZ, _P]),
    "zk_memcpy_d2d": (_I, [_P, _P, _P, _SZ]),
    "zk_dev_zero": (_I, [_P, _P, _SZ]),
    "zk_fr_inverse": (_I, [_P, _P]),
    "zk_fr_pow": (_I, [_P, C.c_uint64, _P]),
    "zk_r1cs_matvec_dev": (_I, [_P, _P, _I, _P, _P, _SZ]),
    "zk_fr_gather_dev": (_I, [_P, _P, _P, _SZ, _P]),
    "zk_marlin_round3_f_evals_dev": (_I, [_P, _P, _SZ, _P, _P, _P, _P, _P]),
    "zk_marlin_round3_ab_evals_dev": (_I, [_P, _P, _SZ, _P, _P, _P, _P, _P, _P]),
    "zk_marlin_proof_max_size": (_SZ, []),
    "zk_marlin_prove": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _SZ, _P]),
    "zk_marlin_prove_shared": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _SZ, _P, _P]),
    "zk_marlin_prove_shared_spdz": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _SZ, _P, _P]),
    "zk_she_vec_op_dev": (_I, [_P, _I, _P, _P, _P, _SZ]),
    "zk_she_vec_scale_dev": (_I, [_P, _P, _P, _P, _SZ]),
    "zk_she_negacyclic_mul_dev": (_I, [_P, _P, _P, _P, _SZ, _SZ]),
    "zk_she_ciphertext_mul_dev": (_I, [_P, _P, _P, _P, _SZ, _SZ]),
    "zk_she_encrypt_dev": (_I, [_P, _P, _P, _P, _P, _P, _P, _SZ, _SZ]),
    "zk_she_decrypt_dev": (_I, [_P, _P, _P, _P, _SZ, _SZ]),
    "zk_she_encode_dev": (_I, [_P, _P, _P, _SZ, _SZ]),
    "zk_she_decode_dev": (_I, [_P, _P, _P, _SZ, _SZ]),
    "zk_fr_sum_parties_dev": (_I, [_P, _P, _SZ, _SZ, _P]),
    "zk_beaver_combine_dev": (_I, [_P, _P, _P, _P, _P, _P, _P, _SZ]),
    "zk_fr_vec_is_zero_dev": (_I, [_P, _P, _SZ, _P]),
    "zk_fr_random_dev": (_I, [_P, _P, C.c_uint64, _P, _SZ]),
    "zk_fsrng_new": (_I, [_P, _SZ, C.POINTER(_P)]),
    "zk_fsrng_absorb": (_I, [_P, _P, _SZ]),
    "zk_rng_from_seed": (_I, [_P, _I, C.POINTER(_P)]),
    "zk_rng_free": (_I, [_P]),
    "zk_rng_next_u64": (_I, [_P, _P]),
    "zk_rng_next_u128": (_I, [_P, _P]),
    "zk_rng_next_fr": (_I, [_P, _P]),
    "zk_rng_fill_fr": (_I, [_P, _P, _SZ]),
    "zk_rng_fill_bytes": (_I, [_P, _P, _SZ]),
    "zk_blake2s": (_I, [_P, _SZ, _P]),
    "zk_chacha_block": (_I, [_P, _P, _I, _P]),
    "zk_set_profiling": (_I, [_P, _I]),
    "zk_last_timers": (_I, [_P, _P, _SZ, _P, _P, _I]),
    "zk_diag_int_mad_peak": (_I, [_P, _I, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "zk_diag_fq_pow_dev": (_I, [_P, _P, _P, _I, _P]),
    "zk_diag_fr_pow_dev": (_I, [_P, _P, _P, _I, _P]),
    "zk_diag_g1_mul_glv": (_I, [_P, _P, _P]),
}

_lib = None


def load():
    """Load libzkmpc_hip.so (once) and attach prototypes.  Raises ZkError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ZkError("libzkmpc_hip.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(expected at %s)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
