#!/usr/bin/env python3
"""Generate the Rust side of the C ABI from include/zkmpc_hip.h -- so that it cannot drift from the header.

  python tools/gen_rust_ffi.py            # writes bindings/hip_ffi.rs and bindings/overrides.rs
  python tools/gen_rust_ffi.py --check    # exit 1 if the committed files differ from what the header generates

bindings/hip_ffi.rs   every typedef (#[repr(C)] structs, opaque handles, the transport vtable), every `extern "C"` prototype and
                      the error / opcode constants of the header, mechanically.
bindings/overrides.rs the trait overrides of INTEGRATION.md section 2 as Rust source against those declarations: the four dispatch
                      points of SURVEY.md 8(b) -- AffineCurve::multi_scalar_mul (arkworks/algebra/ec/src/lib.rs:305-318),
                      EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}_in_place + divide_by_vanishing_poly_on_coset_in_place
                      (poly/src/domain/mod.rs:78-190), Field::batch_product_in_place (ff/src/fields/mod.rs:216-220) and
                      MpcNet::broadcast_bytes (mpc-net/src/lib.rs:60-64) as the library's zk_net_vtable.
There is no Rust toolchain in the build image: neither file is compiled here.  tests/test_abi.py keeps them honest instead --
symbol for symbol and arity for arity against the header and against zk-mpc_amd/_lib.py::PROTOTYPES, and every zk_* call in
overrides.rs against the prototype it names.
"""
from __future__ import annotations

import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "zkmpc_hip.h")
OUT_DIR = os.path.join(ROOT, "bindings")

SCALARS = {"int": "i32", "unsigned": "u32", "uint32_t": "u32", "uint64_t": "u64", "uint8_t": "u8", "uint16_t": "u16", "size_t": "usize",
           "double": "f64", "float": "f32", "char": "c_char", "void": "c_void", "int32_t": "i32", "int64_t": "i64"}


def camel(name: str) -> str:
    return "".join(p.capitalize() for p in name.split("_"))


def strip(text: str) -> str:
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    text = re.sub(r'extern\s+"C"\s*\{', "", text)
    return text


def split_top(s: str, sep: str):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == sep and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


class Header:
    def __init__(self, path=HEADER):
        raw = open(path).read()
        self.consts = [(m.group(1), m.group(2)) for m in re.finditer(r"^#define\s+(ZK_[A-Z0-9_]+)\s+(-?\d+)\b", raw, re.M)]
        text = strip(raw)
        self.opaque, self.structs, self.funcs = [], [], []        # names; (name, [(field, ctype, array)] | fnptr); (name, ret, [(ctype, pname)])
        for decl in self._decls(text):
            self._decl(decl)

    @staticmethod
    def _decls(text):
        out, depth, cur = [], 0, ""
        for ch in text:
            if ch == "{":
                depth += 1
            elif ch == "}":
                depth -= 1
                if depth < 0:          # the closing brace of extern "C" {
                    depth = 0
                    continue
            if ch == ";" and depth == 0:
                if cur.strip():
                    out.append(" ".join(cur.split()))
                cur = ""
            else:
                cur += ch
        return out

    def _decl(self, d: str):
        m = re.match(r"typedef struct (\w+) (\w+)$", d)
        if m:
            self.opaque.append(m.group(2))
            return
        m = re.match(r"typedef struct (\w+ )?\{(.*)\} (\w+)$", d)
        if m:
            fields = []
            for f in split_top(m.group(2), ";"):
                f = f.strip()
                if not f:
                    continue
                fp = re.match(r"(.+?)\(\*(\w+)\)\((.*)\)$", f)
                if fp:
                    fields.append((fp.group(2), ("fnptr", fp.group(1).strip(), [self._param(p) for p in split_top(fp.group(3), ",")]), None))
                    continue
                base, decls = self._split_base(f)
                for dcl in decls:
                    am = re.match(r"(\**)\s*(\w+)(?:\[(\d+)\])?$", dcl.strip())
                    fields.append((am.group(2), base + am.group(1), int(am.group(3)) if am.group(3) else None))
            self.structs.append((m.group(3), fields))
            return
        m = re.match(r"(.+?)\b(zk_\w+)\s*\((.*)\)$", d)
        if m:
            params = [] if m.group(3).strip() in ("", "void") else [self._param(p) for p in split_top(m.group(3), ",")]
            self.funcs.append((m.group(2), m.group(1).strip(), params))
            return
        raise SystemExit("gen_rust_ffi: cannot parse declaration: %r" % d)

    @staticmethod
    def _split_base(f: str):
        # "const void* row" / "zk_fq x[2], y[2]" / "uint64_t l[4]" -> (base type, [declarators])
        first, *rest = split_top(f, ",")
        m = re.match(r"(.*?)(\**\s*\w+(?:\[\d+\])?)$", first.strip())
        base = m.group(1).strip()
        return base, [m.group(2)] + rest

    @staticmethod
    def _param(p: str):
        p = p.strip()
        m = re.match(r"(.*?)(\w+)\[(\d*)\]$", p)          # array parameter: a pointer to the element type
        if m:
            return (m.group(1).strip() + "*", m.group(2), m.group(3))
        m = re.match(r"(.*?)(\w+)$", p)
        return (m.group(1).strip(), m.group(2), None)


def rust_type(ctype: str, names) -> str:
    """C type (pointers included) -> Rust.  `const T*` -> *const T, `T*` -> *mut T, right to left."""
    t = ctype.replace("*", " * ").split()
    # leading qualifiers / base
    const_base = False
    i = 0
    if t[i] == "const":
        const_base = True
        i += 1
    if t[i] == "struct":
        i += 1
    if t[i] == "unsigned" and i + 1 < len(t) and t[i + 1] == "int":
        i += 1
    base = t[i]
    i += 1
    if i < len(t) and t[i] == "const":      # "T const"
        const_base = True
        i += 1
    cur = SCALARS.get(base) or (camel(base) if base in names else None)
    if cur is None:
        raise SystemExit("gen_rust_ffi: unknown C type %r in %r" % (base, ctype))
    is_const = const_base
    while i < len(t):
        if t[i] == "*":
            cur = ("*const " if is_const else "*mut ") + cur
            is_const = False
        elif t[i] == "const":
            is_const = True
        else:
            raise SystemExit("gen_rust_ffi: cannot read %r" % ctype)
        i += 1
    return cur


def generate():
    H = Header()
    names = set(H.opaque) | {n for n, _ in H.structs}
    L = []
    L.append("// hip_ffi.rs -- GENERATED by tools/gen_rust_ffi.py from include/zkmpc_hip.h.  Do not edit: change the header and regenerate.")
    L.append("// The C ABI of libzkmpc_hip.so for a Rust host (mpc-algebra/src/hip_ffi.rs in the reference tree): every typedef, every")
    L.append("// prototype, the error and opcode constants.  What each entry point replaces is documented in the header.")
    L.append("#![allow(non_camel_case_types, dead_code, clippy::too_many_arguments)]")
    L.append("use std::os::raw::{c_char, c_void};")
    L.append("")
    for k, v in H.consts:
        L.append("pub const %s: i32 = %s;" % (k, v))
    L.append("")
    for n in H.opaque:
        L.append("#[repr(C)] pub struct %s { _private: [u8; 0] }          // opaque: %s" % (camel(n), n))
    L.append("")
    for n, fields in H.structs:
        L.append("#[repr(C)]")
        L.append("#[derive(Clone, Copy)]")
        L.append("pub struct %s {          // %s" % (camel(n), n))
        for fname, ctype, arr in fields:
            if isinstance(ctype, tuple):
                _, ret, params = ctype
                ps = ", ".join("%s: %s" % (pn, rust_type(ct, names)) for ct, pn, _ in params)
                r = "" if ret == "void" else " -> " + rust_type(ret, names)
                L.append("    pub %s: Option<unsafe extern \"C\" fn(%s)%s>," % (fname, ps, r))
            else:
                rt = rust_type(ctype, names)
                L.append("    pub %s: %s," % (fname, "[%s; %d]" % (rt, arr) if arr else rt))
        L.append("}")
        L.append("")
    L.append('#[link(name = "zkmpc_hip")]')
    L.append('extern "C" {')
    for name, ret, params in H.funcs:
        ps = []
        for ct, pn, arr in params:
            rt = rust_type(ct, names)
            ps.append("%s: %s%s" % (pn, rt, "" if not arr else " /* [%s] */" % arr))
        r = "" if ret == "void" else " -> " + rust_type(ret, names)
        L.append("    pub fn %s(%s)%s;" % (name, ", ".join(ps), r))
    L.append("}")
    L.append("")
    return "\n".join(L), H


OVERRIDES = r'''// overrides.rs -- GENERATED by tools/gen_rust_ffi.py (the text below is a template in that script; the zk_* calls are checked
// against include/zkmpc_hip.h by tests/test_abi.py).  The trait overrides of INTEGRATION.md section 2 for Bls12_377: what a
// maintainer adds to the reference tree so that every caller in src/ and arkworks/ reaches the GPU unchanged.
// Not compiled in the build image (no Rust toolchain).
use crate::hip_ffi::*;
use ark_bls12_377::{Fr, G1Affine, G1Projective, G2Affine, G2Projective};
use ark_ff::Zero;
use std::os::raw::c_void;

thread_local! {
    // one context per party task / thread (SURVEY.md 8b: no process-global device state); device chosen by the host
    pub static CTX: *mut ZkCtx = unsafe {
        let mut c: *mut ZkCtx = std::ptr::null_mut();
        let rc = zk_ctx_create(device_for_party(), mpc_net::MpcMultiNet::party_id() as i32, mpc_net::MpcMultiNet::n_parties() as i32, &mut c);
        assert_eq!(rc, ZK_OK, "zk_ctx_create failed");
        c
    };
}

fn check(rc: i32) {
    // arkworks panics on failure and MpcSerNet::broadcast unwraps (mpc-algebra/src/channel.rs:20-23): so does the override
    if rc != ZK_OK {
        let msg = CTX.with(|c| unsafe { std::ffi::CStr::from_ptr(zk_last_error(*c)).to_string_lossy().into_owned() });
        panic!("libzkmpc_hip error {}: {}", rc, msg);
    }
}

// GroupAffine<P> is {x, y, infinity: bool} as rustc lays it out (104 / 200 bytes per point for BLS12-377): the library reads the
// caller's slice IN PLACE -- stride and field offsets are taken off a value, whatever order rustc chose -- so nothing is copied per
// call; the library keeps a resident copy keyed by the table's CONTENT (zk_bases_cache_*): a proving key's queries are uploaded once,
// and every later hit is confirmed against the caller's slice under the MSM (or taken on trust: trust_base_tables).
fn layout_of<T, X, Y>(sample: &T, x: &X, y: &Y, infinity: &bool) -> ZkAffineLayout {
    let base = sample as *const T as usize;
    ZkAffineLayout { stride: std::mem::size_of::<T>(), off_x: x as *const X as usize - base, off_y: y as *const Y as usize - base,
                     off_infinity: infinity as *const bool as usize - base }
}

// ---- 1. AffineCurve::multi_scalar_mul (arkworks/algebra/ec/src/lib.rs:305-318); min(len) rule inside (msm/variable_base.rs:15-17) ----
pub fn multi_scalar_mul_g1(bases: &[G1Affine], scalars: &[Fr]) -> G1Projective {
    let mut out = ZkG1Projective { x: ZkFq { l: [0; 6] }, y: ZkFq { l: [0; 6] }, z: ZkFq { l: [0; 6] } };
    if let Some(p) = bases.first() {
        let lay = layout_of(p, &p.x, &p.y, &p.infinity);
        CTX.with(|c| check(unsafe { zk_msm_g1_strided(*c, bases.as_ptr() as *const c_void, bases.len(), &lay, scalars.as_ptr() as *const ZkFr, scalars.len(), &mut out) }));
    } else {
        CTX.with(|c| check(unsafe { zk_msm_g1(*c, std::ptr::null(), 0, scalars.as_ptr() as *const ZkFr, scalars.len(), &mut out) }));
    }
    G1Projective::new(ark_ff::Fp384::new(ark_ff::BigInteger384(out.x.l)), ark_ff::Fp384::new(ark_ff::BigInteger384(out.y.l)),
                      ark_ff::Fp384::new(ark_ff::BigInteger384(out.z.l)))
}
pub fn multi_scalar_mul_g2(bases: &[G2Affine], scalars: &[Fr]) -> G2Projective {
    let z6 = ZkFq { l: [0; 6] };
    let mut out = ZkG2Projective { x: [z6; 2], y: [z6; 2], z: [z6; 2] };
    if let Some(p) = bases.first() {
        let lay = layout_of(p, &p.x, &p.y, &p.infinity);          // Fq2 is {c0, c1}: c0 then c1 at off_x / off_y (quadratic_extension.rs)
        CTX.with(|c| check(unsafe { zk_msm_g2_strided(*c, bases.as_ptr() as *const c_void, bases.len(), &lay, scalars.as_ptr() as *const ZkFr, scalars.len(), &mut out) }));
    } else {
        CTX.with(|c| check(unsafe { zk_msm_g2(*c, std::ptr::null(), 0, scalars.as_ptr() as *const ZkFr, scalars.len(), &mut out) }));
    }
    let f = |a: &[ZkFq; 2]| ark_bls12_377::Fq2::new(ark_ff::Fp384::new(ark_ff::BigInteger384(a[0].l)), ark_ff::Fp384::new(ark_ff::BigInteger384(a[1].l)));
    G2Projective::new(f(&out.x), f(&out.y), f(&out.z))
}
// (a host that rewrites a base table in place needs no call: a verified hit sees it.  In trusted mode it must say so:)
pub fn bases_changed_in_place() { CTX.with(|c| check(unsafe { zk_bases_cache_drop(*c) })); }

// ---- 2. EvaluationDomain::{fft, ifft, coset_fft, coset_ifft}_in_place (poly/src/domain/mod.rs:78,89,138,154; radix2/mod.rs:98-114) ----
pub fn fft_family_in_place(coeffs: &mut Vec<Fr>, size: usize, log_size: u32, inverse: bool, coset: bool) {
    coeffs.resize(size, Fr::zero());
    CTX.with(|c| check(unsafe { zk_fr_fft_in_place(*c, coeffs.as_mut_ptr() as *mut ZkFr, coeffs.len(), log_size, inverse as i32, coset as i32) }));
}
// EvaluationDomain::divide_by_vanishing_poly_on_coset_in_place (poly/src/domain/mod.rs:183-190): the host slice, or a device-resident vector
pub fn divide_by_vanishing_poly_on_coset_in_place(evals: &mut [Fr], log_size: u32) {
    assert_eq!(evals.len(), 1usize << log_size);
    CTX.with(|c| check(unsafe { zk_fr_divide_by_vanishing_on_coset_in_place(*c, evals.as_mut_ptr() as *mut ZkFr, log_size) }));
}
pub fn divide_by_vanishing_poly_on_coset_dev(evals_dev: *mut c_void, log_size: u32) {
    CTX.with(|c| check(unsafe { zk_fr_divide_by_vanishing_on_coset_dev(*c, evals_dev, log_size) }));
}

// ---- 3. Field::batch_product_in_place (arkworks/algebra/ff/src/fields/mod.rs:216-220) ----
pub fn batch_product_in_place(selfs: &mut [Fr], others: &[Fr]) {
    let n = selfs.len().min(others.len());
    CTX.with(|c| check(unsafe { zk_fr_batch_product_in_place(*c, selfs.as_mut_ptr() as *mut ZkFr, others.as_ptr() as *const ZkFr, n) }));
}

// ---- 4. MpcNet::broadcast_bytes (mpc-net/src/lib.rs:60-64) under MpcSerNet::broadcast (mpc-algebra/src/channel.rs:12-28) ----
// The collaborative provers reach the host's transport through this vtable; the D-element opens go over RCCL inside the library.
unsafe extern "C" fn all_gather_bytes(_user: *mut c_void, mine: *const u8, len: usize, out_all: *mut u8) -> i32 {
    let mine = std::slice::from_raw_parts(mine, len);
    match mpc_net::MpcMultiNet::broadcast_bytes(&bytes::Bytes::copy_from_slice(mine)) {        // ordered by party id (multi.rs:469-525)
        Ok(parts) => { for (p, b) in parts.iter().enumerate() { std::ptr::copy_nonoverlapping(b.as_ptr(), out_all.add(p * len), len); } 0 }
        Err(_) => -1,
    }
}
pub fn net_vtable() -> ZkNetVtable {
    ZkNetVtable { user: std::ptr::null_mut(), all_gather_bytes: Some(all_gather_bytes), open_sum_fr_dev: None /* = the context's RCCL communicator */ }
}
// once per session: the leader's 128-byte id travels over the mesh the parties already have, then the opens run GPU to GPU
pub fn comm_init() {
    let mut id = [0u8; 128];
    if mpc_net::MpcMultiNet::am_king() { check(unsafe { zk_comm_unique_id(id.as_mut_ptr()) }); }
    let id = mpc_net::MpcMultiNet::broadcast_bytes(&bytes::Bytes::copy_from_slice(&id)).unwrap()[0].clone();
    CTX.with(|c| check(unsafe { zk_comm_init(*c, id.as_ptr(), mpc_net::MpcMultiNet::party_id() as i32, mpc_net::MpcMultiNet::n_parties() as i32) }));
}
// AdditiveFieldShare::batch_open of a device vector (mpc-algebra/src/share/additive.rs:124-131)
pub fn open_sum_fr_dev(v_dev: *const c_void, n: usize, out_dev: *mut c_void) {
    CTX.with(|c| check(unsafe { zk_open_sum_fr_dev(*c, v_dev, n, out_dev) }));
}

// ---- 5. the same dispatch points for the COLLABORATIVE element types: create_proof::<MpcPairingEngine, C> unchanged ----------------
// Under E = MpcPairingEngine the callers hand over Vec<MpcField<Fr, S>> and &[MpcG1Affine] (enum wrappers: wire/field.rs:37-40,
// wire/pairing.rs).  The library reads and writes those elements IN PLACE; the layouts are taken off values once.  Where the hooks
// sit: (a) Field::batch_product_in_place is overridden by MpcField already (wire/field.rs:917-958) -- its body becomes
// mpc_batch_product_in_place; (b) MpcG1Affine / MpcG2Affine::multi_scalar_mul (wire/pairing.rs:714-777) -- its body becomes
// mpc_msm_g1 / _g2 on the key's OWN slice (no all_public_or_shared copy: wire/group.rs:441-457); (c) the transforms: FftField gains
// `fn fft_in_place_hook(coeffs: &mut Vec<Self>, log_size: u32, inverse: bool, coset: bool) -> bool { false }` (the same kind of patch
// as batch_product_in_place), Radix2EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}_in_place and
// divide_by_vanishing_poly_on_coset_in_place call it first, Fr answers with fft_family_in_place, MpcField<Fr, S> with mpc_fft_in_place.
use mpc_algebra::{AdditiveFieldShare, FieldShare, MpcField, Reveal, SpdzFieldShare};

/// what the share types tell the binding about themselves (implemented next to the private fields: share/additive.rs, share/spdz.rs)
pub trait HipLanes { fn lanes(&self) -> (*const Fr, Option<*const Fr>); }    // (&self.val, None) / (&self.sh.val, Some(&self.mac.val))

pub fn mpc_field_layout<S: FieldShare<Fr> + HipLanes>() -> ZkMpcFieldLayout {
    let p = MpcField::<Fr, S>::Public(Fr::zero());
    let s = MpcField::<Fr, S>::Shared(S::from_add_shared(Fr::zero()));
    let base_p = &p as *const _ as usize;
    let base_s = &s as *const _ as usize;
    let off_public = match &p { MpcField::Public(x) => x as *const Fr as usize - base_p, _ => unreachable!() };
    let (off_share, off_mac) = match &s {
        MpcField::Shared(sh) => { let (a, b) = sh.lanes(); (a as usize - base_s, b.map(|m| m as usize - base_s).unwrap_or(usize::MAX)) }
        _ => unreachable!(),
    };
    // the discriminant: the one byte outside every payload in which the two values differ (all payload words are zero in both)
    let size = std::mem::size_of::<MpcField<Fr, S>>();
    let in_payload = |o: usize| (o >= off_public && o < off_public + 32) || (o >= off_share && o < off_share + 32) || (off_mac != usize::MAX && o >= off_mac && o < off_mac + 32);
    let (bp, bs) = unsafe { (std::slice::from_raw_parts(base_p as *const u8, size), std::slice::from_raw_parts(base_s as *const u8, size)) };
    let off_tag = (0..size).find(|&o| !in_payload(o) && bp[o] != bs[o]).expect("MpcField: no discriminant byte found");
    ZkMpcFieldLayout { stride: size, off_tag, off_public, off_share, off_mac, tag_public: bp[off_tag], tag_shared: bs[off_tag] }
}
/// MpcG1Affine / MpcG2Affine { val: MpcGroup<G, _> }: the GroupAffine of the Public variant and the byte that says Public
pub fn mpc_group_layout<W, G, X, Y>(public_sample: &W, inner: &G, x: &X, y: &Y, infinity: &bool, shared_sample: &W) -> ZkMpcGroupLayout {
    let base = public_sample as *const W as usize;
    let size = std::mem::size_of::<W>();
    let point = ZkAffineLayout { stride: size, off_x: x as *const X as usize - base, off_y: y as *const Y as usize - base, off_infinity: infinity as *const bool as usize - base };
    let lo = inner as *const G as usize - base;
    let hi = lo + std::mem::size_of::<G>();
    let (bp, bs) = unsafe { (std::slice::from_raw_parts(base as *const u8, size), std::slice::from_raw_parts(shared_sample as *const W as *const u8, size)) };
    let off_tag = (0..size).find(|&o| (o < lo || o >= hi) && bp[o] != bs[o]).expect("MpcGroup: no discriminant byte found");
    ZkMpcGroupLayout { point, off_tag, tag_public: bp[off_tag] }
}

// EvaluationDomain::*fft_in_place(&mut Vec<MpcField<Fr, S>>) (src/groth16.rs:278-303)
pub fn mpc_fft_in_place<S: FieldShare<Fr> + HipLanes>(coeffs: &mut Vec<MpcField<Fr, S>>, size: usize, log_size: u32, inverse: bool, coset: bool) {
    let n = coeffs.len();
    coeffs.resize(size, MpcField::Public(Fr::zero()));
    let lay = mpc_field_layout::<S>();
    CTX.with(|c| check(unsafe { zk_mpc_fft_in_place(*c, coeffs.as_mut_ptr() as *mut c_void, n, &lay, log_size, inverse as i32, coset as i32) }));
}
pub fn mpc_divide_by_vanishing_poly_on_coset_in_place<S: FieldShare<Fr> + HipLanes>(evals: &mut [MpcField<Fr, S>], log_size: u32) {
    assert_eq!(evals.len(), 1usize << log_size);
    let lay = mpc_field_layout::<S>();
    CTX.with(|c| check(unsafe { zk_mpc_divide_by_vanishing_on_coset_in_place(*c, evals.as_mut_ptr() as *mut c_void, &lay, log_size) }));
}
// MpcField::batch_product_in_place (wire/field.rs:917-958): Beaver through the vtable when both slices are shared
pub fn mpc_batch_product_in_place<S: FieldShare<Fr> + HipLanes>(selfs: &mut [MpcField<Fr, S>], others: &[MpcField<Fr, S>]) {
    let n = selfs.len().min(others.len());
    let lay = mpc_field_layout::<S>();
    let net = net_vtable();
    let mut sent = 0u64;
    CTX.with(|c| check(unsafe { zk_mpc_batch_product_in_place(*c, selfs.as_mut_ptr() as *mut c_void, others.as_ptr() as *const c_void, n, &lay,
        std::ptr::null() /* DummyFieldTripleSource, as the wire passes (wire/field.rs:941-947) */, &net, &mut sent) }));
}
// MpcG1Affine::multi_scalar_mul (wire/pairing.rs:714-777) on the key's own slice: (share lane, MAC lane, every scalar public?)
pub fn mpc_msm_g1<S: FieldShare<Fr> + HipLanes>(bases: *const c_void, n_bases: usize, base_layout: &ZkMpcGroupLayout, scalars: &[MpcField<Fr, S>]) -> ([G1Projective; 2], bool) {
    let z = ZkG1Projective { x: ZkFq { l: [0; 6] }, y: ZkFq { l: [0; 6] }, z: ZkFq { l: [0; 6] } };
    let mut out = [z, z];
    let mut all_public = 0i32;
    let lay = mpc_field_layout::<S>();
    CTX.with(|c| check(unsafe { zk_mpc_msm_g1(*c, bases, n_bases, base_layout, scalars.as_ptr() as *const c_void, scalars.len(), &lay, out.as_mut_ptr(), &mut all_public) }));
    let f = |o: &ZkG1Projective| G1Projective::new(ark_ff::Fp384::new(ark_ff::BigInteger384(o.x.l)), ark_ff::Fp384::new(ark_ff::BigInteger384(o.y.l)),
                                                   ark_ff::Fp384::new(ark_ff::BigInteger384(o.z.l)));
    ([f(&out[0]), f(&out[1])], all_public != 0)
}
pub fn mpc_msm_g2<S: FieldShare<Fr> + HipLanes>(bases: *const c_void, n_bases: usize, base_layout: &ZkMpcGroupLayout, scalars: &[MpcField<Fr, S>]) -> ([G2Projective; 2], bool) {
    let z6 = ZkFq { l: [0; 6] };
    let z = ZkG2Projective { x: [z6; 2], y: [z6; 2], z: [z6; 2] };
    let mut out = [z, z];
    let mut all_public = 0i32;
    let lay = mpc_field_layout::<S>();
    CTX.with(|c| check(unsafe { zk_mpc_msm_g2(*c, bases, n_bases, base_layout, scalars.as_ptr() as *const c_void, scalars.len(), &lay, out.as_mut_ptr(), &mut all_public) }));
    let q = |a: &[ZkFq; 2]| ark_bls12_377::Fq2::new(ark_ff::Fp384::new(ark_ff::BigInteger384(a[0].l)), ark_ff::Fp384::new(ark_ff::BigInteger384(a[1].l)));
    let f = |o: &ZkG2Projective| G2Projective::new(q(&o.x), q(&o.y), q(&o.z));
    ([f(&out[0]), f(&out[1])], all_public != 0)
}
// The body of `fn multi_scalar_mul(bases: &[Self], scalars: &[Self::ScalarField])` in wire/pairing.rs:714 then reads
//     let lay = mpc_group_layout(&bases[0], ...);                       // once per type: cache it in a OnceCell
//     let (lanes, all_public) = mpc_msm_g1::<PS::FrShare>(bases.as_ptr() as *const c_void, bases.len(), &lay, scalars);
//     $w_pro { val: MpcGroup::Shared(if all_public { <PS::$share_proj as Reveal>::from_public(lanes[0]) }      // :726-741
//                                    else { PS::$share_proj::from_lanes(lanes[0], lanes[1]) }) }                // multi_scale_pub_group, :750-756
// -- the assertion `bases.iter().all(|b| !b.is_shared())` is made by the library (ZK_ERR_ARG -> panic), the Vec copies of
// all_public_or_shared are gone, and the key's tables are found again by content whatever Vec they arrive in.
// A prover whose key outlives it may skip the per-hit comparison of the caller's table with the cached one:
pub fn trust_base_tables(on: bool) { CTX.with(|c| check(unsafe { zk_bases_cache_trust(*c, on as i32) })); }
// calculate_coeff's three MSMs run over ONE `assignment` (src/groth16.rs:137-160): the library starts the next two while the first
// reduces and releases a result only for the same cached table and scalars equal word for word.  Off for provers sharing one GPU:
pub fn start_msms_ahead(on: bool) { CTX.with(|c| check(unsafe { zk_msm_speculate(*c, on as i32) })); }

// ---- whole provers (src/groth16.rs:68-183 over shares; the plain prover with a resident key) ----
pub fn create_proof_shared(pk: *const ZkPk, r1cs: *const ZkR1cs, z_share_dev: *const c_void, r_share: &Fr, s_share: &Fr) -> [u8; 192] {
    let mut proof = [0u8; 192];
    let mut sent = 0u64;
    let net = net_vtable();
    CTX.with(|c| check(unsafe { zk_groth16_prove_shared(*c, pk, r1cs, z_share_dev, r_share as *const Fr as *const ZkFr, s_share as *const Fr as *const ZkFr,
        std::ptr::null(), std::ptr::null(), std::ptr::null() /* DummyFieldTripleSource */, &net, proof.as_mut_ptr(), &mut sent) }));
    proof
}
pub fn create_proof_local(pk: *const ZkPk, r1cs: *const ZkR1cs, full_assignment: &[Fr], r: &Fr, s: &Fr) -> [u8; 192] {
    let mut proof = [0u8; 192];
    CTX.with(|c| check(unsafe { zk_groth16_prove(*c, pk, r1cs, full_assignment.as_ptr() as *const ZkFr, r as *const Fr as *const ZkFr,
        s as *const Fr as *const ZkFr, proof.as_mut_ptr()) }));
    proof
}
'''


def main():
    ffi, _ = generate()
    files = {"hip_ffi.rs": ffi, "overrides.rs": OVERRIDES}
    if "--check" in sys.argv:
        bad = [f for f, text in files.items() if not os.path.exists(os.path.join(OUT_DIR, f)) or open(os.path.join(OUT_DIR, f)).read() != text]
        if bad:
            sys.exit("bindings/%s out of date: run python tools/gen_rust_ffi.py" % ", bindings/".join(bad))
        return
    os.makedirs(OUT_DIR, exist_ok=True)
    for f, text in files.items():
        open(os.path.join(OUT_DIR, f), "w").write(text)
    print("wrote bindings/hip_ffi.rs (%d lines), bindings/overrides.rs (%d lines)" % (ffi.count("\n"), OVERRIDES.count("\n")))


if __name__ == "__main__":
    main()
