"""Collaborative (N-party) Groth16 proving over additive shares: one party per GPU.

Mirrors the reference's MPC path (paths relative to the Yoii-Inc/zk-mpc tree):
  create_proof over MpcField / MpcGroup                  src/groth16.rs:68-183
  FieldShare::batch_mul (Beaver, two vector opens)       mpc-algebra/src/share/field.rs:97-129
  AdditiveFieldShare::{batch_open, reveal, from_public}  mpc-algebra/src/share/additive.rs:81-131
  GroupShare::scale (group Beaver, two scalar opens)     mpc-algebra/src/share/group.rs:72-111
  DummyFieldTripleSource / DummyGroupTripleSource        mpc-algebra/src/wire/{field.rs:49-63, group.rs:41-71}
  shift() adds public constants on the leader only       share/additive.rs:147-152
  Proof::reveal                                          arkworks/groth16/src/reveal.rs:7-10
  MpcSerNet::broadcast / MpcNet::broadcast_bytes         mpc-algebra/src/channel.rs:12-28, mpc-net/src/multi.rs:469-525

Everything that is linear in the shares (sparse mat-vec, 7 NTTs, 5 MSMs) is local to a party and runs
in libzkmpc_hip on that party's GPU.  An "open" of a length-n share vector is an all-gather of the
parties' vectors (RCCL over xGMI when the transport is torch.distributed/nccl; the reference's
broadcast returns all payloads ordered by party id and the caller sums) followed by one HIP kernel
that sums the N vectors mod r.  Elements travel as raw Montgomery limbs: addition commutes with
the Montgomery factor, so no (de)serialisation is needed.

The protocol code below is written against two small interfaces:
  backend : the arithmetic (GpuBackend = libzkmpc_hip through zk_mpc_amd.api.Context).  Tests may
            inject another backend to exercise the protocol and transport logic without a GPU; the
            product ships only GpuBackend and has no CPU arithmetic.
  net     : the transport (DistNet = torch.distributed; LocalNet = N parties as threads of one
            process, the analogue of the reference's LocalTestNet, mpc-net/src/multi.rs:357-453).
"""
from __future__ import annotations

import os
import threading

import numpy as np


# ------------------------------------------------------------------------------------------------
# transports
# ------------------------------------------------------------------------------------------------


class DistNet:
    """torch.distributed transport: backend "nccl" (= RCCL) for device buffers, "gloo" on CPU."""

    def __init__(self, dist, device=None, open_pattern=None):
        """open_pattern: None = by party count (all-gather for two parties, all-to-all of slices for three or more),
        "allgather" or "a2a" to force one; a constructor argument, not an environment variable: every party of a run
        must make the same choice, or the parties wait for each other in different collectives."""
        import torch
        self.torch = torch
        self.dist = dist
        if open_pattern not in (None, "allgather", "a2a"):
            raise ValueError("open_pattern must be None, 'allgather' or 'a2a'")
        self.open_pattern = open_pattern
        self.rank = dist.get_rank()
        self.n = dist.get_world_size()
        self.device = device if device is not None else torch.device("cpu")

    def is_leader(self) -> bool:
        return self.rank == 0

    def new_buffer(self, nbytes: int):
        """A transport-visible buffer of nbytes (multiple of 8); returns (tensor, address)."""
        t = self.torch.empty(nbytes // 8, dtype=self.torch.int64, device=self.device)
        return t, t.data_ptr()

    def all_gather(self, send_tensor, recv_tensor):
        """recv = concat over parties (ordered by party id) of send  (MpcNet::broadcast_bytes)."""
        self.dist.all_gather_into_tensor(recv_tensor, send_tensor)
        if self.device.type == "cuda":
            self.torch.cuda.current_stream().synchronize()

    def open_sum(self, send, n: int, sum_parties, buffer):
        """Every party learns the element-wise sum over parties of `send` (n field elements of 4 int64 words each): the
        "open" of a share vector.  Reduce-scatter then all-gather over point-to-point links: party j receives slice j of
        every party (all_to_all), sums its slice, and the summed slices are all-gathered -- 2 x 32 n bytes in per GPU
        for any number of parties, where all-gather-then-sum moves P x 32 n (xGMI is point-to-point: 7 links per GPU,
        so the all-to-all pattern is the native one).  With 2 parties both patterns move the same bytes and the
        all-gather is one collective instead of two, so it is kept there.
        sum_parties(gathered, n_parts, m, out): out[i] = sum_p gathered[p*m + i] mod r on the caller's arithmetic;
        buffer(name, nbytes) -> int64 tensor that stays valid until the next call with the same name."""
        N = self.n
        words = 4 * n
        if (N < 3 and self.open_pattern != "a2a") or self.open_pattern == "allgather":
            recv = buffer("open_recv", N * n * 32)
            self.dist.all_gather_into_tensor(recv, send[:words])
            out = buffer("open_out", n * 32)
            self._sync()
            sum_parties(recv, N, n, out)
            return out
        chunk = (n + N - 1) // N
        if chunk * N != n:
            padded = buffer("open_pad", N * chunk * 32)
            padded[:words] = send[:words]
            padded[words:] = 0
            send = padded
        recv = buffer("open_recv", N * chunk * 32)
        self.dist.all_to_all_single(recv, send[:N * chunk * 4])          # recv[p] = slice `rank` of party p
        part = buffer("open_part", chunk * 32)
        self._sync()
        sum_parties(recv, N, chunk, part)
        full = buffer("open_full", N * chunk * 32)
        self.dist.all_gather_into_tensor(full, part)
        self._sync()
        return full[:words]

    def scatter(self, parts, nbytes: int):
        """The leader ("king") hands parts[p] to party p; everybody returns its own part
        (MpcNet::worker_receive_or_leader_send_element, used by king_share: mpc-algebra/src/share/additive.rs:98-107).
        parts: list of N int64 tensors of nbytes / 8 words on the leader, None elsewhere."""
        recv = self.torch.empty(nbytes // 8, dtype=self.torch.int64, device=self.device)
        self.dist.scatter(recv, scatter_list=[p.reshape(-1) for p in parts] if self.rank == 0 else None, src=0)
        self._sync()
        return recv

    def _sync(self):
        if self.device.type == "cuda":
            self.torch.cuda.current_stream().synchronize()

    def all_gather_small(self, arr: np.ndarray) -> list:
        t = self.torch.from_numpy(np.ascontiguousarray(arr).view(np.int64).reshape(-1).copy()).to(self.device)
        out = self.torch.empty(self.n * t.numel(), dtype=self.torch.int64, device=self.device)
        self.dist.all_gather_into_tensor(out, t)
        res = out.cpu().numpy().view(arr.dtype).reshape((self.n,) + arr.shape)
        return [res[p] for p in range(self.n)]

    def barrier(self):
        self.dist.barrier()


class LocalNet:
    """N parties as threads of one process sharing host-visible buffers (LocalTestNet analogue)."""

    class _Shared:
        def __init__(self, n):
            self.n = n
            self.barrier = threading.Barrier(n)
            self.slots = [None] * n

    @staticmethod
    def create(n):
        sh = LocalNet._Shared(n)
        return [LocalNet(sh, p) for p in range(n)]

    def __init__(self, shared, rank):
        self.sh, self.rank, self.n = shared, rank, shared.n

    def is_leader(self):
        return self.rank == 0

    def exchange(self, obj) -> list:
        self.sh.slots[self.rank] = obj
        self.sh.barrier.wait()
        out = list(self.sh.slots)
        self.sh.barrier.wait()
        return out

    def all_gather_small(self, arr):
        return self.exchange(np.array(arr, copy=True))

    def scatter(self, parts, nbytes: int):
        """parts: list of N arrays on the leader, None elsewhere; returns this party's part."""
        return self.exchange(parts if self.rank == 0 else None)[0][self.rank]

    def barrier(self):
        self.sh.barrier.wait()


# ------------------------------------------------------------------------------------------------
# GPU backend (the product)
# ------------------------------------------------------------------------------------------------


class GpuBackend:
    """Arithmetic on this party's GPU through libzkmpc_hip.  Vectors are device addresses."""

    def __init__(self, ctx, net):
        self.ctx, self.net = ctx, net
        self._bufs = {}
        self._const = {}
        self._tensors = {}

    # -- buffers --
    def vec(self, name, n):
        """A named device vector of n field elements.  With a torch.distributed transport the storage
        is a torch tensor, so the vector can be handed to all_gather without a copy."""
        key = (name, n)
        if key not in self._bufs:
            if isinstance(self.net, DistNet) and self.net.device.type == "cuda":
                t, ptr = self.net.new_buffer(n * 32)
                self._bufs[key] = (t, ptr)
                self._tensors[ptr] = t
            else:
                b = self.ctx.alloc(n * 32)
                self._bufs[key] = (b, b.ptr)
        return self._bufs[key][1]

    def const_vec(self, value_mont4, n):
        """n copies of one field element (used for the dummy triples)."""
        key = (tuple(int(x) for x in value_mont4), n)
        if key not in self._const:
            self._const[key] = self.ctx.upload(np.tile(np.asarray(value_mont4, dtype=np.uint64), (n, 1)))
        return self._const[key].ptr

    # -- vector arithmetic --
    def add(self, a, b, out, n):
        self.ctx.fr_vec_op_dev(1, a, b, out, n)

    def sub(self, a, b, out, n):
        self.ctx.fr_vec_op_dev(2, a, b, out, n)

    def open_stats(self, proofs: int = 1) -> dict:
        """Wall time this party spent inside the vector opens since the counters were last read (each open ends with a device
        synchronisation, so the figure includes the local sum); `proofs` divides the totals."""
        s, c, e = getattr(self, "_open_s", 0.0), getattr(self, "_open_calls", 0), getattr(self, "_open_elems", 0)
        self._open_s, self._open_calls, self._open_elems = 0.0, 0, 0
        return {"opens_per_proof": round(c / max(proofs, 1), 2), "ms_per_open": round(s / c * 1e3, 3) if c else None,
                "ms_per_proof": round(s / max(proofs, 1) * 1e3, 3), "elements_per_open": (e // c) if c else 0}

    def open_vec(self, v, out, n):
        """out = sum over parties of v (AdditiveFieldShare::batch_open)."""
        import time as _time
        t0 = _time.perf_counter()
        try:
            self._open_vec(v, out, n)
        finally:
            self._open_s = getattr(self, "_open_s", 0.0) + (_time.perf_counter() - t0)
            self._open_calls = getattr(self, "_open_calls", 0) + 1
            self._open_elems = getattr(self, "_open_elems", 0) + n

    def _open_vec(self, v, out, n):
        net, ctx = self.net, self.ctx
        if getattr(self, "native_open", False):
            ctx.open_sum_fr_dev(v, n, out)       # zk_open_sum_fr_dev: RCCL inside the library, on the context's stream
            ctx.sync()
            return
        if isinstance(net, DistNet) and net.device.type == "cuda":
            st = self._tensors.get(v)
            if st is None:                       # not one of our tensors: stage it
                sp = self.vec("xchg_send", n)
                st = self._tensors[sp]
                ctx.fr_vec_op_dev(1, v, self.const_vec(np.zeros(4, dtype=np.uint64), n), sp, n)
            ctx.sync()                           # the vector kernels ran on the context's stream

            def sum_parties(gathered, n_parts, m, dst):
                ctx.fr_sum_parties_dev(gathered.data_ptr(), n_parts, m, dst.data_ptr())
                ctx.sync()

            def buffer(name, nbytes):
                return self._tensors[self.vec(name, nbytes // 32)]
            res = net.open_sum(st, n, sum_parties, buffer)
            ctx.memcpy_d2d(out, res.data_ptr(), n * 32)
            ctx.sync()
        else:
            # LocalNet (parties share one process), or a torch.distributed backend without device collectives (gloo: the
            # exchange crosses host memory, as the reference's TCP mesh does -- mpc-net/src/multi.rs:469-525); the sum runs on
            # the device either way
            mine = ctx.download(v, (n, 4))
            allv = net.all_gather_small(mine) if isinstance(net, DistNet) else net.exchange(mine)
            g = ctx.upload(np.concatenate(allv, axis=0))
            ctx.fr_sum_parties_dev(g.ptr, net.n, n, out)
            ctx.sync()
            g.free()

    def beaver_combine(self, sx, oy, tx, ty, tz, out, n):
        self.ctx.beaver_combine_dev(sx, oy, out, n, triple=(tx, ty, tz))

    def king_share(self, values, n, key32=None):
        """AdditiveFieldShare::king_share over a vector: the leader draws N-1 uniform share vectors (F::rand per element:
        zk_fr_random_dev, ChaCha20 keyed from the operating system's CSPRNG; one stream id per receiving party), sets the
        last to values - sum, and scatters them (share/additive.rs:98-107); `values` is a device vector on the leader,
        ignored elsewhere.  key32: a 32-byte ChaCha20 key for reproducible shares -- tests only; the default (None) must be
        used whenever the values are secret.  Returns this party's share (device vector)."""
        import torch
        ctx, net = self.ctx, self.net
        N = net.n
        parts = None
        if net.is_leader():
            dev = torch.device("cuda", ctx.device)
            if key32 is None:
                key32 = os.urandom(32)          # one key per call; the N - 1 vectors take distinct stream ids under it
            parts = []
            last = torch.empty(n * 4, dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            ctx.memcpy_d2d(last.data_ptr(), values, n * 32)
            for p in range(N - 1):
                t = torch.empty(n * 4, dtype=torch.int64, device=dev)
                torch.cuda.synchronize()
                ctx.fr_random_dev(t.data_ptr(), n, key32, stream_id=p)
                ctx.fr_vec_op_dev(2, last.data_ptr(), t.data_ptr(), last.data_ptr(), n)
                parts.append(t)
            ctx.sync()
            parts.append(last)
        if isinstance(net, DistNet) and net.device.type == "cuda":
            mine = net.scatter(parts, n * 32)
            self._tensors[mine.data_ptr()] = mine
            return mine.data_ptr()
        if isinstance(net, DistNet):          # no device collectives (gloo): the shares cross host memory
            mine = net.scatter([p.cpu() for p in parts] if parts is not None else None, n * 32)
            b = ctx.upload(mine.numpy().view(np.uint64).reshape(n, 4))
            self._bufs[("king_share", n, len(self._bufs))] = (b, b.ptr)
            return b.ptr
        host = [p.cpu().numpy().view(np.uint64).reshape(n, 4) for p in parts] if parts is not None else None
        b = ctx.upload(net.scatter(host, n * 32))
        self._bufs[("king_share", n, len(self._bufs))] = (b, b.ptr)
        return b.ptr

    def is_zero_vec(self, v, n) -> bool:
        return self.ctx.fr_vec_is_zero_dev(v, n)

    # -- Groth16 pieces --
    def domain_size(self, r1cs):
        return 1 << r1cs.domain_log

    def witness_map_pre(self, r1cs, z, a, b, c):
        self.ctx.witness_map_pre_dev(r1cs, z, a, b, c, True)

    def witness_map_post(self, r1cs, ab, c):
        self.ctx.witness_map_post_dev(r1cs, ab, c)

    def msms_presort(self, pk, r1cs, z):
        self.ctx.groth16_msms_presort_dev(pk, r1cs, z)

    def msms_begin(self, pk, r1cs, z):
        self.ctx.groth16_msms_begin_dev(pk, r1cs, z)

    def msms(self, pk, r1cs, z, h):
        return self.ctx.groth16_msms_dev(pk, r1cs, z, h)

    # -- host-side group / field helpers (O(1) per proof) --
    def g1_add(self, a, b): return self.ctx.g1_add(a, b)
    def g2_add(self, a, b): return self.ctx.g2_add(a, b)
    def g1_neg(self, a): return self.ctx.g1_neg(a)
    def g2_neg(self, a): return self.ctx.g2_neg(a)
    def g1_mul(self, a, k): return self.ctx.g1_mul(a, k)
    def g2_mul(self, a, k): return self.ctx.g2_mul(a, k)
    def g1_from_affine(self, a): return self.ctx.g1_from_affine(a)
    def g2_from_affine(self, a): return self.ctx.g2_from_affine(a)
    def g1_serialize(self, a): return self.ctx.g1_serialize(a)
    def g2_serialize(self, a): return self.ctx.g2_serialize(a)
    def fr_add(self, a, b): return self.ctx.fr_op("add", a, b)
    def fr_sub(self, a, b): return self.ctx.fr_op("sub", a, b)
    def fr_mul(self, a, b): return self.ctx.fr_op("mul", a, b)

    def g1_zero(self): return self.ctx.g1_from_affine(np.zeros(12, dtype=np.uint64))
    def g2_zero(self): return self.ctx.g2_from_affine(np.zeros(24, dtype=np.uint64))

    def pk_points(self, pk):
        """Public key elements the assembly needs: alpha_g1, beta_g1, delta_g1, beta_g2, delta_g2 and query[0]s."""
        f1, f2 = self.ctx.g1_from_affine, self.ctx.g2_from_affine
        return dict(alpha_g1=f1(pk.vk_g1(0)), beta_g1=f1(pk.vk_g1(1)), delta_g1=f1(pk.vk_g1(2)),
                    beta_g2=f2(pk.vk_g2(0)), delta_g2=f2(pk.vk_g2(1)),
                    a0=f1(pk.download("a_query", 0, 1)[0]), b0_g1=f1(pk.download("b_g1_query", 0, 1)[0]),
                    b0_g2=f2(pk.download("b_g2_query", 0, 1)[0]))

    def fr_one(self):
        out = np.zeros(4, dtype=np.uint64)
        canon = np.array([1, 0, 0, 0], dtype=np.uint64)
        import ctypes as C
        self.ctx._ck(self.ctx.lib.zk_fr_from_canonical(canon.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)))
        return out


# ------------------------------------------------------------------------------------------------
# the protocol
# ------------------------------------------------------------------------------------------------


class Party:
    """One MPC party: holds additive shares, runs the collaborative prover."""

    def __init__(self, ctx=None, dist=None, net=None, backend=None):
        if net is None:
            import torch
            dev = torch.device("cuda", ctx.device) if (dist.get_backend() == "nccl") else torch.device("cpu")
            net = DistNet(dist, dev)
        self.net = net
        # leadership is decided from net.rank here and from ctx->party_id inside the library (zk_beaver_combine_dev adds
        # the public sx*oy term on party 0): the two must be the same party, or every rank adds it
        if ctx is not None and backend is None and (ctx.party_id != net.rank or ctx.n_parties != net.n):
            raise ValueError("Party: Context(party_id=%d, n_parties=%d) does not match the transport (rank %d of %d)"
                             % (ctx.party_id, ctx.n_parties, net.rank, net.n))
        self.be = backend if backend is not None else GpuBackend(ctx, net)
        self.ctx = ctx
        # ZK_TRANSPORT=native: the share-vector opens go through the library's own RCCL communicator (comm.hip) instead of
        # torch.distributed; the 128-byte id travels over the existing process group, as it would over the reference's
        # TCP mesh.  Small opens (points, scalars) stay on the process group.
        if os.environ.get("ZK_TRANSPORT") == "native" and isinstance(net, DistNet) and backend is None:
            box = [ctx.comm_unique_id() if net.is_leader() else None]
            net.dist.broadcast_object_list(box, src=0)
            ctx.comm_init(box[0], net.rank, net.n)
            self.be.native_open = True
        self.bytes_sent = 0   # payload bytes this party contributed to opens (cf. mpc-net/src/multi.rs:527-536)

    @property
    def leader(self) -> bool:
        return self.net.is_leader()

    def _pk_points(self, pk):
        """The key's O(1) public points, fetched once and kept ON the key object (a cache keyed by id(pk) would hand a new
        key allocated at a recycled address the old key's points)."""
        P = getattr(pk, "_mpc_points", None)
        if P is None:
            P = self.be.pk_points(pk)
            try:
                pk._mpc_points = P
            except AttributeError:
                pass
        return P

    # ---- sharing helpers (input distribution; not on the proving path) ----
    def share_scalars(self, values, seed: int):
        """Deterministic additive shares of public test scalars: every party derives all N shares from
        the seed and keeps its own (stands in for async_king_share, share/additive.rs:98-107)."""
        from .convert import fr_to_mont, R_MOD
        rs = np.random.RandomState(seed & 0x7FFFFFFF)
        out = []
        for v in values:
            sh = [int.from_bytes(rs.bytes(40), "little") % R_MOD for _ in range(self.net.n - 1)]
            sh.append((int(v) - sum(sh)) % R_MOD)
            out.append(fr_to_mont([sh[self.net.rank]])[0])
        return out

    def king_share_vec(self, values, n: int, key32=None):
        """Input distribution by the leader (Reveal::king_share / king_share_batch): see GpuBackend.king_share.
        key32 = None (the default, and the only choice for secret inputs): masks keyed from the OS CSPRNG."""
        self.bytes_sent += (self.net.n - 1) * n * 32 if self.leader else 0
        return self.be.king_share(values, n, key32)

    def share_assignment_dev(self, z_dev, r1cs, seed: int):
        """This party's additive share of a full assignment that is resident on its own device
        (bench / test input generation).  Instance variables are public: the leader holds them, the
        others hold zero (Reveal::from_public, share/additive.rs:89-93); witness variables are split
        into N shares, shares 0..N-2 pseudo-random and share N-1 the difference."""
        import torch
        ctx, net = self.ctx, self.net
        ni, nw = r1cs.num_instance, r1cs.num_witness
        m = ni + nw
        dev = torch.device("cuda", ctx.device)
        mine = torch.zeros((m, 4), dtype=torch.int64, device=dev)

        def rnd(p):
            g = torch.Generator(device=dev)
            g.manual_seed(seed * 1000 + p)
            t = torch.randint(0, 1 << 62, (nw, 4), dtype=torch.int64, device=dev, generator=g)
            t[:, 3] &= (1 << 60) - 1          # < 2^252 < r: a valid residue
            return t

        zsrc = z_dev.ptr if hasattr(z_dev, "ptr") else int(z_dev)
        if net.rank < net.n - 1:
            mine[ni:] = rnd(net.rank)
            torch.cuda.synchronize()
        else:
            acc = torch.empty((nw, 4), dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            ctx._ck(0)
            # acc = z_witness - sum_{p<N-1} share_p
            ctx.fr_vec_op_dev(1, zsrc + ni * 32, self.be.const_vec(np.zeros(4, dtype=np.uint64), nw), acc.data_ptr(), nw)
            for p in range(net.n - 1):
                s = rnd(p)
                torch.cuda.synchronize()
                ctx.fr_vec_op_dev(2, acc.data_ptr(), s.data_ptr(), acc.data_ptr(), nw)
                ctx.sync()
            mine[ni:] = acc
            torch.cuda.synchronize()
        if net.is_leader():
            pub = ctx.download(zsrc, (ni, 4))
            mine[:ni] = torch.from_numpy(pub.view(np.int64)).to(dev)
            torch.cuda.synchronize()
        self._keep = mine
        return mine.data_ptr()

    # ---- field Beaver (vector) ----
    def beaver_batch_mul(self, x, y, out, n, triple=None):
        """FieldShare::batch_mul: out = shares of x*y.  triple = (tx, ty, tz) device vectors or None for
        DummyFieldTripleSource (the leader holds 1, everybody else 0)."""
        be = self.be
        if triple is None:
            c = be.fr_one() if self.leader else np.zeros(4, dtype=np.uint64)
            tx = ty = tz = be.const_vec(c, n)
        else:
            tx, ty, tz = triple
        sx_l, oy_l = be.vec("bv_sx_l", n), be.vec("bv_oy_l", n)
        sx, oy = be.vec("bv_sx", n), be.vec("bv_oy", n)
        be.add(x, tx, sx_l, n)                   # s + x
        be.add(y, ty, oy_l, n)                   # o + y
        be.open_vec(sx_l, sx, n)                 # open(s + x)
        be.open_vec(oy_l, oy, n)                 # open(o + y)
        self.bytes_sent += 2 * n * 32
        be.beaver_combine(sx, oy, tx, ty, tz, out, n)   # z - sx*y - oy*x (+ sx*oy on the leader)

    # ---- group Beaver: shared point * shared scalar (GroupShare::scale) ----
    def _open_g(self, p, add):
        parts = self.net.all_gather_small(np.ascontiguousarray(p, dtype=np.uint64))
        self.bytes_sent += p.nbytes
        acc = parts[0]
        for q in parts[1:]:
            acc = add(acc, q)
        return acc

    def _open_fr(self, s):
        parts = self.net.all_gather_small(np.ascontiguousarray(s, dtype=np.uint64))
        self.bytes_sent += 32
        acc = parts[0]
        for q in parts[1:]:
            acc = self.be.fr_add(acc, q)
        return acc

    def scale_g1(self, s_pt, o_sc, lazy=False):
        """GroupShare::scale with DummyGroupTripleSource: x = 0, y = [leader ? 1 : 0], z = 0.
        lazy: the two opens happen here, in program order; the local arithmetic behind them (a full scalar multiplication on
        the leader) goes to a host thread and the call returns its future."""
        be = self.be
        y = be.fr_one() if self.leader else np.zeros(4, dtype=np.uint64)
        sx = self._open_g(s_pt, be.g1_add)                         # open(s + x), x = 0
        oy = self._open_fr(be.fr_add(o_sc, y))                     # open(o + y)

        def finish():
            out = be.g1_neg(be.g1_mul(sx, y))                      # z - scale_pub_group(sx, y)       (z = 0)
            # - x * oy with x = 0 contributes nothing
            if self.leader:
                out = be.g1_add(out, be.g1_mul(sx, oy))            # shift(sx * oy): leader only
            return out
        return self._early(finish) if lazy else finish()

    def close(self):
        """Release the host helper threads (the pool of _early)."""
        pool = getattr(self, "_pool", None)
        if pool is not None:
            pool.shutdown(wait=True)
            self._pool = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _early(self, fn, *args):
        """fn(*args) on a host thread: a future.  For host-side group algebra that need not wait for the device."""
        pool = getattr(self, "_pool", None)
        if pool is None:
            from concurrent.futures import ThreadPoolExecutor
            pool = self._pool = ThreadPoolExecutor(max_workers=6)
        return pool.submit(fn, *args)

    def _open_many(self, frs=(), g1s=(), g2s=()):
        """Several opens in ONE collective: the sums over parties of scalar / G1 / G2 shares (the values are those of the
        one-by-one opens; a proof has a dozen small opens and each costs a collective with a host round trip)."""
        be = self.be
        items = [np.ascontiguousarray(x, dtype=np.uint64).reshape(-1) for x in list(frs) + list(g1s) + list(g2s)]
        if not items:
            return [], [], []
        arr = np.concatenate(items)
        parts = self.net.all_gather_small(arr)
        self.bytes_sent += arr.nbytes
        outs, pos = ([], [], []), 0
        for k, (group, width, add) in enumerate(((frs, 4, be.fr_add), (g1s, 18, be.g1_add), (g2s, 36, be.g2_add))):
            for _ in group:
                acc = np.array(parts[0][pos:pos + width], dtype=np.uint64)
                for q in parts[1:]:
                    acc = add(acc, np.ascontiguousarray(q[pos:pos + width], dtype=np.uint64))
                outs[k].append(acc)
                pos += width
        return outs

    def _scale_finish(self, sx, oy, y):
        """The local part of GroupShare::scale behind its two opens: z - sx*y (+ sx*oy on the leader), z = 0."""
        be = self.be
        out = be.g1_neg(be.g1_mul(sx, y))
        if self.leader:
            out = be.g1_add(out, be.g1_mul(sx, oy))
        return out

    # ---- reveal ----
    def reveal_g1(self, p): return self._open_g(p, self.be.g1_add)
    def reveal_g2(self, p): return self._open_g(p, self.be.g2_add)

    # ---- the collaborative prover ----
    def _net_vtable(self):
        """zk_net_vtable over this party's transport: (struct, errors, keep-alive).  all_gather_bytes = MpcNet::broadcast_bytes
        (mpc-net/src/lib.rs:60-64), open_sum_fr_dev = the vector open of this backend."""
        import ctypes as C
        be, net = self.be, self.net
        AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_uint8))
        OV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)

        class NetVtable(C.Structure):
            _fields_ = [("user", C.c_void_p), ("all_gather_bytes", AG), ("open_sum_fr_dev", OV)]
        errors = []

        def all_gather(_user, mine, length, out_all):
            try:
                arr = np.ctypeslib.as_array(mine, shape=(length,)).copy()
                parts = net.all_gather_small(arr.view(np.uint64))
                flat = np.concatenate([np.ascontiguousarray(p, dtype=np.uint64).reshape(-1) for p in parts]).view(np.uint8)
                C.memmove(out_all, flat.ctypes.data, flat.nbytes)
                return 0
            except Exception as e:      # an exception must not unwind through the C frames
                errors.append(e)
                return -1

        def open_vec(_user, v, n, out):
            try:
                be.open_vec(v, out, n)          # (the timed form: open_stats counts the opens of the one-call provers too)
                return 0
            except Exception as e:
                errors.append(e)
                return -1
        cbs = (AG(all_gather), OV(open_vec))
        return NetVtable(None, cbs[0], cbs[1]), errors, cbs

    def create_proof_shared_native(self, pk, r1cs, z_share, r_share, s_share, triple=None) -> bytes:
        """create_proof over additive shares as ONE library call (zk_groth16_prove_shared): what a Rust host would do.  The
        library calls back into this party's transport for the two small exchanges (MpcNet::broadcast_bytes) and for the two
        vector opens; everything else -- witness map halves, MSMs, Beaver tail, group algebra on shares -- stays inside.
        Same opened values and the same 192 bytes as create_proof_shared(fused=True), which remains the second implementation
        the tests compare it with."""
        import ctypes as C
        ctx = self.ctx
        vt, errors, _keep = self._net_vtable()
        from .api import _fr_struct
        r, s_ = _fr_struct(r_share), _fr_struct(s_share)
        out = np.zeros(192, dtype=np.uint8)
        sent = C.c_uint64(0)
        t = [C.c_void_p(int(x)) for x in triple] if triple is not None else [None, None, None]
        rc = ctx.lib.zk_groth16_prove_shared(ctx.h, pk.h, r1cs.h, C.c_void_p(int(z_share)), C.byref(r), C.byref(s_), t[0], t[1], t[2],
                                             C.byref(vt) if self.net.n > 1 else None, out.ctypes.data_as(C.c_void_p), C.byref(sent))
        if errors:
            raise errors[0]
        ctx._ck(rc)
        self.bytes_sent += int(sent.value)
        return out.tobytes()

    def create_proof_shared(self, pk, r1cs, z_share, r_share, s_share, triple=None, fused=True) -> bytes:
        """create_proof over additive shares (src/groth16.rs:68-183 with E = MpcPairingEngine).
        z_share: this party's share of the full assignment (device vector); r_share, s_share: (4,) uint64.
        Returns the revealed 192-byte proof (identical on every party).
        fused (default): the nine small opens of the three `scale` calls and of Proof::reveal travel in two collectives (every
        opened value is the same as in the reference's order; A is the opened s + x of the second scale).  fused=False keeps
        the reference's call order, one collective per open."""
        be = self.be
        D = be.domain_size(r1cs)
        P = self._pk_points(pk)
        # public point x shared scalar is local host arithmetic (0.4 ms per G1, 1.2 ms per G2 scalar multiplication): the three
        # that do not depend on the MSMs run on host threads under the device work (the library calls release the GIL)
        early = self._early(be.g1_mul, P["delta_g1"], r_share), self._early(be.g1_mul, P["delta_g1"], s_share), \
            self._early(be.g2_mul, P["delta_g2"], s_share)
        a, b, c = be.vec("wm_a", D), be.vec("wm_b", D), be.vec("wm_c", D)
        be.witness_map_pre(r1cs, z_share, a, b, c)                 # local: linear in the shares
        be.msms_begin(pk, r1cs, z_share)                           # the four MSMs over z run under the open and the second half below
        self.beaver_batch_mul(a, b, a, D, triple)                  # the one shared x shared vector product (:285)
        be.witness_map_post(r1cs, a, c)                            # h shares in `a`
        g1, g2 = be.msms(pk, r1cs, z_share, a)                     # party-local MSMs (multi_scale_pub_group)
        h_acc, l_acc, a_acc, b1_acc = g1[0], g1[1], g1[2], g1[3]
        pub1 = (lambda x: x) if self.leader else (lambda x: be.g1_zero())   # shift(): leader only
        pub2 = (lambda x: x) if self.leader else (lambda x: be.g2_zero())
        r_g1 = early[0].result()                                   # delta_g1 * r: public point * shared scalar, local
        if fused:
            y = be.fr_one() if self.leader else np.zeros(4, dtype=np.uint64)
            g_a = be.g1_add(be.g1_add(be.g1_add(r_g1, pub1(P["a0"])), a_acc), pub1(P["alpha_g1"]))
            g1_b = be.g1_add(be.g1_add(be.g1_add(early[1].result(), pub1(P["b0_g1"])), b1_acc), pub1(P["beta_g1"]))
            g2_b = be.g2_add(be.g2_add(be.g2_add(early[2].result(), pub2(P["b0_g2"])), g2), pub2(P["beta_g2"]))
            (oy_s, oy_r), (sx_rd, sx_a, sx_b), (B,) = self._open_many([be.fr_add(s_share, y), be.fr_add(r_share, y)], [r_g1, g_a, g1_b], [g2_b])
            parts = [self._early(self._scale_finish, sx, oy, y) for sx, oy in ((sx_rd, oy_s), (sx_a, oy_s), (sx_b, oy_r))]
            g_c = be.g1_add(parts[1].result(), parts[2].result())
            g_c = be.g1_add(g_c, be.g1_neg(parts[0].result()))
            g_c = be.g1_add(be.g1_add(g_c, l_acc), h_acc)
            C = self.reveal_g1(g_c)
            return be.g1_serialize(sx_a) + be.g2_serialize(B) + be.g1_serialize(C)
        r_s_delta = self.scale_g1(r_g1, s_share, lazy=True)        # :115
        g_a = be.g1_add(be.g1_add(be.g1_add(r_g1, pub1(P["a0"])), a_acc), pub1(P["alpha_g1"]))   # calculate_coeff
        s_g_a = self.scale_g1(g_a, s_share, lazy=True)             # :140
        s_g1 = early[1].result()
        g1_b = be.g1_add(be.g1_add(be.g1_add(s_g1, pub1(P["b0_g1"])), b1_acc), pub1(P["beta_g1"]))
        s_g2 = early[2].result()
        g2_b = be.g2_add(be.g2_add(be.g2_add(s_g2, pub2(P["b0_g2"])), g2), pub2(P["beta_g2"]))
        r_g1_b = self.scale_g1(g1_b, r_share, lazy=True)           # :161
        g_c = be.g1_add(s_g_a.result(), r_g1_b.result())           # :169-174
        g_c = be.g1_add(g_c, be.g1_neg(r_s_delta.result()))
        g_c = be.g1_add(g_c, l_acc)
        g_c = be.g1_add(g_c, h_acc)
        A, B, C = self.reveal_g1(g_a), self.reveal_g2(g2_b), self.reveal_g1(g_c)     # Proof::reveal
        return be.g1_serialize(A) + be.g2_serialize(B) + be.g1_serialize(C)


    # ---- collaborative Marlin (AHP rounds over additive shares) ----
    SHARED_POLYS = ("w", "z_a", "z_b", "mask_poly", "g_1", "h_1")     # the witness-dependent oracles; t, g_2, h_2 are public

    def marlin_prove_shared(self, index, powers_g, z_share, randomness_share, challenge_fn, triple_fn=None) -> dict:
        """Marlin::prove (arkworks/marlin/src/lib.rs:152-319) with F = MpcField over additive shares, on this party's GPU.

        Every step of the AHP rounds is linear in the witness except z_A * z_B in round 2 (`DensePolynomial::mul` on
        MpcField = FieldShare::batch_mul: one Beaver product of two vectors over the 4|H| multiplication domain); round 3
        involves public values only.  Commitments / evaluations / opening witnesses of witness-dependent polynomials are
        computed on the shares and revealed (`first_comms.publicize()`, `evaluations.publicize()` in the reference).

        z_share: this party's share of the padded assignment (device vector, instance part shared like the rest);
        randomness_share: this party's share of the prover's randomness (3 + 3|H| elements);
        challenge_fn(round, revealed_commitments) -> dict of challenges (the Fiat-Shamir transcript stays with the caller);
        triple_fn(n) -> (tx, ty, tz) device vectors, or None for the reference's DummyFieldTripleSource."""
        from . import marlin as DM
        be = self.be
        ctx = be.ctx
        m = DM.HostField.m

        def reveal_some(comms):
            return {l: (self.reveal_g1(c) if l in self.SHARED_POLYS else c) for l, c in comms.items()}

        def open_is_zero(v, n):
            tmp = be.vec("marlin_open", n)
            be.open_vec(v, tmp, n)
            return be.is_zero_vec(tmp, n)

        def batch_mul(x, y, out, n):
            self.beaver_batch_mul(x, y, out, n, triple_fn(n) if triple_fn else None)

        st = DM.prover_init(index, z_share, shared=True)
        r1 = DM.prover_first_round(st, randomness_share)
        comms = reveal_some(DM.commit(ctx, powers_g, r1))
        ch = dict(challenge_fn(1, comms))
        r2 = DM.prover_second_round(st, ch["alpha"], ch["eta_a"], ch["eta_b"], ch["eta_c"], batch_mul=batch_mul, open_is_zero=open_is_zero)
        comms.update(reveal_some(DM.commit(ctx, powers_g, r2)))
        ch.update(challenge_fn(2, comms))
        r3 = DM.prover_third_round(st, ch["beta"])
        comms.update(DM.commit(ctx, powers_g, r3))
        ch.update(challenge_fn(3, comms))
        polys = {**r1, **r2, **r3}
        ev = lambda l, pt: ctx.poly_evaluate_dev(polys[l].ptr, polys[l].n, m(pt))
        evals = {"g_1": self._open_fr(ev("g_1", ch["beta"])), "z_b": self._open_fr(ev("z_b", ch["beta"])),
                 "t": ev("t", ch["beta"]), "g_2": ev("g_2", ch["gamma"])}
        mine = lambda l: polys[l] if (l in self.SHARED_POLYS or self.leader) else None
        at_beta = [mine(l) for l in ("g_1", "z_b", "t", "mask_poly", "z_a", "w", "h_1")]
        ixp = index.polynomials()
        at_gamma = [polys["g_2"], polys["h_2"]] + [ixp[l] for l in sorted(ixp)]
        w_beta, w_gamma = DM.batch_open(ctx, powers_g, [(at_beta, ch["beta"]), (at_gamma, ch["gamma"])], ch["xi"])
        return {"commitments": comms, "evaluations": evals, "w_beta": self.reveal_g1(w_beta), "w_gamma": w_gamma, "challenges": ch}


    # ---- collaborative Marlin as a PROOF: transcript, hiding commitments, open_combinations over shares ----
    def marlin_prove_full(self, keys, z_share, zk_rng, triple_fn=None, mask_on_device=False):
        """MpcMarlin::prove (src/marlin.rs:56 -> arkworks/marlin/src/lib.rs:152-319 with F = MpcField) over additive shares:
        the complete proof, as `marlin.prove` emits it for one prover.  z_share: this party's share of the padded assignment
        (DevBuf; instance on the leader); zk_rng: this party's OWN generator -- every draw is a share (MpcField::rand), the
        effective randomness is the sum over parties.  The revealed proof is identical on every party and equal to the local
        proof on the summed inputs and summed randomness."""
        return _marlin_prove_full(self, keys, [z_share], zk_rng, triple_fn, spdz=False, mask_on_device=mask_on_device)

    def marlin_prove_shared_native(self, keys, z_share, zk_rng, triple=None, mask_on_device=False) -> bytes:
        """MpcMarlin::prove over additive shares as ONE library call (zk_marlin_prove_shared): what a Rust host would bind.
        Arguments as marlin_prove_full (triple: three device pointers of |MUL| Beaver shares, or None for dummy triples);
        returns Proof::serialize's bytes -- the same bytes as marlin_prove_full(...).to_bytes(), which stays the second
        implementation the tests compare with."""
        return _marlin_prove_native(self, keys, [z_share], zk_rng, triple, mask_on_device)

# ------------------------------------------------------------------------------------------------
# SPDZ (malicious-majority backend): every share carries a MAC share; opens are MAC-checked
# ------------------------------------------------------------------------------------------------


class MacCheckError(RuntimeError):
    pass


class SpdzParty(Party):
    """The reference's `malicious` backend (mpc-algebra/src/share/spdz.rs).  A shared value is a pair
    (sh, mac) of additive shares with sum(mac) = alpha * sum(sh); the MAC key alpha is itself shared
    (mac_share(): the reference's stand-in key is the constant 1 held by the leader, spdz.rs:31-37).

      SpdzFieldShare::{add,sub,scale,shift}   spdz.rs:197-219   both lanes; shift adds mac_share*c to the mac lane
      SpdzFieldShare::batch_open               spdz.rs:177-196   open sh, then dx = mac_share*x - mac, open dx, assert sum = 0
      FieldShare::batch_mul (default body)     share/field.rs:97-129 over SPDZ shares
      SpdzGroupShare::{batch_open, reveal, scale_pub_group, shift}   spdz.rs:284-309,425-480
      multi_scale_pub_group                    spdz.rs:482-488   (the reference feeds the SHARE values to both MSMs; with
                                                                  its key alpha = 1 that equals the MSM over the mac values
                                                                  that is computed here)
    Everything linear runs twice (share lane and mac lane); every open costs a second all-gather."""

    # ---- vectors: a value is a pair (sh, mac) of backend vectors ----
    def spdz_open_vec(self, v, out, n):
        be = self.be
        sh, mac = v
        be.open_vec(sh, out, n)                                    # x = sum of shares
        dx = be.vec("spdz_dx", n)
        zero = be.const_vec(np.zeros(4, dtype=np.uint64), n)
        be.sub(out if self.leader else zero, mac, dx, n)           # mac_share * x - mac
        chk = be.vec("spdz_chk", n)
        be.open_vec(dx, chk, n)
        self.bytes_sent += 2 * n * 32
        if not be.is_zero_vec(chk, n):
            raise MacCheckError("SPDZ MAC check failed on a vector open")

    def spdz_beaver_batch_mul(self, x, y, out, n, triple=None):
        """x, y, out: (sh, mac) pairs.  triple = ((tx_sh, tx_mac), (ty..), (tz..)) or None for the dummy source
        (from_public(1): the leader holds 1 in both lanes)."""
        be = self.be
        if triple is None:
            c = be.fr_one() if self.leader else np.zeros(4, dtype=np.uint64)
            one = be.const_vec(c, n)
            triple = ((one, one), (one, one), (one, one))
        tx, ty, tz = triple
        sxl = (be.vec("sp_sx_l0", n), be.vec("sp_sx_l1", n))
        oyl = (be.vec("sp_oy_l0", n), be.vec("sp_oy_l1", n))
        for lane in (0, 1):
            be.add(x[lane], tx[lane], sxl[lane], n)
            be.add(y[lane], ty[lane], oyl[lane], n)
        sx, oy = be.vec("sp_sx", n), be.vec("sp_oy", n)
        self.spdz_open_vec(sxl, sx, n)
        self.spdz_open_vec(oyl, oy, n)
        for lane in (0, 1):   # z - sx*y - oy*x + [leader] sx*oy : the shift lands on the leader in BOTH lanes (mac_share)
            be.beaver_combine(sx, oy, tx[lane], ty[lane], tz[lane], out[lane], n)

    # ---- scalars and group elements: pairs of host arrays ----
    def _check_fr(self, x, mac):
        dx = self.be.fr_sub(x if self.leader else np.zeros(4, dtype=np.uint64), mac)
        parts = self.net.all_gather_small(np.ascontiguousarray(dx, dtype=np.uint64))
        self.bytes_sent += 32
        acc = parts[0]
        for q in parts[1:]:
            acc = self.be.fr_add(acc, q)
        if np.any(acc):
            raise MacCheckError("SPDZ MAC check failed on a scalar open")

    def spdz_open_fr(self, v):
        x = self._open_fr(v[0])
        self._check_fr(x, v[1])
        return x

    def _spdz_open_g(self, v, add, neg, zero, ser):
        x = self._open_g(v[0], add)
        dx = add(x if self.leader else zero(), neg(v[1]))
        tot = self._open_g(dx, add)
        if ser(tot) != ser(zero()):
            raise MacCheckError("SPDZ MAC check failed on a group open")
        return x

    def _spdz_open_many(self, frs=(), g1s=(), g2s=()):
        """Party._open_many over SPDZ pairs (share, mac): the shares are opened in one collective, all the MAC checks
        (sum over parties of [leader ? x : 0] - mac_i = 0 with key share 1 on the leader: spdz.rs:177-196) in a second one."""
        be = self.be
        xf, x1, x2 = self._open_many([v[0] for v in frs], [v[0] for v in g1s], [v[0] for v in g2s])
        zf = np.zeros(4, dtype=np.uint64)
        df = [be.fr_sub(x if self.leader else zf, v[1]) for x, v in zip(xf, frs)]
        d1 = [be.g1_add(x if self.leader else be.g1_zero(), be.g1_neg(v[1])) for x, v in zip(x1, g1s)]
        d2 = [be.g2_add(x if self.leader else be.g2_zero(), be.g2_neg(v[1])) for x, v in zip(x2, g2s)]
        tf, t1, t2 = self._open_many(df, d1, d2)
        if any(np.any(t) for t in tf) or any(be.g1_serialize(t) != be.g1_serialize(be.g1_zero()) for t in t1) or \
                any(be.g2_serialize(t) != be.g2_serialize(be.g2_zero()) for t in t2):
            raise MacCheckError("SPDZ MAC check failed on a fused open")
        return xf, x1, x2

    def spdz_open_g1(self, v):
        be = self.be
        return self._spdz_open_g(v, be.g1_add, be.g1_neg, be.g1_zero, be.g1_serialize)

    def spdz_open_g2(self, v):
        be = self.be
        return self._spdz_open_g(v, be.g2_add, be.g2_neg, be.g2_zero, be.g2_serialize)

    def _minus_one(self):
        return self.be.fr_sub(np.zeros(4, dtype=np.uint64), self.be.fr_one())

    def spdz_scale_g1(self, s_pt, o_sc, lazy=False):
        """GroupShare::scale over SPDZ shares with DummyGroupTripleSource (x = 0, y = from_add_shared(leader?1:0), z = 0).
        lazy: as Party.scale_g1 (the opens here, the local arithmetic on a host thread, a future back)."""
        be = self.be
        y = be.fr_one() if self.leader else np.zeros(4, dtype=np.uint64)
        sx = self.spdz_open_g1(s_pt)                                              # x = 0
        oy = self.spdz_open_fr((be.fr_add(o_sc[0], y), be.fr_add(o_sc[1], y)))    # from_add_shared: mac = share (key 1)

        def finish():
            t = be.g1_neg(be.g1_mul(sx, y))                                       # - scale_pub_group(sx, y)
            if self.leader:
                t = be.g1_add(t, be.g1_mul(sx, oy))                               # shift: sh on the leader, mac += mac_share * G
            return (t, t)                                                         # both lanes hold the same value (key 1)
        return self._early(finish) if lazy else finish()

    def marlin_prove_shared_spdz(self, index, powers_g, z_share, randomness_share, challenge_fn, triple_fn=None) -> dict:
        """Marlin::prove over SPDZ shares (the `malicious` feature, BASELINE config 5 shape): the AHP rounds run on the
        share lane and on the MAC lane; the two lanes meet in the one Beaver multiplication of round 2 (SPDZ batch_mul:
        both opens MAC-checked) and in the MAC-checked opens of commitments, evaluations and the opening witness.
        z_share, randomness_share: (sh, mac) pairs; triple_fn(n) -> ((tx_sh, tx_mac), (ty..), (tz..)) or None (dummy)."""
        from . import marlin as DM
        be = self.be
        ctx = be.ctx
        m = DM.HostField.m
        lanes = (0, 1)
        shared = Party.SHARED_POLYS

        def commit_open(polys2):
            c = [DM.commit(ctx, powers_g, polys2[lane]) for lane in lanes]
            return {l: (self.spdz_open_g1((c[0][l], c[1][l])) if l in shared else c[0][l]) for l in c[0]}

        st = [DM.prover_init(index, z_share[lane], shared=True) for lane in lanes]
        r1 = [DM.prover_first_round(st[lane], randomness_share[lane]) for lane in lanes]
        comms = commit_open(r1)
        ch = dict(challenge_fn(1, comms))
        steps = [DM.second_round_steps(st[lane], ch["alpha"], ch["eta_a"], ch["eta_b"], ch["eta_c"]) for lane in lanes]
        req = [next(g) for g in steps]                                   # both lanes stop at z_A * z_B
        n_mul = req[0][4]
        self.spdz_beaver_batch_mul((req[0][1], req[1][1]), (req[0][2], req[1][2]), (req[0][3], req[1][3]), n_mul,
                                   triple_fn(n_mul) if triple_fn else None)
        req = [g.send(None) for g in steps]                              # ... and at the zero test of the sum over H
        opened = be.vec("marlin_open", req[0][2])
        self.spdz_open_vec((req[0][1], req[1][1]), opened, req[0][2])
        ok = be.is_zero_vec(opened, req[0][2])
        r2 = []
        for g in steps:
            try:
                g.send(ok)
                raise RuntimeError("second round did not finish")
            except StopIteration as done:
                r2.append(done.value)
        comms.update(commit_open(r2))
        ch.update(challenge_fn(2, comms))
        r3 = DM.prover_third_round(st[0], ch["beta"])                    # public values only: one lane suffices
        comms.update(DM.commit(ctx, powers_g, r3))
        ch.update(challenge_fn(3, comms))
        polys = [{**r1[lane], **r2[lane], **r3} for lane in lanes]
        ev = lambda lane, l, pt: ctx.poly_evaluate_dev(polys[lane][l].ptr, polys[lane][l].n, m(pt))
        evals = {l: self.spdz_open_fr((ev(0, l, ch["beta"]), ev(1, l, ch["beta"]))) for l in ("g_1", "z_b")}
        evals["t"], evals["g_2"] = ev(0, "t", ch["beta"]), ev(0, "g_2", ch["gamma"])
        # public polynomials enter a shared combination through shift(): on the leader, in both lanes (mac_share = 1 there)
        mine = lambda lane, l: polys[lane][l] if (l in shared or self.leader) else None
        ixp = index.polynomials()
        at_gamma = [polys[0]["g_2"], polys[0]["h_2"]] + [ixp[l] for l in sorted(ixp)]
        w = []
        for lane in lanes:
            at_beta = [mine(lane, l) for l in ("g_1", "z_b", "t", "mask_poly", "z_a", "w", "h_1")]
            w.append(DM.batch_open(ctx, powers_g, [(at_beta, ch["beta"])] + ([(at_gamma, ch["gamma"])] if lane == 0 else []), ch["xi"]))
        return {"commitments": comms, "evaluations": evals, "w_beta": self.spdz_open_g1((w[0][0], w[1][0])), "w_gamma": w[0][1],
                "challenges": ch}

    def create_proof_shared_spdz_native(self, pk, r1cs, z_share, r_share, s_share, triple=None) -> bytes:
        """create_proof over SPDZ shares as ONE library call (zk_groth16_prove_shared_spdz); arguments as create_proof_shared_spdz.
        A failed MAC check comes back as ZK_ERR_MAC and is raised as MacCheckError."""
        import ctypes as C
        from . import _lib
        ctx = self.ctx
        vt, errors, _keep = self._net_vtable()
        P2 = C.c_void_p * 2
        FR2 = _lib.Fr * 2
        lanes = lambda v: P2(int(v[0]), int(v[1]))
        frs = lambda v: FR2(*[_lib.Fr((C.c_uint64 * 4)(*[int(w) for w in np.asarray(x, dtype=np.uint64)])) for x in v])
        zl, rl, sl = lanes(z_share), frs(r_share), frs(s_share)
        t = [lanes((triple[k][0], triple[k][1])) for k in range(3)] if triple is not None else [None, None, None]
        out = np.zeros(192, dtype=np.uint8)
        sent = C.c_uint64(0)
        rc = ctx.lib.zk_groth16_prove_shared_spdz(ctx.h, pk.h, r1cs.h, zl, rl, sl, t[0], t[1], t[2],
                                                  C.byref(vt) if self.net.n > 1 else None, out.ctypes.data_as(C.c_void_p), C.byref(sent))
        if errors:
            raise errors[0]
        if rc == -5:
            raise MacCheckError((ctx.lib.zk_last_error(ctx.h) or b"").decode())
        ctx._ck(rc)
        self.bytes_sent += int(sent.value)
        return out.tobytes()

    def create_proof_shared_spdz(self, pk, r1cs, z_share, r_share, s_share, triple=None, fused=True) -> bytes:
        """create_proof with E = MpcPairingEngine<_, SpdzPairingShare> (the `malicious` feature).
        z_share = (sh, mac) device vectors; r_share, s_share = (sh, mac) scalars.
        fused: as create_proof_shared -- the opens of the three scale calls and of B in one collective, their MAC checks in a
        second one, then C and its check."""
        be = self.be
        D = be.domain_size(r1cs)
        P = self._pk_points(pk)
        early = {(k, lane): self._early(fn, P[pt], sc[lane]) for lane in (0, 1)       # see create_proof_shared
                 for k, fn, pt, sc in (("r_g1", be.g1_mul, "delta_g1", r_share), ("s_g1", be.g1_mul, "delta_g1", s_share),
                                       ("s_g2", be.g2_mul, "delta_g2", s_share))}
        lanes = []
        for lane in (0, 1):
            a, b, c = be.vec("wm_a%d" % lane, D), be.vec("wm_b%d" % lane, D), be.vec("wm_c%d" % lane, D)
            be.witness_map_pre(r1cs, z_share[lane], a, b, c)
            lanes.append((a, b, c))
        A = (lanes[0][0], lanes[1][0])
        B = (lanes[0][1], lanes[1][1])
        be.msms_begin(pk, r1cs, z_share[0])                        # the share lane's four MSMs over z run under the opens below
        self.spdz_beaver_batch_mul(A, B, A, D, triple)
        msm = []
        for lane in (0, 1):
            be.witness_map_post(r1cs, lanes[lane][0], lanes[lane][2])
            msm.append(be.msms(pk, r1cs, z_share[lane], lanes[lane][0]))       # 2 x 5 MSMs (spdz.rs:482-488)
        pair = lambda f: tuple(f(lane) for lane in (0, 1))
        pub1 = (lambda x: x) if self.leader else (lambda x: be.g1_zero())      # shift: leader's sh; mac += mac_share * G
        pub2 = (lambda x: x) if self.leader else (lambda x: be.g2_zero())
        add1 = lambda u, v: (be.g1_add(u[0], v[0]), be.g1_add(u[1], v[1]))
        neg1 = lambda u: (be.g1_neg(u[0]), be.g1_neg(u[1]))
        h_acc, l_acc, a_acc, b1_acc = [pair(lambda lane, k=k: msm[lane][0][k]) for k in range(4)]
        b2_acc = pair(lambda lane: msm[lane][1])
        r_g1 = pair(lambda lane: early[("r_g1", lane)].result())
        if fused:
            y = be.fr_one() if self.leader else np.zeros(4, dtype=np.uint64)
            g_a = pair(lambda lane: be.g1_add(be.g1_add(be.g1_add(r_g1[lane], pub1(P["a0"])), a_acc[lane]), pub1(P["alpha_g1"])))
            g1_b = pair(lambda lane: be.g1_add(be.g1_add(be.g1_add(early[("s_g1", lane)].result(), pub1(P["b0_g1"])), b1_acc[lane]), pub1(P["beta_g1"])))
            g2_b = pair(lambda lane: be.g2_add(be.g2_add(be.g2_add(early[("s_g2", lane)].result(), pub2(P["b0_g2"])), b2_acc[lane]), pub2(P["beta_g2"])))
            sy = (be.fr_add(s_share[0], y), be.fr_add(s_share[1], y))              # o + y, y = from_add_shared(leader ? 1 : 0)
            ry = (be.fr_add(r_share[0], y), be.fr_add(r_share[1], y))
            (oy_s, oy_r), (sx_rd, sx_a, sx_b), (B,) = self._spdz_open_many([sy, ry], [r_g1, g_a, g1_b], [g2_b])
            parts = [self._early(self._scale_finish, sx, oy, y) for sx, oy in ((sx_rd, oy_s), (sx_a, oy_s), (sx_b, oy_r))]
            both = lambda f: (lambda t: (t, t))(f.result())                        # scale: both lanes hold the same value (key 1)
            g_c = add1(add1(add1(add1(both(parts[1]), both(parts[2])), neg1(both(parts[0]))), l_acc), h_acc)
            Cp = self.spdz_open_g1(g_c)
            return be.g1_serialize(sx_a) + be.g2_serialize(B) + be.g1_serialize(Cp)
        r_s_delta = self.spdz_scale_g1(r_g1, s_share, lazy=True)
        g_a = pair(lambda lane: be.g1_add(be.g1_add(be.g1_add(r_g1[lane], pub1(P["a0"])), a_acc[lane]), pub1(P["alpha_g1"])))
        s_g_a = self.spdz_scale_g1(g_a, s_share, lazy=True)
        s_g1 = pair(lambda lane: early[("s_g1", lane)].result())
        g1_b = pair(lambda lane: be.g1_add(be.g1_add(be.g1_add(s_g1[lane], pub1(P["b0_g1"])), b1_acc[lane]), pub1(P["beta_g1"])))
        s_g2 = pair(lambda lane: early[("s_g2", lane)].result())
        g2_b = pair(lambda lane: be.g2_add(be.g2_add(be.g2_add(s_g2[lane], pub2(P["b0_g2"])), b2_acc[lane]), pub2(P["beta_g2"])))
        r_g1_b = self.spdz_scale_g1(g1_b, r_share, lazy=True)
        g_c = add1(add1(add1(add1(s_g_a.result(), r_g1_b.result()), neg1(r_s_delta.result())), l_acc), h_acc)
        Ap, Bp, Cp = self.spdz_open_g1(g_a), self.spdz_open_g2(g2_b), self.spdz_open_g1(g_c)   # SpdzGroupShare::reveal
        return be.g1_serialize(Ap) + be.g2_serialize(Bp) + be.g1_serialize(Cp)


    def marlin_prove_full_spdz(self, keys, z_share, zk_rng, triple_fn=None, mask_on_device=False):
        """The same over SPDZ shares (the `malicious` feature; BASELINE config 5's prover): z_share = (share, MAC) DevBufs, every
        open MAC-checked.  The MAC lane of this party's fresh randomness is the share itself (key alpha = 1 on the leader:
        sum of MAC shares = sum of shares), as the reference's from_add_shared does."""
        return _marlin_prove_full(self, keys, list(z_share), zk_rng, triple_fn, spdz=True, mask_on_device=mask_on_device)


    def marlin_prove_shared_spdz_native(self, keys, z_share, zk_rng, triple=None, mask_on_device=False) -> bytes:
        """The same over SPDZ shares (zk_marlin_prove_shared_spdz): z_share = (share, MAC) DevBufs, triple = ((x, x_mac), (y, y_mac),
        (z, z_mac)) device pointers or None.  A failed MAC check raises MacCheckError."""
        return _marlin_prove_native(self, keys, list(z_share), zk_rng, triple, mask_on_device)


def _marlin_prove_native(party, keys, z_lanes, zk_rng, triple, mask_on_device):
    import ctypes as C
    from . import marlin as DM
    ctx = party.be.ctx
    srs = keys.srs
    d, _keep_index = DM.native_index(keys)
    vt, errors, _keep = party._net_vtable()
    net = C.byref(vt) if party.net.n > 1 else None
    cap = ctx.lib.zk_marlin_proof_max_size()
    out = (C.c_uint8 * cap)()
    n, sent = C.c_size_t(), C.c_uint64(0)
    ptr = lambda v: int(getattr(v, "ptr", v))
    if len(z_lanes) == 1:
        t = [C.c_void_p(ptr(x)) for x in triple] if triple is not None else [None, None, None]
        rc = ctx.lib.zk_marlin_prove_shared(ctx.h, C.byref(d), srs.powers_g.h, srs.powers_gamma_g.h, C.c_void_p(ptr(z_lanes[0])), zk_rng.h,
                                            int(mask_on_device), t[0], t[1], t[2], net, out, cap, C.byref(n), C.byref(sent))
    else:
        P2 = C.c_void_p * 2
        lanes = lambda v: P2(ptr(v[0]), ptr(v[1]))
        t = [lanes(triple[k]) for k in range(3)] if triple is not None else [None, None, None]
        rc = ctx.lib.zk_marlin_prove_shared_spdz(ctx.h, C.byref(d), srs.powers_g.h, srs.powers_gamma_g.h, lanes(z_lanes), zk_rng.h,
                                                 int(mask_on_device), t[0], t[1], t[2], net, out, cap, C.byref(n), C.byref(sent))
    if errors:
        raise errors[0]
    if rc == -5:
        raise MacCheckError((ctx.lib.zk_last_error(ctx.h) or b"").decode())
    ctx._ck(rc)
    party.bytes_sent += int(sent.value)
    return bytes(out[:n.value])


def _marlin_prove_full(party, keys, z_lanes, zk_rng, triple_fn, spdz: bool, mask_on_device: bool = False):
    from . import convert as cv
    from . import marlin as DM
    from .api import Rng
    be = party.be
    ctx = be.ctx
    index, srs = keys.index, keys.srs
    m, ival = DM.HostField.m, DM.HostField.i
    R_MOD = DM.R_MOD
    lanes = range(len(z_lanes))
    shared = Party.SHARED_POLYS
    leader = party.leader
    import os as _os
    import time as _time
    _laps, _t = [], [_time.perf_counter()]

    def lap(name):                                      # ZK_MPC_TIMING=1: host wall-clock laps on stderr
        if _os.environ.get("ZK_MPC_TIMING"):
            ctx.sync()
            now = _time.perf_counter()
            _laps.append("%s %.1f" % (name, (now - _t[0]) * 1e3))
            _t[0] = now

    def open_g1(pts):                                   # pts: one projective array per lane
        return party.spdz_open_g1(tuple(pts)) if spdz else party.reveal_g1(pts[0])

    def open_fr(vals):                                  # vals: one (4,) Montgomery array per lane
        return ival(party.spdz_open_fr(tuple(vals)) if spdz else party._open_fr(vals[0]))

    # the public input is the instance part of the assignment: shared as from_public (the leader holds it), opened for the transcript
    ni = index.num_instance
    pub = []
    if ni > 1:
        tmp = be.vec("marlin_pub", ni)
        if spdz:
            party.spdz_open_vec((z_lanes[0].ptr, z_lanes[1].ptr), tmp, ni)
        else:
            be.open_vec(z_lanes[0].ptr, tmp, ni)
        pub = cv.fr_from_mont(ctx.download(tmp, (ni, 4)))[1:]
    fs = Rng.fiat_shamir(DM.PROTOCOL_NAME + keys.ivk_bytes() + b"".join(DM._fr_bytes(v) for v in pub))
    st = [DM.prover_init(index, z, shared=True) for z in z_lanes]
    polys = [dict(index.polynomials()) for _ in lanes]
    rands = {l: ([], None) for l in DM.INDEX_LABELS}
    comms = dict(keys.index_comms)
    ch = {}

    def commit_round(labels, round_polys):
        """Shares of the commitments on every lane under the same draws, then the reveal of the witness-dependent ones
        (`comms.publicize()`, lib.rs:180,205,228); public oracles commit alike on every party."""
        rr = DM._draw_round_randomness(keys, labels, zk_rng)
        # public oracles (t, g_2, h_2) are the same on every lane: committed once
        res = [DM._commit_round(keys, labels if lane == 0 else [l for l in labels if l in shared], round_polys[lane], zk_rng, rands=rr,
                                raw=True)[0] for lane in lanes]
        out = {}
        for l in labels:
            if l in shared:
                c = open_g1([res[lane][l]["comm"] for lane in lanes])
                sc = open_g1([res[lane][l]["shifted_comm"] for lane in lanes]) if res[0][l]["shifted_comm"] is not None else None
            else:
                c, sc = res[0][l]["comm"], res[0][l]["shifted_comm"]
            out[l] = DM.PcCommitment(c, sc)
        rands.update(rr)
        comms.update(out)
        fs.absorb(b"".join(out[l].to_bytes() for l in labels))

    # ---- round 1: every draw is this party's share of the prover's randomness
    md = DM.mask_poly_degree(index)
    if mask_on_device:
        # this party's share of the mask polynomial sampled on the device under a key from its rng (marlin.py::prove:
        # 3 |H| draws from a host ChaCha generator take 0.19 s at 2^20, more than the rest of the proof)
        rnd = ctx.alloc((3 + md + 1) * 32)
        head = ctx.upload(zk_rng.fill_fr(3))
        ctx.memcpy_d2d(rnd.ptr, head.ptr, 96)
        ctx.fr_random_dev(rnd.ptr + 96, md + 1, zk_rng.fill_bytes(32))
        ctx.sync()
    else:
        rnd = ctx.upload(zk_rng.fill_fr(3 + md + 1))
    lap("init+rng")
    r1 = [DM.prover_first_round(st[lane], rnd) for lane in lanes]
    for lane in lanes:
        polys[lane].update(r1[lane])
    lap("round1")
    commit_round(DM.ROUND_LABELS[0], r1)
    lap("commit1")
    ch["alpha"] = DM._sample_outside(index.dom_h, fs)
    ch["eta_a"], ch["eta_b"], ch["eta_c"] = ival(fs.next_fr()), ival(fs.next_fr()), ival(fs.next_fr())
    # ---- round 2: the lanes advance in lock-step around ONE Beaver product and one opened zero test
    steps = [DM.second_round_steps(st[lane], ch["alpha"], ch["eta_a"], ch["eta_b"], ch["eta_c"]) for lane in lanes]
    req = [next(g) for g in steps]
    n_mul = req[0][4]
    if spdz:
        party.spdz_beaver_batch_mul((req[0][1], req[1][1]), (req[0][2], req[1][2]), (req[0][3], req[1][3]), n_mul,
                                    triple_fn(n_mul) if triple_fn else None)
    else:
        party.beaver_batch_mul(req[0][1], req[0][2], req[0][3], n_mul, triple_fn(n_mul) if triple_fn else None)
    req = [g.send(None) for g in steps]
    opened = be.vec("marlin_open", req[0][2])
    if spdz:
        party.spdz_open_vec((req[0][1], req[1][1]), opened, req[0][2])
    else:
        be.open_vec(req[0][1], opened, req[0][2])
    ok = be.is_zero_vec(opened, req[0][2])
    r2 = []
    for g in steps:
        try:
            g.send(ok)
            raise RuntimeError("second round did not finish")
        except StopIteration as done:
            r2.append(done.value)
    for lane in lanes:
        polys[lane].update(r2[lane])
    lap("round2")
    commit_round(DM.ROUND_LABELS[1], r2)
    lap("commit2")
    ch["beta"] = DM._sample_outside(index.dom_h, fs)
    # ---- round 3: public values only
    r3 = DM.prover_third_round(st[0], ch["beta"])
    for lane in lanes:
        polys[lane].update(r3)
    lap("round3")
    commit_round(DM.ROUND_LABELS[2], [r3 for _ in lanes])
    lap("commit3")
    ch["gamma"] = ival(fs.next_fr())
    # ---- evaluations: shared oracles are evaluated on the shares and opened (`evaluations.publicize()`)
    ev = lambda lane, l, pt: ctx.poly_evaluate_dev(polys[lane][l].ptr, polys[lane][l].n, m(pt))
    single = {l: open_fr([ev(lane, l, ch["beta"]) for lane in lanes]) for l in ("z_b", "g_1")}
    single["t"], single["g_2"] = ival(ev(0, "t", ch["beta"])), ival(ev(0, "g_2", ch["gamma"]))
    ba = ch["beta"] * ch["alpha"] % R_MOD
    for mm in "abc":
        single[mm + "_denom"] = (ba - ch["alpha"] * ival(ev(0, mm + "_row", ch["gamma"])) - ch["beta"] * ival(ev(0, mm + "_col", ch["gamma"]))
                                 + ival(ev(0, mm + "_row_col", ch["gamma"]))) % R_MOD
    lcs = DM._linear_combinations(index, pub, ch, lambda l: single[l])
    evaluations = [single[l] for l in DM.EVAL_LABELS]
    fs.absorb(b"".join(DM._fr_bytes(e) for e in evaluations))
    xi = fs.next_u128() % R_MOD
    ch["xi"] = xi
    lap("evals")
    # ---- open_combinations on the shares: the witness of a share combination is a share of the witness; public polynomials
    # enter a shared combination through shift(), i.e. on the leader (in both lanes: mac_share = 1 there)
    point = {"beta": ch["beta"], "gamma": ch["gamma"]}
    pc_proof, keep = [], []
    for pl in ("beta", "gamma"):
        z = point[pl]
        terms, shifted, j = {}, [], 0
        r_comb, sr = [], []
        for label in DM.QUERY_SET[pl]:
            lc = [(c, l) for c, l in lcs[label] if l is not None]
            cj = pow(xi, j, R_MOD); j += 1
            for c, l in lc:
                terms[l] = (terms.get(l, 0) + c * cj) % R_MOD
                blind = rands[l][0]
                r_comb = [((r_comb[i] if i < len(r_comb) else 0) + (blind[i] if i < len(blind) else 0) * c % R_MOD * cj) % R_MOD
                          for i in range(max(len(r_comb), len(blind)))]
            if len(lcs[label]) == 1 and lc[0][1] in keys.bounds:
                src = lc[0][1]
                cj1 = pow(xi, j, R_MOD); j += 1
                shifted.append((src, cj1))
                sb = rands[src][1] or []
                sr = [((sr[i] if i < len(sr) else 0) + (sb[i] if i < len(sb) else 0) * cj1) % R_MOD for i in range(max(len(sr), len(sb)))]
        labels = list(terms)
        any_shared = any(l in shared for l in labels)
        hiding = any(len(rands[l][0]) > 0 for l in labels)
        wit = []
        for lane in (lanes if any_shared else [0]):
            mine = [polys[lane][l] if (l in shared or leader or not any_shared) else None for l in labels]
            comb = DM.linear_combination(ctx, mine, [terms[l] for l in labels])
            q = ctx.alloc(max(comb.n - 1, 1) * 32)
            ctx.poly_divide_by_linear_dev(comb.ptr, comb.n, m(z), q.ptr)
            keep += [comb, q]
            jobs = [(srs.powers_g, 0, q.ptr, comb.n - 1)]
            if hiding:
                rw = DM._host_divide_by_linear(r_comb, z)
                d = ctx.upload(cv.fr_to_mont(rw)); keep.append(d)
                jobs.append((srs.powers_gamma_g, 0, d.ptr, len(rw)))
            srw = []
            for src, cj1 in shifted:
                pp = polys[lane][src]
                if src in shared or leader or not any_shared:
                    wq = ctx.alloc(max(pp.n - 1, 1) * 32)
                    ctx.poly_divide_by_linear_dev(pp.ptr, pp.n, m(z), wq.ptr)
                    ctx.fr_vec_scale_dev(wq.ptr, m(cj1), wq.ptr, pp.n - 1)
                    keep.append(wq)
                    jobs.append((srs.powers_g, srs.max_degree - keys.bounds[src], wq.ptr, pp.n - 1))
                sb = rands[src][1] or []
                if sb:
                    w1 = DM._host_divide_by_linear(sb, z)
                    srw = [((srw[i] if i < len(srw) else 0) + w1[i] * cj1) % R_MOD for i in range(len(w1))]
            if srw:
                d = ctx.upload(cv.fr_to_mont(srw)); keep.append(d)
                jobs.append((srs.powers_gamma_g, 0, d.ptr, len(srw)))
            outs = ctx.msm_batch_dev(jobs)
            w = outs[0]
            for o in outs[1:]:
                w = ctx.g1_add(w, o)
            wit.append(w)
        rv = None
        if hiding:
            rv_share = DM._host_poly_eval(r_comb, z)
            if shifted:
                rv_share = (rv_share + DM._host_poly_eval(sr, z)) % R_MOD
            rv = open_fr([m(rv_share) for _ in lanes])      # the MAC lane of fresh local randomness is the share itself
        w = open_g1(wit) if any_shared else wit[0]
        pc_proof.append((w, rv))
    ctx.sync()
    lap("open")
    if _laps:
        import sys as _sys
        print("marlin_prove_full ms: " + " ".join(_laps), file=_sys.stderr)
    return DM.MarlinProof([[comms[l] for l in rnd_labels] for rnd_labels in DM.ROUND_LABELS], evaluations, pc_proof, ch)
