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

    def __init__(self, dist, device=None, open_pattern=None, group=None):
        """open_pattern: None = by party count (all-gather for two parties, all-to-all of slices for three or more),
        "allgather" or "a2a" to force one; a constructor argument, not an environment variable: every party of a run
        must make the same choice, or the parties wait for each other in different collectives.
        group: the process group that carries the collectives (None = the default group), e.g. an RCCL group beside a gloo
        default group that stays the control plane."""
        import torch
        self.torch = torch
        self.dist = dist
        if open_pattern not in (None, "allgather", "a2a"):
            raise ValueError("open_pattern must be None, 'allgather' or 'a2a'")
        self.open_pattern = open_pattern
        self.group = group
        self.rank = dist.get_rank(group)
        self.n = dist.get_world_size(group)
        self.device = device if device is not None else torch.device("cpu")

    def is_leader(self) -> bool:
        return self.rank == 0

    def new_buffer(self, nbytes: int):
        """A transport-visible buffer of nbytes (multiple of 8); returns (tensor, address)."""
        t = self.torch.empty(nbytes // 8, dtype=self.torch.int64, device=self.device)
        return t, t.data_ptr()

    def all_gather(self, send_tensor, recv_tensor):
        """recv = concat over parties (ordered by party id) of send  (MpcNet::broadcast_bytes)."""
        self.dist.all_gather_into_tensor(recv_tensor, send_tensor, group=self.group)
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
            self.dist.all_gather_into_tensor(recv, send[:words], group=self.group)
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
        self.dist.all_to_all_single(recv, send[:N * chunk * 4], group=self.group)          # recv[p] = slice `rank` of party p
        part = buffer("open_part", chunk * 32)
        self._sync()
        sum_parties(recv, N, chunk, part)
        full = buffer("open_full", N * chunk * 32)
        self.dist.all_gather_into_tensor(full, part, group=self.group)
        self._sync()
        return full[:words]

    def scatter(self, parts, nbytes: int):
        """The leader ("king") hands parts[p] to party p; everybody returns its own part
        (MpcNet::worker_receive_or_leader_send_element, used by king_share: mpc-algebra/src/share/additive.rs:98-107).
        parts: list of N int64 tensors of nbytes / 8 words on the leader, None elsewhere."""
        recv = self.torch.empty(nbytes // 8, dtype=self.torch.int64, device=self.device)
        self.dist.scatter(recv, scatter_list=[p.reshape(-1) for p in parts] if self.rank == 0 else None, src=0, group=self.group)
        self._sync()
        return recv

    def _sync(self):
        if self.device.type == "cuda":
            self.torch.cuda.current_stream().synchronize()

    def all_gather_small(self, arr: np.ndarray) -> list:
        t = self.torch.from_numpy(np.ascontiguousarray(arr).view(np.int64).reshape(-1).copy()).to(self.device)
        out = self.torch.empty(self.n * t.numel(), dtype=self.torch.int64, device=self.device)
        self.dist.all_gather_into_tensor(out, t, group=self.group)
        res = out.cpu().numpy().view(arr.dtype).reshape((self.n,) + arr.shape)
        return [res[p] for p in range(self.n)]

    def barrier(self):
        self.dist.barrier(group=self.group)


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
        Same opened values and the same 192 bytes as tests/pyseq/mpc_seq.py::Party.create_proof_shared(fused=True), the Python
        sequence of the same calls that the tests compare it with."""
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

    def marlin_prove_shared_native(self, keys, z_share, zk_rng, triple=None, mask_on_device=False) -> bytes:
        """MpcMarlin::prove over additive shares as ONE library call (zk_marlin_prove_shared): what a Rust host would bind.
        Arguments as marlin_prove_full (triple: three device pointers of |MUL| Beaver shares, or None for dummy triples);
        returns Proof::serialize's bytes -- the same bytes as tests/pyseq/mpc_seq.py::Party.marlin_prove_full(...).serialize()."""
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
