"""Host-side mirror of GPvecchia's R API for the U_NZentries path, on top of the
C ABI of libgpvecchia_hip.so (R is not installed on either box, so this Python
layer plays the part of the R wrappers; INTEGRATION.md shows the R binding).

Same names and argument meaning as the reference:
    vecchia_specify()    R/vecchia_specify.R:29-240
    createU()            R/createU.R:65-201
    vecchia_likelihood() R/vecchia_likelihood.R:14-27
    U_NZentries()        R/RcppExports.R:22-24   (literal C-ABI drop-in)
    U_NZentries_mat()    R/RcppExports.R:26-28
    MaternFun()/EsqeFun  NAMESPACE:3, R/RcppExports.R
All arithmetic of the hot path runs in the HIP library; nothing here falls back
to NumPy for it.
"""
from __future__ import annotations

import ctypes as C
import warnings

import numpy as np

from . import _lib as L
from . import specify as S
from ._lib import GPV_WANT_DENOM, GPV_WANT_LOGLIK_Z, GPV_WANT_NUMERATOR, GPV_WANT_U, NSUMS, GpvError


# ---------------------------------------------------------------------------
# device plan
# ---------------------------------------------------------------------------
class Plan:
    """Device-resident image of a vecchia.approx object (gpv_plan)."""

    def __init__(self, locsord, revNNarray, revCond, device=0, row_begin=0, row_end=None):
        locs = np.asfortranarray(locsord, dtype=np.float64)
        self.Nlocs, self.dim = locs.shape
        nn = L.as_r_int_matrix(revNNarray)
        cd = _cond_to_r(revCond)
        self.p = nn.shape[1]
        self.row_begin = int(row_begin)
        self.row_end = int(self.Nlocs if row_end is None else row_end)
        self.rows = self.row_end - self.row_begin
        self._h = C.c_void_p()
        st = L.lib().gpv_plan_create(C.byref(self._h), int(device), self.Nlocs, self.dim, self.p, L.dptr(locs),
                                     L.iptr(nn), L.iptr(cd), self.row_begin, self.row_end)
        L.check(st, "gpv_plan_create")
        self.device = device
        self._nn, self._cd = nn, cd          # kept for build_posterior (the R-layout arrays of this plan)
        self.has_posterior = False

    def build_posterior(self):
        """Structure of the U2V pass (R/vecchia_prediction.R:62-83) for GPV_WANT_DENOM evaluations."""
        self.has_posterior = False                       # the library drops any earlier structure on entry: true only on success
        L.check(L.lib().gpv_plan_build_posterior(self._h, L.iptr(self._nn), L.iptr(self._cd)), "gpv_plan_build_posterior")
        self.has_posterior = True
        nl = C.c_int()
        L.check(L.lib().gpv_plan_posterior_levels(self._h, C.byref(nl)), "gpv_plan_posterior_levels")
        return int(nl.value)

    def ensure_posterior(self):
        """True when the structure of the device posterior pass exists (built now if it did not); False when the library
        refuses it for this plan (GPV_ERR_UNSUPPORTED_M: some conditioning set has more than 64 LATENT entries; the row
        length m + 1 itself may exceed 64): the caller then takes the host route, like the reference's CHOLMOD."""
        if self.has_posterior:
            return True
        if getattr(self, "_post_refused", False):
            return False
        self.has_posterior = False                       # the library drops any earlier structure on entry: true only on success
        st = L.lib().gpv_plan_build_posterior(self._h, L.iptr(self._nn), L.iptr(self._cd))
        if st == 5:
            self._post_refused = True
            return False
        L.check(st, "gpv_plan_build_posterior")
        self.has_posterior = True
        return True

    def build_posterior_fill(self, max_fill=4.0):
        """Structure of the U2V pass on the FILLED pattern (cond.yz='y': the factor fills in, R/vecchia_prediction.R:72-83).
        Returns the fill ratio, or None when the library refuses (fill beyond max_fill x the latent block, or a column of the
        factor beyond 64 rows): the caller then factorises on the host like the reference's CHOLMOD."""
        if getattr(self, "_fill_refused", False):
            return None
        ratio = C.c_double(0.0)
        self.has_posterior = False                       # (as in ensure_posterior: a refused or failed rebuild leaves none)
        st = L.lib().gpv_plan_build_posterior_fill(self._h, L.iptr(self._nn), L.iptr(self._cd), float(max_fill), C.byref(ratio))
        if st == 5:                                       # GPV_ERR_UNSUPPORTED_M: bounded out
            self._fill_refused = True
            self.fill_ratio = float(ratio.value)
            return None
        L.check(st, "gpv_plan_build_posterior_fill")
        self.has_posterior = True
        self.fill_ratio = float(ratio.value)
        return self.fill_ratio

    def set_observed(self, obs_ord):
        """vecchia.approx$obs in ordered layout (True = the location carries an observation); None = all observed.  With
        unobserved locations the posterior pass drops 1/tau and z/tau there (U2V + vecchia_mean for plans with prediction
        locations, R/vecchia_prediction.R:62-126); evaluate with per-location nuggets."""
        if obs_ord is None:
            L.check(L.lib().gpv_plan_set_observed(self._h, None), "gpv_plan_set_observed")
            return
        o = np.ascontiguousarray(np.asarray(obs_ord, dtype=bool), dtype=np.int32)
        if o.size != self.Nlocs:
            raise ValueError("obs must have one entry per ordered location")
        L.check(L.lib().gpv_plan_set_observed(self._h, L.iptr(o)), "gpv_plan_set_observed")

    def posterior_levels(self):
        """Number of levels of the posterior pass's schedule (after build_posterior)."""
        nl = C.c_int()
        L.check(L.lib().gpv_plan_posterior_levels(self._h, C.byref(nl)), "gpv_plan_posterior_levels")
        return int(nl.value)

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                L.lib().gpv_plan_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def set_data(self, z_ord):
        z = np.ascontiguousarray(z_ord, dtype=np.float64)
        if z.shape[0] != self.Nlocs:
            raise ValueError("z_ord must have one entry per ordered location")
        self._user_z = None                              # whatever set_user_data remembered is no longer on the device
        L.check(L.lib().gpv_plan_set_data(self._h, L.dptr(z)), "gpv_plan_set_data")

    def set_user_data(self, z, ord_z):
        """set_data(z[ord_z - 1]) unless exactly this z is the data already on the device: an optimiser evaluates the
        likelihood of ONE data vector hundreds of times (R/vecchia_wrappers.R:72-93), and reordering and uploading 8 MB
        per call costs more than the evaluation itself at n = 1e6.  The comparison is by content, against a private copy."""
        prev = getattr(self, "_user_z", None)
        if prev is not None and prev.shape == z.shape and np.array_equal(prev, z):
            return False
        self.set_data(z[ord_z - 1])
        self._user_z = np.array(z, dtype=np.float64, copy=True)
        return True

    def invalidate_data(self):
        """Called by code that changes the plan's data behind set_data (the Vecchia-Laplace device loop)."""
        self._user_z = None

    def eval(self, covmodel, covparms, nuggets, flags, stream=None, d_sums_out=None):
        cp = np.ascontiguousarray(covparms, dtype=np.float64)
        ng = np.ascontiguousarray(np.atleast_1d(nuggets), dtype=np.float64)
        st = L.lib().gpv_plan_eval(self._h, covmodel.encode() if isinstance(covmodel, str) else bytes(covmodel),
                                   cp.ctypes.data, int(cp.size), ng.ctypes.data, int(ng.size), int(flags), stream or 0,
                                   d_sums_out or 0)
        if st:
            L.check(st, "gpv_plan_eval")

    def sums(self):
        s = np.zeros(NSUMS)
        L.check(L.lib().gpv_plan_get_sums(self._h, L.dptr(s)), "gpv_plan_get_sums")
        return s

    def Lentries(self):
        out = np.zeros((self.rows, self.p), dtype=np.float64, order="F")
        L.check(L.lib().gpv_plan_get_Lentries(self._h, L.dptr(out)), "gpv_plan_get_Lentries")
        return out

    def Zentries(self):
        out = np.zeros(2 * self.rows, dtype=np.float64)
        L.check(L.lib().gpv_plan_get_Zentries(self._h, L.dptr(out)), "gpv_plan_get_Zentries")
        return out

    def Lentries_device(self):
        ptr, ld = C.c_void_p(), C.c_int64()
        L.check(L.lib().gpv_plan_Lentries_device(self._h, C.byref(ptr), C.byref(ld)), "gpv_plan_Lentries_device")
        return ptr.value, int(ld.value)

    def posterior_mean(self):
        """mu.ord of R/vecchia_prediction.R:118-126 after an eval with GPV_WANT_MEAN."""
        out = np.zeros(self.Nlocs)
        L.check(L.lib().gpv_plan_get_posterior_mean(self._h, L.dptr(out)), "gpv_plan_get_posterior_mean")
        return out

    kernel_timing = True

    def set_kernel_timing(self, on):
        L.check(L.lib().gpv_plan_set_kernel_timing(self._h, int(bool(on))), "gpv_plan_set_kernel_timing")
        self.kernel_timing = bool(on)

    def last_kernel_ms(self):
        ms = C.c_double()
        L.check(L.lib().gpv_plan_last_kernel_ms(self._h, C.byref(ms)), "gpv_plan_last_kernel_ms")
        return float(ms.value)

    def set_comm(self, comm):
        """Attach (or, with None, detach) a Comm: every eval() then all-reduces its 8 sums over the ranks, inside the library."""
        L.check(L.lib().gpv_plan_set_comm(self._h, comm._h if comm is not None else None), "gpv_plan_set_comm")
        self._comm = comm                                     # keeps the communicator alive as long as the plan uses it


class Comm:
    """gpv_comm: an RCCL communicator owned by the library (one process per GPU).  `exchange(id_bytes_or_None) -> id_bytes`
    carries rank 0's 128-byte id to the other ranks (from_torch() does it over an initialised torch.distributed group)."""

    def __init__(self, device, rank, world, exchange=None, ident=None):
        """exchange(id_bytes_or_None) -> id_bytes carries rank 0's id to the others; or `ident`: the 128 bytes every rank
        already holds (Comm.exchange_id in a first step: negotiate_comm does the exchange on the caller's thread and only the
        collective gpv_comm_create on a helper thread)."""
        raw = ident if ident is not None else Comm.exchange_id(rank, exchange)
        if not isinstance(raw, (bytes, bytearray)) or len(raw) != 128:
            raise ValueError("Comm: the exchange must hand every rank the 128 bytes of rank 0")
        ident = C.create_string_buffer(bytes(raw), 128)
        self._h = C.c_void_p()
        self.device, self.rank, self.world = int(device), int(rank), int(world)
        L.check(L.lib().gpv_comm_create(C.byref(self._h), self.device, self.rank, self.world, ident), "gpv_comm_create")

    @staticmethod
    def exchange_id(rank, exchange):
        """Rank 0 makes the communicator id and `exchange` hands it to every rank.  Rank 0 ALWAYS takes part in the exchange:
        if it could not make the id it hands out an empty one and every rank raises here, instead of rank 0 raising alone
        while the others wait for a broadcast that never comes."""
        ident = C.create_string_buffer(128)
        st0 = L.lib().gpv_comm_unique_id(ident) if rank == 0 else 0
        raw = exchange((bytes(ident.raw) if st0 == 0 else b"") if rank == 0 else None)
        if rank == 0:
            L.check(st0, "gpv_comm_unique_id")
        if raw == b"":
            raise RuntimeError("Comm: rank 0 could not produce the communicator id (no RCCL to bind?)")
        return raw

    @staticmethod
    def torch_exchange(group=None):
        """The exchange over an initialised torch.distributed group (call it on the thread that owns the rank's GPU: with
        the nccl backend the broadcast moves through torch.cuda.current_device() of the CALLING thread)."""
        import torch.distributed as dist

        def exchange(mine):
            box = [mine]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            return box[0]
        return exchange

    @classmethod
    def from_torch(cls, device, group=None):
        """Collective over the torch group.  Raises on EVERY rank alike when rank 0 cannot produce the id; for a route that
        is agreed between the ranks (and a fallback when the library's communicator is unavailable on any of them) use
        gpvecchia_amd.distributed.negotiate_comm."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)

        def exchange(mine):
            box = [mine]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            return box[0]
        return cls(device, rank, world, exchange)

    @staticmethod
    def rccl_version():
        """ncclGetVersion() of the RCCL the library bound at run time; 0 when there is none (gpv_comm_* then return GPV_ERR_STATE)."""
        return int(L.lib().gpv_rccl_version())

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                L.lib().gpv_comm_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass


class MultiPlan:
    """gpv_mplan: the same vecchia.approx spread over several GPUs of one host process (row shards)."""

    def __init__(self, locsord, revNNarray, revCond, devices):
        locs = np.asfortranarray(locsord, dtype=np.float64)
        self.Nlocs, self.dim = locs.shape
        nn = L.as_r_int_matrix(revNNarray)
        cd = _cond_to_r(revCond)
        self.p = nn.shape[1]
        dev = np.ascontiguousarray(devices, dtype=np.int32)
        self._h = C.c_void_p()
        L.check(L.lib().gpv_mplan_create(C.byref(self._h), L.iptr(dev), int(dev.size), self.Nlocs, self.dim, self.p,
                                         L.dptr(locs), L.iptr(nn), L.iptr(cd)), "gpv_mplan_create")

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                L.lib().gpv_mplan_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def set_data(self, z_ord):
        z = np.ascontiguousarray(z_ord, dtype=np.float64)
        L.check(L.lib().gpv_mplan_set_data(self._h, L.dptr(z)), "gpv_mplan_set_data")

    def eval(self, covmodel, covparms, nuggets, flags):
        cp = np.ascontiguousarray(covparms, dtype=np.float64)
        ng = np.ascontiguousarray(np.atleast_1d(nuggets), dtype=np.float64)
        s = np.zeros(NSUMS)
        L.check(L.lib().gpv_mplan_eval(self._h, str(covmodel).encode(), L.dptr(cp), int(cp.size), L.dptr(ng),
                                       int(ng.size), int(flags), L.dptr(s)), "gpv_mplan_eval")
        return s

    def Lentries(self):
        out = np.zeros((self.Nlocs, self.p), dtype=np.float64, order="F")
        L.check(L.lib().gpv_mplan_get_Lentries(self._h, L.dptr(out)), "gpv_mplan_get_Lentries")
        return out


class ReplicaPlans:
    """gpv_mplan in REPLICA mode: one complete plan per listed device, each evaluating its own parameter vector (or its
    own data set), all in flight together.  This is what several GPUs mean for the parts of the path that do not shard:
    the posterior pass of cond.yz='SGV' and every Vecchia-Laplace Newton step (BASELINE.json configs[4])."""

    def __init__(self, locsord, revNNarray, revCond, devices):
        locs = np.asfortranarray(locsord, dtype=np.float64)
        self.Nlocs, self.dim = locs.shape
        self._nn = L.as_r_int_matrix(revNNarray)
        self._cd = _cond_to_r(revCond)
        self.p = self._nn.shape[1]
        dev = np.ascontiguousarray(devices, dtype=np.int32)
        self.count = int(dev.size)
        self._h = C.c_void_p()
        L.check(L.lib().gpv_mplan_create_replicas(C.byref(self._h), L.iptr(dev), self.count, self.Nlocs, self.dim, self.p,
                                                  L.dptr(locs), L.iptr(self._nn), L.iptr(self._cd)), "gpv_mplan_create_replicas")
        self.has_posterior = False

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                L.lib().gpv_mplan_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def set_data(self, z_ord, replica=None):
        z = np.ascontiguousarray(z_ord, dtype=np.float64)
        if z.shape[0] != self.Nlocs:
            raise ValueError("z_ord must have one entry per ordered location")
        if replica is None:
            L.check(L.lib().gpv_mplan_set_data(self._h, L.dptr(z)), "gpv_mplan_set_data")
        else:
            L.check(L.lib().gpv_mplan_set_data_one(self._h, int(replica), L.dptr(z)), "gpv_mplan_set_data_one")

    def build_posterior(self):
        self.has_posterior = False
        L.check(L.lib().gpv_mplan_build_posterior(self._h, L.iptr(self._nn), L.iptr(self._cd)), "gpv_mplan_build_posterior")
        self.has_posterior = True

    def eval_each(self, covmodel, covparms, nuggets, flags):
        """covparms: (count, ncovparms), nuggets: (count,) constant nugget of each replica -> sums (count, NSUMS)."""
        cp = np.ascontiguousarray(covparms, dtype=np.float64).reshape(self.count, -1)
        ng = np.ascontiguousarray(np.broadcast_to(np.asarray(nuggets, dtype=np.float64), (self.count,)))
        s = np.zeros((self.count, NSUMS))
        L.check(L.lib().gpv_mplan_eval_each(self._h, str(covmodel).encode(), L.dptr(cp), int(cp.shape[1]), L.dptr(ng),
                                            int(flags), L.dptr(s)), "gpv_mplan_eval_each")
        return s

    def logliks(self, covmodel, covparms, nuggets, cond_yz="SGV"):
        """vecchia_likelihood of every replica's parameter vector: cond.yz='z' fused, 'SGV' with the posterior pass."""
        if cond_yz == "z":
            s = self.eval_each(covmodel, covparms, nuggets, GPV_WANT_LOGLIK_Z)
            return np.array([loglik_z_from_sums(r, self.Nlocs) for r in s])
        if not self.has_posterior:
            self.build_posterior()
        s = self.eval_each(covmodel, covparms, nuggets, GPV_WANT_DENOM)
        return np.array([loglik_from_sums(r, self.Nlocs) for r in s])


_R_LAYOUT_CACHE = []          # [(weakref(revNNarray), weakref(revCond), fingerprint, nn_r, cd_r)] of the last two index-array pairs


def _fingerprint(a):
    """Content fingerprint of an index array: shape, dtype, strides and the library's 128-bit hash of EVERY byte
    (gpv_hash_bytes: the multi-threaded hash the plan cache of the literal drop-in keys on, ~3-4 ms for the two arrays of
    n = 1e6, m = 30; no optional module, no 32-bit fallback).  An in-place edit of the array between two calls changes it.
    None for an empty array (nothing worth caching)."""
    a = np.asarray(a)
    if a.size == 0:
        return None
    buf = a if a.flags.c_contiguous or a.flags.f_contiguous else np.ascontiguousarray(a)
    out = (C.c_uint64 * 2)()
    L.check(L.lib().gpv_hash_bytes(C.c_void_p(buf.ctypes.data), buf.nbytes, 0x6770765F6670, out), "gpv_hash_bytes")
    return (a.shape, a.dtype.str, a.strides, int(out[0]), int(out[1]))


def _r_layout_cached(revNNarray, revCond):
    """Column-major int32 copies (R's representation) of the two index arrays, kept for the arrays last seen: createU hands
    the SAME objects to U_NZentries at every optimiser step, and converting 2 x 31e6 entries costs ~60 ms, six times the
    library call.  A hit needs the same two objects (held by weak reference: the cache keeps no caller array alive) AND an
    unchanged hash of their whole content, so an array edited in place between two calls is converted again."""
    import weakref
    fp = (_fingerprint(revNNarray), _fingerprint(revCond))
    if fp[0] is None or fp[1] is None:                                 # empty arrays: convert, do not cache
        return L.as_r_int_matrix(revNNarray), _cond_to_r(revCond)
    for wa, wb, f, nn_r, cd_r in _R_LAYOUT_CACHE:
        if wa() is revNNarray and wb() is revCond and f == fp:
            return nn_r, cd_r
    nn_r, cd_r = L.as_r_int_matrix(revNNarray), _cond_to_r(revCond)
    try:
        _R_LAYOUT_CACHE.insert(0, (weakref.ref(revNNarray), weakref.ref(revCond), fp, nn_r, cd_r))
    except TypeError:                                                  # not weakly referenceable (a list, ...): do not cache
        return nn_r, cd_r
    del _R_LAYOUT_CACHE[2:]
    return nn_r, cd_r


def _cond_to_r(revCond):
    """logical matrix -> R's int representation: NaN (float input) or -1 (int8 input) -> NA_INTEGER."""
    rc = np.asarray(revCond)
    if rc.dtype.kind == "f":
        return L.as_r_int_matrix(rc)
    out = np.asarray(rc, dtype=np.int32, order="F")                  # one pass: cast and transpose together
    if out is rc or np.shares_memory(out, rc):
        out = out.copy(order="F")
    out[out < 0] = L.NA_INTEGER
    return out


def loglik_z_from_sums(sums, n):
    out = C.c_double()
    s = np.ascontiguousarray(sums, dtype=np.float64)
    L.check(L.lib().gpv_loglik_z_from_sums(L.dptr(s), int(n), C.byref(out)), "gpv_loglik_z_from_sums")
    return float(out.value)


def loglik_from_sums(sums, n):
    """R/vecchia_likelihood.R:95-96 from sums evaluated with GPV_WANT_DENOM."""
    out = C.c_double()
    s = np.ascontiguousarray(sums, dtype=np.float64)
    L.check(L.lib().gpv_loglik_from_sums(L.dptr(s), int(n), C.byref(out)), "gpv_loglik_from_sums")
    return float(out.value)


def numerator_from_sums(sums):
    a, b = C.c_double(), C.c_double()
    s = np.ascontiguousarray(sums, dtype=np.float64)
    L.check(L.lib().gpv_numerator_from_sums(L.dptr(s), C.byref(a), C.byref(b)), "gpv_numerator_from_sums")
    return float(a.value), float(b.value)


# ---------------------------------------------------------------------------
# literal drop-ins (R/RcppExports.R)
# ---------------------------------------------------------------------------
def U_NZentries(Ncores, n, locs, revNNarray, revCondOnLatent, nuggets, nuggets_obsord, covType, covparms):
    """R/RcppExports.R:22-24: returns dict(Lentries=(Nlocs, m+1), Zentries=(2n,)) (+ n_failed).
    Goes through the C symbol gpv_U_NZentries exactly as the R .C() binding would."""
    locs = np.asfortranarray(locs, dtype=np.float64)
    Nlocs, dim = locs.shape
    nn, cd = _r_layout_cached(revNNarray, revCondOnLatent)
    p = nn.shape[1]
    nug = np.ascontiguousarray(nuggets, dtype=np.float64)
    nugo = np.ascontiguousarray(nuggets_obsord, dtype=np.float64)
    cp = np.ascontiguousarray(covparms, dtype=np.float64)
    Lent = np.empty((Nlocs, p), dtype=np.float64, order="F")          # every entry is written by the library
    Z = np.zeros(2 * int(n), dtype=np.float64)
    ci = lambda v: C.byref(C.c_int(int(v)))
    nfail, status = C.c_int(0), C.c_int(0)
    ct = C.c_char_p(str(covType).encode())
    L.lib().gpv_U_NZentries(ci(Ncores), ci(n), ci(Nlocs), ci(dim), ci(p), L.dptr(locs), L.iptr(nn), L.iptr(cd),
                            L.dptr(nug), L.dptr(nugo), C.byref(ct), L.dptr(cp), ci(cp.size), L.dptr(Lent), L.dptr(Z),
                            C.byref(nfail), C.byref(status))
    L.check(status.value, "gpv_U_NZentries")
    return dict(Lentries=Lent, Zentries=Z, n_failed=int(nfail.value))


def U_NZentries_mat(Ncores, n, locs, revNNarray, revCondOnLatent, nuggets, nuggets_obsord, covVals, covparms):
    """R/RcppExports.R:26-28 (locs/revCond/nuggets/covparms are unused by the reference body)."""
    nn = L.as_r_int_matrix(revNNarray)
    Nlocs, p = nn.shape
    nugo = np.ascontiguousarray(nuggets_obsord, dtype=np.float64)
    cv = np.asfortranarray(covVals, dtype=np.float64)
    if cv.shape != (Nlocs, Nlocs):
        raise ValueError("covVals must be Nlocs x Nlocs")
    Lent = np.zeros((Nlocs, p), dtype=np.float64, order="F")
    Z = np.zeros(2 * int(n), dtype=np.float64)
    ci = lambda v: C.byref(C.c_int(int(v)))
    nfail, status = C.c_int(0), C.c_int(0)
    L.lib().gpv_U_NZentries_mat(ci(Ncores), ci(n), ci(Nlocs), ci(p), L.iptr(nn), L.dptr(nugo), L.dptr(cv),
                                L.dptr(Lent), L.dptr(Z), C.byref(nfail), C.byref(status))
    L.check(status.value, "gpv_U_NZentries_mat")
    return dict(Lentries=Lent, Zentries=Z, n_failed=int(nfail.value))


def _covfun(name, distmat, covparms):
    d = np.ascontiguousarray(distmat, dtype=np.float64)
    cp = np.ascontiguousarray(covparms, dtype=np.float64)
    out = np.empty_like(d)
    status = C.c_int(0)
    getattr(L.lib(), name)(L.dptr(d), C.byref(C.c_int(d.size)), L.dptr(cp), L.dptr(out), C.byref(status))
    L.check(status.value, name)
    return out


def MaternFun(distmat, covparms):
    """src/Matern.cpp:24-86 (closed-form smoothness values)."""
    return _covfun("gpv_MaternFun", distmat, covparms)


def EsqeFun(distmat, covparms):
    """src/Esqe.cpp:17-39."""
    return _covfun("gpv_EsqeFun", distmat, covparms)


# ---------------------------------------------------------------------------
# vecchia_specify — R/vecchia_specify.R:29-240
# ---------------------------------------------------------------------------
def vecchia_specify(locs, m=-1, ordering=None, cond_yz=None, locs_pred=None, ordering_pred=None, pred_cond=None,
                    conditioning=None, mra_options=None, ic0=False, verbose=False, NNarray=None, nn_backend="auto"):
    """Parameter-independent specification of the Vecchia approximation, conditioning='NN':
    ordering in {'coord','maxmin','outsidein','none'}; cond.yz in {'SGV','SGVT','y','z','zy','RVP','LK'}; prediction
    locations with ordering.pred in {'obspred','general'} and pred.cond in {'general','independent'}; the m = 0
    independent case.  MRA / 'firstm' conditioning goes through ic0 in the reference, never through U_NZentries
    (R/createU.R:89): not built (NotImplementedError)."""
    locs = np.asarray(locs, dtype=np.float64)
    if locs.ndim != 2:
        warnings.warn("Locations must be in matrix form")                    # :32-35
        return None
    if m is None or m == -1:
        raise ValueError("neither m nor r defined!")                        # :36-40
    if conditioning not in (None, "NN"):
        raise NotImplementedError("conditioning='mra'/'firstm' goes through ic0, not U_NZentries (R/createU.R:89)")
    spatial_dim = locs.shape[1]
    n = locs.shape[0]
    have_pred = locs_pred is not None
    if have_pred:                                                            # :46-51
        locs_pred = np.asarray(locs_pred, dtype=np.float64).reshape(-1, spatial_dim)
        la = np.vstack([locs, locs_pred])
        if np.unique(la, axis=0).shape[0] < la.shape[0]:
            raise ValueError("Prediction locations contain observed location(s), remove redundancies.")
    if m > n:                                                                # :53-56
        warnings.warn("Conditioning set size m chosen to be larger than n. Changing to m=n-1")
        m = n - 1
    if m == 0:                                                               # :59-73
        if have_pred:
            warnings.warn("Attempting to make predictions with m=0.  Prediction ignored")
        ord_ = np.arange(1, n + 1)
        NN = np.stack([ord_, np.zeros(n, dtype=np.int64)], axis=1).astype(np.int32)
        Cond = np.stack([np.ones(n), -np.ones(n)], axis=1).astype(np.int8)
        obs = np.ones(n, dtype=bool)
        U_prep = S.U_sparsity(locs, NN, obs, Cond)
        return dict(locsord=locs.copy(), obs=obs, ord=ord_, ord_z=ord_.copy(), ord_pred="general", U_prep=U_prep,
                    cond_yz="false", conditioning="NN", ic0=False)
    if ordering is None:                                                     # :83-85
        ordering = "coord" if spatial_dim == 1 else "maxmin"
    if pred_cond is None:                                                    # :86
        pred_cond = "general"
    if cond_yz is None:                                                      # :92-96
        cond_yz = "SGV" if (not have_pred or spatial_dim == 1) else "zy"
    if ordering not in ("coord", "maxmin", "outsidein", "none"):
        raise ValueError(f"ordering='{ordering}' not defined")
    n_p = 0
    if not have_pred:                                                        # :100-117
        if ordering == "coord":                                              # :102
            ord_ = S.order_coordinate(locs)
        elif ordering == "maxmin":                                           # :103-106
            o = S.order_maxmin_exact(locs)
            cut = min(n, 9)
            ord_ = np.concatenate([o[:1], o[cut:], o[1:cut]])
        elif ordering == "outsidein":                                        # :107-108
            ord_ = S.order_outsidein(locs)
        else:                                                                # :109-110
            ord_ = np.arange(1, n + 1)
        ord_z = ord_.copy()
        locsord = locs[ord_ - 1]
        obs = np.ones(n, dtype=bool)
        ordering_pred = "general"
    else:                                                                    # :119-149 prediction is desired
        n_p = locs_pred.shape[0]
        locs_all = np.vstack([locs, locs_pred])
        observed_obspred = np.concatenate([np.ones(n, bool), np.zeros(n_p, bool)])
        if ordering_pred is None:                                            # :124-126
            ordering_pred = "general" if (spatial_dim == 1 and ordering == "coord") else "obspred"
        if ordering_pred == "general":                                       # :127-131
            ord_ = S.order_coordinate(locs_all) if ordering == "coord" else S.order_maxmin_exact(locs_all)
            ord_obs = ord_[ord_ <= n]
        else:                                                                # :132-145
            if ordering == "coord":
                ord_obs, ord_pr = S.order_coordinate(locs), S.order_coordinate(locs_pred)
            elif ordering == "none":
                ord_obs, ord_pr = np.arange(1, n + 1), np.arange(1, n_p + 1)
            else:
                ord_obs, ord_pr = S.order_maxmin_exact_obs_pred(locs, locs_pred)
            ord_ = np.concatenate([ord_obs, ord_pr + n])
        ord_z = ord_obs
        locsord = locs_all[ord_ - 1]
        obs = observed_obspred[ord_ - 1]
    nall = locsord.shape[0]
    if NNarray is None:                                                      # :157-159
        # both searches implement the same exact definition and return identical arrays (tests); "gpu" is the
        # library's brute-force kernel, "host" the cKDTree search
        use_gpu = (nn_backend == "gpu" or (nn_backend == "auto" and L.device_count() > 0 and nall >= 2000)) and \
            spatial_dim <= 8 and m <= 255                                    # limits of the brute-force kernel
        NNarray = S.find_ordered_nn_gpu(locsord, m) if use_gpu else S.find_ordered_nn(locsord, m)
    NNarray = np.asarray(NNarray).astype(np.int32)
    if have_pred and pred_cond == "independent":                             # :168-178
        if ordering_pred == "obspred":
            # every prediction location conditions on itself and its m nearest OBSERVED locations, descending index
            from scipy.spatial import cKDTree
            _, idx = cKDTree(locsord[:n]).query(locsord[n:], k=m)
            idx = np.sort(idx.reshape(n_p, m) + 1, axis=1)[:, ::-1]
            NNarray = NNarray.copy()
            NNarray[n:] = np.concatenate([(n + np.arange(1, n_p + 1))[:, None], idx], axis=1)
        else:
            warnings.warn("indep. conditioning currently only implemented for obspred ordering")
    if cond_yz == "SGV":                                                     # :182-183
        Cond = S.whichCondOnLatent(NNarray, firstind_pred=n + 1)
    elif cond_yz == "SGVT":                                                  # :184-185
        Cond = np.vstack([S.whichCondOnLatent(NNarray[:n]), np.ones((n_p, m + 1), dtype=np.int8)])
    elif cond_yz == "y":                                                     # :186-187
        Cond = np.where(NNarray != 0, 1, -1).astype(np.int8)
    elif cond_yz == "z":                                                     # :189-190
        if have_pred:
            raise ValueError("cond.yz='z' cannot be combined with prediction locations (an unobserved location has no z "
                             "to condition on; the reference fails in U_sparsity)")
        Cond = np.where(NNarray != 0, 0, -1).astype(np.int8)
        Cond[:, 0] = 1
    elif cond_yz in ("RVP", "LK", "zy"):                                     # :191-223 response-latent ('zy') trick
        obs = np.concatenate([np.ones(n, bool), np.zeros(nall, bool)])       # :195
        locsord = np.vstack([locsord[:n], locsord])                          # :196
        NNs = S.get_knn(locsord[:n], m - 1).astype(np.int64)                 # :199
        if cond_yz in ("RVP", "zy"):                                         # :200-203 latent y.obs where it comes earlier
            prev = NNs < np.arange(1, n + 1)[:, None]
            NNs[prev] += n
        NN_z = np.concatenate([np.arange(1, n + 1)[:, None], np.zeros((n, m), dtype=np.int64)], axis=1)       # :206
        NN_y = np.concatenate([(np.arange(1, n + 1) + n)[:, None], np.arange(1, n + 1)[:, None], NNs], axis=1)   # :207
        if not have_pred:                                                    # :208-210
            NN_yp = np.zeros((0, m + 1), dtype=np.int64)
            ordering_pred = "obspred"
        else:
            if ordering_pred != "obspred":
                warnings.warn("ZY only implemented for obspred ordering")
            NN_yp = NNarray[n: n + n_p].astype(np.int64)
            if cond_yz == "zy":                                              # :213-214
                NN_yp = np.where(NN_yp != 0, NN_yp + n, 0)
            else:                                                            # :215-218
                NN_yp = np.where(NN_yp > n, NN_yp + n, NN_yp)
        NNarray = np.vstack([NN_z, NN_y, NN_yp]).astype(np.int32)            # :220
        Cond = np.where(NNarray == 0, -1, (NNarray > n).astype(np.int8)).astype(np.int8)   # :223
        Cond[:, 0] = 1
        cond_yz = "zy"
    else:
        raise ValueError(f"cond.yz='{cond_yz}' not defined")                 # :226
    # a neighbour conditioned on as an observation must have one: the reference would put NA indices into sparseMatrix
    # (R/U_sparsity.R:52) and fail in createU; e.g. cond.yz='SGV' with ordering.pred='general' in two dimensions
    nb = np.where(NNarray > 0, NNarray - 1, 0)
    if np.any((Cond == 0) & (NNarray > 0) & ~np.asarray(obs)[nb]):
        raise ValueError(f"cond.yz='{cond_yz}' with ordering.pred='{ordering_pred}' conditions on the observation of an "
                         "unobserved location; use ordering.pred='obspred' or cond.yz in {'y','zy'}")
    U_prep = S.U_sparsity(locsord, NNarray, obs, Cond)                       # :230
    return dict(locsord=locsord, obs=obs, ord=ord_, ord_z=ord_z, ord_pred=ordering_pred, U_prep=U_prep,
                cond_yz=cond_yz, ic0=ic0, conditioning="NN")                 # :234-235


def _plan_for(va, device=0):
    key = ("_plan", device)
    if key not in va:
        prep = va["U_prep"]
        va[key] = Plan(va["locsord"], prep["revNNarray"], prep["revCond"], device=device)
    return va[key]


def _device_nuggets(va, nug):
    """Nugget argument of Plan.eval: a length-1 array when the nugget is constant (it then travels in the launch
    arguments), else the n-vector in ordering (R/createU.R:75-77)."""
    nug = np.atleast_1d(np.asarray(nug, dtype=np.float64))
    if nug.size == 1 or np.all(nug == nug[0]):
        return nug[:1]
    return nug[va["ord"] - 1]


def _ordered_nuggets(va, nuggets, n):
    """R/createU.R:73-78: (nuggets.all.ord for every row of locsord, nuggets.ord for the observations, nuggets)."""
    nug = np.atleast_1d(np.asarray(nuggets, dtype=np.float64))
    if nug.size == 1:
        nug = np.repeat(nug, n)                                               # :74
    nlat = va["locsord"].shape[0]                                            # = sum(latent)
    nug_all = np.concatenate([nug, np.zeros(nlat - n)])                      # :75 prediction / dummy locations carry none
    ord_ = va["ord"]
    ord_all = np.concatenate([ord_[:n], ord_ + n]) if va["cond_yz"] == "zy" else ord_   # :76
    return nug_all[ord_all - 1], nug_all[va["ord_z"] - 1], nug


# ---------------------------------------------------------------------------
# createU — R/createU.R:65-201 (NN branch)
# ---------------------------------------------------------------------------
def createU(vecchia_approx, covparms, nuggets, covmodel="matern", device=0):
    """Returns the U.obj list of R/createU.R:196-199 as a dict; U is a scipy.sparse CSC matrix."""
    import scipy.sparse as sp
    va = vecchia_approx
    prep = va["U_prep"]
    n = int(np.sum(va["obs"]))
    size = prep["size"]
    latent = np.zeros(size, dtype=bool)
    latent[prep["y_ind"] - 1] = True
    nug_all_ord, nug_ord, nug = _ordered_nuggets(va, nuggets, n)
    revNN, revCond = prep["revNNarray"], prep["revCond"]
    if np.any(nug == 0):                                                     # :83-86
        zero_idx = np.where(nug_ord == 0)[0] + 1
        revCond = revCond.copy()
        revCond[np.isin(revNN, zero_idx)] = 1
    plain = (n == va["locsord"].shape[0]) and va["cond_yz"] != "zy"          # every row observed: the resident plan serves
    if isinstance(covmodel, str):                                            # :152-154
        if np.any(nug == 0) or not plain:
            ent = U_NZentries(prep["n_cores"], n, va["locsord"], revNN, revCond, nug_all_ord, nug_ord, covmodel,
                              covparms)
            Lent, Zent = ent["Lentries"], ent["Zentries"]
        else:
            plan = _plan_for(va, device)
            plan.eval(covmodel, covparms, nug_all_ord if nug.size > 1 and not np.all(nug == nug[0]) else nug[:1],
                      GPV_WANT_U)
            Lent, Zent = plan.Lentries(), plan.Zentries()
    elif isinstance(covmodel, np.ndarray):                                   # :149-151
        ent = U_NZentries_mat(prep["n_cores"], n, va["locsord"], revNN, revCond, nug_all_ord, nug_ord, covmodel,
                              covparms)
        Lent, Zent = ent["Lentries"], ent["Zentries"]
    else:
        raise TypeError("argument 'covmodel' type not supported")            # :155
    # :158-160 keep the first n0 entries of every row (row-major walk), append Zentries
    n0 = (revNN != 0).sum(axis=1)
    keep = np.arange(revNN.shape[1])[None, :] < n0[:, None]
    vals = np.concatenate([np.ascontiguousarray(Lent)[keep], Zent])
    U = sp.csc_matrix((vals, (prep["colindices"] - 1, prep["rowpointers"] - 1)), shape=(size, size))   # :161-162
    ord_, obs, zero_nugg = va["ord"], va["obs"], {}
    if va["cond_yz"] == "zy":                                                # :166-171 rows/columns of the dummy y's
        keepd = np.ones(size, dtype=bool)
        keepd[2 * np.arange(n)] = False                                      # dummy = 2*(1:n)-1 (1-based)
        U = U.tocsr()[keepd][:, keepd].tocsc()
        latent = latent[keepd]
        obs = np.delete(obs, np.arange(n, 2 * n))
        size = int(keepd.sum())
    if np.any(nug == 0):                                                     # :173-193
        # rows/columns of observations with zero noise are removed; the latent variable they pin down
        # takes their place as an "observed" row
        Ucsc = U.tocsc()
        diag = Ucsc.diagonal()
        inds_U = np.where(np.isinf(diag))[0]                                 # :178
        cond_on = np.array([Ucsc.indices[Ucsc.indptr[j]:Ucsc.indptr[j + 1]].min() for j in inds_U])   # :179
        keep = np.ones(size, dtype=bool)
        keep[inds_U] = False
        U = Ucsc[keep][:, keep].tocsc()                                      # :180
        inds_z = np.where(np.isin(np.where(~latent)[0], inds_U))[0]          # :183 (0-based positions)
        inds_locs = np.where(np.isin(np.where(latent)[0], cond_on))[0]       # :184
        zero_nugg = dict(inds_U=inds_U + 1, inds_z=inds_z + 1, inds_locs=inds_locs + 1)
        latent = latent.copy()
        latent[cond_on] = False                                              # :188
        latent = latent[keep]                                                # :189
        rest = np.setdiff1d(np.arange(len(ord_)), inds_locs)
        ord_ = np.concatenate([ord_[rest], ord_[inds_locs]])                 # :190
        obs = np.concatenate([obs[rest], obs[inds_locs]])                    # :191
    return dict(U=U, latent=latent, ord=ord_, obs=obs, zero_nugg=zero_nugg, ord_pred=va["ord_pred"],
                ord_z=va["ord_z"], cond_yz=va["cond_yz"], ic0=va["ic0"], Lentries=Lent, Zentries=Zent)


# ---------------------------------------------------------------------------
# vecchia_likelihood — R/vecchia_likelihood.R
# ---------------------------------------------------------------------------
def _removeNAs(z, nuggets):
    """R/vecchia_likelihood.R:45-58."""
    z = np.asarray(z, dtype=np.float64).copy()
    nug = np.atleast_1d(np.asarray(nuggets, dtype=np.float64)).copy()
    isna = np.isnan(z)
    if isna.any():
        if nug.size < z.size:
            new = np.zeros(z.size)
            new[~isna] = nug if nug.size > 1 else nug[0]
            nug = new
        nug[isna] = np.var(z[~isna], ddof=1) * 1e8
        z[isna] = np.mean(z[~isna])
    return z, nug


def ichol_lower(Wrev):
    """R/ichol.R:16-59 -> src/ic0.cpp:43-64 through the native gpv_ic0: the lower IC(0) factor (CSR) of a sparse SPD
    matrix on its own pattern."""
    import scipy.sparse as sp
    Lw = sp.tril(Wrev, format="csr")
    Lw.sort_indices()
    ptrs = np.ascontiguousarray(Lw.indptr, dtype=np.int32)
    inds = np.ascontiguousarray(Lw.indices, dtype=np.int32)
    vals = np.ascontiguousarray(Lw.data, dtype=np.float64).copy()
    nbad = C.c_int64(0)
    L.check(L.lib().gpv_ic0(Lw.shape[0], L.iptr(ptrs), L.iptr(inds), L.dptr(vals), C.byref(nbad)), "gpv_ic0")
    return sp.csr_matrix((vals, inds, ptrs), shape=Lw.shape)


def _chol_rev(A, ic0):
    """t(chol(revMat(A))) as a sparse lower-triangular matrix (R/vecchia_prediction.R:74-81): SuperLU without pivoting in
    the natural order is L D L^T, the Cholesky factor is L sqrt(D); with ic0 the zero-fill factor of src/ic0.cpp."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    nA = A.shape[0]
    rev = np.arange(nA - 1, -1, -1)
    Arev = A.tocsr()[rev][:, rev].tocsc()
    if ic0:
        return ichol_lower(Arev)
    lu = spla.splu(Arev, permc_spec="NATURAL", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
    if not (np.array_equal(lu.perm_r, np.arange(nA)) and np.array_equal(lu.perm_c, np.arange(nA))):
        raise RuntimeError("U2V: the sparse factorisation pivoted (matrix not positive definite?)")
    d = lu.U.diagonal()
    return (lu.L @ sp.diags(np.sqrt(d))).tocsr()


def U2V(U_obj):
    """R/vecchia_prediction.R:62-111.  Host-side like the reference (CHOLMOD there; sequential sparse factorisation,
    SURVEY §8f-1).  Returns an object with `solve` = (V V^T)^{-1} and `U.diagonal()` = diag(V)^2:
      * general ordering, non-'zy': a SuperLU factor of W.rev = rev(U_y U_y^T) (or the IC(0) factor with ic0 = TRUE, :76-77);
      * 'zy': V.ord is the reversed latent block of U itself, no factorisation (:68-70);
      * obs-pred ordering: prediction columns of U_y unchanged, Cholesky of the observed block only (:85-107)."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    U = U_obj["U"].tocsr()
    latent = U_obj["latent"]
    lat_idx = np.where(latent)[0]
    Uy = U[lat_idx, :]
    if U_obj.get("cond_yz") == "zy":                                        # :68-70
        B = Uy[:, lat_idx]
        r = np.arange(B.shape[0] - 1, -1, -1)
        return _TriFactor(B[r][:, r].tocsr())
    if U_obj.get("ord_pred", "general") != "obspred":                       # :72-83
        W = (Uy @ Uy.T).tocsc()
        nW = W.shape[0]
        rev = np.arange(nW - 1, -1, -1)
        Wrev = W[rev][:, rev].tocsc()
        if U_obj.get("ic0", False):                                         # :76-77
            return _TriFactor(ichol_lower(Wrev))
        return spla.splu(Wrev, permc_spec="NATURAL", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
    last_obs = int(np.max(np.where(~latent)[0])) + 1                        # :87
    lat_before = int(latent[:last_obs].sum())                               # :88
    lat_after = int(latent[last_obs:].sum())                                # :89
    Vpr = Uy[:, last_obs:]
    Vpr = Vpr[np.arange(Vpr.shape[0] - 1, -1, -1)][:, np.arange(Vpr.shape[1] - 1, -1, -1)]   # :92 revMat
    Uoo = Uy[:lat_before, :last_obs]                                        # :95
    Voor = _chol_rev((Uoo @ Uoo.T).tocsc(), U_obj.get("ic0", False))        # :96-100
    Vor = sp.vstack([sp.csr_matrix((lat_after, lat_before)), Voor])         # :103-104
    return _TriFactor(sp.hstack([Vpr, Vor]).tocsr())                        # :106


class _TriFactor:
    """V.ord = t(ichol(W.rev)) with the two members vecchia_likelihood_U / vecchia_mean_host use of a SuperLU object:
    `U.diagonal()` such that sum(log) = log det, and `solve` = (V V^T)^{-1}."""

    def __init__(self, V):
        self.V = V.tocsr()
        d = V.diagonal()

        class _D:
            def diagonal(self_inner):
                return d * d
        self.U = _D()

    def solve(self, b):
        import scipy.sparse.linalg as spla
        y = spla.spsolve_triangular(self.V, np.asarray(b, dtype=np.float64), lower=True)
        return spla.spsolve_triangular(self.V.T.tocsr(), y, lower=False)


def vecchia_mean_host(z, U_obj):
    """R/vecchia_prediction.R:118-126 on the host: mu.ord = -W^{-1} z2 (ordered layout, one entry per latent variable)."""
    U = U_obj["U"].tocsr()
    latent = U_obj["latent"]
    zord = np.asarray(z, dtype=np.float64)[U_obj["ord_z"] - 1]
    z1 = U[np.where(~latent)[0], :].T @ zord
    z2 = U[np.where(latent)[0], :] @ z1
    lu = U2V(U_obj)
    return -(lu.solve(z2[::-1]))[::-1]


def split_mean(mu_ord, U_obj):
    """R/vecchia_prediction.R:134-139: ordered posterior mean -> (mu.obs, mu.pred) in the caller's order."""
    orig_order = np.argsort(U_obj["ord"], kind="stable")
    mu = np.asarray(mu_ord)[orig_order]
    obs_orig = np.asarray(U_obj["obs"], dtype=bool)[orig_order]
    return mu[obs_orig], mu[~obs_orig]


def vecchia_likelihood_U(z, U_obj):
    """R/vecchia_likelihood.R:63-99 on a host sparse U (denominator via U2V)."""
    U = U_obj["U"].tocsr()
    latent = U_obj["latent"]
    zord = np.asarray(z, dtype=np.float64)[U_obj["ord_z"] - 1]
    const = np.sum(~latent) * np.log(2 * np.pi)
    z1 = U[np.where(~latent)[0], :].T @ zord
    quadform_num = float(np.sum(z1 ** 2))
    logdet_num = -2.0 * float(np.sum(np.log(U.diagonal())))
    if latent.sum() == 0:
        logdet_denom = quadform_denom = 0.0
    else:
        z2 = U[np.where(latent)[0], :] @ z1
        lu = U2V(U_obj)
        logdet_denom = -float(np.sum(np.log(np.abs(lu.U.diagonal()))))      # -2 sum log diag(V) = -log det W
        y = lu.solve(z2[::-1])
        quadform_denom = float(z2[::-1] @ y)                                # |V^{-1} rev(z2)|^2 = z2' W^{-1} z2
    neg2loglik = logdet_num - logdet_denom + quadform_num - quadform_denom + const
    return -neg2loglik / 2


def vecchia_likelihood(z, vecchia_approx, covparms, nuggets, covmodel="matern", device=0):
    """R/vecchia_likelihood.R:14-27.  cond.yz='z' (and m=0) evaluates fully on the GPU with the
    fused epilogue; other conditioning modes build U on the GPU and finish the denominator
    on the host like the reference does (Matrix::chol)."""
    va = vecchia_approx
    if va["cond_yz"] == "zy":
        warnings.warn("cond.yz='zy' will produce a poor likelihood approximation. Use 'SGV' instead.")
    z_in = np.asarray(z, dtype=np.float64)
    nug_in = np.atleast_1d(np.asarray(nuggets, dtype=np.float64))
    if nug_in.size == 1 and not np.isnan(z_in.sum()):    # nothing for removeNAs to do (a NaN anywhere makes the sum NaN)
        z, nug = z_in, nug_in
    else:
        z, nug = _removeNAs(z, nuggets)
    n = int(np.sum(va["obs"]))
    plain = n == va["locsord"].shape[0]                  # every row of locsord observed: the fused device paths apply
    if plain and va["cond_yz"] in ("z", "false") and isinstance(covmodel, str) and not np.any(nug == 0):
        plan = _plan_for(va, device)
        plan.set_user_data(z, va["ord_z"])
        plan.eval(covmodel, covparms, _device_nuggets(va, nug), GPV_WANT_LOGLIK_Z)
        return loglik_z_from_sums(plan.sums(), n)
    if plain and va["cond_yz"] == "SGV" and isinstance(covmodel, str) and not np.any(nug == 0) and \
            _plan_for(va, device).ensure_posterior():                       # ic0 changes nothing: no fill
        # default mode: U, the numerator AND the posterior pass (U2V) on the GPU; SGV has no fill, so the
        # fixed-pattern factorisation equals the reference's Matrix::chol (R/vecchia_prediction.R:80).  (Refused only
        # when a conditioning set has more than 64 latent entries: the host route below.)
        plan = _plan_for(va, device)
        plan.set_user_data(z, va["ord_z"])
        plan.eval(covmodel, covparms, _device_nuggets(va, nug), GPV_WANT_DENOM)
        return loglik_from_sums(plan.sums(), n)
    if plain and va["cond_yz"] == "y" and isinstance(covmodel, str) and not np.any(nug == 0) and \
            not va.get("ic0", False):
        # latent conditioning throughout: W = U_y U_y^T fills in.  The device pass runs on the filled pattern when the fill is
        # bounded (symbolic factorisation on the host, once per plan); otherwise the host factorisation below, like CHOLMOD
        plan = _plan_for(va, device)
        if plan.has_posterior or plan.build_posterior_fill() is not None:
            plan.set_user_data(z, va["ord_z"])
            plan.eval(covmodel, covparms, _device_nuggets(va, nug), GPV_WANT_DENOM)
            return loglik_from_sums(plan.sums(), n)
    U_obj = createU(va, covparms, nug, covmodel, device=device)
    return vecchia_likelihood_U(z, U_obj)
