"""CPU tests: the C-ABI library loads and exports every symbol include/gpvecchia.h declares
(no compute without a GPU), error behaviour at the boundary, and the host-side mirror of
the R setup code against the oracle's literal restatement."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_all_exported():
    from gpvecchia_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "gpvecchia.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(gpv_[A-Za-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.EXPORTS)
    L = _lib.lib()
    for name in declared:
        assert getattr(L, name) is not None
    out = os.popen(f"nm -D --defined-only {_lib.LIB_PATH}").read()
    for name in declared:
        assert re.search(rf"\bT {name}\b", out), name


def test_library_contains_gfx950_code_object():
    from gpvecchia_amd import _lib
    _lib.lib()
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"gpv_sets_kernel" in blob


def test_status_strings_and_pure_host_entry_points():
    import gpvecchia_amd as G
    from gpvecchia_amd import _lib
    L = _lib.lib()
    assert L.gpv_version() >= 100
    assert L.gpv_max_p() == 192            # <= 64: unrolled register kernels; up to 192: the generic workgroup-per-set kernel
    for code in range(0, 9):
        assert len(L.gpv_status_string(code)) > 0
    # gpv_loglik_z_from_sums / gpv_numerator_from_sums are host arithmetic: testable without a GPU
    s = np.array([1.5, 2.0, 10.0, 7.0, 3.0, -4.0, 0.0, 5.0])
    assert G.loglik_z_from_sums(s, 5) == pytest.approx(-0.5 * (10.0 + 7.0 + 5 * np.log(2 * np.pi)), rel=1e-15)
    ld, qf = G.numerator_from_sums(s)
    assert ld == pytest.approx(-2 * 1.5 - 4.0) and qf == pytest.approx(5.0)
    s[6] = 1         # a failed block: the reference's zero row gives logdet.num = +Inf, loglik = -Inf (R/vecchia_likelihood.R:76,95-96)
    assert G.loglik_z_from_sums(s, 5) == -np.inf and G.loglik_from_sums(s, 5) == -np.inf


@pytest.mark.skipif(__import__("gpvecchia_amd").device_count() > 0, reason="only meaningful without a GPU")
def test_no_cpu_fallback_without_gpu(kat):
    import gpvecchia_amd as G
    va = G.vecchia_specify(kat["locs"], kat["m"], ordering="none", cond_yz="z")
    with pytest.raises(G.GpvError) as e:
        G.vecchia_likelihood(kat["z"], va, kat["covparms"], kat["nugget"])
    assert e.value.status == 1
    with pytest.raises(G.GpvError):
        G.MaternFun(np.zeros(3), [1, 1, .5])


def test_argument_errors_before_any_device_work():
    import gpvecchia_amd as G
    from gpvecchia_amd import _lib
    L = _lib.lib()
    status = ctypes.c_int(-1)
    L.gpv_U_NZentries(None, None, None, None, None, None, None, None, None, None, None, None, None, None, None, None,
                      ctypes.byref(status))
    assert status.value == 2
    # covType is validated before the device is touched (src/U_NZentries.cpp:27-29)
    locs = np.zeros((2, 2)); nn = np.array([[0, 1], [1, 2]]); cd = np.array([[0, 1], [1, 1]])
    with pytest.raises(G.GpvError) as e:
        G.U_NZentries(1, 2, locs, nn, cd, np.ones(2), np.ones(2), "gauss", [1, 1, .5])
    assert e.value.status == 3
    with pytest.raises(G.GpvError) as e:
        G.U_NZentries(1, 2, locs, nn, cd, np.ones(2), np.ones(2), "matern", [1, 1, -.7])
    assert e.value.status == 4


# ---------------------------------------------------------------------------------------
# host mirror vs the oracle's literal restatement of the R code
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("d", [1, 2, 3])
def test_find_ordered_nn_is_exact(d):
    from gpvecchia_amd import specify as S
    from oracle import r_side as R
    rng = np.random.default_rng(d)
    locs = rng.random((900, d))
    for m in (1, 5, 30):
        a = S.find_ordered_nn(locs, m)
        b = np.nan_to_num(R.findOrderedNN(locs, m)).astype(np.int32)
        assert np.array_equal(a, b)            # bit-exact neighbour arrays


def test_find_ordered_nn_ties_and_duplicates():
    from gpvecchia_amd import specify as S
    from oracle import r_side as R
    g = np.stack(np.meshgrid(np.arange(12.0), np.arange(9.0)), -1).reshape(-1, 2)     # regular grid: many exact ties
    g = np.vstack([g, g[:7]])                                                        # exact duplicates
    a = S.find_ordered_nn(g, 8)
    b = np.nan_to_num(R.findOrderedNN(g, 8)).astype(np.int32)
    assert np.array_equal(a, b)                # lower index wins ties, like R's stable order()
    sh = S.find_ordered_nn(g, 8, rows=(40, 77))
    assert np.array_equal(sh[40:77], a[40:77]) and not sh[:40].any() and not sh[77:].any()


@pytest.mark.parametrize("ordering", ["none", "coord", "maxmin"])
@pytest.mark.parametrize("cond", ["z", "y", "SGV"])
def test_vecchia_specify_matches_oracle(ordering, cond):
    import gpvecchia_amd as G
    from oracle import r_side as R
    rng = np.random.default_rng(4)
    locs = rng.random((300, 2))
    va = G.vecchia_specify(locs, 7, ordering=ordering, cond_yz=cond)
    vb = R.vecchia_specify(locs, 7, ordering=ordering, cond_yz=cond)
    assert np.array_equal(va["ord"], vb["ord"])
    pa, pb = va["U_prep"], vb["U_prep"]
    assert np.array_equal(pa["revNNarray"], np.nan_to_num(pb["revNNarray"]).astype(np.int32))
    assert np.array_equal(pa["revCond"], np.nan_to_num(pb["revCond"], nan=-1).astype(np.int8))
    for k in ("rowpointers", "colindices", "y_ind"):
        assert np.array_equal(pa[k], pb[k])
    assert pa["size"] == pb["size"] == 600


def test_native_whichCondOnLatent_matches_both_restatements():
    from gpvecchia_amd import specify as S
    from oracle import r_side as R
    rng = np.random.default_rng(12)
    for n, m, d in ((60, 3, 1), (400, 9, 2), (250, 30, 3)):
        locs = rng.random((n, d))
        NN = S.find_ordered_nn(locs, m)
        a = S.whichCondOnLatent(NN, native=True)
        b = S.whichCondOnLatent(NN, native=False)
        c = np.nan_to_num(R.whichCondOnLatent(np.where(NN == 0, np.nan, NN.astype(float))), nan=-1).astype(np.int8)
        assert np.array_equal(a, b) and np.array_equal(a, c)
    # prediction-style call: indices >= firstind_pred are always conditioned on as latent
    a = S.whichCondOnLatent(NN, firstind_pred=200, native=True)
    c = np.nan_to_num(R.whichCondOnLatent(np.where(NN == 0, np.nan, NN.astype(float)), firstind_pred=200), nan=-1)
    assert np.array_equal(a, c.astype(np.int8))


def test_vecchia_specify_edge_cases():
    import gpvecchia_amd as G
    locs = np.random.default_rng(0).random((20, 2))
    with pytest.warns(UserWarning):
        va = G.vecchia_specify(locs, 50, ordering="none", cond_yz="z")      # m > n -> m = n-1
    assert va["U_prep"]["revNNarray"].shape == (20, 20)
    v0 = G.vecchia_specify(locs, 0)                                          # independent case
    assert v0["cond_yz"] == "false" and v0["U_prep"]["revNNarray"].shape == (20, 2)
    assert np.array_equal(v0["U_prep"]["revNNarray"][:, 1], np.arange(1, 21))
    with pytest.raises(ValueError):
        G.vecchia_specify(locs, 3, ordering="spiral")
    with pytest.raises(ValueError):
        G.vecchia_specify(locs, 3, cond_yz="q")
    with pytest.warns(UserWarning):
        assert G.vecchia_specify(np.zeros(5), 2) is None                     # "Locations must be in matrix form"
    v1 = G.vecchia_specify(np.sort(np.random.default_rng(1).random((30, 1)), axis=0)[::-1].copy(), 3)
    assert np.array_equal(v1["ord"], np.arange(30, 0, -1))                   # 1-D default ordering = 'coord'


def test_orderings():
    from gpvecchia_amd import specify as S
    from oracle import r_side as R
    locs = np.random.default_rng(9).random((150, 2))
    assert np.array_equal(S.order_maxmin_exact(locs), R.order_maxmin_exact(locs))
    assert np.array_equal(S.order_coordinate(locs), R.order_coordinate(locs))
    o = S.order_maxmin_exact(locs)
    assert sorted(o.tolist()) == list(range(1, 151))
    # max-min property: each point maximises the distance to the already chosen ones
    d = R.rdist(locs)
    for t in (1, 5, 40):
        chosen = o[:t] - 1
        mind = d[:, chosen].min(axis=1)
        mind[chosen] = -1
        assert mind[o[t] - 1] == mind.max()
    mo = S.order_middleout(locs)
    assert np.array_equal(S.order_outsidein(locs), mo[::-1])


@pytest.mark.parametrize("n,d", [(1, 2), (2, 2), (17, 1), (800, 2), (3000, 2), (1500, 3), (600, 5)])
def test_native_maxmin_matches_definition(n, d):
    from gpvecchia_amd import specify as S
    from oracle import r_side as R
    locs = np.random.default_rng(n + d).random((n, d))
    a = S.order_maxmin_exact(locs, native=True)
    assert np.array_equal(a, S.order_maxmin_exact(locs, native=False))
    if n <= 1500:
        assert np.array_equal(a, R.order_maxmin_exact(locs))


def test_native_maxmin_grid_ties_and_clusters():
    from gpvecchia_amd import specify as S
    g = np.stack(np.meshgrid(np.arange(30.0), np.arange(20.0)), -1).reshape(-1, 2)     # all distances tie
    assert np.array_equal(S.order_maxmin_exact(g, native=True), S.order_maxmin_exact(g, native=False))
    rng = np.random.default_rng(3)
    c = np.vstack([rng.normal(0, 1e-3, (400, 2)), rng.normal(5, 1.0, (400, 2)), np.zeros((5, 2))])   # clusters + duplicates
    assert np.array_equal(S.order_maxmin_exact(c, native=True), S.order_maxmin_exact(c, native=False))


def test_removeNAs_semantics():
    from gpvecchia_amd.api import _removeNAs
    z = np.array([1.0, np.nan, 3.0, 5.0])
    z2, nug = _removeNAs(z, 0.1)
    assert z2[1] == 3.0 and nug[1] == np.var([1.0, 3.0, 5.0], ddof=1) * 1e8        # R/vecchia_likelihood.R:55-56
    assert np.array_equal(nug[[0, 2, 3]], [0.1, 0.1, 0.1])
    z3, n3 = _removeNAs(np.array([1.0, 2.0]), 0.3)
    assert np.array_equal(z3, [1.0, 2.0]) and n3.tolist() == [0.3]


def test_host_likelihood_U_matches_oracle():
    # the host-side (sparse) vecchia_likelihood_U / U2V of the mirror vs the oracle's dense restatement
    import scipy.sparse as sp
    import gpvecchia_amd as G
    from oracle import r_side as R
    rng = np.random.default_rng(2)
    locs = rng.random((200, 2)); z = rng.standard_normal(200)
    for cond in ("SGV", "y", "z"):
        vb = R.vecchia_specify(locs, 6, ordering="maxmin", cond_yz=cond)
        Uo = R.createU(vb, [1.0, 0.2, 1.5], 0.1)
        U_obj = dict(U=sp.csc_matrix(Uo["U"]), latent=Uo["latent"], ord_z=Uo["ord_z"])
        assert G.vecchia_likelihood_U(z, U_obj) == pytest.approx(R.vecchia_likelihood_U(z, Uo), rel=1e-11)


def test_ic0_native_matches_reference_restatement():
    # src/ic0.cpp:43-64 (called by R/ichol.R:54; U2V with ic0 = TRUE, R/vecchia_prediction.R:76-77)
    import ctypes as C
    import scipy.sparse as sp
    from gpvecchia_amd import _lib as L
    from gpvecchia_amd import api as A
    from oracle import r_side as R
    rng = np.random.default_rng(3)
    n = 60
    B = sp.random(n, n, density=0.08, random_state=4, format="csr")
    M = (B @ B.T + sp.identity(n) * 2.0).tocsr()
    ref_upper = R.ichol(M.toarray())                                  # upper factor, dense
    Lw = A.ichol_lower(M.tocsc())
    np.testing.assert_allclose(Lw.toarray(), ref_upper.T, rtol=1e-13, atol=1e-14)
    assert np.array_equal(Lw.toarray() != 0, np.tril(M.toarray()) != 0)          # zero fill: the pattern is kept
    # full pattern => IC(0) is the Cholesky factor
    D = rng.standard_normal((12, 12)); S = D @ D.T + 12 * np.eye(12)
    np.testing.assert_allclose(A.ichol_lower(sp.csc_matrix(S)).toarray(), np.linalg.cholesky(S), rtol=1e-12)
    # malformed structure (diagonal missing) is an error, not a crash
    ptrs = np.array([0, 1, 2], dtype=np.int32); inds = np.array([0, 0], dtype=np.int32); vals = np.ones(2)
    assert L.lib().gpv_ic0(2, L.iptr(ptrs), L.iptr(inds), L.dptr(vals), None) == 8          # GPV_ERR_INDEX (include/gpvecchia.h)
    # a non-positive pivot is counted
    ptrs = np.array([0, 1, 3], dtype=np.int32); inds = np.array([0, 0, 1], dtype=np.int32)
    vals = np.array([1.0, 2.0, 1.0]); nbad = C.c_int64(0)
    assert L.lib().gpv_ic0(2, L.iptr(ptrs), L.iptr(inds), L.dptr(vals), C.byref(nbad)) == 0 and nbad.value == 1


def test_nelder_mead_reproduces_the_known_answer_of_R_optim():
    # stats::optim(method = "Nelder-Mead") is what vecchia_estimate drives (R/vecchia_wrappers.R:87-93).  R's documentation
    # of optim prints, for `fr <- function(x) 100 * (x[2] - x[1]^2)^2 + (1 - x[1])^2; optim(c(-1.2, 1), fr)`:
    #   $par 1.000260 1.000506   $value 8.825241e-08   $counts function 195   $convergence 0
    from gpvecchia_amd.wrappers import _nelder_mead_nash
    fr = lambda x: 100 * (x[1] - x[0] ** 2) ** 2 + (1 - x[0]) ** 2
    x, f, count, code = _nelder_mead_nash(fr, [-1.2, 1.0], reltol=np.sqrt(np.finfo(float).eps), maxit=500)
    assert count == 195 and code == 0
    assert abs(x[0] - 1.000260) < 5e-7 and abs(x[1] - 1.000506) < 5e-7
    assert abs(f - 8.825241e-08) < 5e-14
    # evaluation limit -> convergence code 1 (optim's maxit counts function evaluations for this method)
    _, _, c2, code2 = _nelder_mead_nash(fr, [-1.2, 1.0], reltol=1e-12, maxit=50)
    assert code2 == 1 and 50 < c2 <= 53


def _build_c_caller(tmp_path):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "kat")
    libdir = os.path.join(root, "gpvecchia_amd")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "c_abi", "kat.c"), "-o", exe, "-L", libdir, "-lgpvecchia_hip",
                           f"-Wl,-rpath,{libdir}", "-lm"])
    return exe


@pytest.mark.skipif(__import__("gpvecchia_amd").device_count() > 0, reason="only meaningful without a GPU")
def test_plain_c_caller_links_and_reports_no_device(tmp_path):
    # include/gpvecchia.h is consumable by a C compiler and the library links into a plain C program (what R's .C() needs)
    import subprocess
    import gpvecchia_amd  # noqa: F401  (builds the library if missing)
    r = subprocess.run([_build_c_caller(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 77 and "no HIP device" in r.stdout


@pytest.mark.gpu
def test_plain_c_caller_known_answer(tmp_path):
    # the known-answer case of SURVEY.md §8c from a C program, through the .C()-style entry and the plan API
    import subprocess
    import gpvecchia_amd  # noqa: F401
    r = subprocess.run([_build_c_caller(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0 and "known-answer test ok" in r.stdout, r.stdout + r.stderr


def test_r_layout_cache_sees_in_place_edits_and_holds_no_reference():
    """api._r_layout_cached: the converted copies of the index arrays are reused only for the same objects with unchanged
    content (an in-place edit between two U_NZentries calls must not be answered from the stale copy), and the cache keeps
    the caller's arrays alive no longer than the caller does."""
    import gc
    import weakref
    from gpvecchia_amd import api as A
    rng = np.random.default_rng(0)
    nn = rng.integers(0, 50, size=(200, 6)).astype(np.int32)
    cd = rng.integers(-1, 2, size=(200, 6)).astype(np.int8)
    a1, b1 = A._r_layout_cached(nn, cd)
    a2, b2 = A._r_layout_cached(nn, cd)
    assert a2 is a1 and b2 is b1                                        # hit: same objects, same content
    nn[0, 0] += 1                                                       # first row is part of the fingerprint
    a3, _ = A._r_layout_cached(nn, cd)
    assert a3 is not a1 and a3[0, 0] == nn[0, 0]
    nn[17, 3] += 1                                                      # 1200 entries: every entry is in the sample
    a4, _ = A._r_layout_cached(nn, cd)
    assert a4 is not a3 and a4[17, 3] == nn[17, 3]
    w = weakref.ref(nn)
    del nn, a1, a2, a3, a4
    gc.collect()
    assert w() is None                                                  # only weak references inside the cache


def test_vecchia_laplace_data_support_checks():
    # R/vecchia_laplace_NR.R:48-54: the data must lie in the support of the likelihood (poisson: any(z < 0) | any(z %% 1 > 0);
    # logistic: values in {0, 1}; gamma: z > 0; beta: 0 <= z <= 1); host-side checks, written without numpy's fmod / isin
    from gpvecchia_amd.laplace import _families
    lp = dict(alpha=2, sigma=0.3, beta=0.5)
    cases = {
        "poisson": [([0.0, 3.0, 7.0], False), ([1.0, 2.5], True), ([-1.0, 2.0], True), ([-0.5], True), ([np.inf], False)],
        "logistic": [([0.0, 1.0, 1.0], False), ([0.0, 2.0], True), ([0.5], True)],
        "gamma": [([0.1, 3.0], False), ([0.0, 1.0], True)],
        "gamma_alt": [([0.1, 3.0], False), ([-1.0], True)],
        "beta": [([0.0, 0.5, 1.0], False), ([1.5], True), ([-0.1], True)],
        "gaussian": [([-3.0, 0.5], False)],
    }
    for model, rows in cases.items():
        bad = _families(model, lp)["bad"]
        for z, want in rows:
            assert bool(bad(np.asarray(z))) is want, (model, z)
    big = np.random.default_rng(0).poisson(1.5, 200_000).astype(float)
    assert not _families("poisson", lp)["bad"](big)
    big[123_456] += 0.25
    assert _families("poisson", lp)["bad"](big)


def test_hash_bytes_and_index_array_fingerprint():
    """gpv_hash_bytes (include/gpvecchia.h): deterministic, independent of the thread count by construction, every byte
    counts, empty input valid; gpvecchia_amd.api._fingerprint keys the R-layout cache on it (no optional module, no
    32-bit fallback) and skips empty arrays."""
    import ctypes as C
    from gpvecchia_amd import _lib as L, api as A
    rng = np.random.default_rng(0)
    a = rng.integers(0, 1000, size=(70_000, 31), dtype=np.int32)          # > 4 MiB: the 64-chunk threaded form

    def h(x, seed=5):
        out = (C.c_uint64 * 2)()
        assert L.lib().gpv_hash_bytes(C.c_void_p(x.ctypes.data), x.nbytes, seed, out) == 0
        return int(out[0]), int(out[1])
    assert h(a) == h(a.copy()) and h(a) != h(a, seed=6)
    b = a.copy(); b[69_999, 30] ^= 1
    c = a.copy(); c[0, 0] ^= 1 << 30
    assert len({h(a), h(b), h(c)}) == 3
    small = np.arange(13, dtype=np.uint8)                                  # tail bytes (13 % 8 != 0), single chunk
    s2 = small.copy(); s2[12] += 1
    assert h(small) != h(s2)
    out = (C.c_uint64 * 2)()
    assert L.lib().gpv_hash_bytes(None, 0, 1, out) == 0 and L.lib().gpv_hash_bytes(None, 8, 1, out) != 0
    assert A._fingerprint(np.zeros((0, 5), np.int32)) is None
    f1, f2 = A._fingerprint(a), A._fingerprint(b)
    assert f1 != f2 and f1 == A._fingerprint(a) and A._fingerprint(np.asfortranarray(a)) != f1   # (strides are part of it)
    nn_r, cd_r = A._r_layout_cached(np.zeros((0, 5), np.int32), np.zeros((0, 5), np.int8))
    assert nn_r.shape == (0, 5) and cd_r.shape == (0, 5)


def test_built_set_kernels_have_no_dpp_read_hazard():
    """The sweeps' DPP operands sit in inline asm, which hipcc's hazard recogniser does not see: a VGPR written by one of the
    two preceding instructions (a value fetched back from an AGPR under register pressure, say) must never be a DPP source.
    gpvecchia_amd.build.dpp_hazards disassembles every built set-kernel object; round 6 found two such reads in the P = 41,
    generic-dimension, general-nu instantiation while the pivot row's read was being moved, and pinned the sources ahead."""
    import glob
    from gpvecchia_amd import build as B
    objs = sorted(glob.glob(os.path.join(B.CSRC, "build", "sets_p*.o")))
    if not objs:
        pytest.skip("no object files in this tree (library built elsewhere)")
    seen = 0
    for o in objs:
        nd, bad = B.dpp_hazards(o)
        if nd is None:
            continue
        seen += nd
        assert bad == 0, (os.path.basename(o), nd, bad)
    assert seen > 10_000                                             # the DPP sweeps were really looked at
