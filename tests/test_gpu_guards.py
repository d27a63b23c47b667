"""Failure behaviour of the device-pointer boundary (include/shpair.h): what must be refused loudly instead of
producing silent garbage — output arrays that are not ordinary device memory (the FP64 hardware atomics of the force
accumulation are unreliable on host-coherent / managed allocations), atom types or shape indices outside their
tables reaching the kernel, and bounding radii below the shape's true maximum."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(n=400, lmax=4, nq=8):
    import torch
    from shpair import ShPair, ShPairError, shapes, bed  # noqa: F401
    a = shapes.random_shape(lmax, 5)
    sp = ShPair(0)
    sp.settings(nq)
    sp.set_ntypes(1, 1)
    sp.set_shape(0, lmax, a)
    sp.coeff("*", "*", 500.0, 1.25)
    rmax = [sp.rmax(0)]
    b = bed.make_bed(n, rmax)
    il, of, jl = bed.half_neighbor_list(b["x"], b["shtype"], rmax)
    sp.set_neighbors_csr(il, of, jl)
    dev = torch.device("cuda:0")
    t = {k: torch.from_numpy(b[k]).to(dev) for k in ("x", "quat", "type", "shtype")}
    return sp, b, t, dev


def test_output_arrays_must_be_device_memory():
    import torch
    from shpair import ShPairError
    sp, b, t, dev = _setup()
    n = b["x"].shape[0]
    f = torch.zeros(n, 3, dtype=torch.float64, device=dev)
    tq = torch.zeros_like(f)
    pinned = torch.zeros(n, 3, dtype=torch.float64).pin_memory()      # host-coherent: atomics may be dropped there
    with pytest.raises(ShPairError) as e:
        sp.compute_device(n, 0, t["x"].data_ptr(), t["quat"].data_ptr(), t["type"].data_ptr(), t["shtype"].data_ptr(),
                          pinned.data_ptr(), tq.data_ptr())
    assert "device memory" in str(e.value)
    plain = np.zeros((n, 3))                                          # not known to HIP at all
    with pytest.raises(ShPairError):
        sp.compute_device(n, 0, t["x"].data_ptr(), t["quat"].data_ptr(), t["type"].data_ptr(), t["shtype"].data_ptr(),
                          f.data_ptr(), plain.ctypes.data)
    # and the good case still runs
    sp.compute_device(n, 0, t["x"].data_ptr(), t["quat"].data_ptr(), t["type"].data_ptr(), t["shtype"].data_ptr(), f.data_ptr(),
                      tq.data_ptr())
    sp.synchronize()
    torch.cuda.synchronize()
    assert float(f.abs().max()) > 0
    sp.close()


@pytest.mark.parametrize("what", ["shtype", "type"])
def test_bad_index_on_the_device_path_raises_instead_of_reading_out_of_bounds(what):
    import torch
    from shpair import ShPairError
    sp, b, t, dev = _setup()
    n = b["x"].shape[0]
    f = torch.zeros(n, 3, dtype=torch.float64, device=dev)
    tq = torch.zeros_like(f)
    bad = t[what].clone()
    bad[17] = 7 if what == "shtype" else 0
    args = dict(t)
    args[what] = bad
    sp.compute_device(n, 0, args["x"].data_ptr(), args["quat"].data_ptr(), args["type"].data_ptr(), args["shtype"].data_ptr(),
                      f.data_ptr(), tq.data_ptr())
    torch.cuda.synchronize()
    with pytest.raises(ShPairError) as e:
        sp.synchronize()
    assert "outside its table" in str(e.value)
    assert bool(torch.isfinite(f).all())
    # the flag is cleared by the report: a clean compute afterwards passes
    f.zero_()
    sp.compute_device(n, 0, t["x"].data_ptr(), t["quat"].data_ptr(), t["type"].data_ptr(), t["shtype"].data_ptr(), f.data_ptr(),
                      tq.data_ptr())
    torch.cuda.synchronize()
    sp.synchronize()
    sp.close()


def test_bounding_radius_below_the_true_maximum_is_refused():
    from shpair import ShPair, ShPairError, shapes, capi
    lmax = 6
    a = shapes.random_shape(lmax, 11, amp=0.25)
    sp = ShPair(0)
    sp.set_ntypes(1, 1)
    sp.set_shape(0, lmax, a)                       # default: fine
    rdef = sp.rmax(0)
    # the true maximum (dense sample) lies below the default and above 0.98 of it
    rng = np.random.default_rng(0)
    u = rng.normal(size=(20000, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    rtrue = max(capi.shape_radius(lmax, a, v) for v in u)
    assert rtrue < rdef and rtrue > 0.97 * rdef
    sp.set_shape(0, lmax, a, rmax=rtrue * 1.001)   # a tight but valid user radius
    with pytest.raises(ShPairError) as e:
        sp.set_shape(0, lmax, a, rmax=0.99 * rtrue)
    assert "below the shape's largest radius" in str(e.value)
    sp.close()


def test_exactly_tight_bounding_radius_of_a_sphere_is_accepted():
    """rmax = a00 / sqrt(4 pi) is the analytic radius of an L = 0 shape; the refined maximum the
    library compares with is a rounded evaluation of the same number and may sit an ulp above it."""
    from shpair import ShPair
    sp = ShPair(0)
    sp.set_ntypes(1, 1)
    for r in (1.0, 0.7312345678901234, 3.3):
        a00 = r * np.sqrt(4.0 * np.pi)
        anm = np.zeros(2)
        anm[0] = a00
        for rm in (r, np.nextafter(r, 0.0), np.nextafter(r, 10.0)):
            sp.set_shape(0, 0, anm, rmax=float(rm))
            assert sp.rmax(0) == float(rm)
    sp.close()


def test_coincident_centres_are_skipped_and_reported(oracle):
    """docs/SPEC.md 2, step 1: a listed pair with separation 0 has no line of centres.  It must contribute nothing, raise
    the error bit (SHPAIR_EINVAL, 'coincident centres') and leave every other pair's forces untouched and finite —
    it used to divide by rho and add NaN into f[i], f[j].  The oracle skips such a pair silently."""
    import torch
    from shpair import ShPair, ShPairError, shapes, bed
    lmax, nq, n = 4, 8, 400
    a = shapes.random_shape(lmax, 5)
    sp = ShPair(0)
    sp.settings(nq)
    sp.set_ntypes(1, 1)
    sp.set_shape(0, lmax, a)
    sp.coeff("*", "*", 500.0, 1.25)
    rmax = [sp.rmax(0)]
    b = bed.make_bed(n, rmax)
    il, of, jl = bed.half_neighbor_list(b["x"], b["shtype"], rmax)
    sp.set_neighbors_csr(il, of, jl)
    x = b["x"].copy()
    i0, j0 = int(il[5]), int(jl[of[5]])
    x[j0] = x[i0]                                  # the listed pair (i0, j0) now has rho = 0
    dev = torch.device("cuda:0")
    t = {k: torch.from_numpy(v).to(dev) for k, v in (("x", x), ("quat", b["quat"]), ("type", b["type"]), ("shtype", b["shtype"]))}
    f = torch.zeros(n, 3, dtype=torch.float64, device=dev)
    tq = torch.zeros_like(f)
    sp.compute_device(n, 0, t["x"].data_ptr(), t["quat"].data_ptr(), t["type"].data_ptr(), t["shtype"].data_ptr(), f.data_ptr(), tq.data_ptr())
    torch.cuda.synchronize()
    with pytest.raises(ShPairError) as e:
        sp.synchronize()
    assert "coincident centres" in str(e.value)
    assert bool(torch.isfinite(f).all()) and bool(torch.isfinite(tq).all())
    K, E = np.full((2, 2), 500.0), np.full((2, 2), 1.25)
    o = oracle.compute([(lmax, a, rmax[0])], K, E, nq, n, x, b["quat"], b["type"], b["shtype"], il, of, jl)
    assert np.all(np.isfinite(o["f"]))
    fs = np.abs(o["f"]).max()
    assert np.abs(f.cpu().numpy() - o["f"]).max() < 1e-9 * fs and np.abs(tq.cpu().numpy() - o["torque"]).max() < 1e-9 * fs
    # the host-pointer entry point reports it from the same call
    with pytest.raises(ShPairError) as e:
        sp.compute(n, x, b["quat"], b["type"], b["shtype"])
    assert "coincident centres" in str(e.value)
    # a separation that is not a number is the same case
    x2 = b["x"].copy()
    x2[i0, 1] = np.nan
    with pytest.raises(ShPairError) as e:
        sp.compute(n, x2, b["quat"], b["type"], b["shtype"])
    assert "coincident centres" in str(e.value)
    sp.close()
