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
