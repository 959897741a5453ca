"""GPU: the specialised kernels compute with ONE representative per magnitude class (coefficients within QGS_HIP_MAGNITUDE_ULP
units in the last place of each other, default 2 -- qgs_amd/csrc/codegen.h Canonical); the generic kernels, the contraction
kernel and the reference use the caller's values as given (`value` is a run-time operand of sparse_mul3,
qgs/functions/sparse_mul.py:76-81).  What that means for results, made visible on tensors built to sit on the class boundaries:

* with the default, a coefficient within 2 ulp of an earlier one IS that earlier one in the specialised kernels -- the change of
  the input is at most 2 ulp of a coefficient, inside the 1e-14 gate of the path;
* QGS_HIP_MAGNITUDE_ULP=0 switches it off: specialised = generic to summation-order rounding;
* tensors that differ only inside a class share their code objects; their coefficient tables differ when the representative does.
"""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

C0 = 0.3711                                             # any normal number


def _ulps(c, n):
    for _ in range(n):
        c = np.nextafter(c, np.inf)
    return float(c)


def _jacobian(coo, val, ndim):
    from qgs_amd.tensors.qgtensor import CooTensor, QgsTensor
    j = QgsTensor.jacobian_from_tensor(CooTensor.from_coords(coo.T, val, (ndim + 1,) * 3))
    return np.ascontiguousarray(j.coords.T, dtype=np.int32), j.data


def _model(coo, val, ndim):
    from qgs_amd import _lib
    coo = np.asarray(coo, dtype=np.int32)
    val = np.asarray(val, dtype=np.float64)
    jcoo, jval = _jacobian(coo, val, ndim)
    return _lib.HipModel(ndim, coo, val, jcoo, jval), coo, val, jcoo, jval


def _boundary_tensor(ndim=6, first=0):
    """Every row holds c, c(1 + 1 ulp), c(1 + 2 ulp), c(1 + 3 ulp) (in an order that starts at `first`), a linear and a constant term."""
    cs = [_ulps(C0, k) for k in range(4)]
    coo, val = [], []
    for i in range(1, ndim + 1):
        pairs = [(1, 1), (1, 2), (2, 2), (2, 3)]
        for n, (j, k) in enumerate(pairs):
            coo.append((i, j, k))
            val.append(cs[(first + n + i) % 4] * (-1.0 if (i + n) % 3 == 0 else 1.0))
        coo.append((i, 0, i)); val.append(-0.1 * i)
        coo.append((i, 0, 0)); val.append(0.01 * i)
    return np.array(coo), np.array(val)


def test_default_classes_stay_inside_the_gate_and_knob_zero_is_exact(monkeypatch):
    from oracle.oracle import OracleModel                                   # the checker
    ndim = 6
    coo, val = _boundary_tensor(ndim)
    x = np.random.RandomState(5).rand(200, ndim) + 0.5
    res = {}
    for ulp in ('2', '0'):
        monkeypatch.setenv('QGS_HIP_MAGNITUDE_ULP', ulp)
        m, coo_, val_, jcoo, jval = _model(coo, val, ndim)
        assert m.specialised_available
        ref = OracleModel(ndim, coo_, val_, jcoo, jval)
        want_f, want_j = ref.f(0., x), ref.Df(0., x[:16])
        for kind, name in ((2, 'spec'), (1, 'gen')):
            m.set_kernel(kind)
            f, jac = m.tendencies(x), m.jacobian(x[:16])
            assert rel_err(f, want_f) < 1e-14 and rel_err(jac, want_j) < 1e-14, (ulp, name)
            res[ulp, name] = f
        # a short trajectory through the specialised stepper, against the oracle on the caller's values
        m.set_kernel(2)
        t = np.linspace(0., 0.2, 21)
        b = np.array([1 / 6, 1 / 3, 1 / 3, 1 / 6]); c = np.array([0., .5, .5, 1.]); a = np.zeros((4, 4)); a[1, 0] = a[2, 1] = .5; a[3, 2] = 1.
        traj = m.rk_integrate(t, x[:8] * 0.1, 1, 5, b, c, a)
        want = ref.integrate_runge_kutta_jit(t, x[:8] * 0.1, 1, 5, b, c, a)
        assert rel_err(traj, want) < 1e-13, ulp
        m.close()
    # the generic kernels never see the classes
    assert np.array_equal(res['2', 'gen'], res['0', 'gen'])
    # knob 0: the specialised kernel has the caller's coefficients; what is left against the generic kernel is the order of the sums
    # (factored groups, FMAs): within one unit in the last place of the sum of the absolute terms
    mags = np.zeros_like(res['0', 'gen'])
    xx = np.concatenate((np.ones((x.shape[0], 1)), x), axis=1)
    for (i, j, k), v in zip(coo, val):
        mags[:, i - 1] += np.abs(v * xx[:, j] * xx[:, k])
    assert (np.abs(res['0', 'spec'] - res['0', 'gen']) <= np.spacing(mags)).all()


def test_representative_is_what_the_specialised_kernel_multiplies_with(monkeypatch, tmp_path):
    """Rows of one term each, x = 1: f_i is the coefficient the kernel used, bit for bit.  Row 1 holds c (the first value to
    appear: the representative), rows 2 .. 4 hold c + 1, 2, 3 ulp."""
    from test_kernel_cache import _objs, _structs
    monkeypatch.setenv('QGS_HIP_CACHE_DIR', str(tmp_path))
    ndim = 4
    coo = np.array([(i, 1, 1) for i in range(1, ndim + 1)])
    x = np.ones((64, ndim))

    def f_of(vals, ulp, kind):
        monkeypatch.setenv('QGS_HIP_MAGNITUDE_ULP', str(ulp))
        m, *_ = _model(coo, np.array(vals), ndim)
        m.set_kernel(kind)
        out = m.tendencies(x)
        assert (out == out[0]).all()
        m.close()
        return out[0]
    c = [_ulps(C0, k) for k in range(4)]
    # default: c + 1 ulp and c + 2 ulp are c; c + 3 ulp is a class of its own
    assert f_of(c, 2, 2).tolist() == [c[0], c[0], c[0], c[3]]
    assert f_of(c, 2, 1).tolist() == c                                          # generic: the values as given
    n_obj, n_struct = len(_objs(tmp_path)), len(_structs(tmp_path))
    # the same tensor one ulp up in its first entry: same classes, same code objects and layouts, another representative
    c4 = _ulps(C0, 4)
    shifted = [c[1], c[1], c[2], c4]
    assert f_of(shifted, 2, 2).tolist() == [c[1], c[1], c[1], c4]
    assert len(_objs(tmp_path)) == n_obj and len(_structs(tmp_path)) == n_struct
    # knob 0: every value is its own class -- exact, and a structure of its own in the cache
    assert f_of(c, 0, 2).tolist() == c
    assert len(_structs(tmp_path)) > n_struct
    # subnormal magnitudes are never merged (a unit in the last place is not small against them)
    tiny = [5e-324, 1.5e-323, 1.5e-323, 2.5e-323]
    assert f_of(tiny, 2, 2).tolist() == tiny
