"""Tendencies of the model: `create_tendencies(params)` -> `[f, Df, ...]`.

Same contract as the reference's qgs/functions/tendencies.py:20-130, but `f` and `Df` are callable
*objects* that carry the COO operands (``.coo``, ``.val``, ``.ndim``) instead of numba closures, so an
integrator can hand the tensor to the GPU once.  Calling them evaluates the tendencies with the HIP
kernels (`qgs_tendencies` / `qgs_jacobian` of include/qgs_hip.h) for one state ``(ndim,)`` -- as the
reference's closures do -- or for a whole batch ``(n, ndim)``.  There is no CPU evaluation path.
"""
import numpy as np

from qgs_amd import _lib


class TensorOperands(object):
    """The arrays the reference's closures capture (tendencies.py:92-96) plus the lazily created GPU handle."""

    def __init__(self, ndim, coo, val, jcoo, jval):
        self.ndim = int(ndim)
        self.coo = np.ascontiguousarray(coo, dtype=np.int32)
        self.val = np.ascontiguousarray(val, dtype=np.float64)
        self.jcoo = None if jcoo is None else np.ascontiguousarray(jcoo, dtype=np.int32)
        self.jval = None if jval is None else np.ascontiguousarray(jval, dtype=np.float64)
        self._models = {}

    def hip_model(self, device=0):
        """GPU handle of these tensors (created on first use, then shared by f, Df and integrators): a `HipModel` on
        GPU `device` (an index), or a `HipModelGroup` over several GPUs for ``device='all'`` (every visible GPU) or a
        list / tuple of indices (an index may repeat: two pipelines on that GPU)."""
        if isinstance(device, str):
            if device != 'all':
                raise ValueError("device must be a GPU index, a list of indices or 'all'")
            device = tuple(_lib.visible_devices())
        if isinstance(device, (list, tuple)):
            key = tuple(int(d) for d in device)
            m = self._models.get(key)
            if m is None:
                m = _lib.HipModelGroup(self.ndim, self.coo, self.val, self.jcoo, self.jval, devices=key)
                self._models[key] = m
            return m
        device = int(device)
        m = self._models.get(device)
        if m is None:
            m = _lib.HipModel(self.ndim, self.coo, self.val, self.jcoo, self.jval, device=device)
            self._models[device] = m
        return m

    def release(self):
        for m in self._models.values():
            m.close()
        self._models = {}

    # picklable like the reference's f/Df/params (user_guide.rst "saving the model"): drop the GPU handles
    def __getstate__(self):
        d = dict(self.__dict__)
        d['_models'] = {}
        return d


class _TensorFunction(object):
    def __init__(self, operands, device=0):
        self.operands = operands
        self.device = device

    ndim = property(lambda self: self.operands.ndim)

    def hip_model(self, device=None):
        return self.operands.hip_model(self.device if device is None else device)


class TendenciesFunction(_TensorFunction):
    """f(t, x): dx_i = sum_jk T_ijk x_j x_k with x_0 = 1 (tendencies.py:111-115).  `t` is ignored (autonomous)."""

    coo = property(lambda self: self.operands.coo)
    val = property(lambda self: self.operands.val)

    def __call__(self, t, x):
        return self.hip_model().tendencies(np.asarray(x, dtype=np.float64))


class JacobianFunction(_TensorFunction):
    """Df(t, x): J_ij = sum_k (T_ijk + T_ikj) x_k (tendencies.py:117-121); (ndim, ndim), or (n, ndim, ndim)."""

    coo = property(lambda self: self.operands.jcoo)
    val = property(lambda self: self.operands.jval)

    def __call__(self, t, x):
        return self.hip_model().jacobian(np.asarray(x, dtype=np.float64))


def tendencies_from_tensor(ndim, coo, val, jcoo=None, jval=None, device=0):
    """(f, Df) from raw COO operands, e.g. ``aotensor.tensor.coords.T, aotensor.tensor.data`` -- what users of
    the reference pass to `sparse_mul3` by hand (documentation user_guide.rst:437-458)."""
    ops = TensorOperands(ndim, coo, val, jcoo, jval)
    return TendenciesFunction(ops, device), (JacobianFunction(ops, device) if jcoo is not None else None)


def create_tendencies(params, return_inner_products=False, return_qgtensor=False):
    """Build inner products -> tendencies tensor -> `[f, Df, (aip, oip, gip)?, qgtensor?]`.

    Mirrors qgs/functions/tendencies.py:20-130: analytic inner products for models configured from spectral
    blocks, quadrature ("symbolic"-mode) ones for models configured from bases; `QgsTensor`, or the rank-5
    `QgsTensorDynamicT` / `QgsTensorT4` whose closures the reference evaluates with sparse_mul5 / sparse_mul4
    (tendencies.py:98-109) -- here the same `f` / `Df` objects on a rank-5 device model.  Returns a **list**, like
    the reference.
    """
    from qgs_amd.inner_products.analytic import (AtmosphericAnalyticInnerProducts, OceanicAnalyticInnerProducts,
                                                   GroundAnalyticInnerProducts)
    from qgs_amd.inner_products.symbolic import (AtmosphericSymbolicInnerProducts, OceanicSymbolicInnerProducts,
                                                   GroundSymbolicInnerProducts)
    from qgs_amd.tensors.qgtensor import QgsTensor, QgsTensorDynamicT, QgsTensorT4

    def pick(blocks, basis, analytic, symbolic):
        if blocks is not None:
            return analytic(params)
        return symbolic(params) if basis is not None else None

    aip = pick(params.ablocks, params.atmospheric_basis, AtmosphericAnalyticInnerProducts, AtmosphericSymbolicInnerProducts)
    oip = pick(params.oblocks, params.oceanic_basis, OceanicAnalyticInnerProducts, OceanicSymbolicInnerProducts)
    gip = pick(params.gblocks, params.ground_basis, GroundAnalyticInnerProducts, GroundSymbolicInnerProducts)
    if aip is not None and oip is not None:
        if not aip.connected_to_ocean:
            aip.connect_to_ocean(oip)
    elif aip is not None and gip is not None:
        if not aip.connected_to_ground:
            aip.connect_to_ground(gip)

    if params.T4:
        agotensor = QgsTensorT4(params, aip, oip, gip)
    elif params.dynamic_T:
        agotensor = QgsTensorDynamicT(params, aip, oip, gip)
    else:
        agotensor = QgsTensor(params, aip, oip, gip)
    ops = TensorOperands(params.ndim, agotensor.tensor.coords.T, agotensor.tensor.data,
                         agotensor.jacobian_tensor.coords.T, agotensor.jacobian_tensor.data)
    ret = [TendenciesFunction(ops), JacobianFunction(ops)]
    if return_inner_products:
        ret.append((aip, oip, gip))
    if return_qgtensor:
        ret.append(agotensor)
    return ret
