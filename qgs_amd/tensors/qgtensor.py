"""Tendencies tensor (reference: qgs/tensors/qgtensor.py, class QgsTensor).

The model equations are d eta_i/dt = sum_{j,k=0}^{ndim} T_ijk eta_j eta_k with eta_0 = 1.  `QgsTensor`
assembles T from the inner products and the parameters (the reference's `_compute_tensor_dicts`,
qgtensor.py:143-409, for stored inner products and the non-dynamic temperature scheme), then derives

    jacobian_tensor = T + T.swapaxes(1, 2)              (before simplification, qgtensor.py:700-722)
    tensor          = T with (j, k) sorted so j <= k, duplicates merged, zeros pruned (qgtensor.py:724-746)

Both are exposed as `CooTensor` objects with the attributes `create_tendencies` and user code read from
pydata/sparse arrays in the reference: `.coords` (3, nnz) in lexicographic order, `.data`, `.shape`, `.nnz`.

The assembly is written with whole-matrix NumPy products per equation block instead of the reference's
per-row Python loops; every product involves the same diagonal matrices (a^-1, (sig0 a - u)^-1, U^-1,
(M + G U)^-1) so the entries come out bit-identical to the reference's.
"""
import pickle
from contextlib import redirect_stdout

import numpy as np


class CooTensor(object):
    """Minimal COO container: coords (rank, nnz) sorted lexicographically, data (nnz,)."""

    def __init__(self, dense):
        dense = np.asarray(dense, dtype=np.float64)
        self.shape = dense.shape
        nz = np.nonzero(dense)
        self.coords = np.array(nz)
        self.data = dense[nz]

    nnz = property(lambda self: int(self.data.shape[0]))
    ndim = property(lambda self: len(self.shape))

    @classmethod
    def from_coords(cls, coords, data, shape):
        """From a (rank, n) coordinate list with duplicates: entries are summed, zeros dropped, order lexicographic
        (what the pydata/sparse COO constructor does for the reference, qgtensor.py:693)."""
        coords = np.asarray(coords, dtype=np.int64).reshape(len(shape), -1)
        data = np.asarray(data, dtype=np.float64)
        lin = np.ravel_multi_index(tuple(coords), shape)
        order = np.argsort(lin, kind='stable')
        lin, data = lin[order], data[order]
        uniq, start = np.unique(lin, return_index=True)
        summed = np.add.reduceat(data, start) if len(data) else data
        keep = summed != 0.
        out = cls.__new__(cls)
        out.shape = tuple(shape)
        out.coords = np.array(np.unravel_index(uniq[keep], shape))
        out.data = summed[keep]
        return out

    @classmethod
    def sum_of(cls, parts, shape):
        """Element-wise sum ``((parts[0] + parts[1]) + parts[2]) + ...`` of COO operands given as (coords, data) pairs, each with
        unique coordinates: what a chain of binary additions of pydata/sparse arrays gives (an entry absent from an operand
        adds nothing), zeros dropped at the end."""
        lins = [np.ravel_multi_index(tuple(np.asarray(c, dtype=np.int64).reshape(len(shape), -1)), shape) for c, _ in parts]
        uniq = np.unique(np.concatenate(lins)) if lins else np.zeros(0, dtype=np.int64)
        acc = np.zeros(len(uniq))
        for lin, (_, d) in zip(lins, parts):
            acc[np.searchsorted(uniq, lin)] += np.asarray(d, dtype=np.float64)
        return cls.from_sorted_keys(uniq, acc, shape)

    @classmethod
    def from_sorted_keys(cls, keys, data, shape):
        """From linearised, strictly increasing indices: zeros dropped."""
        keep = data != 0.
        out = cls.__new__(cls)
        out.shape = tuple(shape)
        out.coords = np.array(np.unravel_index(keys[keep], shape))
        out.data = np.ascontiguousarray(data[keep])
        return out

    def todense(self):
        out = np.zeros(self.shape)
        out[tuple(self.coords)] = self.data
        return out


class _BlockAccumulator(object):
    """Sparse stand-in for the dense `(ndim+1)^3` array the tensor used to be assembled in (0.1 GB per copy at ndim 228, 8 GB at
    ndim 1000; the reference assembles sparse dictionaries).  The assembly adds whole blocks, `T[rows, cols, k] += values`;
    here a block contributes its non-zero entries as (linear index, value) pairs, and `result()` replays the blocks in
    program order on the union of the touched entries -- every entry sees the same sequence of additions as in the dense
    array, so the sums are bit-identical to it.  (All plain assignments of the assembly are first touches of their block, i.e.
    additions to zero.)"""

    def __init__(self, shape):
        self.shape = tuple(shape)
        self._keys, self._vals = [], []

    def add(self, index, values, sign=1.):
        idx = np.broadcast_arrays(*[np.asarray(q) for q in index])
        vals = np.broadcast_to(np.asarray(values, dtype=np.float64), idx[0].shape)
        nz = vals != 0.
        if not nz.any():
            return
        self._keys.append(np.ravel_multi_index(tuple(q[nz] for q in idx), self.shape))
        self._vals.append(vals[nz] if sign > 0 else -vals[nz])

    def sub(self, index, values):
        self.add(index, values, -1.)

    def result(self):
        """(sorted unique linear indices, summed values)"""
        if not self._keys:
            return np.zeros(0, dtype=np.int64), np.zeros(0)
        uniq = np.unique(np.concatenate(self._keys))
        acc = np.zeros(len(uniq))
        for k, v in zip(self._keys, self._vals):
            pos = np.searchsorted(uniq, k)
            if len(np.unique(pos)) == len(pos):
                acc[pos] += v
            else:                                   # an index twice in one block: sequential accumulation
                np.add.at(acc, pos, v)
        return uniq, acc


def _swap12(keys, shape):
    """linear index of (i, k, j) for linear indices of (i, j, k)"""
    i, j, k = np.unravel_index(keys, shape)
    return np.ravel_multi_index((i, k, j), shape)


def _lookup(keys, vals, query):
    """values of `query` indices in the sparse (sorted keys, vals), 0 where absent"""
    pos = np.searchsorted(keys, query)
    pos_c = np.minimum(pos, max(len(keys) - 1, 0))
    hit = (pos < len(keys)) & (keys[pos_c] == query) if len(keys) else np.zeros(len(query), dtype=bool)
    out = np.zeros(len(query))
    out[hit] = vals[pos_c[hit]]
    return out


class QgsTensor(object):
    """qgs tendencies tensor.

    Parameters: ``QgsTensor(params, atmospheric_inner_products, oceanic_inner_products=None,
    ground_inner_products=None)``.  Attributes: ``params``, ``tensor``, ``jacobian_tensor`` and the three
    inner-products objects.
    """

    def __init__(self, params=None, atmospheric_inner_products=None, oceanic_inner_products=None,
                 ground_inner_products=None):
        self.atmospheric_inner_products = atmospheric_inner_products
        self.oceanic_inner_products = oceanic_inner_products
        self.ground_inner_products = ground_inner_products
        self.params = params
        self.tensor = None
        self.jacobian_tensor = None
        self.compute_tensor()

    # index of a variable in the tensor (0 is the constant slot) -- qgtensor.py:67-141
    def _psi_a(self, i):
        return i + 1

    def _theta_a(self, i):
        return i + self.params.variables_range[0] + 1

    def _psi_o(self, i):
        return i + self.params.variables_range[1] + 1

    def _deltaT_o(self, i):
        return i + self.params.variables_range[2] + 1

    def _deltaT_g(self, i):
        return i + self.params.variables_range[1] + 1

    def compute_tensor(self):
        """Assemble the rank-3 tensor and set `tensor` / `jacobian_tensor`."""
        par = self.params
        if par is None or self.atmospheric_inner_products is None:
            return
        shape = (par.ndim + 1,) * 3
        keys, vals = self._assemble().result()
        self.jacobian_tensor, self.tensor = self.derive(keys, vals, shape)

    @staticmethod
    def derive(keys, vals, shape):
        """(jacobian tensor, simplified tensor) of the un-simplified tensor given as sorted linear indices + values:
        J = T + T.swapaxes(1, 2) (qgtensor.py:700-722); the tensor itself with T_ijk, j > k folded onto T_ikj
        (qgtensor.py:724-746).  Entry by entry the same two-operand sums as on dense arrays."""
        swapped = _swap12(keys, shape)
        union = np.union1d(keys, swapped)
        own = _lookup(keys, vals, union)
        mirror = _lookup(keys, vals, _swap12(union, shape))
        jac = CooTensor.from_sorted_keys(union, own + mirror, shape)
        _, j, k = np.unravel_index(union, shape)
        folded = np.where(j < k, own + mirror, np.where(j == k, own, 0.))
        return jac, CooTensor.from_sorted_keys(union, folded, shape)

    def _assemble(self):
        """Rank-3 part of the tensor as a `_BlockAccumulator` (un-simplified).  With dynamic reference temperatures (`offset` = 1) the temperature fields
        carry a 0-th (constant) mode: their inner products arrays are one larger than the streamfunction ones, and the
        streamfunction equations skip index 0 (qgtensor.py:175-178: `jo = j + offset`)."""
        par = self.params
        aips = self.atmospheric_inner_products
        ocean = self.oceanic_inner_products is not None
        ground_temp = self.ground_inner_products is not None
        bips = self.oceanic_inner_products if ocean else self.ground_inner_products
        atp, ap, op, scp, gp = (par.atemperature_params, par.atmospheric_params, par.oceanic_params,
                                par.scale_params, par.ground_params)
        nvar = par.number_of_variables
        ndim = par.ndim
        natm = nvar[0]
        o = 1 if par.dynamic_T else 0
        T = _BlockAccumulator((ndim + 1, ndim + 1, ndim + 1))

        psi = np.arange(natm) + 1                                   # tensor index of psi_a,j
        theta_all = np.arange(nvar[1]) + par.variables_range[0] + 1  # theta_a,j (j = 0: T_a,0 with dynamic_T)
        theta = theta_all[o:]                                       # ... the ones paired with psi_a,j
        kd, kdp, sig0, beta = float(ap.kd), float(ap.kdp), float(ap.sig0), float(scp.beta)

        a, u, c, b, g = (np.asarray(t) for t in (aips._a, aips._u, aips._c, aips._b, aips._g))
        a_inv = np.linalg.inv(a[o:, o:])
        a_theta = np.linalg.inv(sig0 * a - u)
        eye = np.eye(natm, dtype=int)
        hk = None
        g_oro = g
        if gp is not None and gp.hk is not None:
            if gp.orographic_basis != "atmospheric":
                g_oro = getattr(aips, '_gh', None)
                if g_oro is None:
                    raise NotImplementedError('orography on a non-atmospheric basis needs the symbolic-mode inner products')
            hk = np.asarray(gp.hk, dtype=float)

        # ---- psi_a equations (qgtensor.py:231-265) -------------------------------------------------
        rows = psi
        v = a_inv @ c[o:, o:]                                       # [i, j]
        T.sub((rows[:, None], psi[None, :], 0), v * beta)
        T.sub((rows[:, None], psi[None, :], 0), (kd * eye) / 2)
        T.add((rows[:, None], theta[None, :], 0), (kd * eye) / 2)
        if hk is not None:
            oro = np.einsum('il,ljk->ijk', a_inv, g_oro[o:, o:, o:]) @ hk      # a_inv[i,:] @ g[:, j, :] @ hk
            T.sub((rows[:, None], psi[None, :], 0), oro / 2)
            T.add((rows[:, None], theta[None, :], 0), oro / 2)
        vb = np.einsum('il,ljk->ijk', a_inv, b[o:, o:, o:])
        T.add((rows[:, None, None], psi[None, :, None], psi[None, None, :]), - vb)
        T.add((rows[:, None, None], theta[None, :, None], theta[None, None, :]), - vb)
        if ocean:
            noc = nvar[2]
            psio = np.arange(noc) + par.variables_range[1] + 1
            v = a_inv @ np.asarray(aips._d)[o:, o:]
            T.add((rows[:, None], psio[None, :], 0), v * kd / 2)

        # ---- theta_a equations (qgtensor.py:268-338) -------------------------------------------------
        rows = theta_all
        if par.Cpa is not None:
            T.sub((rows, 0, 0), a_theta @ u @ np.asarray(par.Cpa, dtype=float))
        if atp.hd is not None and atp.thetas is not None:
            val = - a_theta @ u @ np.asarray(atp.thetas, dtype=float)
            T.add((rows, 0, 0), val * float(atp.hd))
        v = a_theta @ a[:, o:]
        T.add((rows[:, None], psi[None, :], 0), v * kd * sig0 / 2)
        T.sub((rows[:, None], theta[None, :], 0), v * (kd / 2 + 2 * kdp) * sig0)
        v = - a_theta @ c[:, o:]
        T.add((rows[:, None], theta[None, :], 0), v * beta * sig0)
        if hk is not None:
            oro = np.einsum('il,ljk->ijk', a_theta, g_oro[:, o:, o:]) @ hk
            T.sub((rows[:, None], theta[None, :], 0), sig0 * oro / 2)
            T.add((rows[:, None], psi[None, :], 0), sig0 * oro / 2)
        vb = np.einsum('il,ljk->ijk', a_theta, b[:, o:, o:])
        vg = np.einsum('il,ljk->ijk', a_theta, g[:, o:, o:])
        T.add((rows[:, None, None], psi[None, :, None], theta[None, None, :]), - vb * sig0)
        T.add((rows[:, None, None], theta[None, :, None], psi[None, None, :]), - vb * sig0)
        T.add((rows[:, None, None], psi[None, :, None], theta[None, None, :]), vg)
        v = a_theta @ u
        if par.Lpa is not None:
            T.add((rows[:, None], theta_all[None, :], 0), v * float(atp.sc) * par.Lpa)
        if par.LSBpa is not None:
            T.add((rows[:, None], theta_all[None, :], 0), v * par.LSBpa)
        if atp.hd is not None:
            T.add((rows[:, None], theta_all[None, :], 0), v * float(atp.hd))
        if ocean:
            v = - a_theta @ np.asarray(aips._d)[:, o:]
            T.add((rows[:, None], psio[None, :], 0), v * sig0 * kd / 2)
        if ocean or ground_temp:
            nsurf = nvar[3] if ocean else nvar[2]
            dT_all = np.arange(nsurf) + (par.variables_range[2] if ocean else par.variables_range[1]) + 1
            dT = dT_all[o:]
        if (ocean or ground_temp) and par.Lpa is not None:
            v = - a_theta @ np.asarray(aips._s)
            T.add((rows[:, None], dT_all[None, :], 0), v * par.Lpa / 2)
            if par.LSBpgo is not None:
                T.add((rows[:, None], dT_all[None, :], 0), v * par.LSBpgo)

        if ocean:
            # ---- psi_o equations (qgtensor.py:342-364) ---------------------------------------------------
            bU, bM, bN, bO, bC, bK, bW = (np.asarray(t) for t in (bips._U, bips._M, bips._N, bips._O, bips._C, bips._K, bips._W))
            U_inv = np.linalg.inv(bU)
            M_psio = np.linalg.inv(bM[o:, o:] + par.G * bU[o:, o:])
            rows = psio
            v = M_psio @ bK[o:, o:] * float(op.d)
            T.add((rows[:, None], psi[None, :], 0), v)
            T.sub((rows[:, None], theta[None, :], 0), v)
            v = - M_psio @ bN[o:, o:]
            T.add((rows[:, None], psio[None, :], 0), v * beta)
            v = - M_psio @ bM[o:, o:]
            T.add((rows[:, None], psio[None, :], 0), v * (float(op.r) + float(op.d)))
            T.sub((rows[:, None, None], psio[None, :, None], psio[None, None, :]), np.einsum('il,ljk->ijk', M_psio, bC[o:, o:, o:]))

            # ---- delta T_o equations (qgtensor.py:367-389) -----------------------------------------------
            rows = dT_all
            T.add((rows, 0, 0), U_inv @ bW @ np.asarray(par.Cpgo, dtype=float))
            v = U_inv @ bW
            T.add((rows[:, None], theta_all[None, :], 0), v * 2 * float(atp.sc) * par.Lpgo)
            if par.sbpa is not None:
                T.add((rows[:, None], theta_all[None, :], 0), v * par.sbpa)
            eye_o = np.eye(nvar[3], dtype=int)
            T.add((rows[:, None], dT_all[None, :], 0), - par.Lpgo * eye_o)
            if par.sbpgo is not None:
                T.add((rows[:, None], dT_all[None, :], 0), - par.sbpgo * eye_o)
            T.sub((rows[:, None, None], psio[None, :, None], dT[None, None, :]), np.einsum('il,ljk->ijk', U_inv, bO[:, o:, o:]))

        if ground_temp:
            # ---- delta T_g equations (qgtensor.py:392-409) -----------------------------------------------
            ngr = nvar[2]
            bU, bW = np.asarray(bips._U), np.asarray(bips._W)
            U_inv = np.linalg.inv(bU)
            rows = dT_all
            T.add((rows, 0, 0), U_inv @ bW @ np.asarray(par.Cpgo, dtype=float))
            v = U_inv @ bW
            T.add((rows[:, None], theta_all[None, :], 0), v * 2 * float(atp.sc) * par.Lpgo)
            if par.sbpa is not None:
                T.add((rows[:, None], theta_all[None, :], 0), v * par.sbpa)
            eye_g = np.eye(ngr, dtype=int)
            T.add((rows[:, None], dT_all[None, :], 0), - par.Lpgo * eye_g)
            if par.sbpgo is not None:
                T.add((rows[:, None], dT_all[None, :], 0), - par.sbpgo * eye_g)
        return T

    # ---- I/O ------------------------------------------------------------------------------------------
    def save_to_file(self, filename, **kwargs):
        with open(filename, 'wb') as f:
            pickle.dump(self.__dict__, f, **kwargs)

    def load_from_file(self, filename, **kwargs):
        with open(filename, 'rb') as f:
            tmp = pickle.load(f, **kwargs)
        self.__dict__.clear()
        self.__dict__.update(tmp)

    @staticmethod
    def _string_format(func, symbol, indices, value):
        if abs(value) >= np.finfo(np.float64).eps:
            func(symbol + "".join("[" + str(i) + "]" for i in indices) + " = % .5E" % value)

    def print_tensor(self, tensor_name=""):
        """Print the non-zero entries as ``name[i][j][k] = value`` (qgtensor.py:779-790)."""
        name = tensor_name or 'QgsTensor'
        for coo, val in zip(self.tensor.coords.T, self.tensor.data):
            self._string_format(print, name, coo, val)

    def print_jacobian_tensor(self, tensor_name=""):
        name = tensor_name or 'QgsTensorJacobian'
        for coo, val in zip(self.jacobian_tensor.coords.T, self.jacobian_tensor.data):
            self._string_format(print, name, coo, val)

    def print_tensor_to_file(self, filename, tensor_name=""):
        """`print_tensor` into the text file `filename` (qgtensor.py:792-804)."""
        with open(filename, 'w') as f:
            with redirect_stdout(f):
                self.print_tensor(tensor_name)

    def print_jacobian_tensor_to_file(self, filename, tensor_name=""):
        """`print_jacobian_tensor` into the text file `filename` (qgtensor.py:819-831)."""
        with open(filename, 'w') as f:
            with redirect_stdout(f):
                self.print_jacobian_tensor(tensor_name)

    # ---- the two derivations as the reference's static helpers, on any COO tensor (rank 3 or 5) ---------
    @staticmethod
    def jacobian_from_tensor(tensor):
        """Jacobian tensor of `tensor` (anything with `.coords` (rank, nnz), `.data`, `.shape`): the tensor plus its copies
        with axis 1 swapped with every later axis, ``T + T.swapaxes(1, 2) [+ T.swapaxes(1, 3) + T.swapaxes(1, 4)]``
        (qgtensor.py:701-722), added one after the other; zeros dropped.  `tensor` has unique coordinates (a COO array);
        returns a `CooTensor`."""
        coords = np.asarray(tensor.coords, dtype=np.int64)
        data = np.asarray(tensor.data, dtype=np.float64)
        parts = [(coords, data)]
        for ax in range(2, len(tensor.shape)):
            sw = coords.copy()
            sw[[1, ax]] = sw[[ax, 1]]
            parts.append((sw, data))
        return CooTensor.sum_of(parts, tuple(tensor.shape))

    @staticmethod
    def simplify_tensor(tensor):
        """Upper-triangularised `tensor`: for every entry the indices after the first are sorted, entries that land on one
        coordinate are summed in their incoming order, zeros dropped (qgtensor.py:725-746); returns a `CooTensor`."""
        coords = np.asarray(tensor.coords, dtype=np.int64).copy()
        coords[1:, :] = np.sort(coords[1:, :], axis=0)
        return CooTensor.from_coords(coords, np.asarray(tensor.data, dtype=np.float64).copy(), tuple(tensor.shape))


class QgsTensorDynamicT(QgsTensor):
    """Tendencies tensor of the models with dynamic reference temperatures (reference: qgtensor.py:843-1170): rank 5,

        d eta_i/dt = sum T_ijklm eta_j eta_k eta_l eta_m ,     eta_0 = 1,

    the rank-3 tensor of `QgsTensor` (entries (i, j, k, 0, 0)) plus the quartic long-wave radiation terms
    sigma_B T^4 of the temperature equations.  With `dynamic_T` only the 0-th mode is kept to fourth order,
    T_0^4 + 4 T_0^3 dT_m (qgtensor.py:1009-1154); `QgsTensorT4` keeps every product.

    `tensor` / `jacobian_tensor` are rank-5 `CooTensor`s with the reference's conventions: the Jacobian tensor is
    T + T.swapaxes(1, 2) + T.swapaxes(1, 3) + T.swapaxes(1, 4) of the un-simplified tensor (qgtensor.py:700-722), the
    tensor itself has its last four indices sorted and duplicates merged (qgtensor.py:724-746).
    """

    def _quartic_entries(self):
        """[(coords (5, n), data)]: the T^4 terms of the temperature equations as un-simplified tensor entries,
        qgtensor.py:916-1007 (the same contraction serves dynamic T and T4: the inner products decide what is kept)."""
        from qgs_amd.inner_products.symbolic import DynTQuartic
        par = self.params
        aips = self.atmospheric_inner_products
        ocean = self.oceanic_inner_products is not None
        bips = self.oceanic_inner_products if ocean else self.ground_inner_products
        nvar = par.number_of_variables
        sig0 = float(par.atmospheric_params.sig0)
        theta = np.arange(nvar[1]) + par.variables_range[0] + 1

        def contract(rows, var, mat, ip, factor):
            """factor * sum_jj mat[i, jj] ip[jj, j, k, l, m] at (rows[i], var[j], var[k], var[l], var[m])"""
            if isinstance(ip, DynTQuartic):
                c, d = DynTQuartic(factor * (mat @ ip.val), ip.shape[1]).entries()
            else:
                block = factor * np.einsum('ij,jklmn->iklmn', mat, np.asarray(ip))
                nz = np.nonzero(block)
                c, d = np.array(nz), block[nz]
            return np.vstack((rows[c[0]], var[c[1]], var[c[2]], var[c[3]], var[c[4]])), d

        out = []
        a_theta = np.linalg.inv(sig0 * np.asarray(aips._a) - np.asarray(aips._u))
        if par.T4LSBpa is not None:
            out.append(contract(theta, theta, a_theta, aips._z, par.T4LSBpa))
        if bips is not None:
            nsurf = nvar[3] if ocean else nvar[2]
            dT = np.arange(nsurf) + (par.variables_range[2] if ocean else par.variables_range[1]) + 1
            if par.T4LSBpgo is not None:
                out.append(contract(theta, dT, a_theta, aips._v, - par.T4LSBpgo))
            U_inv = np.linalg.inv(np.asarray(bips._U))
            out.append(contract(dT, theta, U_inv, bips._Z, par.T4sbpa))
            out.append(contract(dT, dT, U_inv, bips._V, - par.T4sbpgo))
        return out

    def compute_tensor(self):
        par = self.params
        if par is None or self.atmospheric_inner_products is None:
            return
        n1 = par.ndim + 1
        shape = (n1,) * 5
        keys3, vals3 = self._assemble().result()
        keep = vals3 != 0.
        nz = np.unravel_index(keys3[keep], (n1,) * 3)
        coords = [np.vstack((np.array(nz), np.zeros((2, len(nz[0])), dtype=np.int64)))]    # (i, j, k, 0, 0)
        data = [vals3[keep]]
        for c, d in self._quartic_entries():
            coords.append(c)
            data.append(d)
        coords, data = np.hstack(coords), np.concatenate(data)
        # as the reference (qgtensor.py:662-666): the un-simplified entries become one COO array (duplicates summed), from which
        # the Jacobian tensor and the simplified tensor are derived
        raw = CooTensor.from_coords(coords, data, shape)
        self.jacobian_tensor = self.jacobian_from_tensor(raw)
        self.tensor = self.simplify_tensor(raw)


class QgsTensorT4(QgsTensorDynamicT):
    """Tendencies tensor of the models with the full T^4 long-wave radiation terms (reference: qgtensor.py:1173-1363).
    Same assembly as `QgsTensorDynamicT`: the inner products z, v, Z, V then hold every index combination."""
