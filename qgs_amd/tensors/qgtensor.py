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

    def todense(self):
        out = np.zeros(self.shape)
        out[tuple(self.coords)] = self.data
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
        full = self._assemble_dense()
        self.jacobian_tensor = CooTensor(full + np.swapaxes(full, 1, 2))
        self.tensor = CooTensor(self.simplify_dense(full))

    @staticmethod
    def simplify_dense(full):
        """Fold T_ijk with j > k onto T_ikj (upper-triangular in the last two indices)."""
        lower = np.tril(np.ones(full.shape[1:], dtype=bool), -1)
        out = full.copy()
        out += np.swapaxes(np.where(lower[np.newaxis], full, 0.), 1, 2)
        out[:, lower] = 0.
        return out

    def _assemble_dense(self):
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
        T = np.zeros((ndim + 1, ndim + 1, ndim + 1))

        psi = np.arange(natm) + 1                                   # tensor index of psi_a,j
        theta = np.arange(natm) + par.variables_range[0] + 1        # theta_a,j
        kd, kdp, sig0, beta = float(ap.kd), float(ap.kdp), float(ap.sig0), float(scp.beta)

        a, u, c, b, g = aips._a, aips._u, aips._c, aips._b, aips._g
        a_inv = np.linalg.inv(a)
        a_theta = np.linalg.inv(sig0 * a - u)
        eye = np.eye(natm, dtype=int)
        hk = None
        if gp is not None and gp.hk is not None:
            if gp.orographic_basis != "atmospheric":
                raise NotImplementedError('orography on a non-atmospheric basis needs symbolic inner products')
            hk = np.asarray(gp.hk, dtype=float)

        # ---- psi_a equations (qgtensor.py:231-265) -------------------------------------------------
        rows = psi
        v = a_inv @ c                                               # [i, j]
        T[rows[:, None], psi[None, :], 0] -= v * beta
        T[rows[:, None], psi[None, :], 0] -= (kd * eye) / 2
        T[rows[:, None], theta[None, :], 0] = (kd * eye) / 2
        if hk is not None:
            oro = np.einsum('il,ljk->ijk', a_inv, g) @ hk          # a_inv[i,:] @ g[:, j, :] @ hk
            T[rows[:, None], psi[None, :], 0] -= oro / 2
            T[rows[:, None], theta[None, :], 0] += oro / 2
        vb = np.einsum('il,ljk->ijk', a_inv, b)
        T[rows[:, None, None], psi[None, :, None], psi[None, None, :]] = - vb
        T[rows[:, None, None], theta[None, :, None], theta[None, None, :]] = - vb
        if ocean:
            noc = nvar[2]
            psio = np.arange(noc) + par.variables_range[1] + 1
            v = a_inv @ aips._d
            T[rows[:, None], psio[None, :], 0] += v * kd / 2

        # ---- theta_a equations (qgtensor.py:268-338) -------------------------------------------------
        rows = theta
        if par.Cpa is not None:
            T[rows, 0, 0] -= a_theta @ u @ np.asarray(par.Cpa, dtype=float)
        if atp.hd is not None and atp.thetas is not None:
            val = - a_theta @ u @ np.asarray(atp.thetas, dtype=float)
            T[rows, 0, 0] += val * float(atp.hd)
        v = a_theta @ a
        T[rows[:, None], psi[None, :], 0] += v * kd * sig0 / 2
        T[rows[:, None], theta[None, :], 0] -= v * (kd / 2 + 2 * kdp) * sig0
        v = - a_theta @ c
        T[rows[:, None], theta[None, :], 0] += v * beta * sig0
        if hk is not None:
            oro = np.einsum('il,ljk->ijk', a_theta, g) @ hk
            T[rows[:, None], theta[None, :], 0] -= sig0 * oro / 2
            T[rows[:, None], psi[None, :], 0] += sig0 * oro / 2
        vb = np.einsum('il,ljk->ijk', a_theta, b)
        vg = np.einsum('il,ljk->ijk', a_theta, g)
        T[rows[:, None, None], psi[None, :, None], theta[None, None, :]] = - vb * sig0
        T[rows[:, None, None], theta[None, :, None], psi[None, None, :]] = - vb * sig0
        T[rows[:, None, None], psi[None, :, None], theta[None, None, :]] += vg
        v = a_theta @ u
        if par.Lpa is not None:
            T[rows[:, None], theta[None, :], 0] += v * float(atp.sc) * par.Lpa
        if par.LSBpa is not None:
            T[rows[:, None], theta[None, :], 0] += v * par.LSBpa
        if atp.hd is not None:
            T[rows[:, None], theta[None, :], 0] += v * float(atp.hd)
        if ocean:
            v = - a_theta @ aips._d
            T[rows[:, None], psio[None, :], 0] += v * sig0 * kd / 2
        if ocean or ground_temp:
            nsurf = nvar[3] if ocean else nvar[2]
            dT = np.arange(nsurf) + (par.variables_range[2] if ocean else par.variables_range[1]) + 1
        if (ocean or ground_temp) and par.Lpa is not None:
            v = - a_theta @ aips._s
            T[rows[:, None], dT[None, :], 0] += v * par.Lpa / 2
            if par.LSBpgo is not None:
                T[rows[:, None], dT[None, :], 0] += v * par.LSBpgo

        if ocean:
            # ---- psi_o equations (qgtensor.py:342-364) ---------------------------------------------------
            U_inv = np.linalg.inv(bips._U)
            M_psio = np.linalg.inv(bips._M + par.G * bips._U)
            rows = psio
            v = M_psio @ bips._K * float(op.d)
            T[rows[:, None], psi[None, :], 0] += v
            T[rows[:, None], theta[None, :], 0] -= v
            v = - M_psio @ bips._N
            T[rows[:, None], psio[None, :], 0] += v * beta
            v = - M_psio @ bips._M
            T[rows[:, None], psio[None, :], 0] += v * (float(op.r) + float(op.d))
            T[rows[:, None, None], psio[None, :, None], psio[None, None, :]] -= np.einsum('il,ljk->ijk', M_psio, bips._C)

            # ---- delta T_o equations (qgtensor.py:367-389) -----------------------------------------------
            rows = dT
            T[rows, 0, 0] += U_inv @ bips._W @ np.asarray(par.Cpgo, dtype=float)
            v = U_inv @ bips._W
            T[rows[:, None], theta[None, :], 0] += v * 2 * float(atp.sc) * par.Lpgo
            if par.sbpa is not None:
                T[rows[:, None], theta[None, :], 0] += v * par.sbpa
            eye_o = np.eye(noc, dtype=int)
            T[rows[:, None], dT[None, :], 0] = - par.Lpgo * eye_o
            if par.sbpgo is not None:
                T[rows[:, None], dT[None, :], 0] += - par.sbpgo * eye_o
            T[rows[:, None, None], psio[None, :, None], dT[None, None, :]] -= np.einsum('il,ljk->ijk', U_inv, bips._O)

        if ground_temp:
            # ---- delta T_g equations (qgtensor.py:392-409) -----------------------------------------------
            ngr = nvar[2]
            U_inv = np.linalg.inv(bips._U)
            rows = dT
            T[rows, 0, 0] += U_inv @ bips._W @ np.asarray(par.Cpgo, dtype=float)
            v = U_inv @ bips._W
            T[rows[:, None], theta[None, :], 0] += v * 2 * float(atp.sc) * par.Lpgo
            if par.sbpa is not None:
                T[rows[:, None], theta[None, :], 0] += v * par.sbpa
            eye_g = np.eye(ngr, dtype=int)
            T[rows[:, None], dT[None, :], 0] = - par.Lpgo * eye_g
            if par.sbpgo is not None:
                T[rows[:, None], dT[None, :], 0] += - par.sbpgo * eye_g
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
