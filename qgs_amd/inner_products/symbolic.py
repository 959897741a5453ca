"""Inner products of arbitrary Fourier bases, by quadrature (reference: qgs/inner_products/symbolic.py).

The reference integrates the projections with SymPy -- or, by default (``quadrature=True``), hands each
integrand to ``scipy.integrate.dblquad`` (symbolic.py:1581-1632) -- one inner product at a time, in a
process pool: 50 s for the 38-variable dynamic-temperature MAOOAM, 2 min for its T^4 variant.  Here all inner
products of a model are evaluated at once on one tensor grid of Gauss-Legendre nodes: the modes and their
derivatives are sampled once, every inner product is then a (batched) matrix product over the grid.  The
integrands are trigonometric polynomials, for which Gauss-Legendre with a node count proportional to the
highest total wavenumber converges to rounding error, so the values are exact to ~1e-15 (checked against the
closed-form `analytic` inner products in tests/test_host_model.py); the reference's own quadrature values agree
with them to its error tolerance.

Definitions (qgs/inner_products/definition.py:360-404; F_i atmospheric, phi_i oceanic / ground modes):
    (S, G)  = n / (2 pi^2)  int_0^pi int_0^{2 pi / n} S G dx dy
    a = (F_i, lap F_j)      u = (F_i, F_j)       c = (F_i, d_x F_j)     b = (F_i, J(F_j, lap F_k))   g = (F_i, J(F_j, F_k))
    s = (F_i, phi_j)        d = (F_i, lap phi_j) gh = (F_i, J(F_j, phi_k))
    z = (F_i, F_j F_k F_l F_m)                   v = (F_i, phi_j phi_k phi_l phi_m)
    M = (phi_i, lap phi_j)  U = (phi_i, phi_j)   N = (phi_i, d_x phi_j) O = (phi_i, J(phi_j, phi_k)) C = (phi_i, J(phi_j, lap phi_k))
    K = (phi_i, lap F_j)    W = (phi_i, F_j)     Z = (phi_i, F_j F_k F_l F_m)   V = (phi_i, phi_j phi_k phi_l phi_m)
with J(S, G) = d_x S d_y G - d_x G d_y S.  With dynamic reference temperatures the bases carry the constant
function at index 0 (params.py:1389-1392, 1413-1415); the quartic products are then stored only for
(i, 0, 0, 0, m) and its permutations (symbolic.py:497-502), with the full T^4 scheme for all indices.
"""
import itertools

import numpy as np


class SampledBasis(object):
    """Values, x / y derivatives and Laplacians of a list of Fourier modes on a tensor grid.

    Modes are the `WaveNumber` records of qgs_amd/basis/fourier.py ('A': sqrt(2) cos(P y); 'K': 2 cos(nx n x) sin(P y);
    'L': 2 sin(nx n x) sin(P y), nx = H for the channel and H / 2 for the closed basin); None stands for the
    constant function 1."""

    def __init__(self, modes, n, x, y):
        self.modes = list(modes)
        nf = len(self.modes)
        X, Y = x[np.newaxis, :], y[:, np.newaxis]
        shape = (nf, len(y), len(x))
        self.F, self.Fx, self.Fy, self.lap = (np.zeros(shape) for _ in range(4))
        self.lam = np.zeros(nf)                       # every mode is an eigenfunction of the Laplacian: lap F = lam F
        for q, wn in enumerate(self.modes):
            if wn is None:
                self.F[q] = 1.
            elif wn.type == 'A':
                self.F[q] = np.sqrt(2.) * np.cos(wn.P * Y) + 0. * X
                self.Fy[q] = -np.sqrt(2.) * wn.P * np.sin(wn.P * Y) + 0. * X
                self.lam[q] = -(wn.P ** 2)
            else:
                kx = wn.nx * n
                cx, sx = np.cos(kx * X), np.sin(kx * X)
                sy, cy = np.sin(wn.P * Y), np.cos(wn.P * Y)
                if wn.type == 'K':
                    self.F[q], self.Fx[q], self.Fy[q] = 2. * cx * sy, -2. * kx * sx * sy, 2. * wn.P * cx * cy
                else:
                    self.F[q], self.Fx[q], self.Fy[q] = 2. * sx * sy, 2. * kx * cx * sy, 2. * wn.P * sx * cy
                self.lam[q] = -(kx ** 2 + wn.P ** 2)
            self.lap[q] = self.lam[q] * self.F[q]
        self.F, self.Fx, self.Fy, self.lap = (t.reshape(nf, -1) for t in (self.F, self.Fx, self.Fy, self.lap))

    def __len__(self):
        return len(self.modes)

    @staticmethod
    def max_wavenumbers(modes, n):
        kx = max([0.] + [wn.nx * n for wn in modes if wn is not None and wn.type != 'A'])
        ky = max([0.] + [float(wn.P) for wn in modes if wn is not None])
        return kx, ky


def _grid(mode_lists, n, factors=5):
    """Gauss-Legendre nodes / weights on [0, 2 pi / n] x [0, pi] fine enough for products of `factors` modes."""
    kx = max(SampledBasis.max_wavenumbers(m, n)[0] for m in mode_lists)
    ky = max(SampledBasis.max_wavenumbers(m, n)[1] for m in mode_lists)
    lx, ly = 2. * np.pi / n, np.pi

    def nodes(kmax, length):
        # sin(w t) over a length l: the Gauss-Legendre error decays like (w l / 2)^(2N) / (2N)!
        half = factors * kmax * length / 2.
        N = int(1.5 * half) + 32
        t, w = np.polynomial.legendre.leggauss(N)
        return 0.5 * length * (t + 1.), 0.5 * length * w

    x, wx = nodes(kx, lx)
    y, wy = nodes(ky, ly)
    w = (wy[:, np.newaxis] * wx[np.newaxis, :]).reshape(-1) * n / (2. * np.pi ** 2)
    return x, y, w


def _chop(t, eps=1.e-14):
    """Exact zeros for the structurally vanishing integrals (the reference drops |value| <= its error estimate,
    symbolic.py:1628-1631)."""
    t = np.asarray(t)
    t[np.abs(t) < eps * max(1., float(np.abs(t).max()) if t.size else 1.)] = 0.
    return t


def _jac(A, B):
    """J(A_j, B_k) on the grid: (j, k, g)."""
    return A.Fx[:, None, :] * B.Fy[None, :, :] - B.Fx[None, :, :] * A.Fy[:, None, :]


def _jac_lap(A, B):
    """J(A_j, lap B_k) = lam_k J(A_j, B_k)."""
    return _jac(A, B) * B.lam[None, :, None]


class DynTQuartic(object):
    """Quartic inner products of the dynamic-T scheme: (L_i, R_0^3 R_m), stored for (i, 0, 0, 0, m) and every
    permutation of the last four indices (symbolic.py:497-502) -- kept as the (i, m) matrix instead of an n^5 array
    (37^5 doubles = 555 MB for a 4x4 / 4x4 model).  Indexable like the array it stands for; `todense()` / `np.asarray`
    expand it."""

    def __init__(self, val, n):
        self.val = np.asarray(val, dtype=float)
        self.shape = (self.val.shape[0], n, n, n, n)

    @staticmethod
    def _perms(m):
        return sorted(set(itertools.permutations((0, 0, 0, m))))

    def __getitem__(self, idx):
        i, rest = idx[0], tuple(int(q) for q in idx[1:])
        m = max(rest)
        return self.val[i, m] if sorted(rest) == [0, 0, 0, m] else 0.

    def entries(self):
        """(coords (5, n), data) of the non-zero entries."""
        coords, data = [], []
        for i, m in zip(*np.nonzero(self.val)):
            for perm in self._perms(int(m)):
                coords.append((int(i),) + perm)
                data.append(self.val[i, m])
        return np.array(coords, dtype=np.int64).reshape(-1, 5).T, np.array(data)

    def todense(self):
        out = np.zeros(self.shape)
        c, d = self.entries()
        out[tuple(c)] = d
        return out

    def __array__(self, dtype=None, copy=None):
        return self.todense()


def _quartic(left, w, R, full):
    """(L_i, R_j R_k R_l R_m).  full: every index, dense (T4, symbolic.py:490-495); else only (i, 0, 0, 0, m) and its
    permutations (dynamic T, symbolic.py:497-502), as a `DynTQuartic`."""
    n = len(R)
    Lw = left.F * w[None, :]
    if not full:
        return DynTQuartic(_chop((Lw * (R.F[0] ** 3)[None, :]) @ R.F.T), n)
    out = np.zeros((len(left), n, n, n, n))
    PP = (R.F[:, None, :] * R.F[None, :, :]).reshape(n * n, -1)              # (jk, g)
    for i in range(len(left)):
        out[i] = ((PP * Lw[i][None, :]) @ PP.T).reshape(n, n, n, n)
    return _chop(out)


class _QuadratureInnerProducts(object):
    stored = True
    return_symbolic = False

    def _get(self, name, idx):
        arr = getattr(self, '_' + name)
        return 0 if arr is None else arr[idx]

    def save_to_file(self, filename, **kwargs):
        import pickle
        with open(filename, 'wb') as f:
            pickle.dump(self.__dict__, f, **kwargs)

    def load_from_file(self, filename, **kwargs):
        import pickle
        with open(filename, 'rb') as f:
            tmp = pickle.load(f, **kwargs)
        self.__dict__.clear()
        self.__dict__.update(tmp)


def _modes_of(basis, dynamic_T):
    """Mode list of a basis record; the constant function leads it with dynamic reference temperatures."""
    modes = list(basis.wavenumbers)
    return ([None] + modes) if dynamic_T else modes


class AtmosphericSymbolicInnerProducts(_QuadratureInnerProducts):
    """Atmospheric inner products of a `mode='symbolic'` model (reference class of the same name,
    symbolic.py:36-680).  ``AtmosphericSymbolicInnerProducts(params, stored=True, ...)``; the keyword arguments that
    steer SymPy in the reference (`quadrature`, `timeout`, `num_threads`, `make_substitution`) are accepted and unused.
    """

    def __init__(self, params=None, stored=True, inner_product_definition=None, interaction_inner_product_definition=None,
                 num_threads=None, quadrature=True, timeout=None, dynTinnerproducts=None, T4innerproducts=None,
                 return_symbolic=False, make_substitution=True):
        if return_symbolic:
            raise NotImplementedError('symbolic (SymPy) output of the inner products is out of scope; values only')
        self.n = float(params.scale_params.n)
        self.atmospheric_basis = params.atmospheric_basis
        self.oceanic_basis = self.ground_basis = None
        self.connected_to_ocean = self.connected_to_ground = False
        self._T4 = bool(params.T4) if T4innerproducts is None else bool(T4innerproducts)
        self._dynamic_T = bool(params.dynamic_T) if dynTinnerproducts is None else bool(dynTinnerproducts)
        self._modes = _modes_of(params.atmospheric_basis, params.dynamic_T)
        self._s = self._d = self._v = self._gh = self._z = None
        self.compute_inner_products()

    natm = property(lambda self: len(self._modes))

    def compute_inner_products(self):
        x, y, w = _grid([self._modes], self.n)
        B = SampledBasis(self._modes, self.n, x, y)
        Fw = B.F * w[None, :]
        self._u = _chop(Fw @ B.F.T)
        self._a = _chop(Fw @ B.lap.T)
        self._c = _chop(Fw @ B.Fx.T)
        n = len(B)
        self._g = _chop((Fw @ _jac(B, B).reshape(n * n, -1).T).reshape(n, n, n))
        self._b = _chop((Fw @ _jac_lap(B, B).reshape(n * n, -1).T).reshape(n, n, n))
        if self._T4 or self._dynamic_T:
            self._z = _quartic(B, w, B, self._T4)

    def _connect(self, other_modes, laplacian, orography):
        x, y, w = _grid([self._modes, other_modes], self.n)
        B, P = SampledBasis(self._modes, self.n, x, y), SampledBasis(other_modes, self.n, x, y)
        Fw = B.F * w[None, :]
        self._s = _chop(Fw @ P.F.T)
        self._d = _chop(Fw @ P.lap.T) if laplacian else None
        if orography:
            self._gh = _chop((Fw @ _jac(B, P).reshape(len(B) * len(P), -1).T).reshape(len(B), len(B), len(P)))
        if self._T4 or self._dynamic_T:
            self._v = _quartic(B, w, P, self._T4)

    def connect_to_ocean(self, ocean_basis, num_threads=None, timeout=None):
        """`ocean_basis`: the oceanic inner products object or the basis itself (symbolic.py:214-299)."""
        basis = getattr(ocean_basis, 'oceanic_basis', ocean_basis)
        self.oceanic_basis, self.ground_basis = basis, None
        self.connected_to_ocean, self.connected_to_ground = True, False
        self._connect(_modes_of(basis, self._dynamic_T), True, False)

    def connect_to_ground(self, ground_basis, orographic_basis="atmospheric", num_threads=None, timeout=None):
        """`ground_basis`: the ground inner products object or the basis itself (symbolic.py:301-395)."""
        basis = getattr(ground_basis, 'ground_basis', ground_basis)
        self.ground_basis, self.oceanic_basis = basis, None
        self.connected_to_ground, self.connected_to_ocean = True, False
        self._connect(_modes_of(basis, self._dynamic_T), False, orographic_basis != "atmospheric")

    def a(self, i, j): return self._get('a', (i, j))
    def u(self, i, j): return self._get('u', (i, j))
    def c(self, i, j): return self._get('c', (i, j))
    def b(self, i, j, k): return self._get('b', (i, j, k))
    def g(self, i, j, k): return self._get('g', (i, j, k))
    def gh(self, i, j, k): return self._get('gh', (i, j, k))
    def s(self, i, j): return self._get('s', (i, j))
    def d(self, i, j): return self._get('d', (i, j))
    def z(self, i, j, k, l, m): return self._get('z', (i, j, k, l, m))
    def v(self, i, j, k, l, m): return self._get('v', (i, j, k, l, m))


class OceanicSymbolicInnerProducts(_QuadratureInnerProducts):
    """Oceanic inner products of a `mode='symbolic'` model (symbolic.py:683-1180)."""

    def __init__(self, params=None, stored=True, inner_product_definition=None, interaction_inner_product_definition=None,
                 num_threads=None, quadrature=True, timeout=None, dynTinnerproducts=None, T4innerproducts=None,
                 return_symbolic=False, make_substitution=True):
        if return_symbolic:
            raise NotImplementedError('symbolic (SymPy) output of the inner products is out of scope; values only')
        self.n = float(params.scale_params.n)
        self.oceanic_basis = params.oceanic_basis
        self.atmospheric_basis = None
        self.connected_to_atmosphere = False
        self._T4 = bool(params.T4) if T4innerproducts is None else bool(T4innerproducts)
        self._dynamic_T = bool(params.dynamic_T) if dynTinnerproducts is None else bool(dynTinnerproducts)
        self._modes = _modes_of(params.oceanic_basis, params.dynamic_T)
        self._K = self._W = self._Z = self._V = None
        self.compute_inner_products()
        if params.atmospheric_basis is not None:            # symbolic.py:837-838
            self.connect_to_atmosphere(params.atmospheric_basis)

    noc = property(lambda self: len(self._modes))

    def compute_inner_products(self):
        x, y, w = _grid([self._modes], self.n)
        B = SampledBasis(self._modes, self.n, x, y)
        Fw = B.F * w[None, :]
        n = len(B)
        self._U = _chop(Fw @ B.F.T)
        self._M = _chop(Fw @ B.lap.T)
        self._N = _chop(Fw @ B.Fx.T)
        self._O = _chop((Fw @ _jac(B, B).reshape(n * n, -1).T).reshape(n, n, n))
        self._C = _chop((Fw @ _jac_lap(B, B).reshape(n * n, -1).T).reshape(n, n, n))
        if self._T4 or self._dynamic_T:
            self._V = _quartic(B, w, B, self._T4)

    def connect_to_atmosphere(self, atmosphere_basis, num_threads=None, timeout=None):
        basis = getattr(atmosphere_basis, 'atmospheric_basis', atmosphere_basis)
        self.atmospheric_basis = basis
        self.connected_to_atmosphere = True
        amodes = _modes_of(basis, self._dynamic_T)
        x, y, w = _grid([self._modes, amodes], self.n)
        B, A = SampledBasis(self._modes, self.n, x, y), SampledBasis(amodes, self.n, x, y)
        Fw = B.F * w[None, :]
        self._W = _chop(Fw @ A.F.T)
        self._K = _chop(Fw @ A.lap.T)
        if self._T4 or self._dynamic_T:
            self._Z = _quartic(B, w, A, self._T4)

    def M(self, i, j): return self._get('M', (i, j))
    def U(self, i, j): return self._get('U', (i, j))
    def N(self, i, j): return self._get('N', (i, j))
    def O(self, i, j, k): return self._get('O', (i, j, k))
    def C(self, i, j, k): return self._get('C', (i, j, k))
    def K(self, i, j): return self._get('K', (i, j))
    def W(self, i, j): return self._get('W', (i, j))
    def V(self, i, j, k, l, m): return self._get('V', (i, j, k, l, m))
    def Z(self, i, j, k, l, m): return self._get('Z', (i, j, k, l, m))


class GroundSymbolicInnerProducts(_QuadratureInnerProducts):
    """Ground inner products of a `mode='symbolic'` model (symbolic.py:1183-1560): only U, W (and Z, V) exist."""

    def __init__(self, params=None, stored=True, inner_product_definition=None, interaction_inner_product_definition=None,
                 num_threads=None, quadrature=True, timeout=None, dynTinnerproducts=None, T4innerproducts=None,
                 return_symbolic=False, make_substitution=True):
        if return_symbolic:
            raise NotImplementedError('symbolic (SymPy) output of the inner products is out of scope; values only')
        self.n = float(params.scale_params.n)
        self.ground_basis = params.ground_basis
        self.atmospheric_basis = None
        self.connected_to_atmosphere = False
        self._T4 = bool(params.T4) if T4innerproducts is None else bool(T4innerproducts)
        self._dynamic_T = bool(params.dynamic_T) if dynTinnerproducts is None else bool(dynTinnerproducts)
        self._modes = _modes_of(params.ground_basis, params.dynamic_T)
        self._W = self._Z = self._V = None
        self.compute_inner_products()
        if params.atmospheric_basis is not None:
            self.connect_to_atmosphere(params.atmospheric_basis)

    ngr = property(lambda self: len(self._modes))

    def compute_inner_products(self):
        x, y, w = _grid([self._modes], self.n)
        B = SampledBasis(self._modes, self.n, x, y)
        self._U = _chop((B.F * w[None, :]) @ B.F.T)
        if self._T4 or self._dynamic_T:
            self._V = _quartic(B, w, B, self._T4)

    def connect_to_atmosphere(self, atmosphere_basis, num_threads=None, timeout=None):
        basis = getattr(atmosphere_basis, 'atmospheric_basis', atmosphere_basis)
        self.atmospheric_basis = basis
        self.connected_to_atmosphere = True
        amodes = _modes_of(basis, self._dynamic_T)
        x, y, w = _grid([self._modes, amodes], self.n)
        B, A = SampledBasis(self._modes, self.n, x, y), SampledBasis(amodes, self.n, x, y)
        self._W = _chop((B.F * w[None, :]) @ A.F.T)
        if self._T4 or self._dynamic_T:
            self._Z = _quartic(B, w, A, self._T4)

    def K(self, i, j): return 0
    def M(self, i, j): return 0
    def N(self, i, j): return 0
    def O(self, i, j, k): return 0
    def C(self, i, j, k): return 0
    def U(self, i, j): return self._get('U', (i, j))
    def W(self, i, j): return self._get('W', (i, j))
    def V(self, i, j, k, l, m): return self._get('V', (i, j, k, l, m))
    def Z(self, i, j, k, l, m): return self._get('Z', (i, j, k, l, m))
