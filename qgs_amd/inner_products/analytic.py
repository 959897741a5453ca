"""Analytic inner products of the Fourier basis functions (reference: qgs/inner_products/analytic.py).

Closed-form values of the integrals that couple the spectral modes (Cehelsky & Tung 1987; De Cruz et al.
2016 for the ocean): for the atmosphere a (Laplacian), u (identity), c (d/dx), g (Jacobian), b = g * a_kk,
s/d (coupling to the ocean or ground); for the ocean M, U, N, O, C = O * M_kk, K, W.  The storage is dense
NumPy (the reference stores pydata/sparse COO); values are computed with the reference's operation order
so that they agree bit for bit.  Accessors `a(i,j)`, `b(i,j,k)`, ... keep the reference's names.
"""
import numpy as np

from qgs_amd.basis.fourier import channel_wavenumbers, basin_wavenumbers


def _delta(r):
    return 1. if r == 0 else 0.


def _odd(r):
    """1 for odd r, 0 for even r (the reference's `_flambda`, analytic.py:921-927)."""
    return 0. if r % 2 == 0 else 1.


def _sort_with_parity(seq):
    """Insertion sort returning (sorted list, permutation parity) (analytic.py:880-897)."""
    out = list(seq)
    par = 1
    for i in range(1, len(out)):
        a = out[i]
        j = i - 1
        while j >= 0 and out[j] > a:
            out[j + 1] = out[j]
            par = -par
            j -= 1
        out[j + 1] = a
    return out, par


def _lll_bracket(Ti, Tj, Tk):
    """Selection-rule bracket shared by the atmospheric L-L-L triads and the oceanic triads
    (analytic.py:320-334 and 648-660)."""
    s3 = (Tk.P * Tj.H + Tj.P * Tk.H) / 2.
    s4 = (Tk.P * Tj.H - Tj.P * Tk.H) / 2.
    return s3 * ((_delta(Tk.H - Tj.H - Ti.H) - _delta(Tk.H - Tj.H + Ti.H)) * _delta(Tk.P + Tj.P - Ti.P)
                 + _delta(Tk.H + Tj.H - Ti.H) * (_delta(Tk.P - Tj.P + Ti.P) - _delta(Tk.P - Tj.P - Ti.P))) \
        + s4 * ((_delta(Tk.H + Tj.H - Ti.H) * _delta(Tk.P - Tj.P - Ti.P))
                + (_delta(Tk.H - Tj.H + Ti.H) - _delta(Tk.H - Tj.H - Ti.H))
                * (_delta(Tk.P - Tj.P - Ti.P) - _delta(Tk.P - Tj.P + Ti.P)))


class _Stored(object):
    """Dense arrays with the reference's accessor style."""

    def _get(self, name, idx):
        arr = getattr(self, '_' + name, None)
        if arr is None:
            return 0
        return arr[idx]


class AtmosphericAnalyticInnerProducts(_Stored):
    """Inner products of the channel (atmosphere) modes (analytic.py:48-436).

    `params` is a `QgParams` or the tuple ``(n, ablocks, natm)``.
    """

    def __init__(self, params=None, stored=True):
        if params is None:
            raise ValueError('params is required')
        if hasattr(params, 'scale_params'):
            self.n = float(params.scale_params.n)
            self._natm = params.nmod[0]
            ams = params.ablocks
        else:
            self.n, ams, self._natm = float(params[0]), params[1], params[2]
        self.atmospheric_wavenumbers = channel_wavenumbers(ams)
        self.connected_to_ocean = False
        self.connected_to_ground = False
        self.ocean_inner_products = None
        self.ground_inner_products = None
        self.stored = True
        self._s = self._d = None
        self.compute_inner_products()

    natm = property(lambda self: self._natm)

    # -- closed forms ------------------------------------------------------------------------------
    def _a_comp(self, i, j):
        if i != j:
            return 0
        T = self.atmospheric_wavenumbers[i]
        return - (self.n ** 2) * T.nx ** 2 - T.ny ** 2

    def _u_comp(self, i, j):
        return _delta(i - j)

    def _c_comp(self, i, j):
        Ti, Tj = self.atmospheric_wavenumbers[i], self.atmospheric_wavenumbers[j]
        if (Ti.type, Tj.type) == ('K', 'L'):
            return self.n * Ti.M * (_delta(Ti.M - Tj.H) * _delta(Ti.P - Tj.P))
        if (Ti.type, Tj.type) == ('L', 'K'):
            return - self.n * Tj.M * (_delta(Tj.M - Ti.H) * _delta(Tj.P - Ti.P))
        return 0.

    def _g_comp(self, i, j, k):
        wn = self.atmospheric_wavenumbers
        idx = [i, j, k]
        types = [wn[q].type for q in idx]
        val, par = 0., 1
        if types == ['L', 'L', 'L']:
            order, par = _sort_with_parity(idx)
            val = _lll_bracket(wn[order[0]], wn[order[1]], wn[order[2]])
        elif 'A' in types and 'K' in types and 'L' in types:
            Ti, Tj, Tk = wn[idx[types.index('A')]], wn[idx[types.index('K')]], wn[idx[types.index('L')]]
            _, par = _sort_with_parity(types)
            b1 = (Tk.P + Tj.P) / float(Ti.P)
            b2 = (Tk.P - Tj.P) / float(Ti.P)
            val = -2 * (np.sqrt(2.) / np.pi) * Tj.M * _delta(Tj.M - Tk.H) * _odd(Ti.P + Tj.P + Tk.P)
            if val != 0:
                val = val * (((b1 ** 2) / (b1 ** 2 - 1)) - ((b2 ** 2) / (b2 ** 2 - 1)))
        elif 'A' not in types and types.count('K') == 2:
            _, par = _sort_with_parity(types)
            perm = np.argsort(types)
            Ti, Tj, Tk = wn[idx[perm[0]]], wn[idx[perm[1]]], wn[idx[perm[2]]]      # K, K, L
            s1 = -(Tk.P * Tj.M + Tj.P * Tk.H) / 2.
            s2 = (Tk.P * Tj.M - Tj.P * Tk.H) / 2.
            val = s1 * (_delta(Ti.M - Tk.H - Tj.M) * _delta(Ti.P - Tk.P + Tj.P)
                        - _delta(Ti.M - Tk.H - Tj.M) * _delta(Ti.P + Tk.P - Tj.P)
                        + (_delta(Tk.H - Tj.M + Ti.M) + _delta(Tk.H - Tj.M - Ti.M)) * _delta(Tk.P + Tj.P - Ti.P)) \
                + s2 * (_delta(Ti.M - Tk.H - Tj.M) * _delta(Ti.P - Tk.P - Tj.P)
                        + (_delta(Tk.H - Tj.M - Ti.M) + _delta(Ti.M + Tk.H - Tj.M))
                        * (_delta(Ti.P - Tk.P + Tj.P) - _delta(Tk.P - Tj.P + Ti.P)))
        return val * self.n * par

    def _b_comp(self, i, j, k):
        return self._a_comp(k, k) * self._g_comp(i, j, k)

    def _s_comp(self, i, j):
        if self.connected_to_ocean:
            Ti = self.atmospheric_wavenumbers[i]
            Dj = self.ocean_inner_products.oceanic_wavenumbers[j]
            val = 0.
            if Ti.type == 'A':
                val = _odd(Dj.H) * _odd(Dj.P + Ti.P)
                if val != 0.:
                    val = val * 8 * np.sqrt(2.) * Dj.P / (np.pi ** 2 * (Dj.P ** 2 - Ti.P ** 2) * Dj.H)
            if Ti.type == 'K':
                val = _odd(2 * Ti.M + Dj.H) * _delta(Dj.P - Ti.P)
                if val != 0:
                    val = val * 4 * Dj.H / (np.pi * (-4 * Ti.M ** 2 + Dj.H ** 2))
            if Ti.type == 'L':
                val = _delta(Dj.P - Ti.P) * _delta(2 * Ti.H - Dj.H)
            return val
        if self.connected_to_ground:
            return 1 if i == j else 0
        return 0

    def _d_comp(self, i, j):
        if self.connected_to_ocean:
            return self._s_comp(i, j) * self.ocean_inner_products._M_comp(j, j)
        return 0

    # -- storage -------------------------------------------------------------------------------------
    def compute_inner_products(self):
        n = self.natm
        self._a = np.array([[self._a_comp(i, j) for j in range(n)] for i in range(n)], dtype=float)
        self._u = np.array([[self._u_comp(i, j) for j in range(n)] for i in range(n)], dtype=float)
        self._c = np.array([[self._c_comp(i, j) for j in range(n)] for i in range(n)], dtype=float)
        self._g = np.zeros((n, n, n))
        wn = self.atmospheric_wavenumbers
        for i in range(n):
            for j in range(n):
                for k in range(n):
                    # cheap necessary condition first: an 'A' function can appear at most once
                    if (wn[i].type == 'A') + (wn[j].type == 'A') + (wn[k].type == 'A') > 1:
                        continue
                    self._g[i, j, k] = self._g_comp(i, j, k)
        self._b = self._g * np.diag(self._a)[np.newaxis, np.newaxis, :]      # b_ijk = g_ijk * a_kk

    def connect_to_ocean(self, ocean_inner_products):
        """Compute s, d against the oceanic basis (analytic.py:115-150)."""
        self.ground_inner_products = None
        self.connected_to_ground = False
        self.ocean_inner_products = ocean_inner_products
        self.connected_to_ocean = True
        noc = ocean_inner_products.noc
        self._s = np.array([[self._s_comp(i, j) for j in range(noc)] for i in range(self.natm)], dtype=float)
        self._d = np.array([[self._d_comp(i, j) for j in range(noc)] for i in range(self.natm)], dtype=float)
        if not ocean_inner_products.connected_to_atmosphere:
            ocean_inner_products.connect_to_atmosphere(self)
        self.ocean_inner_products = None

    def connect_to_ground(self, ground_inner_products):
        """s against the ground basis = identity (analytic.py:152-182)."""
        self.ocean_inner_products = None
        self.connected_to_ocean = False
        self.ground_inner_products = ground_inner_products
        self.connected_to_ground = True
        ngr = ground_inner_products.ngr
        self._s = np.array([[self._s_comp(i, j) for j in range(ngr)] for i in range(self.natm)], dtype=float)
        self._d = None
        if not ground_inner_products.connected_to_atmosphere:
            ground_inner_products.connect_to_atmosphere(self)
        self.ground_inner_products = None

    # -- accessors ---------------------------------------------------------------------------------------
    def a(self, i, j): return self._get('a', (i, j))
    def u(self, i, j): return self._get('u', (i, j))
    def c(self, i, j): return self._get('c', (i, j))
    def b(self, i, j, k): return self._get('b', (i, j, k))
    def g(self, i, j, k): return self._get('g', (i, j, k))
    def s(self, i, j): return self._get('s', (i, j))
    def d(self, i, j): return self._get('d', (i, j))


class OceanicAnalyticInnerProducts(_Stored):
    """Inner products of the closed-basin (ocean) modes (analytic.py:439-694)."""

    def __init__(self, params=None, stored=True):
        if params is None:
            raise ValueError('params is required')
        if hasattr(params, 'scale_params'):
            self.n = float(params.scale_params.n)
            self._noc = params.nmod[1]
            oms = params.oblocks
        else:
            self.n, oms, self._noc = float(params[0]), params[1], params[2]
        self.oceanic_wavenumbers = basin_wavenumbers(oms)
        self.connected_to_atmosphere = False
        self.atmosphere_inner_products = None
        self.stored = True
        self._K = self._W = None
        self.compute_inner_products()

    noc = property(lambda self: self._noc)

    def _M_comp(self, i, j):
        if i != j:
            return 0
        D = self.oceanic_wavenumbers[i]
        return - (self.n ** 2) * D.nx ** 2 - D.ny ** 2

    def _U_comp(self, i, j):
        return _delta(i - j)

    def _N_comp(self, i, j):
        Di, Dj = self.oceanic_wavenumbers[i], self.oceanic_wavenumbers[j]
        val = _delta(Di.P - Dj.P) * _odd(Di.H + Dj.H)
        if val != 0:
            val = val * (-2) * Dj.H * Di.H * self.n / ((Dj.H ** 2 - Di.H ** 2) * np.pi)
        return val

    def _O_comp(self, i, j, k):
        order, par = _sort_with_parity([i, j, k])
        wn = self.oceanic_wavenumbers
        return par * _lll_bracket(wn[order[0]], wn[order[1]], wn[order[2]]) * self.n / 2

    def _C_comp(self, i, j, k):
        return self._M_comp(k, k) * self._O_comp(i, j, k)

    def _K_comp(self, i, j):
        if self.connected_to_atmosphere:
            aip = self.atmosphere_inner_products
            return aip._s_comp(j, i) * aip._a_comp(j, j)
        return 0

    def _W_comp(self, i, j):
        if self.connected_to_atmosphere:
            return self.atmosphere_inner_products._s_comp(j, i)
        return 0

    def compute_inner_products(self):
        n = self.noc
        self._M = np.array([[self._M_comp(i, j) for j in range(n)] for i in range(n)], dtype=float)
        self._U = np.array([[self._U_comp(i, j) for j in range(n)] for i in range(n)], dtype=float)
        self._N = np.array([[self._N_comp(i, j) for j in range(n)] for i in range(n)], dtype=float)
        self._O = np.array([[[self._O_comp(i, j, k) for k in range(n)] for j in range(n)] for i in range(n)], dtype=float)
        self._C = self._O * np.diag(self._M)[np.newaxis, np.newaxis, :]      # C_ijk = O_ijk * M_kk

    def connect_to_atmosphere(self, atmosphere_inner_products):
        """K, W against the atmospheric basis (analytic.py:500-528)."""
        self.atmosphere_inner_products = atmosphere_inner_products
        self.connected_to_atmosphere = True
        natm = atmosphere_inner_products.natm
        self._K = np.array([[self._K_comp(i, j) for j in range(natm)] for i in range(self.noc)], dtype=float)
        self._W = np.array([[self._W_comp(i, j) for j in range(natm)] for i in range(self.noc)], dtype=float)
        self.atmosphere_inner_products = None

    def K(self, i, j): return self._get('K', (i, j))
    def M(self, i, j): return self._get('M', (i, j))
    def U(self, i, j): return self._get('U', (i, j))
    def N(self, i, j): return self._get('N', (i, j))
    def O(self, i, j, k): return self._get('O', (i, j, k))
    def C(self, i, j, k): return self._get('C', (i, j, k))
    def W(self, i, j): return self._get('W', (i, j))


class GroundAnalyticInnerProducts(_Stored):
    """Inner products of the ground temperature modes: U = identity, W = s^T (analytic.py:697-877)."""

    def __init__(self, params=None, stored=True):
        if params is None:
            raise ValueError('params is required')
        if hasattr(params, 'scale_params'):
            self.n = float(params.scale_params.n)
            self._ngr = params.nmod[1]
        else:
            self.n, self._ngr = float(params[0]), params[2]
        self.connected_to_atmosphere = False
        self.atmosphere_inner_products = None
        self.stored = True
        self._W = None
        self._U = np.eye(self._ngr)

    ngr = property(lambda self: self._ngr)

    def _U_comp(self, i, j):
        return _delta(i - j)

    def _W_comp(self, i, j):
        if self.connected_to_atmosphere:
            return self.atmosphere_inner_products._s_comp(j, i)
        return 0

    def connect_to_atmosphere(self, atmosphere_inner_products):
        self.atmosphere_inner_products = atmosphere_inner_products
        self.connected_to_atmosphere = True
        natm = atmosphere_inner_products.natm
        self._W = np.array([[self._W_comp(i, j) for j in range(natm)] for i in range(self.ngr)], dtype=float)
        self.atmosphere_inner_products = None

    def K(self, i, j): return 0
    def M(self, i, j): return 0
    def N(self, i, j): return 0
    def O(self, i, j, k): return 0
    def C(self, i, j, k): return 0
    def U(self, i, j): return self._get('U', (i, j))
    def W(self, i, j): return self._get('W', (i, j))
