"""Model parameters (reference: qgs/params/params.py).

`QgParams` and its parameter blocks with the attribute and method names the reference exposes on the
path QgParams -> create_tendencies: scale / atmospheric / temperature / oceanic / ground blocks,
`set_params`, the Fourier mode setters with their documented side effects on the defaults, the derived
nondimensional quantities (`LR, G, Cpgo, Lpgo, Cpa, Lpa, sbpgo, sbpa, LSBpgo, LSBpa`) and the dimension
bookkeeping (`ndim, nmod, variables_range, number_of_variables, var_string`).

Out of scope here: symbolic bases (`set_*_modes` with SymPy bases), dynamic-T / T4
temperature schemes, LaTeX strings and unit pretty-printing.
"""
import pickle
import warnings

import numpy as np

from qgs_amd.params.parameter import Parameter, ParametersArray, ScalingParameter
from qgs_amd.basis.fourier import ChannelFourierBasis, BasinFourierBasis


class Params(object):
    """Base container: `set_params(dict)` re-creates an existing `Parameter` with its own units/flags
    (params.py:75-109); unknown keys are ignored."""

    _name = ""

    def __init__(self, dic=None):
        self.set_params(dic)

    def _assign(self, key, val):
        cur = self.__dict__[key]
        if isinstance(cur, Parameter) and not isinstance(val, Parameter):
            val = Parameter(val, input_dimensional=cur._input_dimensional, units=cur._units,
                            description=cur._description, scale_object=cur._scale_object, symbol=cur._symbol,
                            return_dimensional=cur._return_dimensional)
        elif isinstance(cur, ScalingParameter) and not isinstance(val, ScalingParameter):
            val = ScalingParameter(val, units=cur._units, description=cur._description, symbol=cur._symbol,
                                   dimensional=cur._dimensional)
        self.__dict__[key] = val

    def set_params(self, dic):
        if dic is not None:
            for key, val in dic.items():
                if key in self.__dict__:
                    self._assign(key, val)

    def _list_params(self):
        lines = []
        for key, val in self.__dict__.items():
            if key.startswith('_'):
                continue
            if isinstance(val, (Parameter, ScalingParameter)):
                lines.append("'%s': %s  %s  (%s)" % (key, float(val), val.units, val.description))
            elif isinstance(val, np.ndarray):
                for i, v in enumerate(val):
                    lines.append("'%s'[%d]: %s" % (key, i, float(v)))
        return "\n".join(lines)

    def print_params(self):
        print(self._name + " Parameters:\n" + self._list_params())

    def __str__(self):
        return self._list_params()

    def save_to_file(self, filename, **kwargs):
        with open(filename, 'wb') as f:
            pickle.dump(self.__dict__, f, **kwargs)

    def load_from_file(self, filename, **kwargs):
        with open(filename, 'rb') as f:
            tmp = pickle.load(f, **kwargs)
        self.__dict__.clear()
        self.__dict__.update(tmp)


class ScaleParams(Params):
    """Scales and domain geometry (params.py:193-273)."""
    _name = "Scale"

    def __init__(self, dic=None):
        Params.__init__(self, dic)
        self.scale = ScalingParameter(5.e6, units='[m]', description="space scale of the model, pi times the length L", dimensional=True)
        self.f0 = ScalingParameter(1.032e-4, units='[s^-1]', description="Coriolis parameter at the central latitude of the domain",
                                   dimensional=True)
        self.n = ScalingParameter(1.3e0, description="domain aspect ratio n = 2 Ly / Lx")
        self.rra = ScalingParameter(6370.e3, units='[m]', description="radius of the Earth", dimensional=True)
        self.phi0_npi = ScalingParameter(0.25e0, description="central latitude as a fraction of pi")
        self.deltap = ScalingParameter(5.e4, units='[Pa]', description='pressure difference separating the two layers of the atmosphere',
                                       dimensional=True)
        self.Ha = ScalingParameter(8500., units='[m]', description="mean mid-latitude height of the 500 hPa surface",
                                   dimensional=True)
        self.set_params(dic)

    @property
    def L(self):
        return ScalingParameter(float(self.scale) / np.pi, units='[m]', description='length scale L of the non-dimensionalisation', dimensional=True)

    @property
    def L_y(self):
        return ScalingParameter(float(self.scale), units='[m]', description='north-south extent of the domain',
                                dimensional=True)

    @property
    def L_x(self):
        return ScalingParameter(2 * float(self.scale) / float(self.n), units='[m]',
                                description='east-west extent of the domain', dimensional=True)

    @property
    def phi0(self):
        return ScalingParameter(float(self.phi0_npi) * np.pi, units='[rad]',
                                description="latitude of the centre of the domain", dimensional=True)

    @property
    def beta(self):
        phi0 = float(self.phi0)
        return Parameter(float(self.L) / float(self.rra) * np.cos(phi0) / np.sin(phi0), input_dimensional=False,
                         units='[m^-1][s^-1]', scale_object=self,
                         description="beta: meridional derivative of the Coriolis parameter at phi_0")


class AtmosphericParams(Params):
    """Friction and static stability of the atmosphere (params.py:276-318)."""
    _name = "Atmospheric"

    def __init__(self, scale_params, dic=None):
        Params.__init__(self, dic)
        self._scale_params = scale_params
        self.kd = Parameter(0.1, input_dimensional=False, scale_object=scale_params, units='[s^-1]',
                            description="friction coefficient between the lower atmospheric layer and the surface")
        self.kdp = Parameter(0.01, input_dimensional=False, scale_object=scale_params, units='[s^-1]',
                             description="friction coefficient between the two atmospheric layers")
        self.sigma = Parameter(0.2e0, input_dimensional=False, scale_object=scale_params, units='[m^2][s^-2][Pa^-2]',
                               description="atmospheric static stability")
        self.set_params(dic)

    @property
    def sig0(self):
        return Parameter(float(self.sigma) / 2, input_dimensional=False, scale_object=self._scale_params,
                         units='[m^2][s^-2][Pa^-2]', description="half the atmospheric static stability")


class _SpectralField(object):
    """Helper for the blocks that own a spectral decomposition (`C`, `thetas`, `hk`)."""

    @staticmethod
    def _values(values):
        if hasattr(values, "__iter__"):
            return list(values)
        return int(values) * [0.]

    @staticmethod
    def _warn_no_pos():
        warnings.warn('A scalar value was provided, but without the `pos` argument indicating in which component of '
                      'the spectral decomposition to put it: Spectral decomposition unchanged !'
                      'Please specify it or give a vector as `value`.')


class AtmosphericTemperatureParams(Params, _SpectralField):
    """Atmospheric temperature scheme (params.py:321-469): Newtonian cooling (`hd`, `thetas`) or the
    radiative/heat-exchange scheme (`gamma, C, eps, T0, sc, hlambda`) installed by the ocean/ground setters."""
    _name = "Atmospheric Temperature"

    def __init__(self, scale_params, dic=None):
        Params.__init__(self, dic)
        self._scale_params = scale_params
        self.hd = Parameter(0.045, input_dimensional=False, units='[s]', scale_object=scale_params,
                            description="coefficient of the Newtonian cooling")
        self.thetas = None
        self.gamma = None
        self.C = None
        self.eps = None
        self.T0 = None
        self.sc = None
        self.hlambda = None
        self.dynamic_T = None
        self.set_params(dic)

    def set_insolation(self, value, pos=None, dynamic_T=False):
        """Short-wave radiation C_a: one spectral component (`value`, `pos`) or the whole vector."""
        if isinstance(value, (float, int)) and pos is not None and self.C is not None:
            self.C[pos] = Parameter(value, units='[W][m^-2]', scale_object=self._scale_params, return_dimensional=True)
        elif hasattr(value, "__iter__"):
            if dynamic_T:
                self.dynamic_T = True
            self.C = ParametersArray(self._values(value), units='[W][m^-2]', scale_object=self._scale_params,
                                     return_dimensional=True)
        else:
            self._warn_no_pos()

    def set_thetas(self, value, pos=None):
        """Radiative equilibrium temperature profile theta* (nondimensional)."""
        if isinstance(value, (float, int)) and pos is not None and self.thetas is not None:
            self.thetas[pos] = Parameter(value, scale_object=self._scale_params, return_dimensional=False,
                                         input_dimensional=False)
        elif hasattr(value, "__iter__"):
            self.thetas = ParametersArray(self._values(value), scale_object=self._scale_params,
                                          return_dimensional=False, input_dimensional=False)
        else:
            self._warn_no_pos()


class OceanicParams(Params):
    """Shallow-water ocean layer (params.py:472-510)."""
    _name = "Oceanic"

    def __init__(self, scale_params, dic=None):
        Params.__init__(self, dic)
        self._scale_params = scale_params
        self.gp = Parameter(3.1e-2, units='[m][s^-2]', return_dimensional=True, scale_object=scale_params,
                            description='reduced gravity of the ocean layer')
        self.r = Parameter(1.e-8, units='[s^-1]', scale_object=scale_params,
                           description="friction coefficient at the ocean bottom")
        self.h = Parameter(5.e2, units='[m]', return_dimensional=True, scale_object=scale_params,
                           description="depth of the active ocean layer")
        self.d = Parameter(1.e-8, units='[s^-1]', scale_object=scale_params,
                           description="mechanical coupling strength between ocean and atmosphere")
        self.set_params(dic)


class _SurfaceTemperatureParams(Params, _SpectralField):
    _what = ""

    def __init__(self, scale_params, dic=None):
        Params.__init__(self, dic)
        self._scale_params = scale_params
        self.gamma = Parameter(2.e8, units='[J][m^-2][K^-1]', scale_object=scale_params, return_dimensional=True,
                               description='heat capacity (specific) of the ' + self._what)
        self.C = None
        self.T0 = None
        self.dynamic_T = None
        self.set_params(dic)

    def set_insolation(self, value, pos=None, dynamic_T=False):
        """Short-wave radiation C_go: one spectral component (`value`, `pos`) or the whole vector."""
        if isinstance(value, (float, int)) and pos is not None and self.C is not None:
            self.C[pos] = Parameter(value, units='[W][m^-2]', scale_object=self._scale_params, return_dimensional=True)
        elif hasattr(value, "__iter__"):
            if dynamic_T:
                self.dynamic_T = True
            self.C = ParametersArray(self._values(value), units='[W][m^-2]', scale_object=self._scale_params,
                                     return_dimensional=True)
        else:
            self._warn_no_pos()


class OceanicTemperatureParams(_SurfaceTemperatureParams):
    """Ocean temperature scheme (params.py:513-599)."""
    _name = "Oceanic Temperature"
    _what = "ocean"


class GroundTemperatureParams(_SurfaceTemperatureParams):
    """Ground temperature scheme (params.py:683-771)."""
    _name = "Ground Temperature"
    _what = "ground"


class GroundParams(Params, _SpectralField):
    """Orography (params.py:602-680)."""
    _name = "Ground"

    def __init__(self, scale_params, dic=None):
        Params.__init__(self, dic)
        self._scale_params = scale_params
        self.hk = None
        self.orographic_basis = "atmospheric"
        self.set_params(dic)

    def set_orography(self, value, pos=None, basis="atmospheric"):
        """Spectral orography h_k (nondimensional): one component (`value`, `pos`) or the whole vector."""
        self.orographic_basis = basis
        if isinstance(value, (float, int)) and pos is not None and self.hk is not None:
            self.hk[pos] = Parameter(value, scale_object=self._scale_params, return_dimensional=False,
                                     input_dimensional=False)
        elif hasattr(value, "__iter__"):
            self.hk = ParametersArray(self._values(value), scale_object=self._scale_params, return_dimensional=False,
                                      input_dimensional=False)
        else:
            self._warn_no_pos()


def _spectral_blocks(nxmax, nymax):
    """All (nx, ny) wavenumber blocks up to the truncation, nx-major (params.py:1969-1975)."""
    return np.array([[nx, ny] for nx in range(1, nxmax + 1) for ny in range(1, nymax + 1)], dtype=int)


class QgParams(Params):
    """General qgs parameters container (params.py:774-2063).

    ``QgParams(dic=None, scale_params=None, atmospheric_params=True, atemperature_params=True, oceanic_params=None,
    otemperature_params=None, ground_params=True, gtemperature_params=None, dynamic_T=False, T4=False)``
    """
    _name = "General"

    def __init__(self, dic=None, scale_params=None, atmospheric_params=True, atemperature_params=True,
                 oceanic_params=None, otemperature_params=None, ground_params=True, gtemperature_params=None,
                 dynamic_T=False, T4=False):
        Params.__init__(self, dic)
        self.scale_params = ScaleParams(dic) if scale_params is None else scale_params
        sp = self.scale_params
        self.atmospheric_params = AtmosphericParams(sp, dic=dic) if atmospheric_params is True else atmospheric_params
        # sic: the reference keys the temperature block on `atmospheric_params` (params.py:879-882)
        self.atemperature_params = AtmosphericTemperatureParams(sp, dic=dic) if atmospheric_params is True \
            else atemperature_params
        self.oceanic_params = OceanicParams(sp, dic) if oceanic_params is True else oceanic_params
        self.ground_params = GroundParams(sp, dic) if ground_params is True else ground_params
        # sic: the reference assigns the oceanic temperature block and then overwrites it with the ground one
        # (params.py:893-899), so only `gtemperature_params` survives; the mode setters (auto=True) recreate it.
        self.gotemperature_params = GroundTemperatureParams(sp, dic) if gtemperature_params is True else gtemperature_params

        self._atmospheric_basis = None
        self._oceanic_basis = None
        self._ground_basis = None
        self._number_of_atmospheric_modes = 0
        self._number_of_oceanic_modes = 0
        self._number_of_ground_modes = 0
        self._ams = None
        self._oms = None
        self._gms = None
        # dynamic reference temperatures (a 0-th, constant mode in the temperature fields) and the full T^4 radiative
        # terms: both lead to the rank-5 tensor; T4 forces dynamic_T (params.py:919-923)
        self.dynamic_T = bool(dynamic_T) or bool(T4)
        self.T4 = bool(T4)
        self._atmospheric_var_string = list()
        self._oceanic_var_string = list()
        self._ground_var_string = list()
        self.time_unit = 'days'

        self.rr = Parameter(287.058e0, return_dimensional=True, units='[J][kg^-1][K^-1]', scale_object=sp,
                            description="dry-air gas constant")
        self.sb = Parameter(5.67e-8, return_dimensional=True, units='[J][m^-2][s^-1][K^-4]', scale_object=sp,
                            description="constant of Stefan and Boltzmann")
        self.set_params(dic)

    # ---- derived nondimensional quantities (params.py:946-1076) ----------------------------------
    @staticmethod
    def _try(fn):
        try:
            return fn()
        except Exception:
            return None

    @property
    def LR(self):
        """Reduced Rossby deformation radius."""
        op, scp = self.oceanic_params, self.scale_params
        if op is None:
            return None
        return self._try(lambda: (float(op.gp) * float(op.h)) ** 0.5 / float(scp.f0))

    @property
    def G(self):
        """gamma = -L^2 / L_R^2."""
        if self.LR is None:
            return None
        return self._try(lambda: -float(self.scale_params.L) ** 2 / self.LR ** 2)

    @property
    def Cpgo(self):
        gotp, scp = self.gotemperature_params, self.scale_params
        if gotp is None:
            return None
        return self._try(lambda: np.asarray(gotp.C) / (float(gotp.gamma) * float(scp.f0)) * float(self.rr)
                         / (float(scp.f0) ** 2 * float(scp.L) ** 2))

    @property
    def Lpgo(self):
        atp, gotp, scp = self.atemperature_params, self.gotemperature_params, self.scale_params
        if atp is None or gotp is None:
            return None
        return self._try(lambda: float(atp.hlambda) / (float(gotp.gamma) * float(scp.f0)))

    @property
    def Cpa(self):
        atp, scp = self.atemperature_params, self.scale_params
        if atp is None:
            return None
        return self._try(lambda: np.asarray(atp.C) / (float(atp.gamma) * float(scp.f0)) * float(self.rr)
                         / (float(scp.f0) ** 2 * float(scp.L) ** 2) / 2)

    @property
    def Lpa(self):
        atp, scp = self.atemperature_params, self.scale_params
        if atp is None:
            return None
        return self._try(lambda: float(atp.hlambda) / (float(atp.gamma) * float(scp.f0)))

    @property
    def sbpgo(self):
        if self.dynamic_T:
            return None
        gotp, scp = self.gotemperature_params, self.scale_params
        if gotp is None:
            return None
        return self._try(lambda: 4 * float(self.sb) * float(gotp.T0) ** 3 / (float(gotp.gamma) * float(scp.f0)))

    @property
    def sbpa(self):
        if self.dynamic_T:
            return None
        atp, gotp, scp = self.atemperature_params, self.gotemperature_params, self.scale_params
        if gotp is None or atp is None:
            return None
        return self._try(lambda: 8 * float(atp.eps) * float(self.sb) * float(atp.T0) ** 3 / (float(gotp.gamma) * float(scp.f0)))

    @property
    def LSBpgo(self):
        if self.dynamic_T:
            return None
        atp, gotp, scp = self.atemperature_params, self.gotemperature_params, self.scale_params
        if gotp is None or atp is None:
            return None
        return self._try(lambda: 2 * float(atp.eps) * float(self.sb) * float(gotp.T0) ** 3 / (float(atp.gamma) * float(scp.f0)))

    @property
    def LSBpa(self):
        if self.dynamic_T:
            return None
        atp, scp = self.atemperature_params, self.scale_params
        if atp is None:
            return None
        return self._try(lambda: 8 * float(atp.eps) * float(self.sb) * float(atp.T0) ** 3 / (float(atp.gamma) * float(scp.f0)))

    # T^4 / dynamic-T radiative coefficients (params.py:1078-1129)
    def _t4_coefficient(self, factor, gamma):
        scp = self.scale_params
        return self._try(lambda: factor * float(self.sb) * float(scp.L) ** 6 * float(scp.f0) ** 5
                         / (float(gamma()) * float(self.rr) ** 3))

    @property
    def T4sbpgo(self):
        gotp = self.gotemperature_params
        return None if gotp is None else self._t4_coefficient(1., lambda: gotp.gamma)

    @property
    def T4sbpa(self):
        atp, gotp = self.atemperature_params, self.gotemperature_params
        if gotp is None or atp is None:
            return None
        return self._try(lambda: 16 * float(atp.eps) * float(self.sb) * float(self.scale_params.L) ** 6
                         * float(self.scale_params.f0) ** 5 / (float(gotp.gamma) * float(self.rr) ** 3))

    @property
    def T4LSBpgo(self):
        atp = self.atemperature_params
        if atp is None:
            return None
        return self._try(lambda: 0.5 * float(atp.eps) * float(self.sb) * float(self.scale_params.L) ** 6
                         * float(self.scale_params.f0) ** 5 / (float(atp.gamma) * float(self.rr) ** 3))

    @property
    def T4LSBpa(self):
        atp = self.atemperature_params
        if atp is None:
            return None
        return self._try(lambda: 16 * float(atp.eps) * float(self.sb) * float(self.scale_params.L) ** 6
                         * float(self.scale_params.f0) ** 5 / (float(atp.gamma) * float(self.rr) ** 3))

    @property
    def streamfunction_scaling(self):
        return float(self.scale_params.L) ** 2 * float(self.scale_params.f0)

    @property
    def temperature_scaling(self):
        return self.streamfunction_scaling * float(self.scale_params.f0) / float(self.rr)

    @property
    def geopotential_scaling(self):
        return float(self.scale_params.f0) / 9.81

    @property
    def dimensional_time(self):
        c = 24 * 3600
        if self.time_unit == 'years':
            c *= 365
        return 1 / (float(self.scale_params.f0) * c)

    # ---- parameters I/O ---------------------------------------------------------------------------
    def set_params(self, dic):
        """Set parameters in this container and in every sub-block that knows the key (params.py:1147-1198)."""
        if dic is None:
            return
        Params.set_params(self, dic)
        for name in ('scale_params', 'atmospheric_params', 'atemperature_params', 'oceanic_params', 'ground_params',
                     'gotemperature_params'):
            block = self.__dict__.get(name)
            if block is not None:
                block.set_params(dic)

    def print_params(self):
        print("Qgs v0 parameters summary\n=========================\n")
        for name in ('scale_params', 'atmospheric_params', 'atemperature_params', 'oceanic_params', 'ground_params',
                     'gotemperature_params'):
            block = self.__dict__.get(name)
            if block is not None:
                block.print_params()
                print("")
        Params.print_params(self)

    # ---- dimensions (params.py:1229-1282) ---------------------------------------------------------------
    @property
    def nmod(self):
        if self._number_of_oceanic_modes != 0:
            return [self._number_of_atmospheric_modes, self._number_of_oceanic_modes]
        return [self._number_of_atmospheric_modes, self._number_of_ground_modes]

    @property
    def variables_range(self):
        natm, ngoc = self.nmod
        extra = 1 if self.dynamic_T else 0            # the 0-th temperature modes (params.py:1268-1275)
        vr = [natm, 2 * natm + extra]
        if ngoc > 0:
            vr.append(vr[-1] + ngoc)
            if self._oceanic_basis is not None:
                vr.append(vr[-1] + ngoc)
            vr[-1] += extra
        return vr

    @property
    def ndim(self):
        return self.variables_range[-1]

    @property
    def number_of_variables(self):
        vr = self.variables_range
        return [vr[0]] + [vr[i] - vr[i - 1] for i in range(1, len(vr))]

    @property
    def var_string(self):
        return list(self._atmospheric_var_string + self._oceanic_var_string + self._ground_var_string)

    # ---- bases ---------------------------------------------------------------------------------------------
    # `mode='symbolic'`: the model is configured from basis objects instead of spectral blocks; the spectral-block
    # tables stay None, which is what selects the quadrature inner products in create_tendencies
    # (tendencies.py:57-76).  Required for dynamic_T / T4 (the constant mode has no closed-form inner products).
    @property
    def atmospheric_basis(self):
        return self._atmospheric_basis

    @atmospheric_basis.setter
    def atmospheric_basis(self, basis):
        """params.py:1383-1399"""
        self._ams = self._oms = self._gms = None
        self._atmospheric_basis = basis
        self._number_of_atmospheric_modes = len(basis)
        if self.ground_params is not None and self.ground_params.orographic_basis == "atmospheric":
            self.ground_params.set_orography(self._number_of_atmospheric_modes * [0.e0])
        if self.atemperature_params is not None:
            self.atemperature_params.set_thetas(self._number_of_atmospheric_modes * [0.e0])

    @property
    def oceanic_basis(self):
        return self._oceanic_basis

    @oceanic_basis.setter
    def oceanic_basis(self, basis):
        """params.py:1406-1460"""
        self._ams = self._oms = self._gms = None
        self._oceanic_basis = basis
        self._number_of_ground_modes = 0
        self._number_of_oceanic_modes = len(basis)
        self._heat_exchange_defaults()
        self._surface_insolation_defaults()
        if self.gotemperature_params is not None and self.ground_params is not None:
            self.ground_params.hk = None

    @property
    def ground_basis(self):
        return self._ground_basis

    @ground_basis.setter
    def ground_basis(self, basis):
        """params.py:1467-1527"""
        self._ams = self._oms = self._gms = None
        self._ground_basis = basis
        self._number_of_ground_modes = len(basis)
        self._number_of_oceanic_modes = 0
        self._heat_exchange_defaults()
        if self.gotemperature_params is not None:
            gp = self.ground_params
            if gp is not None and gp.hk is None:
                n = self._number_of_atmospheric_modes if gp.orographic_basis == 'atmospheric' else self._number_of_ground_modes
                gp.set_orography(n * [0.e0], basis=gp.orographic_basis)
                gp.set_orography(0.1, 1, basis=gp.orographic_basis)
        self._surface_insolation_defaults()

    def _surface_insolation_defaults(self):
        gotp = self.gotemperature_params
        if gotp is None:
            return
        if self.dynamic_T:
            gotp.set_insolation((self.nmod[0] + 1) * [0.e0], None, True)
            gotp.set_insolation(350.0, 0, True)
            gotp.set_insolation(350.0, 1, True)
        else:
            gotp.set_insolation(self.nmod[0] * [0.e0])
            gotp.set_insolation(350.0, 0)
            gotp.T0 = Parameter(285.0, units='[K]', scale_object=self.scale_params, return_dimensional=True,
                                description="zeroth-order stationary ocean temperature")

    def set_atmospheric_modes(self, basis, auto=False):
        """Configure the atmosphere from a basis object (params.py:1529-1568)."""
        if auto:
            if self.atemperature_params is None:
                self.atemperature_params = AtmosphericTemperatureParams(self.scale_params)
            if self.atmospheric_params is None:
                self.atmospheric_params = AtmosphericParams(self.scale_params)
        self.atmospheric_basis = basis
        n = self.nmod[0]
        self._atmospheric_var_string = (['psi_a_%d' % (i + 1) for i in range(n)] + (['T_a_0'] if self.dynamic_T else [])
                                        + ['theta_a_%d' % (i + 1) for i in range(n)])

    def set_oceanic_modes(self, basis, auto=True):
        """Configure the ocean from a basis object (params.py:1570-1627)."""
        if self._atmospheric_basis is None:
            print('Atmosphere modes not set up. Add an atmosphere before adding an ocean!')
            print('Oceanic setup aborted.')
            return
        if auto:
            if self.gotemperature_params is None or isinstance(self.gotemperature_params, GroundTemperatureParams):
                self.gotemperature_params = OceanicTemperatureParams(self.scale_params)
            if self.oceanic_params is None:
                self.oceanic_params = OceanicParams(self.scale_params)
            self.ground_params = None
            self._ground_basis = None
        self.oceanic_basis = basis
        n = self.nmod[1]
        self._oceanic_var_string = (['psi_o_%d' % (i + 1) for i in range(n)] + (['T_o_0'] if self.dynamic_T else [])
                                    + ['delta_T_o_%d' % (i + 1) for i in range(n)])
        self._ground_var_string = list()

    def set_ground_modes(self, basis=None, auto=True):
        """Configure the ground from a basis object, default: the atmospheric one (params.py:1629-1681)."""
        if self._atmospheric_basis is None:
            print('Atmosphere modes not set up. Add an atmosphere before adding the ground!')
            print('Ground setup aborted.')
            return
        if auto:
            if self.gotemperature_params is None or isinstance(self.gotemperature_params, OceanicTemperatureParams):
                self.gotemperature_params = GroundTemperatureParams(self.scale_params)
            if self.ground_params is None:
                self.ground_params = GroundParams(self.scale_params)
            self.oceanic_params = None
            self._oceanic_basis = None
        self.ground_basis = basis if basis is not None else self._atmospheric_basis
        self._oceanic_var_string = ['T_g_0'] if self.dynamic_T else []       # sic: the reference files it with the oceanic names
        self._ground_var_string = ['delta_T_g_%d' % (i + 1) for i in range(self.nmod[1])]

    def set_atmospheric_channel_fourier_modes(self, nxmax, nymax, auto=False, mode='analytic'):
        """Fourier modes of the channel atmosphere up to wavenumbers (nxmax, nymax) (params.py:1688-1725)."""
        if mode == 'symbolic':
            return self.set_atmospheric_modes(ChannelFourierBasis(_spectral_blocks(nxmax, nymax), self.scale_params.n), auto)
        self._require_analytic(mode)
        if auto:
            if self.atemperature_params is None:
                self.atemperature_params = AtmosphericTemperatureParams(self.scale_params)
            if self.atmospheric_params is None:
                self.atmospheric_params = AtmosphericParams(self.scale_params)
        self.ablocks = _spectral_blocks(nxmax, nymax)
        n = self.nmod[0]
        self._atmospheric_var_string = ['psi_a_%d' % (i + 1) for i in range(n)] + ['theta_a_%d' % (i + 1) for i in range(n)]

    def set_oceanic_basin_fourier_modes(self, nxmax, nymax, auto=True, mode='analytic'):
        """Fourier modes of the closed-basin ocean (params.py:1727-1768); `auto` creates the ocean blocks and
        removes the ground block."""
        if mode == 'symbolic':
            return self.set_oceanic_modes(BasinFourierBasis(_spectral_blocks(nxmax, nymax), self.scale_params.n), auto)
        self._require_analytic(mode)
        if self._ams is None:
            print('Atmosphere modes not set up. Add an atmosphere before adding an ocean!')
            print('Oceanic setup aborted.')
            return
        if auto:
            if self.gotemperature_params is None or isinstance(self.gotemperature_params, GroundTemperatureParams):
                self.gotemperature_params = OceanicTemperatureParams(self.scale_params)
            if self.oceanic_params is None:
                self.oceanic_params = OceanicParams(self.scale_params)
            self.ground_params = None
        self.oblocks = _spectral_blocks(nxmax, nymax)
        n = self.nmod[1]
        self._oceanic_var_string = ['psi_o_%d' % (i + 1) for i in range(n)] + ['delta_T_o_%d' % (i + 1) for i in range(n)]
        self._ground_var_string = list()

    def set_ground_channel_fourier_modes(self, nxmax=None, nymax=None, auto=True, mode='analytic'):
        """Fourier modes of the ground temperature field (default: the atmospheric ones) (params.py:1770-1819)."""
        if mode == 'symbolic':
            basis = None
            if nxmax is not None and nymax is not None:
                basis = ChannelFourierBasis(_spectral_blocks(nxmax, nymax), self.scale_params.n)
            return self.set_ground_modes(basis, auto)
        self._require_analytic(mode)
        if self._ams is None:
            print('Atmosphere modes not set up. Add an atmosphere before adding the ground!')
            print('Ground setup aborted.')
            return
        blocks = self._ams.copy() if nxmax is None or nymax is None else _spectral_blocks(nxmax, nymax)
        if auto:
            if self.gotemperature_params is None or isinstance(self.gotemperature_params, OceanicTemperatureParams):
                self.gotemperature_params = GroundTemperatureParams(self.scale_params)
            if self.ground_params is None:
                self.ground_params = GroundParams(self.scale_params)
            self.oceanic_params = None
        self.gblocks = blocks
        self._oceanic_var_string = list()
        self._ground_var_string = ['delta_T_g_%d' % (i + 1) for i in range(self.nmod[1])]

    def _require_analytic(self, mode):
        if mode != 'analytic':
            raise ValueError("mode must be 'analytic' or 'symbolic'")
        if self.dynamic_T:
            raise ValueError("dynamic_T / T4 models need mode='symbolic': the constant temperature mode has no "
                             "closed-form inner products (see notebooks/maooam_dynamic_temperature.ipynb)")

    @staticmethod
    def _count_channel_modes(blocks):
        # a block with nx == 1 carries the three functions A, K, L; the others K, L (fourier.py:245-253)
        return int(sum(3 if b[0] == 1 else 2 for b in blocks))

    def _heat_exchange_defaults(self):
        """Side effect shared by the ocean and ground setters (params.py:1866-1891, 1921-1946): replace the
        Newtonian cooling by the radiative + heat exchange scheme with its default coefficients."""
        atp, sp = self.atemperature_params, self.scale_params
        if atp is None:
            return
        atp.thetas = None
        atp.hd = None
        atp.gamma = Parameter(1.e7, units='[J][m^-2][K^-1]', scale_object=sp, return_dimensional=True,
                              description='heat capacity (specific) of the atmosphere')
        if self.dynamic_T:
            atp.set_insolation((self.nmod[0] + 1) * [0.e0], None, True)
            atp.set_insolation(100.0, 0, True)
            atp.set_insolation(100.0, 1, True)
        else:
            atp.set_insolation(self.nmod[0] * [0.e0])
            atp.set_insolation(100.0, 0)
            atp.T0 = Parameter(270.0, units='[K]', scale_object=sp, return_dimensional=True,
                               description="zeroth-order stationary atmospheric temperature")
        atp.eps = Parameter(0.76e0, input_dimensional=False, description="grey-body emissivity of the atmosphere")
        atp.sc = Parameter(1., input_dimensional=False, description="surface-to-atmosphere temperature ratio")
        atp.hlambda = Parameter(20.00, units='[W][m^-2][K^-1]', scale_object=sp, return_dimensional=True,
                                description="sensible and turbulent heat exchange coefficient between the surface (ocean or ground) and the atmosphere")

    @property
    def ablocks(self):
        return self._ams

    @ablocks.setter
    def ablocks(self, value):
        """Atmospheric spectral blocks; installs the default orography h_2 = 0.1 and theta*_1 = 0.1
        (params.py:1830-1850)."""
        self._ams = value
        self._atmospheric_basis = ChannelFourierBasis(self._ams, self.scale_params.n)
        namod = self._count_channel_modes(self._ams)
        self._number_of_atmospheric_modes = namod
        if self.ground_params is not None:
            self.ground_params.orographic_basis = 'atmospheric'
            self.ground_params.set_orography(namod * [0.e0])
            self.ground_params.set_orography(0.1, 1)
        if self.atemperature_params is not None:
            self.atemperature_params.set_thetas(namod * [0.e0])
            self.atemperature_params.set_thetas(0.1, 0)

    @property
    def oblocks(self):
        return self._oms

    @oblocks.setter
    def oblocks(self, value):
        """Oceanic spectral blocks; switches the temperature scheme and disables the orography
        (params.py:1858-1902)."""
        self._oms = value
        self._gms = None
        self._oceanic_basis = BasinFourierBasis(self._oms, self.scale_params.n)
        self._ground_basis = None
        self._heat_exchange_defaults()
        if self.gotemperature_params is not None:
            self._number_of_ground_modes = 0
            self._number_of_oceanic_modes = self._oms.shape[0]
            self.gotemperature_params.set_insolation(self.nmod[0] * [0.e0])     # sic: on the atmospheric basis
            self.gotemperature_params.set_insolation(350.0, 0)
            self.gotemperature_params.T0 = Parameter(285.0, units='[K]', scale_object=self.scale_params,
                                                     return_dimensional=True,
                                                     description="zeroth-order stationary ocean temperature")
            if self.ground_params is not None:
                self.ground_params.hk = None

    @property
    def gblocks(self):
        return self._gms

    @gblocks.setter
    def gblocks(self, value):
        """Ground spectral blocks (params.py:1910-1965)."""
        self._oms = None
        self._gms = value
        self._oceanic_basis = None
        self._ground_basis = ChannelFourierBasis(self._gms, self.scale_params.n)
        self._heat_exchange_defaults()
        if self.gotemperature_params is not None:
            self._number_of_ground_modes = self._count_channel_modes(self._ams[:self._gms.shape[0]])
            self._number_of_oceanic_modes = 0
            if self.ground_params is not None:
                self.ground_params.orographic_basis = 'atmospheric'
                if self.ground_params.hk is None:
                    self.ground_params.set_orography(self.nmod[0] * [0.e0])
                    self.ground_params.set_orography(0.1, 1)
            self.gotemperature_params.set_insolation(self.nmod[0] * [0.e0])
            self.gotemperature_params.set_insolation(350.0, 0)
            self.gotemperature_params.T0 = Parameter(285.0, units='[K]', scale_object=self.scale_params,
                                                     return_dimensional=True,
                                                     description="zeroth-order stationary ocean temperature")
