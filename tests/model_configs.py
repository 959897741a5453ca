"""The four model configurations of the goldens, built with qgs_amd's own QgParams (same calls as
tests/golden/make_golden.py makes on the reference)."""
import numpy as np

from qgs_amd.params.params import QgParams


def params_rp20():
    p = QgParams({'phi0_npi': np.deg2rad(50.) / np.pi, 'hd': 0.1})
    p.set_atmospheric_channel_fourier_modes(2, 2)
    p.ground_params.set_orography(0.2, 1)
    p.atemperature_params.set_thetas(0.2, 0)
    return p


def params_a36():
    p = QgParams({'rr': 287.e0, 'sb': 5.6e-8})
    p.set_atmospheric_channel_fourier_modes(2, 2)
    p.set_oceanic_basin_fourier_modes(2, 4)
    p.set_params({'kd': 0.04, 'kdp': 0.04, 'n': 1.5})
    return p


def params_m36():
    p = QgParams()
    p.set_atmospheric_channel_fourier_modes(2, 2)
    p.set_oceanic_basin_fourier_modes(2, 4)
    p.set_params({'kd': 0.0290, 'kdp': 0.0290, 'n': 1.5, 'r': 1.e-7, 'h': 136.5, 'd': 1.1e-7})
    p.atemperature_params.set_params({'eps': 0.7, 'T0': 289.3, 'hlambda': 15.06, })
    p.gotemperature_params.set_params({'gamma': 5.6e8, 'T0': 301.46})
    p.atemperature_params.set_insolation(103.3333, 0)
    p.gotemperature_params.set_insolation(310., 0)
    return p


def params_t228():
    p = QgParams({'rr': 287.e0, 'sb': 5.6e-8})
    p.set_atmospheric_channel_fourier_modes(6, 6)
    p.set_oceanic_basin_fourier_modes(6, 6)
    p.set_params({'kd': 0.04, 'kdp': 0.04, 'n': 1.5})
    return p


def params_g30():
    p = QgParams({'phi0_npi': np.deg2rad(50.) / np.pi, 'n': 1.3, 'oro_scale': 1})
    p.set_atmospheric_channel_fourier_modes(2, 2)
    p.set_ground_channel_fourier_modes()
    p.ground_params.set_orography(0.2, 1)
    p.gotemperature_params.set_params({'gamma': 1.6e7, 'T0': 300})
    p.atemperature_params.set_params({'hlambda': 10, 'T0': 290})
    p.atmospheric_params.set_params({'sigma': 0.2, 'kd': 0.085, 'kdp': 0.02})
    p.atemperature_params.set_insolation(0.4 * 300., 0)
    p.gotemperature_params.set_insolation(300., 0)
    return p


def _params_notebook_T(**flags):
    """notebooks/maooam_dynamic_temperature.ipynb / maooam_T4.ipynb: MAOOAM 2x2 / 2x4 with dynamic reference
    temperatures (rank-5 tensor); the modes must be set in `symbolic` mode."""
    p = QgParams({'n': 1.5}, **flags)
    p.set_atmospheric_channel_fourier_modes(2, 2, mode="symbolic")
    p.set_oceanic_basin_fourier_modes(2, 4, mode="symbolic")
    p.set_params({'kd': 0.0290, 'kdp': 0.0290, 'r': 1.e-7, 'h': 136.5, 'd': 1.1e-7})
    p.atemperature_params.set_params({'eps': 0.7, 'hlambda': 15.06})
    p.gotemperature_params.set_params({'gamma': 5.6e8})
    p.atemperature_params.set_insolation(103., 0)
    p.atemperature_params.set_insolation(103., 1)
    p.gotemperature_params.set_insolation(310., 0)
    p.gotemperature_params.set_insolation(310., 1)
    return p


def params_d38():
    return _params_notebook_T(dynamic_T=True)


def params_q38():
    return _params_notebook_T(T4=True)


# rank-5 configurations: the reference computes their inner products by numerical quadrature, so its tensors are
# reproduced to its quadrature accuracy instead of bit for bit
MAKERS_RANK5 = {'d38': params_d38, 'q38': params_q38}

MAKERS = {'rp20': params_rp20, 'a36': params_a36, 'm36': params_m36, 't228': params_t228, 'g30': params_g30}


def params_a72():
    """Atmosphere only, 4x4 modes with orography (ndim 72): between the register-resident kernels (ndim <= 64) and the
    MAOOAM 6x6 case, served by the LDS-resident JIT kernels; not a golden configuration (checked against the oracle)."""
    p = QgParams({'phi0_npi': np.deg2rad(50.) / np.pi, 'hd': 0.1})
    p.set_atmospheric_channel_fourier_modes(4, 4)
    p.ground_params.set_orography(0.2, 1)
    p.atemperature_params.set_thetas(0.2, 0)
    return p


def params_d106():
    """Dynamic-temperature MAOOAM at 4x4 / 4x4 resolution (ndim 106, rank-5 tensor): beyond the register-resident kernels,
    served by the LDS-resident JIT kernels with the derived monomials as LDS nodes; checked against the oracle (the
    reference needs ~10 min of quadratures for this model, qgs_amd 1.5 s)."""
    p = QgParams({'n': 1.5}, dynamic_T=True)
    p.set_atmospheric_channel_fourier_modes(4, 4, mode="symbolic")
    p.set_oceanic_basin_fourier_modes(4, 4, mode="symbolic")
    p.set_params({'kd': 0.0290, 'kdp': 0.0290, 'r': 1.e-7, 'h': 136.5, 'd': 1.1e-7})
    p.atemperature_params.set_params({'eps': 0.7, 'hlambda': 15.06})
    p.gotemperature_params.set_params({'gamma': 5.6e8})
    p.atemperature_params.set_insolation(103., 0)
    p.atemperature_params.set_insolation(103., 1)
    p.gotemperature_params.set_insolation(310., 0)
    p.gotemperature_params.set_insolation(310., 1)
    return p
