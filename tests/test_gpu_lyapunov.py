"""GPU: Benettin Lyapunov estimator (qgs_amd/toolbox/lyapunov.py) against goldens captured from the reference's
jitted loops (qgs/toolbox/lyapunov.py:471-632) with the same np.random seed, and the batched QR kernel against
np.linalg.qr."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, rel_err

pytestmark = pytest.mark.gpu


def test_batched_qr_matches_lapack():
    import torch
    from qgs_amd.functions.tendencies import tendencies_from_tensor
    f, _ = tendencies_from_tensor(2, np.array([[1, 0, 1]], dtype=np.int32), np.array([1.0]))
    m = f.hip_model()
    rng = np.random.RandomState(0)
    # shapes: register-resident generated kernels (rows <= 64), the LDS kernel (cols <= 64, rows <= 300), the global-memory kernel
    # (round 5: two generated designs -- four matrices per wavefront with DPP broadcasts where 2 * rows * ceil(cols / 16) registers
    # fit, 16- / 8-member tiles with an LDS broadcast otherwise; member counts around the 4 / 16 / 64 boundaries; column counts that
    # are and are not multiples of 16)
    shapes = ((36, 36, 70), (20, 5, 3), (7, 1, 64), (64, 64, 2), (228, 40, 3), (100, 80, 2), (228, 228, 2), (320, 10, 2),
              (36, 36, 1), (36, 36, 17), (32, 32, 33), (16, 16, 5), (38, 38, 20), (36, 12, 19), (24, 24, 9), (36, 20, 7), (48, 48, 4),
              (40, 40, 18), (12, 12, 130), (33, 17, 6), (2, 2, 4), (1, 1, 3), (228, 40, 9), (100, 36, 70), (228, 64, 3), (300, 64, 2), (65, 1, 5),
              (130, 50, 6), (70, 70, 3), (228, 10, 17), (64, 20, 9), (48, 16, 33), (40, 3, 5),
              # row design with operands parked in accumulation registers or close to the register limit: where the register
              # allocator's copies in front of the DPP instructions showed (44 x 40 was wrong in round 5 before those instructions came
              # in runs behind a wait, tests/test_qr_codegen_cpu.py)
              (44, 40, 37), (64, 20, 33), (60, 30, 37), (56, 20, 34), (38, 34, 35), (64, 32, 33), (64, 16, 21), (64, 40, 33),
              (100, 16, 19), (52, 52, 9), (42, 37, 66),
              # blocked kernel (cols > 64 or rows > 300, up to 400 rows: dgeqrf + dorgqr with 16-column panels): panel counts 1 ... 15, a
              # short last panel, member counts around the XCD mapping; beyond 400 rows the unblocked global-memory kernel
              (228, 228, 9), (300, 200, 3), (97, 65, 4), (400, 17, 3), (80, 80, 17), (129, 128, 2), (401, 70, 2), (500, 3, 2))
    for n_rows, n_cols, n in shapes:
        a = rng.randn(n, n_rows, n_cols)
        ld = (n + 63) // 64 * 64
        d = torch.zeros((n_rows, n_cols, ld), dtype=torch.float64, device='cuda')
        d[:, :, :n] = torch.from_numpy(np.ascontiguousarray(a.transpose(1, 2, 0))).cuda()
        rd = torch.zeros((n_cols, ld), dtype=torch.float64, device='cuda')
        m.batched_qr_device(n, ld, n_rows, n_cols, d.data_ptr(), rd.data_ptr())
        torch.cuda.synchronize()
        expect = ('qgs_spec_qr_%dx%d' % (n_rows, n_cols) if (n_cols <= 64 and n_rows <= 300)
                  else ('batched_qr_blocked_kernel' if n_rows <= 400 else 'batched_qr_global_kernel'))
        if expect:
            assert m.last_kernel_info()['name'] == expect
        q = d[:, :, :n].cpu().numpy().transpose(2, 0, 1)
        r_diag = rd[:, :n].cpu().numpy().T
        for i in range(n):
            qr, rr = np.linalg.qr(a[i])
            assert np.abs(q[i] - qr).max() < 1e-13 and np.abs(r_diag[i] - np.diag(rr)).max() < 1e-13
    f.operands.release()


@pytest.mark.parametrize('name,device', [('rp20', None), ('m36', None), ('d38', None), ('rp20', [0, 0]), ('m36', 'all')])
def test_lyapunov_estimator_vs_reference(name, device):
    """d38: the dynamic-T model (rank-5 tensor); its goldens come from the reference's loops on the reference's own tensor,
    which differs from ours by its quadrature error (2e-14), well inside the tolerances below.  device=[0, 0] / 'all': the
    members sharded over a device list (two shards on the one GPU of the test box), same goldens, same random draws."""
    from model_configs import MAKERS, MAKERS_RANK5
    from qgs_amd.functions.tendencies import create_tendencies
    from qgs_amd.toolbox.lyapunov import LyapunovsEstimator
    g = np.load(os.path.join(GOLDEN_DIR, 'lyap_%s.npz' % name))
    meta = json.loads(bytes(g['meta_json']).decode())
    f, Df = create_tendencies(dict(MAKERS, **MAKERS_RANK5)[name]())
    est = LyapunovsEstimator(num_threads=1, device=device)
    est.set_func(f, Df)
    for cs in meta['cases']:
        np.random.seed(cs['seed'])
        est.compute_lyapunovs(meta['t0'], meta['tw'], meta['t'], meta['dt'], meta['mdt'], ic=g['ic'], write_steps=cs['ws'],
                              n_vec=cs['n_vec'], forward=cs['forward'], adjoint=cs['adjoint'], inverse=cs['inverse'])
        tt, traj, exps, vecs = est.get_lyapunovs()
        tag = cs['tag']
        assert rel_err(traj, np.squeeze(g[tag + '_traj'])) < 1e-12, tag
        assert rel_err(vecs, np.squeeze(g[tag + '_vec'])) < 1e-9, tag
        assert np.abs(exps - np.squeeze(g[tag + '_exp'])).max() < 1e-8 * max(1.0, np.abs(g[tag + '_exp']).max()), tag
        grid = g['pretime'] if cs['forward'] else g['time']
        if cs['ws'] > 0:
            assert np.shape(tt)[0] == traj.shape[-1]
        else:
            assert tt == grid[-1]
    est.terminate()


def test_full_lyapunov_spectrum_at_ndim228():
    """MAOOAM 6x6: the full basis of 228 vectors (the reference's default n_vec = n_dim) -- the QR of a 228 x 228 matrix per
    member runs in the global-memory kernel.  No golden at this size (the reference needs hours); checked: the backward
    vectors are orthonormal and the exponents finite and ordered like a spectrum (largest first on average)."""
    from conftest import load_golden
    from qgs_amd.functions.tendencies import tendencies_from_tensor
    from qgs_amd.toolbox.lyapunov import LyapunovsEstimator
    g = load_golden('t228')
    f, Df = tendencies_from_tensor(g.ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
    est = LyapunovsEstimator(num_threads=1)
    est.set_func(f, Df)
    np.random.seed(5)
    ic = np.random.RandomState(1).rand(2, g.ndim) * 0.01
    est.compute_lyapunovs(0., 0.4, 1.0, 0.1, 0.05, ic=ic, write_steps=2, n_vec=None, forward=False)
    tt, traj, exps, vecs = est.get_lyapunovs()
    assert vecs.shape[:3] == (2, g.ndim, g.ndim) and exps.shape[:2] == (2, g.ndim)
    assert np.isfinite(exps).all() and np.isfinite(vecs).all()
    for i in range(2):
        q = vecs[i, :, :, -1]
        assert np.abs(q.T @ q - np.eye(g.ndim)).max() < 1e-11
    est.terminate()
    f.operands.release()


def test_lyapunov_estimator_at_ndim228_vs_reference():
    """MAOOAM 6x6 against the reference's own Benettin loops (tests/golden/lyap_t228.npz, make_golden.py gen_lyapunov_t228: one
    trajectory, four intervals of two sub-steps): 5 backward vectors, 3 forward vectors, and the full 228-vector spectrum --
    the LDS-resident tangent kernels and the 228 x 228 global-memory QR with the same random start matrices as the reference."""
    from conftest import load_golden
    from qgs_amd.functions.tendencies import tendencies_from_tensor
    from qgs_amd.toolbox.lyapunov import LyapunovsEstimator
    z = np.load(os.path.join(GOLDEN_DIR, 'lyap_t228.npz'))
    meta = json.loads(bytes(z['meta_json']).decode())
    g = load_golden('t228')
    f, Df = tendencies_from_tensor(g.ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
    est = LyapunovsEstimator(num_threads=1)
    est.set_func(f, Df)
    for cs in meta['cases']:
        np.random.seed(cs['seed'])
        est.compute_lyapunovs(meta['t0'], meta['tw'], meta['t'], meta['dt'], meta['mdt'], ic=z['ic'], write_steps=cs['ws'],
                              n_vec=cs['n_vec'], forward=cs['forward'], adjoint=cs['adjoint'], inverse=cs['inverse'])
        tt, traj, exps, vecs = est.get_lyapunovs()
        tag = cs['tag']
        assert rel_err(traj, np.squeeze(z[tag + '_traj'])) < 1e-12, tag
        assert rel_err(vecs, np.squeeze(z[tag + '_vec'])) < 1e-9, tag
        assert np.abs(exps - np.squeeze(z[tag + '_exp'])).max() < 1e-8 * max(1.0, np.abs(z[tag + '_exp']).max()), tag
    est.terminate()
    f.operands.release()


def test_lyapunov_at_ndim228_on_a_device_list():
    """Two shards, one host thread each, at a size whose batched QR needs more than 64 KB of dynamic LDS (228 x 40: 118 KB, the
    limit is raised per device and kernel, generic_kernels.hip DynLdsLimit) and whose tangent model runs in the LDS-resident
    kernels: the sharded run agrees with one model holding both members."""
    from conftest import load_golden
    from qgs_amd.functions.tendencies import tendencies_from_tensor
    from qgs_amd.toolbox.lyapunov import LyapunovsEstimator
    g = load_golden('t228')
    f, Df = tendencies_from_tensor(g.ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
    ic = np.random.RandomState(2).rand(2, g.ndim) * 0.01
    res = []
    for device in (None, [0, 0]):
        est = LyapunovsEstimator(num_threads=1, device=device)
        est.set_func(f, Df)
        np.random.seed(9)
        est.compute_lyapunovs(0., 0.2, 0.6, 0.1, 0.05, ic=ic, write_steps=1, n_vec=40, forward=False)
        res.append(est.get_lyapunovs())
        est.terminate()
    (t0, traj0, exp0, vec0), (t1, traj1, exp1, vec1) = res
    assert vec0.shape == (2, g.ndim, 40, 5) == vec1.shape
    assert np.abs(traj0 - traj1).max() < 1e-13 and np.abs(vec0 - vec1).max() < 1e-10 and np.abs(exp0 - exp1).max() < 1e-9
    for i in range(2):
        q = vec1[i, :, :, -1]
        assert np.abs(q.T @ q - np.eye(40)).max() < 1e-11
    f.operands.release()

