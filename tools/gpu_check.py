#!/usr/bin/env python3
"""Developer script: run the HIP path against the committed goldens (all stepper cases) and print the
worst relative error per config / kernel family.  The pytest version is tests/test_gpu_parity.py."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qgs_amd import _lib  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def main(names):
    print('backend:', _lib.backend_info())
    for nm in names:
        g = np.load(os.path.join(GOLD, nm + '.npz'))
        meta = json.loads(bytes(g['meta_json']).decode())
        ndim = int(g['ndim'])
        t0 = time.time()
        m = _lib.HipModel(ndim, g['coo'], g['val'], g['jcoo'], g['jval'])
        kinds = [('generic', 1)] + ([('spec', 2)] if m.specialised_available else [])
        for kname, kind in kinds:
            m.set_kernel(kind)
            t1 = time.time()
            f = m.tendencies(g['fx_x'])
            e_f = np.abs(f - g['fx_f']).max() / np.abs(g['fx_f']).max()
            nj = g['fx_Df'].shape[0]
            J = m.jacobian(g['fx_x'][:nj])
            e_J = rel(J, g['fx_Df'])
            worst_rk = (0, '')
            for cs in meta['rk_cases']:
                t = cs['tag']
                ic = g['rk_ic'][:cs['n_traj']]
                rec = m.rk_integrate(g['rk_%s_time' % t], ic, 1 if cs['forward'] else -1, cs['ws'], g['rk_%s_b' % t],
                                     g['rk_%s_c' % t], g['rk_%s_a' % t])
                ref = g['rk_%s_traj' % t]
                assert rec.shape == ref.shape, (t, rec.shape, ref.shape)
                e = rel(rec, ref)
                if e > worst_rk[0]:
                    worst_rk = (e, t)
            worst_tg = (0, '')
            for cs in meta['tgls_cases']:
                t = cs['tag']
                rec, fm = m.rk_tgls_integrate(g['tgls_%s_time' % t], g['tgls_ic'], g['tgls_%s_tgic' % t],
                                              1 if cs['forward'] else -1, cs['ws'], g['tgls_%s_b' % t], g['tgls_%s_c' % t],
                                              g['tgls_%s_a' % t], cs['adjoint'], -1. if cs['inverse'] else 1.)
                e = max(rel(rec, g['tgls_%s_traj' % t]), rel(fm, g['tgls_%s_fm' % t]))
                if e > worst_tg[0]:
                    worst_tg = (e, t)
            print('%-5s %-8s f %.2e  Df %.2e  rk %.2e (%s)  tgls %.2e (%s)  [%.1fs, last kernel %s]' %
                  (nm, kname, e_f, e_J, worst_rk[0], worst_rk[1], worst_tg[0], worst_tg[1], time.time() - t1,
                   m.last_kernel_info()), flush=True)
        m.close()
        print('  total %.1fs' % (time.time() - t0))


if __name__ == '__main__':
    main(sys.argv[1:] or ['rp20', 'a36', 'm36', 't228'])
