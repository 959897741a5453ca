#!/usr/bin/env python3
"""GPU box: the LDS-resident tangent / adjoint kernels of MAOOAM 6x6 (ndim 228), compiler-scheduled (qgs_spec_tgllds16 /
adjlds16) against hand-scheduled (qgs_spec_tglldsa8 / adjldsa8), on the Benettin interval of bench.py's f-row entry:
1 024 members x 228 vectors x 10 sub-steps.  Every variant is a process of its own (the choice is made when the model is
created); the first prints its sampled propagators to a file, the others their largest difference from it.

    python tools/r06_tgllds_ab.py [variant ...]      variant = name:ENV=V,ENV=V     default: the two kernels, twice
"""
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/tmp/r06_tgllds_ref.npz'


def child(tag):
    sys.path.insert(0, HERE)
    import numpy as np
    import torch
    from qgs_amd import _lib
    if os.environ.get('RK_AB_LIB'):                    # another build of the library (developer knobs)
        _lib.LIB_PATH = os.path.abspath(os.environ['RK_AB_LIB'])
    g = np.load(os.path.join(HERE, 'tests', 'golden', 't228.npz'))
    nd = int(g['ndim'])
    dev = torch.device('cuda:0')
    m = _lib.HipModel(nd, g['coo'], g['val'], g['jcoo'], g['jval'], device=0)
    m.set_kernel(2)
    b = np.array([1., 2., 2., 1.]) / 6
    c = np.array([0., .5, .5, 1.])
    a = np.zeros((4, 4)); a[1, 0] = .5; a[2, 1] = .5; a[3, 2] = 1.
    n, nv, steps = 1024, nd, 10
    t = np.arange(steps + 1) * 0.01
    ic = torch.from_numpy(np.random.RandomState(2).rand(nd, n) * 0.01).to(dev)
    q = torch.zeros((nd, nv, n), dtype=torch.float64, device=dev)
    for d in range(nd):
        q[d, d, :] = 1.0
    q += 0.01 * torch.from_numpy(np.random.RandomState(3).randn(nd, nv, 1)).to(dev)
    qn = torch.empty((1, nd, nv, n), dtype=torch.float64, device=dev)
    yend = torch.empty((1, nd, n), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    out = {}
    prof = bool(os.environ.get('R06_PROF'))             # under the profiler: the tangent kernel only, three launches
    for adj, inv, d in (((False, 1., 1),) if prof else ((False, 1., 1), (True, -1., -1))):
        def run():
            m.rk_tgls_integrate_device(n, n, nv, ic.data_ptr(), q.data_ptr(), t, d, 0, b, c, a, adj, inv, yend.data_ptr(), qn.data_ptr(), st)
        t0 = time.time()
        run()
        torch.cuda.synchronize()
        first = time.time() - t0
        name = m.last_kernel_info()['name']
        info = m.last_kernel_info()
        ms = []
        for _ in range(0 if prof else 3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1) / 3)
        sel = torch.tensor([0, 15, 16, 63, 64, 1000, 1023], device=dev)
        out['adj' if adj else 'tgl'] = qn[0][:, :, sel].cpu().numpy()
        if prof:
            run(); run()
            torch.cuda.synchronize()
        print('%-28s %-20s first call %6.2f s   ms per call (trajectory pass included): %s   vgpr %s scratch %s' % (
            tag, name, first, ' '.join('%.2f' % v for v in ms), info.get('vgprs'), info.get('scratch_bytes')), flush=True)
    if prof:
        m.close()
        return
    if not os.path.exists(REF):
        np.savez(REF, **out)
    else:
        r = np.load(REF)
        for k in out:
            den = np.abs(r[k]).max()
            print('%-28s %s: max|diff| vs first variant / max|value| = %.3e' % (tag, k, np.abs(out[k] - r[k]).max() / den), flush=True)
    m.close()


def main():
    if len(sys.argv) > 2 and sys.argv[1] == '--child':
        return child(sys.argv[2])
    variants = sys.argv[1:] or ['compiler-scheduled:QGS_HIP_LDS_TGL_ASM=0', 'hand-scheduled:QGS_HIP_LDS_TGL_ASM=1',
                                'compiler-scheduled:QGS_HIP_LDS_TGL_ASM=0', 'hand-scheduled:QGS_HIP_LDS_TGL_ASM=1']
    if os.path.exists(REF):
        os.remove(REF)
    for v in variants:
        tag, _, envs = v.partition(':')
        env = dict(os.environ)
        for kv in envs.split(','):
            if kv:
                k, _, val = kv.partition('=')
                env[k] = val
        subprocess.run([sys.executable, os.path.abspath(__file__), '--child', tag + ' ' + envs], env=env)


if __name__ == '__main__':
    main()
