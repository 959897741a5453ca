"""CPU: trajectory output (binary and the reference scripts' text layout) and the on-disk tensor cache."""
import os

import numpy as np

from model_configs import params_rp20


def test_binary_trajectories_round_trip(tmp_path):
    from qgs_amd.io import load_trajectories, save_trajectories
    rng = np.random.RandomState(0)
    time = np.arange(0., 1.05, 0.1)
    traj = rng.rand(7, 20, len(time))
    base = str(tmp_path / 'ens')
    save_trajectories(base, time, traj)
    t2, x2 = load_trajectories(base)
    assert np.array_equal(t2, time) and np.array_equal(np.asarray(x2), traj)
    assert isinstance(x2, np.memmap)
    save_trajectories(base, 1.0, traj[0, :, -1])                 # write_steps = 0: scalar time, one state
    t3, x3 = load_trajectories(base, mmap_mode=None)
    assert float(t3) == 1.0 and np.array_equal(x3, traj[0, :, -1])


def test_text_layout_of_the_reference_scripts(tmp_path):
    """qgs_rp.py:114-131: rows [time, x_1 .. x_n], written with np.savetxt."""
    from qgs_amd.io import save_trajectory_txt
    time = np.array([0., 0.5, 1.0])
    traj = np.arange(12.).reshape(4, 3)
    fn = str(tmp_path / 'evol.dat')
    save_trajectory_txt(fn, time, traj)
    back = np.loadtxt(fn)
    ref = np.insert(traj.T, 0, time, axis=1)
    assert back.shape == (3, 5) and np.array_equal(back, ref)


def test_tensor_cache_is_keyed_by_the_parameter_set(tmp_path):
    from qgs_amd.functions.tendencies import create_tendencies
    from qgs_amd.io import cached_tendencies, params_key
    p = params_rp20()
    f, Df = cached_tendencies(p, str(tmp_path))
    files = os.listdir(str(tmp_path))
    assert files == ['qgs_tensor_%s.npz' % params_key(p)]
    f2, Df2 = cached_tendencies(params_rp20(), str(tmp_path))         # second call: read back, same operands
    assert os.listdir(str(tmp_path)) == files
    f0, Df0 = create_tendencies(p)
    for a, b in ((f2.coo, f0.coo), (f2.val, f0.val), (Df2.coo, Df0.coo), (Df2.val, Df0.val)):
        assert np.array_equal(a, b)
    q = params_rp20()
    q.set_params({'kd': 0.05})
    assert params_key(q) != params_key(p)
    # a miss and a hit return functions bound to the same device
    miss = cached_tendencies(q, str(tmp_path), device=3)
    hit = cached_tendencies(q, str(tmp_path), device=3)
    assert miss[0].device == 3 and miss[1].device == 3 and hit[0].device == 3
    assert params_key(pickle_round_trip(q)) == params_key(q)         # an equal parameter set that took another road hashes equally


def pickle_round_trip(obj):
    import pickle
    return pickle.loads(pickle.dumps(obj))
