"""Pin the CPU oracle against golden vectors produced by the imported reference (gen_golden.py).

Float64 oracle vs float64 reference on the same float64 clip table: agreement to ~1e-12
(a few quantities divide tiny differences by 1/120 s or by a small angle; those get 1e-9).
"""
import numpy as np

from conftest import PMC_REWARD_WEIGHTS, make_oracle_batch

TIGHT = dict(rtol=1e-11, atol=1e-12)
OBS = dict(rtol=1e-9, atol=1e-10)
RW = [PMC_REWARD_WEIGHTS[k] for k in ['joint_pos', 'joint_vel', 'end_effector', 'root_pose', 'root_vel']]


def quat_close(a, b, tol):
    a = np.asarray(a); b = np.asarray(b)
    return min(np.abs(a - b).max(), np.abs(a + b).max()) < tol


def test_k1_motionlib_meta(golden, orc, model_blob, mocap_table):
    B = make_oracle_batch(orc, model_blob, mocap_table)
    margin, frame_rate, max_steps = B.meta()
    assert margin == int(golden['k1_margin']) == 125
    assert frame_rate == 120
    assert mocap_table.frame_step == float(golden['k1_frame_step'])
    np.testing.assert_array_equal(mocap_table.clip_len, golden['k1_data_len'])
    np.testing.assert_allclose(max_steps, golden['k1_max_steps'], rtol=1e-15)
    np.testing.assert_allclose(mocap_table.max_steps, golden['k1_max_steps'], rtol=1e-15)
    assert list(mocap_table.names) == [str(n) for n in golden['clip_names']]
    assert mocap_table.frames.dtype == np.float64 and mocap_table.frames.shape == (167076, 19)


def test_g1_mocap_interpolation(golden, orc, mocap_table):
    t = mocap_table
    for k in range(len(golden['g1_clip'])):
        clip = t.clip(int(golden['g1_clip'][k]))
        t0, n = float(golden['g1_t0'][k]), int(golden['g1_n'][k])
        fid, frac = orc.mocap_locate(t0, t.frame_step)
        tt = t0
        for _ in range(n):                      # PLE:208-210: locate at the time BEFORE each increment
            fid, frac = orc.mocap_locate(tt, t.frame_step)
            tt += 0.002
        assert fid == int(golden['g1_frame_id'][k])
        assert abs(frac - float(golden['g1_frac'][k])) < 1e-12
        s = orc.mocap_interp(clip[fid], clip[fid + 1], frac, t.frame_step)
        g = golden['g1_state'][k]
        np.testing.assert_allclose(s[:3], g[:3], **TIGHT)
        assert quat_close(s[3:7], g[3:7], 1e-12)
        np.testing.assert_allclose(s[7:], g[7:], **OBS)
        f = orc.mocap_future(clip[fid:fid + 123], frac, t.frame_step)
        gf = golden['g1_future'][k].reshape(4, 19)
        np.testing.assert_allclose(f[:, 0:3], gf[:, 0:3], **TIGHT)
        for h in range(4):
            assert quat_close(f[h, 3:7], gf[h, 3:7], 1e-12)
        np.testing.assert_allclose(f[:, 13:25], gf[:, 7:19], **TIGHT)
        ended = fid >= len(clip) - t.margin - 1
        assert ended == bool(golden['g1_ended'][k])
    assert golden['g1_ended'].any()


def test_g2_reset_obs(golden, orc, model_blob, mocap_table):
    B = make_oracle_batch(orc, model_blob, mocap_table)
    for k in range(len(golden['g2_seed'])):
        obs = B.reset_env(0, int(golden['g2_clip'][k]), float(golden['g2_t0'][k]))
        np.testing.assert_allclose(obs, golden['g2_obs'][k], **OBS)
        kin = B.get_ref_state(0)
        g = golden['g2_kin'][k]
        assert quat_close(kin[3:7], g[3:7], 1e-12)
        np.testing.assert_allclose(np.delete(kin, [3, 4, 5, 6]), np.delete(g, [3, 4, 5, 6]), **OBS)
        np.testing.assert_array_equal(B.get_state(0), kin)          # PLE:162-163 robot starts on the mocap state
    assert len(set(golden['g2_clip'].tolist())) > 20


def test_k4_reset_walkrun_seed123(golden, orc, model_blob, mocap_table):
    B = make_oracle_batch(orc, model_blob, mocap_table)
    c = mocap_table.names.index('dog_quad_walkrun_001_ret.txt')
    obs = B.reset_env(0, c, float(golden['k4_t0']))
    np.testing.assert_allclose(obs, golden['k4_obs'], **TIGHT)
    assert abs(float(golden['k4_t0']) - 2.4345688415361453) < 1e-15
    assert abs(np.abs(obs[:99]).sum() - 121.0138273226796) < 1e-9
    assert abs(np.abs(obs[135:]).sum() - 42.56634331679517) < 1e-9
    assert np.abs(obs[99:135]).sum() == 0.0


def test_g3_prop_and_future(golden, orc):
    for k in range(len(golden['g3_state'])):
        s = golden['g3_state'][k]
        np.testing.assert_allclose(orc.prop(s), golden['g3_prop'][k], **TIGHT)
        fut = orc.calc_future(s[0:3], s[3:7], golden['g3_future_in'][k])
        np.testing.assert_allclose(fut, golden['g3_future'][k], **OBS)


def test_prop_order_and_subsets(golden, orc):
    s = golden['g3_state'][5]
    full = orc.prop(s, (0, 1, 2, 3, 4))
    assert full.shape == (33,)
    np.testing.assert_array_equal(orc.prop(s, (4, 0)), np.concatenate([full[30:33], full[0:12]]))
    np.testing.assert_array_equal(orc.prop(s, (3,)), full[27:30])


def test_g4_reward_and_terminations(golden, orc):
    for k in range(len(golden['g4_dyn'])):
        d, kn = golden['g4_dyn'][k], golden['g4_kin'][k]
        r = orc.reward(d, kn, golden['g4_feet_dyn'][k], golden['g4_feet_kin'][k], RW)
        assert abs(r - float(golden['g4_reward'][k])) < 1e-12, k
        assert orc.check_fall(d[3:7]) == bool(golden['g4_fall'][k]), k
        assert orc.check_diverged(d, kn) == bool(golden['g4_diverged'][k]), k
    assert golden['g4_fall'].any() and not golden['g4_fall'].all()
    assert golden['g4_diverged'].any() and not golden['g4_diverged'].all()


def test_k3_reward_constants(golden, orc):
    r = orc.reward(golden['k3_dyn'], golden['k3_kin'], np.zeros((4, 3)), np.full((4, 3), 0.01), RW)
    assert abs(r - 0.43603701632970404) < 1e-14
    assert orc.check_diverged(golden['k3_dyn'], golden['k3_kin']) is True
    assert orc.check_fall(golden['k3_dyn'][3:7]) is False


def test_g5_scripted_episodes_and_sampling_table(golden, orc, model_blob, mocap_table):
    """Full step() control flow (history stacking, raw-action history, Q2 phase lag, done, PLE:235-240)."""
    B = make_oracle_batch(orc, model_blob, mocap_table)
    n_done = 0
    for e in range(len(golden['g5_seed'])):
        obs0 = B.reset_env(0, int(golden['g5_clip'][e]), float(golden['g5_t0'][e]))
        np.testing.assert_allclose(obs0, golden['g5_reset_obs'][e], **OBS)
        n = int(golden['g5_n'][e])
        for t in range(n):
            obs, r, d = B.step_env(0, golden['g5_actions'][e, t], scripted_dyn=golden['g5_dyn'][e, t],
                                   feet_dyn=golden['g5_feet_dyn'][e, t], feet_kin=golden['g5_feet_kin'][e, t])
            np.testing.assert_allclose(obs, golden['g5_obs'][e, t], err_msg='ep %d step %d' % (e, t), **OBS)
            assert abs(r - golden['g5_reward'][e, t]) < 1e-11
            assert d == bool(golden['g5_done'][e, t]), (e, t)
        n_done += d
        prob, avg_r, avg_len = B.sampling_table()
        np.testing.assert_allclose(prob, golden['g5_prob_after'][e], rtol=1e-10, atol=1e-14)
        np.testing.assert_allclose(avg_len, golden['g5_avg_len_after'][e], rtol=1e-12)
    assert n_done >= 6


def test_g7_obstacle_extraction(golden, mocap_table):
    """utils/obstacle.py:6-33 (find_peaks + get_obstacle_pose) restated in mocap.MocapTable.obstacles()."""
    from scipy.spatial.transform import Rotation as R
    cnt, tab = mocap_table.obstacles()
    assert cnt.sum() == len(golden['g7_clip']) == 78 and (cnt > 0).sum() == 20
    np.testing.assert_array_equal(np.repeat(np.arange(mocap_table.n_clips), cnt), golden['g7_clip'])
    np.testing.assert_allclose(tab[:, 0:2], golden['g7_pose'][:, 0:2], rtol=0, atol=0)
    np.testing.assert_allclose(tab[:, 3], golden['g7_time'], rtol=1e-15)
    yaw = R.from_quat(golden['g7_pose'][:, 3:7]).as_euler('xyz')[:, 2]
    assert np.abs(np.angle(np.exp(1j * (tab[:, 2] - yaw)))).max() < 1e-12
    np.testing.assert_array_equal(golden['g7_pose'][:, 2], 0.0)           # boxes sit on the ground (obstacle.py:28)
