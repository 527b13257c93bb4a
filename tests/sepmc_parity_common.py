"""SEPMC parity checks shared by the oracle test, the CPU run of the kernel source (tests/emul) and the GPU run of libllenv.so:
(a) the reference's own outputs in tests/golden/sepmc_golden.npz, (b) the float64 NumPy oracle (oracle/sepmc_oracle.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import sepmc_oracle as SO  # noqa: E402
from oracle import epmc_oracle as EO  # noqa: E402

TOL = 1e-9
NOISE = {'pos_x_bias': [-0.1, 0.1], 'pos_y_bias': [-0.1, 0.1], 'yaw_bias': [-0.2, 0.2], 'pos_z_bias': [-0.02, 0.02]}
N_FULL = 3


def load_golden():
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'sepmc_golden.npz'))


def env_config(elements, noisy=False, max_steps=1000):      # the dict gen_sepmc_golden.py passed to the reference
    return {
        'arena_id': 'CTG', 'render': False, 'control_freq': 50.0,
        'prop_type': ['joint_pos', 'joint_vel', 'root_ang_vel_loc', 'root_lin_vel_loc', 'e_g'],
        'kp': 50.0, 'kd': 0.5, 'max_tau': 16, 'max_steps': int(max_steps), 'obs_randomization': dict(NOISE) if noisy else {},
        'env_randomize_config': {
            'friction_range': [0.4, 3.0],
            'disturb_force_config': {'start_time': 0.5, 'interval_time': 1.0, 'duration_time': 0.2, 'horizontal_force': [0, 50], 'vertical_force': [0, 10]},
        },
        'element_config': {'rand_cube': bool(elements[0]), 'hurdle': bool(elements[1]), 'hole': bool(elements[2])},
    }


def scripted_rays(call, n):          # the fake client's rayTestBatch answers (same arithmetic as gen_epmc_golden.scripted_rays)
    i = np.arange(n)
    return ((i * 7 + call * 13) % 10) < 7, (((i * 37 + call * 101) % 1009) + 0.5) / 1009.0


def vis_blocked(drill, slot):        # the fake client's rayTest answers (gen_sepmc_golden.vis_blocked)
    if drill % 4 == 0:
        return True
    if drill % 4 == 1:
        return slot != 0
    if drill % 4 == 2:
        return slot != 1 + (drill % 10)
    return slot != 11 + ((drill * 3) % 10)


def drill_rays(drill, robot):
    """(hit[778], frac[778]) of one robot in the engine's / oracle's ray order (height, horizontal, front); the reference casts
    front r0, front r1, height r0, height r1, fan r0, fan r1 per drill (CTG:515-531)."""
    hits, fracs = [], []
    for call, n in ((6 * drill + 2 + robot, 325), (6 * drill + 4 + robot, 128), (6 * drill + robot, 325)):
        h, f = scripted_rays(call, n)
        hits.append(h); fracs.append(f)
    return np.concatenate(hits), np.concatenate(fracs)


def percep_checks(obs):
    out = []
    for a, b in ((135, 460), (460, 588), (588, 913)):
        v = obs[a:b]
        i = np.arange(len(v))
        out += [v.sum(), (v * (i + 1)).sum() / len(v), (v * np.where(i % 2 == 0, 1.0, -1.0)).sum()]
    return np.array(out)


def core_of(obs):
    return np.concatenate([obs[:135], obs[913:]])


def contacts_of(rows):
    return [tuple(int(x) for x in r) for r in rows if r[0] > -8.5]


def make_oracle(g, model, elements, noisy, prev_orn, max_steps=1000):
    init = g['init_states_info'].copy()
    init[3:7] = prev_orn
    return SO.SepmcOracleEnv(env_config(elements, noisy, max_steps), init, model)


# ------------------------------------------------------------------------------------------------ oracle vs goldens
def check_oracle_reset_case(g, model, k):
    env = make_oracle(g, model, g['t_elements'][k], bool(g['t_noise_on'][k]), g['t_prev_orn'][k])
    draws = SO.LoggedDraws(g['t_draws'][k][:g['t_n_draws'][k]])
    obs = env.reset(draws, rays=lambda r, f, t: drill_rays(0, r), vis=lambda s, f, t: vis_blocked(0, s), contacts=lambda: [])
    assert draws.exhausted()
    n = g['t_n_boxes'][k]
    assert len(env.statics) == n
    np.testing.assert_allclose(env.statics, g['t_boxes'][k][:n], rtol=0, atol=1e-12)
    np.testing.assert_allclose(env.target_pos, g['t_flag'][k], atol=1e-12)
    assert env.with_flag[0] == bool(g['t_with_flag'][k])
    assert abs(env.foot_friction - g['t_friction'][k]) < 1e-12 and abs(env.episodic_fix_spd - g['t_fix_spd'][k]) < 1e-12
    np.testing.assert_allclose(np.array(env.states), g['t_state'][k], atol=1e-12)
    np.testing.assert_allclose(env.push.curr_force, g['t_push'][k], atol=1e-12)
    assert abs(env.last_two_rob_pos_diff_len - g['t_last_two'][k]) < 1e-12 and abs(env.last_esc_flag_pos_diff_len - g['t_last_esc'][k]) < 1e-12
    assert list(env.oppo_visible) == list(g['t_vis'][k])
    if g['t_noise_on'][k]:
        np.testing.assert_allclose([env.noise[key] for key in NOISE], g['t_noise'][k], atol=1e-12)
    np.testing.assert_allclose(np.array(obs), g['t_obs'][k], rtol=0, atol=TOL)


def check_oracle_episode(g, model, e):
    n = int(g['e_n'][e])
    env = make_oracle(g, model, g['e_elements'][e], bool(g['e_noise_on'][e]), g['e_prev_orn'][e], g['e_max_steps'][e])
    draws = SO.LoggedDraws(g['e_draws'][e][:g['e_n_draws'][e]])
    drill = [0]
    kw = dict(rays=lambda r, f, t: drill_rays(drill[0], r), vis=lambda s, f, t: vis_blocked(drill[0], s))
    obs = env.reset(draws, contacts=lambda: [], **kw)
    np.testing.assert_allclose(np.array(obs), g['e_reset_obs'][e], rtol=0, atol=TOL)
    np.testing.assert_allclose(np.array(env.states), g['e_init_state'][e], atol=1e-12)
    vis_rows = g['e_vis'][e][:g['e_n_vis'][e]]

    def check_rays(d):
        for r in range(2):
            np.testing.assert_allclose(env.ray_ends[r][0], g['e_ray_from'][e][d][r], atol=1e-9)
            np.testing.assert_allclose(env.ray_ends[r][1], g['e_ray_to'][e][d][r], atol=1e-9)
        want = vis_rows[vis_rows[:, 0] == d]
        assert [s for (s, _, _) in env.vis_log] == [int(x) for x in want[:, 1]]            # same rayTest calls in the same order (early exits)
        for (s, f, t), w in zip(env.vis_log, want):
            np.testing.assert_allclose(np.concatenate([f, t]), w[2:8], atol=1e-9)
    check_rays(0)
    for t in range(n):
        drill[0] = t + 1
        st = [g['e_state'][e][0][t], g['e_state'][e][1][t]]
        obs, rew, done, info = env.step([g['e_action'][e][0][t], g['e_action'][e][1][t]], draws, lambda k, tgt, f: st if k == env.n_sub - 1 else None,
                                        contacts=lambda: contacts_of(g['e_contacts'][e][t]), **kw)
        if t < N_FULL:
            np.testing.assert_allclose(np.array(obs), g['e_obs_full'][e][t], rtol=0, atol=TOL)
            check_rays(t + 1)
        for r in range(2):
            np.testing.assert_allclose(core_of(obs[r]), g['e_obs_core'][e][t][r], rtol=0, atol=TOL)
            np.testing.assert_allclose(percep_checks(obs[r]), g['e_obs_checks'][e][t][r], rtol=1e-9, atol=1e-7)
        np.testing.assert_allclose(rew, g['e_reward'][e][t], atol=1e-12)
        assert done == bool(g['e_done'][e][t]), t
        np.testing.assert_allclose(env.target_pos, g['e_flag'][e][t + 1], atol=1e-12)
        assert env.with_flag[0] == bool(g['e_with_flag'][e][t + 1]) and env.switch_flag_at_this_frame == bool(g['e_switch'][e][t])
        assert list(env.oppo_visible) == list(g['e_visible'][e][t + 1])
        np.testing.assert_allclose(info, g['e_info'][e][t], atol=1e-12)
        for k in range(10):
            f = env.applied[k]
            for r in range(2):
                assert (f is not None) == bool(g['e_force_on'][e][t][k][r])
                if f is not None:
                    np.testing.assert_allclose(f[r], g['e_force'][e][t][k][r], atol=1e-12)
    assert done and draws.exhausted()


# ------------------------------------------------------------------------------------------------ engine vs goldens
OBS_TOL = 3e-5        # float32 engine vs float64 reference: ray lengths reach 20 m
REW_TOL = 2e-6


def draws_to_uniforms(log):
    """(kind, a, b, value) rows of the reference's np.random log -> the U[0,1) numbers that reproduce them in the engine."""
    u = np.zeros(len(log), dtype=np.float64)
    for i, (k, a, b, v) in enumerate(log):
        if int(k) == 0:
            u[i] = (v - a) / (b - a) if b > a else 0.0
        elif int(k) == 1:
            u[i] = (v - a + 0.5) / (b - a)
        else:
            u[i] = v
    return u.astype(np.float32)


def make_engine(cfg_dict, n_arenas, lib_path, **kw):
    from lifelike_agility_and_play_amd import sepmc_capi, urdf_model
    cfg = sepmc_capi.make_sepmc_config(n_arenas, cfg_dict, **kw)
    E = sepmc_capi.SepmcEngine(cfg, urdf_model.default_model_blob(), lib_path=lib_path)
    from epmc_parity_common import BASE_SPEC
    E.set_spec(**BASE_SPEC)                     # (epmc_parity_common.spec_variant)
    return E


def script_of(drill):
    hit = np.array([drill_rays(drill, r)[0] for r in range(2)])
    frac = np.array([drill_rays(drill, r)[1] for r in range(2)])
    vis = np.array([vis_blocked(drill, s) for s in range(21)])
    return hit, frac, vis


def _reset_engine(E, g, draws_log, prev_orn):
    from lifelike_agility_and_play_amd import sepmc_capi
    u = np.full(sepmc_capi.LLS_MAX_DRAWS, 0.5, np.float32)
    u[:len(draws_log)] = draws_to_uniforms(draws_log)
    E.script_reset(*script_of(0))
    E.reset(draws=u[None], prev_orn=np.asarray(prev_orn)[None])


def _check_reset_state(E, boxes, flag, with_flag, friction, fix_spd, state, obs, push, noise, last_two, last_esc, visible):
    rows, n = E.boxes()
    assert n[0] == len(boxes)
    np.testing.assert_allclose(rows[0][:n[0]], boxes[:, 1:7], atol=1e-6)
    ep = E.episode()
    np.testing.assert_allclose([ep['flag_x'][0], ep['flag_y'][0], ep['flag_z'][0]], flag, atol=1e-6)
    assert bool(ep['with_flag0'][0] > 0.5) == bool(with_flag)
    assert abs(ep['friction'][0] - friction) < 1e-6 and abs(ep['fix_spd'][0] - fix_spd) < 1e-6
    np.testing.assert_allclose([ep['push_fx'][0], ep['push_fy'][0], ep['push_fz'][0]], push, atol=2e-5)
    if noise is not None:
        np.testing.assert_allclose([ep[k][0] for k in NOISE], noise, atol=1e-7)
    assert abs(ep['last_two_rob_pos_diff_len'][0] - last_two) < 1e-5 and abs(ep['last_esc_flag_pos_diff_len'][0] - last_esc) < 1e-5
    assert [bool(ep['visible0'][0] > 0.5), bool(ep['visible1'][0] > 0.5)] == [bool(x) for x in visible]
    np.testing.assert_allclose(E.state()[0], state, atol=2e-6)
    np.testing.assert_allclose(E.obs()[0], obs, rtol=0, atol=OBS_TOL)


def check_engine_reset_cases(lib_path):
    """CTG.reset() inside the engine from the reference's own draws: arena, flag, roles, friction, push, noise, both start poses
    (incl. the shared in-place start orientation) and both first observations of the 15 golden cases."""
    g = load_golden()
    for k in range(len(g['t_seed'])):
        noisy = bool(g['t_noise_on'][k])
        E = make_engine(env_config(g['t_elements'][k], noisy), 1, lib_path)
        _reset_engine(E, g, g['t_draws'][k][:g['t_n_draws'][k]], g['t_prev_orn'][k])
        _check_reset_state(E, g['t_boxes'][k][:g['t_n_boxes'][k]], g['t_flag'][k], g['t_with_flag'][k], g['t_friction'][k], g['t_fix_spd'][k], g['t_state'][k],
                           g['t_obs'][k], g['t_push'][k], g['t_noise'][k] if noisy else None, g['t_last_two'][k], g['t_last_esc'][k], g['t_vis'][k])
        E.close()


def check_engine_episodes(lib_path, model):
    """The 4 scripted golden episodes through ll_sepmc_step_scripted: observations, ray and visibility end points, flag hand-over,
    first-contact-wins bookkeeping, rewards, termination, info and the two-robot push schedule, step by step."""
    from lifelike_agility_and_play_amd import sepmc_capi
    g = load_golden()
    for e in range(len(g['e_n'])):
        n = int(g['e_n'][e])
        noisy = bool(g['e_noise_on'][e])
        cfg = env_config(g['e_elements'][e], noisy, g['e_max_steps'][e])
        E = make_engine(cfg, 1, lib_path)
        # the oracle runs alongside only to cut the reference's draw log into per-call pieces
        orc = make_oracle(g, model, g['e_elements'][e], noisy, g['e_prev_orn'][e], g['e_max_steps'][e])
        log = g['e_draws'][e][:g['e_n_draws'][e]]
        draws = SO.LoggedDraws(log)
        drill = [0]
        kw = dict(rays=lambda r, f, t: drill_rays(drill[0], r), vis=lambda s, f, t: vis_blocked(drill[0], s))
        orc.reset(draws, contacts=lambda: [], **kw)
        _reset_engine(E, g, log[:draws.i], g['e_prev_orn'][e])
        np.testing.assert_allclose(E.obs()[0], g['e_reset_obs'][e], rtol=0, atol=OBS_TOL)
        np.testing.assert_allclose(E.state()[0], g['e_init_state'][e], atol=2e-6)
        vis_rows = g['e_vis'][e][:g['e_n_vis'][e]]

        def check_rays(d):
            f, t, _, _ = E.rays()
            np.testing.assert_allclose(f[0], g['e_ray_from'][e][d], atol=OBS_TOL)
            np.testing.assert_allclose(t[0], g['e_ray_to'][e][d], atol=OBS_TOL)
            v = E.vis()[0]
            for w in vis_rows[vis_rows[:, 0] == d]:                         # every segment the reference asked about
                s = int(w[1])
                assert v[s][7] > 0.5
                np.testing.assert_allclose(v[s][:6], w[2:8], atol=OBS_TOL)
                assert bool(v[s][6] > 0.5) == bool(w[8])
        check_rays(0)
        for t in range(n):
            drill[0] = t + 1
            i0 = draws.i
            st = [g['e_state'][e][0][t], g['e_state'][e][1][t]]
            orc.step([g['e_action'][e][0][t], g['e_action'][e][1][t]], draws, lambda k, tgt, f: st if k == orc.n_sub - 1 else None,
                     contacts=lambda: contacts_of(g['e_contacts'][e][t]), **kw)
            u = draws_to_uniforms(log[i0:draws.i])
            cl = np.full((sepmc_capi.LLS_MAX_CONTACTS, 4), -9, np.int32)
            rows = contacts_of(g['e_contacts'][e][t])
            if rows:
                cl[:len(rows)] = rows
            hit, frac, vis = script_of(t + 1)
            E.step_scripted(np.array([g['e_action'][e][0][t], g['e_action'][e][1][t]]), np.array(st), hit, frac, vis, cl, draws=u[None] if len(u) else None)
            obs = E.obs()[0].astype(np.float64)
            if t < N_FULL:
                np.testing.assert_allclose(obs, g['e_obs_full'][e][t], rtol=0, atol=OBS_TOL)
                check_rays(t + 1)
            for r in range(2):
                np.testing.assert_allclose(core_of(obs[r]), g['e_obs_core'][e][t][r], rtol=0, atol=OBS_TOL)
                np.testing.assert_allclose(percep_checks(obs[r]), g['e_obs_checks'][e][t][r], rtol=1e-5, atol=2e-3)
            rew, done, why = E.reward_done()
            np.testing.assert_allclose(rew[0], g['e_reward'][e][t], atol=REW_TOL)
            assert bool(done[0]) == bool(g['e_done'][e][t]), (e, t)
            ep = E.episode()
            np.testing.assert_allclose([ep['flag_x'][0], ep['flag_y'][0], ep['flag_z'][0]], g['e_flag'][e][t + 1], atol=1e-6)
            assert bool(ep['with_flag0'][0] > 0.5) == bool(g['e_with_flag'][e][t + 1]) and bool(ep['switch'][0] > 0.5) == bool(g['e_switch'][e][t])
            assert [bool(ep['visible0'][0] > 0.5), bool(ep['visible1'][0] > 0.5)] == [bool(x) for x in g['e_visible'][e][t + 1]], (e, t)
            np.testing.assert_allclose(E.info()[0], g['e_info'][e][t], atol=2e-5)
            pt = E.push_trace()[0]                                        # [robot][substep][4]
            for r in range(2):
                assert [bool(x > 0.5) for x in pt[r][:, 0]] == [bool(x) for x in g['e_force_on'][e][t][:, r]], (e, t, r)
                np.testing.assert_allclose(pt[r][:, 1:4], g['e_force'][e][t][:, r], atol=5e-5)
        assert done[0] and draws.exhausted()
        if e == 0:
            assert why[0] == sepmc_capi.DONE_CATCH
        E.close()


# ------------------------------------------------------------------------------------------------ free-running checks
ALL_ELEMENTS = (1, 1, 1)


def boxes_with_flag(E, a):
    rows, n = E.boxes()
    ep = E.episode()
    st = [[0.0, *rows[a][b][:3], *rows[a][b][3:6], 0.0] for b in range(n[a])]
    st.append(list(SO.flag_box([ep['flag_x'][a], ep['flag_y'][a], ep['flag_z'][a]])))
    return np.array(st, dtype=np.float64)


def check_free_running(lib_path, n_arenas=8, steps=120, seed=3):
    """Real physics, rays and contacts: invariants between the two robots' views, analytic rays and visibility segments against
    the oracle's cast_rays on the engine's own arena, episodes ending and restarting."""
    E = make_engine(env_config(ALL_ELEMENTS), n_arenas, lib_path, auto_reset=1, seed=seed)
    E.reset()
    ends = 0
    for t in range(steps):
        E.fill_random_actions(np.exp(-2.0))
        E.step()
        if t % 20 == 19 or t == steps - 1:
            obs = E.obs().astype(np.float64)
            assert np.isfinite(obs).all()
            rew, done, why = E.reward_done()
            assert np.all(np.abs(rew[:, 0] + rew[:, 1]) < 1e-6)                                  # zero sum
            np.testing.assert_allclose(obs[:, 0, 913 + 35:913 + 39], obs[:, 1, 913 + 35:913 + 39], atol=0)      # the same flag
            np.testing.assert_allclose(obs[:, 0, 913 + 49], obs[:, 1, 913 + 50], atol=0)          # with_flag rows mirror each other
            np.testing.assert_allclose(obs[:, 0, 913 + 21:913 + 24], obs[:, 1, 913:913 + 3], atol=0)            # my oppo_pos is its position
            np.testing.assert_allclose(obs[:, 1, 913 + 21:913 + 24], obs[:, 0, 913:913 + 3], atol=0)
            f, tt, hit, frac = E.rays()
            vis = E.vis()
            same = tot = 0
            for a in range(min(n_arenas, 3)):
                if done[a]:
                    continue                      # the rays of a re-seeded arena saw the new arena; episode() reports it too, but keep it simple
                bx = boxes_with_flag(E, a)
                ep = E.episode()
                if ep['switch'][a] > 0.5:
                    continue                      # the flag moved after the rays were cast
                for r in range(2):
                    h2, f2 = SO.cast_rays(f[a][r], tt[a][r], bx)
                    same += int((h2 == hit[a][r]).sum()); tot += len(h2)
                    both = h2 & hit[a][r]
                    np.testing.assert_allclose(frac[a][r][both], f2[both], atol=3e-4)
                asked = vis[a][:, 7] > 0.5
                hb, _ = SO.cast_rays(vis[a][asked, 0:3], vis[a][asked, 3:6], bx)
                assert (hb == (vis[a][asked, 6] > 0.5)).mean() > 0.9
            if tot:
                assert same / tot > 0.995, same / tot
        ends = E.counters()['episodes']
    assert ends > 0                                 # random actions make robot 0 fall within the run
    st = E.state()
    assert np.isfinite(st).all() and np.all(np.abs(st[:, :, 0:2]) < 2.6)                             # nobody left through a wall
    out = dict(episodes=ends)
    E.close()
    return out


def check_flag_handover_physical(lib_path):
    """The robot that may take the flag is put with a thigh onto it: the engine's own contact test hands the flag over (reward
    +-1, a new flag position), and a robot that already holds it does not trigger anything."""
    E = make_engine(env_config((0, 0, 0)), 2, lib_path, auto_reset=0, seed=11)
    E.reset()
    ep = E.episode()
    st = E.state().astype(np.float64)
    flag0 = np.array([ep['flag_x'], ep['flag_y']]).T.copy()
    wf0 = ep['with_flag0'].copy()
    for a in range(2):
        taker = 1 if wf0[a] > 0.5 else 0                       # CTG:579-581
        who = taker if a == 0 else 1 - taker                   # arena 0: the taker touches; arena 1: the holder touches (nothing happens)
        st[a, who, 0:2] = flag0[a] - np.array([0.195, -0.15])
        st[a, who, 2] = 0.36                                   # the standing height of the start pose
        st[a, who, 3:7] = [0, 0, 0, 1]
        st[a, 1 - who, 0:2] = -np.sign(flag0[a]) * 1.5       # the other one far away
    E.set_state(st)
    for t in range(3):               # (in the first step a shank end within 2 cm of the ground may be listed before the flag: CTG:426-440)
        E.step_host(np.zeros((2, 2, 12)))
        ep2 = E.episode()
        rew, done, why = E.reward_done()
        assert ep2['switch'][1] < 0.5 and ep2['with_flag0'][1] == wf0[1] and rew[1][0] == 0.0        # the holder touching it: nothing
        if ep2['switch'][0] > 0.5:
            break
    assert ep2['switch'][0] > 0.5 and ep2['with_flag0'][0] != wf0[0] and ep2['who_taker'][0] == 2
    assert abs(ep2['flag_x'][0] - flag0[0][0]) + abs(ep2['flag_y'][0] - flag0[0][1]) > 1e-3
    taker = 1 if wf0[0] > 0.5 else 0
    assert rew[0][taker] == 1.0 and rew[0][1 - taker] == -1.0
    E.close()


def check_free_running_big(lib_path, n_arenas, steps):
    E = make_engine(env_config(ALL_ELEMENTS), n_arenas, lib_path, auto_reset=1, seed=5)
    E.reset()
    for t in range(steps):
        E.fill_random_actions(np.exp(-2.0))
        E.step()
    obs = E.obs().astype(np.float64)
    rew, done, why = E.reward_done()
    st = E.state()
    assert np.isfinite(obs).all() and np.isfinite(st).all()
    assert np.all(np.abs(rew[:, 0] + rew[:, 1]) < 1e-6)
    np.testing.assert_allclose(obs[:, 0, 913 + 35:913 + 39], obs[:, 1, 913 + 35:913 + 39], atol=0)
    np.testing.assert_allclose(obs[:, 0, 913 + 21:913 + 24], obs[:, 1, 913:913 + 3], atol=0)
    c = E.counters()
    assert c['episodes'] > 0 and c['nonfinite'] == 0
    E.close()
    return c


def check_engine_against_emulation(emul_lib_path, n_arenas=2048, steps=2, seed=5, spec=None, gpu_lib=None, report_only=False):
    """Every field of every observation the HIP kernels write, at full size, against the HOST build of the very same kernel source (tests/emul), arena by arena --
    the net under the one-wave-per-SIMD chase-tag kernels, which sit at 256 + 255 registers (round 4: a seven-rays-per-chunk build of them wrote garbage into the
    flag_info fields of arenas that re-seed -- right state, right episode record, wrong observation; the three-ray build is shipped, and this test is what would
    catch the same failure in it or in any later build: HISTORY.md).  Same config, seed and actions on both sides; before every step the emulation takes the
    engine's state, so physics rounding stays one step old; arenas whose done flag / reason / flag holder differ after a step (a contact class on the last
    bit) are counted, capped at 1 % and left out.  About 4 % of the arenas are caught at spawn and RE-SEED in the very first step: the path that failed."""
    import epmc_parity_common as ec
    with ec.spec_variant(**(spec or {})):
        G = make_engine(env_config(ALL_ELEMENTS), n_arenas, gpu_lib, auto_reset=1, seed=seed)        # (gpu_lib: another build of the HIP library -- tools/diag_sepmc_chunk7.py)
        H = make_engine(env_config(ALL_ELEMENTS), n_arenas, emul_lib_path, auto_reset=1, seed=seed)
    G.reset(); H.reset()
    P3, NR = 135, 778
    out = dict(reseeded=0, left_out=0, ray_mismatch=0.0, worst_tail=0.0)

    # tail layout (sepmc_step.hpp observe): percept_vec 5 | oppo_info 15 | oppo_info_cheat 15 | flag_info 7 | flag_info_cheat 7 | with_flag 2 | control_spd 1; velocities
    # (prop: joint rates, base twist; oppo_info: the other robot's twist) are compared relative to the robot's fastest joint -- a violent contact step differs
    # between two float32 builds by fused-multiply-add rounding alone -- everything else absolutely
    tail_vel = np.zeros(52, bool); tail_vel[14:20] = True; tail_vel[29:35] = True
    prop_vel = np.zeros(33, bool); prop_vel[12:30] = True
    prop_vel = np.concatenate([np.tile(prop_vel, 3), np.zeros(36, bool)])
    FIELDS = (('percept_vec', 0, 5), ('oppo_info', 5, 20), ('oppo_info_cheat', 20, 35), ('flag_info', 35, 42), ('flag_info_cheat', 42, 49), ('with_flag', 49, 51), ('control_spd', 51, 52))

    def compare(label, keep):
        og, oh = G.obs().astype(np.float64)[keep], H.obs().astype(np.float64)[keep]
        if not keep.any():
            return
        assert np.isfinite(og).all(), label
        scale = 1.0 + np.abs(oh[..., :P3][..., prop_vel]).max(-1, keepdims=True)
        dprop = np.abs(og[..., :P3] - oh[..., :P3]) / np.where(prop_vel, scale, 1.0)
        tail = np.abs(og[..., P3 + NR:] - oh[..., P3 + NR:]) / np.where(tail_vel, scale, 1.0)
        rays = np.abs(og[..., P3:P3 + NR] - oh[..., P3:P3 + NR]) > 2e-3             # a ray that grazes an edge may answer differently: counted
        # robots whose contact step is ill-conditioned between the two builds show it in their own state first: left to the physics parity tests, counted here
        rough = dprop.max(-1) > 5e-3
        out['rough'] = out.get('rough', 0) + int(rough.sum())
        if report_only and keep.all():
            # where the differences sit (tools/diag_sepmc_chunk7.py): by the robot's 16-lane row inside its wavefront (row = (2 arena + robot) % 4), by
            # observation entry, and the rays by family
            wr = (2 * np.arange(n_arenas)[:, None] + np.arange(2)[None, :]) % 4
            det = out.setdefault('detail', {}).setdefault(label, {})
            det['rough_by_wave_row'] = [int(rough[wr == k].sum()) for k in range(4)]
            det['rough_prop_entries'] = [int(x) for x in (dprop[rough] > 5e-3).sum(0).reshape(-1)[:33]] if rough.any() else []
            det['ray_mismatch_by_wave_row'] = [float(rays[wr == k].mean()) for k in range(4)]
            det['ray_mismatch_height_fan_front'] = [float(rays[..., 0:325].mean()), float(rays[..., 325:453].mean()), float(rays[..., 453:778].mean())]
            det['robots_with_ray_mismatch'] = int(rays.any(-1).sum())
        out['ray_mismatch'] = max(out['ray_mismatch'], rays.mean())
        # the visibility flag (oppo_info[0] = oppo_info_cheat[0]; it also masks oppo_info) is a ray test and an angle test decided on the last bit now and then: counted, capped, set aside
        vis_tie = (og[..., P3 + NR + 20] != oh[..., P3 + NR + 20]) & ~rough
        out['visibility_ties'] = out.get('visibility_ties', 0) + int(vis_tie.sum())
        assert vis_tie.sum() <= max(2, int(1e-3 * vis_tie.size)), (label, int(vis_tie.sum()))
        settled = ~rough & ~vis_tie
        out['worst_tail'] = max(out['worst_tail'], tail[settled].max() if settled.any() else 0.0)
        per_field = {name: int((tail[..., a:b][settled].max(-1) > 5e-3).sum()) for name, a, b in FIELDS}
        out.setdefault('per_field', {})[label] = per_field
        if report_only:
            return
        assert sum(per_field.values()) == 0, (label, 'percept_vec .. control_spd differ in robots whose own state agrees', per_field)
        assert rough.mean() < 0.01, (label, rough.mean())
        assert rays.mean() < 2e-3, (label, rays.mean())
    compare('reset', np.ones(n_arenas, bool))
    rng = np.random.default_rng(seed)
    for t in range(steps):
        act = (rng.normal(size=(n_arenas, 2, 12)) * 0.135).astype(np.float32)
        H.set_state(G.state())
        G.step_host(act); H.step_host(act)
        (rg, dg, wg), (rh, dh, wh) = G.reward_done(), H.reward_done()
        eg, eh = G.episode(), H.episode()
        same = (dg.reshape(n_arenas, -1)[:, 0] == dh.reshape(n_arenas, -1)[:, 0]) & (wg.reshape(n_arenas, -1)[:, 0] == wh.reshape(n_arenas, -1)[:, 0]) & \
               (eg['with_flag0'] == eh['with_flag0']) & (np.abs(eg['flag_x'] - eh['flag_x']) < 1e-6)
        out['left_out'] += int((~same).sum())
        out['reseeded'] += int((dg.reshape(n_arenas, -1)[:, 0] != 0)[same].sum())
        compare('step %d' % t, same)
        compare('step %d, re-seeding arenas only' % t, same & (dg.reshape(n_arenas, -1)[:, 0] != 0))
        compare('step %d, the others' % t, same & (dg.reshape(n_arenas, -1)[:, 0] == 0))
    assert out['left_out'] <= max(2, int(0.01 * n_arenas * steps)), out
    assert out['reseeded'] >= n_arenas // 64, out                      # the re-seed path was exercised (catches at spawn)
    G.close(); H.close()
    return out


def check_robot_robot_contact(lib_path):
    """This build's robot-robot contact (capsule pairs between the two rows of an arena).  Arena 0: robot 0 is dropped onto robot 1
    and is carried by it instead of falling through.  Arena 1: robot 0 walks its front legs into robot 1's hind legs -- a catch
    (CTG:442-450) with +-1 on top of the reward.  Arena 2: far apart, nothing happens."""
    from lifelike_agility_and_play_amd import sepmc_capi
    E = make_engine(env_config((0, 0, 0)), 3, lib_path, auto_reset=0, seed=21)
    E.reset()
    st = E.state().astype(np.float64)
    wf0 = E.episode()['with_flag0'].copy()
    for a in range(3):
        for r in range(2):
            st[a, r, 3:7] = [0, 0, 0, 1]
            st[a, r, 7:13] = 0.0
    st[0, 1, 0:3] = [1.0, 1.0, 0.36]; st[0, 0, 0:3] = [1.0, 1.0, 0.63]       # (0.36 = the standing height of the start pose)
    st[1, 1, 0:3] = [0.0, 1.0, 0.36]; st[1, 0, 0:3] = [-0.40, 1.0, 0.36]
    st[2, 1, 0:3] = [1.5, -1.5, 0.36]; st[2, 0, 0:3] = [-1.5, -1.5, 0.36]
    # keep the flag out of the way of all three
    E.set_state(st)
    zero = np.zeros((3, 2, 12))
    E.step_host(zero)
    rew, done, why = E.reward_done()
    assert done[1] and (why[1] & sepmc_capi.DONE_CATCH), why
    want = 1.0 if wf0[1] > 0.5 else -1.0
    assert abs(rew[1][0] - want) < 1e-6 and abs(rew[1][1] + want) < 1e-6, rew[1]
    assert not done[2] and rew[2][0] == 0.0
    zs = []
    for t in range(40):
        E.step_host(zero)
        s = E.state()
        assert np.isfinite(s).all()
        zs.append(s[0, 0, 2] - s[0, 1, 2])
    # robot 1 carries robot 0: its legs give way under the double load (alone it keeps standing at 0.3 m), and robot 0 stays above it
    # (without the contact rows it would drop through and stand at the same height)
    assert min(zs) > 0.04 and E.state()[0, 1, 2] < 0.2, (zs[-5:], E.state()[0, :, 2])
    assert abs(E.state()[2, 0, 0] + 1.5) < 0.2          # the far pair stayed where it was
    E.close()
    return dict(stack_gap=float(zs[-1]))


def check_arena_corners_are_closed(lib_path):
    """A robot pressed into a wall right next to a corner is pushed back into the arena.  (Round 5: the walls were thickened outwards but not lengthened -- a point inside one thickened wall
    within a few centimetres of its end was nearest to the END face and left through the pocket between the two walls: one robot in 16 M robot-steps of the soak run.)"""
    E = make_engine(env_config((0, 0, 0)), 4, lib_path, auto_reset=0, seed=23)
    E.reset()
    st = E.state().astype(np.float64)
    for a, (sx, sy) in enumerate(((-1, -1), (1, -1), (-1, 1), (1, 1))):
        for r in range(2):
            st[a, r, 3:7] = [0, 0, 0, 1]
            st[a, r, 7:13] = 0.0
        st[a, 1, 0:3] = [-sx * 1.5, -sy * 1.5, 0.36]
        st[a, 0, 0:3] = [sx * 2.44, sy * 2.62, 0.36]          # the base 12 cm inside the thickened wall along x, 6 cm from that wall's end face
    E.set_state(st)
    zero = np.zeros((4, 2, 12))
    far = 0.0
    for t in range(50):
        E.step_host(zero)
        s = E.state()
        assert np.isfinite(s).all()
        far = max(far, float(np.abs(s[:, 0, 0:2]).max()))
    s = E.state()
    assert far < 2.75 and np.abs(s[:, 0, 0:2]).max() < 2.6, (far, s[:, 0, 0:3])
    E.close()
    return dict(farthest=far, final=float(np.abs(s[:, 0, 0:2]).max()))


def check_trained_policy_plays_chase_tag(lib_path, n_arenas=4, horizon=380, min_caught=0.25):
    """SURVEY.md 8f-3 for the strategic level -- the only Bullet-facing check of this build's robot-robot contact model: the reference's TRAINED
    SEPMC policy (data/models/strategic_level.model, trained against PyBullet; NumPy restatement oracle/sepmc_policy.py) drives BOTH robots of our
    chase-tag arenas under the protocol of test_strategic_level_env.py.  The robots run at the commanded 1 m/s, find each other across the 5 m
    arena and the games end the way they are meant to: by a catch -- a leg of one robot touching the other -- not by falls or time-outs."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import rollout_sepmc_policy as R
    o = R.rollout(n_arenas, horizon, lib_path)
    why = o['why']
    out = dict(caught=float(((why & 8) != 0).mean()), fell=float(((why & 1) != 0).mean()), running=float(o['alive'].mean()),
               speed=float((o['path'] / (o['steps'][:, None] * 0.02)).mean()), closest=float(np.median(o['closest'])), steps=float(o['steps'].mean()),
               touch_frac=float(o['touch_frac']))
    assert out['caught'] >= min_caught and out['fell'] <= 0.3, out        # (MI355X, 512 arenas: 79 % caught, 17 % robot 0 knocked over or fallen, 5 % timed out)
    assert 0.4 < out['speed'] < 1.5 and out['closest'] < 1.2, out
    return out


def check_per_robot_torque_limit(lib_path, n_arenas=6):
    """max_tau given as a [lo, hi] list draws one torque limit per LeggedRobot (LR:244, CTG:62-72): ll_sepmc_config.max_tau_robot1.  With large
    actions, robot 0 of an engine with limits (16, 4) moves exactly like robot 0 of a (16, 16) engine and its robot 1 exactly like robot 1 of a
    (4, 4) engine (arenas whose robots are too far apart to touch: they share nothing but the arena)."""
    def run(t0, t1):
        cfg = env_config((0, 0, 0))
        cfg['max_tau'] = t0
        cfg['max_tau_robot1'] = t1
        E = make_engine(cfg, n_arenas, lib_path, seed=12)
        E.reset()
        s0 = E.state()
        rng = np.random.default_rng(3)
        for _ in range(3):
            E.step_host((rng.normal(size=(n_arenas, 2, 12)) * 1.5).astype(np.float32))
        out = E.state()
        E.close()
        return s0, out
    s0, mixed = run(16.0, 4.0)
    _, hi = run(16.0, 0.0)                              # 0: robot 1 shares robot 0's limit
    _, lo = run(4.0, 4.0)
    far = np.linalg.norm(s0[:, 0, 0:2] - s0[:, 1, 0:2], axis=1) > 1.6
    assert far.sum() >= 2
    np.testing.assert_array_equal(mixed[far, 0], hi[far, 0])
    np.testing.assert_array_equal(mixed[far, 1], lo[far, 1])
    assert np.abs(mixed[far, 1] - hi[far, 1]).max() > 1e-3            # the limit matters for these actions


def check_multi_step_launch(lib_path, sizes=(6,), k=5, n_launches=4):
    """ll_sepmc_step_random_n(sigma, k) == k x {ll_sepmc_fill_random_actions(sigma); ll_sepmc_step()}, bit for bit (both robots' states and
    965-float observations, rewards, done, episode records, counters), with episodes timing out and re-seeding inside the launches."""
    sg = float(np.exp(-2.0))
    for n in sizes:
        cfg = env_config(ALL_ELEMENTS, max_steps=3 * k)
        A = make_engine(cfg, n, lib_path, auto_reset=1, seed=4)
        B = make_engine(cfg, n, lib_path, auto_reset=1, seed=4)
        A.reset(); B.reset()
        for L in range(n_launches):
            for _ in range(k):
                A.fill_random_actions(sg); A.step()
            B.step_random_n(sg, k)
            A.sync(); B.sync()
            np.testing.assert_array_equal(A.state(), B.state())
            np.testing.assert_array_equal(A.obs(), B.obs())
            for x, y in zip(A.reward_done(), B.reward_done()):
                np.testing.assert_array_equal(x, y)
            ea, eb = A.episode(), B.episode()
            for key in ea:
                np.testing.assert_array_equal(ea[key], eb[key])
            assert A.counters() == B.counters()
        assert A.counters()['episodes'] > 0
        A.close(); B.close()


def check_split_rays_equal_fused(lib_path, n=4, n_steps=30, multi=(1, 3)):
    """The 2 x 778 perception rays of an arena cast by the ray kernel behind the step kernel (LL_SPLIT_RAYS; epmc_parity_common.check_split_rays_equal_fused) against the
    step kernel casting them itself, bit for bit: every observation entry, the ray traces, states, rewards, episode records; arenas with elements, games ending and re-seeding."""
    import os
    sg = float(np.exp(-2.0))
    cfg = env_config(ALL_ELEMENTS, max_steps=12)
    prev = os.environ.get('LL_SPLIT_RAYS')
    try:
        os.environ['LL_SPLIT_RAYS'] = '0'
        A = make_engine(cfg, n, lib_path, auto_reset=1, seed=4)
        os.environ['LL_SPLIT_RAYS'] = '2'
        B = make_engine(cfg, n, lib_path, auto_reset=1, seed=4)
        A.reset(); B.reset()
        for t in range(n_steps):
            k = multi[t % len(multi)]
            for X, mode in ((A, '0'), (B, '2')):
                os.environ['LL_SPLIT_RAYS'] = mode
                if k == 1:
                    X.fill_random_actions(sg); X.step()
                else:
                    X.step_random_n(sg, k)
            np.testing.assert_array_equal(A.obs(), B.obs())
            np.testing.assert_array_equal(A.state(), B.state())
            for x, y in zip(A.reward_done(), B.reward_done()):
                np.testing.assert_array_equal(x, y)
            ea, eb = A.episode(), B.episode()
            for key in ea:
                np.testing.assert_array_equal(ea[key], eb[key])
            if 2 * n <= 512:
                for x, y in zip(A.rays(), B.rays()):
                    np.testing.assert_array_equal(x, y)
        assert A.counters() == B.counters() and A.counters()['episodes'] > 0, A.counters()
        A.close(); B.close()
    finally:
        if prev is None:
            os.environ.pop('LL_SPLIT_RAYS', None)
        else:
            os.environ['LL_SPLIT_RAYS'] = prev


def check_parked_variant_equals_plain(lib_path, n=4, n_steps=30):
    """step_env<PARK = true> (the larger-batch GPU build: episode scalars in the row scratch during the substep loop, history read after it) against
    the plain variant on the HOST build, bit for bit (LL_EMUL_PARK=1; see epmc_parity_common.check_parked_variant_equals_plain)."""
    import os
    sg = float(np.exp(-2.0))
    cfg = env_config(ALL_ELEMENTS, max_steps=12)
    cfg['env_randomize_config']['disturb_force_config'] = {'start_time': 0.0, 'interval_time': 0.1, 'duration_time': 0.06, 'horizontal_force': [10, 50], 'vertical_force': [0, 10]}
    A = make_engine(cfg, n, lib_path, auto_reset=1, seed=4)
    B = make_engine(cfg, n, lib_path, auto_reset=1, seed=4)
    A.reset(); B.reset()
    pushed = 0
    try:
        for t in range(n_steps):
            os.environ.pop('LL_EMUL_PARK', None)
            A.fill_random_actions(sg); A.step()
            os.environ['LL_EMUL_PARK'] = '1'
            B.fill_random_actions(sg); B.step()
            np.testing.assert_array_equal(A.state(), B.state())
            np.testing.assert_array_equal(A.obs(), B.obs())
            for x, y in zip(A.reward_done(), B.reward_done()):
                np.testing.assert_array_equal(x, y)
            ea, eb = A.episode(), B.episode()
            for key in ea:
                np.testing.assert_array_equal(ea[key], eb[key])
            np.testing.assert_array_equal(np.asarray(A.push_trace()), np.asarray(B.push_trace()))
            pushed += int(np.asarray(A.push_trace())[..., 0].sum())
    finally:
        os.environ.pop('LL_EMUL_PARK', None)
    assert A.counters() == B.counters() and A.counters()['episodes'] > 0 and pushed > 0, (A.counters(), pushed)
    A.close(); B.close()


def check_pair_physics_against_oracle(lib_path, n_arenas=24, seed=5, total_arenas=None, cap_ill=1, spec=None):
    """Two robots within reach of each other (side by side, nose to tail, one partly above the other), random joint states and
    velocities, the push active: one control step of real physics, engine (float32, two rows exchanging registers) vs the float64
    oracle's two-robot substep (orc_substep_pair: explicit Jacobians, M^-1 by unit responses) given the same arena records,
    friction and push forces.  The two share the spec (capsules, pair order, row order) and nothing else.
    total_arenas: the engine runs that many arenas (above 2048: the larger-batch kernel build) and the n_arenas cases are spread over the
    first, middle and last wavefronts of its grid.
    spec: switches set on BOTH sides (round 6: pair_friction = 0.25, max_pair = 4, self_friction = 0.25 -- the engine twins of what had been oracle-only switches; the
    engine then runs its XROWS build)."""
    from conftest import make_oracle_batch
    from oracle import oracle as orc
    if not spec:                       # (possibly inside epmc_parity_common.spec_variant, which has set both sides)
        return _check_pair_physics_against_oracle(lib_path, n_arenas, seed, total_arenas, cap_ill, spec)
    orc.reset_spec()
    orc.set_spec(**spec)
    try:
        return _check_pair_physics_against_oracle(lib_path, n_arenas, seed, total_arenas, cap_ill, spec)
    finally:
        orc.reset_spec()


def _check_pair_physics_against_oracle(lib_path, n_arenas, seed, total_arenas, cap_ill, spec):
    from conftest import make_oracle_batch
    from oracle import oracle as orc
    from lifelike_agility_and_play_amd import mocap, urdf_model
    from parity_common import quat_align
    cfg = env_config((1, 0, 0))
    cfg['env_randomize_config']['disturb_force_config'] = {'start_time': 0.0, 'interval_time': 1.0, 'duration_time': 0.5, 'horizontal_force': [10, 50], 'vertical_force': [0, 10]}
    NA = total_arenas or n_arenas
    third = n_arenas // 3
    idx = np.arange(n_arenas) if not total_arenas else np.concatenate([np.arange(third), NA // 2 - 5 + np.arange(third), NA - (n_arenas - 2 * third) + np.arange(n_arenas - 2 * third)])
    E = make_engine(cfg, NA, lib_path, seed=seed)
    if spec:
        E.set_spec(**spec)
    E.reset()
    rng = np.random.default_rng(seed)
    st = E.state().astype(np.float64)
    from scipy.spatial.transform import Rotation as R
    for a in idx:
        c = rng.uniform(-1.2, 1.2, 2)
        ang = rng.uniform(0, 2 * np.pi)
        dist = rng.uniform(0.18, 0.62)
        off = 0.5 * dist * np.array([np.cos(ang), np.sin(ang)])
        for r in range(2):
            st[a, r, 0:2] = c + (off if r == 0 else -off)
            st[a, r, 2] = rng.uniform(0.35, 0.41) + (0.12 if (a % 4 == 3 and r == 0) else 0.0)      # (standing height of the start pose: 0.356)
            st[a, r, 3:7] = R.from_euler('xyz', [rng.normal() * 0.1, rng.normal() * 0.1, rng.uniform(0, 2 * np.pi)]).as_quat()
            st[a, r, 7:13] = rng.normal(size=6) * 0.3
            st[a, r, 13:25] += rng.normal(size=12) * 0.15
            st[a, r, 25:37] = rng.normal(size=12)
    E.set_state(st)
    st32 = E.state().astype(np.float64)
    act = (rng.normal(size=(NA, 2, 12)) * 0.135).astype(np.float32)
    ep = E.episode()
    rows, cnt = E.boxes()
    E.step_host(act)
    es = E.state().astype(np.float64)
    tr = E.push_trace().astype(np.float64)
    B = make_oracle_batch(orc, urdf_model.default_model_blob(), mocap.load_mocap('', 0.02), n_envs=1, kd=0.5, max_tau=16.0)
    out = dict(config=[], vel=[], n_rows=0, n_felt=0, who=[])
    ep_after = E.episode()
    for a in idx:
        rec = np.array([[rows[a][b][0] - rows[a][b][3], rows[a][b][0] + rows[a][b][3], rows[a][b][1] - rows[a][b][4], rows[a][b][1] + rows[a][b][4],
                         rows[a][b][2] - rows[a][b][5], rows[a][b][2] + rows[a][b][5], 0.0, 0.0] for b in range(cnt[a])], dtype=np.float64).reshape(-1, 8)
        fl = np.array([ep['flag_x'][a], ep['flag_y'][a], ep['flag_z'][a]], dtype=np.float64)
        rec = np.vstack([rec, [[fl[0] - 0.05, fl[0] + 0.05, fl[1] - 0.05, fl[1] + 0.05, fl[2] - 0.25, fl[2] + 0.25, 0.0, 0.0]]]).astype(np.float32).astype(np.float64)
        rec[0, 3] += 1.0; rec[1, 2] -= 1.0; rec[2, 1] += 1.0; rec[3, 0] -= 1.0        # the walls are solid outwards for contacts (SEPMC_WALL_SOLID)
        rec[0:2, 0] -= 1.0; rec[0:2, 1] += 1.0; rec[2:4, 2] -= 1.0; rec[2:4, 3] += 1.0      # ... and longer by the same at both ends: the corners are closed
        rec = rec.astype(np.float32).astype(np.float64)
        near, s, s_free, flag_at = [], [], [], []
        for r in range(2):
            p = st32[a, r, 0:3]
            sel = np.nonzero((p[0] >= rec[:, 0] - 0.9) & (p[0] <= rec[:, 1] + 0.9) & (p[1] >= rec[:, 2] - 0.9) & (p[1] <= rec[:, 3] + 0.9) & (p[2] <= rec[:, 5] + 0.9))[0][:8]
            near.append(rec[sel]); flag_at.append(int(np.nonzero(sel == len(rec) - 1)[0][0]) if (len(rec) - 1) in sel else -1)
            s.append(st32[a, r].copy()); s_free.append(st32[a, r].copy())
        tgt = [np.clip(s[r][13:25] + act[a, r].astype(np.float64), -3.0, 3.0) for r in range(2)]
        mu = float(np.float32(ep['friction'][a]) * np.float32(0.9))
        nrows = 0
        for k in range(10):
            tau = [np.clip(50.0 * (tgt[r] - s[r][13:25]) - 0.5 * s[r][25:37], -16.0, 16.0) for r in range(2)]
            push = [tr[a, r, k, 1:4] if tr[a, r, k, 0] > 0.5 else None for r in range(2)]
            if k == 9:                          # the contact list the env reads is that of the last substep's collision detection (CTG:426-450)
                tc = B.touch(s[0], s[1], near[0], flag_at[0], near[1], flag_at[1])
                who = [1 if tc[r][0] else (2 if tc[r][1] else ((4 - r) if tc[r][2] else -1)) for r in range(2)]
                taker = 1 if ep['with_flag0'][a] > 0.5 else 0
                out['who'].append((int(ep_after['who0'][a]) == who[0], int(ep_after['who_taker'][a]) == who[taker], who[0], who[taker]))
            s[0], s[1], pr = B.substep_pair(s[0], s[1], tau[0], tau[1], mu, near[0], near[1], 0.5 / 0.9, push[0], push[1])
            nrows += len(pr)
            for r in range(2):                                   # the same step with the other robot ignored
                tf = np.clip(50.0 * (tgt[r] - s_free[r][13:25]) - 0.5 * s_free[r][25:37], -16.0, 16.0)
                s_free[r], _, _ = B.substep_terrain(s_free[r], tf, mu, near[r], 0.5 / 0.9, push[r])
        out['n_rows'] += 1 if nrows else 0
        if max(np.abs(s[r] - s_free[r]).max() for r in range(2)) > 1e-3:
            out['n_felt'] += 1
        for r in range(2):
            err = np.abs(quat_align(es[a, r], s[r]) - s[r])
            ce, ve = max(err[0:7].max(), err[13:25].max()), max(err[7:13].max(), err[25:37].max()) / (1.0 + np.abs(s[r][25:37]).max())
            if ce >= 1e-4 or ve >= 1e-3:
                # outside the bars: legitimate only if the step is ill-conditioned in the ORACLE itself -- rounding its state to float32 between
                # substeps moves its own result by at least a quarter of the engine's deviation (epmc_parity_common.assert_within_bars)
                # (... or, round 5, starting one float32 ulp up / down in every coordinate: two capsules that start deep inside each other have a
                # closest-point normal that hangs on the last bit, which the rounding between substeps does not always disturb)
                oc = ov = 0.0
                for kind in ('r32', 'up', 'down'):
                    s32 = oracle_pair_step(B, st32[a], act[a], rec, mu, tr[a], r32=True, ulp=dict(r32=0, up=1, down=-1)[kind])
                    own = np.abs(quat_align(s32[r], s[r]) - s[r])
                    oc, ov = max(oc, own[0:7].max(), own[13:25].max()), max(ov, max(own[7:13].max(), own[25:37].max()) / (1.0 + np.abs(s[r][25:37]).max()))
                print('pair physics: arena %d robot %d outside the bars (config %.2e, velocity %.2e); the oracle under float32 rounding of its own state / one ulp up / down: %.2e, %.2e' % (a, r, ce, ve, oc, ov))
                assert ce < max(1e-4, 4.0 * oc) and ve < max(1e-3, 4.0 * ov), (a, r, ce, ve, oc, ov)
                out['n_ill'] = out.get('n_ill', 0) + 1
                continue
            out['config'].append(ce)
            out['vel'].append(ve)
    E.close()
    c, v = np.array(out['config']), np.array(out['vel'])
    assert out['n_rows'] >= n_arenas // 2 and out['n_felt'] >= n_arenas // 3, (out['n_rows'], out['n_felt'])
    assert c.max() < 1e-4 and v.max() < 1e-3, (np.sort(c)[-6:], np.sort(v)[-6:])     # every robot of every arena (measured: 4e-6 / 2e-5) ...
    assert out.get('n_ill', 0) <= cap_ill, out['n_ill']                                     # ... but at most one that is ill-conditioned in the oracle itself (observed: 0 - 1 of 96)
    w = np.array(out['who'])
    # who-touches-whom from this build's contact classes (the env's real, unscripted bookkeeping path) against the oracle's restatement
    assert w[:, 0].mean() > 0.9 and w[:, 1].mean() > 0.9 and (w[:, 2] == 4).sum() >= 3 and (w[:, 2] == 1).sum() >= 1, (w[:, 0].mean(), w[:, 1].mean(), w[:, 2].tolist())
    return dict(who0_agree=float(w[:, 0].mean()), who_taker_agree=float(w[:, 1].mean()), who0_hist=np.bincount(w[:, 2] + 1, minlength=6).tolist(), config_median=float(np.median(c)), config_max=float(c.max()), vel_median=float(np.median(v)), vel_max=float(v.max()), arenas_with_rows=out['n_rows'],
                arenas_felt=out['n_felt'])


def arena_records(rows_a, cnt_a, ep, a):
    """the arena's boxes + the flag as the oracle's shape records (what check_pair_physics_against_oracle hands to B.substep_pair)"""
    rec = np.array([[rows_a[b][0] - rows_a[b][3], rows_a[b][0] + rows_a[b][3], rows_a[b][1] - rows_a[b][4], rows_a[b][1] + rows_a[b][4],
                     rows_a[b][2] - rows_a[b][5], rows_a[b][2] + rows_a[b][5], 0.0, 0.0] for b in range(cnt_a)], dtype=np.float64).reshape(-1, 8)
    fl = np.array([ep['flag_x'][a], ep['flag_y'][a], ep['flag_z'][a]], dtype=np.float64)
    rec = np.vstack([rec, [[fl[0] - 0.05, fl[0] + 0.05, fl[1] - 0.05, fl[1] + 0.05, fl[2] - 0.25, fl[2] + 0.25, 0.0, 0.0]]]).astype(np.float32).astype(np.float64)
    rec[0, 3] += 1.0; rec[1, 2] -= 1.0; rec[2, 1] += 1.0; rec[3, 0] -= 1.0        # the walls are solid outwards for contacts (SEPMC_WALL_SOLID)
    rec[0:2, 0] -= 1.0; rec[0:2, 1] += 1.0; rec[2:4, 2] -= 1.0; rec[2:4, 3] += 1.0      # ... and longer by the same at both ends: the corners are closed
    return rec.astype(np.float32).astype(np.float64)


def oracle_pair_step(B, st_pair, act_pair, rec, mu, push_trace, r32=False, ulp=0):
    """One control step (ten substeps) of the oracle's two-robot physics from the given states; r32 rounds both states to float32 between
    substeps -- how far that moves the result is the step's conditioning in the oracle itself."""
    near, s = [], []
    for r in range(2):
        p = st_pair[r, 0:3]
        sel = np.nonzero((p[0] >= rec[:, 0] - 0.9) & (p[0] <= rec[:, 1] + 0.9) & (p[1] >= rec[:, 2] - 0.9) & (p[1] <= rec[:, 3] + 0.9) & (p[2] <= rec[:, 5] + 0.9))[0][:8]
        near.append(rec[sel]); s.append(st_pair[r].copy())
    tgt = [np.clip(s[r][13:25] + np.asarray(act_pair[r], np.float64), -3.0, 3.0) for r in range(2)]
    if ulp:                                   # the same step from one float32 ulp up (+1) / down (-1) in every position coordinate of robot 0
        x = s[0].astype(np.float32)
        x[0:3] = np.nextafter(x[0:3], np.float32(np.inf * ulp)); x[13:25] = np.nextafter(x[13:25], np.float32(np.inf * ulp))
        s[0] = x.astype(np.float64)
    for k in range(10):
        tau = [np.clip(50.0 * (tgt[r] - s[r][13:25]) - 0.5 * s[r][25:37], -16.0, 16.0) for r in range(2)]
        push = [push_trace[r, k, 1:4] if push_trace[r, k, 0] > 0.5 else None for r in range(2)]
        s[0], s[1], _ = B.substep_pair(s[0], s[1], tau[0], tau[1], mu, near[0], near[1], 0.5 / 0.9, push[0], push[1])
        if r32:
            s = [x.astype(np.float32).astype(np.float64) for x in s]
    return s


def check_free_running_against_oracle_env(lib_path, n_steps=4, prop_type=None, element_sets=((0, 0, 0), (1, 1, 1)), noisy=False):
    """End to end, nothing scripted: the engine and the oracles assembled into a CPU chase-tag env (oracle/free_run.py: NumPy env logic,
    analytic rays and visibility segments on the real arena, the two-robot C physics, the oracle's contact classes) start from the same
    uniforms, get the same actions and are compared after every control step: both observations, both states, flag, roles, rewards, done."""
    from oracle import free_run as FR
    from lifelike_agility_and_play_amd import mocap, urdf_model, epmc_capi, sepmc_capi
    from parity_common import quat_align
    blob, table, init = urdf_model.default_model_blob(), mocap.load_mocap('', 0.02), epmc_capi.default_init_state()
    worst = dict(state=0.0, percep_same=1.0, ill_conditioned=0)
    from conftest import make_oracle_batch
    from oracle import oracle as orc
    B1 = make_oracle_batch(orc, blob, table, n_envs=1, kd=0.5, max_tau=16.0)
    for elements in element_sets:
        cfg = env_config(elements, noisy)
        if prop_type is not None:
            cfg['prop_type'] = list(prop_type)
        P3 = 3 * sum({'joint_pos': 12, 'joint_vel': 12, 'root_lin_vel_loc': 3, 'root_ang_vel_loc': 3, 'e_g': 3}[k] for k in cfg['prop_type']) + 36   # prop | prop_a
        cfg['env_randomize_config']['disturb_force_config'] = {'start_time': 0.0, 'interval_time': 1.0, 'duration_time': 0.5, 'horizontal_force': [10, 50], 'vertical_force': [0, 10]}
        n = 3
        E = make_engine(cfg, n, lib_path, seed=1)
        runs = [FR.SepmcFreeRun(cfg, blob, table, init, seed=0) for _ in range(n)]
        U = np.full((n, sepmc_capi.LLS_MAX_DRAWS), 0.5, np.float32)
        obs_o = []
        for i, r in enumerate(runs):
            r.draws = FR.SharedDraws(10 * sum(elements) + i)
            obs_o.append(r.reset())
            u = r.draws.take()
            assert len(u) <= sepmc_capi.LLS_MAX_DRAWS
            U[i, :len(u)] = u
        E.reset(draws=U)
        rng = np.random.default_rng(sum(elements))

        parted = set()          # arenas whose two trajectories have parted on a step that is ill-conditioned in the oracle itself (below)

        def compare(t, obs_o, rew_o=None, done_o=None, pre=None):
            obs_e, st_e, ep = E.obs().astype(np.float64), E.state().astype(np.float64), E.episode()
            tol = 1e-5 * 3.0 ** t
            for i, r in enumerate(runs):
                if i in parted:
                    continue
                errs = [np.abs(quat_align(st_e[i][k], r.env.states[k]) - r.env.states[k]) for k in range(2)]
                if pre is not None and any(e[:7].max() >= tol or e[13:25].max() >= 5 * tol for e in errs):
                    # Outside the bars.  Legitimate only if the step is ill-conditioned in the ORACLE: from the engine's own pre-step state the
                    # engine must agree with the oracle's step to the single-step bars, and merely rounding the oracle's state to float32
                    # between substeps must move the oracle's result by at least a quarter of what separates the two trajectories.
                    st_pre, ep_pre, (rows, cnt), tr, act = pre
                    rec = arena_records(rows[i], cnt[i], ep_pre, i)
                    mu = float(np.float32(ep_pre['friction'][i]) * np.float32(0.9))
                    s = oracle_pair_step(B1, st_pre[i], act[i], rec, mu, tr[i])
                    s32 = oracle_pair_step(B1, st_pre[i], act[i], rec, mu, tr[i], r32=True)
                    own = max(np.abs(quat_align(s32[k], s[k]) - s[k])[[*range(7), *range(13, 25)]].max() for k in range(2))
                    sep = max(max(e[:7].max(), e[13:25].max()) for e in errs)
                    from_engine_state = max(np.abs(quat_align(st_e[i][k], s[k]) - s[k])[[*range(7), *range(13, 25)]].max() for k in range(2))
                    print('free run: arena %d of %s parts from the oracle env at step %d (%.2e); the oracle step from the engine state agrees to %.2e, '
                          'its own float32-rounding deviation is %.2e' % (i, elements, t, sep, from_engine_state, own))
                    assert from_engine_state < max(1e-4, 4.0 * own) and sep < 4.0 * own, (elements, i, t, sep, from_engine_state, own)
                    parted.add(i); worst['ill_conditioned'] += 1
                    continue
                np.testing.assert_allclose([ep['flag_x'][i], ep['flag_y'][i]], r.env.target_pos[:2], atol=1e-6)
                assert bool(ep['with_flag0'][i] > 0.5) == r.env.with_flag[0]
                for k in range(2):
                    err = errs[k]
                    worst['state'] = max(worst['state'], err[:7].max() / tol)
                    assert err[:7].max() < tol and err[13:25].max() < 5 * tol, (elements, i, k, t, err[:7].max(), err[13:25].max())
                    assert obs_e[i][k].shape == np.asarray(obs_o[i][k]).shape == (P3 + 830,)
                    np.testing.assert_allclose(obs_e[i][k][:P3], np.asarray(obs_o[i][k])[:P3], atol=2e-3 + 500 * tol)              # prop (joint rates up to 30 rad/s), prop_a
                    pe, po = obs_e[i][k][P3:P3 + 778], np.asarray(obs_o[i][k])[P3:P3 + 778]
                    same = np.abs(pe - po) < 2e-3 + 20 * tol
                    worst['percep_same'] = min(worst['percep_same'], same.mean())
                    assert same.mean() > 0.97, (elements, i, k, t, same.mean())
                    np.testing.assert_allclose(obs_e[i][k][P3 + 778:], np.asarray(obs_o[i][k])[P3 + 778:], atol=2e-3 + 20 * tol)      # percept_vec .. control_spd
            if rew_o is not None:
                rew_e, done_e, _ = E.reward_done()
                for i in range(n):
                    if i in parted:
                        continue
                    assert bool(done_e[i]) == bool(done_o[i]), (elements, i, t)
                    np.testing.assert_allclose(rew_e[i], rew_o[i], atol=1e-6)
        compare(0, obs_o)
        for t in range(n_steps):
            act = (rng.normal(size=(n, 2, 12)) * 0.135).astype(np.float32)
            outs = [r.step([act[i][0].astype(np.float64), act[i][1].astype(np.float64)]) for i, r in enumerate(runs)]
            used = [r.draws.take() for r in runs]
            k = max(1, max(len(u) for u in used))
            D = np.full((n, k), 0.5, np.float32)
            for i, u in enumerate(used):
                D[i, :len(u)] = u
            E.set_step_draws(D)
            pre = (E.state().astype(np.float64), E.episode(), E.boxes())
            E.step_host(act)
            compare(t + 1, [o[0] for o in outs], [o[1] for o in outs], [o[2] for o in outs], pre=pre + (E.push_trace().astype(np.float64), act))
            if any(o[2] for o in outs):
                break
        assert len(parted) <= 1, parted                    # (observed: one arena of the (1, 1, 1) set under the cone, none under the pyramid)
        E.close()
    return worst


# ------------------------------------------------------------------------------------------------------------------------------------------
# Game-level statistics, engine against oracle env, ONE protocol (round-3 review, "What's weak" #3)
# ------------------------------------------------------------------------------------------------------------------------------------------
GAME_MAX_STEPS = 700


def _game_cfg():
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import rollout_sepmc_policy as R
    cfg = R.env_config(1)
    cfg['max_steps'] = GAME_MAX_STEPS              # every game is played to ITS end: a catch, robot 0 down, or this many steps
    return cfg


def _who0(touch_row):
    """CTG:426-440: the FIRST contact record of robot 0 decides; this build lists plane / boxes, then the flag, then the other robot"""
    return 1 if touch_row[0] else (2 if touch_row[1] else (4 if touch_row[2] else -1))


def _oracle_game(seed):
    """One chase-tag game of the float64 oracle env (oracle/free_run.py), both robots driven by the reference's trained policy, drawing from a
    recorded stream: (length, end reason as the engine's bits, arena-steps whose first contact record of robot 0 names robot 1, the uniforms)."""
    from oracle import free_run as FR, epmc_oracle as EO
    from oracle.sepmc_policy import SepmcPolicy
    from lifelike_agility_and_play_amd import epmc_capi, mocap, urdf_model
    cfg = _game_cfg()
    run = FR.SepmcFreeRun(cfg, urdf_model.default_model_blob(), mocap.load_mocap('', 0.02), epmc_capi.default_init_state(), seed=0)
    run.draws = FR.SharedDraws(seed)
    pol = SepmcPolicy(os.path.join(ROOT, 'tests', 'golden', 'sepmc_policy.npz'), 2)
    obs = run.reset()
    u0, us, why, named = run.draws.take(), [], 0, 0
    for t in range(cfg['max_steps'] + 1):
        a = pol.act(np.asarray(obs, np.float64).reshape(2, -1))
        o = run.step([a[0], a[1]])
        obs = o[0]
        us.append(run.draws.take())
        named += int(_who0(run.touch[0]) == 4)
        if o[2]:
            why = 1 if EO.check_fall(run.env.states[0][3:7]) else (2 if run.env.counter >= run.env.max_steps else 8)
            break
    assert why, 'the oracle game did not end by max_steps'
    return len(us), why, named, u0, us


def blocked_policy(npz_path, n_rows, block=64):
    """oracle.sepmc_policy.SepmcPolicy whose conv stacks run over `block` rows at a time, and only over the rows named in `.alive` (a boolean mask, None = all; the
    rows left out get zero conv features and their actions mean nothing).  For the rows that are evaluated: bit for bit the same actions and LSTM state as the plain
    policy (every matrix product of a conv layer is a stack of per-row products, so a row's numbers do not depend on which rows share the call; the dense layers and
    the LSTMs still see the whole batch, whose SIZE is what a BLAS kernel choice could depend on -- tests/test_oracle_game_cache.py holds all of that).  The 512-game
    statistic evaluates 1024 rows a step for up to 700 steps while its games end after 290 on average: the conv stacks are most of that time."""
    from oracle.sepmc_policy import SepmcPolicy

    class Blocked(SepmcPolicy):
        alive = None

        def _percepts(self, p2d, p1d, pfr, k):
            n = p2d.shape[0]
            rows = np.arange(n) if self.alive is None else np.flatnonzero(self.alive)
            if len(rows) == n and n <= block:
                return SepmcPolicy._percepts(self, p2d, p1d, pfr, k)
            out = None
            for i in range(0, len(rows), block):
                r = rows[i:i + block]
                part = SepmcPolicy._percepts(self, p2d[r], p1d[r], pfr[r], k)
                if out is None:
                    out = tuple(np.zeros((n, q.shape[1])) for q in part)
                for o, q in zip(out, part):
                    o[r] = q
            return out
    return Blocked(npz_path, n_rows)


def game_cache_extra():
    """What besides the source files decides an oracle game: the game config as the tests build it"""
    import json
    return 'sepmc ' + json.dumps(_game_cfg(), sort_keys=True, default=str)


def oracle_games(seeds, procs=None):
    """The oracle env's games of `seeds` as (length, why, named, reset uniforms, step uniforms): from tests/golden/oracle_games.npz while that fixture is the CURRENT
    oracle's (tests/oracle_game_cache.py: a sha256 over every file a game depends on; LL_LIVE_ORACLE_GAMES=1 ignores it), otherwise played now on `procs` host processes."""
    import gc
    import multiprocessing as mp
    import bench
    import oracle_game_cache as C
    res = C.load('sepmc', game_cache_extra(), seeds)
    if res is not None:
        return res, 'the oracle games come from tests/golden/oracle_games.npz, whose source key is current'
    procs = procs or bench.effective_cores()[0]
    gc.collect()                                              # (no dead engine objects for the forked workers to finalise)
    with mp.get_context('fork').Pool(procs) as p:
        return p.map(_oracle_game, list(seeds), chunksize=1), 'played live on %d processes' % procs


def check_game_statistics(lib_path, n_arenas=512, procs=None, frac_tol=0.03, len_tol=0.03, ks_p=0.5, n_se=2.0, seed0=5000):
    """The strategic level's counterpart of parity_common.check_rollout_statistics, at the level the game is decided on: the engine and the float64
    oracle env play the SAME games -- same spawn poses, friction and pushes (the oracle env's uniforms are recorded and handed to the engine draw by
    draw), the reference's trained policy on both robots acting on each side's own observations -- every game to its end on both sides (a catch
    CTG:426-470 / robot 0 down / max_steps).  Distributions must agree: end-reason fractions, mean length and the Kolmogorov-Smirnov test on the
    lengths under two-sample bars (`frac_tol` / `len_tol` or `n_se` standard errors, whichever is larger -- round 6: 512 games at TWO standard errors; the games are
    seeded seed0 + i, so a run is reproducible game by game), and the fraction of arena-steps whose first contact record of robot 0 names the other robot."""
    import gc
    import multiprocessing as mp
    from scipy import stats as sst
    from oracle.sepmc_policy import SepmcPolicy
    from lifelike_agility_and_play_amd import sepmc_capi
    import bench
    n = n_arenas
    res, src = oracle_games([seed0 + i for i in range(n)], procs)
    print('chase-tag game statistics: %d games, oracle seeds %d .. %d (%s), engine seed 3, bars at %.1f standard errors (floors %.3f / %.3f)' % (n, seed0, seed0 + n - 1, src, n_se, frac_tol, len_tol))
    len_o, why_o, named_o = np.array([r[0] for r in res]), np.array([r[1] for r in res]), np.array([r[2] for r in res])
    cfg = _game_cfg()
    E = make_engine(cfg, n, lib_path, seed=3)
    U = np.full((n, sepmc_capi.LLS_MAX_DRAWS), 0.5, np.float32)
    for i, r in enumerate(res):
        assert len(r[3]) <= sepmc_capi.LLS_MAX_DRAWS
        U[i, :len(r[3])] = r[3]
    E.reset(draws=U)
    pol = blocked_policy(os.path.join(ROOT, 'tests', 'golden', 'sepmc_policy.npz'), 2 * n)
    obs = E.obs()
    alive = np.ones(n, bool); len_e = np.zeros(n, int); why_e = np.zeros(n, int); named_e = np.zeros(n, int)
    for t in range(cfg['max_steps'] + 1):
        used = [r[4][t] if t < len(r[4]) else [] for r in res]
        D = np.full((n, max(1, max(len(u) for u in used))), 0.5, np.float32)
        for i, u in enumerate(used):
            D[i, :len(u)] = u
        pol.alive = np.repeat(alive, 2)                       # (rows 2 i, 2 i + 1 are arena i's robots; a finished arena's actions no longer matter)
        a = pol.act(obs.astype(np.float64).reshape(2 * n, -1)).reshape(n, 2, 12)
        E.set_step_draws(D)
        E.step_host(a.astype(np.float32))
        obs = E.obs()
        r_, d_, w_ = E.reward_done()
        named_e += alive & (E.episode()['who0'] == 4)
        len_e += alive
        newly = alive & d_
        why_e[newly] = w_[newly]
        alive &= ~d_
        if not alive.any():
            break
    E.close()
    assert not alive.any(), int(alive.sum())
    fr = lambda w, bit: float(((w & bit) != 0).mean())
    o = dict(n=n, caught=(fr(why_e, 8), fr(why_o, 8)), fell=(fr(why_e, 1), fr(why_o, 1)), timed_out=(fr(why_e & ~9, 2), fr(why_o & ~9, 2)),
             mean_len=(float(len_e.mean()), float(len_o.mean())), ks_p=float(sst.ks_2samp(len_e, len_o).pvalue),
             named=(float(named_e.sum() / len_e.sum()), float(named_o.sum() / len_o.sum())),
             same_end=float(((why_e & 11) == (why_o & 11)).mean()), same_step=float((len_e == len_o).mean()))
    print('game statistics, chase tag, %d games (engine / oracle): caught %.3f / %.3f, robot 0 down %.3f / %.3f, timed out %.3f / %.3f, mean length %.1f / %.1f, KS p %.3f; '
          'arena-steps whose first contact record names the other robot %.4f / %.4f; same end reason %.3f, same end step %.3f'
          % ((n,) + o['caught'] + o['fell'] + o['timed_out'] + o['mean_len'] + (o['ks_p'],) + o['named'] + (o['same_end'], o['same_step'])))
    # Games decorrelate between the two simulators (a fifth end at the same step): two samples of one distribution, two-sample bars
    # (parity_common.two_sample_bars; measured on MI355X, 256 games: caught 0.742 / 0.727, robot 0 down 0.109 / 0.109, timed out 0.148 / 0.164,
    # mean length 312.7 / 335.2 with a standard error of the difference of 18, KS p 0.12)
    from parity_common import two_sample_bars
    two_sample_bars('chase tag', {k: o[k] for k in ('caught', 'fell', 'timed_out')}, len_e, len_o, o['ks_p'], n, floor_frac=frac_tol, floor_len=len_tol, n_se=n_se, ks_floor=min(ks_p, 0.01))
    assert abs(o['named'][0] - o['named'][1]) <= max(0.002, 0.5 * o['named'][1]), o['named']
    return o
