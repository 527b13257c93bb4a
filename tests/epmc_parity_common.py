"""EPMC parity checks shared by the CPU run of the kernel source (tests/emul) and the GPU run of libllenv.so.
float32 engine vs (a) the reference's own outputs in tests/golden/epmc_golden.npz, (b) the float64 NumPy oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lifelike_agility_and_play_amd import epmc_capi, urdf_model  # noqa: E402
from oracle import epmc_oracle as eo  # noqa: E402
from test_epmc_oracle_golden import env_config, scripted_rays, ScriptedRays, percep_checks, make_env  # noqa: E402

OBS_TOL = 3e-5        # float32 engine vs float64 reference: positions reach 25 m, ray lengths 20 m
REW_TOL = 2e-6


def golden():
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'epmc_golden.npz'))


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


BASE_SPEC = {}         # physics-spec switches (include/llenv_model.h LLM_SPEC_*) every engine and every oracle call of this module runs under: spec_variant()


class spec_variant:
    """with spec_variant(friction_mode=0): ... -- the checks of this module (and of sepmc_parity_common) on a spec VARIANT both implementations
    carry (ll_epmc_set_spec_param / ll_sepmc_set_spec_param on the engines, orc.set_spec on the oracle)."""

    def __init__(self, **spec):
        self.spec = spec

    def __enter__(self):
        from oracle import oracle as orc
        self.saved = dict(BASE_SPEC)
        BASE_SPEC.update(self.spec)
        orc.reset_spec(); orc.set_spec(**BASE_SPEC)

    def __exit__(self, *a):
        from oracle import oracle as orc
        BASE_SPEC.clear(); BASE_SPEC.update(self.saved)
        orc.reset_spec(); orc.set_spec(**BASE_SPEC)


def make_engine(cfg_dict, n_envs, lib_path, **kw):
    cfg = epmc_capi.make_epmc_config(n_envs, cfg_dict, **kw)
    E = epmc_capi.EpmcEngine(cfg, urdf_model.default_model_blob(), lib_path=lib_path)
    E.set_spec(**BASE_SPEC)
    return E


def script3(call0):
    hits, fracs = [], []
    for k, n in enumerate((325, 128, 325)):
        h, f = scripted_rays(call0 + k, n)
        hits.append(h); fracs.append(f)
    return np.concatenate(hits), np.concatenate(fracs)


def check_terrain_and_reset_against_goldens(lib_path):
    """BSE reset() inside the engine, from the reference's own draws: bodies, target, friction, command period, start pose,
    push force and first observation of the 24 golden cases."""
    g = golden()
    for k in range(len(g['t_element'])):
        aux = None if np.isnan(g['t_aux'][k]) else float(g['t_aux'][k])
        E = make_engine(env_config(int(g['t_element'][k]), aux=aux), 1, lib_path)
        u = np.full(epmc_capi.LLE_MAX_DRAWS, 0.5, np.float32)
        n = int(g['t_n_draws'][k])
        u[:n] = draws_to_uniforms(g['t_draws'][k][:n])
        h, f = script3(0)
        E.script_reset_rays(h[None], f[None])
        E.reset(draws=u[None], prev_orn=g['t_prev_orn'][k][None])
        rows, cnt = E.statics()
        ns = int(g['t_n_statics'][k])
        assert cnt[0] == ns, (k, cnt[0], ns)
        np.testing.assert_allclose(rows[0, :ns], g['t_statics'][k][:ns], rtol=1e-5, atol=2e-5)
        ep = E.episode()
        np.testing.assert_allclose([ep['target_x'][0], ep['target_y'][0], ep['target_z'][0]], g['t_target'][k], atol=2e-5)
        assert abs(ep['friction'][0] - g['t_friction'][k]) < 1e-5 and int(ep['cmd_vary_freq'][0]) == int(g['t_cmd_freq'][k])
        np.testing.assert_allclose([ep['push_fx'][0], ep['push_fy'][0], ep['push_fz'][0]], g['t_push_force'][k], rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(E.state()[0], g['t_init_state'][k], atol=1e-6)
        np.testing.assert_allclose(E.obs()[0], g['t_reset_obs'][k], rtol=OBS_TOL, atol=OBS_TOL)
        E.close()


def check_scripted_episodes_against_goldens(lib_path):
    """The engine's whole step() control flow against the reference's own outputs: 8 scripted episodes driven exactly as
    gen_epmc_golden.py drove PlayGroundEnv through its fake BulletClient -- robot state and ray answers supplied, everything
    else (ray end points, observation, rewards, termination, joystick targets, push schedule, info) computed by the kernel."""
    g = golden()
    n_done = 0
    for e in range(len(g['e_element'])):
        noise_on = not np.isnan(g['e_noise'][e][0])
        obs_rand = {'pos_x_bias': [-0.1, 0.1], 'pos_y_bias': [-0.1, 0.1], 'yaw_bias': [-0.2, 0.2], 'pos_z_bias': [-0.02, 0.02]} if noise_on else None
        cmd = {0: (7, 8), 1: (25, 200)}.get(e, (9999, 10000))
        cfg = env_config(int(g['e_element'][e]), obs_rand=obs_rand, cmd_range=cmd)
        # replay the oracle alongside, only to learn how many draws the reset and each step consume
        log = g['e_draws'][e][:g['e_n_draws'][e]]
        orc = make_env(g, cfg, g['e_prev_orn'][e])
        od = eo.LoggedDraws(log)
        orays = ScriptedRays(int(g['e_ray_call0'][e]))
        orc.reset(od, orays)
        n_reset = od.i
        uni = draws_to_uniforms(log)
        E = make_engine(cfg, 1, lib_path)
        u = np.full(epmc_capi.LLE_MAX_DRAWS, 0.5, np.float32)
        u[:n_reset] = uni[:n_reset]
        call = int(g['e_ray_call0'][e])
        h, f = script3(call); call += 3
        E.script_reset_rays(h[None], f[None])
        E.reset(draws=u[None], prev_orn=g['e_prev_orn'][e][None])
        np.testing.assert_allclose(E.obs()[0], g['e_reset_obs'][e], rtol=OBS_TOL, atol=OBS_TOL)
        rf, rt, _, _ = E.rays()
        np.testing.assert_allclose(rf[0], g['e_ray_from'][e][0], atol=5e-5); np.testing.assert_allclose(rt[0], g["e_ray_to"][e][0], atol=5e-5)
        for t in range(int(g['e_n'][e])):
            state = g['e_state'][e][t]
            i0 = od.i
            orc.step(g['e_action'][e][t], od, lambda k, tgt, fo: state if k == 9 else None, orays)
            step_u = uni[i0:od.i]
            h, f = script3(call); call += 3
            E.step_scripted(g['e_action'][e][t][None], state[None], h[None], f[None], draws=(step_u[None] if len(step_u) else None))
            obs = E.obs()[0].astype(np.float64)
            r, d, why = E.reward_done()
            if t < g['e_obs_full'].shape[1]:
                np.testing.assert_allclose(obs, g['e_obs_full'][e][t], rtol=OBS_TOL, atol=OBS_TOL)
                rf, rt, _, _ = E.rays()
                np.testing.assert_allclose(rf[0], g['e_ray_from'][e][t + 1], atol=5e-5); np.testing.assert_allclose(rt[0], g["e_ray_to"][e][t + 1], atol=5e-5)
            np.testing.assert_allclose(np.concatenate([obs[:135], obs[913:]]), g['e_obs_core'][e][t], rtol=OBS_TOL, atol=OBS_TOL)
            np.testing.assert_allclose(percep_checks(obs), g['e_obs_checks'][e][t], rtol=1e-4, atol=2e-2)      # sums of 325 float32 terms
            assert abs(float(r[0]) - g['e_reward'][e][t]) < REW_TOL and bool(d[0]) == bool(g['e_done'][e][t]), (e, t, r[0], g['e_reward'][e][t], why)
            ep = E.episode()
            np.testing.assert_allclose([ep['target_x'][0], ep['target_y'][0], ep['target_z'][0]], g['e_target'][e][t], rtol=1e-5, atol=1e-4)
            assert abs(ep['target_spd'][0] - g['e_target_spd'][e][t]) < 1e-5
            tr = E.push_trace()[0]
            assert ((tr[:, 0] > 0.5) == g['e_force_on'][e][t]).all(), (e, t)
            np.testing.assert_allclose(tr[:, 1:4], g['e_force'][e][t], rtol=1e-5, atol=1e-4)
            if d[0]:
                n_done += 1
                np.testing.assert_allclose(E.info()[0], g['e_info'][e], rtol=2e-4, atol=2e-6)
        assert od.exhausted()
        E.close()
    assert n_done == 3


def check_ray_casting_against_oracle(lib_path, n_envs=12, n_steps=6, seed=3):
    """Free-running engine (Philox terrain, analytic rays) vs the oracle's cast_rays() on the engine's own terrain and ray end
    points: hit flags and fractions; then the percep part of the observation from those."""
    n_checked = 0
    for element in (1, 2, 3, 0):
        E = make_engine(env_config(element), n_envs, lib_path, seed=seed + element)
        E.reset()
        rng = np.random.default_rng(seed)
        for t in range(n_steps):
            s = E.state()
            s[:, 0] += rng.uniform(0.5, 2.5, n_envs) * (t > 0); s[:, 1] = rng.uniform(-0.4, 0.4, n_envs); s[:, 2] = rng.uniform(0.2, 0.45, n_envs)
            if t >= 2:      # any heading, rolled and pitched: the box lists of the ray families follow the bundles' bounds as they lie in the world
                from scipy.spatial.transform import Rotation as Rot
                s[:, 3:7] = Rot.from_euler('zyx', np.c_[rng.uniform(-np.pi, np.pi, n_envs), rng.uniform(-0.6, 0.6, n_envs), rng.uniform(-0.6, 0.6, n_envs)]).as_quat()
            E.set_state(s)
            E.step_host(np.zeros((n_envs, 12), np.float32))
            r, d, why = E.reward_done()
            rows, cnt = E.statics()
            f, to, hit, frac = E.rays()
            obs = E.obs()
            for i in range(n_envs):
                if d[i]:
                    continue                                    # re-seeded inside the step only with auto_reset; here its rays are of the old episode
                oh, ofr = eo.cast_rays(f[i].astype(np.float64), to[i].astype(np.float64), rows[i, :cnt[i]].astype(np.float64))
                agree = oh == hit[i]
                assert agree.mean() > 0.995, (element, t, i, agree.mean())       # grazing rays may fall either side in float32
                np.testing.assert_allclose(frac[i][agree & oh], ofr[agree & oh], rtol=2e-4, atol=2e-5)
                p2d, p1d, pf = eo.percep_from_rays(f[i].astype(np.float64), to[i].astype(np.float64), hit[i], frac[i].astype(np.float64))
                a0 = 135
                np.testing.assert_allclose(obs[i, a0:a0 + 325], p2d, atol=3e-5)
                np.testing.assert_allclose(obs[i, a0 + 325:a0 + 453], p1d, rtol=1e-5, atol=3e-5)
                np.testing.assert_allclose(obs[i, a0 + 453:a0 + 778], pf, rtol=1e-5, atol=3e-5)
                n_checked += 1
        E.close()
    assert n_checked > 100
    return n_checked


def check_engine_against_host_build(emul_lib_path, n_envs=4096, steps=7, element=1, seed=5, gpu_lib=None, report_only=False):
    """The net under the EPMC step kernels at BASELINE config 4's size (the chase-tag one: sepmc_parity_common.check_engine_against_emulation): every entry of every observation, reward, done flag and reason,
    episode record and terrain the HIP kernel writes, env by env against the HOST build of the same kernel source.  max_steps = 3: every env runs out of time at steps 2 and 5 and RE-SEEDS inside the step --
    new terrain, new target, new friction, the first observation of the new episode -- the rare path where a misplaced register copy would show (HISTORY.md).  The host build takes the engine's state before
    every step; envs whose own state differs by more than 5e-3 after it are counted, capped at 1 % and left to the physics parity tests."""
    cfg = env_config(element, cmd_range=(25, 200))
    cfg['max_steps'] = 3
    G = make_engine(cfg, n_envs, gpu_lib, auto_reset=1, seed=seed)
    H = make_engine(cfg, n_envs, emul_lib_path, auto_reset=1, seed=seed)
    G.reset(); H.reset()
    out = dict(reseeded=0, left_out=0, rough=0, worst_prop=0.0, worst_tail=0.0, ray_mismatch=0.0)
    vel = np.zeros(33, bool); vel[12:30] = True
    prop_vel = np.concatenate([np.tile(vel, 3), np.zeros(36, bool)])

    recent = []

    def compare(label, keep):
        og, oh = G.obs().astype(np.float64), H.obs().astype(np.float64)
        assert np.isfinite(og).all(), label
        sg, sh = G.state().astype(np.float64), H.state().astype(np.float64)
        scale = 1.0 + np.maximum(np.abs(sh[:, 7:13]).max(-1, keepdims=True), np.abs(sh[:, 25:37]).max(-1, keepdims=True))
        ds = np.abs(sg - sh); ds[:, 7:13] /= scale; ds[:, 25:37] /= scale            # (base twist and joint rates relative to the fastest of them)
        rough = ds.max(-1) > 5e-3
        out['rough'] += int((keep & rough).sum())
        if not label.endswith('only'):
            recent.append(rough)
        rough = np.logical_or.reduce(recent[-3:])                  # (the observation carries the two older proprioceptive frames)
        ok = keep & ~rough
        dprop = np.abs(og[:, :135] - oh[:, :135]) / np.where(prop_vel, scale, 1.0)
        rays = np.abs(og[:, 135:913] - oh[:, 135:913]) > 2e-3                       # a ray grazing an edge may answer differently: counted
        tail = np.abs(og[:, 913:] - oh[:, 913:])
        eg, eh = G.episode(), H.episode()
        rec = np.zeros(n_envs, bool)
        for k in eg:
            if k not in ('total_spd', 'max_spd', 'last_pos_diff_len'):
                rec |= np.abs(eg[k].astype(np.float64) - eh[k]) > 1e-5 * (1.0 + np.abs(eh[k]))
        res = dict(prop=int((ok & (dprop.max(-1) > 5e-3)).sum()), tail=int((ok & (tail.max(-1) > 5e-3)).sum()), episode_record=int((ok & rec).sum()))
        if ok.any():
            out['worst_prop'] = max(out['worst_prop'], float(dprop[ok].max())); out['worst_tail'] = max(out['worst_tail'], float(tail[ok].max()))
            out['ray_mismatch'] = max(out['ray_mismatch'], float(rays[ok].mean()))
            res['rays'] = int(rays[ok].mean() > 2e-3)
        out.setdefault('per_step', {})[label] = res
        if not report_only:
            assert sum(res.values()) == 0, (label, res)
            assert (keep & rough).mean() < 0.01, (label, (keep & rough).mean())

    def terrain_off():
        (rg_, ng_), (rh_, nh_) = G.statics(), H.statics()
        live = np.arange(rg_.shape[1])[None, :] < ng_[:, None]
        return (ng_ != nh_) | (np.abs(rg_ - rh_).max(-1) * live > 1e-4).any(-1)

    compare('reset', np.ones(n_envs, bool))
    assert not terrain_off().any()
    rng = np.random.default_rng(seed)
    for t in range(steps):
        act = (rng.normal(size=(n_envs, 12)) * 0.135).astype(np.float32)
        H.set_state(G.state())
        G.step_host(act); H.step_host(act)
        (rg, dg, wg), (rh, dh, wh) = G.reward_done(), H.reward_done()
        same = (dg == dh) & (wg == wh)
        out['left_out'] += int((~same).sum())
        out['reseeded'] += int((dg & same).sum())
        compare('step %d' % t, same)
        if dg.any():
            compare('step %d, re-seeding envs only' % t, same & dg)
            assert not (terrain_off() & same).any(), t                                                   # the terrain of the new episodes
        calm = same & (np.abs(G.state() - H.state()).max(-1) < 5e-3)
        if not report_only:
            assert np.abs(rg - rh)[calm].max() < 5e-3, (t, np.abs(rg - rh)[calm].max())
    if not report_only:
        assert out['left_out'] <= max(2, int(0.01 * n_envs * steps)), out
        assert out['reseeded'] >= n_envs, out                                   # every env re-seeded at least once
    G.close(); H.close()
    return out


def check_free_running_invariants(lib_path, n_envs=64, n_steps=80, element=1):
    """Real physics, real rays, Philox draws, auto-reset: size-independent properties of a random-policy run."""
    cfg = env_config(element, cmd_range=(25, 200))
    cfg['max_steps'] = 60                                    # so that episodes also end by time within the run
    E = make_engine(cfg, n_envs, lib_path, auto_reset=1, seed=9)
    E.reset()
    o0 = E.obs()
    assert np.isfinite(o0).all() and E.obs_dim == 916
    ep0 = E.episode()
    assert (ep0['friction'] >= 0.4).all() and (ep0['friction'] <= 3.0).all() and (ep0['counter'] == 0).all()
    rng = np.random.default_rng(1)
    done_total, pushed = 0, 0
    reasons = np.zeros(32, int)
    for t in range(n_steps):
        E.step_host(rng.normal(size=(n_envs, 12)).astype(np.float32) * 0.135)
        o, s = E.obs(), E.state()
        r, d, why = E.reward_done()
        assert np.isfinite(o).all() and np.isfinite(s).all() and np.isfinite(r).all()
        np.testing.assert_allclose(np.linalg.norm(s[:, 3:7], axis=1), 1.0, atol=1e-5)
        assert ((why != 0) == d).all()
        tgt = o[:, 913:915]
        np.testing.assert_allclose(np.linalg.norm(tgt, axis=1), 1.0, atol=1e-5)          # PGE:402 unit direction to the target
        assert (o[:, 135:460] >= -1e-6).all() and (o[:, 135:460] <= 2.0 + 1e-5).all()   # heights: ground .. wall tops
        assert (o[:, 588:913] >= 0).all() and (o[:, 588:913] <= 3.0 + 1e-4).all()       # front rays: at most their 3 m length
        ep = E.episode()
        assert (ep['counter'][d] == 0).all() and (ep['counter'][~d] >= 1).all()         # re-seeded inside the kernel
        done_total += int(d.sum())
        for w in why[d]:
            reasons[w] += 1
        pushed += int((E.push_trace()[:, :, 0] > 0.5).any(axis=1).sum())
    c = E.counters()
    assert c['env_steps'] == n_steps * n_envs and c['episodes'] == done_total and c['nonfinite'] == 0
    if n_steps >= 70:
        assert done_total > 0 and reasons[2] > 0 and pushed > 0                          # some episodes ran out of time; pushes happened
    E.close()
    return done_total


def check_multi_step_launch(lib_path, sizes=(12,), k=5, n_launches=4, element=1):
    """ll_epmc_step_random_n(sigma, k) == k x {ll_epmc_fill_random_actions(sigma); ll_epmc_step()}, bit for bit: state, the 916-float
    observation (rays included), rewards, done reasons, episode records (terrain seeds, targets, push schedule), counters.
    (Round 6: with the rays split off -- the default, LL_SPLIT_RAYS=2 -- a multi-step call IS a sequence of single launches; the fused multi-step build is what
    LL_SPLIT_RAYS=0 runs, and that is the one this check is about: it pins the switch for its engines.)"""
    import os
    prev_split = os.environ.get('LL_SPLIT_RAYS')
    os.environ['LL_SPLIT_RAYS'] = '0'
    try:
        return _check_multi_step_launch(lib_path, sizes, k, n_launches, element)
    finally:
        if prev_split is None:
            os.environ.pop('LL_SPLIT_RAYS', None)
        else:
            os.environ['LL_SPLIT_RAYS'] = prev_split


def _check_multi_step_launch(lib_path, sizes, k, n_launches, element):
    sg = float(np.exp(-2.0))
    for n in sizes:
        cfg = env_config(element)
        cfg['max_steps'] = 3 * k                                     # episodes end (time-out) and re-seed inside the launches
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
            ra, rb = A.reward_done(), B.reward_done()
            for x, y in zip(ra, rb):
                np.testing.assert_array_equal(x, y)
            ea, eb = A.episode(), B.episode()
            for key in ea:
                np.testing.assert_array_equal(ea[key], eb[key])
            assert A.counters() == B.counters()
        assert A.counters()['episodes'] > 0
        A.close(); B.close()


def check_parked_variant_equals_plain(lib_path, n=10, n_steps=40, element=1):
    """step_env<PARK = true> -- what the larger-batch GPU build runs: the episode scalars wait in the row scratch during the substep loop and the
    observation history is read after it -- against the plain variant on the HOST build (LL_EMUL_PARK=1 switches; there the two are the same
    float arithmetic, so everything must agree bit for bit: a field the substep loop changes and the parked variant forgets to keep would show
    here).  Pushes on, episodes ending and re-seeding inside the run."""
    import os
    sg = float(np.exp(-2.0))
    cfg = env_config(element)
    cfg['max_steps'] = 15
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
    assert A.counters() == B.counters() and A.counters()['episodes'] > 0 and pushed > 0
    A.close(); B.close()


def check_split_rays_equal_fused(lib_path, n=10, n_steps=40, elements=(1, 2, 3), multi=(1, 4), noise=False):
    """Round 6: the 778 rays of a row's observation cast by a kernel of their own behind the step kernel (epmc_step.hpp percept_rays, LL_SPLIT_RAYS) against the step
    kernel casting them itself, BIT FOR BIT: observation (every percept column), ray traces (end points, hit, fraction), state, rewards, episode records -- over runs in which
    episodes end and re-seed (new terrain) and, with `noise`, the observation noise of PGE:388-393 and :443-446 is on.  Per ray the two paths run the same expressions; a ray's
    answer is a minimum / maximum over boxes, so the order and pre-selection of the boxes cannot show.  On the GPU the switch is read when the engine is created, on the host
    build at every call: set for both."""
    import os
    sg = float(np.exp(-2.0))
    for element in elements:
        cfg = env_config(element)
        cfg['max_steps'] = 15
        if noise:
            cfg['obs_randomization'] = {'pos_x_bias': [-0.05, 0.05], 'pos_y_bias': [-0.05, 0.05], 'yaw_bias': [-0.1, 0.1], 'pos_z_bias': [-0.02, 0.02]}
        prev = os.environ.get('LL_SPLIT_RAYS')
        try:
            os.environ['LL_SPLIT_RAYS'] = '0'
            A = make_engine(cfg, n, lib_path, auto_reset=1, seed=4)
            os.environ['LL_SPLIT_RAYS'] = '2'
            B = make_engine(cfg, n, lib_path, auto_reset=1, seed=4)
            A.reset(); B.reset()
            for t in range(n_steps):
                k = multi[t % len(multi)]
                os.environ['LL_SPLIT_RAYS'] = '0'
                if k == 1:
                    A.fill_random_actions(sg); A.step()
                else:
                    A.step_random_n(sg, k)
                os.environ['LL_SPLIT_RAYS'] = '2'
                if k == 1:
                    B.fill_random_actions(sg); B.step()
                else:
                    B.step_random_n(sg, k)
                np.testing.assert_array_equal(A.obs(), B.obs())
                np.testing.assert_array_equal(A.state(), B.state())
                ra, rb = A.reward_done(), B.reward_done()
                # (k > 1: A ran the fused MULTI-step build, B single-step launches.  The two builds round the reward's exp / division chain one ulp apart on the GPU -- state
                #  and observation are bit-equal, and so is the reward between the fused and the split SINGLE-step paths: profiles/r06_epmc_multi_vs_single.txt)
                if k == 1:
                    np.testing.assert_array_equal(ra[0], rb[0])
                else:
                    np.testing.assert_allclose(ra[0], rb[0], rtol=3e-7, atol=1e-9)
                np.testing.assert_array_equal(ra[1], rb[1]); np.testing.assert_array_equal(ra[2], rb[2])
                ea, eb = A.episode(), B.episode()
                for key in ea:
                    if np.asarray(ea[key]).dtype.kind == 'f' and k > 1:
                        np.testing.assert_allclose(ea[key], eb[key], rtol=1e-6, atol=1e-7)      # (the episode's reward sums carry the same ulp)
                    else:
                        np.testing.assert_array_equal(ea[key], eb[key])
                if n <= 512:                                                              # (the ray trace is kept for engines of at most 512 rows)
                    for x, y in zip(A.rays(), B.rays()):                                  # end points, hit flags and fractions of the step's 778 rays per row
                        np.testing.assert_array_equal(x, y)
            assert A.counters() == B.counters() and A.counters()['episodes'] > 0
            o = A.obs()[:, 135:135 + 778]
            assert np.isfinite(o).all() and (o[:, :325] > 0.02).any() and (o[:, 453:] < 2.99).any()       # boxes were seen from above and ahead
            A.close(); B.close()
        finally:
            if prev is None:
                os.environ.pop('LL_SPLIT_RAYS', None)
            else:
                os.environ['LL_SPLIT_RAYS'] = prev


def check_trained_policies_traverse(lib_path, n_envs=8, horizon=(360, 560)):
    """SURVEY.md 8f-3 for the environmental level -- the only Bullet-facing check of this build's terrain contacts and 778 analytic rays: the
    reference's TRAINED EPMC policies (data/models/environmental_level_{hurdle,cube}.model, trained against PyBullet; the hole checkpoint
    does not traverse our bars course and is reported, not asserted: DESIGN.md 2, profiles/r03_epmc_hole_policy.txt) drive our PlayGround env closed-loop under the protocol of test_environmental_level_env.py: target speed
    3 m/s, pushes, friction 0.4 .. 1, argmax code.  They run 10 m over hurdles / up and down 10 and 25 cm steps and reach the target.  The
    LSTM cell is tpolicies' published lnlstm restated in oracle/epmc_policy.py; with any other gate order the same weights fall within 3 s
    (tools/rollout_epmc_policy.py, profiles/r03_epmc_policy_rollout.txt), so a passing run vouches for cell and physics together."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import rollout_epmc_policy as R
    out = {}
    for which, hz in zip(('hurdle', 'cube'), horizon):
        o = R.rollout(which, n_envs, hz, lib_path)
        reached, fell = int(((o['why'] & 4) != 0).sum()), int(((o['why'] & 1) != 0).sum())
        out[which] = dict(reached=reached, fell=fell, running=int(o['alive'].sum()), dist=float(o['dist'].mean()), steps=float(o['steps'].mean()))
        assert reached >= 0.7 * n_envs and fell <= 0.15 * n_envs, (which, out[which])
        assert o['dist'].mean() > 6.0, (which, out[which])                                      # metres along the course
    return out


def statics_to_records(rows):
    """ll_epmc_get_statics rows (creation order: a box, then its two edge cylinders if any) -> the kernel's box records."""
    recs, i = [], 0
    while i < len(rows):
        r = rows[i]
        assert r[0] == 0
        if r[4] > 0:
            rec = [r[1] - r[4], r[1] + r[4], r[2] - r[5], r[2] + r[5], r[3] - r[6], r[3] + r[6], 0.0, 0.0]
            if i + 2 < len(rows) + 0 and i + 1 < len(rows) and rows[i + 1][0] == 1:
                rec[6] = 1.0 if rows[i + 1][3] > r[3] else -1.0
                rec[7] = rows[i + 1][4]
                i += 2
            recs.append(rec)
        i += 1
    return np.array(recs, dtype=np.float64).reshape(-1, 8)


def oracle_control_step(B, orc, s0, act, push_trace, mu, near, r32=False, **spec):
    """Ten substeps of the oracle from state s0 with the PD target of one action (what one engine step does); r32 rounds the state to float32
    between substeps.  Returns the end state and how close the deepest-K picks came to their discontinuity (oracle.selection_margin)."""
    orc.reset_spec()
    orc.set_spec(**{**BASE_SPEC, **spec})
    s = s0.copy()
    tgt = np.clip(s[13:25] + np.asarray(act, np.float64), -3.0, 3.0)
    sel = np.inf
    for k in range(10):
        tau = np.clip(50.0 * (tgt - s[13:25]) - 0.5 * s[25:37], -16.0, 16.0)
        push = push_trace[k, 1:4] if push_trace[k, 0] > 0.5 else None
        B.selection_margin()
        s = B.substep_terrain(s, tau, mu, near, 0.5 / 0.9, push)[0]
        sel = min(sel, B.selection_margin())
        if r32:
            s = s.astype(np.float32).astype(np.float64)
    orc.reset_spec(); orc.set_spec(**BASE_SPEC)
    return s, sel


TIE_ZONE = 2e-6        # float32 resolution of a contact depth a few metres from the origin


def score_case(out, B, orc, es_i, s0, act, push_trace, mu, near):
    """Engine end state es_i against the oracle's for one case; appends to out['config' / 'vel' / 'cond_*' / 'on_tie']."""
    from parity_common import quat_align

    def errs(a, b):
        e = np.abs(quat_align(a, b) - b)
        return max(e[0:7].max(), e[13:25].max()), max(e[7:13].max(), e[25:37].max()) / (1.0 + np.abs(b[25:37]).max())
    s, sel = oracle_control_step(B, orc, s0, act, push_trace, mu, near)
    c, v = errs(es_i, s)
    on_tie = sel < TIE_ZONE
    c_nominal, v_nominal = c, v
    if on_tie:
        # Some candidate's depth came within float32 resolution of (deepest + LLM_SELECT_EPS), where the deepest-K pick changes hands: float32
        # and float64 may legitimately keep different points.  The engine must then agree with the oracle for SOME tie tolerance within
        # +-2 TIE_ZONE of the nominal one.
        from lifelike_agility_and_play_amd import capi
        for d in (-2.0 * TIE_ZONE, 2.0 * TIE_ZONE):
            s2, _ = oracle_control_step(B, orc, s0, act, push_trace, mu, near, select_eps=capi.LL_SELECT_EPS + d)
            c2, v2 = errs(es_i, s2)
            if c2 < c:
                c, v, s = c2, v2, s2
    s_r32, _ = oracle_control_step(B, orc, s0, act, push_trace, mu, near, r32=True)
    cc, cv = errs(s_r32, oracle_control_step(B, orc, s0, act, push_trace, mu, near)[0])
    out['config'].append(c); out['vel'].append(v)
    out.setdefault('cond_config', []).append(cc); out.setdefault('cond_vel', []).append(cv)
    out.setdefault('on_tie', []).append(on_tie)
    # ... and whether the allowance was NEEDED: the nominal comparison outside the plain bars, a shifted tolerance inside them (round 5: this is what the cap counts;
    # that some depth of ten substeps came within 4 um of the rule's boundary happens to a few per cent of all cases and decides nothing in most of them)
    out.setdefault('tie_used', []).append(bool(on_tie and (c_nominal >= 1e-4 or v_nominal >= 1e-3) and (c < c_nominal)))
    return s


def assert_within_bars(out, cfg_bar=1e-4, vel_bar=1e-3, factor=4.0, max_ill=0.03, max_tie=0.04, cap_ill=None, cap_tie=None):
    """Every case within the bars of flat-ground motion (1e-4 configuration, 1e-3 relative velocity) -- unless the case is ill-conditioned in
    the ORACLE itself: a stick-slip or make-and-break contact step in which merely rounding the oracle's state to float32 between substeps
    moves its own result by more than a quarter of the bar.  Such a case (at most `max_ill` of the cases) must stay within `factor` times
    that self-deviation.  Exact and near ties in depth (symmetric poses) are chosen by index in both implementations (LLM_SELECT_EPS); a case
    that lands on the rule's remaining discontinuity is compared as score_case describes."""
    c, v = np.array(out['config']), np.array(out['vel'])
    cc, cv = np.array(out['cond_config']), np.array(out['cond_vel'])
    bar_c, bar_v = np.maximum(cfg_bar, factor * cc), np.maximum(vel_bar, factor * cv)
    ill = (bar_c > cfg_bar) | (bar_v > vel_bar)
    out['n_ill_conditioned'], out['n_on_selection_tie'] = int(ill.sum()), int(np.sum(out.get('tie_used', out['on_tie'])))
    out['n_near_selection_boundary'] = int(np.sum(out['on_tie']))
    # the two allowances are counted, printed and capped, so that they cannot quietly absorb a regression: ill-conditioned cases at most
    # max_ill of the cases (observed: 1 of 32 standing / 2 of 32 dropped), cases on the deepest-K rule's discontinuity at most max_tie (observed: 0 / 3 of 32)
    # caps: what the case set was observed to need + 1 where the caller knows it (the GPU-sized sets, round 4), a fraction of the cases otherwise
    cap_ill = cap_ill if cap_ill is not None else max(2, int(max_ill * len(ill)))
    cap_tie = cap_tie if cap_tie is not None else max(2, int(max_tie * len(ill)))
    print('assert_within_bars: %d cases, %d ill-conditioned in the oracle itself (cap %d), %d decided by a selection tie (cap %d; %d came within 4 um of the rule\'s boundary); worst config %.2e, velocity %.2e'
          % (len(ill), ill.sum(), cap_ill, out['n_on_selection_tie'], cap_tie, out['n_near_selection_boundary'], c.max(), v.max()))
    assert out['n_on_selection_tie'] <= cap_tie, out['n_on_selection_tie']
    assert ill.sum() <= cap_ill, (ill.sum(), len(ill))
    assert (c < bar_c).all(), (np.sort(c)[-5:], cc[np.argsort(c)[-5:]])
    assert (v < bar_v).all(), (np.sort(v)[-5:], cv[np.argsort(v)[-5:]])


def check_terrain_physics_against_oracle(lib_path, n_envs=16, seed=11, total_envs=None, max_tie=0.04, cap_ill=None, cap_tie=None):
    """Robots standing on, straddling and pressed into cube steps and hurdles, with the push active: one control step of real
    physics, engine (float32) vs the float64 oracle given the same terrain records, friction and push forces.  The two share the
    spec (shape_sdf, nearest-surface normal, btPlaneSpace1 tangents) and nothing else.
    total_envs: the engine runs that many envs (above 4096: the larger-batch kernel build) and the n_envs cases are spread over the first,
    middle and last wavefronts of its grid."""
    from conftest import make_oracle_batch
    from oracle import oracle as orc
    from lifelike_agility_and_play_amd import mocap
    from scipy.spatial.transform import Rotation as Rot
    blob_ = urdf_model.default_model_blob()
    qlo, qhi = blob_[241:253], blob_[253:265]                              # LLM_OFF_Q_LO / _HI
    out = dict(config=[], vel=[], n_terrain=0)
    N = total_envs or n_envs
    third = n_envs // 3
    idx = np.arange(n_envs) if not total_envs else np.concatenate([np.arange(third), N // 2 - 5 + np.arange(third), N - (n_envs - 2 * third) + np.arange(n_envs - 2 * third)])
    for element in (3, 1):
        cfg = env_config(element)
        cfg['env_randomize_config']['disturb_force_config'] = {'start_time': 0.0, 'interval_time': 1.0, 'duration_time': 0.5, 'horizontal_force': [10, 50], 'vertical_force': [0, 10]}
        E = make_engine(cfg, N, lib_path, seed=seed)
        E.reset()
        rows, cnt = E.statics()
        rng = np.random.default_rng(seed)
        st = E.state().astype(np.float64)
        recs_all = [statics_to_records(rows[e, :cnt[e]].astype(np.float64)) for e in idx]
        for k, i in enumerate(idx):
            rec = recs_all[k]
            b = rec[2 + rng.integers(0, min(6, len(rec) - 2))]                      # one of the first obstacles (0, 1 are the walls)
            st[i, 0] = rng.uniform(b[0] - 0.35, b[1] + 0.35); st[i, 1] = rng.uniform(-0.1, 0.1)
            st[i, 2] = b[5] + rng.uniform(0.22, 0.33) if b[4] < 0.01 else rng.uniform(0.2, b[4] + 0.05)   # above a step / bar, or under a hanging bar
            st[i, 7:13] = rng.normal(size=6) * 0.3
            st[i, 25:37] = rng.normal(size=12)
            # (round 5) not the reset pose itself: a level trunk on four identically bent legs puts the left and right vertices of every link box -- and
            # all four feet -- at the same depth up to rounding, and the deepest-4 rule of a leg was decided by the last bit in 6 - 9 % of such cases
            # (the rounds' tie counts were a property of this case set, not of the rule).  A few degrees of roll / pitch / yaw and of every joint:
            st[i, 3:7] = (Rot.from_euler('xyz', [rng.normal() * 0.06, rng.normal() * 0.06, rng.normal() * 0.3]) * Rot.from_quat(st[i, 3:7])).as_quat()
            st[i, 13:25] = np.clip(st[i, 13:25] + rng.normal(size=12) * 0.1, qlo + 0.02, qhi - 0.02)
        E.set_state(st)
        st32 = E.state().astype(np.float64)
        act = (rng.normal(size=(N, 12)) * 0.135).astype(np.float32)
        ep = E.episode()
        E.step_host(act)
        es = E.state().astype(np.float64)
        tr = E.push_trace().astype(np.float64)
        B = make_oracle_batch(orc, urdf_model.default_model_blob(), mocap.load_mocap('', 0.02), n_envs=1, kd=0.5, max_tau=16.0)
        for k, i in enumerate(idx):
            rec = recs_all[k]
            p = st32[i, 0:3]
            near = rec[(p[0] >= rec[:, 0] - 0.9) & (p[0] <= rec[:, 1] + 0.9) & (p[1] >= rec[:, 2] - 0.9) & (p[1] <= rec[:, 3] + 0.9) & (p[2] <= rec[:, 5] + 0.9)][:8]
            mu_i = float(np.float32(ep['friction'][i]) * np.float32(0.9))
            s = score_case(out, B, orc, es[i], st32[i], act[i], tr[i], mu_i, near)
            s_flat, _ = oracle_control_step(B, orc, st32[i], act[i], tr[i], mu_i, near[:0])     # the same step with the terrain ignored
            if np.abs(s - s_flat).max() > 1e-3:
                out['n_felt'] = out.get('n_felt', 0) + 1                              # the obstacle changed the motion
            if len(near) and ((near[:, 1] - near[:, 0]) < 10.0).any():        # an obstacle (not just a side wall) within reach
                out['n_terrain'] += 1
        E.close()
    assert out['n_terrain'] >= 10 and out.get('n_felt', 0) >= 8, (out['n_terrain'], out.get('n_felt', 0))
    assert_within_bars(out, max_tie=max_tie, cap_ill=cap_ill, cap_tie=cap_tie)
    return out


def check_trunk_on_edges_against_oracle(lib_path, n_envs=16, seed=5, cap_ill=None, cap_tie=None):
    """DESIGN 8 "edges under the trunk" and the side point of a sphere on a wall: robots dropped belly-first across the edges of cube steps
    and hurdles (legs folded back so that the trunk gets there first), one control step, engine vs oracle -- and against the oracle with the
    reverse candidates switched off, to show that the cases are what they claim to be."""
    from conftest import make_oracle_batch
    from oracle import oracle as orc
    from lifelike_agility_and_play_amd import mocap
    from parity_common import quat_align
    from scipy.spatial.transform import Rotation as Rot
    out = dict(config=[], vel=[], n_edge_felt=0)
    blob = urdf_model.default_model_blob()
    box = blob[urdf_model.OFF_BASE_PRIMS:urdf_model.OFF_BASE_PRIMS + urdf_model.PRIM_STRIDE]
    hz, cz = box[3], box[6]
    for element in (3, 1, 2):
        cfg = env_config(element)
        E = make_engine(cfg, n_envs, lib_path, seed=seed)
        E.reset()
        rows, cnt = E.statics()
        rng = np.random.default_rng(seed)
        st = E.state().astype(np.float64)
        recs_all = [statics_to_records(rows[i, :cnt[i]].astype(np.float64)) for i in range(n_envs)]
        felt0 = out['n_edge_felt']
        for i in range(n_envs):
            rec = recs_all[i]
            if element == 2:
                # (round 5) under a HANGING bar (BSE:366-412: 0.1 m long, its underside 0.25 m up): the flat of the back within the margin of, or a little
                # into, the underside, the robot on its way up -- the bar's bottom edges against the body box (LLM_FLOATING_MIN_Z)
                cand = [b for b in rec[2:12] if b[4] > 0.05]
                b = cand[rng.integers(0, len(cand))]
                st[i, 0] = 0.5 * (b[0] + b[1]) + rng.uniform(-0.12, 0.12); st[i, 1] = rng.uniform(-0.1, 0.1)
                st[i, 2] = b[4] - hz - cz + rng.uniform(-0.012, 0.015)
            else:
                cand = [b for b in rec[2:10] if b[4] < 0.01 and b[5] > 0.05]              # boxes standing on the ground
                b = cand[rng.integers(0, len(cand))]
                edge = b[0] if rng.uniform() < 0.5 else b[1]
                st[i, 0] = edge + rng.uniform(-0.1, 0.1); st[i, 1] = rng.uniform(-0.1, 0.1)
                st[i, 2] = b[5] + hz - cz + rng.uniform(-0.015, 0.012)                     # the belly within the margin of, or a little into, the top
            st[i, 3:7] = Rot.from_euler('zyx', [rng.uniform(-0.5, 0.5), rng.uniform(-0.15, 0.15), rng.uniform(-0.1, 0.1)]).as_quat()
            st[i, 7:13] = rng.normal(size=6) * 0.3; st[i, 9] += 0.5 if element == 2 else -0.5
            st[i, 13:25] = np.tile([0.0, 1.4, -2.4], 4) + rng.normal(size=12) * 0.05     # legs folded up and back
            st[i, 25:37] = rng.normal(size=12) * 0.5
        E.set_state(st)
        st32 = E.state().astype(np.float64)
        act = (rng.normal(size=(n_envs, 12)) * 0.05).astype(np.float32)
        ep = E.episode()
        E.step_host(act)
        es = E.state().astype(np.float64)
        tr = E.push_trace().astype(np.float64)
        B = make_oracle_batch(orc, blob, mocap.load_mocap('', 0.02), n_envs=1, kd=0.5, max_tau=16.0)
        for i in range(n_envs):
            rec = recs_all[i]
            p = st32[i, 0:3]
            near = rec[(p[0] >= rec[:, 0] - 0.9) & (p[0] <= rec[:, 1] + 0.9) & (p[1] >= rec[:, 2] - 0.9) & (p[1] <= rec[:, 3] + 0.9) & (p[2] <= rec[:, 5] + 0.9)][:8]
            mu_i = float(np.float32(ep['friction'][i]) * np.float32(0.9))
            s = score_case(out, B, orc, es[i], st32[i], act[i], tr[i], mu_i, near)
            s_off, _ = oracle_control_step(B, orc, st32[i], act[i], tr[i], mu_i, near, trunk_edges=0)     # without the reverse candidates
            if np.abs(s - s_off).max() > 1e-3:
                out['n_edge_felt'] += 1
        E.close()
        out.setdefault('felt_by_element', {})[element] = out['n_edge_felt'] - felt0
    assert out['n_edge_felt'] >= 8 and out['felt_by_element'][2] >= max(2, n_envs // 8), (out['n_edge_felt'], out['felt_by_element'])
    assert_within_bars(out, max_ill=0.07, max_tie=0.12, cap_ill=cap_ill, cap_tie=cap_tie)         # (bodies dropped flat onto edges: more make-and-break steps and more equal depths than among standing robots)
    return out


def check_legs_on_edges_against_oracle(lib_path, n_envs=16, seed=7, total_envs=None, cap_ill=None, cap_tie=None):
    """DESIGN 8, round 6 "edges across the leg boxes" (LLM_SPEC_LEG_EDGES = 1 on both sides: the engine runs its XROWS build; BSE:310-364: a shank laid across a hurdle): robots lowered
    onto hurdles with their shanks level, one shank's flat bottom within the margin of -- or a little into -- a hurdle's top edge somewhere between that shank's own candidate points; one
    control step of real physics, engine vs oracle, standing bars -- and against the oracle with the leg edges switched off, to show that the cases are what they claim to be.
    total_envs: as check_terrain_physics_against_oracle (the larger-batch build, cases spread over its grid)."""
    with spec_variant(leg_edges=1):
        return _check_legs_on_edges_against_oracle(lib_path, n_envs, seed, total_envs, cap_ill, cap_tie)


def _check_legs_on_edges_against_oracle(lib_path, n_envs, seed, total_envs, cap_ill, cap_tie):
    from conftest import make_oracle_batch
    from oracle import oracle as orc
    from lifelike_agility_and_play_amd import mocap
    from scipy.spatial.transform import Rotation as Rot
    blob = urdf_model.default_model_blob()
    jo = blob[urdf_model.OFF_JOINT_ORIGIN:urdf_model.OFF_JOINT_ORIGIN + 36].reshape(12, 3)
    ax = blob[urdf_model.OFF_JOINT_AXIS:urdf_model.OFF_JOINT_AXIS + 36].reshape(12, 3)

    def shank_box(s, l):                                                    # world centre of leg l's shank box, its lowest half extent, by an independent chain product
        pos, rot = s[0:3].copy(), Rot.from_quat(s[3:7])
        for j in range(3):
            pos = pos + rot.apply(jo[3 * l + j]); rot = rot * Rot.from_rotvec(ax[3 * l + j] * s[13 + 3 * l + j])
        bx = blob[urdf_model.OFF_LEG_PRIMS + (l * urdf_model.N_LEG_PRIMS + 5) * urdf_model.PRIM_STRIDE:][:16]
        axes = (rot * Rot.from_matrix(bx[7:16].reshape(3, 3))).as_matrix()
        return pos + rot.apply(bx[4:7]), float(sum(abs(axes[2, a]) * bx[1 + a] for a in range(3)))
    out = dict(config=[], vel=[], n_edge_felt=0)
    N = total_envs or n_envs
    third = n_envs // 3
    idx = np.arange(n_envs) if not total_envs else np.concatenate([np.arange(third), N // 2 - 5 + np.arange(third), N - (n_envs - 2 * third) + np.arange(n_envs - 2 * third)])
    cfg = env_config(1)
    E = make_engine(cfg, N, lib_path, seed=seed)
    E.reset()
    rows, cnt = E.statics()
    rng = np.random.default_rng(seed)
    st = E.state().astype(np.float64)
    recs_all = [statics_to_records(rows[e, :cnt[e]].astype(np.float64)) for e in idx]
    for k, i in enumerate(idx):
        rec = recs_all[k]
        cand = [b for b in rec[2:12] if b[4] < 0.01 and 0.04 < b[5] < 0.2]                # hurdles standing on the ground
        b = cand[rng.integers(0, len(cand))]
        l = int(rng.integers(0, 4))
        st[i, 0:3] = [0.0, rng.uniform(-0.1, 0.1), 1.0]
        st[i, 3:7] = Rot.from_euler('zyx', [rng.uniform(-0.4, 0.4), rng.uniform(-0.08, 0.08), rng.uniform(-0.08, 0.08)]).as_quat()
        st[i, 13:25] = np.tile([0.0, -0.8, 0.8 + np.pi / 2], 4) + rng.normal(size=12) * 0.04     # thigh + shank = pi / 2: the shanks lie level
        c, low = shank_box(st[i], l)
        edge = b[0] if rng.uniform() < 0.5 else b[1]
        st[i, 0] += edge - c[0] + rng.uniform(-0.03, 0.03)                                # the hurdle's edge under the middle third of that shank
        st[i, 2] += b[5] - (c[2] - low) + rng.uniform(-0.012, 0.012)                       # its bottom within the margin of, or a little into, the hurdle's top
        st[i, 7:13] = rng.normal(size=6) * 0.2; st[i, 9] -= 0.4
        st[i, 25:37] = rng.normal(size=12) * 0.5
    E.set_state(st)
    st32 = E.state().astype(np.float64)
    act = (rng.normal(size=(N, 12)) * 0.05).astype(np.float32)
    ep = E.episode()
    E.step_host(act)
    es = E.state().astype(np.float64)
    tr = E.push_trace().astype(np.float64)
    B = make_oracle_batch(orc, blob, mocap.load_mocap('', 0.02), n_envs=1, kd=0.5, max_tau=16.0)
    for k, i in enumerate(idx):
        rec = recs_all[k]
        p = st32[i, 0:3]
        near = rec[(p[0] >= rec[:, 0] - 0.9) & (p[0] <= rec[:, 1] + 0.9) & (p[1] >= rec[:, 2] - 0.9) & (p[1] <= rec[:, 3] + 0.9) & (p[2] <= rec[:, 5] + 0.9)][:8]
        mu_i = float(np.float32(ep['friction'][i]) * np.float32(0.9))
        s = score_case(out, B, orc, es[i], st32[i], act[i], tr[i], mu_i, near)
        s_off, _ = oracle_control_step(B, orc, st32[i], act[i], tr[i], mu_i, near, leg_edges=0)        # rounds 1 - 5: without the leg edges
        if np.abs(s - s_off).max() > 1e-3:
            out['n_edge_felt'] += 1
    E.close()
    assert out['n_edge_felt'] >= n_envs // 3, out['n_edge_felt']
    assert_within_bars(out, max_ill=0.07, max_tie=0.12, cap_ill=cap_ill, cap_tie=cap_tie)
    return out


def check_free_running_against_oracle_env(lib_path, n_steps=4, elements=(1, 3, 0), aux=0.02, obs_rand=None):
    """End to end, nothing scripted: the engine and the oracles assembled into a CPU env (oracle/free_run.py) start from the same uniforms,
    get the same actions and are compared after every control step -- terrain, rays on the real boxes, ten substeps of terrain physics with the
    push, rewards, termination.  Contact dynamics amplify float32 rounding, so the bars widen with the step index."""
    from oracle import free_run as FR
    from lifelike_agility_and_play_amd import mocap
    blob, table, init = urdf_model.default_model_blob(), mocap.load_mocap('', 0.02), epmc_capi.default_init_state()
    worst = dict(state=0.0, percep_same=1.0, reward=0.0)
    for element in elements:
        cfg = env_config(element, aux=aux, obs_rand=obs_rand, cmd_range=(3, 5))
        cfg['env_randomize_config']['disturb_force_config'] = {'start_time': 0.0, 'interval_time': 1.0, 'duration_time': 0.5, 'horizontal_force': [10, 50], 'vertical_force': [0, 10]}
        n = 3
        E = make_engine(cfg, n, lib_path, seed=1)
        runs = [FR.EpmcFreeRun(cfg, blob, table, init, seed=0) for _ in range(n)]
        U = np.full((n, epmc_capi.LLE_MAX_DRAWS), 0.5, np.float32)
        obs_o = []
        for i, r in enumerate(runs):
            r.draws = FR.SharedDraws(100 * element + i)
            obs_o.append(r.reset())
            u = r.draws.take()
            U[i, :len(u)] = u
        E.reset(draws=U)
        rng = np.random.default_rng(element)

        def compare(t, obs_o, rew_o=None, done_o=None):
            obs_e = E.obs().astype(np.float64)
            st_e = E.state().astype(np.float64)
            from parity_common import quat_align
            for i, r in enumerate(runs):
                tol = 1e-5 * 3.0 ** t                                    # 1e-5 at the reset, 8e-4 after four steps (measured: < 1 % of it)
                err = np.abs(quat_align(st_e[i], r.env.state) - r.env.state)
                worst['state'] = max(worst['state'], err[:7].max() / tol)
                assert err[:7].max() < tol and err[13:25].max() < 5 * tol, (element, i, t, err[:7].max(), err[13:25].max())
                pe, po = obs_e[i][135:913], np.asarray(obs_o[i])[135:913]
                same = np.abs(pe - po) < 2e-3 + 20 * tol
                worst['percep_same'] = min(worst['percep_same'], same.mean())
                assert same.mean() > 0.97, (element, i, t, same.mean())      # a ray grazing a box edge may fall either side
                np.testing.assert_allclose(obs_e[i][913:916], np.asarray(obs_o[i])[913:916], atol=2e-3 + 20 * tol)
            if rew_o is not None:
                rew_e, done_e, _ = E.reward_done()
                for i in range(n):
                    assert bool(done_e[i]) == bool(done_o[i])
                    worst['reward'] = max(worst['reward'], abs(rew_e[i] - rew_o[i]))
                    assert abs(rew_e[i] - rew_o[i]) < 1e-5 + 0.05 * 1e-5 * 3.0 ** t
        compare(0, obs_o)
        for t in range(n_steps):
            act = (rng.normal(size=(n, 12)) * 0.135).astype(np.float32)
            outs = [r.step(act[i].astype(np.float64)) for i, r in enumerate(runs)]
            used = [r.draws.take() for r in runs]
            k = max(1, max(len(u) for u in used))
            D = np.full((n, k), 0.5, np.float32)
            for i, u in enumerate(used):
                D[i, :len(u)] = u
            E.set_step_draws(D)
            E.step_host(act)
            compare(t + 1, [o[0] for o in outs], [o[1] for o in outs], [o[2] for o in outs])
            if any(o[2] for o in outs):
                break
        E.close()
    return worst


# ------------------------------------------------------------------------------------------------------------------------------------------
# Game-level statistics, engine against oracle env, ONE protocol (round-3 review, "What's weak" #3)
# ------------------------------------------------------------------------------------------------------------------------------------------
GAME_HORIZON = {'hurdle': 420, 'cube': 600, 'hole': 500}


def _game_cfg(which):
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import rollout_epmc_policy as R
    cfg = R.env_config(R.ELEMENT[which], 1)
    cfg['max_steps'] = GAME_HORIZON[which]                 # every episode is played to ITS end: target reached, fall, or this many steps
    return cfg


def _oracle_game(job):
    """One episode of the float64 oracle env (oracle/free_run.py) under the reference's trained policy, drawing from a recorded stream:
    returns (length, end reason as the engine's bits, the uniforms of the reset, the uniforms of every step)."""
    which, seed = job
    from oracle import free_run as FR
    from oracle.epmc_policy import EpmcPolicy
    from lifelike_agility_and_play_amd import mocap
    cfg = _game_cfg(which)
    run = FR.EpmcFreeRun(cfg, urdf_model.default_model_blob(), mocap.load_mocap('', 0.02), epmc_capi.default_init_state(), seed=0)
    run.draws = FR.SharedDraws(seed)
    pol = EpmcPolicy(os.path.join(ROOT, 'tests', 'golden', 'epmc_policy_%s.npz' % which), 1)
    obs = run.reset()
    u0, us, why = run.draws.take(), [], 0
    for t in range(cfg['max_steps'] + 1):
        a = pol.act(np.asarray(obs, np.float64).reshape(1, -1))[0]
        obs, r, done, info = run.step(a)
        us.append(run.draws.take())
        if done:
            s = run.env.state
            why = 1 if eo.check_fall(s[3:7]) else (4 if np.linalg.norm((run.env.target_pos - s[0:3])[:2]) < 0.5 else 2)
            break
    assert why, 'the oracle episode did not end by max_steps'
    return len(us), why, u0, us


def game_cache_extra(which):
    """What besides the source files decides an oracle episode: the game config as the tests build it"""
    import json
    return 'epmc %s ' % which + json.dumps(_game_cfg(which), sort_keys=True, default=str)


def oracle_games(which, seeds, procs=None):
    """The oracle env's episodes of `seeds` under policy `which` as (length, why, 0, reset uniforms, step uniforms): from tests/golden/oracle_games.npz while that fixture
    is the CURRENT oracle's (tests/oracle_game_cache.py; LL_LIVE_ORACLE_GAMES=1 ignores it), otherwise played now on `procs` host processes."""
    import gc
    import multiprocessing as mp
    import bench
    import oracle_game_cache as C
    res = C.load('epmc_' + which, game_cache_extra(which), seeds)
    if res is not None:
        return res, 'the oracle episodes come from tests/golden/oracle_games.npz, whose source key is current'
    procs = procs or bench.effective_cores()[0]
    gc.collect()                                              # (no dead engine objects for the forked workers to finalise)
    with mp.get_context('fork').Pool(procs) as p:
        res = p.map(_oracle_game, [(which, s) for s in seeds], chunksize=1)
    return [(r[0], r[1], 0, r[2], r[3]) for r in res], 'played live on %d processes' % procs


def check_game_statistics(lib_path, n_per_policy=256, procs=None, policies=('hurdle', 'cube', 'hole'), frac_tol=0.03, len_tol=0.03, ks_p=0.5, n_se=2.0):
    """PMC has check_rollout_statistics; this is the same for the environmental level, at the level the game is decided on.  The engine and the
    float64 oracle env play the SAME episodes -- same terrain, friction, pushes (the oracle env's uniforms are recorded and handed to the engine
    draw by draw), same trained policy of the reference acting on each side's own observations -- every episode to its end on both sides
    (target reached / fell / max_steps).  Chaos decorrelates individual episodes; the DISTRIBUTIONS must agree: end-reason fractions within
    `frac_tol`, mean length within `len_tol`, Kolmogorov-Smirnov p > 0.5 on the lengths -- for the hurdle and stairs policies, whose episodes stay
    correlated between the two simulators (measured on MI355X, 128 episodes each: reached 0.984 / 0.984 and 0.992 / 0.969, mean length 188.6 / 188.0
    and 231.1 / 233.7, KS p 1.0).  The overhead-bars ('hole') policy, whose episodes split between reaching and falling (0.266 / 0.258 reached,
    0.578 / 0.617 fell), gets two-sample bars (see below)."""
    import gc
    import multiprocessing as mp
    from scipy import stats as sst
    from oracle.epmc_policy import EpmcPolicy
    import bench
    out = {}
    for which in policies:
        n = n_per_policy
        res, src = oracle_games(which, [1000 * (1 + policies.index(which)) + i for i in range(n)], procs)
        print('%s policy: %s' % (which, src))
        len_o, why_o = np.array([r[0] for r in res]), np.array([r[1] for r in res])
        print('%s policy: %d episodes, oracle seeds %d .. %d, engine seed 3, bars at %.1f standard errors (floors %.3f / %.3f)' % (which, n, 1000 * (1 + policies.index(which)), 1000 * (1 + policies.index(which)) + n - 1, n_se, frac_tol, len_tol))
        cfg = _game_cfg(which)
        E = make_engine(cfg, n, lib_path, seed=3)
        U = np.full((n, epmc_capi.LLE_MAX_DRAWS), 0.5, np.float32)
        for i, r in enumerate(res):
            U[i, :len(r[3])] = r[3]
        E.reset(draws=U)
        pol = EpmcPolicy(os.path.join(ROOT, 'tests', 'golden', 'epmc_policy_%s.npz' % which), n)
        obs = E.obs()
        alive = np.ones(n, bool); len_e = np.zeros(n, int); why_e = np.zeros(n, int)
        for t in range(cfg['max_steps'] + 1):
            used = [r[4][t] if t < len(r[4]) else [] for r in res]
            D = np.full((n, max(1, max(len(u) for u in used))), 0.5, np.float32)      # (beyond the oracle episode's end: mid-range pushes)
            for i, u in enumerate(used):
                D[i, :len(u)] = u
            a = pol.act(obs.astype(np.float64))
            E.set_step_draws(D)
            E.step_host(a.astype(np.float32))
            obs = E.obs()
            r_, d_, w_ = E.reward_done()
            len_e += alive
            newly = alive & d_
            why_e[newly] = w_[newly]
            alive &= ~d_
            if not alive.any():
                break
        E.close()
        assert not alive.any(), (which, int(alive.sum()))
        fr = lambda w, bit: float(((w & bit) != 0).mean())
        o = dict(n=n, reached=(fr(why_e, 4), fr(why_o, 4)), fell=(fr(why_e, 1), fr(why_o, 1)), timed_out=(fr(why_e & ~5, 2), fr(why_o & ~5, 2)),
                 mean_len=(float(len_e.mean()), float(len_o.mean())), ks_p=float(sst.ks_2samp(len_e, len_o).pvalue),
                 same_end=float(((why_e & 7) == (why_o & 7)).mean()), same_step=float((len_e == len_o).mean()))
        out[which] = o
        print('game statistics, %s policy, %d episodes (engine / oracle): reached %.3f / %.3f, fell %.3f / %.3f, timed out %.3f / %.3f, mean length %.1f / %.1f, '
              'KS p %.3f; same end reason %.3f, same end step %.3f' % ((which, n) + o['reached'] + o['fell'] + o['timed_out'] + o['mean_len'] + (o['ks_p'], o['same_end'], o['same_step'])))
        # The episodes of the two simulators decorrelate: a contact that makes or breaks one step apart sends the rest of a run down another path
        # (same end step: 0.2 - 0.3 of the runs for every policy; under cone-coupled friction the stairs policy ends 6 % of its runs for another
        # reason).  Engine and oracle are two SAMPLES of one distribution: two-sample bars at `n_se` standard errors (round 6: two, on 256 episodes a policy), never tighter than the
        # floors (parity_common.two_sample_bars).  What the populations are was measured at 1024 episodes a side (profiles/r04_cone_decision.md:
        # stairs reached 988 engine / 981 oracle, hurdles 1013 / 1014).
        from parity_common import two_sample_bars
        two_sample_bars(which, {k: o[k] for k in ('reached', 'fell', 'timed_out')}, len_e, len_o, o['ks_p'], n, floor_frac=frac_tol, floor_len=len_tol, n_se=n_se)
    return out
