"""tests/golden/oracle_games.npz (the oracle's side of the game-statistics tests, tests/oracle_game_cache.py) must be the CURRENT oracle's: its source key
equals the key of the tree, one game of each kind played live now equals its record, and a stale or disabled fixture sends the callers to the live path."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import oracle_game_cache as C   # noqa: E402
import epmc_parity_common as EC   # noqa: E402
import sepmc_parity_common as SC   # noqa: E402

REGEN = 'tests/golden/oracle_games.npz is stale: a file the oracle games depend on changed -- run `python tests/golden/gen_oracle_games.py` (7 min) and commit the result'


def kinds():
    return [('sepmc', SC.game_cache_extra())] + [('epmc_' + w, EC.game_cache_extra(w)) for w in ('hurdle', 'cube', 'hole')]


def test_the_fixture_is_the_current_oracles():
    z = np.load(C.PATH)
    for kind, extra in kinds():
        assert str(z[kind + '_key']) == C.source_key(extra), '%s: %s' % (kind, REGEN)
    # ... and it holds what the GPU tests ask for: 512 chase-tag games from seed 5000, 256 episodes a policy from 1000 / 2000 / 3000
    assert list(z['sepmc_seed']) == list(range(5000, 5512))
    for k, w in enumerate(('hurdle', 'cube', 'hole')):
        assert list(z['epmc_%s_seed' % w]) == list(range(1000 * (1 + k), 1000 * (1 + k) + 256))
    assert SC.oracle_games([5000, 5001])[1].startswith('the oracle games come from')
    assert EC.oracle_games('hole', [3000, 3255])[1].startswith('the oracle episodes come from')


def test_one_live_game_of_each_kind_equals_its_record():
    """A recorded game of median length of each kind, played now by the live path's own worker: same length, end reason, contact-record count and the same uniforms
    at the same steps (float64 without contraction or fast-math: the same on every host)."""
    z = np.load(C.PATH)
    i = int(np.argsort(z['sepmc_len'])[len(z['sepmc_len']) // 2])
    seed = int(z['sepmc_seed'][i])
    rec = C.load('sepmc', SC.game_cache_extra(), [seed])
    assert rec is not None, REGEN
    live = SC._oracle_game(seed)
    assert (live[0], live[1], live[2]) == rec[0][:3], (seed, live[:3], rec[0][:3])
    assert list(live[3]) == rec[0][3] and [list(u) for u in live[4]] == rec[0][4]
    for w in ('hurdle', 'cube', 'hole'):
        i = int(np.argsort(z['epmc_%s_len' % w])[len(z['epmc_%s_len' % w]) // 2])
        seed = int(z['epmc_%s_seed' % w][i])
        rec = C.load('epmc_' + w, EC.game_cache_extra(w), [seed])
        assert rec is not None, REGEN
        live = EC._oracle_game((w, seed))
        assert (live[0], live[1]) == rec[0][:2], (w, seed, live[:2], rec[0][:2])
        assert list(live[2]) == rec[0][3] and [list(u) for u in live[3]] == rec[0][4]


def test_a_stale_or_disabled_fixture_is_not_used(monkeypatch):
    assert C.load('sepmc', SC.game_cache_extra() + ' (another config)', [5000]) is None          # another key
    assert C.load('sepmc', SC.game_cache_extra(), [4999]) is None                                 # a seed the fixture does not hold
    assert C.load('sepmc', SC.game_cache_extra(), [5000], path=os.path.join(ROOT, 'tests', 'golden', 'no_such_file.npz')) is None
    monkeypatch.setenv('LL_LIVE_ORACLE_GAMES', '1')
    assert C.load('sepmc', SC.game_cache_extra(), [5000]) is None
    res, src = SC.oracle_games([5000 + int(np.argmin(np.load(C.PATH)['sepmc_len']))], procs=1)    # the caller then plays the game itself
    assert src.startswith('played live') and len(res) == 1 and res[0][0] > 0


def test_the_key_sees_a_changed_source(tmp_path, monkeypatch):
    """source_key is a function of the bytes of the files: a tree whose oracle differs by one byte has another key"""
    import shutil
    root2 = tmp_path / 'tree'
    for g in ('oracle', 'include', 'tools'):
        shutil.copytree(os.path.join(ROOT, g), root2 / g, ignore=shutil.ignore_patterns('_build', '__pycache__', '_ref'))
    k0 = None
    monkeypatch.setattr(C, 'ROOT', str(root2))
    k0 = C.source_key('x')
    with open(root2 / 'oracle' / 'pmc_oracle.c', 'ab') as f:
        f.write(b'\n')
    assert C.source_key('x') != k0
    assert C.source_key('y') != C.source_key('x')


def test_the_row_blocked_policy_of_the_game_statistics_is_the_policy_bit_for_bit():
    """sepmc_parity_common.blocked_policy (the engine side of the 512-game statistic evaluates up to 1024 rows a step) against oracle.sepmc_policy.SepmcPolicy:
    the same actions, heading and LSTM state to the last bit over several steps, with a last block that is not full; and with rows dropping out of `.alive`
    step by step the rows still alive keep the plain policy's numbers (the others are not looked at)"""
    from oracle.sepmc_policy import SepmcPolicy
    npz = os.path.join(ROOT, 'tests', 'golden', 'sepmc_policy.npz')
    n = 150
    rng = np.random.default_rng(4)
    plain, blocked, masked = SepmcPolicy(npz, n), SC.blocked_policy(npz, n, block=64), SC.blocked_policy(npz, n, block=64)
    alive = np.ones(n, bool)
    for step in range(4):
        obs = rng.normal(size=(n, 965)) * 0.5
        a, b = plain.act(obs), blocked.act(obs)
        assert np.array_equal(a, b) and np.array_equal(plain.last_heading, blocked.last_heading)
        for k in plain.h:
            assert np.array_equal(plain.h[k], blocked.h[k]) and np.array_equal(plain.c[k], blocked.c[k])
        alive &= rng.random(n) > 0.3                               # rows leave for good, as finished arenas do
        assert 0 < alive.sum() < n
        masked.alive = alive.copy()
        c = masked.act(obs)
        assert np.array_equal(a[alive], c[alive]) and np.array_equal(plain.last_heading[alive], masked.last_heading[alive])
        for k in plain.h:
            assert np.array_equal(plain.h[k][alive], masked.h[k][alive]) and np.array_equal(plain.c[k][alive], masked.c[k][alive])
        assert np.isfinite(c).all()
