"""The ORACLE's side of the game-statistics tests, kept as a fixture (test infrastructure; nothing of the product reads this).

tests/epmc_parity_common.check_game_statistics and tests/sepmc_parity_common.check_game_statistics let the HIP engine and the float64 oracle env play
the same seeded games and compare the distributions.  The oracle's games are a pure function of (oracle sources, trained weights, game config, seed):
float64, -ffp-contract=off, no fast-math (oracle/Makefile), every uniform drawn from numpy's default_rng(seed) through oracle/free_run.SharedDraws.  They
cost 15 ms of host time per step -- 216 s of the GPU suite's 556 s for the 512 chase-tag games, 56 s for the 3 x 256 playground episodes -- and they
do not depend on the GPU at all.  So the summary of each oracle game (length, end reason, contact-record count, how many uniforms the reset and every
step consumed) is generated once by tests/golden/gen_oracle_games.py (it calls the very `_oracle_game` the live path calls) and committed as
tests/golden/oracle_games.npz (a few hundred KB: the uniforms themselves are re-drawn from the seed, only the COUNTS are stored).

The fixture is used only while it is provably the current oracle's: it carries a sha256 over every file the games depend on (`source_key`), and
`load` returns None when the key differs, when the file is missing, or when LL_LIVE_ORACLE_GAMES=1 -- the callers then run the games live as before.
tests/test_oracle_game_cache.py (CPU suite) fails when the committed fixture is stale and replays one game of each kind live against it."""
import glob
import hashlib
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, 'tests', 'golden', 'oracle_games.npz')

_KEY_GLOBS = ('oracle/*.py', 'oracle/*.c', 'oracle/Makefile', 'include/*.h', 'lifelike_agility_and_play_amd/*.py', 'lifelike_agility_and_play_amd/assets/*',
              'tests/golden/epmc_policy_*.npz', 'tests/golden/sepmc_policy.npz', 'tools/rollout_epmc_policy.py', 'tools/rollout_sepmc_policy.py', 'tools/env_configs.py')


def source_key(extra=''):
    """sha256 over the names and bytes of every file an oracle game depends on, and over `extra` (the game configs and seeds as the callers build them)"""
    h = hashlib.sha256()
    files = sorted({f for g in _KEY_GLOBS for f in glob.glob(os.path.join(ROOT, g)) if os.path.isfile(f)})
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode() + b'\0')
        with open(f, 'rb') as fh:
            h.update(hashlib.sha256(fh.read()).digest())
    h.update(extra.encode())
    return h.hexdigest()


def redraw(seed, n_reset, counts):
    """The uniforms SharedDraws(seed) handed out: the first n_reset at reset, then counts[t] during step t (the same float32-rounded values, in order)"""
    g = np.random.default_rng(int(seed))
    total = int(n_reset) + int(np.sum(counts))
    u = [float(np.float32(g.random() * 0.999)) for _ in range(total)]
    u0, us, k = u[:int(n_reset)], [], int(n_reset)
    for c in counts:
        us.append(u[k:k + int(c)])
        k += int(c)
    return u0, us


def pack(kind, extra, seeds, results):
    """results[i] = (length, why, named, u0, us) of seed seeds[i] (EPMC: named = 0) -> the arrays of one kind, names prefixed by it"""
    counts = [np.array([len(u) for u in r[4]], np.uint16) for r in results]
    for s, r, c in zip(seeds, results, counts):                       # the stored counts must reproduce the recorded uniforms exactly
        u0, us = redraw(s, len(r[3]), c)
        assert u0 == list(r[3]) and us == [list(u) for u in r[4]], (kind, s)
    return {kind + '_key': np.array(source_key(extra)), kind + '_seed': np.asarray(seeds, np.int64),
            kind + '_len': np.array([r[0] for r in results], np.int32), kind + '_why': np.array([r[1] for r in results], np.int32),
            kind + '_named': np.array([r[2] for r in results], np.int32), kind + '_n_reset': np.array([len(r[3]) for r in results], np.int32),
            kind + '_off': np.concatenate([[0], np.cumsum([len(c) for c in counts])]).astype(np.int64),
            kind + '_counts': np.concatenate(counts) if counts else np.zeros(0, np.uint16)}


def load(kind, extra, seeds, path=None):
    """[(length, why, named, u0, us)] for `seeds`, or None when there is no CURRENT fixture for them (the caller plays the games live)"""
    if os.environ.get('LL_LIVE_ORACLE_GAMES') == '1':
        return None
    path = path or PATH
    if not os.path.exists(path):
        return None
    z = np.load(path)
    if kind + '_key' not in z.files or str(z[kind + '_key']) != source_key(extra):
        return None
    have = {int(s): i for i, s in enumerate(z[kind + '_seed'])}
    if any(int(s) not in have for s in seeds):
        return None
    off, cnt = z[kind + '_off'], z[kind + '_counts']
    out = []
    for s in seeds:
        i = have[int(s)]
        c = cnt[off[i]:off[i + 1]]
        assert len(c) == int(z[kind + '_len'][i])
        u0, us = redraw(s, z[kind + '_n_reset'][i], c)
        out.append((int(z[kind + '_len'][i]), int(z[kind + '_why'][i]), int(z[kind + '_named'][i]), u0, us))
    return out
