#!/usr/bin/env python3
"""Generates tests/golden/oracle_games.npz: the float64 ORACLE env's side of the game-statistics tests (tests/oracle_game_cache.py says why and how it is
kept honest).  Needs neither the reference nor a GPU: it plays the very games tests/sepmc_parity_common._oracle_game / tests/epmc_parity_common._oracle_game
play in the live path -- 512 chase-tag games (seeds 5000 ..) and 256 playground episodes for each of the three trained policies (seeds 1000 / 2000 / 3000 ..),
the numbers tests/test_gpu_sepmc.py and tests/test_gpu_epmc.py ask for -- and stores per game: length, end reason, contact-record count, the number of
uniforms the reset and every step drew.  About 7 minutes on 8 cores.

    python tests/golden/gen_oracle_games.py            # regenerate after ANY change to oracle/, the package's .py / assets, the policies or the game configs
"""
import gc
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

N_SEPMC, SEPMC_SEED0 = 512, 5000            # tests/test_gpu_sepmc.py::test_game_statistics_against_the_oracle_env_gpu
N_EPMC = 256                                # tests/test_gpu_epmc.py::test_game_statistics_against_the_oracle_env, per policy
EPMC_POLICIES = ('hurdle', 'cube', 'hole')  # seeds 1000 x (1 + index) + i, as check_game_statistics numbers them


def main():
    import bench
    import oracle_game_cache as C
    import epmc_parity_common as EC
    import sepmc_parity_common as SC
    from oracle import oracle as O
    O.lib()                                                   # (build the C library once, ahead of the fork)
    procs = bench.effective_cores()[0]
    out = {}
    gc.collect()
    with mp.get_context('fork').Pool(procs) as p:
        t = time.time()
        seeds = [SEPMC_SEED0 + i for i in range(N_SEPMC)]
        res = p.map(SC._oracle_game, seeds, chunksize=1)
        out.update(C.pack('sepmc', SC.game_cache_extra(), seeds, res))
        print('sepmc: %d games in %.0f s, mean length %.1f' % (len(res), time.time() - t, np.mean([r[0] for r in res])), flush=True)
        for k, which in enumerate(EPMC_POLICIES):
            t = time.time()
            seeds = [1000 * (1 + k) + i for i in range(N_EPMC)]
            res = [(r[0], r[1], 0, r[2], r[3]) for r in p.map(EC._oracle_game, [(which, s) for s in seeds], chunksize=1)]
            out.update(C.pack('epmc_' + which, EC.game_cache_extra(which), seeds, res))
            print('epmc %s: %d episodes in %.0f s, mean length %.1f' % (which, len(res), time.time() - t, np.mean([r[0] for r in res])), flush=True)
    np.savez_compressed(C.PATH, **out)
    print('wrote %s (%d bytes)' % (C.PATH, os.path.getsize(C.PATH)))


if __name__ == '__main__':
    main()
