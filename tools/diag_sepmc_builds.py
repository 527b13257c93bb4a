"""Which builds of the one-wave-per-SIMD chase-tag kernels compute a wrong state (the seven-rays-per-chunk failure, HISTORY.md)?  Every library given on the command line
steps the same 2048 arenas (same seed, same actions, all elements) as the shipped library, twice; after every control step the robots whose state differs from the shipped
build's by more than 5e-3 are counted, with where they sit (wave row), how far apart the two robots of their arena are, and whether the set is the same in both runs.

    gpurun -- 'python tools/diag_sepmc_builds.py tools/_build/diag/libllenv_c7.so ... > gpurun_out/sepmc_builds.txt 2>&1'
"""
import hashlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import sepmc_parity_common as SC  # noqa: E402

N, STEPS, SEED = int(os.environ.get("LL_DIAG_N", "2048")), int(os.environ.get("LL_DIAG_STEPS", "3")), 5


def run(lib):
    E = SC.make_engine(SC.env_config(SC.ALL_ELEMENTS), N, lib, auto_reset=1, seed=SEED)
    E.reset()
    rng = np.random.default_rng(SEED)
    out = [E.state().astype(np.float64).copy()]
    obs = [E.obs().astype(np.float64).copy()]
    for t in range(STEPS):
        E.step_host((rng.normal(size=(N, 2, 12)) * 0.135).astype(np.float32))
        out.append(E.state().astype(np.float64).copy())
        obs.append(E.obs().astype(np.float64).copy())
    E.close()
    OBS[lib] = obs
    return out


OBS = {}
P3, NR = 135, 778
FIELDS = (('prop', 0, P3), ('height rays', P3, P3 + 325), ('fan rays', P3 + 325, P3 + 453), ('front rays', P3 + 453, P3 + NR), ('percept_vec', P3 + NR, P3 + NR + 5), ('oppo_info', P3 + NR + 5, P3 + NR + 20),
          ('oppo_info_cheat', P3 + NR + 20, P3 + NR + 35), ('flag_info', P3 + NR + 35, P3 + NR + 42), ('flag_info_cheat', P3 + NR + 42, P3 + NR + 49), ('with_flag', P3 + NR + 49, P3 + NR + 51), ('control_spd', P3 + NR + 51, P3 + NR + 52))


REF = os.environ.get('LL_DIAG_REF') or None          # another library as the reference (default: the shipped one)
ref = run(REF)
ref2 = run(REF)
print('reference %s, run to run: states identical' % (REF or 'shipped library'), all(np.array_equal(a, b) for a, b in zip(ref, ref2)), flush=True)
for lib in sys.argv[1:]:
    if not os.path.exists(lib):
        print(lib, 'missing'); continue
    runs = [run(lib), run(lib)]
    print('%s: run to run identical %s' % (os.path.basename(lib), all(np.array_equal(a, b) for a, b in zip(*runs))))
    for t in range(STEPS + 1):
        d = np.abs(runs[0][t] - ref[t]); d[..., 7:] /= (1.0 + np.abs(ref[t][..., 7:]).max(-1, keepdims=True))
        bad = d.max(-1) > 5e-3                                    # [arena, robot]
        wr = (2 * np.arange(N)[:, None] + np.arange(2)[None, :]) % 4
        dist = np.linalg.norm(ref[t][:, 0, 0:2] - ref[t][:, 1, 0:2], axis=-1)
        ba = bad.any(-1)
        print('    after step %d: robots off %4d (arenas %4d, both robots in %4d), by wave row %s, bitwise-equal robots %d of %d; base distance of those arenas: median %.2f max %.2f (all arenas: median %.2f, within 1.5 m: %d); set %s'
              % (t, bad.sum(), ba.sum(), bad.all(-1).sum(), [int(bad[wr == k].sum()) for k in range(4)], int((d.max(-1) == 0).sum()), 2 * N,
                 np.median(dist[ba]) if ba.any() else 0, dist[ba].max() if ba.any() else 0, np.median(dist), int((dist < 1.5).sum()), hashlib.md5(np.packbits(bad).tobytes()).hexdigest()[:8]), flush=True)
        og, orf = OBS[lib][t], OBS[REF][t]
        print('      observations: robots with an entry off by > 2e-3, per field: %s; bitwise-equal observation rows %d of %d'
              % ({n: int((np.abs(og[..., a:b] - orf[..., a:b]) > 2e-3).any(-1).sum()) for n, a, b in FIELDS}, int((og == orf).all(-1).sum()), 2 * N), flush=True)
        if t == 1 and ba.any():
            a = np.flatnonzero(ba)[:6]
            print('      first arenas off:', a.tolist(), 'largest state entry off:', [int(d[i].max(0).argmax()) for i in a], 'by', ['%.2e' % d[i].max() for i in a])
