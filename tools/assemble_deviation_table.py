"""profiles/r03_deviation_table.md from the runs of tools/deviation_table.py (PMC), tools/deviation_envs.py (EPMC, SEPMC): engine legs from the GPU box
(gpurun_out/r03d, r03e, r03f), oracle legs run on the build container's 8 cores (gpurun_out/dev)."""
import os
import re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, 'gpurun_out')


def rows(path):
    out = {}
    if not os.path.exists(path):
        return out
    for l in open(path):
        if l.startswith('| ') and not l.startswith('| variant') and not l.startswith('| simulator'):
            c = [x.strip() for x in l.strip().strip('|').split('|')]
            out[c[0]] = c
    return out


eng = rows(os.path.join(G, 'r03d', 'dev_pmc_engine.md'))
eng.update(rows(os.path.join(G, 'r03e', 'dev_pmc_engine_dirs.md')))
orc = {}
for f in ('r03_oracle_table.md', 'r03_oracle_table2.md', 'r03_oracle_table3.md'):
    for k, c in rows(os.path.join(G, 'dev', f)).items():
        if c[4] != '-':
            orc[k] = c
order = []
for f in ('r03d/dev_pmc_engine.md', 'r03e/dev_pmc_engine_dirs.md', 'dev/r03_oracle_table.md', 'dev/r03_oracle_table2.md', 'dev/r03_oracle_table3.md'):
    for k in rows(os.path.join(G, f)):
        if k not in order:
            order.append(k)
L = ['# Deviation study, round 3: the reference\'s trained policies in our simulator, one spec switch moved at a time', '',
     '## PMC: the trained tracking policy (DESIGN.md 4)', '',
     'Protocol: episodes started at uniformly random (clip, t0) over all 62 clips, horizon 500 control steps; engine = float32 HIP kernel on MI355X, 4096 episodes '
     'per variant; oracle = float64 CPU restatement, 1024 episodes per variant.  tools/deviation_table.py.  Noise of the tracked fraction at 1024 episodes: about +-0.003 '
     '(the oracle rows of switches that cannot act -- max coordinate velocity -- reproduce the baseline to the digit).', '',
     '| variant | engine reward | engine tracked | engine length | oracle reward | oracle tracked | oracle length | note |', '|---|---|---|---|---|---|---|---|']
for k in order:
    e, o = eng.get(k), orc.get(k)
    note = (e or o)[7] if len(e or o) > 7 else ''
    if k == 'friction cone, sequential':
        note = 'ORACLE ONLY: the spec\'s rounds, each friction row bounded by what the contact\'s other row leaves of the cone (bounding t2 only: 0.7453 / 0.626 / 248.9): not the cone'
    L.append('| %s | %s | %s | %s |' % (k, ' | '.join(e[1:4]) if e and e[1] != '-' else '- | - | -', ' | '.join(o[4:7]) if o else '- | - | -', note))
L += ['', '## EPMC: the trained hurdle and stairs policies on terrain (DESIGN.md 2, 4)', '']
for f in ('r03d/dev_epmc_engine.md', 'dev/r03_epmc_oracle.md'):
    p = os.path.join(G, f)
    if os.path.exists(p):
        txt = [l.rstrip() for l in open(p) if l.strip()]
        if f.startswith('r03d'):
            L += [l for l in txt if not l.startswith('#')]
        else:
            L += [l for l in txt if l.startswith('| oracle') and 'warm start' not in l]
L += ['', '(engine: 4096 envs per policy on MI355X; oracle: 64 episodes per cell, so one episode is 1.6 %.  The warm-start switch does not reach the stateless terrain entry point the CPU env '
      'steps through and is left out.)', '', '## SEPMC: the trained chase-tag policy on both robots (DESIGN.md 2, 4)', '']
for f in ('r03f/dev_sepmc_engine.md', 'dev/r03_sepmc_oracle2.md'):
    p = os.path.join(G, f)
    if os.path.exists(p):
        txt = [l.rstrip() for l in open(p) if l.strip()]
        if f.startswith('r03f'):
            L += [l for l in txt if not l.startswith('#')]
        else:
            L += [l for l in txt if l.startswith('| oracle')]
L += ['', '(engine: 512 arenas, horizon 1000, every game played to its end; oracle: 64 arenas x 900 arena-steps per variant, games cut off by the budget are not counted, so long games are '
      'under-represented there -- compare oracle rows with each other, not with the engine row.  The last column counts different things: the engine row the arena-steps whose FIRST '
      'contact record names the other robot (what the env reads, CTG:426-440), the oracle rows the arena-steps in which a leg of either robot is within the margin of the other.  '
      'With about 150 games per variant one standard error of the caught fraction is 0.03: no robot-robot contact variant moves a game-level metric by more than that.)']
open(os.path.join(ROOT, 'profiles', 'r03_deviation_table.md'), 'w').write('\n'.join(L) + '\n')
print('\n'.join(L))
