"""Issue-slot ledger of the contract kernel (round-5 review, "Next round" #1): the static instructions of pmc_step_kernel<1, false, true, true> attributed to the
phases of a control step, multiplied by trip counts, priced per instruction class with the single-wave issue costs of tools/issue_probe.hip, and reconciled with the
instruction and cycle counters of the committed rocprofv3 --pmc pass.  No GPU needed (hipcc cross-compiles the listing in seconds).

    python tools/issue_ledger.py [--kernel 'pmc_step_kernel<1, false, true, true>(StepParams)'] [--costs profiles/r06_issue_probe.txt] [--counters profiles/r06_pmc_step_kernel_counters.json]
                                 [--define LL_MFMA_GRAM=1 ...] [--md profiles/r06_issue_ledger.md]

How: the source carries PMC_PHASE("name") marks (pmc_params.hpp); compiled with -DPMC_MARKS each becomes a comment in the listing fenced by scheduling barriers, so the
instructions of a phase stay between its marks.  The marked listing is a close cousin of the shipped code (which schedules across those points): its totals are printed
next to the unmarked listing's.  Trip counts: substep phases x n_sub (10), solver rounds x n_sub x n_iter (100); phases under a wave-uniform branch x the share of
wave-substeps that take it (WEIGHTS below: from the stamped ablation build, profiles/r05_timeline.txt, and fitted to the counters where that build has no stamp).
"""
import argparse
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g   # noqa: E402

N_SUB, N_ITER = 10, 10
# share of wave-substeps (or waves, for tail phases) in which the phase's conditional part runs; 1.0 = unconditional
WEIGHTS = {
    'sub.self_rows': 0.42,            # "substeps with a self-collision row somewhere in the wave: 42.0 %" (profiles/r05_timeline.txt)
    'pgs.self_turns': 0.42,
    'tail.episode_end_reseed': 0.073, # "re-seeding wave, 7.3 % of waves"
    'tail.obstacle_check': 0.0,       # set_obstacle is off in BASELINE config 2
    'tail.unroll_row': 0.0,           # P.traj is null at N = 1 (the unroll rows are recorded for the gather: N > 1); --traj 1 counts it
    'kernel.entry': 1.0 / 32,         # once per launch of 32 control steps
}
# single-wave issue cost per instruction class in shader cycles (tools/issue_probe.hip on MI355X: profiles/r06_issue_probe.txt; defaults = profiles/r04_valu_issue.txt)
DEFAULT_COST = {'vop2_add': 4.44, 'vop2_mul': 5.75, 'vop2': 5.38, 'vop3': 5.38, 'dpp': 5.38, 'pk': 5.61, 'trans': 8.16, 'accvgpr': 5.38, 'mov': 4.44, 'lane': 5.63, 'mfma': 30.3, 's_nop': 4.44, 's_nop1': 8.16,
                's_waitcnt': 2.9, 'salu': 2.9, 'branch': 4.44, 'lds': 10.9, 'vmem': 4.44, 'smem': 2.9, 'other': 4.44}
CLASSES = ['vop2_add', 'vop2_mul', 'vop2', 'vop3', 'dpp', 'pk', 'trans', 'accvgpr', 'mov', 'lane', 'mfma', 's_nop', 's_nop1', 's_waitcnt', 'salu', 'branch', 'lds', 'vmem', 'smem', 'other']


def classify(line):
    m = re.match(r'\s+([a-z_0-9]+)', line)
    if not m:
        return None
    op = m.group(1)
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('v_'):
        if 'dpp' in op or 'quad_perm' in line or 'row_' in line: return 'dpp'
        if op.startswith('v_pk_'): return 'pk'
        if op.startswith('v_accvgpr'): return 'accvgpr'
        if re.match(r'v_(rcp|rsq|sqrt|sin|cos|exp|log)_', op): return 'trans'
        if op.startswith(('v_readlane', 'v_readfirstlane', 'v_writelane', 'v_permlane')): return 'lane'
        if op.startswith('v_mov_b32'): return 'mov'
        if op.endswith('_e32') and 'lit' not in line and not re.search(r'0x[0-9a-f]{5,}', line):
            if re.match(r'v_(add|sub|subrev)_f32', op): return 'vop2_add'
            if re.match(r'v_(mul|fmac|mac|lshlrev|lshrrev)_', op): return 'vop2_mul'
            return 'vop2'
        return 'vop3'
    if op == 's_nop': return 's_nop' if re.search(r's_nop\s+0', line) else 's_nop1'
    if op == 's_waitcnt': return 's_waitcnt'
    if op.startswith(('s_cbranch', 's_branch')): return 'branch'
    if op.startswith(('s_load', 's_buffer_load', 's_memtime', 's_memrealtime')): return 'smem'
    if op.startswith('s_'): return 'salu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem'
    return 'other' if op[:2] in ('v_', 's_') else None


def compile_listing(kernel, defines, out):
    flags = [f for f in g.HIP_FLAGS if f not in ('-shared', '-fPIC')]
    cmd = [g.HIPCC] + flags + ['--cuda-device-only', '-S', '-DLL_KERNELS_ONLY=' + kernel] + ['-D' + d for d in defines] + ['-o', out, os.path.join(g.CSRC, 'llenv.hip')]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    txt = open(out).read().split('\n')
    name = kernel.split('<')[0].split('(')[0]
    starts = [i for i, l in enumerate(txt) if re.match(r'^_Z\d+' + name + r'\w*:', l)]
    i = starts[0]
    j = next(k for k in range(i, len(txt)) if txt[k].startswith('.Lfunc_end'))
    return txt[i:j]


def loop_ranges(lines):
    """(first, last, depth) of every annotated loop: header label up to the last backward branch to it."""
    out = []
    for i, l in enumerate(lines):
        m = re.match(r'^(\.LBB\d+_\d+):.*Loop Header: Depth=(\d+)', l)
        if not m:
            continue
        head, depth, last = m.group(1), int(m.group(2)), i
        for j in range(i, len(lines)):
            b = re.match(r'\s+s_c?branch\w*\s+(\.LBB\d+_\d+)', lines[j])
            if b and b.group(1) == head:
                last = j
        out.append((i, last, depth))
    return out


def ledger(lines):
    """{phase: {class: static count}} in listing order, with the phase's position (for the table's order)."""
    led, order, cur = {}, [], 'kernel.entry'
    for l in lines:
        m = re.search(r'; LLPHASE (\S+)', l)
        if m:
            cur = m.group(1)
            continue
        c = classify(l)
        if c is None:
            continue
        if cur not in led:
            led[cur] = dict.fromkeys(CLASSES, 0)
            order.append(cur)
        led[cur][c] += 1
    return led, order


def trips(phase, p_limit):
    w = WEIGHTS.get(phase, 1.0)
    if phase in ('pgs.limit_round', 'sub.limit_gram'): w = p_limit
    if phase.startswith('pgs.'): return N_SUB * N_ITER * w
    if phase.startswith('sub.'): return N_SUB * w
    return w


def read_costs(path):
    """cycles per instruction by class from an issue_probe output (falls back to the defaults per class)."""
    cost = dict(DEFAULT_COST)
    if not path or not os.path.exists(path):
        return cost, 'defaults (profiles/r04_valu_issue.txt)'
    rows = {}
    for l in open(path):
        m = re.match(r'^(.{62})\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s*$', l)
        if m:
            rows[m.group(1).strip()] = float(m.group(4))

    def pick(*keys):
        for k, v in rows.items():
            if all(x in k for x in keys):
                return v
        return None
    base = pick('v_add_f32_e32 (VOP2')
    k = (4.44 / base) if base else 1.0           # the probe's counter runs at a fixed clock, the shader does not: normalised to 4.44 shader cycles per v_add_f32 (profiles/r04_valu_issue.txt)
    for cls, keys in (('vop2_add', ('v_add_f32_e32 (VOP2',)), ('vop2_mul', ('v_fmac_f32_e32',)), ('vop3', ('v_fma_f32 (VOP3',)), ('dpp', ('v_fmac_f32_dpp',)), ('pk', ('v_pk_fma',)), ('trans', ('v_rsq',)),
                      ('accvgpr', ('v_accvgpr_read_b32',)), ('mov', ('v_mov_b32_e32',)), ('lane', ('v_readlane',)), ('s_nop', ('s_nop 0',)), ('s_nop1', ('s_nop 1',)), ('lds', ('ds_read_b32 (16',)),
                      ('mfma', ('16x16x1_4b_f32, one acc',))):
        v = pick(*keys)
        if v:
            cost[cls] = v * k
    pair, fma = pick('v_fma_f32 + s_mov_b32'), pick('v_fma_f32 (VOP3')
    if pair and fma:
        cost['salu'] = (2 * pair - fma) * k         # what a scalar instruction ADDS next to a vector one (a pair costs less than the sum of the two alone)
    cost['vop2'] = cost['vop3']
    cost['s_waitcnt'] = cost['smem'] = cost['salu']
    cost['branch'] = cost['other'] = cost['s_nop']
    cost['vmem'] = cost['s_nop']
    return cost, path


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--kernel', default='pmc_step_kernel<1, false, true, true>(StepParams)')
    ap.add_argument('--costs', default=os.path.join(ROOT, 'profiles', 'r06_issue_probe.txt'))
    ap.add_argument('--counters', default=os.path.join(ROOT, 'profiles', 'r06_pmc_step_kernel_counters.json'))
    ap.add_argument('--define', action='append', default=[])
    ap.add_argument('--p-limit', type=float, default=None, help='share of wave-substeps with a limit row (default: fitted to the instruction counter)')
    ap.add_argument('--traj', type=float, default=0.0, help='1: the step records unroll rows (N > 1)')
    ap.add_argument('--clock-ghz', type=float, default=2.33)
    ap.add_argument('--md', default=None)
    ap.add_argument('--keep', default='/tmp/issue_ledger')
    args = ap.parse_args()
    os.makedirs(args.keep, exist_ok=True)
    WEIGHTS['tail.unroll_row'] = args.traj
    marked = compile_listing(args.kernel, ['PMC_MARKS'] + args.define, os.path.join(args.keep, 'marked.s'))
    plain = compile_listing(args.kernel, args.define, os.path.join(args.keep, 'plain.s'))
    led, order = ledger(marked)
    cost, cost_src = read_costs(args.costs)
    n_plain = sum(1 for l in plain if classify(l))
    n_marked = sum(sum(v.values()) for v in led.values())
    counters = json.load(open(args.counters)) if args.counters and os.path.exists(args.counters) else None
    meas_inst = meas_slots = None
    if counters:
        meas_inst = counters['_notes']['instructions_per_wave_per_control_step']
        meas_slots = counters['_notes']['issue_slots_per_wave_per_control_step']

    def total(p_limit, what='n'):
        t = 0.0
        for ph in order:
            k = trips(ph, p_limit)
            for c in CLASSES:
                if what == 'n' and c in ('vmem', 'branch', 'smem', 's_waitcnt', 's_nop', 's_nop1'):
                    continue            # the counter sum is SQ_INSTS_VALU + SALU + LDS: no memory, branch, wait or nop instructions (SALU excludes s_nop / s_waitcnt on gfx9)
                t += led[ph][c] * k * (cost[c] if what == 'cycles' else 1.0)
        return t
    p_limit = args.p_limit
    if p_limit is None:
        p_limit = 0.85
        if meas_inst:
            lo, hi = 0.0, 1.0
            for _ in range(40):
                mid = 0.5 * (lo + hi)
                if total(mid) < meas_inst: lo = mid
                else: hi = mid
            p_limit = 0.5 * (lo + hi)
    out = []
    w = out.append
    w('# Issue-slot ledger: %s' % args.kernel)
    w('')
    w('`python tools/issue_ledger.py%s` -- static instructions of the marked listing (hipcc -S -DPMC_MARKS) by phase and class x trip counts; cycles = count x the single-wave issue cost of the class'
      % ''.join(' --define ' + d for d in args.define))
    w('(%s; %.2f GHz).  Marked listing: %d instructions; unmarked (as shipped): %d.' % (os.path.relpath(cost_src, ROOT) if os.path.exists(str(cost_src)) else cost_src, args.clock_ghz, n_marked, n_plain))
    w('')
    w('Cost per instruction class (cycles, one wave alone on its SIMD): ' + ', '.join('%s %.2f' % (c, cost[c]) for c in CLASSES if c != 'other'))
    w('')
    w('| phase | trips per control step | static | ' + ' | '.join(CLASSES[:-1]) + ' | dynamic instr | est. us |')
    w('|---|---|---|' + '---|' * (len(CLASSES) - 1) + '---|---|')
    grand_n = grand_c = 0.0
    groups = {}
    for ph in order:
        k = trips(ph, p_limit)
        st = sum(led[ph].values())
        dyn = st * k
        cyc = sum(led[ph][c] * cost[c] for c in CLASSES) * k
        grand_n += dyn; grand_c += cyc
        gname = ph.split('.')[0]
        groups.setdefault(gname, [0.0, 0.0])
        groups[gname][0] += dyn; groups[gname][1] += cyc
        w('| %s | %.2f | %d | ' % (ph, k, st) + ' | '.join(str(led[ph][c]) if led[ph][c] else '' for c in CLASSES[:-1]) + ' | %.0f | %.2f |' % (dyn, cyc / args.clock_ghz / 1e3))
    w('| **total** | | %d | ' % n_marked + ' | '.join(str(sum(led[ph][c] for ph in order)) for c in CLASSES[:-1]) + ' | **%.0f** | **%.1f** |' % (grand_n, grand_c / args.clock_ghz / 1e3))
    w('')
    w('By group: ' + '; '.join('%s %.0f instr = %.1f us' % (k_, v[0], v[1] / args.clock_ghz / 1e3) for k_, v in groups.items()))
    w('')
    w('Conditional phases: pgs.limit_round and sub.limit_gram x %.3f (%s), sub.self_rows / pgs.self_turns x 0.42, tail.episode_end_reseed x 0.073 (profiles/r05_timeline.txt); tail.obstacle_check and tail.unroll_row do not run in config 2 at N = 1; kernel.entry once per 32 steps.'
      % (p_limit, 'fitted so that the VALU + SALU + LDS sum meets the counter' if args.p_limit is None and meas_inst else 'given'))
    if meas_inst:
        w('')
        w('Reconciliation with %s: counted VALU + SALU + LDS instructions per wave per control step %.0f (ledger, same classes: %.0f); issue slots %.0f quad-cycles = %.0f cycles = %.1f us'
          % (os.path.relpath(args.counters, ROOT), meas_inst, total(p_limit), meas_slots, meas_slots * 4, meas_slots * 4 / args.clock_ghz / 1e3))
        ratio = meas_slots * 4 / grand_c
        w('(ledger estimate from per-class costs: %.0f cycles = %.1f us, %.2f x the measured cycles: the class costs are those of 64 instructions of ONE form back to back; a mixed stream pays a little less per '
          'instruction.  Scaled by that ratio the groups read: %s.)' % (grand_c, grand_c / args.clock_ghz / 1e3, 1.0 / ratio,
                                                                       '; '.join('%s %.1f us (%.0f %%)' % (k_, v[1] * ratio / args.clock_ghz / 1e3, 100.0 * v[1] / grand_c) for k_, v in groups.items())))
    text = '\n'.join(out) + '\n'
    print(text)
    if args.md:
        open(args.md, 'w').write(text)


if __name__ == '__main__':
    main()
