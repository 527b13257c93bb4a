"""Static look at the gfx950 code of the step kernels (no GPU needed): registers, scratch, code size, and the instruction
count of the substep loop / the solver loop inside it -- the quantities DESIGN.md 5.1 argues with.  hipcc cross-compiles here.

    python tools/isa_stats.py [pmc|epmc|sepmc] [--keep /tmp/isa]      -> one line per kernel variant + loop sizes

The dynamic count per control step is roughly  tail + n_sub * (substep body - solver loop + n_iter * solver loop).
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g   # noqa: E402

OPS = ('v_', 's_', 'ds_', 'global_', 'buffer_', 'flat_', 'scratch_')


def compile_asm(out_dir, extra=()):
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, 'llenv.s')
    flags = [f for f in g.HIP_FLAGS if f not in ('-shared', '-fPIC')]
    subprocess.check_call([g.HIPCC] + flags + list(extra) + ['--cuda-device-only', '-S', '-o', out, os.path.join(g.CSRC, 'llenv.hip')],
                          stderr=subprocess.DEVNULL)
    return out


def kernels(path):
    txt = open(path).read().split('\n')
    starts = [(i, l.split(':')[0]) for i, l in enumerate(txt) if re.match(r'^_Z\w+:', l)]
    for n, (i, name) in enumerate(starts):
        j = starts[n + 1][0] if n + 1 < len(starts) else len(txt)
        yield name, txt[i:j]


def is_inst(l):
    m = re.match(r'\s+([a-z_0-9]+)', l)
    return bool(m) and m.group(1).startswith(OPS)


def loops(lines):
    """(header label, first line, last line, depth) of every loop the compiler annotated."""
    lab = {}
    for i, l in enumerate(lines):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            lab[m.group(1)] = i
    out = []
    for i, l in enumerate(lines):
        m = re.match(r'^(\.LBB\d+_\d+):.*Loop Header: Depth=(\d+)', l)
        if not m:
            continue
        head, depth = m.group(1), int(m.group(2))
        last = i
        for j in range(i, len(lines)):                       # the last backward branch to the header closes the loop
            b = re.match(r'\s+s_c?branch\w*\s+(\.LBB\d+_\d+)', lines[j])
            if b and b.group(1) == head:
                last = j
        out.append((head, i, last, depth))
    return out


def main():
    which = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith('-') else 'pmc'
    keep = sys.argv[sys.argv.index('--keep') + 1] if '--keep' in sys.argv else '/tmp/isa'
    path = compile_asm(keep)
    for name, lines in kernels(path):
        if which + '_step_kernel' not in name:
            continue
        meta = {k: v for k, v in re.findall(r'; (NumVgprs|NumAgprs|ScratchSize|Occupancy|codeLenInByte)[ =:]+(\d+)', '\n'.join(lines))}
        n_all = sum(is_inst(l) for l in lines)
        print('%s: vgpr %s agpr %s scratch %s occupancy %s code %s B, %d instructions' % (
            name, meta.get('NumVgprs'), meta.get('NumAgprs'), meta.get('ScratchSize'), meta.get('Occupancy'), meta.get('codeLenInByte'), n_all))
        for head, a, b, depth in loops(lines):
            n = sum(is_inst(l) for l in lines[a:b + 1])
            if n >= 100:
                nops = sum(1 for l in lines[a:b + 1] if re.match(r'\s+s_nop', l))
                waits = sum(1 for l in lines[a:b + 1] if re.match(r'\s+s_waitcnt', l))
                acc = sum(1 for l in lines[a:b + 1] if re.match(r'\s+v_accvgpr', l))
                mov = sum(1 for l in lines[a:b + 1] if re.match(r'\s+v_mov_b32_e32', l))
                print('   loop %-10s depth %d: %5d instructions (s_nop %d, s_waitcnt %d, v_accvgpr %d, v_mov %d)' % (head, depth, n, nops, waits, acc, mov))


if __name__ == '__main__':
    main()
