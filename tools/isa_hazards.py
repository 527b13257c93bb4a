"""Static check of a hipcc -save-temps .s for the software-visible hazards the compiler cannot see across inline-asm boundaries (cdna_hip_programming.md 5.7 item 2):
  A  a VALU write of a VGPR followed within 2 wait states by a DPP read of it (src0 of a *_dpp instruction)
  B  a VALU write of a VGPR followed within 2 wait states by v_permlane16/32_swap touching it (either operand)
  C  a transcendental result read by a non-transcendental VALU in the next wait state
  D  a VALU write of a VGPR followed within 1 wait state by v_readlane / v_readfirstlane of it
Linear scan per function (fall-through order; a label does not reset the window, a branch does not follow its target).
    python tools/isa_hazards.py file.s [function-substring]"""
import re, sys
TRANS = ('v_rsq_', 'v_rcp_', 'v_sqrt_', 'v_sin_', 'v_cos_', 'v_exp_', 'v_log_')
def regs(tok):
    tok = tok.strip().lstrip('-|').rstrip('|')
    m = re.match(r'^v(\d+)$', tok)
    if m: return [int(m.group(1))]
    m = re.match(r'^v\[(\d+):(\d+)\]$', tok)
    if m: return list(range(int(m.group(1)), int(m.group(2)) + 1))
    return []
def scan(lines, name):
    age = {}          # vgpr -> (wait states since its last VALU write, writer text, was transcendental, inside asm)
    found = []
    in_asm = False
    for ln, raw in lines:
        s = raw.split(';')[0].strip() if not raw.strip().startswith(';') else ''
        if ';;#ASMSTART' in raw or '#ASMSTART' in raw: in_asm = True
        if ';;#ASMEND' in raw or '#ASMEND' in raw: in_asm = False
        if not s or s.endswith(':') or s.startswith('.'): continue
        parts = s.split(None, 1)
        op = parts[0]; args = [a.strip() for a in (parts[1] if len(parts) > 1 else '').split(',')]
        states = 1
        if op == 's_nop': states = int(args[0], 0) + 1
        is_valu = op.startswith('v_')
        dpp = any(k in s for k in ('quad_perm:', 'row_shl:', 'row_shr:', 'row_ror:', 'row_newbcast:', 'row_share:', 'row_mirror', 'row_half_mirror', 'row_bcast:', 'wave_'))
        def chk(kind, rr, need):
            for r in rr:
                if r in age and age[r][0] < need:
                    found.append((name, ln, kind, s, 'v%d written %d state(s) earlier by: %s%s%s' % (r, age[r][0], age[r][1], ' [writer inside asm]' if age[r][3] else '', ' [reader inside asm]' if in_asm else '')))
        if is_valu:
            if dpp and len(args) > 1: chk('A dpp', regs(args[1].split()[0]), 2)
            if 'permlane' in op and 'swap' in op: chk('B swap', regs(args[0]) + regs(args[1].split()[0]), 2)
            if op.startswith(('v_readlane', 'v_readfirstlane')) and len(args) > 1: chk('D readlane', regs(args[1].split()[0]), 1)
            if not op.startswith(TRANS):
                for a in args[1:]:
                    for r in regs(a.split()[0] if a else ''):
                        if r in age and age[r][2] and age[r][0] < 1:
                            found.append((name, ln, 'C trans', s, 'v%d from %s' % (r, age[r][1])))
        for r in list(age):
            age[r] = (age[r][0] + states,) + age[r][1:]
        if is_valu and not op.startswith(('v_cmp', 'v_readlane', 'v_readfirstlane', 'v_nop')):
            dst = regs(args[0]) if args else []
            if 'permlane' in op and 'swap' in op: dst += regs(args[1].split()[0])
            for r in dst: age[r] = (0, s, op.startswith(TRANS), in_asm)
    return found
def main():
    path = sys.argv[1]; want = sys.argv[2] if len(sys.argv) > 2 else ''
    cur, buf, funcs = None, [], []
    for i, l in enumerate(open(path), 1):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            if cur: funcs.append((cur, buf))
            cur, buf = m.group(1), []
        elif cur: buf.append((i, l.rstrip('\n')))
    if cur: funcs.append((cur, buf))
    tot = 0
    for n, b in funcs:
        if want not in n: continue
        f = scan(b, n)
        kinds = {}
        for x in f: kinds[x[2]] = kinds.get(x[2], 0) + 1
        print('%-70s %s' % (n[:70], kinds or 'clean'))
        for x in f[:int(sys.argv[3]) if len(sys.argv) > 3 else 4]: print('     line %d %s: %s   <- %s' % (x[1], x[2], x[3], x[4]))
        tot += len(f)
    print('total', tot)
main()
