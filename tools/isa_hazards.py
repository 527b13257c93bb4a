"""Static check of a hipcc -save-temps .s for the software-visible hazards the compiler cannot see across inline-asm boundaries (cdna_hip_programming.md 5.7 item 2):
  A  a VALU write of a VGPR followed within 2 wait states by a DPP read of it (src0 of a *_dpp instruction)
  B  a VALU write of a VGPR followed within 2 wait states by v_permlane16/32_swap touching it (either operand)
  C  a transcendental result read by a non-transcendental VALU in the next wait state
  D  a VALU write of a VGPR followed within 1 wait state by v_readlane / v_readfirstlane of it
Linear scan per function (fall-through order; a label does not reset the window, a branch does not follow its target).
  E  (a code-generation defect, not a hazard -- the root cause of round 4's seven-rays-per-chunk chase-tag failure, HISTORY.md)  a block that is the target of an
     s_cbranch_execz -- the join block behind a masked region -- and runs VALU writes AHEAD of the s_or_b64 exec that re-converges the wavefront there: those
     writes reach only the lanes that were inside the region.  hipcc (ROCm 7.2) placed a live-range copy there in one build of sepmc_step_kernel<1, false>;
     check_library() looks for it in the code object of the built library itself (llvm-objcopy + clang-offload-bundler + llvm-objdump), __graft_entry__.build()
     and tests/test_built_code.py call it.
    python tools/isa_hazards.py file.s [function-substring] [lines shown]        (a hipcc -save-temps .s)
    python tools/isa_hazards.py --library [path/to/libllenv.so]                  (check E on the shipped code object)"""
import os, re, subprocess, sys, tempfile
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

LABEL = re.compile(r'^(?:(\.LBB\d+_\d+)|<(L\d+)>):')
FUNC = re.compile(r'^(?:(_Z\w+):|<(_Z\w+)>:)')


def exec_restore_scan(text):
    """[(function, label, line number, [VALU instructions ahead of the exec restore])] over a -save-temps .s or an llvm-objdump --symbolize-operands listing."""
    lines = text.split('\n')
    fn, targets = None, {}
    for l in lines:
        m = FUNC.match(l)
        if m: fn = m.group(1) or m.group(2)
        m = re.match(r'\s+s_cbranch_execz\s+(\.LBB\d+_\d+|L\d+)', l)
        if m and fn: targets.setdefault(fn, set()).add(m.group(1))
    fn, found = None, []
    for i, l in enumerate(lines):
        m = FUNC.match(l)
        if m: fn = m.group(1) or m.group(2)
        m = LABEL.match(l)
        lab = m and (m.group(1) or m.group(2))
        if not lab or lab not in targets.get(fn, ()): continue
        pre, j = [], i + 1
        while j < len(lines):
            s = re.split(r';|//', lines[j])[0].strip()
            j += 1
            if not s: continue
            if s.startswith('s_or_b64 exec, exec'):
                if pre: found.append((fn, lab, i + 1, pre))
                break
            if s.startswith('v_') and not s.startswith(('v_readlane', 'v_writelane', 'v_readfirstlane', 'v_cmp', 'v_nop')): pre.append(s); continue
            if s.startswith(('scratch_', 'global_', 'flat_', 'buffer_', 'ds_')): pre.append(s); continue            # a spill reload or a store under the region's mask is the same defect
            if s.startswith(('v_readlane', 's_mov', 's_nop', 's_waitcnt')) and j - i < 16: continue       # an SGPR reload of the saved mask
            break
    return found


def check_library(lib, llvm_bin='/opt/rocm/lib/llvm/bin'):
    """Check E on the gfx950 code object inside a built HIP library; returns the findings (empty = clean)."""
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, 'fat.bin'), os.path.join(d, 'dev.co')
        subprocess.check_call([os.path.join(llvm_bin, 'llvm-objcopy'), '--dump-section', '.hip_fatbin=' + fat, lib, os.path.join(d, 'copy.so')])
        subprocess.check_call([os.path.join(llvm_bin, 'clang-offload-bundler'), '--type=o', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--input=' + fat, '--output=' + co, '--unbundle'])
        dis = subprocess.run([os.path.join(llvm_bin, 'llvm-objdump'), '-d', '--symbolize-operands', '--no-show-raw-insn', '--no-leading-addr', co], check=True, capture_output=True, text=True).stdout
    if dis.count('s_cbranch_execz') < 100:
        raise RuntimeError('the disassembly of %s does not look like the step kernels (%d s_cbranch_execz)' % (lib, dis.count('s_cbranch_execz')))
    return exec_restore_scan(dis)


def main():
    if sys.argv[1] == '--library':
        lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'lifelike_agility_and_play_amd', 'csrc', 'libllenv.so')
        f = check_library(lib)
        for fn, lab, ln, pre in f: print('%s %s (listing line %d): %d VALU writes ahead of the exec restore: %s' % (fn, lab, ln, len(pre), '; '.join(pre[:4])))
        print('check E on %s: %d finding(s)' % (lib, len(f)))
        sys.exit(1 if f else 0)
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
    e = exec_restore_scan(open(path).read())
    for fn, lab, ln, pre in e:
        if want in fn: print('E  %s %s (line %d): %d VALU writes ahead of the exec restore: %s' % (fn[:60], lab, ln, len(pre), '; '.join(pre[:4])))
    print('total', tot, '+ E', len([x for x in e if want in x[0]]))


if __name__ == '__main__':
    main()
