"""Static check of a register that a returning memory operation fills asynchronously.

    python tools/isa_reg_lifetime.py <file.s> <kernel-name-substring> <opcode-prefix> [dest-operand-index]

For every instruction of the kernel whose mnemonic starts with <opcode-prefix> (e.g.
global_atomic_add) the script takes its destination VGPR and walks forward to the first instruction
that reads it.  It prints every instruction in between that mentions the register (a write there
means the compiler reused the register while the hardware may still write it back; a read before
the wait that covers the operation means a copy of a value that has not arrived) and what kind of
wait, if any, stands between the issue and the first read.  Straight-line listing order is used,
which is the execution order of the fully unrolled kernels this is meant for; branches are reported.
"""
import re
import sys


def regs_of(tok):
    tok = tok.strip().rstrip(',')
    m = re.fullmatch(r'v(\d+)', tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def main():
    path, kname, op = sys.argv[1:4]
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and kname in l and l.rstrip().endswith(tuple(':')) or (l.startswith('_Z') and kname in l and ':' in l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
    body = [(i, l.split(';')[0].strip()) for i, l in enumerate(lines[start:end + 1], start)]
    body = [(i, l) for i, l in body if l and not l.endswith(':') and not l.startswith('.')]
    issues = [k for k, (i, l) in enumerate(body) if l.startswith(op)]
    print(f"{len(issues)} x {op} in {lines[start].split(':')[0]}")
    for k in issues:
        i, l = body[k]
        ops = l.split(None, 1)[1].split(',')
        dst = regs_of(ops[0])
        (d,) = dst
        waits, between, branches = [], [], 0
        first_read = None
        for kk in range(k + 1, len(body)):
            j, m = body[kk]
            mn = m.split(None, 1)[0]
            if mn.startswith('s_cbranch') or mn == 's_branch':
                branches += 1
            if mn == 's_waitcnt':
                waits.append((j, m))
            if len(m.split(None, 1)) < 2:
                continue
            toks = [t for t in re.split(r'[ ,]+', m.split(None, 1)[1]) if t]
            mentions = [t for t in toks if d in regs_of(t)]
            if not mentions:
                continue
            # the first operand of a VALU/VMEM-load instruction is its destination
            is_write = d in regs_of(toks[0]) and not mn.startswith(('global_store', 'ds_write', 'buffer_store', 'v_cmp', 's_'))
            is_read = any(d in regs_of(t) for t in toks[1:]) or not is_write
            if is_write and not is_read:
                between.append((j, 'WRITE', m))
                if mn.startswith(op):
                    break
                continue
            first_read = (j, m)
            break
        vm = [w for w in waits if 'vmcnt' in w[1]]
        print(f"line {i + 1}: {l}")
        print(f"   first read of v{d}: line {first_read[0] + 1 if first_read else None}: {first_read[1] if first_read else None}")
        print(f"   instructions between: {(first_read[0] - i) if first_read else '-'}; branches: {branches}; s_waitcnt with vmcnt in between: {len(vm)}; "
              f"last before the read: {vm[-1][1] if vm else None}")
        for j, kind, m in between:
            print(f"   !! {kind} of v{d} before its first read, line {j + 1}: {m}")


if __name__ == '__main__':
    main()
