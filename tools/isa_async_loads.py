"""Static check of every register-returning vector-memory operation in one kernel of a gfx950
disassembly (llvm-objdump -d of the code object, or hipcc -S output).

    python tools/isa_async_loads.py <file> <kernel-name-substring>

An operation issued from inline asm is unknown to the compiler's wait bookkeeping: as far as the
compiler can tell its destination registers hold their value when the asm statement ends, so it
may read, copy or reuse them while the data is still on its way; when the late write-back lands in
a register that by then holds something else (an address, say), the kernel stores or loads through
garbage.  For every global_load_dword* / global_atomic_* with a VGPR destination this script walks
forward in listing order, counts the vector-memory operations issued after it on the path every
wave takes (operations inside short forward-branched regions are not counted: some waves skip them), and
finds the first s_waitcnt vmcnt(N) with N <= that count, i.e. the first wait that covers the
operation.  It reports every instruction before that wait that reads or writes a destination
register.  Listing order is execution order for the fully unrolled bodies this is meant for; a
backward branch (loop) before the covering wait is reported as such.
"""
import re
import sys


def regs(tok):
    tok = tok.strip().rstrip(',')
    m = re.fullmatch(r'v(\d+)', tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


VMEM = ('global_load', 'global_store', 'global_atomic', 'buffer_load', 'buffer_store', 'buffer_atomic', 'flat_',
        'scratch_load', 'scratch_store')


def parse(path, kname):
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if kname in l and l.rstrip().endswith(':'))
    out = []
    for i in range(start + 1, len(lines)):
        raw = lines[i]
        ins = raw.split('//')[0].split(';')[0].strip()
        if not ins or ins.endswith(':') or ins.startswith('.'):
            if raw.rstrip().endswith('>:'):
                break
            continue
        addr = None
        m = re.search(r'//\s*([0-9A-Fa-f]+):', raw)
        if m:
            addr = int(m.group(1), 16)
        out.append((i + 1, ins, addr))
        if ins.startswith('s_endpgm'):
            break
    return out


def main():
    path, kname = sys.argv[1:3]
    body = parse(path, kname)
    addr_to_idx = {a: k for k, (_, _, a) in enumerate(body) if a is not None}
    # forward-branched regions: [k+1, target) for s_cbranch at k with a target further down
    skipped = [False] * len(body)
    backward = [False] * len(body)
    for k, (ln, ins, addr) in enumerate(body):
        mn = ins.split()[0]
        if mn.startswith('s_cbranch') or mn == 's_branch':
            m = re.search(r'<[^>]*\+0x([0-9a-f]+)>', open(path).read().split('\n')[ln - 1])
            tgt = None
            if m and body[0][2] is not None:
                base = body[0][2]
                tgt = addr_to_idx.get(base + int(m.group(1), 16))
                # the symbol offset is relative to the kernel's first instruction
            if tgt is not None and tgt > k:
                # a short forward branch guards a conditional region; a long one leaves a loop or
                # skips a whole pass, whose body is the path being checked
                if tgt - k <= 400:
                    for j in range(k + 1, tgt):
                        skipped[j] = True
            elif tgt is not None:
                backward[k] = True
    bad = 0
    checked = 0
    for k, (ln, ins, _) in enumerate(body):
        mn = ins.split()[0]
        if not (mn.startswith('global_load_dword') or (mn.startswith('global_atomic') and ' sc0' in ins)):
            continue
        dst = regs(ins.split(None, 1)[1].split(',')[0])
        if not dst:
            continue
        checked += 1
        issued_after = 0
        events = []
        covered = None
        loops = 0
        same_region = skipped[k]   # still inside the conditional region the operation itself sits in
        for j in range(k + 1, len(body)):
            ln2, ins2, _ = body[j]
            mn2 = ins2.split()[0]
            if backward[j]:
                loops += 1
            if not skipped[j]:
                same_region = False
            if mn2 == 's_waitcnt':
                m = re.search(r'vmcnt\((\d+)\)', ins2)
                if m and int(m.group(1)) <= issued_after and (same_region or not skipped[j]):
                    covered = (ln2, ins2, issued_after)
                    break
                continue
            if mn2.startswith(VMEM) and (same_region or not skipped[j]):
                issued_after += 1
            ops = ins2.split(None, 1)[1] if ' ' in ins2 else ''
            toks = [t for t in re.split(r'[ ,]+', ops) if t]
            named = [t for t in toks if regs(t) & dst]
            if named:
                is_store = mn2.startswith(('global_store', 'buffer_store', 'ds_write', 'scratch_store'))
                writes = bool(toks and regs(toks[0]) & dst) and not is_store and not mn2.startswith(('v_cmp', 's_'))
                reads = any(regs(t) & dst for t in (toks if is_store else toks[1:]))
                kind = ('WRITE' if writes else '') + ('+' if writes and reads else '') + ('READ' if reads else '')
                events.append((ln2, kind, ins2, skipped[j]))
        status = 'ok' if not events else 'HAZARD'
        if events or covered is None:
            bad += 1
        print(f"line {ln}: {ins}")
        if covered:
            print(f"   covered by line {covered[0]}: {covered[1]} ({covered[2]} vector-memory operations issued in between on every wave's path; loops passed: {loops})")
        else:
            print("   NOT covered by any counted wait before the kernel ends")
        for ln2, kind, ins2, sk in events:
            print(f"   !! {kind} before the covering wait, line {ln2}{' (inside a conditional region)' if sk else ''}: {ins2}")
    print(f"{checked} operations checked, {bad} with findings")


if __name__ == '__main__':
    main()
