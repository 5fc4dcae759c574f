"""Summarise the instruction interleave of a kernel's loops in a hipcc -save-temps .s file.
    python tools/asm_loop.py <file.s> <mangled-name-substring>"""
import sys


def cls(l):
    op = l.split()[0]
    if 'mfma' in op:
        return 'M'
    if op.startswith(('global_load', 'buffer_load')):
        return 'L'
    if op.startswith(('global_store', 'buffer_store')):
        return 'S'
    if op.startswith('scratch_'):
        return 'SCRATCH'
    if op.startswith('s_waitcnt'):
        return 'W(' + l.split(None, 1)[1].strip() + ')'
    if op.startswith('s_'):
        return 's'
    if op.startswith('v_'):
        return 'v'
    if op.startswith('ds_'):
        return 'D'
    return op


def main():
    s = open(sys.argv[1]).read()
    key = sys.argv[2]
    i = s.index(key)
    i = s.index(':', i)
    j = s.index('.end_amdhsa_kernel', i)
    lines = [l for l in s[i:j].split('\n') if l.strip() and not l.strip().startswith(';')]
    labs = [n for n, l in enumerate(lines) if 'Loop Header' in l]
    for n0 in labs:
        end = [n for n, l in enumerate(lines) if n > n0 and 's_cbranch' in l][0]
        loop = lines[n0 + 1:end + 1]
        seq = [cls(l) for l in loop]
        out, prev, c = [], None, 0
        for x in seq:
            if x == prev and x in ('v', 's', 'L', 'S', 'D'):
                c += 1
            else:
                if prev is not None:
                    out.append(prev + (str(c) if c > 1 else ''))
                prev, c = x, 1
        out.append(prev + (str(c) if c > 1 else ''))
        print(f"loop at {n0}: {len(loop)} instructions, {seq.count('M')} MFMA, {seq.count('v')} VALU, {seq.count('L')} loads")
        print(' '.join(out))


main()
