#!/usr/bin/env python3
"""Instruction histogram per kernel from a hipcc -save-temps .s file (static counts, not dynamic)."""
import collections
import re
import sys


def main(path, pattern=''):
    text = open(path).read()
    # kernels: label "name:" ... "s_endpgm"
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\n\s*s_endpgm', text, flags=re.S | re.M):
        name, body = m.group(1), m.group(2)
        if pattern and pattern not in name:
            continue
        ops = [ln.split()[0] for ln in body.splitlines() if ln.startswith('\t') and not ln.strip().startswith(('.', ';'))]
        hist = collections.Counter(ops)
        valu = sum(c for o, c in hist.items() if o.startswith('v_'))
        salu = sum(c for o, c in hist.items() if o.startswith('s_'))
        mem = sum(c for o, c in hist.items() if o.startswith(('global_', 'buffer_', 'flat_', 'ds_', 'scratch_')))
        mul = sum(c for o, c in hist.items() if o.startswith(('v_mul_lo', 'v_mul_hi', 'v_mad_u64')))
        print(f'{name}: total {len(ops)} valu {valu} (32-bit mul {mul}) salu {salu} mem {mem}')
        print('   ' + ', '.join(f'{o}:{c}' for o, c in hist.most_common(28)))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else '')
