#!/usr/bin/env python3
"""Dev helper: print a Fortran source with `if (use_ATRC) then ... else ... end if`
blocks reduced to their else-branch (use_ATRC is false in every build we target).
Reads only; used for studying the reference, never shipped."""
import sys, re
lines = open(sys.argv[1]).read().split('\n')
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 1
hi = int(sys.argv[3]) if len(sys.argv) > 3 else len(lines)
out = []
stack = []  # entries: ('atrc', state) or ('other',)
skip_depth = 0
for n, ln in enumerate(lines, 1):
    s = ln.strip().lower()
    is_if_then = re.match(r'(else\s+)?if\s*\(.*\)\s*then\s*$', s) and not s.startswith('else')
    if is_if_then:
        if re.match(r'if\s*\(\s*use_atrc\s*\)\s*then', s) or re.match(r'if\s*\(use_trc\s*\.and\.\s*use_atrc\)\s*then', s):
            stack.append(['atrc', 'then'])
            continue
        stack.append(['other'])
    elif s.startswith('else') and stack and stack[-1][0] == 'atrc' and not s.startswith('else if'):
        stack[-1][1] = 'else'
        continue
    elif re.match(r'end\s*if', s):
        if stack:
            top = stack.pop()
            if top[0] == 'atrc':
                continue
    skipping = any(e[0] == 'atrc' and e[1] == 'then' for e in stack)
    if not skipping and lo <= n <= hi:
        out.append(f"{n:5d} {ln}")
print('\n'.join(out))
