#!/usr/bin/env python3
"""Resolve the conditional-compilation blocks of a source file whose conditions only involve a given set of macros
(NAME=value, or NAME= for "not defined"), leaving every other block alone.  Used once in round 4 to take the timing-only
ablation switches (KQ_DIAG_*, KQ_STAMP, PM_ABL_*, PM_BPIPE ...) out of the product sources; the sources WITH the
switches are kept under profiles/tools/r03_src/.

    strip_ifdefs.py file NAME=1 OTHER= ...   (rewrites the file in place)
"""
import re
import sys


def main():
    path, known = sys.argv[1], {}
    for a in sys.argv[2:]:
        k, v = a.split('=', 1)
        known[k] = v if v != '' else None          # None: not defined
    out, stack = [], []                             # stack entries: dict(resolved, taken, done, keep_lines)
    ident = re.compile(r'[A-Za-z_]\w*')

    def evaluate(kind, expr):
        if kind in ('ifdef', 'ifndef'):
            name = expr.strip()
            if name not in known:
                return None
            d = known[name] is not None
            return d if kind == 'ifdef' else not d
        e = re.sub(r'defined\s*\(\s*(\w+)\s*\)|defined\s+(\w+)', lambda m: ('@%s@' % (m.group(1) or m.group(2))), expr)
        names = set(re.findall(r'@(\w+)@', e))
        if any(n not in known for n in names):
            return None
        e = re.sub(r'@(\w+)@', lambda m: '1' if known[m.group(1)] is not None else '0', e)
        for n in set(ident.findall(e)):
            if n in known and known[n] is not None:
                e = re.sub(r'\b%s\b' % n, known[n], e)
            elif n in known:
                e = re.sub(r'\b%s\b' % n, '0', e)
            else:
                return None
        e = e.replace('&&', ' and ').replace('||', ' or ')
        e = re.sub(r'!(?!=)', ' not ', e)
        try:
            return bool(eval(e, {'__builtins__': {}}))
        except Exception:
            return None

    for line in open(path).read().split('\n'):
        m = re.match(r'\s*#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)', line)
        active = all(f['emit'] for f in stack)
        if not m:
            if active:
                out.append(line)
            continue
        kind, rest = m.group(1), re.sub(r'//.*', '', m.group(2))
        if kind in ('if', 'ifdef', 'ifndef'):
            r = evaluate(kind, rest) if active else False
            if active and r is None:
                stack.append(dict(resolved=False, emit=True, taken=True))
                out.append(line)
            else:
                stack.append(dict(resolved=True, emit=bool(r) and active, taken=bool(r), outer=active))
        elif kind == 'elif':
            f = stack[-1]
            if not f['resolved']:
                out.append(line)
            else:
                r = evaluate('if', rest)
                if r is None:
                    raise SystemExit('unresolvable #elif after a resolved #if: ' + line)
                f['emit'] = f['outer'] and (not f['taken']) and bool(r)
                f['taken'] = f['taken'] or bool(r)
        elif kind == 'else':
            f = stack[-1]
            if not f['resolved']:
                out.append(line)
            else:
                f['emit'] = f['outer'] and not f['taken']
                f['taken'] = True
        else:
            f = stack.pop()
            if not f['resolved']:
                out.append(line)
    open(path, 'w').write('\n'.join(out))


if __name__ == '__main__':
    main()
