#!/usr/bin/env python
"""The "current numbers" table at the top of DESIGN.md, generated (never typed): one row per round from the DRIVER's bench
records (BENCH_rNN.json: the line bench.py printed on the driver's box, found in `tail` / `parsed`), then the builder's own
lines of this round (profiles/rNN_bench_*.json, marked as such).   python tools/current_numbers.py [--write]

--write replaces the text between the markers <!-- current-numbers:begin --> / <!-- current-numbers:end --> in DESIGN.md."""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def grab(text, key):
    """the JSON object that follows "key": in a (possibly truncated) bench line"""
    i = text.find('"%s": {' % key)
    if i < 0:
        return None
    j = text.index('{', i)
    depth = 0
    for k in range(j, len(text)):
        depth += text[k] == '{'
        depth -= text[k] == '}'
        if depth == 0:
            try:
                return json.loads(text[j:k + 1])
            except ValueError:
                return None
    return None


def row(label, line):
    st, rs = line.get('stages_ms') or {}, line.get('roofline_stencil') or {}
    f = lambda v, fmt='%.2f': (fmt % v) if isinstance(v, (int, float)) else '-'
    return '| %s | %s | %s | %s / %s | %s | %s | %s | %s | %s | %s | %s |' % (
        label, f(line.get('ms_per_step')), f(line.get('value'), '%.0f'), f(rs.get('avg_kernel_ms'), '%.3f'), f(rs.get('back_to_back_ms'), '%.3f'),
        f(rs.get('frac'), '%.3f'), f(rs.get('valu_roofline_frac'), '%.2f'), f(st.get('flats_ms')), f(st.get('graph_ms')), f(st.get('pits_ms')),
        f(st.get('sweep_ms')), f(st.get('twi_ms')))


def main():
    out = ['| source | ms / step | Mcells/s | stencil in the pipeline / back to back (ms) | stencil: fraction of 8 TB/s | ... of its vector-ALU issue rate | flats | graph | pits | sweep | twi |',
           '|---|---|---|---|---|---|---|---|---|---|---|---|']
    for fn in sorted(glob.glob(os.path.join(ROOT, 'BENCH_r*.json'))):
        d = json.load(open(fn))
        p = d.get('parsed') or {}
        tail = d.get('tail') if isinstance(d.get('tail'), str) else ''
        line = {'ms_per_step': p.get('ms_per_step'), 'value': p.get('value'), 'stages_ms': grab(tail, 'stages_ms'), 'roofline_stencil': grab(tail, 'roofline_stencil')}
        out.append(row('driver, round %s (`%s`)' % (re.search(r'r(\d+)', os.path.basename(fn)).group(1).lstrip('0'), os.path.basename(fn)), line))
    for fn in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_bench_16384.json')) + glob.glob(os.path.join(ROOT, 'profiles', 'r*_bench_16384_other_box.json'))):
        rnd = re.search(r'r(\d+)_', os.path.basename(fn)).group(1)
        if os.path.exists(os.path.join(ROOT, 'BENCH_r%s.json' % rnd)):
            continue                                   # the driver's line of that round is above
        try:
            line = json.load(open(fn))
        except ValueError:
            continue
        who = "builder's own run" + (", another box" if 'other_box' in fn else "")
        out.append(row("%s, round %s (`profiles/%s`)" % (who, rnd.lstrip('0'), os.path.basename(fn)), line))
    text = '\n'.join(out)
    if '--write' in sys.argv:
        fn = os.path.join(ROOT, 'DESIGN.md')
        s = open(fn).read()
        a, b = '<!-- current-numbers:begin -->', '<!-- current-numbers:end -->'
        i, j = s.index(a) + len(a), s.index(b)
        open(fn, 'w').write(s[:i] + '\n' + text + '\n' + s[j:])
    print(text)


if __name__ == '__main__':
    main()
