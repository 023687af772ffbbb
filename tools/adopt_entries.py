#!/usr/bin/env python3
"""
Merges re-tuned entries into a shipped tile table only where the new configuration wins by a margin:
    python tools/adopt_entries.py <shipped.json> <retuned.json> <autotune table .txt> [--min-gain 0.03] [--prefix v7:]
For every entry of <retuned.json> whose configuration name starts with --prefix and differs from the shipped entry of the
same (batch, m, n, k, taps, stride, residual), the TFLOP/s of both configurations are read from the autotune table of the
same run (one line per op, one column per configuration id) and the entry is adopted only if new >= old * (1 + min-gain):
a configuration of another summation-order family fixes the layer's kernel for every batch size, so ties stay where
they are.  Prints what it did; rewrites <shipped.json> in place.
"""
import argparse
import json
import re
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('shipped')
    ap.add_argument('retuned')
    ap.add_argument('table')
    ap.add_argument('--min-gain', type=float, default=0.03)
    ap.add_argument('--prefix', default='v7:')
    ap.add_argument('--no-propagate', action='store_true',
                    help='the adopted configurations belong to the bitwise family of the entries they replace (same K order): the '
                         'entries of the other batch sizes keep their own measured configurations')
    ap.add_argument('--names', required=True, help='json list of configuration names by id (mdhip_conv_cfg_name), for the table columns')
    a = ap.parse_args()
    names = json.load(open(a.names))
    key = lambda e: (int(e.get('batch', 32)), e['m'], e['n'], e['k'], e['ntaps'], e['stride'], e['has_res'])
    shipped = json.load(open(a.shipped))
    old = {key(e): e for e in shipped['entries']}
    rows = {}
    for line in open(a.table):
        m = re.match(r'(.{34}) M=\s*(\d+) N=\s*(\d+) K=\s*(\d+) .*?\| (.*)$', line)
        if m:
            # (a stride-2 3x3 conv and a stride-1 one can share M, N and K: the op name tells them apart)
            stride = 2 if re.search(r'conv \dx\ds2', m.group(1)) else 1
            rows[(int(m.group(2)), int(m.group(3)), int(m.group(4)), stride)] = [float(v) for v in m.group(5).split()]
    adopted = kept = 0
    for e in json.load(open(a.retuned))['entries']:
        k = key(e)
        if not str(e.get('name', '')).startswith(a.prefix) or k not in old or old[k].get('name') == e.get('name'):
            continue
        tf = rows.get((e['m'], e['n'], e['k'], e['stride']))
        if not tf or old[k].get('name') not in names:
            continue
        t_new, t_old = tf[names.index(e['name'])], tf[names.index(old[k]['name'])]
        if t_old > 0 and t_new >= t_old * (1.0 + a.min_gain):
            old[k] = e
            adopted += 1
            print('adopt  M={:8d} N={:4d} K={:5d} batch {:2d}: {} {:.1f} -> {} {:.1f} TFLOP/s'.format(e['m'], e['n'], e['k'], k[0], shipped_name(shipped, k), t_old, e['name'], t_new))
        else:
            kept += 1
            print('keep   M={:8d} N={:4d} K={:5d} batch {:2d}: {} {:.1f} vs {} {:.1f} TFLOP/s'.format(e['m'], e['n'], e['k'], k[0], old[k]['name'], t_old, e['name'], t_new))
    # one kernel family per layer geometry and image shape over ALL batch sizes (an image's result must not depend on the
    # batch it travels in): the entries of the other batches follow the entry of the largest batch where that one now is
    # of the --prefix family (the only configuration of that family there is; their own measurements no longer apply)
    geo = lambda e: (e['n'], e['k'], e['ntaps'], e['stride'], e['has_res'], round(e['m'] / max(1, int(e.get('batch', 32)))))
    lead = {}
    for e in old.values():
        g = geo(e)
        if g not in lead or int(e.get('batch', 32)) > int(lead[g].get('batch', 32)):
            lead[g] = e
    moved = 0
    for e in old.values():
        top = lead[geo(e)]
        if not a.no_propagate and str(top.get('name', '')).startswith(a.prefix) and e.get('name') != top['name']:
            e['name'], e['cfg'] = top['name'], top['cfg']
            e.pop('ms', None)
            e.pop('tflops', None)
            e['note'] = 'family of the batch-{} entry'.format(top.get('batch', 32))
            moved += 1
    shipped['entries'] = list(old.values())
    json.dump(shipped, open(a.shipped, 'w'), indent=1, sort_keys=True)
    print('{} adopted, {} kept, {} entries of other batch sizes moved to the family of their layer'.format(adopted, kept, moved))


def shipped_name(shipped, k):
    for e in shipped['entries']:
        if (int(e.get('batch', 32)), e['m'], e['n'], e['k'], e['ntaps'], e['stride'], e['has_res']) == k:
            return e.get('name')
    return '?'


if __name__ == '__main__':
    sys.exit(main())
