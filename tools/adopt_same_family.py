#!/usr/bin/env python3
"""
Round 5: adopts re-tuned batch-B entries into a shipped tile table where the new configuration is of the SAME summation-order
family as the shipped one (implicit GEMM / conv_v2 <-> each other; conv_v5 tiles <-> each other: bit-identical results, so the
entries of the other batch sizes keep their own measured configurations) and wins by a margin IN THE SAME autotune run:
    python tools/adopt_same_family.py <shipped.json> <retuned.json> <autotune table .txt> --names <names.json> [--min-gain 0.04]
The 1x1 convs that read an upsampled tensor in place (C3 entry convs of layers 15 / 19 / 23) are left alone: mdhip_time_op times
an op without that absorption, and any configuration but conv_v2's 160x160 brings the upsample launch back (round 4: + 0.25 ms).
Family changes (conv_v7) go through tools/adopt_entries.py, which moves the entries of the other batch sizes with them.
"""
import argparse
import json
import re
import sys


def family(name):
    for p in ('v5:', 'v7:', 'f8:'):
        if name.startswith(p):
            return p
    return 'gemm'                      # conv_igemm / conv_v2 / stem: one K order


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('shipped')
    ap.add_argument('retuned')
    ap.add_argument('table')
    ap.add_argument('--names', required=True)
    ap.add_argument('--min-gain', type=float, default=0.04)
    a = ap.parse_args()
    names = json.load(open(a.names))
    key = lambda e: (int(e.get('batch', 32)), e['m'], e['n'], e['k'], e['ntaps'], e['stride'], e['has_res'])
    shipped = json.load(open(a.shipped))
    old = {key(e): e for e in shipped['entries']}
    rows = {}
    for line in open(a.table):
        m = re.match(r'(.{34}) M=\s*(\d+) N=\s*(\d+) K=\s*(\d+) .*?\| (.*)$', line)
        if m:
            stride = 2 if re.search(r'conv \dx\ds2', m.group(1)) else 1
            k = (int(m.group(2)), int(m.group(3)), int(m.group(4)), stride)
            rows.setdefault(k, []).append((m.group(1).strip(), [float(v) for v in m.group(5).split()]))
    adopted = kept = 0
    for e in json.load(open(a.retuned))['entries']:
        k = key(e)
        if k not in old or old[k].get('name') == e.get('name'):
            continue
        r = rows.get((e['m'], e['n'], e['k'], e['stride']))
        if not r or old[k].get('name') not in names or e.get('name') not in names:
            continue
        if any(re.match(r'L(15|19|23) C3\.cv1\|cv2', nm) for nm, _ in r):
            print('skip   {} (reads an upsampled tensor in place)'.format([nm for nm, _ in r if 'cv1|cv2' in nm][0]))
            continue
        if old[k]['name'].startswith('v5:strip'):
            # the bottleneck 3x3s of an 80-channel block run as fused launches (1x1 + 3x3): an isolated op timing prices the
            # plain strip kernel, not what runs (tools/c80_ab.py A/Bs those inside whole forwards)
            print('skip   {} (fused bottleneck launch)'.format(r[0][0]))
            continue
        if family(old[k]['name']) != family(e['name']):
            continue
        # (ops with and without residual share M, N, K: the row whose best figure is the re-tuned entry's own)
        best = [row for row in r if abs(max(row[1]) - float(e.get('tflops', -1))) < 0.06]
        tf = (best[0] if best else r[0])[1]
        t_new, t_old = tf[names.index(e['name'])], tf[names.index(old[k]['name'])]
        if t_old > 0 and t_new >= t_old * (1.0 + a.min_gain):
            print('adopt  {:30s} M={:8d} N={:4d} K={:5d} batch {:2d}: {} {:.1f} -> {} {:.1f} TFLOP/s (+{:.1f} %)'.format(
                r[0][0], e['m'], e['n'], e['k'], k[0], old[k]['name'], t_old, e['name'], t_new, (t_new / t_old - 1) * 100))
            old[k] = e
            adopted += 1
        else:
            kept += 1
    shipped['entries'] = list(old.values())
    json.dump(shipped, open(a.shipped, 'w'), indent=1, sort_keys=True)
    print('{} adopted, {} kept'.format(adopted, kept))


if __name__ == '__main__':
    sys.exit(main())
