#!/usr/bin/env python3
"""
bench.py reads the counter rows of its line (roofline.traffic, *.pmc) from the newest summaries under profiles/ -- they
cannot be collected from inside the process.  A measurement session runs the bench BEFORE its counter passes, so the line
it saved cites the previous session's summaries.  This re-derives those fields of a saved line from the summaries that
are under profiles/ now (same code path as bench.py).  Usage: python tools/rederive_bench_counters.py profiles/rN_bench.json ...
"""
import glob
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def main():
    for path in sys.argv[1:]:
        d = json.load(open(path))
        r = d['roofline']
        for key in ('dominant_kernel', 'fastest_heavy_kernel'):
            if r.get(key):
                r[key]['pmc'] = bench.pmc_row_for_cfg(r[key]['name'])
        if r.get('traffic') is not None:
            tpaths = sorted(glob.glob(os.path.join(REPO, 'profiles', 'r*_hbm_traffic.json')), reverse=True)
            if tpaths:                                   # (the newest summary; bench.py checks its workload key as well)
                tj = json.load(open(tpaths[0]))
                r['traffic'] = tj.get('hbm_bytes_per_step')
                r['traffic_source'] = 'profiles/' + os.path.basename(tpaths[0]) + r['traffic_source'][r['traffic_source'].index(' '):]
                if r.get('algorithmic_bytes_per_step'):
                    r['traffic_over_algorithmic'] = round(r['traffic'] / r['algorithmic_bytes_per_step'], 3)
        open(path, 'w').write(json.dumps(d) + '\n')
        print(path, 'traffic', r.get('traffic'), 'dominant pmc', (r.get('dominant_kernel') or {}).get('pmc', {}) and
              {k: v for k, v in r['dominant_kernel']['pmc'].items() if k != 'instantiations'})


if __name__ == '__main__':
    main()
