#!/usr/bin/env python3
"""
HBM traffic of one bench.py step from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE are
collected in separate runs: TCC has 4 counter slots, FETCH_SIZE takes 3, WRITE_SIZE 2 --
MI355X_MICROARCH.md 'rocprofv3 PMC slots').

usage: hbm_traffic.py <fetch_pass_dir> <write_pass_dir> <key> <out.json>

Both passes ran `bench.py --lean --steps K --warmup W`, i.e. exactly W+K identical steps; a step has
exactly one letterbox_s2d_kernel (or letterbox_copy_s2d_kernel) dispatch, which is how the steps are counted.  Units and the gfx950
correction follow the guide's HBM section: the counters are in KiB; FETCH_SIZE tallies the 128-byte
requests of a wide coalesced stream at 64 bytes, i.e. reports half of the bytes such streams move, so
it is doubled; WRITE_SIZE is taken as reported (uncalibrated per the guide).
"""
import collections
import csv
import glob
import json
import sys


def load(pass_dir, counter):
    files = glob.glob(pass_dir + '/**/*counter_collection.csv', recursive=True)
    per_kernel = collections.defaultdict(float)
    steps = 0
    seen = set()
    for f in files:
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != counter:
                continue
            name = r['Kernel_Name'].split('(')[0]
            per_kernel[name] += float(r['Counter_Value'])
            if 'letterbox_' in name and 's2d' in name and r['Dispatch_Id'] not in seen:
                seen.add(r['Dispatch_Id'])
                steps += 1
    return per_kernel, steps


def main():
    fdir, wdir, key, out = sys.argv[1:5]
    fetch, steps_f = load(fdir, 'FETCH_SIZE')
    write, steps_w = load(wdir, 'WRITE_SIZE')
    assert steps_f > 0 and steps_f == steps_w, (steps_f, steps_w)
    ours = lambda d: {k: v for k, v in d.items() if 'mdhip::' in k}
    fetch, write = ours(fetch), ours(write)
    fetch_kib = sum(fetch.values()) / steps_f
    write_kib = sum(write.values()) / steps_w
    conv = lambda d: sum(v for k, v in d.items() if 'conv_' in k)
    res = {
        'key': key,
        'steps_counted': steps_f,
        'fetch_size_kib_per_step_raw': fetch_kib,
        'write_size_kib_per_step_raw': write_kib,
        'hbm_bytes_per_step': (2.0 * fetch_kib + write_kib) * 1024.0,
        'conv_kernels_hbm_bytes_per_step': (2.0 * conv(fetch) + conv(write)) / steps_f * 1024.0,
        'correction': 'FETCH_SIZE x2 (gfx950: 128-B requests tallied at 64 B, MI355X_MICROARCH.md HBM section); '
                      'WRITE_SIZE as reported (uncalibrated)',
        'per_kernel_kib_per_step': {k: {'fetch_raw': fetch.get(k, 0.0) / steps_f, 'write_raw': write.get(k, 0.0) / steps_f}
                                    for k in sorted(set(fetch) | set(write))},
    }
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps({k: res[k] for k in ('key', 'steps_counted', 'hbm_bytes_per_step', 'conv_kernels_hbm_bytes_per_step')}))


if __name__ == '__main__':
    main()
