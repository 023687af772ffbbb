#!/usr/bin/env python3
"""
HBM traffic of one bench.py step from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE are
collected in separate runs: TCC has 4 counter slots, FETCH_SIZE takes 3, WRITE_SIZE 2 --
MI355X_MICROARCH.md 'rocprofv3 PMC slots').

usage: hbm_traffic.py <fetch_pass_dir> <write_pass_dir> <key> <out.json> [<ops.json> [<table.txt>]]

With <ops.json> (bench.py --profile-out of the same build: one row per op of the forward, the library's own accounting of
every op's algorithmic bytes, mdhip_get_op_info) the dispatches of a step are aligned with the ops in launch order and the
output gains `per_op` and `per_instantiation`: counter bytes next to ALGORITHMIC read / write bytes and their ratio --
read amplification per kernel instantiation (halo re-reads across tiles and XCDs, operands that miss L2) instead of one
figure for the whole step.

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


FORWARD_KERNELS = ('conv_', 'sppf_pool', 'upsample2x', 'detect_decode', 'copy_view')


def load_sequences(pass_dir, counter):
    """per step: the forward's dispatches in launch order as (kernel name, counter value)"""
    rows = []
    for f in glob.glob(pass_dir + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                rows.append((int(r['Dispatch_Id']), r['Kernel_Name'].split('(')[0], float(r['Counter_Value'])))
    rows.sort()
    steps, cur = [], None
    for _, name, v in rows:
        if 'letterbox_' in name and 's2d' in name:
            cur = []
            steps.append(cur)
        elif cur is not None and 'mdhip::' in name and any(k in name for k in FORWARD_KERNELS):
            cur.append((name, v))
    return steps


def op_algorithmic(op, prev):
    """(read, write) algorithmic bytes of one op from the library's accounting (`bytes` = reads + writes of the op as it
    is launched: fused bottlenecks and upsamples read in place are already taken out)"""
    b = float(op['bytes'])
    if op['kind'] == 0:
        out_bytes = 4.0 if (op['n'] <= 32 and op['ntaps'] == 1 and 'Detect' in op['name']) else 2.0
        w = float(op['m']) * op['n'] * out_bytes
    elif op['kind'] == 1:
        w = b * 3.0 / 4.0            # SPPF pools: one slice read, three written
    elif op['kind'] == 2:
        w = b * 4.0 / 5.0            # nearest x2: one pixel read, four written
    else:
        w = b / 2.0                  # decode, copies
    return max(b - w, 0.0), w


def per_op_table(fdir, wdir, ops_path):
    ops = json.load(open(ops_path))
    launching = [o for o in ops if float(o.get('bytes', 0)) > 0 and not (o['kind'] == 0 and o['cfg'] < 0)]
    fseq, wseq = load_sequences(fdir, 'FETCH_SIZE'), load_sequences(wdir, 'WRITE_SIZE')
    fseq = [s for s in fseq if len(s) == len(launching)]
    wseq = [s for s in wseq if len(s) == len(launching)]
    if not fseq or not wseq:
        return None, 'no step of the counter passes has {} forward dispatches (fetch pass: {}, write pass: {})'.format(
            len(launching), sorted({len(s) for s in load_sequences(fdir, "FETCH_SIZE")}),
            sorted({len(s) for s in load_sequences(wdir, "WRITE_SIZE")}))
    per_op, inst = [], collections.OrderedDict()
    for i, o in enumerate(launching):
        names = {s[i][0] for s in fseq} | {s[i][0] for s in wseq}
        assert len(names) == 1, (o['name'], names)                 # the same kernel in every step and in both passes
        name = names.pop()
        assert ('conv_' in name) == (o['kind'] == 0), (o['name'], name)
        rd = 2.0 * 1024.0 * sum(s[i][1] for s in fseq) / len(fseq)
        wr = 1024.0 * sum(s[i][1] for s in wseq) / len(wseq)
        ar, aw = op_algorithmic(o, None)
        row = {'op': o['op'], 'name': o['name'], 'kernel': name, 'm': o['m'], 'n': o['n'], 'k': o['k'],
               'read': rd, 'read_algorithmic': ar, 'write': wr, 'write_algorithmic': aw,
               'read_ratio': rd / ar if ar > 0 else None, 'write_ratio': wr / aw if aw > 0 else None}
        per_op.append(row)
        e = inst.setdefault(name, {'launches': 0, 'read': 0.0, 'read_algorithmic': 0.0, 'write': 0.0, 'write_algorithmic': 0.0})
        e['launches'] += 1
        for k in ('read', 'read_algorithmic', 'write', 'write_algorithmic'):
            e[k] += row[k]
    for e in inst.values():
        e['read_ratio'] = e['read'] / e['read_algorithmic'] if e['read_algorithmic'] > 0 else None
        e['write_ratio'] = e['write'] / e['write_algorithmic'] if e['write_algorithmic'] > 0 else None
    return {'per_op': per_op, 'per_instantiation': inst, 'steps_aligned': [len(fseq), len(wseq)]}, None


def format_table(tab):
    out = ['# HBM bytes per step by kernel instantiation: rocprofv3 counters (FETCH_SIZE x 2, WRITE_SIZE; two separate passes) next to the',
           '# ALGORITHMIC bytes of the ops each instantiation ran (mdhip_get_op_info: input read once, weights, residual; output written once)',
           '{:>4s} {:>9s} {:>9s} {:>6s} {:>9s} {:>9s} {:>6s}  {}'.format('n', 'read GB', 'alg GB', 'ratio', 'write GB', 'alg GB', 'ratio', 'kernel')]
    tot = [0.0] * 4
    for name, e in sorted(tab['per_instantiation'].items(), key=lambda kv: -(kv[1]['read'] - kv[1]['read_algorithmic'])):
        out.append('{:4d} {:9.3f} {:9.3f} {:>6s} {:9.3f} {:9.3f} {:>6s}  {}'.format(
            e['launches'], e['read'] / 1e9, e['read_algorithmic'] / 1e9, '{:.2f}'.format(e['read_ratio']) if e['read_ratio'] else '-',
            e['write'] / 1e9, e['write_algorithmic'] / 1e9, '{:.2f}'.format(e['write_ratio']) if e['write_ratio'] else '-',
            name.replace('void ', '').replace('mdhip::st_bf16::', '').replace('mdhip::', '')))
        for j, k in enumerate(('read', 'read_algorithmic', 'write', 'write_algorithmic')):
            tot[j] += e[k]
    out.append('{:>4s} {:9.3f} {:9.3f} {:6.2f} {:9.3f} {:9.3f} {:6.2f}  all forward kernels'.format(
        '', tot[0] / 1e9, tot[1] / 1e9, tot[0] / tot[1], tot[2] / 1e9, tot[3] / 1e9, tot[2] / tot[3]))
    out.append('# the ten ops with the largest excess reads')
    for r in sorted(tab['per_op'], key=lambda r: -(r['read'] - r['read_algorithmic']))[:10]:
        out.append('  op {:3d} {:28s} M {:8d} N {:4d} K {:5d}: read {:.3f} GB for {:.3f} algorithmic ({:.2f}x)'.format(
            r['op'], r['name'], r['m'], r['n'], r['k'], r['read'] / 1e9, r['read_algorithmic'] / 1e9, r['read_ratio'] or 0.0))
    return '\n'.join(out) + '\n'


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
    if len(sys.argv) > 5:
        tab, why = per_op_table(fdir, wdir, sys.argv[5])
        if tab is None:
            res['per_op_error'] = why
        else:
            res.update(tab)
            res['forward_kernels_read_over_algorithmic'] = sum(r['read'] for r in tab['per_op']) / sum(r['read_algorithmic'] for r in tab['per_op'])
            if len(sys.argv) > 6:
                open(sys.argv[6], 'w').write(format_table(tab))
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps({k: res[k] for k in ('key', 'steps_counted', 'hbm_bytes_per_step', 'conv_kernels_hbm_bytes_per_step')}))


if __name__ == '__main__':
    main()
