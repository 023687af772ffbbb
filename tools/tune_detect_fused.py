#!/usr/bin/env python3
"""
Tile choice for the Detect 1x1 convs WITH the decode in their epilogue (mdhip_set_option "fuse_decode"): tools/autotune.py
times an op in isolation (mdhip_time_op), where nothing is fused, so the table entries of these four ops were picked for a
conv that only stores logits.  Here every configuration that has a decoding instantiation is forced onto each Detect conv
in turn and conv + decode are timed inside whole forwards (per-op events); the best fused choice is printed next to the
separate launches with the table's own tile, and written into the tile table of the storage type with --adopt when it wins
by at least --min-gain.  GPU box:  python tools/tune_detect_fused.py --dtype bf16 --shape 1280x1280 [--adopt]
"""
import argparse
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--shape', default='1280x1280')
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--adopt', action='store_true')
    ap.add_argument('--min-gain', type=float, default=0.03)
    args = ap.parse_args()
    import parity_util as PU
    from megadetector_amd import weights_io, yolo_yaml
    from megadetector_amd.hip_backend import HipContext
    h, w = (int(v) for v in args.shape.split('x'))
    B = args.batch
    W = weights_io.synthetic_weights(yolo_yaml.YOLOV5X6_MD, seed=0)
    ctx = HipContext(W, device=0, dtype=args.dtype, max_batch=B, max_h=h, max_w=w)
    imgs = PU.random_images(B, h, w, seed=5)
    ctx.preprocess(imgs, [(h, w, h, w, 0, 0)] * B, h, w)
    if args.dtype == 'fp8':
        ctx.calibrate(B, h, w)
    infos = ctx.op_infos()
    convs = [o['op'] for o in infos if o['kind'] == 0 and 'Detect' in o['name']]

    def timed():
        ms = np.zeros(len(infos))
        for _ in range(args.reps):
            ms += ctx.forward_timed(B, h, w)
        return ms / args.reps

    ctx.forward(B, h, w)
    ctx.set_option('fuse_decode', 0)
    base = timed()
    table_cfg = {c: ctx.op_infos()[c]['cfg'] for c in convs}
    ctx.set_option('fuse_decode', 1)
    fused_default = timed()
    out = []
    for c in convs:
        rows = []
        for cfg in range(ctx.num_conv_cfgs()):
            if not ctx.op_supports_cfg(c, cfg):
                continue
            ctx.set_op_cfg(c, cfg)
            ctx.forward(B, h, w)
            if ctx.op_infos()[c + 1]['bytes'] != 0:          # this configuration does not decode in place
                continue
            ms = timed()
            rows.append((float(ms[c] + ms[c + 1]), ctx.conv_cfg_name(cfg)))
        ctx.set_op_cfg(c, -1)
        rows.sort()
        o = ctx.op_infos()[c]
        rec = {'dtype': args.dtype, 'shape': args.shape, 'batch': B, 'op': o['name'], 'm': o['m'], 'n': o['n'], 'k': o['k'],
               'separate_ms': float(base[c] + base[c + 1]), 'separate_cfg': ctx.conv_cfg_name(table_cfg[c]),
               'fused_table_cfg_ms': float(fused_default[c] + fused_default[c + 1]),
               'fused_best_ms': rows[0][0] if rows else None, 'fused_best_cfg': rows[0][1] if rows else None,
               'fused_top3': rows[:3]}
        out.append(rec)
        print(json.dumps(rec))
    ctx.close()
    if args.adopt:
        path = os.path.join(REPO, 'megadetector_amd', 'tuned_cfgs{}.json'.format('' if args.dtype == 'bf16' else '_' + args.dtype))
        doc = json.load(open(path))
        n = 0
        for rec in out:
            if not rec['fused_best_cfg'] or rec['fused_best_ms'] > (1.0 - args.min_gain) * min(rec['separate_ms'], rec['fused_table_cfg_ms']):
                continue
            for e in doc['entries']:
                if (e['m'], e['n'], e['k'], e['ntaps'], e['stride'], e['has_res'], e.get('batch', 32)) == (rec['m'], rec['n'], rec['k'], 1, 1, 0, B):
                    if e['name'] != rec['fused_best_cfg']:
                        e['name'] = rec['fused_best_cfg']
                        e['ms'] = round(rec['fused_best_ms'], 5)
                        e['note'] = 'conv + decode in one launch (tools/tune_detect_fused.py)'
                        n += 1
        json.dump(doc, open(path, 'w'), indent=1, sort_keys=True)
        print('adopted {} entries into {} (run tools/normalize_tables.py)'.format(n, os.path.basename(path)), file=sys.stderr)


if __name__ == '__main__':
    main()
