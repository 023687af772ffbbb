#!/usr/bin/env python3
"""
Rewrites the numeric 'cfg' id of every entry of megadetector_amd/tuned_cfgs*.json from the entry's 'name' in THIS build.

The Python loader (hip_backend.HipContext.load_tuned) resolves entries by name, so stale ids never reached the library
through it; a C-ABI caller that hands the JSON's ids to mdhip_set_tuned would get other configurations after a kernel
family was added or removed (ids are positions in the build's dispatch table, include/mdhip.h).  Run after every change
of the configuration tables (no GPU needed: mdhip_num_conv_cfgs / mdhip_conv_cfg_name are host functions); entries naming
a configuration this build does not have are listed and dropped with --drop-unknown, kept otherwise.
tests/test_cabi_exports.py::test_table_ids_agree_with_names keeps the tables honest.
"""
import argparse
import glob
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--drop-unknown', action='store_true')
    ap.add_argument('--check', action='store_true', help='exit 1 if a file would change')
    args = ap.parse_args()
    from megadetector_amd import _lib
    lib = _lib.load()
    by_name = {lib.mdhip_conv_cfg_name(i).decode(): i for i in range(lib.mdhip_num_conv_cfgs())}
    changed = 0
    for path in sorted(glob.glob(os.path.join(REPO, 'megadetector_amd', 'tuned_cfgs*.json'))):
        doc = json.load(open(path))
        out, moved, unknown = [], 0, []
        for e in doc.get('entries', []):
            name = e.get('name')
            if name and name in by_name:
                if e.get('cfg') != by_name[name]:
                    e = dict(e, cfg=by_name[name])
                    moved += 1
            elif name:
                unknown.append(name)
                if args.drop_unknown:
                    continue
            out.append(e)
        print('{}: {} entries, {} ids rewritten, {} unknown names {}'.format(
            os.path.basename(path), len(out), moved, len(unknown), sorted(set(unknown)) if unknown else ''))
        if moved or len(out) != len(doc.get('entries', [])):
            changed += 1
            if not args.check:
                doc['entries'] = out
                with open(path, 'w') as f:
                    json.dump(doc, f, indent=1, sort_keys=True)
                    f.write('\n')
    return 1 if (args.check and changed) else 0


if __name__ == '__main__':
    sys.exit(main())
