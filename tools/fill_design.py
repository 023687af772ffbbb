#!/usr/bin/env python3
"""Fills the «TOKEN» fields of DESIGN.md section 6 from the files of a final session under profiles/ (tools/collect_final.sh),
so that the document's figures ARE the committed files'.  usage: python tools/fill_design.py [round] ; docs/DESIGN.md.in -> DESIGN.md"""
import csv
import json
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else '6'
P = os.path.join(REPO, 'profiles', 'r' + R + '_')


def j(name):
    return json.load(open(P + name))


def jl(name):
    return [json.loads(l) for l in open(P + name) if l.startswith('{')]


def main():
    b = j('bench.json')
    r = b['roofline']
    d = r['dominant_kernel']
    c = b['cpu_baseline']
    t = j('hbm_traffic.json')
    kstats = {}
    with open(P + 'bench_kernel_stats.csv') as f:
        for row in csv.reader(f):
            if len(row) > 3 and 'conv_v5_kernel<320, 160, 4, 2, 0, ' in row[0] and row[0].rstrip(')').endswith('true>(mdhip::ConvArgs'):
                kstats[row[0].split('<')[1].split('>')[0]] = float(row[3]) / 1e3
    st = r['stages']
    dec = st.get('decode')
    small = jl('bench_small_batches.jsonl')
    f16real = jl('bench_fp16_real_shapes.jsonl')      # 1080p, 4:3, 3:2
    nms = open(P + 'bench_nms_stream.txt').read()
    ni, no = re.findall(r': ([0-9.]+) images/s', nms)
    gt = open(P + 'pytest_gpu.log').read()
    m = re.search(r'(\d+ passed[^\n]*) in ', gt)
    f8 = j('bench_fp8_b64.json')
    tok = {
        'H_IMG': '{:,.1f}'.format(b['value']).replace(',', ' '), 'H_MS': '{:.2f}'.format(b['ms_per_step']),
        'H_FWD': '{:.2f}'.format(r['kernel_ms_per_step']), 'H_TF': '{:.0f}'.format(r['achieved']), 'H_FRAC': '{:.3f}'.format(r['frac']),
        'D_MS': '{:.2f}'.format(d['ms_per_step']), 'D_US': '{:.0f}'.format(d['avg_launch_us']),
        'D_TF': '{:,.0f}'.format(d['achieved_tflops']).replace(',', ' '), 'D_FRAC': '{:.3f}'.format(d['frac']),
        'K0': '{:.1f}'.format(kstats.get('320, 160, 4, 2, 0, 0, true', float('nan'))),
        'K2': '{:.1f}'.format(kstats.get('320, 160, 4, 2, 0, 2, true', float('nan'))),
        'T_GB': '{:.1f}'.format(t['hbm_bytes_per_step'] / 1e9), 'T_X': '{:.2f}'.format(t['hbm_bytes_per_step'] / 63.96e9),
        'P_MS': '{:.3f}'.format(st['preprocess']['ms']), 'P_FRAC': '{:.2f}'.format(st['preprocess']['frac']),
        'DEC': 'n/a' if not dec else '{:.2f} ms = {:.1f} TB/s = {:.2f}'.format(dec['ms'], dec['achieved'] / 1e3, dec['frac']),
        'N_MS': '{:.2f}'.format(st['nms']['ms']),
        'C_IMG': '{:.2f}'.format(c['value']), 'C_1': '{:.2f}'.format(c['single_thread']['value']),
        'C_F': '{:.2f}'.format(c['forward_s_per_image']), 'C_N': '{:.2f}'.format(c['nms_format_s_per_image']),
        'F16': '{:,.1f}'.format(j('bench_fp16.json')['value']).replace(',', ' '),
        'F8': '{:,.1f}'.format(f8['value']).replace(',', ' '), 'F8_FRAC': '{:.3f}'.format(f8['roofline']['frac']),
        'R43': '{:,.0f}'.format(j('bench_real_1536x2048.json')['value']).replace(',', ' '),
        'R32': '{:,.0f}'.format(j('bench_real_1600x2400.json')['value']).replace(',', ' '),
        'RV': '{:,.0f}'.format(j('bench_video_1080p.json')['value']).replace(',', ' '),
        'RVH': '{:,.0f}'.format(f16real[0]['value']).replace(',', ' '), 'R43H': '{:,.0f}'.format(f16real[1]['value']).replace(',', ' '),
        'R32H': '{:,.0f}'.format(f16real[2]['value']).replace(',', ' '),
        'HF': '{:,.1f}'.format(j('bench_hostfed.json')['value']).replace(',', ' '),
        'SB': ' / '.join('{:.0f}'.format(x['value']) for x in small),
        'NI': ni, 'NO': no,
        'GT': m.group(1) if m else '?',
        'CT': os.environ.get('CPU_SUITE', '?'),
    }
    src = open(os.path.join(REPO, 'docs', 'DESIGN.md.in')).read()
    missing = sorted(set(re.findall(r'«([A-Z0-9_]+)»', src)) - set(tok))
    if missing:
        sys.exit('no value for ' + ', '.join(missing))
    out = re.sub(r'«([A-Z0-9_]+)»', lambda mm: tok[mm.group(1)], src)
    open(os.path.join(REPO, 'DESIGN.md'), 'w').write(out)
    print('DESIGN.md written ({} tokens)'.format(len(tok)))


if __name__ == '__main__':
    main()
