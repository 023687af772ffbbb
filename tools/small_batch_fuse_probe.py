import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '.')
from megadetector_amd import weights_io, yolo_yaml
from megadetector_amd.hip_backend import HipContext
W = weights_io.synthetic_weights(yolo_yaml.YOLOV5X6_MD, seed=0)
for B in (1, 2, 4):
    S = 1280
    ctx = HipContext(W, device=0, max_batch=B, max_h=S, max_w=S)
    x = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, device='cuda')
    ctx.preprocess([int(x[i].data_ptr()) for i in range(B)], [(S, S, S, S, 0, 0)] * B, S, S)
    def run(tag):
        for _ in range(5): ctx.forward(B, S, S)
        torch.cuda.synchronize(); t = time.time()
        for _ in range(50): ctx.forward(B, S, S)
        torch.cuda.synchronize(); ms = (time.time() - t) / 50 * 1e3
        print('B', B, tag, round(ms, 3), 'ms', round(B / ms * 1e3, 1), 'img/s (forward only)')
    run('table')
    names = {c: ctx.conv_cfg_name(c) for c in range(ctx.num_conv_cfgs())}
    for nm in ('v5:strip160x80/2x5', 'v5:strip128x80/2x5'):
        cfg = [c for c, n in names.items() if n == nm][0]
        ops = [o['op'] for o in ctx.op_infos() if o['kind'] == 0 and ctx.op_supports_cfg(o['op'], cfg)]
        for op in ops: ctx.set_op_cfg(op, cfg)
        run(nm + ' fused')
        ctx.set_fuse(False); run(nm + ' unfused'); ctx.set_fuse(True)
        for op in ops: ctx.set_op_cfg(op, -1)
    ctx.close()
