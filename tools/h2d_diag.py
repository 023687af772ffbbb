import sys, time, torch
sys.path.insert(0, '.')
from megadetector_amd import weights_io, yolo_yaml
from megadetector_amd.hip_backend import HipContext
B, S = 32, 1280
W = weights_io.synthetic_weights(yolo_yaml.YOLOV5X6_MD, seed=0)
ctx = HipContext(W, device=0, max_batch=B, max_h=S, max_w=S)
host = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8).pin_memory()
dev = torch.empty((B, S, S, 3), dtype=torch.uint8, device='cuda')
ptrs = [int(dev[i].data_ptr()) for i in range(B)]
geoms = [(S, S, S, S, 0, 0)] * B
cs, ks = torch.cuda.Stream(), torch.cuda.Stream()
def t(fn, n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def copy():
    with torch.cuda.stream(cs): dev.copy_(host, non_blocking=True)
def fwd():
    ctx.preprocess(ptrs, geoms, S, S, stream=ks.cuda_stream); ctx.forward(B, S, S, stream=ks.cuda_stream)
fwd(); copy()
print('copy alone ms', t(copy), 'GB/s', host.numel() / t(copy) / 1e6)
print('forward alone ms', t(fwd))
def both():
    copy(); fwd()
print('copy || forward ms', t(both))
ctx.close()

# the bench pipeline with host-fed inputs, with and without the host-side formatting
from megadetector_amd.postprocess import format_detections
ctx = HipContext(W, device=0, max_batch=B, max_h=S, max_w=S)
dev2 = [dev, torch.empty_like(dev)]
dptr = [[int(d[i].data_ptr()) for i in range(B)] for d in dev2]
copied = [torch.cuda.Event() for _ in range(2)]
consumed = [torch.cuda.Event() for _ in range(2)]
def enqueue(i):
    k = i % 2
    with torch.cuda.stream(cs):
        if i >= 2: cs.wait_event(consumed[k])
        dev2[k].copy_(host, non_blocking=True)
        copied[k].record(cs)
    ks.wait_event(copied[k])
    ctx.preprocess(dptr[k], geoms, S, S, stream=ks.cuda_stream)
    consumed[k].record(ks)
    ctx.forward(B, S, S, stream=ks.cuda_stream)
    ctx.nms_enqueue(B, 1e-5, 0.45, 300, slot=i % 4, stream=ks.cuda_stream)
def run(n, fmt):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        enqueue(i)
        if i > 0:
            det, cnt = ctx.nms_wait(slot=(i - 1) % 4)
            if fmt:
                for b in range(B): format_detections(det[b, :cnt[b]], (S, S), (S, S, 3), (S, S, 3), 1e-5)
    det, cnt = ctx.nms_wait(slot=(n - 1) % 4)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
run(3, False)
print('pipeline host-fed, no formatting: ms/step', run(10, False))
print('pipeline host-fed, with formatting: ms/step', run(10, True))
ctx.close()

ctx = HipContext(W, device=0, max_batch=B, max_h=S, max_w=S)
run(3, True)
print('again, no forward events: ms/step', run(20, True))
ctx.time_forwards(True)
print('with forward events: ms/step', run(20, True), 'fwd', float(ctx.forward_times(20).mean()))
ctx.time_forwards(False)
hosts = [torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8).pin_memory() for _ in range(4)]
_enq = enqueue
def enqueue(i):
    global host
    host = hosts[i % 4]
    _enq(i)
print('4 different batches: ms/step', run(20, True))
ctx.close()
