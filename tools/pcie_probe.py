"""What the headline would be if the proxy inputs arrived from HOST memory (they do not: the boundary takes device pointers):
pinned host batch -> device copy alone, the copy serialised with the step, and the copy of batch i + 1 on a copy stream under the step of batch i.
    python tools/pcie_probe.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import humaniflow_amd as H
from humaniflow_amd.synthetic import synthetic_proxy_input, init_benchmark_weights

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device('cuda:0')
B, N = 32, 100
cfg = H.get_humaniflow_cfg_defaults()
cfg.MODEL.NUM_RESNET_LAYERS = 50
model = init_benchmark_weights(H.HumaniflowModel(torch.device('cpu'), cfg.MODEL, H.SMPL_PARENTS), seed=0).eval().to(dev)
smpl = H.SMPL(model_data=H.synthetic_smpl_model(seed=0), batch_size=B, create_transl=False).to(dev)
host = [synthetic_proxy_input(B, 256, seed=i).pin_memory() for i in range(3)]
dbuf = [torch.empty_like(h, device=dev) for h in host]
mb = host[0].numel() * 4 / 1e6


def step(x):
    with torch.no_grad():
        out = model(x, num_samples=N)
        return smpl.forward_samples_and_point_est(out)


for i in range(3):
    dbuf[i].copy_(host[i], non_blocking=True)
    step(dbuf[i])
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(steps):
    dbuf[k % 3].copy_(host[k % 3], non_blocking=True)
torch.cuda.synchronize()
t_copy = (time.perf_counter() - t0) / steps
t0 = time.perf_counter()
for k in range(steps):
    step(dbuf[k % 3])
torch.cuda.synchronize()
t_step = (time.perf_counter() - t0) / steps
t0 = time.perf_counter()
for k in range(steps):
    dbuf[k % 3].copy_(host[k % 3], non_blocking=True)
    step(dbuf[k % 3])
torch.cuda.synchronize()
t_serial = (time.perf_counter() - t0) / steps
cs = torch.cuda.Stream()
ev = [torch.cuda.Event() for _ in range(3)]
done = [torch.cuda.Event() for _ in range(3)]
main = torch.cuda.current_stream()
t0 = time.perf_counter()
with torch.cuda.stream(cs):
    dbuf[0].copy_(host[0], non_blocking=True)
    ev[0].record(cs)
for k in range(steps):
    i, j = k % 3, (k + 1) % 3
    with torch.cuda.stream(cs):
        if k >= 2:
            cs.wait_event(done[j])                   # the step that last read buffer j has finished
        dbuf[j].copy_(host[j], non_blocking=True)
        ev[j].record(cs)
    main.wait_event(ev[i])
    step(dbuf[i])
    done[i].record(main)
torch.cuda.synchronize()
t_over = (time.perf_counter() - t0) / steps
print('batch %.0f MB pinned host -> device: %.2f ms (%.1f GB/s) | step alone (eager, one stream) %.2f ms | copy then step %.2f ms = %.0f k images*samples/s | '
      'copy of the next batch under the step %.2f ms = %.0f k images*samples/s'
      % (mb, t_copy * 1e3, mb / 1e3 / t_copy, t_step * 1e3, t_serial * 1e3, B * N / t_serial / 1e3, t_over * 1e3, B * N / t_over / 1e3))
