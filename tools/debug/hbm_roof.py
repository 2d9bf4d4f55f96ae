"""What this box's HBM delivers to plain streaming kernels (torch elementwise ops, 512 MB operands): the practical ceiling for the
training step's activation traffic.   python tools/debug/hbm_roof.py"""
import json, torch
n = 128 * 1024 * 1024
x = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
y = torch.empty_like(x)
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
B = 4.0 * n
out = {"copy_read+write_TBps": 2 * B / t(lambda: y.copy_(x)) / 1e12,
       "fill_write_TBps": B / t(lambda: y.fill_(1.0)) / 1e12,
       "sum_read_TBps": B / t(lambda: x.sum()) / 1e12,
       "add_inplace_read+write_TBps": 2 * B / t(lambda: x.add_(1.0)) / 1e12}
print(json.dumps({k: round(v, 2) for k, v in out.items()}))
