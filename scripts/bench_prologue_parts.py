#!/usr/bin/env python3
"""Parts of the config-2 prologue timed on their own: python scripts/bench_prologue_parts.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cirkit_amd.parameters import ParamBatch

def timeit(pb, n=50):
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        pb.launch(s)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        pb.launch(s)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

src = torch.randn(784, 32, 256, device="cuda")
dense = torch.randn(784, 32, 32, device="cuda")
dst = torch.empty(784, 257, 32, device="cuda")
scale = torch.empty(784, 257, device="cuda")
ws = [(torch.randn(f, 32, 32, device="cuda"), torch.empty(f, 32, 32, device="cuda")) for f in (392, 196, 98, 49, 24, 11, 6, 4, 2)]
def table(pb, n=784):
    pb.add_log_table_dense(src[:n], dense[:n], None, dst[:n], scale[:n])
def weights(pb):
    for a, b in ws:
        pb.add_softmax(a, b, layout=1)
for name, f in (("table+dense kind 5 (784)", lambda pb: table(pb)), ("weights (782 folds, tiled)", weights),
                ("both", lambda pb: (table(pb), weights(pb))), ("table 392", lambda pb: table(pb, 392)),
                ("table 256", lambda pb: table(pb, 256)), ("table 1", lambda pb: table(pb, 1))):
    pb = ParamBatch(); f(pb)
    print(f"{name}: {timeit(pb):.1f} us")
# reference points: a copy of the same bytes
x = torch.empty(62_000_000 // 8, dtype=torch.float32, device="cuda"); y = torch.empty_like(x)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(5): y.copy_(x)
a.record()
for _ in range(50): y.copy_(x)
b.record(); torch.cuda.synchronize()
print(f"copy of 31 MB (62 MB moved): {a.elapsed_time(b) / 50 * 1e3:.1f} us")
