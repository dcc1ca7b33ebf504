#!/usr/bin/env python3
"""Microbenchmark of the batched softmax prologue jobs:  python scripts/bench_prologue.py
table jobs (kind 1) at K = 32 / 64 / 128 and row jobs (kind 0) at several row lengths."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cirkit_amd.parameters import ParamBatch  # noqa: E402


def timeit(pb):
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        pb.launch(s)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        pb.launch(s)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / 20


for K in (32, 64, 128):
    src = torch.randn(784, K, 256, device="cuda")
    dst = torch.empty(784, 257, K, device="cuda")
    pb = ParamBatch()
    pb.add_log_table(src, dst)
    ms = timeit(pb)
    print(f"table K={K}: {ms * 1e3:.1f} us  {2 * src.numel() * 4 / ms / 1e6:.0f} GB/s")
for rows, ln in ((784 * 32, 32), (784 * 64, 64), (1500 * 64, 64), (784 * 128, 128), (300 * 256, 256), (100 * 64, 4096)):
    src = torch.randn(rows, ln, device="cuda")
    dst = torch.empty_like(src)
    pb = ParamBatch()
    pb.add_softmax(src, dst) if hasattr(pb, "add_softmax") else pb._jobs.append((src.data_ptr(), dst.data_ptr(), rows, ln, 0, 0, None, None, None))
    ms = timeit(pb)
    print(f"rows {rows} x {ln}: {ms * 1e3:.1f} us  {2 * src.numel() * 4 / ms / 1e6:.0f} GB/s")
for K, Fd in ((32, 784), (32, 1568), (64, 784), (64, 1568)):
    src = torch.randn(784, K, 256, device="cuda")
    dense = torch.randn(Fd, K, K, device="cuda")
    idx = (torch.arange(Fd, device="cuda") % 784).to(torch.int64)
    dst = torch.empty(Fd, 257, K, device="cuda")
    pb = ParamBatch()
    pb.add_log_table_dense(src, dense, idx, dst)
    ms = timeit(pb)
    print(f"table+dense K={K} Fd={Fd}: {ms * 1e3:.1f} us  {(src.numel() * Fd / 784 + dst.numel()) * 4 / ms / 1e6:.0f} GB/s")
