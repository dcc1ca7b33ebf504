#!/usr/bin/env python3
"""Tucker layers of the reference's compilation-options notebook (K = 64, batch 128) one by one through the C ABI:
the one-workgroup-per-tile launch against the stream-K launch (workspace lent), on normalised weights and on raw logits
normalised online.  python scripts/bench_tucker_layers.py [B] [K]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cirkit_amd import _capi as capi  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
K = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
stream = torch.cuda.current_stream(dev).cuda_stream
for F, Ko in ((784, K), (392, K), (196, K), (98, K), (42, K), (22, K), (12, K), (8, K), (4, K), (2, 1)):
    g = torch.Generator().manual_seed(F)
    theta = torch.randn(F, Ko, K * K, generator=g).to(dev)
    w = torch.softmax(theta, dim=-1)
    x = (torch.randn(F, 2, B, K, generator=g) * 3 - 4).to(dev)
    row_off = (torch.arange(F * 2, dtype=torch.int64) * (B * K)).reshape(F, 2).to(dev)
    out = torch.empty(F, B, Ko, device=dev)
    tiles = F * ((Ko + 31) // 32) * ((B + 127) // 128)
    ws = torch.zeros(n_cu * 3 * 2 * (4 * 1024 + 64) + tiles, dtype=torch.int32, device=dev)
    res = {}
    for name, use in (("tile", False), ("streamk", True), ("online", True)):
        capi.call("ck_set_workspace", ws.data_ptr() if use else None, ws.numel() * 4 if use else 0)
        def go():
            if name == "online":
                capi.call("ck_tucker_logits_fwd", x.data_ptr(), row_off.data_ptr(), theta.data_ptr(), out.data_ptr(), F, B, K, Ko, stream)
            else:
                capi.call("ck_sum_lse_fwd", x.data_ptr(), row_off.data_ptr(), w.data_ptr(), out.data_ptr(), F, 2, B, K, Ko,
                          capi.CK_SUM_KRON, capi.CK_W_ROWMAJOR, stream)
        for _ in range(3):
            go()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            go()
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 20 * 1e3
        capi.call("ck_set_workspace", None, 0)
    ideal = F * ((Ko + 31) // 32) * ((B + 31) // 32) * K * (K // 2) * 64 / (n_cu * 4 * 2.4e9) * 1e6
    print(f"F={F:4d} Ko={Ko:3d}: tile {res['tile']:7.1f} us   stream-K {res['streamk']:7.1f} us   on logits {res['online']:7.1f} us   MFMA floor {ideal:6.1f} us", flush=True)
