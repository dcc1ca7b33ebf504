import sys, torch
sys.path.insert(0, "/root/repo")
from cirkit_amd import _capi as capi
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream(dev).cuda_stream
for K, F, S, B in ((64, 196, 2, 4096), (64, 49, 2, 4096), (32, 196, 2, 4096), (64, 196, 2, 512)):
    g = torch.Generator().manual_seed(1)
    arena = (torch.randn(F * S, B, K, generator=g) * 3 - 5).to(dev)
    row_off = (torch.arange(F * S, dtype=torch.int64) * (B * K)).reshape(F, S).to(dev)
    w = torch.softmax(torch.randn(F * S, K, K, generator=g), dim=-1).to(dev)
    addr = torch.tensor([w.data_ptr() + i * K * K * 4 for i in range(F * S)], dtype=torch.int64).reshape(F, S).to(dev)
    out = torch.empty(F, B, K, device=dev)
    res = {}
    for force in (0, 1):
        capi.call("ck_debug_force_generic", force)
        def go():
            capi.call("ck_cp_lse_fwd", arena.data_ptr(), row_off.data_ptr(), addr.data_ptr(), None, None, out.data_ptr(),
                      None, None, None, 0, F, S, 1, B, K, stream)
        for _ in range(3): go()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): go()
        e1.record(); torch.cuda.synchronize()
        res[force] = e0.elapsed_time(e1) / 20 * 1e3
    capi.call("ck_debug_force_generic", 0)
    floor = F * (B / 32) * S * (K // 32) ** 2 * 16 * 64 / (256 * 4 * 2.4e9) * 1e6
    print(f"K={K} F={F} S={S} B={B}: dma {res[0]:.1f} us  register path {res[1]:.1f} us  MFMA floor {floor:.1f} us")
