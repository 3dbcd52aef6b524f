import sys, torch, statistics
sys.path.insert(0, "/root/repo")
from omnivggt_official_amd import ops
dt = torch.bfloat16; DEV = "cuda"
g = torch.Generator().manual_seed(0)
BH, n = 16, 64 * 1374
q, k, vt = ops.alloc_qkv(BH, 90112, n, dt, DEV)
q[:, :90112] = (torch.randn(BH, 90112, 64, generator=g) * 1.3).to(dt).to(DEV)
k[:, :n] = torch.randn(BH, n, 64, generator=g).to(dt).to(DEV)
ops.set_vt(vt, torch.randn(BH, 64, n, generator=g).to(dt))
def timed(nq, variant, iters=6):
    o = torch.empty(nq, 1024, device=DEV, dtype=dt)
    f = lambda: ops.flash_attn(q, [(k, vt, n)], nq, dt, out=o, variant=variant)
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters)
    return statistics.median(ts)
for variant in (57, 50):
    base = None
    for nq in (81920, 87936, 90112):     # 160 / 171.75 / 176 tiles of 512 rows per head = 10.0 / 10.75 / 11.0 rounds of 256 workgroups
        ms = timed(nq, variant)
        rounds512 = 16 * ((nq + 511) // 512) / 256
        print("variant %d nq=%d (%.2f rounds of 512-row tiles): %.3f ms  -> %.4f ms per 1000 q rows" % (variant, nq, rounds512, ms, ms / nq * 1000), flush=True)
