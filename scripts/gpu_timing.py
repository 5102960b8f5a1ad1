"""Debug: shader clock and cycles per step inside the fused EKF kernel (needs a -DCRX_EKF_TIMING build in CRX_LIB_PATH)."""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import cpprobotics_amd as crx
from common import *
Q, R = ekf_QR()
lib = crx.lib()
for n, T, hist in [(65536, 1000, True), (65536, 1000, False), (131072, 500, True), (262144, 250, True), (1048576, 100, True)]:
    u, x0, P0 = ekf_agents(n, 1)
    z = torch.randn((T, n, 2), device='cuda') * 0.3; ud = torch.randn((T, n, 2), device='cuda') * 0.1 + 1
    xh = torch.empty((T, n, 4), device='cuda') if hist else None
    for rep in range(3):
        x = torch.from_numpy(x0).cuda(); P = torch.from_numpy(P0).cuda()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); crx.ekf_run(x, P, z, ud, Q, R, x_hist=xh); e1.record(); torch.cuda.synchronize()
    nb = min(4096, n // 64)
    out = np.zeros((nb, 2), dtype=np.int64)
    lib.crx_debug_ekf_timing(out.ctypes.data_as(C.c_void_p), C.c_int(nb))
    ms = e0.elapsed_time(e1)
    clk, real = out[:, 0].astype(float), out[:, 1].astype(float)
    print(f"n={n} T={T} hist={int(hist)}: kernel {ms:.3f} ms; per-wave shader ticks mean {clk.mean():.0f} (min {clk.min():.0f} max {clk.max():.0f}); "
          f"real-time ticks mean {real.mean():.0f} (100 MHz -> {real.mean()/100:.1f} us, max {real.max()/100:.1f} us); "
          f"shader clock = {clk.mean()/real.mean()*100:.0f} MHz; ticks/step/wave = {clk.mean()/T:.0f}")
