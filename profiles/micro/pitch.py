"""Does the row pitch matter?  Jacobi sweep (f64, f32) on N x 512 x 512 point grids, N = 512 / 515 / 520 / 528: ns per cell."""
import sys, time, torch
sys.path.insert(0, '.')
import levelsetfortran_amd as lsf
dev = torch.device('cuda', 0)
for dt in (torch.float64, torch.float32):
    for N in (512, 515, 520, 528, 544):
        f = (torch.rand(N * 512 * 512, dtype=torch.float64, device=dev) * 0.1).to(dt)
        dx = 3.0 / 511; h = 0.5 * dx
        lsf.reinit(f, None, None, N - 1, 511, 511, 3, dx, h, tol=0.0, order='jacobi'); torch.cuda.synchronize()
        t0 = time.perf_counter(); lsf.reinit(f, None, None, N - 1, 511, 511, 31, dx, h, tol=0.0, order='jacobi'); torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 32 * 1e3
        print(dt, N, round(ms, 3), 'ms', round(ms * 1e6 / ((N - 2) * 510 * 510), 4), 'ns/cell', flush=True)
