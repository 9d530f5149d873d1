"""Does the row / plane pitch matter?  Jacobi sweep (f64, f32) on N x M x M point grids: ns per cell.
usage: pitch.py [cube]   (cube: N = M, else M = 512)"""
import sys, time, torch
sys.path.insert(0, '.')
import levelsetfortran_amd as lsf
dev = torch.device('cuda', 0)
cube = len(sys.argv) > 1 and sys.argv[1] == 'cube'
for dt in (torch.float64, torch.float32):
    for N in (512, 515, 516, 520, 528, 544, 576):
        M = N if cube else 512
        f = (torch.rand(N * M * M, dtype=torch.float64, device=dev) * 0.1).to(dt)
        dx = 3.0 / 511; h = 0.5 * dx
        lsf.reinit(f, None, None, N - 1, M - 1, M - 1, 3, dx, h, tol=0.0, order='jacobi'); torch.cuda.synchronize()
        t0 = time.perf_counter(); lsf.reinit(f, None, None, N - 1, M - 1, M - 1, 31, dx, h, tol=0.0, order='jacobi'); torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 32 * 1e3
        print(dt, (N, M, M), round(ms, 3), 'ms', round(ms * 1e6 / ((N - 2) * (M - 2) * (M - 2)), 5), 'ns/cell', flush=True)
