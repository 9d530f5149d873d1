"""Cost of the FIRST exact-ordering call of a process at 512^3 (work-field allocation, tables, task list) against a warm one."""
import sys, time, torch
sys.path.insert(0, '.')
import levelsetfortran_amd as lsf
from levelsetfortran_amd import fields
dev = torch.device('cuda', 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
phi, dx = fields.two_sphere_phi0_device((N, N, N), dev)
h = fields.reinit_step(dx)
torch.cuda.synchronize()
for label, sweeps in (("cold, 1 sweep", 1), ("warm, 1 sweep", 1), ("warm, 128 sweeps (new sweep count)", 128), ("warm, 128 sweeps again", 128), ("warm, 1 sweep", 1)):
    t0 = time.perf_counter()
    lsf.reinit(phi, None, None, N - 1, N - 1, N - 1, sweeps - 1, dx, h, tol=0.0)
    torch.cuda.synchronize()
    print(f"{label}: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
