import sys; sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import ctypes as C, numpy as np, torch
from open_kinematics_amd import _lib
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.workloads import bump_sweep_problem
prog, t = bump_sweep_problem(16384)
dp = DeviceProgram(prog)
lib = _lib.load()
tg = torch.as_tensor(t, device='cuda'); out = torch.empty((16384, prog.n_out, 3), dtype=torch.float64, device='cuda')
info = torch.empty((16384,40), dtype=torch.uint8, device='cuda'); ph = torch.zeros(12, dtype=torch.int64, device='cuda')
opts = dp.default_opts()
for _ in range(2):
    rc = lib.okx_debug_phase_profile(dp._handle, C.byref(opts), 16384, tg.data_ptr(), out.data_ptr(), info.data_ptr(), ph.data_ptr(), None)
torch.cuda.synchronize()
c = ph.cpu().numpy().astype(float)
names = ['staging','setup','x->pos+derived','rows','reduce+LM logic','normal eq','factorisation','substitutions','output','-','-','-']
tot = c.sum()
for nme, v in zip(names, c): print(f'{nme:14s} {v:12.0f} ticks {100*v/tot:5.1f}%')
inf = info.cpu().numpy().view(np.dtype([('a','<f8'),('b','<f8'),('c','<f8'),('it','<i4'),('nfev','<i4'),('fl','<i4'),('r','<i4')])).reshape(-1)
print('total ticks of workgroup 0:', tot, '| mean LM evaluations per solve', inf['nfev'].mean())
