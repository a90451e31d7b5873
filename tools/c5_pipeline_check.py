#!/usr/bin/env python3
# the chunked pipeline on one rank's data against the unchunked solve: bit for bit (world 1 has no exchange: use the class directly)
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_kinematics_amd.batch import DeviceProgram
from open_kinematics_amd.dist import ShardedEnsemble
from open_kinematics_amd.workloads import ensemble_problem
program, table, rel = ensemble_problem(67, 64)
dp = DeviceProgram(program, "cuda:0")
t = torch.as_tensor(table, device="cuda:0")
a = ShardedEnsemble(dp, t, rel, 64, chunks=1, direct=False, chain_len=1, predictor=False)
pa = a.step().clone(); fa = a.free_full.clone(); ia = a.info_full.clone()
for c in (3, 8, 67, 100):
    b = ShardedEnsemble(dp, t, rel, 64, chunks=c, direct=False, chain_len=1, predictor=False)
    pb = b.step(); torch.cuda.synchronize()
    assert torch.equal(pa, pb) and torch.equal(fa, b.free_full) and torch.equal(ia, b.info_full), c
    f = ShardedEnsemble(dp, t, rel, 64, chunks=c, records=False, chain_len=1, predictor=False)
    assert torch.equal(f.step(), fa)
    lean = ShardedEnsemble(dp, t, rel, 64, chunks=c, records=False, info="status", chain_len=1, predictor=False)
    assert torch.equal(lean.step(), fa) and torch.equal(lean.status_full, ia[:, 32]) and torch.equal(lean.info_local, ia)
ref = dp.solve(dp.ensemble_targets(*[dp.rebind(t)[0]], rel), geom_pos=dp.rebind(t)[0], geom_row_param=dp.rebind(t)[1], steps_per_geometry=64, chain_len=1, predictor=False)
assert torch.equal(ref.positions, pa)
d = ShardedEnsemble(dp, t, rel, 64, chain_len=1, predictor=False)
assert d.direct and torch.equal(d.step(), pa) and torch.equal(d.info_full, ia)
print("chunked == unchunked == plain solve, bit for bit")
