import ctypes as C, os, sys, numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import torch, fsae_mpc_amd as fm
tr = fm.Track.load("fsg2019")
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for model, N, B in ((0, 40, 4096), (1, 60, 4096)):
    x0, xl, ul, xr = fm.instances(model, N, 0.05, tr.L, 20190, range(B))
    q = fm.LtvBatch(model, N, 0.05, tr, B).build_qp(dev(x0), dev(xr), dev(xl), dev(ul))
    dump = torch.zeros(B * 8, dtype=torch.float64, device="cuda")
    fm.lib().fsaempc_debug_set_dump(C.c_void_p(dump.data_ptr()), 8)
    fm.qp_solve_batch_device(*(q[k] for k in ("H", "g", "A", "lb", "ub", "lbA", "ubA")))
    torch.cuda.synchronize(); fm.lib().fsaempc_debug_set_dump(None, 0)
    d = dump.cpu().numpy().reshape(B, 8)
    dd = np.diff(d, axis=1)
    print("model", model, "N", N, "prep phases (cycles, median over WGs): col-scale %d | row pass %d | sort %d | trips %d | rows/bounds %d | A repack %d | H repack %d | total %d" % (*np.median(dd, axis=0), np.median(d[:, 7] - d[:, 0])))
