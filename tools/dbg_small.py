import numpy as np, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import fsae_mpc_amd as fm
H = 2 * np.eye(2); g = np.array([-2., -4.])
for args in ((H, g, [0, 0], [1.5, 1.5]), (H, g, np.array([[1., 1.]]), [0, 0], [1.5, 1.5], [-np.inf], [2.0]),
             (H, g, np.array([[1., 1.],[1.,-1.]]), [0, 0], [1.5, 1.5], [-np.inf,-5], [2.0,5])):
    for mi in (0, 1, 2, 3, 50):
        x, f, fl, it, lam, aux = fm.qpOASES(*args, options=fm.default_opts(max_iter=mi))
        print(len(args), "max_iter", mi, "->", x, f, fl, it)
