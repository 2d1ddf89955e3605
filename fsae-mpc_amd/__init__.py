"""fsaempc for MI355X: the batched LTV-MPC QP hot path of kerry-he/fsae-mpc (linearise -> condense ->
solve) as hand-written gfx950 HIP kernels behind a C ABI (include/fsaempc.h, lib/libfsaempc.so).  This
package is the host-side mirror of the reference's interfaces for that path; it contains no CPU compute path."""
from . import _lib
from ._lib import FsaempcError, default_opts, lib
from .closed_loop import ClosedLoop, monte_carlo, monte_carlo_carts
from .ltvmpc import LtvBatch, dims, ltvmpc_dynamic_curvilinear, ltvmpc_kinetmatic_curvilinear
from .qpoases import qp_solve_batch_device, qpOASES, qpOASES_sequence
from .reference import obtain_reference, obtain_reference_batch_device, reference_live_batch_device
from .synthetic import DYNAMIC, KINEMATIC, instances, reference_live
from .tracks import Track

__all__ = ["FsaempcError", "default_opts", "lib", "LtvBatch", "dims", "ltvmpc_dynamic_curvilinear",
           "ltvmpc_kinetmatic_curvilinear", "qp_solve_batch_device", "qpOASES", "qpOASES_sequence", "DYNAMIC", "KINEMATIC",
           "instances", "reference_live", "Track", "obtain_reference", "obtain_reference_batch_device",
           "reference_live_batch_device", "ClosedLoop", "monte_carlo", "monte_carlo_carts"]
