"""One rank through the RCCL communicator (PAROPT_AMD_FORCE_RCCL=1): collectives issued per iteration with batching."""
import ctypes as C
import os
import sys
import time

os.environ["PAROPT_AMD_FORCE_RCCL"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import paropt_amd as pa
from paropt_amd.lib import check, lib

ctx = pa.Context(0)
buf = (C.c_char * 128)()
check(lib.po_rccl_unique_id(buf))
check(lib.po_ctx_comm_init_rccl(ctx.handle, 0, 1, buf))
prob = pa.SeparableProblem(ctx, "convex", 4_000_000, 32)
prob.setLinearConstraints(True)
ip = pa.InteriorPoint(prob, {"qn_type": "sr1", "qn_subspace_size": 10, "abs_res_tol": 1e-30,
                             "start_affine_multiplier_min": 0.01, "max_major_iters": 40, "write_output_frequency": 0})
k0 = ctx.comm_info()
r0 = ctx.counters()
t0 = time.perf_counter()
ip.optimize()
ctx.synchronize()
dt = time.perf_counter() - t0
k1 = ctx.comm_info()
r1 = ctx.counters()
print("kind %d: %.1f ncclAllReduce + %.1f ncclAllGather per iteration, %.1f host syncs, %.1f launches, %.2f ms/iteration"
      % (k1[0], (k1[1] - k0[1]) / 40.0, (k1[2] - k0[2]) / 40.0, (r1[0] - r0[0]) / 40.0, (r1[1] - r0[1]) / 40.0, dt / 40 * 1e3))
