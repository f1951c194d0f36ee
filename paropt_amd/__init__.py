"""paropt_amd -- MI355X-native interior-point hot path behind ParOpt's API.

Importing the package loads libparopt_amd.so; it raises if the library has not been built.
"""
from .api import (  # noqa: F401
    LBFGS,
    LSR1,
    MMA,
    Context,
    InteriorPoint,
    Problem,
    PVec,
    SeparableProblem,
    TrustRegion,
    TrustRegionSubproblem,
    QuadraticSubproblem,
    UserTrustRegionSubproblem,
    EigenSubproblem,
    InfeasSubproblem,
    EigenQuasiNewton,
    CompactEigenApprox,
    EigenApprox,
    CsrSymbolic,
    UserLibraryProblem,
    live_host_mirrors,
    live_objects,
    quasidef_factor,
    quasidef_apply,
    quasidef_factor_info,
    bench_mdot,
    bench_kernels,
    bench_stream,
    bench_vec_api,
    bench_wgram,
    wgram,
    wgram_with_groups,
    group_panel,
)
from .lib import LIB_PATH, ParOptAMDError  # noqa: F401
