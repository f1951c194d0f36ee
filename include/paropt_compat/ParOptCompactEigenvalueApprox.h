/* ParOptCompactEigenvalueApprox.h -- the reference's header name (src/ParOptCompactEigenvalueApprox.h:7-206).  Provides,
 * from the MI355X facade (include/ParOptAMD.hpp): ParOptCompactEigenApprox(problem, N) with getApproximation / multAdd /
 * evalApproximation / evalApproximationGradient, ParOptEigenQuasiNewton(qn, approx, index) and
 * ParOptEigenSubproblem(problem, eig_qn) with setEigenModelUpdate(data, fn) -- the objects the reference's own user code
 * for BASELINE config 5 assembles (examples/eigenvalue/eigenvalue_opt.py:298-308).
 * Build: -I include/paropt_compat -I <mpi include>, link -lparopt_amd and the MPI library. */
#ifndef PAROPT_AMD_USE_MPI
#define PAROPT_AMD_USE_MPI 1
#endif
#include "../ParOptAMD.hpp"
