/* ParOptTrustRegion.h -- the reference's header name (src/ParOptTrustRegion.h), so that code written against smdogroup/paropt recompiles
 * unchanged: MPI_Comm communicators (PAROPT_AMD_USE_MPI), the whole class set from the MI355X facade.
 * Build: -I include/paropt_compat -I <mpi include>, link -lparopt_amd and the MPI library. */
#ifndef PAROPT_AMD_USE_MPI
#define PAROPT_AMD_USE_MPI 1
#endif
#include "../ParOptAMD.hpp"
