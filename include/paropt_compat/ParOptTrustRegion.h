/* ParOptTrustRegion.h -- the reference's header name (src/ParOptTrustRegion.h:15-480).  Provides, from the MI355X facade
 * (include/ParOptAMD.hpp): ParOptTrustRegionSubproblem (the interface), ParOptQuadraticSubproblem(problem, qn),
 * ParOptTrustRegion(subproblem, options) with optimize(ParOptInteriorPoint*), initialize, setPenaltyGamma (both forms),
 * getPenaltyGamma, getOptimizedPoint, addDefaultOptions.  Not provided: ParOptInfeasSubproblem as a user-visible class
 * (the steering problem lives inside the library) and user-written subclasses of ParOptTrustRegionSubproblem under
 * ParOptTrustRegion.
 * Build: -I include/paropt_compat -I <mpi include>, link -lparopt_amd and the MPI library. */
#ifndef PAROPT_AMD_USE_MPI
#define PAROPT_AMD_USE_MPI 1
#endif
#include "../ParOptAMD.hpp"
