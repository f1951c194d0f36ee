# per-rank workloads of the strong-scaling metric on one GPU, final build of round 6: self communicator and the forced
# single-rank RCCL communicator (the real ncclAllReduce / ncclAllGather + publish kernel + polled flag), one call
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{ echo "## self communicator"; bash tools/per_rank_sizes.sh; echo "## PAROPT_AMD_FORCE_RCCL=1"; PAROPT_AMD_FORCE_RCCL=1 bash tools/per_rank_sizes.sh; } 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" > gpurun_out/r06_per_rank_sizes.txt
cat gpurun_out/r06_per_rank_sizes.txt
