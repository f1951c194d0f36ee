gcc -std=c99 -Iinclude examples/c_abi_quadratic.c -Lparopt_amd -lparopt_amd -Wl,-rpath,$PWD/paropt_amd -Wl,-rpath-link,/opt/rocm/lib -o /tmp/cq && /tmp/cq 20000 2>&1 | tail -3
PAROPT_AMD_NO_LEAN_STEP=1 /tmp/cq 20000 2>&1 | tail -1
PAROPT_AMD_NO_FUSED_MERIT=1 /tmp/cq 20000 2>&1 | tail -1
