#!/bin/bash
# round 5, session 2: same-box alternations of cache-hint variants of the headline kernels
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/s2
{
bash tools/ab_variants.sh nts ntsp
bash tools/ab_variants.sh nts ntsp
} 2>&1 | tee gpurun_out/s2/ab_nt2.txt
