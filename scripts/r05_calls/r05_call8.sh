cd $GRAFT_REPO_ROOT
timeout 100 python scripts/icp_overlap_probe.py small ra 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | tail -22
timeout 100 python scripts/icp_overlap_probe.py big host 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | tail -22
timeout 100 python scripts/icp_overlap_probe.py big ra 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | tail -22
