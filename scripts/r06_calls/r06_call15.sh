#!/bin/bash
# robustness: soak runs in every odometry mode and gate setting, and the GPU suite three times
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c15; mkdir -p $O
export TMPDIR=/tmp
for mode in icp rgbd_icp rgbd; do
  timeout 900 python tests/tools/soak.py $mode 2>&1 | tail -2 | tee -a $O/soak.log
done
KT_SIDE_GATE=1 timeout 900 python tests/tools/soak.py icp 2>&1 | tail -2 | tee -a $O/soak.log
KT_SIDE_GATE=1 KT_ICP_LEVELS=0 timeout 900 python tests/tools/soak.py icp 2>&1 | tail -2 | tee -a $O/soak.log
KT_RI_LEVELS=1 timeout 900 python tests/tools/soak.py rgbd_icp 2>&1 | tail -2 | tee -a $O/soak.log
timeout 900 python tests/tools/soak.py icp 320 160 2>&1 | tail -2 | tee -a $O/soak.log
for rep in 1 2 3; do
  timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/pytest_$rep.log 2>&1; echo "suite $rep rc $?" | tee -a $O/suite.log; tail -1 $O/pytest_$rep.log | tee -a $O/suite.log
done
