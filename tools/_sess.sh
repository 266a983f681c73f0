cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/gpu_session.sh r04_n tests:"-s -k pickplace_8192" > /dev/null 2>&1
grep -E "gripper:|objects:|arm:|oracle fed|PickPlace step|passed|failed|Error" gpurun_out/r04_n_pytest_gpu.txt | cut -c1-250
bash tools/gpu_session.sh r04_n ab:librsim_hip_prev.so:librsim_hip.so:pickplace ab:librsim_hip_prev.so:librsim_hip.so:stack 2>&1 | grep value | cut -c1-150
bash tools/gpu_session.sh r04_n2 tests 2>&1 | tail -8 | cut -c1-200
