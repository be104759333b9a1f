for cfg in "5e-9 1" "5e-9 0" "1e-7 0" "1e-7 1"; do set -- $cfg
echo "== VTOL=$1 ADTAU=$2"
RBP_JQ_VTOL=$1 RBP_JQ_ADTAU=$2 python tools/r05_joint_vs_golden.py joint64_sweep.npz 2>&1 | grep -v amdgpu.ids
done
