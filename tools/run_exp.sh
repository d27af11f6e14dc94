for e in ${EXPS:-0 1 2 3}; do
  if [ $e = 0 ]; then L=oryon_amd/liboryon_hip_dev.so; else L=oryon_amd/liboryon_hip_dev_exp$e.so; fi
  echo "== exp $e"; ORYON_DEVLIB=$PWD/$L timeout 120 python ${SCRIPT:-tools/r5_time_reg.py} 2>&1 | grep -v amdgpu | tail -2
done
