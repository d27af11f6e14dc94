#!/bin/bash
# development aid: tools/build_exp.sh N [file] -> oryon_amd/liboryon_hip_dev_expN.so = the dev build with <file> (default pdsc_encoder.hip)
# compiled with -DPDSC_EXP=N (temporary experiment switches).  Select it with ORYON_DEVLIB=<path> (tools/_devlib.py).
set -e
cd "$(dirname "$0")/../oryon_amd/csrc"
N=$1; F=${2:-pdsc_encoder}
make -s dev >/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -Xclang -target-feature -Xclang -packed-fp32-ops \
  -DORYON_DEV -DPDSC_EXP=$N -c $F.hip -o obj_dev/${F}_exp$N.o 2>&1 | grep -v "packed-fp32-ops\|warning\|^ \|^$\|generated" || true
OBJS=$(ls obj_dev/*.o | grep -v "_exp[0-9]*.o" | grep -v "obj_dev/$F.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../liboryon_hip_dev_exp$N.so $OBJS obj_dev/${F}_exp$N.o
echo built ../liboryon_hip_dev_exp$N.so
