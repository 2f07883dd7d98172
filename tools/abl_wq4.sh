#!/bin/bash
# builds ablated variants of the quad-channel weight-gradient kernel into separate libraries and times each
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in NONE NOMFMA NOX NODY "NOX -DWQ4_ABL_NODY" "NOX -DWQ4_ABL_NODY -DWQ4_ABL_NOMFMA"; do
  flags=""; [ "$v" != "NONE" ] && flags="-DWQ4_ABL_$v"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $flags -c xlstm-hved_amd/csrc/conv3d_wgrad_q4.hip -o /tmp/wq4_abl.o
  objs=$(ls xlstm-hved_amd/lib/obj/*.o | grep -v conv3d_wgrad_q4)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o xlstm-hved_amd/lib/libxlstm_hved_hip.so $objs /tmp/wq4_abl.o
  echo "== variant: $v"
  XH_S=128 timeout -k 10 100 python tools/microbench_wgrad_q4.py 2>&1 | grep "wgrad 4->4\|wgrad 16->16"
done
