#!/bin/bash
# register / scratch / LDS usage of one kernel instantiation under extra compile flags (compile only, no GPU)
# usage: tools/kres.sh <HP> '<symbol regex>' [extra hipcc flags...]
HP=$1; PAT=$2; shift 2
cd /root/repo/pydens_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Wno-unused-result --cuda-device-only \
  -Rpass-analysis=kernel-resource-usage -DPINN_INST_HP=$HP "$@" -c pinn_inst.inc -o /tmp/kres_$$.o 2>&1 \
  | grep -A12 "Function Name: .*$PAT" | grep -v "^--" | sed 's/.*remark: [^ ]* //' | paste -sd' ' | sed 's/Function Name/\nFunction Name/g'
rm -f /tmp/kres_$$.o
