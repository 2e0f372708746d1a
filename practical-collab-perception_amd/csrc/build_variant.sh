#!/bin/bash
# usage: build_variant.sh NAME "-DFLAG ..."  -> lib/variants/libpcp_hip_NAME.so  (kernel A/B experiments on one GPU box)
set -e
cd "$(dirname "$0")"
mkdir -p ../lib/variants ../build/var_$1
for f in abi voxelize pfn conv wino headconv pointhead decode nms fusion hunter; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fhip-fp32-correctly-rounded-divide-sqrt $2 -c $f.hip -o ../build/var_$1/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC ../build/var_$1/*.o -o ../lib/variants/libpcp_hip_$1.so
