#!/bin/bash
# usage: build_variant.sh NAME "-DFLAG ..."  -> lib/variants/libpcp_hip_NAME.so  (kernel A/B experiments on one GPU box; select it
# with PCP_HIP_LIB=<path>)
set -e
cd "$(dirname "$0")"
mkdir -p ../lib/variants
make -j8 OBJDIR=../build/var_$1 LIB=../lib/variants/libpcp_hip_$1.so EXTRA="$2"
