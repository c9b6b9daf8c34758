#!/bin/bash
# Step time of experimental builds (variants/<name>/, made with GSPLAT_VARIANT=<name> python .../build.py) against the
# product library, each in its own process, rounds interleaved over the list twice.  Usage: tools/variant_cmp.sh name...
for pass in 1 2; do
  for v in product "$@"; do
    if [ $v = product ]; then unset GSPLAT_LIB_PATH; else export GSPLAT_LIB_PATH=$PWD/variants/$v/libgsplat_mi355.so; fi
    echo -n "$v: "; python tools/ab_step.py lib:xcd_map=1 --rounds 3 --steps 200 2>/dev/null | tail -1
  done
done
