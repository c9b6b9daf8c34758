#!/bin/bash
# tools/tw_parts.sh build (here, no GPU needed) | run (on the GPU box): tile_write_kernel with its later parts left out
if [ "$1" = build ]; then
  for k in 1 2 3 99; do hipcc --offload-arch=gfx950 -O3 -ffp-contract=fast -DTW_STOP_AFTER=$k -I3dgs-avatar-release_amd/csrc -o tools/tw_parts_$k.bin tools/tw_parts.hip > /tmp/tw_$k.log 2>&1 || { grep error -A3 /tmp/tw_$k.log | head; exit 1; }; done
else
  for k in 99 1 2 3; do timeout -k 10 60 tools/tw_parts_$k.bin; done
fi
