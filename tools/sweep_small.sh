#!/bin/bash
# frames/s of small device-resident batches under different small-launch thresholds (tuning of the host-side form selection)
for B in 2 4 8 16 32; do
  for pg in 0 64; do
    for fw in 1 2 4; do
      echo -n "B=$B pyr_group_max=$pg fast_waves=$fw: "
      ORBX_PYR_GROUP_MAX_IMAGES=$pg ORBX_FAST_WAVES=$fw python tools/bench_b1.py $B 400 2>/dev/null | tail -1
    done
  done
done
