#!/bin/bash
# Run ON THE GPU BOX: the headline shape over k, super-k-mer form and (for a few k) the key-array form.
# tools/k_sweep.sh > gpurun_out/k_sweep.txt
for K in 17 18 19 20 21 24 27 30 31 32 34 37 41 49 63; do
  tools/ab_env_k.sh sweep $K "KHOICE_SKM_DEBUG=1" 2>/dev/null
  grep "skm\]" gpurun_out/abek_sweep_${K}_1.log | tail -1 | cut -c1-260
done
for K in 15 31 41 63; do tools/ab_env_k.sh sweepka $K "KHOICE_NO_SKM=1" 2>/dev/null; done
