#!/bin/bash
# A/B of the single-step EKF kernel's build variants (through gpurun): bash scripts/experiments/gpu_ekf_step_ab.sh OUTDIR
OUT=${1:-gpurun_out/step_ab}; mkdir -p $OUT
for round in 1 2; do
  for lib in cpprobotics_amd/libcrx.so scripts/_diag/libcrx_step_*.so; do
    CRX_LIB_PATH=$PWD/$lib timeout 120 python scripts/experiments/gpu_ekf_step_ab.py 2>>$OUT/err.txt | tee -a $OUT/ekf_step_ab.jsonl | cut -c1-400
  done
done
