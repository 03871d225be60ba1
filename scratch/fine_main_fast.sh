#!/bin/bash
# A/B on the GPU: the fine pass's main query on the fast kernel (IBLNERF_FINE_MAIN_FAST=1) under the default mode
cd "$(dirname "$0")/.."
echo "== default"; python scratch/worst_rays_direct.py 2>&1 | grep "f16x3_mxfp6x" 
echo "== fine main fast"; IBLNERF_FINE_MAIN_FAST=1 python scratch/worst_rays_direct.py 2>&1 | grep "f16x3_mxfp6x"
IBLNERF_FINE_MAIN_FAST=1 timeout 900 python -m pytest tests/test_gpu_fitted.py tests/test_gpu_parity.py -q -x -k "f16x3_mxfp6x" 2>&1 | tail -5
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | cut -c1-400
IBLNERF_FINE_MAIN_FAST=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | cut -c1-400
