#!/bin/bash
mkdir -p gpurun_out/r04l
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r04l/gpu_tests.log
cat gpurun_out/r04l/gpu_tests.log
