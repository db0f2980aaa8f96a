#!/bin/bash
mkdir -p gpurun_out/r04k
timeout 900 python -m pytest tests/test_geo_decode.py -x -q -m gpu 2>&1 | tail -30 > gpurun_out/r04k/geo_tests.log
cat gpurun_out/r04k/geo_tests.log
