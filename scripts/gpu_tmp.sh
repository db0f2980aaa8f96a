#!/bin/bash
timeout 600 python -m pytest tests/test_geo_decode.py tests/test_pipeline.py -x -q -m gpu 2>&1 | tail -8
timeout 600 python scripts/geo_bench.py 2>&1 | tail -1
