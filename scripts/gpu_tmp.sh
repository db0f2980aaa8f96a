#!/bin/bash
timeout 600 python -m pytest tests/test_geo_decode.py -x -q -m gpu 2>&1 | tail -12
timeout 300 python scripts/geo_bench.py --parts 2>&1 | grep "^attention" | tail -2
timeout 300 python scripts/geo_bench.py --fb 2>&1 | tail -4
