#!/bin/bash
timeout 1200 python -m pytest tests/test_bench_gpu.py -x -q -m gpu 2>&1 | tail -6
