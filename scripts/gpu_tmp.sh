#!/bin/bash
timeout 900 python scripts/gpu_tmp.py 2>&1 | tail -6
