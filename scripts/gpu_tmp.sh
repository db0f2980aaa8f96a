#!/bin/bash
timeout 600 python -m pytest tests/test_geo_decode.py -x -q -m gpu 2>&1 | tail -12
