#!/bin/bash
cd /root/repo
timeout 400 python -m pytest tests -m gpu -x -q -p no:cacheprovider --ignore=tests/test_ref_pipeline_gpu.py 2>&1 | tail -4
