#!/bin/bash
mkdir -p gpurun_out/p
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/p/tests.log 2>&1
tail -8 gpurun_out/p/tests.log
