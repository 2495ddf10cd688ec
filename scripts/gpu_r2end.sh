#!/bin/bash
# round 2, closing run: the whole GPU suite exactly as the driver runs it, then smoke
mkdir -p gpurun_out/r2end
timeout 230 python -m pytest tests/ -x -q -m gpu > gpurun_out/r2end/tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r2end/tests.log
tail -4 gpurun_out/r2end/tests.log
