#!/bin/bash
OUT=/root/repo/gpurun_out/r2o; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
timeout 300 python tools/host_sampler_rate.py cfg2 > $OUT/host_sampler.txt 2>&1; timeout 300 python tools/host_sampler_rate.py cfg4 >> $OUT/host_sampler.txt 2>&1; grep -v amdgpu $OUT/host_sampler.txt
timeout 600 python -m pytest tests/test_edge_shapes.py -m gpu -q > $OUT/pytest_edge.log 2>&1; tail -2 $OUT/pytest_edge.log
