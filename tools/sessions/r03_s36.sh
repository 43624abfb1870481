#!/bin/bash
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -w -o /tmp/bf16_mfma_probe tools/probes/bf16_mfma_probe.hip > /tmp/probe_build.log 2>&1; tail -2 /tmp/probe_build.log; /tmp/bf16_mfma_probe
