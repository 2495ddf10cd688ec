#!/bin/bash
# builds the three gfx950 micro-benchmarks next to their sources (no GPU needed to compile)
cd "$(dirname "$0")"
for f in mfma_ceiling mfma_power dma_patterns issue_model issue_classes mfma_patterns issue_vmem; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w $f.hip -o $f || exit 1
done
echo built: mfma_ceiling mfma_power dma_patterns issue_model issue_classes mfma_patterns issue_vmem
