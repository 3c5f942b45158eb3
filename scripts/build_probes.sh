#!/bin/bash
# compile the stand-alone hardware probes (scripts/probes/*.hip) into openglue_amd/lib/probe_<name> (travels with gpurun)
cd "$(dirname "$0")/.."
for f in scripts/probes/*.hip; do
  n=$(basename $f .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w $f -o openglue_amd/lib/probe_$n || echo "FAILED $n"
done
ls openglue_amd/lib/probe_*
