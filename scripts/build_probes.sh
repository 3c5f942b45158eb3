#!/bin/bash
# compile the stand-alone hardware probes (scripts/probes/*.hip) into openglue_amd/lib/probe_<name> (travels with gpurun);
# *_lib.hip sources become shared libraries openglue_amd/lib/libprobe_<name>.so for the python drivers
cd "$(dirname "$0")/.."
for f in scripts/probes/*.hip; do
  n=$(basename $f .hip)
  case $n in
    *_lib) /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -shared -fPIC $f -o openglue_amd/lib/libprobe_${n%_lib}.so || echo "FAILED $n" ;;
    *) /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w $f -o openglue_amd/lib/probe_$n || echo "FAILED $n" ;;
  esac
done
ls openglue_amd/lib/probe_* openglue_amd/lib/libprobe_*
