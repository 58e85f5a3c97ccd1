#!/bin/bash
# Builds a variant of the forward library for A/B runs and diagnostics (forward + pack + rays only):
#   tools/build_variant.sh NAME [-DFLAGS...]   ->  tools/bin/libnerfhip_NAME.so
# e.g. -DFWD_STAMPS (clock and layer stamps for tools/clock_probe.py), -DNERF_FWD_NPT=2,
# -DEXP_STALE_RING / -DEXP_NO_LDS_READ (timing-only knock-outs, results wrong).
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin
N=$1; shift
for f in mlp_forward_wide mlp_pack rays; do
  hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -fvisibility=hidden -std=c++17 -Iinclude "$@" -c nerf_amd/csrc/$f.hip -o tools/bin/${f}_$N.o &
done
wait
hipcc --offload-arch=gfx950 -shared -o tools/bin/libnerfhip_$N.so tools/bin/mlp_forward_wide_$N.o tools/bin/mlp_pack_$N.o tools/bin/rays_$N.o
rm -f tools/bin/*_$N.o
