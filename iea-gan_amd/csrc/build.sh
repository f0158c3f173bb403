#!/usr/bin/env bash
# Build libieagan_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
# The library links libamdhip64 by SONAME only (no rpath): inside a PyTorch process it binds to the
# HIP runtime torch has already loaded.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${HERE}/../libieagan_hip.so"
OBJ="${HERE}/_obj"
mkdir -p "${OBJ}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-value -ffp-contract=fast"
pids=()
for f in api conv_igemm conv_wgrad conv1x1_stream conv1x1_tile conv1x1_bwd conv3x3_bwd conv3x3_lds conv3x3_ws attention bn_elem conv_c1 d_stem sn aug_optim small_ops rrm_fused ortho; do
  [ -f "${HERE}/${f}.hip" ] || continue
  if [ ! -f "${OBJ}/${f}.o" ] || [ "${HERE}/${f}.hip" -nt "${OBJ}/${f}.o" ] || [ -n "$(find "${HERE}" "${HERE}/../../include" -name '*.h' -newer "${OBJ}/${f}.o" 2>/dev/null)" ]; then
    ${HIPCC} ${FLAGS} -c "${HERE}/${f}.hip" -o "${OBJ}/${f}.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
${HIPCC} --offload-arch=gfx950 -shared -fPIC -o "${OUT}" "${OBJ}"/*.o
echo "built ${OUT}"
