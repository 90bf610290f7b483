#!/bin/bash
# tools/scale_probe.py with every library variant under build_variants/:  tools/variant_probe.sh n d k [mc]
cd "$(dirname "$0")/.."
for lib in build_variants/libital_*.so; do
  echo "== $(basename $lib .so)"
  ITAL_HIP_LIB=$PWD/$lib python tools/scale_probe.py "$@" 2>&1 | grep "fetch_un\|score_generic\|qmc_" | cut -c1-1500
done
