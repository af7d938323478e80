#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY (oracle).  Builds the reference's own C++ kd-tree KNN
# (randlanet/utils/src/{bindings,knn}.cpp + nanoflann.hpp, bound as knn_tpk.knn,
# bindings.cpp:5-7) from the sources WHERE THEY LIE under /root/reference into
# oracle/_ref/knn_tpk.so.  No reference source is copied into this repo; the
# reference's own CMake build is not run.  -ffp-contract=off and no -ffast-math make
# d2 the plain IEEE fp32 ((dx*dx)+(dy*dy))+(dz*dz) of nanoflann.hpp:488-497
# (SURVEY.md 8c).  Skips silently when /root/reference is absent (GPU box).
set -euo pipefail
REF=${REF_ROOT:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
SRC="$REF/randlanet/utils/src"
if [ ! -d "$SRC" ]; then echo "[build_ref] $SRC absent - skipped"; exit 0; fi
mkdir -p "$OUT"
if [ -f "$OUT/knn_tpk.so" ] && [ "$OUT/knn_tpk.so" -nt "$SRC/knn.cpp" ]; then
  echo "[build_ref] up to date"; exit 0; fi
TORCH=$(python3 -c "import torch,os;print(os.path.dirname(torch.__file__))")
PYINC=$(python3 -c "import sysconfig;print(sysconfig.get_paths()['include'])")
PB11=$(python3 -c "import pybind11;print(pybind11.get_include())" 2>/dev/null || echo "$TORCH/include")
g++ -O3 -std=c++17 -fPIC -shared -ffp-contract=off \
  -DTORCH_EXTENSION_NAME=knn_tpk -D_GLIBCXX_USE_CXX11_ABI=1 \
  -I"$SRC" -I"$TORCH/include" -I"$TORCH/include/torch/csrc/api/include" -I"$PYINC" -I"$PB11" \
  "$SRC/bindings.cpp" "$SRC/knn.cpp" \
  -L"$TORCH/lib" -Wl,-rpath,"$TORCH/lib" -ltorch -ltorch_cpu -ltorch_python -lc10 \
  -o "$OUT/knn_tpk.so"
echo "[build_ref] built $OUT/knn_tpk.so"
