#!/bin/bash
# Measurement-only builds of libgdr_hip.so with one piece of K6/K7 removed (render.hip GDR_ABLATE=k; the results of
# these libraries are WRONG by construction).  build/libgdr_abl<k>.so; use with GDR_LIB_PATH=... python bench.py
set -e
cd "$(dirname "$0")/.."
L=generativedensification_amd/lib; S=generativedensification_amd/csrc; mkdir -p build
make -C $S -j4 >/dev/null
for k in ${KS:-1 2 3 4 5 6}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fhip-fp32-correctly-rounded-divide-sqrt \
      -DGDR_ABLATE=$k -c $S/render.hip -o build/render_abl$k.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libgdr_abl$k.so $L/preprocess.o $L/preprocess_surfel.o \
      $L/binning.o build/render_abl$k.o $L/render_surfel.o $L/loss.o $L/maps_surfel.o $L/knn.o $L/select.o $L/api.o
done
ls -la build/*.so
