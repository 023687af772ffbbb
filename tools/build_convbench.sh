#!/bin/bash
# builds build/convbench (developer micro-benchmark) against the in-tree conv objects
set -e
cd "$(dirname "$0")/.."
make -C megadetector_amd/csrc -j4 >/dev/null
mkdir -p build
hipcc --offload-arch=gfx950 -O3 -std=c++17 -c tools/convbench.cpp -o build/convbench.o
hipcc --offload-arch=gfx950 build/convbench.o megadetector_amd/csrc/conv_igemm.o megadetector_amd/csrc/conv_v2.o megadetector_amd/csrc/conv_v5.o megadetector_amd/csrc/conv_v5s.o megadetector_amd/csrc/conv_v5c.o megadetector_amd/csrc/conv_v6.o megadetector_amd/csrc/conv_v7.o megadetector_amd/csrc/conv_f8.o -o build/convbench
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/peaks.cpp -o build/peaks
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dma_peak.cpp -o build/dma_peak
