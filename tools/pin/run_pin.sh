#!/bin/sh
# One command to pin the oracle against a real OpenCV (3.3.1 is what the reference names):  tools/pin/run_pin.sh [opencv prefix]
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"; ROOT="$(cd "$HERE/../.." && pwd)"
OUT="$ROOT/tests/golden/opencv_pin"; IN="$OUT/inputs"
mkdir -p "$OUT"
python3 "$HERE/dump_inputs.py" "$IN"
if [ -n "$1" ]; then export PKG_CONFIG_PATH="$1/lib/pkgconfig:$PKG_CONFIG_PATH"; fi
PC=opencv; pkg-config --exists opencv4 2>/dev/null && PC=opencv4
g++ -std=c++11 -O2 "$HERE/pin_against_opencv.cpp" -o "$OUT/pin_against_opencv" $(pkg-config --cflags --libs $PC)
"$OUT/pin_against_opencv" "$IN" "$OUT"
# stage 2 (optional): the reference's OWN extractors, compiled where they lie:   PLI_SLAM_ROOT=/path/to/PLI-SLAM tools/pin/run_pin.sh [prefix]
if [ -n "$PLI_SLAM_ROOT" ]; then
  R="$PLI_SLAM_ROOT"
  g++ -std=c++11 -O3 -march=native "$HERE/pin_reference_extractors.cpp" "$R/src/ORBextractor.cc" "$R/src/LineExtractor.cc" "$R/src/Config.cpp" \
      "$R/Thirdparty/line_descriptor/src/LSDDetector_custom.cpp" "$R/Thirdparty/line_descriptor/src/binary_descriptor_custom.cpp" \
      -I"$R" -I"$R/include" -I"$R/Thirdparty/line_descriptor/include" -I/usr/include/eigen3 -o "$OUT/pin_reference_extractors" \
      $(pkg-config --cflags --libs $PC)
  "$OUT/pin_reference_extractors" "$IN" "$OUT"
fi
# stage 3 (with PLI_SLAM_ROOT; round 5): the C++ drop-in against the REAL OpenCV headers — the GPU suite compiles the adapters against
# tests/stubs/opencv2 only — and the field offsets of cv::KeyPoint / cv::line_descriptor::KeyLine against pli_keypoint / pli_keyline
if [ -n "$PLI_SLAM_ROOT" ]; then
  R="$PLI_SLAM_ROOT"
  INC="-I$ROOT -I$ROOT/include -I$R/Thirdparty/line_descriptor/include $(pkg-config --cflags $PC)"
  g++ -std=c++11 -fsyntax-only $INC "$HERE/adapters_against_opencv.cpp" && echo "stage 3: the adapters compile against the installed OpenCV headers"
  # (the linked run prints the offsets; the library is only needed to resolve the symbols: none of it is called)
  if [ -f "$ROOT/pli_slam_amd/csrc/libpli_frontend.so" ]; then
    g++ -std=c++11 -O1 $INC "$HERE/adapters_against_opencv.cpp" -o "$OUT/adapters_against_opencv" -L"$ROOT/pli_slam_amd/csrc" -lpli_frontend \
        -Wl,-rpath,"$ROOT/pli_slam_amd/csrc" $(pkg-config --libs $PC) -pthread && "$OUT/adapters_against_opencv" | tee "$OUT/adapter_layouts.txt"
  else
    g++ -std=c++11 -O1 $INC "$HERE/adapters_against_opencv.cpp" "$ROOT/tests/cpp/mock_pli.cpp" -o "$OUT/adapters_against_opencv" \
        $(pkg-config --libs $PC) -pthread && "$OUT/adapters_against_opencv" | tee "$OUT/adapter_layouts.txt"
  fi
fi
python3 "$HERE/pin_compare.py" "$OUT"
