// ORACLE — TEST INFRASTRUCTURE ONLY.
// C entry points over the two reference translation units on the path that
// build without OpenCV: src/LineIterator.cpp and src/gridStructure.cpp.  They
// are compiled FROM /root/reference (never copied) by oracle/Makefile into
// oracle/_ref/libpli_ref.so and used to pin the oracle's Bresenham cell walk
// and grid window query, and to generate tests/golden/grid_*.json.
#include "gridStructure.h"
#include <set>
#include <list>
#include <unordered_set>

extern "C" {

int ref_line_coords(double x1, double y1, double x2, double y2, int* out, int cap) {
  std::list<std::pair<int, int>> lc;
  ORB_SLAM3::getLineCoords(x1, y1, x2, y2, lc);
  int n = 0;
  for (auto& p : lc) {
    if (n < cap) { out[2 * n] = p.first; out[2 * n + 1] = p.second; }
    ++n;
  }
  return n;
}

int ref_grid_query(const double* segs, int nseg, int rows, int cols, int qx, int qy, int wl, int wr, int hu, int hd,
                   int* out, int cap) {
  ORB_SLAM3::GridStructure grid(rows, cols);
  std::list<std::pair<int, int>> lc;
  for (int i = 0; i < nseg; ++i) {
    ORB_SLAM3::getLineCoords(segs[4 * i], segs[4 * i + 1], segs[4 * i + 2], segs[4 * i + 3], lc);
    for (auto& p : lc) grid.at(p.first, p.second).push_back(i);
  }
  ORB_SLAM3::GridWindow w;
  w.width = std::make_pair(wl, wr);
  w.height = std::make_pair(hu, hd);
  std::unordered_set<int> c;
  grid.get(qx, qy, w, c);
  std::set<int> s(c.begin(), c.end());
  int n = 0;
  for (int v : s) { if (n < cap) out[n] = v; ++n; }
  return n;
}
}
