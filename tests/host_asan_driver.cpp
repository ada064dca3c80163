// Host-side AddressSanitizer driver (tests/test_host_asan.py): calls the library's host-only logic -- the grouped weight-gradient
// planner, the split planner of the streaming 3x3 kernel, argument checks that return before any launch -- in a build of the
// library whose HOST code is ASan-instrumented (-fsanitize=address -fno-gpu-sanitize; GPU ASan is not available on this pool).
// Needs no GPU.  Prints one line per check; the test reads them.
#include "camradepth_hip.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
int main() {
  // host-only planning: the grouped weight-gradient table
  std::vector<crd_wgrad_desc> descs;
  static char dummy[64];
  for (int i = 0; i < 37; ++i) {
    crd_wgrad_desc d; memset(&d, 0, sizeof d);
    d.x = dummy; d.dy = dummy; d.dw = (crd_sum_t*)dummy;
    d.B = 8; d.IH = d.OH = 4 + i % 5; d.IW = d.OW = 13; d.Cin = 64 * (1 + i % 4); d.Cout = 32 * (1 + i % 7); d.x_ld = d.Cin; d.dy_ld = d.Cout;
    d.KH = d.KW = 1; d.stride = 1; d.pad = 0;
    descs.push_back(d);
  }
  crd_wgrad_group_info info;
  int rc = crd_wgrad_group_build(descs.data(), (int)descs.size(), nullptr, 0, &info);
  printf("size query rc %d bytes %lld problems %d\n", rc, (long long)info.bytes, info.n_problems);
  std::vector<unsigned char> table(info.bytes);
  rc = crd_wgrad_group_build(descs.data(), (int)descs.size(), table.data(), info.bytes, &info);
  printf("build rc %d items %d %d %d %d\n", rc, info.n_items[0], info.n_items[1], info.n_items[2], info.n_items[3]);
  rc = crd_wgrad_group_build(descs.data(), (int)descs.size(), table.data(), info.bytes - 8, &info);      // too small: must refuse, not overrun
  printf("short table rc %d (%s)\n", rc, crd_last_error());
  crd_wgrad_desc w; memset(&w, 0, sizeof w);
  w.x = dummy; w.dy = dummy; w.dw = (crd_sum_t*)dummy; w.B = 8; w.IH = w.OH = 256; w.IW = w.OW = 416; w.Cin = 304; w.Cout = 128; w.x_ld = 304; w.dy_ld = 128;
  w.KH = w.KW = 3; w.stride = 1; w.pad = 1;
  for (int cap : {0, 1, 12, 51, 1000}) { w.dw_partial_capacity = cap; printf("splits(cap %d) = %d\n", cap, crd_conv_wgrad_splits(&w)); }
  // argument checks return before any launch
  printf("null conv: %d (%s)\n", crd_conv_igemm(nullptr, nullptr), crd_last_error());
  printf("null wgrad: %d (%s)\n", crd_conv_wgrad(nullptr, nullptr), crd_last_error());
  printf("version %d arch %s\n", crd_version(), crd_arch());
  return 0;
}
