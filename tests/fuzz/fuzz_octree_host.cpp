#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "pcc.h"
void pcc_set_error(const char* fmt, ...) {}
#include "octree_host.cpp"
int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  // build a blob from a random-ish tree via pack: 3 levels
  std::vector<uint8_t> occ; std::vector<int64_t> ln;
  uint64_t seed = 12345; auto rnd = [&]() { seed = seed * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(seed >> 33); };
  int depth = 6; int64_t cur = 1, n = 0;
  for (int L = 0; L < depth; ++L) { ln.push_back(cur); int64_t nxt = 0; for (int64_t i = 0; i < cur; ++i) { uint8_t b = (uint8_t)(rnd() & 0xFF); if (!b) b = 1 << (rnd() & 7); if (L > 2) b &= (uint8_t)(rnd() | 0x11); if (!b) b = 1; occ.push_back(b); nxt += __builtin_popcount(b); } cur = nxt; }
  n = cur;
  std::vector<uint8_t> blob(occ.size() * 2 + 256); int64_t len = 0; int32_t org[3] = {0, -64, 8};
  int rc = pcc_octree_pack(occ.data(), ln.data(), depth, n, org, blob.data(), (int64_t)blob.size(), &len);
  printf("pack rc %d n %lld nodes %zu len %lld\n", rc, (long long)n, occ.size(), (long long)len);
  std::vector<int32_t> pts((size_t)n * 3 + 3);
  int64_t lv[16];
  rc = pcc_octree_unpack_levels(blob.data(), len, pts.data(), n, lv);
  printf("unpack rc %d\n", rc);
  int errs = 0, oks = 0;
  for (int it = 0; it < iters; ++it) {
    std::vector<uint8_t> b(blob.begin(), blob.begin() + len);
    int flips = 1 + (rnd() % 3);
    for (int f = 0; f < flips; ++f) b[rnd() % len] ^= (uint8_t)(1u << (rnd() & 7));
    int64_t cut = (it % 7 == 0) ? (int64_t)(rnd() % len) : len;
    std::vector<int32_t> p2((size_t)n * 3 + 3);
    int r = pcc_octree_unpack_levels(b.data(), cut, p2.data(), n, lv);
    if (r == 0) ++oks; else ++errs;
    std::vector<int32_t> v; int64_t l2[16];
    pcc_octree_unpack_vec(b.data(), cut, &v, l2);
  }
  printf("fuzz: %d ok, %d errors\n", oks, errs);
  return 0;
}
