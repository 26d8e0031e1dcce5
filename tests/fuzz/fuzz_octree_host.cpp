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
  // blob version 3: two parts under one root (the first under octant 0 of the root, the second under octant 7), damaged
  // and cut the same way — the envelope, the length table and the order check read untrusted bytes too
  auto tree = [&](uint8_t root_byte, std::vector<uint8_t>* out, int64_t* n_out) {
    std::vector<uint8_t> o; std::vector<int64_t> l; int64_t c = 1;
    for (int L = 0; L < depth; ++L) { l.push_back(c); int64_t nx = 0; for (int64_t i = 0; i < c; ++i) { uint8_t b = L == 0 ? root_byte : (uint8_t)(rnd() & 0xFF); if (!b) b = 1 << (rnd() & 7); if (L > 2) b &= (uint8_t)(rnd() | 0x11); if (!b) b = 1; o.push_back(b); nx += __builtin_popcount(b); } c = nx; }
    out->resize(o.size() * 2 + 256); int64_t ln2 = 0;
    int r2 = pcc_octree_pack(o.data(), l.data(), depth, c, org, out->data(), (int64_t)out->size(), &ln2);
    out->resize((size_t)ln2); *n_out = c;
    return r2;
  };
  std::vector<uint8_t> part[2]; int64_t pn[2];
  int ra = tree(0x01, &part[0], &pn[0]), rb = tree(0x80, &part[1], &pn[1]);
  std::vector<uint8_t> blob3(part[0].size() + part[1].size() + 64); int64_t len3 = 0;
  int rj = pcc_octree_join_parts(depth, org, pn[0] + pn[1], part, 2, blob3.data(), (int64_t)blob3.size(), &len3);
  const int64_t n3 = pn[0] + pn[1];
  std::vector<int32_t> pts3((size_t)n3 * 3 + 3);
  int ru = pcc_octree_unpack_levels(blob3.data(), len3, pts3.data(), n3, lv);
  const int64_t lv1 = lv[depth - 1], lv2 = lv[depth - 2];
  // the parts the other way round must be refused
  std::vector<uint8_t> swp[2] = {part[1], part[0]}; std::vector<uint8_t> blob3s(blob3.size()); int64_t len3s = 0;
  pcc_octree_join_parts(depth, org, n3, swp, 2, blob3s.data(), (int64_t)blob3s.size(), &len3s);
  int rs = pcc_octree_unpack_levels(blob3s.data(), len3s, pts3.data(), n3, lv);
  printf("version 3: pack %d %d join %d unpack %d (n %lld, lower levels %lld %lld) swapped parts %d\n", ra, rb, rj, ru, (long long)n3,
         (long long)lv1, (long long)lv2, rs);
  if (ra || rb || rj || ru || rs == 0 || lv1 <= 0 || lv2 <= 0 || lv1 > n3) return 1;
  int errs3 = 0, oks3 = 0;
  for (int it = 0; it < iters; ++it) {
    std::vector<uint8_t> b(blob3.begin(), blob3.begin() + len3);
    int flips = 1 + (rnd() % 3);
    for (int f = 0; f < flips; ++f) b[(it % 3 == 0 ? rnd() % 64 : rnd() % len3)] ^= (uint8_t)(1u << (rnd() & 7));   // a third of them in the envelope
    int64_t cut = (it % 7 == 0) ? (int64_t)(rnd() % len3) : len3;
    std::vector<int32_t> p2((size_t)n3 * 3 + 3);
    int r = pcc_octree_unpack_levels(b.data(), cut, p2.data(), n3, lv);
    if (r == 0) ++oks3; else ++errs3;
    std::vector<int32_t> v; int64_t l2[16];
    pcc_octree_unpack_vec(b.data(), cut, &v, l2);
  }
  printf("version 3: %d ok, %d errors\n", oks3, errs3);
  printf("fuzz: %d ok, %d errors\n", oks, errs);
  return 0;
}
