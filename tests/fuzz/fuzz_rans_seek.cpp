#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "pcc.h"
void pcc_set_error(const char* fmt, ...) {}
#include "rans_host.cpp"
int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  uint64_t seed = 99; auto rnd = [&]() { seed = seed * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(seed >> 33); };
  // two CDF tables
  const int pitch = 20; int32_t cdfs[2 * pitch] = {0}; int32_t sizes[2] = {12, 7}, offs[2] = {-5, -2};
  { int32_t c = 0; for (int i = 0; i < 12; ++i) { cdfs[i] = c; c += (i == 11) ? 0 : (i == 5 ? 40000 : 2321); } cdfs[11] = 65536; }
  { int32_t v[7] = {0, 100, 5000, 60000, 65000, 65500, 65536}; for (int i = 0; i < 7; ++i) cdfs[pitch + i] = v[i]; }
  const int64_t n = 20000;
  std::vector<int32_t> sym(n), idx(n);
  for (int64_t i = 0; i < n; ++i) { idx[i] = rnd() & 1; int32_t s = (int32_t)(rnd() % 9) - 4; if (rnd() % 300 == 0) s = (int32_t)(rnd() % 100000) - 50000; sym[i] = s; }
  std::vector<uint8_t> out(n * 8 + 64); int64_t len = 0;
  int64_t si[5] = {64, 5000, 5001, 12000, 19999}; uint64_t st[5]; int64_t wd[5];
  int rc = pcc_rans_encode_seek(sym.data(), idx.data(), n, cdfs, pitch, sizes, offs, 2, out.data(), (int64_t)out.size(), &len, si, 5, st, wd);
  printf("enc rc %d len %lld\n", rc, (long long)len);
  std::vector<int32_t> dec(n, -1);
  uint64_t xo; int64_t wo; int64_t cuts[7] = {0, 64, 5000, 5001, 12000, 19999, n};
  for (int k = 0; k < 6; ++k) {
    rc = pcc_rans_decode_range(out.data(), len, idx.data(), n, cdfs, pitch, sizes, offs, 2, dec.data(), cuts[k], cuts[k + 1], k ? st[k - 1] : 0, k ? wd[k - 1] : 0, &xo, &wo);
    if (rc || (k < 5 && (xo != st[k] || wo != wd[k]))) printf("piece %d rc %d mismatch\n", k, rc);
  }
  printf("pieces equal serial: %d\n", (int)(dec == sym));
  int errs = 0, oks = 0;
  for (int it = 0; it < iters; ++it) {
    std::vector<uint8_t> b(out.begin(), out.begin() + len);
    if (it & 1) b[rnd() % len] ^= (uint8_t)(1u << (rnd() & 7));
    const int64_t lo = rnd() % n, hi = lo + rnd() % (n - lo + 1);
    const uint64_t x0 = (it % 3) ? st[rnd() % 5] ^ ((uint64_t)rnd() << (rnd() % 40)) : ((uint64_t)rnd() << 32) | rnd();
    const int64_t w0 = (it % 5) ? wd[rnd() % 5] : (int64_t)(rnd() % (len / 2)) - 100;
    const int64_t cut = (it % 11 == 0) ? (int64_t)(rnd() % len) : len;
    std::vector<int32_t> d2(n);
    int r = pcc_rans_decode_range(b.data(), cut, idx.data(), n, cdfs, pitch, sizes, offs, 2, d2.data(), lo, hi, x0, w0, &xo, &wo);
    if (r == 0) ++oks; else ++errs;
  }
  printf("fuzz: %d ok, %d errors\n", oks, errs);
  return 0;
}
