// Drives the whole-GOP C-ABI of include/pcc.h from a plain C++ program: no Python, no torch.
//   cabi_main <ckpt.pccw> <coords.i32> <feats.f32> <n> <n_frames> <out_prefix> [container_version]
// Encodes the GOP at the three settings of shared/config.yaml:12-15, writes <out_prefix>.q{1,2,3}.bin, decodes
// quality 3 again and writes <out_prefix>.xyz.i32 / <out_prefix>.rgb.f32 / <out_prefix>.offsets.i64.
// Then the host-memory forms of the same operators: pcc_encode_gop_host_frames on the frames as per-frame host arrays
// (its containers must equal the first ones: checked here) and pcc_container_points + pcc_decode_gop_packed into host
// arrays (<out_prefix>.pxyz.i32 / .prgb.f32).  tests/test_gpu_cabi.py compares all of it with the Python pipelines and
// the oracle.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "pcc.h"

static std::vector<uint8_t> slurp(const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<uint8_t> b((size_t)n);
  if (n && fread(b.data(), 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "short read %s\n", path); exit(2); }
  fclose(f);
  return b;
}

static void spit(const std::string& path, const void* p, size_t n) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f || (n && fwrite(p, 1, n, f) != n)) { fprintf(stderr, "cannot write %s\n", path.c_str()); exit(2); }
  fclose(f);
}

#define HIPOK(x) do { if ((x) != hipSuccess) { fprintf(stderr, "HIP error at %s\n", #x); return 3; } } while (0)
#define PCCOK(x) do { int rc_ = (x); if (rc_ != PCC_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, pcc_last_error()); return 4; } } while (0)

int main(int argc, char** argv) {
  if (argc != 7 && argc != 8) { fprintf(stderr, "usage: cabi_main ckpt coords feats n n_frames out_prefix [version]\n"); return 1; }
  const std::vector<uint8_t> ckpt = slurp(argv[1]), coords = slurp(argv[2]), feats = slurp(argv[3]);
  const int64_t n = atoll(argv[4]);
  const int n_frames = atoi(argv[5]);
  const std::string prefix = argv[6];
  if ((int64_t)coords.size() != n * 16 || (int64_t)feats.size() != n * 16) { fprintf(stderr, "size mismatch\n"); return 1; }

  pcc_codec* codec = pcc_codec_create(ckpt.data(), ckpt.size(), 0, nullptr);
  if (!codec) { fprintf(stderr, "pcc_codec_create: %s\n", pcc_last_error()); return 4; }
  if (argc == 8) PCCOK(pcc_codec_set_container_version(codec, atoi(argv[7])));   // 1: y / z strings coded on the GPU
  int32_t* d_coords = nullptr;
  float* d_feats = nullptr;
  HIPOK(hipMalloc((void**)&d_coords, coords.size()));
  HIPOK(hipMalloc((void**)&d_feats, feats.size()));
  HIPOK(hipMemcpy(d_coords, coords.data(), coords.size(), hipMemcpyHostToDevice));
  HIPOK(hipMemcpy(d_feats, feats.data(), feats.size(), hipMemcpyHostToDevice));

  const double q[6] = {1.0, 0.0, 0.0, 1.0, 1.0, 1.0};
  pcc_buf out[3];
  std::vector<int64_t> k((size_t)3 * n_frames);
  double enc_s[7], dec_s[6];
  PCCOK(pcc_encode_gop(codec, d_coords, d_feats, n, n_frames, q, 3, out, k.data(), enc_s));
  std::vector<std::vector<uint8_t>> cont(3);
  for (int i = 0; i < 3; ++i) {
    cont[i].assign(out[i].data, out[i].data + out[i].len);   // valid until the next call: copy
    spit(prefix + ".q" + std::to_string(i + 1) + ".bin", cont[i].data(), cont[i].size());
  }
  spit(prefix + ".k.i64", k.data(), k.size() * 8);

  pcc_cloud_info info;
  PCCOK(pcc_decode_gop(codec, cont[2].data(), (int64_t)cont[2].size(), &info, dec_s));
  std::vector<int32_t> xyz((size_t)info.n_points * 4);
  std::vector<float> rgb((size_t)info.n_points * 3);
  int32_t* d_xyz = nullptr;
  float* d_rgb = nullptr;
  HIPOK(hipMalloc((void**)&d_xyz, xyz.size() * 4 + 4));
  HIPOK(hipMalloc((void**)&d_rgb, rgb.size() * 4 + 4));
  PCCOK(pcc_decode_fetch(codec, d_xyz, d_rgb));
  HIPOK(hipMemcpy(xyz.data(), d_xyz, xyz.size() * 4, hipMemcpyDeviceToHost));
  HIPOK(hipMemcpy(rgb.data(), d_rgb, rgb.size() * 4, hipMemcpyDeviceToHost));
  spit(prefix + ".xyz.i32", xyz.data(), xyz.size() * 4);
  spit(prefix + ".rgb.f32", rgb.data(), rgb.size() * 4);
  spit(prefix + ".offsets.i64", info.h_offsets, (size_t)info.n_offsets * 8);
  printf("ok n=%lld frames=%d bytes=%lld,%lld,%lld decoded=%lld enc_ms=%.3f dec_ms=%.3f\n", (long long)n, n_frames,
         (long long)cont[0].size(), (long long)cont[1].size(), (long long)cont[2].size(), (long long)info.n_points,
         1e3 * (enc_s[0] + enc_s[1] + enc_s[2] + enc_s[3] + enc_s[4] + enc_s[5] + enc_s[6]),
         1e3 * (dec_s[0] + dec_s[1] + dec_s[2] + dec_s[3] + dec_s[4] + dec_s[5]));
  // ---- the same GOP from host memory: per-frame points int32 [n_f,3] / colours float32 [n_f,3]
  {
    const int32_t* c = (const int32_t*)coords.data();
    const float* ft = (const float*)feats.data();
    std::vector<std::vector<int32_t>> pts((size_t)n_frames);
    std::vector<std::vector<float>> cols((size_t)n_frames);
    for (int64_t i = 0; i < n; ++i) {
      const int f = c[4 * i];
      if (f < 0 || f >= n_frames) { fprintf(stderr, "frame index %d\n", f); return 1; }
      for (int a = 0; a < 3; ++a) {
        pts[f].push_back(c[4 * i + 1 + a]);
        cols[f].push_back(ft[4 * i + 1 + a]);
      }
    }
    std::vector<const void*> pp, cp;
    std::vector<int64_t> ns;
    for (int f = 0; f < n_frames; ++f) {
      pp.push_back(pts[f].data());
      cp.push_back(cols[f].data());
      ns.push_back((int64_t)pts[f].size() / 3);
    }
    pcc_buf out2[3];
    std::vector<int64_t> k2((size_t)3 * n_frames);
    PCCOK(pcc_encode_gop_host_frames(codec, pp.data(), 0, cp.data(), 0, ns.data(), n_frames, q, 3, out2, k2.data(), enc_s));
    for (int i = 0; i < 3; ++i)
      if ((size_t)out2[i].len != cont[i].size() || memcmp(out2[i].data, cont[i].data(), cont[i].size()) != 0) {
        fprintf(stderr, "pcc_encode_gop_host_frames: container %d differs from pcc_encode_gop's\n", i + 1);
        return 5;
      }
    if (k2 != k) { fprintf(stderr, "pcc_encode_gop_host_frames: k differs\n"); return 5; }
    int64_t cap = 0;
    int32_t nf = 0;
    PCCOK(pcc_container_points(cont[2].data(), (int64_t)cont[2].size(), &cap, &nf));
    if (nf != n_frames || cap < info.n_points) { fprintf(stderr, "pcc_container_points: %lld points, %d frames\n", (long long)cap, nf); return 5; }
    std::vector<int32_t> pxyz((size_t)cap * 3);
    std::vector<float> prgb((size_t)cap * 3);
    pcc_cloud_info info2;
    PCCOK(pcc_decode_gop_packed(codec, cont[2].data(), (int64_t)cont[2].size(), pxyz.data(), prgb.data(), cap, &info2, dec_s));
    if (info2.n_points != info.n_points) { fprintf(stderr, "pcc_decode_gop_packed: %lld points\n", (long long)info2.n_points); return 5; }
    spit(prefix + ".pxyz.i32", pxyz.data(), (size_t)info2.n_points * 12);
    spit(prefix + ".prgb.f32", prgb.data(), (size_t)info2.n_points * 12);
  }
  (void)hipFree(d_coords); (void)hipFree(d_feats); (void)hipFree(d_xyz); (void)hipFree(d_rgb);
  pcc_codec_destroy(codec);
  return 0;
}
