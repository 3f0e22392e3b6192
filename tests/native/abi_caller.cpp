// A host program in the reference's own host language (C++; the reference calls its rasterizer from C++ snippets that Jittor
// JIT-compiles: gaussian_renderer/diff_gaussian_rasterizater/rasterize_points.py:88-274, 276-401) that drives the path through
// include/gmesh_hip.h and NOTHING else: no Python, no torch.  Device memory comes from the HIP runtime; the call sequence is the
// reference bridge's:
//   geom = buffer of required<GeometryState>(P)          -> gm_geom_bytes
//   Rasterizer::forward_0(...) -> num_rendered            -> gm_forward_0
//   binning = buffer of required<BinningState>(R), img = buffer of required<ImageState>(W * H)
//   Rasterizer::forward_1(...)                            -> gm_forward_1
//   Rasterizer::backward(...)                             -> gm_backward
// Inputs are read from and outputs written to raw little-endian files in a directory (tests/test_gpu_native_abi.py writes the
// scene, runs this program and compares what it wrote with the oracle).
#include <hip/hip_runtime_api.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "gmesh_hip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define GM_OK_(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, gm_last_error()); return 3; } } while (0)

template <typename T>
static std::vector<T> read_file(const std::string& path, size_t count) {
  std::vector<T> v(count);
  FILE* f = fopen(path.c_str(), "rb");
  if (!f || fread(v.data(), sizeof(T), count, f) != count) { fprintf(stderr, "cannot read %zu items from %s\n", count, path.c_str()); exit(4); }
  fclose(f);
  return v;
}
template <typename T>
static void write_dev(const std::string& path, const T* dev, size_t count) {
  std::vector<T> h(count);
  if (hipMemcpy(h.data(), dev, count * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) { fprintf(stderr, "copy back failed for %s\n", path.c_str()); exit(5); }
  FILE* f = fopen(path.c_str(), "wb");
  fwrite(h.data(), sizeof(T), count, f);
  fclose(f);
}
template <typename T>
static T* to_dev(const std::vector<T>& h) {
  T* d = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&d), h.size() * sizeof(T)) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); exit(6); }
  if (hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) { fprintf(stderr, "upload failed\n"); exit(6); }
  return d;
}
template <typename T>
static T* dev_alloc(size_t count) {
  T* d = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&d), (count ? count : 1) * sizeof(T)) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); exit(6); }
  return d;
}

int main(int argc, char** argv) {
  if (argc != 2) { fprintf(stderr, "usage: %s <directory>\n", argv[0]); return 1; }
  const std::string dir = std::string(argv[1]) + "/";
  if (gm_abi_version() != GM_ABI_VERSION) { fprintf(stderr, "header / library ABI mismatch\n"); return 1; }
  const std::vector<int> meta = read_file<int>(dir + "meta.bin", 5);                 // P, W, H, D, M
  const int P = meta[0], W = meta[1], H = meta[2], D = meta[3], M = meta[4];
  const std::vector<float> cam = read_file<float>(dir + "camera.bin", 16 + 16 + 3 + 2 + 3);   // view, proj, campos, tanx, tany, background
  float* means = to_dev(read_file<float>(dir + "means.bin", 3 * (size_t)P));
  float* shs = to_dev(read_file<float>(dir + "shs.bin", 3 * (size_t)M * P));
  float* opac = to_dev(read_file<float>(dir + "opac.bin", (size_t)P));
  float* scales = to_dev(read_file<float>(dir + "scales.bin", 3 * (size_t)P));
  float* rots = to_dev(read_file<float>(dir + "rots.bin", 4 * (size_t)P));
  float* dpix = to_dev(read_file<float>(dir + "dpix.bin", 3 * (size_t)W * H));
  float* view = to_dev(std::vector<float>(cam.begin(), cam.begin() + 16));
  float* proj = to_dev(std::vector<float>(cam.begin() + 16, cam.begin() + 32));
  float* campos = to_dev(std::vector<float>(cam.begin() + 32, cam.begin() + 35));
  const float tanx = cam[35], tany = cam[36];
  float* bg = to_dev(std::vector<float>(cam.begin() + 37, cam.begin() + 40));

  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  // ---- forward, first half: the caller owns every buffer (the reference's python allocates geomBuffer / binningBuffer / imgBuffer)
  void* geom = dev_alloc<char>(gm_geom_bytes(P));
  int* radii = dev_alloc<int>(P);
  int num_rendered = -1;
  GM_OK_(gm_forward_0(geom, P, D, M, bg, W, H, means, shs, nullptr, opac, scales, 1.0f, rots, nullptr, view, proj, campos, tanx, tany, 0,
                      radii, 0, stream, &num_rendered));
  // ---- second half
  void* binning = dev_alloc<char>(gm_binning_bytes(num_rendered));
  void* img = dev_alloc<char>(gm_image_bytes(W, H));
  float* color = dev_alloc<float>(3 * (size_t)W * H);
  GM_OK_(gm_forward_1(geom, binning, img, P, D, M, num_rendered, bg, W, H, means, shs, nullptr, opac, scales, 1.0f, rots, nullptr, view, proj,
                      campos, tanx, tany, 0, color, radii, 0, stream));
  // ---- backward
  float* d_m2d = dev_alloc<float>(3 * (size_t)P); float* d_conic = dev_alloc<float>(4 * (size_t)P); float* d_op = dev_alloc<float>(P);
  float* d_col = dev_alloc<float>(3 * (size_t)P); float* d_m3d = dev_alloc<float>(3 * (size_t)P); float* d_cov = dev_alloc<float>(6 * (size_t)P);
  float* d_sh = dev_alloc<float>(3 * (size_t)M * P); float* d_sc = dev_alloc<float>(3 * (size_t)P); float* d_rot = dev_alloc<float>(4 * (size_t)P);
  GM_OK_(gm_backward(P, D, M, num_rendered, bg, W, H, means, shs, nullptr, scales, 1.0f, rots, nullptr, view, proj, campos, tanx, tany, radii,
                     geom, binning, img, dpix, d_m2d, d_conic, d_op, d_col, d_m3d, d_cov, d_sh, d_sc, d_rot, 0, stream));
  HIP_OK(hipStreamSynchronize(stream));
  FILE* f = fopen((dir + "num_rendered.bin").c_str(), "wb"); fwrite(&num_rendered, sizeof(int), 1, f); fclose(f);
  write_dev(dir + "color.bin", color, 3 * (size_t)W * H);
  write_dev(dir + "radii.bin", radii, (size_t)P);
  write_dev(dir + "d_means3D.bin", d_m3d, 3 * (size_t)P);
  write_dev(dir + "d_opacity.bin", d_op, (size_t)P);
  write_dev(dir + "d_sh.bin", d_sh, 3 * (size_t)M * P);
  write_dev(dir + "d_scale.bin", d_sc, 3 * (size_t)P);
  write_dev(dir + "d_rot.bin", d_rot, 4 * (size_t)P);
  write_dev(dir + "d_mean2D.bin", d_m2d, 3 * (size_t)P);
  printf("abi_caller: P=%d %dx%d D=%d num_rendered=%d\n", P, W, H, D, num_rendered);
  return 0;
}
