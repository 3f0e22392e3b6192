// The round-6 entry points driven from C++ through include/gmesh_hip.h and the HIP runtime alone (no Python, no torch), as
// tests/native/abi_caller.cpp does for the reference-shaped calls:
//   gm_mesh_rs_packed_batch            K deformed proxy meshes -> K gather tables                 (pyACAP.GetRS, edittool/__init__.py:102, 109)
//   gm_forward_deformed_batch_async    K frames of a view stream in one launch chain              (edittool/__init__.py:103-131, 421-472 per frame)
// and, frame by frame into a second set of buffers, the single-frame calls the batch stands for (gm_mesh_rs_packed,
// gm_forward_0_deformed_async, gm_forward_1_geom): the program itself compares the two byte for byte (images, radii) and writes the batch's
// outputs for tests/test_gpu_native_abi.py to compare with the oracle.
#include <hip/hip_runtime_api.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
static std::vector<T> from_dev(const T* dev, size_t count) {
  std::vector<T> h(count);
  if (hipMemcpy(h.data(), dev, count * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) { fprintf(stderr, "copy back failed\n"); exit(5); }
  return h;
}
template <typename T>
static void write_host(const std::string& path, const std::vector<T>& h) {
  FILE* f = fopen(path.c_str(), "wb");
  fwrite(h.data(), sizeof(T), h.size(), f);
  fclose(f);
}
template <typename T>
static T* to_dev(const std::vector<T>& h) {
  T* d = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&d), (h.size() ? h.size() : 1) * sizeof(T)) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); exit(6); }
  if (!h.empty() && hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) { fprintf(stderr, "upload failed\n"); exit(6); }
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
  const std::vector<int> meta = read_file<int>(dir + "meta.bin", 7);                  // P, W, H, K, Vm, nfaces, binning capacity
  const int P = meta[0], W = meta[1], H = meta[2], K = meta[3], Vm = meta[4], NF = meta[5], CAP = meta[6];
  if (K < 1 || K > GM_BATCH_MAX) { fprintf(stderr, "K out of range\n"); return 1; }
  int* tri = to_dev(read_file<int>(dir + "tri.bin", 3 * (size_t)P));
  float* w = to_dev(read_file<float>(dir + "weights.bin", 3 * (size_t)P));
  float* cov = to_dev(read_file<float>(dir + "cov.bin", 9 * (size_t)P));
  float* pos = to_dev(read_file<float>(dir + "pos.bin", 3 * (size_t)P));
  float* shs = to_dev(read_file<float>(dir + "shs.bin", 48 * (size_t)P));
  float* opac = to_dev(read_file<float>(dir + "opac.bin", (size_t)P));
  float* verts = to_dev(read_file<float>(dir + "verts.bin", 3 * (size_t)Vm));
  int* faces = to_dev(read_file<int>(dir + "faces.bin", 3 * (size_t)NF));
  int* adj_off = to_dev(read_file<int>(dir + "adj_offsets.bin", (size_t)Vm + 1));
  int* adj = to_dev(read_file<int>(dir + "adj_faces.bin", 3 * (size_t)NF));
  const std::vector<float> v1_all = read_file<float>(dir + "deformed.bin", 3 * (size_t)Vm * K);       // K deformed meshes
  const std::vector<float> cams = read_file<float>(dir + "cameras.bin", (size_t)K * 37);             // per frame: view 16, proj 16, campos 3, tanx, tany
  float* bg = to_dev(read_file<float>(dir + "background.bin", 3));
  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));

  const float* v1[GM_BATCH_MAX]; float* packed[GM_BATCH_MAX]; float* packed_one[GM_BATCH_MAX];
  gm_batch_frame fr[GM_BATCH_MAX];
  int* status = nullptr;                                                               // K x 4 page-locked status words
  HIP_OK(hipHostMalloc(reinterpret_cast<void**>(&status), sizeof(int) * 4 * GM_BATCH_MAX, hipHostMallocDefault));
  memset(status, 0xFF, sizeof(int) * 4 * GM_BATCH_MAX);
  float* view[GM_BATCH_MAX]; float* proj[GM_BATCH_MAX]; float* campos[GM_BATCH_MAX];
  for (int k = 0; k < K; k++) {
    v1[k] = to_dev(std::vector<float>(v1_all.begin() + 3 * (size_t)Vm * k, v1_all.begin() + 3 * (size_t)Vm * (k + 1)));
    packed[k] = dev_alloc<float>(24 * (size_t)Vm); packed_one[k] = dev_alloc<float>(24 * (size_t)Vm);
    const float* c = cams.data() + 37 * (size_t)k;
    view[k] = to_dev(std::vector<float>(c, c + 16)); proj[k] = to_dev(std::vector<float>(c + 16, c + 32)); campos[k] = to_dev(std::vector<float>(c + 32, c + 35));
    fr[k].packed = packed[k]; fr[k].viewmatrix = view[k]; fr[k].projmatrix = proj[k]; fr[k].cam_pos = campos[k];
    fr[k].tan_fovx = c[35]; fr[k].tan_fovy = c[36];
    fr[k].geom_buffer = dev_alloc<char>(gm_geom_bytes(P)); fr[k].binning_buffer = dev_alloc<char>(gm_binning_bytes(CAP));
    fr[k].image_buffer = dev_alloc<char>(gm_image_bytes(W, H));
    fr[k].out_color = dev_alloc<float>(3 * (size_t)W * H); fr[k].radii = dev_alloc<int>(P); fr[k].status_host = status + 4 * k;
  }
  // ---- the batch: K tables in one launch, K frames in one launch chain
  GM_OK_(gm_mesh_rs_packed_batch(K, Vm, NF, verts, v1, faces, adj_off, adj, packed, stream));
  GM_OK_(gm_forward_deformed_batch_async(GM_POLICY_DEFAULT, K, fr, P, 3, 16, W, H, tri, w, cov, pos, shs, opac, bg, CAP, 0, nullptr, 0, stream));
  HIP_OK(hipStreamSynchronize(stream));
  // ---- the calls it stands for, frame by frame, into buffers of their own
  void* geom1 = dev_alloc<char>(gm_geom_bytes(P)); void* bin1 = dev_alloc<char>(gm_binning_bytes(CAP)); void* img1 = dev_alloc<char>(gm_image_bytes(W, H));
  float* color1 = dev_alloc<float>(3 * (size_t)W * H); int* radii1 = dev_alloc<int>(P);
  int mismatches = 0;
  std::vector<float> colors; std::vector<int> radii_all, status_all;
  for (int k = 0; k < K; k++) {
    GM_OK_(gm_mesh_rs_packed(Vm, NF, verts, v1[k], faces, adj_off, adj, packed_one[k], stream));
    GM_OK_(gm_forward_0_deformed_async(GM_POLICY_DEFAULT, geom1, P, 3, 16, W, H, tri, w, packed_one[k], cov, pos, shs, opac, view[k], proj[k], campos[k],
                                       fr[k].tan_fovx, fr[k].tan_fovy, nullptr, nullptr, nullptr, radii1, 0, stream, nullptr, nullptr));
    GM_OK_(gm_forward_1_geom(GM_POLICY_DEFAULT, geom1, bin1, img1, P, -1, CAP, bg, W, H, color1, 0, stream, nullptr, 0, nullptr));
    HIP_OK(hipStreamSynchronize(stream));
    const std::vector<float> ca = from_dev(fr[k].out_color, 3 * (size_t)W * H), cb = from_dev(color1, 3 * (size_t)W * H);
    const std::vector<int> ra = from_dev(fr[k].radii, (size_t)P), rb = from_dev(radii1, (size_t)P);
    const std::vector<float> ta = from_dev(packed[k], 24 * (size_t)Vm), tb = from_dev(packed_one[k], 24 * (size_t)Vm);
    if (memcmp(ca.data(), cb.data(), ca.size() * sizeof(float)) || memcmp(ra.data(), rb.data(), ra.size() * sizeof(int)) ||
        memcmp(ta.data(), tb.data(), ta.size() * sizeof(float))) {
      fprintf(stderr, "frame %d of the batch differs from the single-frame calls\n", k);
      mismatches++;
    }
    colors.insert(colors.end(), ca.begin(), ca.end());
    radii_all.insert(radii_all.end(), ra.begin(), ra.end());
    for (int j = 0; j < 4; j++) status_all.push_back(status[4 * k + j]);
  }
  write_host(dir + "colors.bin", colors);
  write_host(dir + "radii.bin", radii_all);
  write_host(dir + "status.bin", status_all);
  printf("batch_caller: P=%d %dx%d K=%d: %d frame(s) differ from the single-frame calls; num_rendered of frame 0: %d\n", P, W, H, K, mismatches, status[0]);
  return mismatches ? 7 : 0;
}
