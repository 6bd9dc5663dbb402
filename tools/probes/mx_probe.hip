// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 (gfx950): operand k-mapping and scale association, determined empirically.
// The kernel runs one instruction per test case on operands / scales the HOST wrote lane by lane, and returns the full C.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mx_probe.hip -o tools/probes/mx_probe && tools/probes/mx_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void run(const int* a, const int* b, const int* sa, const int* sb, float* c) {
  const int lane = threadIdx.x, t = blockIdx.x;
  i32x8 va, vb;
  for (int i = 0; i < 8; ++i) { va[i] = a[(t * 64 + lane) * 8 + i]; vb[i] = b[(t * 64 + lane) * 8 + i]; }
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, acc, 0, 0, 0, sa[t * 64 + lane], 0, sb[t * 64 + lane]);
  for (int r = 0; r < 16; ++r) c[(t * 64 + lane) * 16 + r] = acc[r];
}
struct Case { int a[64][8], b[64][8], sa[64], sb[64]; };
static void set_byte(int (&v)[8], int byte, unsigned char val) { v[byte >> 2] |= (int)val << (8 * (byte & 3)); }
// C[row][col] from the lane-major result: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
static float cval(const float* c, int row, int col) {
  for (int h = 0; h < 2; ++h)
    for (int r = 0; r < 16; ++r)
      if ((r & 3) + 8 * (r >> 2) + 4 * h == row) return c[(h * 32 + col) * 16 + r];
  return -1.f;
}
int main() {
  std::vector<Case> cs;
  auto blank = [] { Case k; memset(&k, 0, sizeof(k)); for (int l = 0; l < 64; ++l) k.sa[l] = k.sb[l] = 127; return k; };
  // cases 0..63: A row 0 has 1.0 at (half hA, byte bA) only; B column 0: value code(half, byte) = e4m3 powers -> identifies the partner
  // two passes: B[half][byte] = 2^(byte % 8) (pass 0), 2^(byte / 8 + 4 * half) (pass 1)  (e4m3: 2^n = (n + 7) << 3)
  for (int pass = 0; pass < 2; ++pass)
    for (int ia = 0; ia < 64; ++ia) {
      Case k = blank();
      set_byte(k.a[(ia >> 5) * 32 + 0], ia & 31, 0x38);
      for (int h = 0; h < 2; ++h)
        for (int bb = 0; bb < 32; ++bb) {
          const int n = pass == 0 ? (bb % 8) : (bb / 8 + 4 * h);
          set_byte(k.b[h * 32 + 0], bb, (unsigned char)((n + 7) << 3));
        }
      cs.push_back(k);
    }
  // cases 128..255: scale association.  A row 0 = 1.0 at (hA, bA), B col 0 all ones; only lane L = hs * 32 has A scale 128
  for (int hs = 0; hs < 2; ++hs)
    for (int ia = 0; ia < 64; ++ia) {
      Case k = blank();
      set_byte(k.a[(ia >> 5) * 32 + 0], ia & 31, 0x38);
      for (int l = 0; l < 64; l += 32)
        for (int i = 0; i < 8; ++i) k.b[l][i] = 0x38383838;
      k.sa[hs * 32] = 128;
      cs.push_back(k);
    }
  // cases 256..383: the same for the B scale (only lane hs * 32 has B scale 128)
  for (int hs = 0; hs < 2; ++hs)
    for (int ia = 0; ia < 64; ++ia) {
      Case k = blank();
      set_byte(k.a[(ia >> 5) * 32 + 0], ia & 31, 0x38);
      for (int l = 0; l < 64; l += 32)
        for (int i = 0; i < 8; ++i) k.b[l][i] = 0x38383838;
      k.sb[hs * 32] = 128;
      cs.push_back(k);
    }
  // case 384: rows: A row r all ones scaled by lane r's scale = 127 + (r % 4), lane r+32's = 127: C[r][0] tells per-row scale use
  {
    Case k = blank();
    for (int l = 0; l < 64; ++l)
      for (int i = 0; i < 8; ++i) { k.a[l][i] = 0x38383838; k.b[l][i] = 0x38383838; }
    for (int r = 0; r < 32; ++r) k.sa[r] = 127 + (r % 4);
    cs.push_back(k);
  }
  const int n = (int)cs.size();
  std::vector<int> ha(n * 64 * 8), hb(n * 64 * 8), hsa(n * 64), hsb(n * 64);
  for (int t = 0; t < n; ++t)
    for (int l = 0; l < 64; ++l) {
      for (int i = 0; i < 8; ++i) { ha[(t * 64 + l) * 8 + i] = cs[t].a[l][i]; hb[(t * 64 + l) * 8 + i] = cs[t].b[l][i]; }
      hsa[t * 64 + l] = cs[t].sa[l]; hsb[t * 64 + l] = cs[t].sb[l];
    }
  int *da, *db, *dsa, *dsb;
  float* dc;
  hipMalloc(&da, ha.size() * 4); hipMalloc(&db, hb.size() * 4); hipMalloc(&dsa, hsa.size() * 4); hipMalloc(&dsb, hsb.size() * 4); hipMalloc(&dc, (size_t)n * 64 * 16 * 4);
  hipMemcpy(da, ha.data(), ha.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dsa, hsa.data(), hsa.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb.data(), hsb.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(run, dim3(n), dim3(64), 0, 0, da, db, dsa, dsb, dc);
  std::vector<float> hc((size_t)n * 64 * 16);
  hipMemcpy(hc.data(), dc, hc.size() * 4, hipMemcpyDeviceToHost);
  printf("A (half, byte) -> partner B (half, byte)   | A-scale lane that applies | B-scale lane that applies\n");
  for (int ia = 0; ia < 64; ++ia) {
    const float p0 = cval(&hc[(size_t)(0 * 64 + ia) * 1024], 0, 0), p1 = cval(&hc[(size_t)(64 + ia) * 1024], 0, 0);
    int lo = -1, hi = -1;
    for (int q = 0; q < 8; ++q) { if (p0 == (float)(1 << q)) lo = q; if (p1 == (float)(1 << q)) hi = q; }
    const int bb = (hi >= 0 && lo >= 0) ? (hi % 4) * 8 + lo : -1, bh = hi >= 0 ? hi / 4 : -1;
    const float s0 = cval(&hc[(size_t)(128 + ia) * 1024], 0, 0), s1 = cval(&hc[(size_t)(192 + ia) * 1024], 0, 0);
    const float t0 = cval(&hc[(size_t)(256 + ia) * 1024], 0, 0), t1 = cval(&hc[(size_t)(320 + ia) * 1024], 0, 0);
    printf("  A(%d,%2d) -> B(%d,%2d) [%.0f %.0f] | lane 0: x%.0f  lane 32: x%.0f | lane 0: x%.0f  lane 32: x%.0f\n", ia >> 5, ia & 31, bh, bb, p0, p1, s0, s1, t0, t1);
  }
  printf("all-ones, A scale of lane r = 127 + r %% 4 (lanes 32.. = 127): C[r][0] =");
  for (int r = 0; r < 32; ++r) printf(" %.0f", cval(&hc[(size_t)384 * 1024], r, 0));
  printf("\n");
  return 0;
}
