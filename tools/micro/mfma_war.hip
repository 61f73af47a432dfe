// Ground truth for tools/isa_hazards.py / DESIGN.md 3.9: how many wait states behind a v_mfma_f32_16x16x32_bf16 may a VALU
// instruction overwrite one of its SOURCE registers (A, B) - or touch its D / C registers - without changing the result?
//   hipcc --offload-arch=gfx950 -O2 -o tools/micro/mfma_war tools/micro/mfma_war.hip && tools/micro/mfma_war
// For every distance k = 0..11 (k s_nop wait states between the MFMA and the VALU write), every victim register (A0..A3, B0..B3)
// and two pipe states (MFMA issued into an idle matrix pipe / behind three other MFMAs that keep it busy), 64 lanes x 4 values are
// compared with the undisturbed result.  Prints the smallest k without a mismatch per victim.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int K, int VICTIM, bool BUSY>
__global__ void probe(const u32x4* a_in, const u32x4* b_in, f32x4* out) {
  const int lane = threadIdx.x;
  u32x4 a = a_in[lane], b = b_in[lane];
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, d0 = acc, d1 = acc, d2 = acc;
  u32x4 a2 = a, b2 = b;
  const unsigned junk = 0x7fc07fc0u;                       // NaN pairs
#define NOPS(k) ".rept " #k "\n\ts_nop 0\n\t.endr\n\t"
  // fixed registers: acc v[0:3], A v[4:7], B v[8:11], busy-pipe accumulators v[12:23], second operand pair v[24:31]; the victim is
  // named in the asm text (v4..v11)
  constexpr int VREG = 4 + VICTIM;
  if (BUSY) {
    asm volatile(
        "v_mfma_f32_16x16x32_bf16 v[12:15], v[24:27], v[28:31], v[12:15]\n\t"
        "v_mfma_f32_16x16x32_bf16 v[16:19], v[24:27], v[28:31], v[16:19]\n\t"
        "v_mfma_f32_16x16x32_bf16 v[20:23], v[24:27], v[28:31], v[20:23]\n\t"
        "v_mfma_f32_16x16x32_bf16 v[0:3], v[4:7], v[8:11], v[0:3]\n\t"
        ".rept %c8\n\ts_nop 0\n\t.endr\n\t"
        "v_mov_b32 v%c9, %10\n\t"
        "s_nop 15\n\ts_nop 15\n\ts_nop 15"
        : "+{v[0:3]}"(acc), "+{v[12:15]}"(d0), "+{v[16:19]}"(d1), "+{v[20:23]}"(d2), "+{v[4:7]}"(a), "+{v[8:11]}"(b), "+{v[24:27]}"(a2),
          "+{v[28:31]}"(b2)
        : "n"(K), "n"(VREG), "v"(junk));
  } else {
    asm volatile(
        "v_mfma_f32_16x16x32_bf16 v[0:3], v[4:7], v[8:11], v[0:3]\n\t"
        ".rept %c3\n\ts_nop 0\n\t.endr\n\t"
        "v_mov_b32 v%c4, %5\n\t"
        "s_nop 15\n\ts_nop 15\n\ts_nop 15"
        : "+{v[0:3]}"(acc), "+{v[4:7]}"(a), "+{v[8:11]}"(b)
        : "n"(K), "n"(VREG), "v"(junk));
  }
  out[lane] = acc;
  if (d0[0] == 12345.f) out[lane] = d0 + d1 + d2;          // keep the busy-pipe MFMAs alive
}

// D-register probes: RD = true: a VALU read of D[0] K wait states behind the MFMA (expected: the MFMA's result);
// RD = false: a VALU write of D[0] K wait states behind it (expected: the written constant survives - the MFMA's own write-back comes first)
template <int K, bool RD, bool BUSY>
__global__ void probe_d(const u32x4* a_in, const u32x4* b_in, float* out) {
  const int lane = threadIdx.x;
  u32x4 a = a_in[lane], b = b_in[lane];
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, d0 = acc, d1 = acc, d2 = acc;
  u32x4 a2 = a, b2 = b;
  float got = -1.f;
  const float c42 = 42.f;
  if (BUSY) {
    asm volatile(
        "v_mfma_f32_16x16x32_bf16 v[12:15], v[24:27], v[28:31], v[12:15]\n\t"
        "v_mfma_f32_16x16x32_bf16 v[16:19], v[24:27], v[28:31], v[16:19]\n\t"
        "v_mfma_f32_16x16x32_bf16 v[20:23], v[24:27], v[28:31], v[20:23]\n\t"
        "v_mfma_f32_16x16x32_bf16 v[0:3], v[4:7], v[8:11], v[0:3]\n\t"
        ".rept %c9\n\ts_nop 0\n\t.endr\n\t"
        ".if %c10\n\tv_mov_b32 %8, v0\n\t.else\n\tv_mov_b32 v0, %11\n\t.endif\n\t"
        "s_nop 15\n\ts_nop 15\n\ts_nop 15"
        : "+{v[0:3]}"(acc), "+{v[12:15]}"(d0), "+{v[16:19]}"(d1), "+{v[20:23]}"(d2), "+{v[4:7]}"(a), "+{v[8:11]}"(b), "+{v[24:27]}"(a2),
          "+{v[28:31]}"(b2), "+v"(got)
        : "n"(K), "n"(RD ? 1 : 0), "v"(c42));
  } else {
    asm volatile(
        "v_mfma_f32_16x16x32_bf16 v[0:3], v[4:7], v[8:11], v[0:3]\n\t"
        ".rept %c4\n\ts_nop 0\n\t.endr\n\t"
        ".if %c5\n\tv_mov_b32 %3, v0\n\t.else\n\tv_mov_b32 v0, %6\n\t.endif\n\t"
        "s_nop 15\n\ts_nop 15\n\ts_nop 15"
        : "+{v[0:3]}"(acc), "+{v[4:7]}"(a), "+{v[8:11]}"(b), "+v"(got)
        : "n"(K), "n"(RD ? 1 : 0), "v"(c42));
  }
  out[lane] = RD ? got : acc[0];
  if (d0[0] == 12345.f) out[lane] = d0[0] + d1[0] + d2[0];
}

static std::vector<float> ref;
template <int K, bool RD, bool BUSY>
static bool run_d(const u32x4* a, const u32x4* b, float* out) {
  hipLaunchKernelGGL((probe_d<K, RD, BUSY>), dim3(1), dim3(64), 0, 0, a, b, out);
  std::vector<float> h(64);
  hipMemcpy(h.data(), out, 256, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l)
    if (h[l] != (RD ? ref[4 * l] : 42.f)) return false;
  return true;
}
template <bool RD, bool BUSY, int... Ks>
static void sweep_d(const u32x4* a, const u32x4* b, float* out, std::integer_sequence<int, Ks...>) {
  bool ok[] = {run_d<Ks, RD, BUSY>(a, b, out)...};
  int first_ok = -1;
  for (int k = (int)sizeof...(Ks) - 1; k >= 0 && ok[k]; --k) first_ok = k;
  printf("  VALU %s of D, %s pipe: results by distance", RD ? "read " : "write", BUSY ? "busy" : "idle");
  for (bool o : ok) printf(" %c", o ? '.' : 'X');
  printf("   -> safe from %d wait states\n", first_ok);
}

template <int K, int VICTIM, bool BUSY>
static bool run(const u32x4* a, const u32x4* b, f32x4* out) {
  hipLaunchKernelGGL((probe<K, VICTIM, BUSY>), dim3(1), dim3(64), 0, 0, a, b, out);
  std::vector<float> h(256);
  hipMemcpy(h.data(), out, 1024, hipMemcpyDeviceToHost);
  return memcmp(h.data(), ref.data(), 1024) == 0;
}

template <int VICTIM, bool BUSY, int... Ks>
static void sweep(const u32x4* a, const u32x4* b, f32x4* out, std::integer_sequence<int, Ks...>) {
  bool ok[] = {run<Ks, VICTIM, BUSY>(a, b, out)...};
  int first_ok = -1;
  for (int k = (int)sizeof...(Ks) - 1; k >= 0 && ok[k]; --k) first_ok = k;
  printf("  victim %s%d, %s pipe: results by distance", VICTIM < 4 ? "A" : "B", VICTIM & 3, BUSY ? "busy" : "idle");
  for (bool o : ok) printf(" %c", o ? '.' : 'X');
  printf("   -> safe from %d wait states\n", first_ok);
}

int main() {
  std::vector<unsigned> ha(256), hb(256);
  srand(7);
  auto bf = [] { float f = (rand() % 2001 - 1000) / 512.f; unsigned u; memcpy(&u, &f, 4); return u >> 16; };
  for (auto& v : ha) v = bf() | (bf() << 16);
  for (auto& v : hb) v = bf() | (bf() << 16);
  u32x4 *a, *b;
  f32x4* out;
  hipMalloc(&a, 1024), hipMalloc(&b, 1024), hipMalloc(&out, 1024);
  hipMemcpy(a, ha.data(), 1024, hipMemcpyHostToDevice);
  hipMemcpy(b, hb.data(), 1024, hipMemcpyHostToDevice);
  // reference: the victim write far behind the MFMA
  hipLaunchKernelGGL((probe<24, 0, false>), dim3(1), dim3(64), 0, 0, a, b, out);
  ref.resize(256);
  hipMemcpy(ref.data(), out, 1024, hipMemcpyDeviceToHost);
  double s = 0;
  for (float v : ref) s += v;
  printf("reference checksum %.4f (must be finite and non-zero)\n", s);
  using KS = std::make_integer_sequence<int, 12>;
#define SW(V) sweep<V, false>(a, b, out, KS{}); sweep<V, true>(a, b, out, KS{});
  SW(0) SW(1) SW(2) SW(3) SW(4) SW(5) SW(6) SW(7)
  using KD = std::make_integer_sequence<int, 20>;
  sweep_d<true, false>(a, b, (float*)out, KD{});
  sweep_d<true, true>(a, b, (float*)out, KD{});
  sweep_d<false, false>(a, b, (float*)out, KD{});
  sweep_d<false, true>(a, b, (float*)out, KD{});
  return 0;
}
