// K5, throughput form: numpy's legacy RandomState stream (MT19937 init_genrand + legacy_gauss, sklearn_gpr.py:460-464 ->
// RandomState(seed).standard_normal((S, Lg))) with the generator state in REGISTERS, FOUR streams per wave, no workgroup
// barrier and no LDS round trip per word.
//
// Why.  k_mt_normals (gpet_kernels.hip) walks one stream with one workgroup of three waves: the state lives in LDS, a block
// of 624 words costs three twist phases with a barrier each plus one for the accept counts, only 227 of 256 lanes twist and
// 156 of 192 make attempts: 342 vector + 243 scalar instructions per block, two thirds of them bookkeeping,
// 23.8 ms per step of 1 024 traces = 24 % of the vector-ALU roof.  A batch that fills the GPU has thousands of streams per
// launch, so latency does not matter there -- only instructions per word do.
//
// Layout.  A stream owns one DPP row (16 lanes) of a wave and 40 registers: word i = 64 g + 4 t + j of the 624-word state
// sits in register 4 g + j of lane t (g = 0..9 "groups" of 64 words; the last group holds 48: lanes 12..15 of its four
// registers are kept ZERO).  Attempt n = 16 g + t of a block (legacy_gauss takes words 4 n .. 4 n + 3) is then LANE-LOCAL:
// its four words are registers 4 g .. 4 g + 3 of lane t.  The recurrence x[i] = x[i + 397 | i - 227] ^ f(x[i], x[i + 1])
// becomes, per register, one row-shift DPP operand per source: i + 1 is the next register of the same lane (j < 3) or the
// next lane's first register; i + 397 = 6 groups + 3 lanes + 1 register, i - 227 = -(3 groups + 9 lanes) + 1 register,
// each split over two source registers whose DPP reads return 0 outside the row (bound_ctrl) -- and the zero lanes of the
// last group make the ragged end (word 226 / 227, word 623 -> new word 0) come out of the same instructions: no masks,
// no index arithmetic, ~7 vector instructions per 64 words x 4 streams (v_bitop3, v_xor_dpp).
// Accept / reject (r2 = x1^2 + x2^2 < 1, decided in double by numpy) is pre-filtered in float32 from the two high words
// with the last tempering step dropped: |error| < 3e-5 on r2, so an estimate farther than 4e-5 from 1 (and above 1e-4)
// decides exactly what numpy decides; the ~8e-5 of the attempts inside the band (5e-5 in the code) take the double path (wave-uniform branch).
// Only zs of Lg columns of a row are stored (the structured loop keeps the r0 <= 96 leading normals): stream positions
// are SCALAR state (pair index within the row, row), advanced by popcounts of the accept mask; a group of 16 attempts
// touches stored columns for one stream in five, and only then that stream's row ranks its accepted attempts and queues
// the raw words of the stored ones in LDS; the queue is evaluated 64 records at a time (exact doubles, log, sqrt: the
// arithmetic of legacy_gauss with contraction off) -- every lane busy.
// Bit for bit the numbers of k_mt_normals and of numpy (tests/test_gpu_stages.py::test_normals_stream*).
// Requirements (the launcher checks them, otherwise k_mt_normals runs): every edge of the launch has the same even grid
// length Lg >= 64, the same S and z_cols.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "gpet_dev.h"
#include "gpet_kernels.h"
#include "gpet_options.h"

namespace gpet {

namespace {

template <int CTRL>
__device__ __forceinline__ unsigned int r4_dppz(unsigned int v) {  // row shift; lanes whose source is outside the row read 0
  return (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
#define R4_SHL(n) (0x100 + (n))
#define R4_SHR(n) (0x110 + (n))

// The constants of the recurrence and of the tempering as OPAQUE scalar registers: with literal constants the compiler
// folds the masks into and / and-or / xor sequences; with a register it selects the three-operand forms (v_bfi_b32 for
// the 1 + 31 bit merge, v_bitop3_b32 for y ^ (t & mask)) -- a vector instruction less per word and step.
struct R4Consts {
  unsigned int lo31, mag, tb, tc;
};
__device__ __forceinline__ R4Consts r4_consts() {
  R4Consts k;
  asm volatile("s_mov_b32 %0, 0x7fffffff" : "=s"(k.lo31));
  asm volatile("s_mov_b32 %0, 0x9908b0df" : "=s"(k.mag));
  asm volatile("s_mov_b32 %0, 0x9d2c5680" : "=s"(k.tb));
  asm volatile("s_mov_b32 %0, 0xefc60000" : "=s"(k.tc));
  return k;
}
// (the compiler keeps and + and-or / and + xor even with register masks, so the two three-operand instructions are spelled
//  out; plain vector ALU operations whose results feed ordinary instructions -- never a DPP source or a lane read, the
//  cases whose wait states the compiler's hazard pass would not see behind inline assembly)
__device__ __forceinline__ unsigned int r4_bfi(unsigned int mask, unsigned int one, unsigned int zero) {  // (one & mask) | (zero & ~mask)
  unsigned int d;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "s"(mask), "v"(one), "v"(zero));
  return d;
}
__device__ __forceinline__ unsigned int r4_xor_and(unsigned int y, unsigned int t, unsigned int mask) {  // y ^ (t & mask)
  unsigned int d;
  asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x78" : "=v"(d) : "v"(y), "v"(t), "s"(mask));
  return d;
}
__device__ __forceinline__ unsigned int r4_mix(const R4Consts& k, unsigned int a, unsigned int b) {
  const unsigned int y = r4_bfi(k.lo31, b, a);
  return (y >> 1) ^ ((unsigned int)(-(int)(b & 1u)) & k.mag);
}
// tempering without its last step (y ^= y >> 18): what the float32 pre-filter reads
__device__ __forceinline__ unsigned int r4_temper3(const R4Consts& k, unsigned int y) {
  y ^= y >> 11;
  y = r4_xor_and(y, y << 7, k.tb);
  y = r4_xor_and(y, y << 15, k.tc);
  return y;
}
__device__ __forceinline__ unsigned int r4_temper(unsigned int y) {  // (the rare paths: literal constants)
  y ^= y >> 11;
  y ^= (y << 7) & 0x9D2C5680u;
  y ^= (y << 15) & 0xEFC60000u;
  return y ^ (y >> 18);
}

__device__ __forceinline__ int r4_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ const void* r4_uni_ptr(const void* p) {
  const unsigned long long u = (unsigned long long)p;
  const unsigned int lo = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)u);
  const unsigned int hi = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(u >> 32));
  return (const void*)(((unsigned long long)hi << 32) | lo);
}

#define R4_QCAP 256  // pending records: < 64 left over + what half a block can push (sparse: 4 x 48; dense: drained per group)

// legacy_gauss on `count` queued records (lane = record): the doubles from the four words, f = sqrt(-2 log(r2) / r2), the
// second normal first (numpy returns f x2 and keeps f x1 for the next call).
__device__ __forceinline__ void r4_drain_body(const uint4* q_w, const unsigned int* q_d, double* z0, double* z1, double* z2,
                                              double* z3, unsigned int qhead, unsigned int count) {
#pragma clang fp contract(off)
  __builtin_amdgcn_wave_barrier();  // (the records were written by other lanes of this wave: LDS operations of a wave execute in order)
  const unsigned int lane = threadIdx.x & 63u;
  if (lane < count) {
    const unsigned int slot = (qhead + lane) & (R4_QCAP - 1);
    const uint4 wv = q_w[slot];
    const unsigned int dst = q_d[slot];
    const unsigned int a = (wv.x ^ (wv.x >> 18)) >> 5, b = r4_temper(wv.y) >> 6;
    const unsigned int c = (wv.z ^ (wv.z >> 18)) >> 5, d = r4_temper(wv.w) >> 6;
    const double u1 = ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
    const double u2 = ((double)c * 67108864.0 + (double)d) / 9007199254740992.0;
    const double x1 = 2.0 * u1 - 1.0, x2 = 2.0 * u2 - 1.0;
    const double r2 = x1 * x1 + x2 * x2;
    const double f = sqrt(-2.0 * log(r2) / r2);
    const unsigned int ds = dst >> 30;
    double* zb = ds == 0 ? z0 : (ds == 1 ? z1 : (ds == 2 ? z2 : z3));
    double* z = zb + (dst & 0x0FFFFFFFu);
    z[0] = f * x2;
    if (dst & 0x10000000u) z[1] = f * x1;
  }
}
// Not inlined: the generator calls it from several places of its unrolled block (the dense form after every group), and
// every copy of the double-precision logarithm is ~3 KB of code and ~40 more live registers.  The price of a call: the ~60
// values live across it sit in callee-saved registers, of which the convention offers 8 in 16 -- 127 registers, four
// waves per SIMD (inlined at two sites the kernel needs 130: three waves).
__device__ __attribute__((noinline)) void r4_drain_call(const uint4* q_w, const unsigned int* q_d, double* z0, double* z1, double* z2,
                                                        double* z3, unsigned int qhead, unsigned int count) {
  r4_drain_body(q_w, q_d, z0, z1, z2, z3, qhead, count);
}

// The double-precision accept test of legacy_gauss for the attempts the float32 pre-filter cannot decide (~1e-4 of them):
// numpy's own comparison.  Not inlined: ten copies (one per group of the unrolled block) would be 4 KB of code that runs
// once in 250 groups.
__device__ __attribute__((noinline)) bool r4_exact_accept(unsigned int ta, unsigned int wb, unsigned int tc, unsigned int wd) {
#pragma clang fp contract(off)
  const unsigned int a = (ta ^ (ta >> 18)) >> 5, b = r4_temper(wb) >> 6;
  const unsigned int c = (tc ^ (tc >> 18)) >> 5, d = r4_temper(wd) >> 6;
  const double u1 = ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
  const double u2 = ((double)c * 67108864.0 + (double)d) / 9007199254740992.0;
  const double x1 = 2.0 * u1 - 1.0, x2 = 2.0 * u2 - 1.0;
  const double r2 = x1 * x1 + x2 * x2;
  return !(r2 >= 1.0 || r2 == 0.0);
}

}  // namespace

// grid = ceil(streams / 4) single-wave workgroups; stream L = 4 blockIdx.x + (lane >> 4) -> edge L / n_ahead, iteration
// ahead L % n_ahead (the seeds of future iterations are known a priori, gpet.py:839).
// SPARSE: at most 96 leading normals of a row are stored and at least 80 pairs of a row are not (D >= 80): the one-comparison
// test below is valid (D > the 16 attempts of a group) and the queue is evaluated twice per block, inline.  Otherwise (the
// dense form: any-rank factors multiply every column) every group ranks its attempts and the queue is evaluated per group.
template <bool SPARSE>
__global__ void __launch_bounds__(64) k_mt_normals4(EdgeDev* edges, const unsigned int* seeds, int add_iter, int iter_abs,
                                                    int n_ahead, int n_streams, int Lg, int S, int zc, int zs) {
#pragma clang fp contract(off)
  const int lane = (int)threadIdx.x, tl = lane & 15, my_s = lane >> 4;
  __shared__ uint4 q_w[R4_QCAP];
  __shared__ unsigned int q_d[R4_QCAP];
  const int H = Lg >> 1;         // pairs per sample row (Lg even: a pair never straddles two rows)
  const int Hs = (zs + 1) >> 1;  // pairs of a row with at least one stored normal
  const int D = H - Hs;          // pairs of a row of which nothing is stored
  unsigned int R[40];
#pragma unroll
  for (int r = 0; r < 40; ++r) R[r] = 0u;
  double* Zb[4];
  // position of MY stream (the same value in the sixteen lanes of its row): pairs LEFT in the current sample row (H at its
  // start), the row, and whether the stream still has rows to fill.  One vector instruction advances all four streams.
  int rem = H, row = 0;
  bool live = false;
  // ---- per stream: edge, ring slot, seed -> init_genrand into the stream's row (word i -> register 4 g + j, lane 16 s + t)
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int L = 4 * (int)blockIdx.x + s;
    Zb[s] = nullptr;
    if (L < n_streams) {
      const int e_idx = L / n_ahead, ahead = L - e_idx * n_ahead;
      // (loads through pointers the compiler cannot prove read-only come back in vector registers and would make everything
      //  derived from them -- the streams' positions, every branch on them -- "divergent": r4_uni declares them wave-uniform)
      const EdgeDev& E = edges[e_idx];
      const gpet_scalars* sc = (const gpet_scalars*)r4_uni_ptr(E.sc);
      const int s_done = r4_uni(sc->done), s_force = r4_uni(sc->force), s_status = r4_uni(sc->status);
      const bool skip = (s_done && !s_force) || s_status != GPET_OK;
      if (!skip) {
        const int iter_idx = (iter_abs >= 0 ? iter_abs : r4_uni(sc->iter)) + ahead;
        const int ring = r4_uni(E.z_ring);
        Zb[s] = (double*)r4_uni_ptr(E.Z) + (size_t)(iter_idx % ring) * ((size_t)S * zc);
        if (my_s == s) live = true;
        unsigned int p = (unsigned int)r4_uni((int)seeds[e_idx]) + (add_iter ? (unsigned int)(iter_idx + 1) : 0u);  // gpet.py:839
#pragma unroll
        for (int g = 0; g < 10; ++g) {
          const int nt = g == 9 ? 12 : 16;
#pragma unroll 1
          for (int t = 0; t < nt; ++t) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int i = 64 * g + 4 * t + j;
              if (i > 0) p = 1812433253u * (p ^ (p >> 30)) + (unsigned int)i;
              R[4 * g + j] = (lane == 16 * s + t) ? p : R[4 * g + j];
            }
          }
        }
      }
    }
  }
  const unsigned int keep12 = tl < 12 ? 0xFFFFFFFFu : 0u;  // the last group's four registers hold words in lanes 0..11 only
  unsigned int qhead = 0, qtail = 0;                       // (wave-uniform)
  auto drain = [&](unsigned int count) {
    r4_drain_call(q_w, q_d, Zb[0], Zb[1], Zb[2], Zb[3], qhead, count);
    qhead += count;
  };
  const int max_blocks = (int)(((long long)S * H) / 100) + 64;  // (an acceptance below 64 % does not happen: a guard, not a limit)
  const unsigned int below = (1u << tl) - 1u;  // the lanes of my row before me
  const R4Consts K = r4_consts();
  for (int blk = 0; blk < max_blocks && __builtin_amdgcn_ballot_w64(live) != 0ull; ++blk) {
    // ---- twist: the next 624 words, in place (ascending groups: sources 6-7 groups ahead are still old, 3-4 back already new)
#pragma unroll
    for (int g = 0; g < 10; ++g) {
      // successor of this group's words j = 3: lane t + 1 of its register 0, lane 15: lane 0 of the next group's register 0
      // -- taken BEFORE register 0 is rewritten; the last group's word 623 is followed by the NEW word 0
      unsigned int t1;
      if (g < 9) t1 = r4_dppz<R4_SHL(1)>(R[4 * g]) | r4_dppz<R4_SHR(15)>(R[4 * g + 4]);
      else t1 = r4_dppz<R4_SHL(1)>(R[36]) | r4_dppz<R4_SHR(11)>(R[0]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned int x = r4_mix(K, R[4 * g + j], j < 3 ? R[4 * g + j + 1] : t1);
        const int js = (j + 1) & 3;  // the source word's register within its group
        if (g <= 3) {                // words <= 226: old word i + 397 = 6 groups + 3 lanes + 1 register on
          if (j < 3) {
            x ^= r4_dppz<R4_SHL(3)>(R[4 * (g + 6) + js]);
            if (g + 7 <= 9) x ^= r4_dppz<R4_SHR(13)>(R[4 * (g + 7) + js]);
          } else {
            x ^= r4_dppz<R4_SHL(4)>(R[4 * (g + 6) + js]);
            if (g + 7 <= 9) x ^= r4_dppz<R4_SHR(12)>(R[4 * (g + 7) + js]);
          }
        }
        if (g >= 3) {  // words >= 227: new word i - 227 = 3 groups + 9 lanes back, 1 register on
          if (j < 3) {
            x ^= r4_dppz<R4_SHR(9)>(R[4 * (g - 3) + js]);
            if (g >= 4) x ^= r4_dppz<R4_SHL(7)>(R[4 * (g - 4) + js]);
          } else {
            x ^= r4_dppz<R4_SHR(8)>(R[4 * (g - 3) + js]);
            if (g >= 4) x ^= r4_dppz<R4_SHL(8)>(R[4 * (g - 4) + js]);
          }
        }
        if (g == 9) x &= keep12;
        R[4 * g + j] = x;
      }
    }
    // ---- 156 polar attempts per stream: group g = attempts 16 g + t, words in registers 4 g .. 4 g + 3 of lane t
#pragma unroll
    for (int g = 0; g < 10; ++g) {
      const unsigned int ta = r4_temper3(K, R[4 * g]), tc = r4_temper3(K, R[4 * g + 2]);
      // x ~ (word - 2^31) / 2^31 (exactly: (a - 2^26) / 2^26 + b / 2^52 with a = word >> 5): r2 in units of 2^62
      const float xf = (float)(int)(ta ^ 0x80000000u), yf = (float)(int)(tc ^ 0x80000000u);
      const float rf = xf * xf + yf * yf;
      constexpr float kOne = 4611686018427387904.0f;  // 2^62
      // the accept mask of the 64 attempts as a scalar: accepted for certain, corrected where the estimate cannot decide
      // (within 5e-5 of 1, or tiny: r2 == 0 is rejected) by numpy's own comparison in double
      unsigned long long M = __builtin_amdgcn_ballot_w64(rf < kOne * (1.0f - 5e-5f));
      const bool grey = !(rf < kOne * (1.0f - 5e-5f) || rf > kOne * (1.0f + 5e-5f)) || rf < kOne * 1e-4f;
      const unsigned long long G = __builtin_amdgcn_ballot_w64(grey);
      if (G != 0ull) M = (M & ~G) | (__builtin_amdgcn_ballot_w64(r4_exact_accept(ta, R[4 * g + 1], tc, R[4 * g + 3])) & G);
      // (lanes 12..15 of the last group hold zero words: x = y = -1, r2 = 2, rejected without a mask)
      // accepted attempts of MY stream in this group (bits 16 s .. 16 s + 15 of the mask) and what they do to its position
      const unsigned int m16 = ((my_s & 2 ? (unsigned int)(M >> 32) : (unsigned int)M) >> (16 * (my_s & 1))) & 0xFFFFu;
      const int c = __popc(m16);
      const int rem0 = rem, u = rem0 - c;
      // the common case in ONE comparison per stream: the group lies strictly inside the columns that are not stored (pair
      // index >= Hs before it, the row not finished by it): rem0 <= D and u >= 1  <=>  (unsigned)(u - 1) < (unsigned)(D - c)
      const bool easy = SPARSE && (unsigned int)(u - 1) < (unsigned int)(D - c);
      rem = u;
      if (__builtin_amdgcn_ballot_w64(!easy) != 0ull) {  // (some stream of the wave is at its stored columns or at the end of a row)
        const int pp0 = H - rem0;     // pair index of my stream's first accepted attempt of the group within its row
        const int k = __popc(m16 & below);  // my rank among them
        int q = pp0 + k, r = row;
        if (q >= H) {
          q -= H;
          r += 1;
        }
        // stored pairs are [0, Hs) of every row (the group may run over the end of the row into the next one's)
        const bool push = live && ((m16 >> tl) & 1u) && q < Hs && r < S;
        const unsigned long long pm = __builtin_amdgcn_ballot_w64(push);
        if (push) {
          const unsigned int idx = __builtin_amdgcn_mbcnt_hi((unsigned int)(pm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)pm, 0u));
          const unsigned int slot = (qtail + idx) & (R4_QCAP - 1);
          q_w[slot] = make_uint4(ta, R[4 * g + 1], tc, R[4 * g + 3]);
          q_d[slot] = (unsigned int)(r * zc + 2 * q) | ((2 * q + 1 < zs) ? 0x10000000u : 0u) | ((unsigned int)my_s << 30);
        }
        qtail += (unsigned int)__popcll(pm);
        if (u <= 0) {  // my stream's row is complete
          rem = u + H;
          row += 1;
          if (row >= S) live = false;
        }
      }
      // sparse: half a block (80 attempts per stream) meets at most ONE run of stored columns per stream (runs are D >= 80
      // pairs apart): <= 4 Hs <= 192 new records + < 64 left over fit the queue, so it is evaluated twice per block
      if (SPARSE) {
        if (g == 4 || g == 9)
          while (qtail - qhead >= 64u) drain(64u);
      } else if (qtail - qhead >= 64u) {
        drain(64u);
      }
    }
  }
  if (qtail != qhead) drain(qtail - qhead);
}

// every edge of the launch fits k_mt_normals4's requirements (the host copy of the edge table decides, once per batch)
bool normals4_applies(const EdgeDev* h_edges, int B) {
  if (B < 1) return false;
  const int Lg = h_edges[0].Lg, S = h_edges[0].S, zc = h_edges[0].z_cols;
  if (Lg < 64 || (Lg & 1) || S < 1 || zc < 2 || (long long)S * zc >= (1ll << 28)) return false;
  for (int e = 1; e < B; ++e)
    if (h_edges[e].Lg != Lg || h_edges[e].S != S || h_edges[e].z_cols != zc) return false;
  return true;
}

int& gpet_opt_rng4() {
  static const int i_ = option_index("rng4");
  int& v = option_at(i_);
  return v;
}

hipError_t launch_normals4(hipStream_t st, EdgeDev* d_edges, int B, const unsigned int* d_seeds, int add_iter, int iter_abs,
                           int n_ahead, int z_store, int Lg, int S, int zc) {
  (void)hipGetLastError();
  const int zs = (z_store > 0 && z_store < zc) ? z_store : zc;
  const int streams = B * n_ahead;
  if (Lg / 2 - (zs + 1) / 2 >= 80 && zs <= 96)
    hipLaunchKernelGGL(k_mt_normals4<true>, dim3((streams + 3) / 4), dim3(64), 0, st, d_edges, d_seeds, add_iter, iter_abs,
                       n_ahead, streams, Lg, S, zc, zs);
  else
    hipLaunchKernelGGL(k_mt_normals4<false>, dim3((streams + 3) / 4), dim3(64), 0, st, d_edges, d_seeds, add_iter, iter_abs,
                       n_ahead, streams, Lg, S, zc, zs);
  return hipGetLastError();
}

}  // namespace gpet
