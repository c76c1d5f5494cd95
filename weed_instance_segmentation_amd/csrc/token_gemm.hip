// Token GEMM of the pixel-decoder encoder with its epilogue fused, on the fp32 matrix cores:
//     out (M, N) = epilogue( x (M, K) . W (N, K)^T + bias )
// for the three narrow Linears of Mask2FormerPixelDecoderEncoderMultiscaleDeformableAttention and the layer around it
// (transformers modeling_mask2former.py):
//     value_proj                  :978     N = 256          epilogue: bias
//     sampling_offsets | attention_weights (merged, :983-991)  N = 288   epilogue: bias
//     output_proj + residual + self_attn_layer_norm  :1012, :1076-1078   N = 256   epilogue: bias, + residual, LayerNorm
//     fc2 + residual + final_layer_norm              :1086-1088          K = 1024  same epilogue (+ the next layer's
//                                                                         `hidden + pos`, :972)
// M = B * S tokens (172 032 at config 2), K = 256 or 1024.  The library runs the K = 256 shapes at 86-89 TFLOP/s
// (profiles/r02_library_gemm_probe.jsonl) and leaves the residual + LayerNorm to a separate pass over the tokens.
//
// Layout of the work (MI355X: 256 CUs, 160 KiB LDS, v_mfma_f32_16x16x4_f32 = exact fp32):
//   * one persistent workgroup per CU: 8 compute waves + 1 loader wave.  A CU owns a contiguous range of 16-token column
//     tiles; its waves split that range as evenly as column tiles allow (172 032 tokens = 42 tiles per CU = 6,6,5,5,5,5,5,5:
//     every SIMD carries 10 or 11) and walk it CT tiles at a time (CT = 1: nine waves put three on one SIMD, i.e. at most
//     168 registers per wave, and two column tiles of accumulators alone are 128 / 144);
//   * a wave owns ALL N output features of its tokens -- N/16 row tiles x CT column tiles of accumulators (64 / 72
//     registers per column tile) -- so the LayerNorm of a token is a reduction inside the wave (4 lane groups x the lane's
//     registers);
//   * W is the A operand.  It does not fit LDS whole (256 KiB), so K is walked in phases of 64: the loader wave brings the
//     (N x 64) panel of the NEXT phase into the other half of a two-panel ring by LDS-DMA, already in MFMA fragment order
//     [row tile][super-step][lane][4] (a lane's 16 bytes are contiguous in its W row), while the compute waves work on the
//     current one; one workgroup barrier per phase.  The main loop's ds_read_b128 are lane-linear (conflict-free);
//   * x is the B operand, streamed HBM -> registers: lane (j, g) loads the 16 bytes x[token j][16 s + 4 g ..], which are
//     its B values for the 4 k-steps of super-step s (k order within a super-step: lane group g takes channels
//     16 s + 4 g + t at step t -- the same fixed permutation for A and B, as in mask_einsum.hip).  Every x element is
//     loaded by exactly one wave, one super-step ahead of its use.
// Roofline: fp32 MFMA, 2 M N K flop against 157.3 TFLOP/s.  HBM side: x once, out once (+ residual, + pos): 0.35-0.53 GB
// per call, far from binding.
//
// Where it stands (one MI355X, config-2 shapes, tools/kbench.py --only tg; profiles/r02_kbench_token_gemm.jsonl): 260 us for
// the 256 -> 256 projections (86.6 TFLOP/s, the library: 262 us), 280 us for the merged 288-wide one (library 292), 339 us
// with the residual + LayerNorm epilogue (library GEMM + the separate LayerNorm pass: 311), 992 us for fc2 + LayerNorm + pos
// (library 842: its K = 1024 kernel runs at 80 % of the peak).  Parity, not a win -- so the model keeps the library GEMMs by
// default (WM2F_TOKEN_GEMM=1 switches).  Timing ablations of the profiling build (WM2F_TG_MODE, outputs not valid): without
// the epilogue's loads and stores 231 us, without x loads 255, without LDS reads 261, without barriers 260, with none of
// them 219 -- against 151 us of pure MFMA issue at the 2.38 GHz the chip holds under this load (tools/probes/mfma_clock.hip
// measures 154.6 TFLOP/s for the same instruction mix on register operands): the epilogue (all waves of all CUs store at
// the same moments, the turns being in lock-step) and the turn structure around the MFMA stream are what is left, not the
// operand paths.  Next: stagger the waves' turns by a phase each so that one wave's epilogue runs under the others' MFMAs.
#include "common.h"
#include <stdlib.h>

namespace wm2f {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int kTgWaves = 8;                    // compute waves
constexpr int kTgThreads = (kTgWaves + 1) * 64;  // + the loader wave
constexpr int kPhaseK = 64;                    // K columns per panel: 4 super-steps of 16
constexpr unsigned kOob = 0x80000000u;

struct TgArgs {
  const float *x, *w, *bias, *residual, *gamma, *beta, *pos;
  float *out, *out_pos;
  int64_t M;
  int K, N, relu;
  int out_group;  // 0: out is (M, N) row-major; G > 0 (G % 4 == 0, N % G == 0): out is (N / G, M, G) -- feature group major
  int64_t pos_rows;
  float eps;
  int tiles_total;  // ceil(M / 16)
};

__device__ __forceinline__ void tg_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// NRT = N / 16 row tiles (16: N = 256, 18: N = 288); CT = column tiles (16 tokens each) a wave works on per turn
// MODE bits (profiling build only; outputs NOT valid): 1 = no x loads (B operand from registers), 2 = no LDS reads of W
// (A operand from registers), 4 = no barriers (the panels are read while they land), 8 = no epilogue loads / stores
template <int NRT, int CT, int MODE = 0>
__global__ __launch_bounds__(kTgThreads) void token_gemm_kernel(TgArgs a) {
  extern __shared__ __attribute__((aligned(16))) float panels[];  // [2][NRT][4][64][4], then bias | gamma | beta (N floats each)
  constexpr int kPanelFloats = NRT * 4 * 64 * 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_cu = gridDim.x, cu = blockIdx.x;
  // this CU's column tiles [t0, t1), then this wave's share of them
  const int t0 = (int)(((int64_t)a.tiles_total * cu) / n_cu), t1 = (int)(((int64_t)a.tiles_total * (cu + 1)) / n_cu);
  const int n_t = t1 - t0;
  const int n_phase = a.K / kPhaseK;
  // every wave runs the same number of turns (it takes part in every barrier): the largest share, CT tiles a turn
  const int share_max = (n_t + kTgWaves - 1) / kTgWaves;
  const int n_iter = (share_max + CT - 1) / CT;
  const int n_q = n_iter * n_phase;  // panel phases of this workgroup
  float* vec = panels + 2 * kPanelFloats;  // the epilogue's per-feature vectors: read from LDS (lgkmcnt), so that no load of
                                           // them sits in the vmcnt queue between the stores (the compiler answered that
                                           // with vmcnt(0) per row tile: store, wait, load, wait, ...)
  for (int i = tid; i < 3 * NRT * 16; i += kTgThreads) {
    const int which = i / (NRT * 16), f = i - which * (NRT * 16);
    const float* srcv = which == 0 ? a.bias : (which == 1 ? a.gamma : a.beta);
    vec[i] = srcv ? srcv[f] : 0.f;
  }
  __syncthreads();
  if (n_q == 0) return;

  if (wave == kTgWaves) {
    // ------------------------------------------------------------------ loader wave: W panels, fragment order, by LDS-DMA
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.N * a.K * 4, 0x00020000);
    const unsigned lane_off = (unsigned)(((lane & 15) * a.K + 4 * (lane >> 4)) * 4);  // row (lane & 15), 16 B of lane group g
    for (int q = 0; q < n_q; ++q) {
      const int p = q % n_phase;
      float* dst = panels + (q & 1) * kPanelFloats;
#pragma unroll 4
      for (int f = 0; f < NRT * 4; ++f) {
        const int rt = f >> 2, s = f & 3;
        const unsigned voff = lane_off + (unsigned)(rt * 16 * a.K * 4);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (lds_ptr_t)(dst + f * 256), 16, (int)voff, (p * kPhaseK + s * 16) * 4, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (!(MODE & 4)) tg_barrier();  // panel q is complete; the compute waves are done with panel q - 1 (its buffer is the next target)
    }
    return;
  }

  // -------------------------------------------------------------------- compute waves
  // share of this wave: tiles [w0, w1) of the CU's range, the first (n_t % 8) waves get one more -- waves w and w + 4 sit on
  // one SIMD, so the extras go to waves 0, 1, 2, 3 first (one per SIMD)
  const int base = n_t / kTgWaves, extra = n_t % kTgWaves;
  const int w0 = t0 + wave * base + (wave < extra ? wave : extra);
  const int w1 = w0 + base + (wave < extra ? 1 : 0);
  const int g = lane >> 4, j = lane & 15;
  const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)(a.M * a.K * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t o_rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, (int)(a.M * a.N * 4), 0x00020000);

  // x offsets of a turn's column tiles (this lane: token j of each tile, lane group g's 16 bytes)
  auto turn_offsets = [&](int it, unsigned (&xo)[CT]) {
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      const int64_t t = (int64_t)(w0 + CT * it + c) * 16 + j;
      xo[c] = (it < n_iter && w0 + CT * it + c < w1 && t < a.M) ? (unsigned)((t * a.K + 4 * g) * 4) : kOob;
    }
  };
  // B ring: 4 register sets, one per super-step of a phase; the loads run THREE super-steps ahead of the MFMAs that use
  // them (one super-step is 16 * NRT MFMAs = 2k cycles at CT = 1: a single one ahead does not cover a loaded HBM's latency),
  // and the ring runs on across phases and turns -- the next turn's first loads fly under this turn's last phase and epilogue.
  f32x4 bq[4][CT];
  unsigned xo_cur[CT], xo_nxt[CT];
  turn_offsets(0, xo_cur);
  auto load_b = [&](f32x4 (&dst)[CT], const unsigned (&xo)[CT], int kk) {  // kk = first K column of the super-step
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      if (MODE & 1) dst[c] = (f32x4){__uint_as_float(xo[c]), (float)kk, 1.f, 2.f};
      else dst[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, xo[c], kk * 4, 0));
    }
  };
  load_b(bq[0], xo_cur, 0);
  load_b(bq[1], xo_cur, 16);
  load_b(bq[2], xo_cur, 32);

  int q = 0;
  for (int it = 0; it < n_iter; ++it) {
    const int ct0 = w0 + CT * it;              // first column tile of this turn; tiles >= w1 are empty (all-zero B, no stores)
    bool live[CT];
    int64_t tok[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      live[c] = ct0 + c < w1;
      tok[c] = (int64_t)(ct0 + c) * 16 + j;
    }
    turn_offsets(it + 1, xo_nxt);
    f32x4 acc[NRT][CT];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
      for (int c = 0; c < CT; ++c) acc[rt][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int p = 0; p < n_phase; ++p, ++q) {
      if (!(MODE & 4)) tg_barrier();  // panel q has landed (the loader's barrier q)
      const float* panel = panels + (q & 1) * kPanelFloats;
      const int kk = p * kPhaseK;
      const bool last = p + 1 == n_phase;
      // where the three loads issued during this phase come from: the next phase of this turn, or -- branch-free, so that
      // the compiler keeps COUNTED vmcnt waits (a control-flow merge made it wait for vmcnt(0)) -- the next turn's tiles
      unsigned xs[CT];
#pragma unroll
      for (int c = 0; c < CT; ++c) xs[c] = last ? xo_nxt[c] : xo_cur[c];
      const int k_next = last ? 0 : kk + kPhaseK;
      // One stream of (super-step, row-tile pair) groups through the panel: 8 * CT MFMAs per group.  The A fragments of group
      // G + 1 are read from LDS before the MFMAs of group G issue (two register sets): left to itself the compiler emitted
      // read, wait, 8 MFMAs, read, wait ... and every group paid the LDS latency.
      constexpr int NG = NRT / 2, kGroups = 4 * NG;
      f32x4 av[2][2];
      auto read_a = [&](f32x4 (&dst)[2], int G) {
        const int s = G / NG, r2 = (G % NG) * 2;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (MODE & 2) dst[u] = (f32x4){(float)(r2 + u), (float)s, (float)lane, 1.f};
          else dst[u] = *reinterpret_cast<const f32x4*>(panel + (((r2 + u) * 4 + s) * 64 + lane) * 4);
        }
      };
      if (!live[0]) {
        // a wave whose share is used up still meets every barrier, but issues no MFMA on its empty tile (the matrix pipe
        // is its SIMD partner's); its B ring keeps turning so that the load count per phase stays the same
        load_b(bq[3], xo_cur, kk + 48);
        load_b(bq[0], xs, k_next);
        load_b(bq[1], xs, k_next + 16);
        load_b(bq[2], xs, k_next + 32);
        continue;
      }
      read_a(av[0], 0);
#pragma unroll
      for (int G = 0; G < kGroups; ++G) {
        const int s = G / NG, r2 = (G % NG) * 2;
        if (G + 1 < kGroups) read_a(av[(G + 1) & 1], G + 1);
        if (G % NG == 0) {  // entering super-step s: request the B operand of super-step s + 3 into ring set (s + 3) % 4
          if (s == 0) load_b(bq[3], xo_cur, kk + 48);
          else load_b(bq[s - 1], xs, k_next + (s - 1) * 16);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int c = 0; c < CT; ++c)
              acc[r2 + u][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[G & 1][u][t], bq[s][c][t], acc[r2 + u][c], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int c = 0; c < CT; ++c) xo_cur[c] = xo_nxt[c];

    // ---- epilogue.  Lane (j, g) holds, per row tile rt and column tile c, features rt*16 + 4g .. +3 of token (ct0 + c)*16 + j.
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      if (!live[c]) continue;  // wave-uniform
      if (MODE & 8) {  // ablation: no epilogue traffic (one store keeps the sums alive)
        f32x4 v = acc[0][c];
#pragma unroll
        for (int rt = 1; rt < NRT; ++rt) v += acc[rt][c];
        if (v[0] == 12345.f) a.out[0] = v[1] + v[2] + v[3];
        continue;
      }
      const int64_t tk = tok[c];
      const bool tok_ok = tk < a.M;
      const unsigned row_o = tok_ok ? (unsigned)(tk * a.N * 4) : kOob;
      constexpr int NF = NRT * 16;
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt) {
        f32x4 v = acc[rt][c] + *reinterpret_cast<const f32x4*>(vec + rt * 16 + 4 * g);
        if (a.relu) v = __builtin_elementwise_max(v, (f32x4){0.f, 0.f, 0.f, 0.f});
        acc[rt][c] = v;
      }
      if (a.gamma) {  // LayerNorm over the token's N features: this lane's NRT * 4 values, then the 4 lane groups
        constexpr int kEB = NRT % 4 == 0 ? 4 : 3;  // row tiles per batch of epilogue loads (16 / 12 registers: the kernel sits at its 168-register cap)
        if (a.residual) {  // loads in batches; the accumulators hold the sums
          const __amdgpu_buffer_rsrc_t r_rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.residual, 0, (int)(a.M * a.N * 4), 0x00020000);
#pragma unroll
          for (int h0 = 0; h0 < NRT; h0 += kEB) {
            f32x4 rv[kEB];
#pragma unroll
            for (int i = 0; i < kEB; ++i)
              rv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rs, row_o + (unsigned)(((h0 + i) * 16 + 4 * g) * 4), 0, 0));
#pragma unroll
            for (int i = 0; i < kEB; ++i) acc[h0 + i][c] += rv[i];
          }
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) {
          const f32x4 v = acc[rt][c];
          s1 += (v[0] + v[1]) + (v[2] + v[3]);
          s2 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        }
        s1 += __shfl_xor(s1, 16, 64);
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 16, 64);
        s2 += __shfl_xor(s2, 32, 64);
        const float inv_n = 1.f / (float)a.N;
        const float mean = s1 * inv_n;
        float var = s2 * inv_n - mean * mean;
        var = var < 0.f ? 0.f : var;
        const float rstd = rsqrtf(var + a.eps);
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) {
          const int f0 = rt * 16 + 4 * g;
          const f32x4 gm = *reinterpret_cast<const f32x4*>(vec + NF + f0), bt = *reinterpret_cast<const f32x4*>(vec + 2 * NF + f0);
          acc[rt][c] = (acc[rt][c] - mean) * rstd * gm + bt;
        }
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[rt][c]), o_rs, row_o + (unsigned)((rt * 16 + 4 * g) * 4), 0, 0);
        if (a.out_pos) {  // the next layer's hidden + pos: pos rows in two batches, then the stores
          const __amdgpu_buffer_rsrc_t p_rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.pos, 0, (int)(a.pos_rows * a.N * 4), 0x00020000);
          const __amdgpu_buffer_rsrc_t q_rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.out_pos, 0, (int)(a.M * a.N * 4), 0x00020000);
          const unsigned prow = tok_ok ? (unsigned)((tk % a.pos_rows) * a.N * 4) : kOob;
#pragma unroll
          for (int h0 = 0; h0 < NRT; h0 += kEB) {
            f32x4 pv[kEB];
#pragma unroll
            for (int i = 0; i < kEB; ++i)
              pv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(p_rs, prow + (unsigned)(((h0 + i) * 16 + 4 * g) * 4), 0, 0));
#pragma unroll
            for (int i = 0; i < kEB; ++i)
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[h0 + i][c] + pv[i]), q_rs,
                                                     row_o + (unsigned)(((h0 + i) * 16 + 4 * g) * 4), 0, 0);
          }
        }
      } else if (a.out_group > 0) {
        // group-major output (N / G, M, G): K1's operand rows head-major, so that a head's 144 bytes of consecutive tokens
        // are contiguous.  A lane's 4 consecutive features stay inside one group (G % 4 == 0).
        const unsigned G = (unsigned)a.out_group;
        const unsigned grp_bytes = (unsigned)(a.M * G * 4);
        const unsigned tok_o = tok_ok ? (unsigned)(tk * G * 4) : kOob;
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) {
          const unsigned f0 = (unsigned)(rt * 16 + 4 * g);
          const unsigned grp = f0 / G;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[rt][c]), o_rs, tok_o + grp * grp_bytes + (f0 - grp * G) * 4, 0, 0);
        }
      } else {
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[rt][c]), o_rs, row_o + (unsigned)((rt * 16 + 4 * g) * 4), 0, 0);
      }
    }
  }
}

}  // namespace
}  // namespace wm2f

using namespace wm2f;

extern "C" int wm2f_token_linear_fwd(const void* x, const void* w, const void* bias, const void* residual, const void* ln_gamma,
                                     const void* ln_beta, const void* pos, void* out, void* out_plus_pos, int64_t M, int K, int N,
                                     int relu, int64_t pos_rows, float eps, int out_group, void* stream) {
  const char* who = "wm2f_token_linear_fwd";
  WM2F_REQUIRE(x && w && bias && out, "%s: null pointer", who);
  WM2F_REQUIRE(M > 0 && K > 0 && N > 0, "%s: non-positive size", who);
  WM2F_REQUIRE(N == 256 || N == 288, "%s: N = %d is not built (256 and 288 are: the widths of the pixel decoder's narrow Linears)", who, N);
  WM2F_REQUIRE(K % kPhaseK == 0, "%s: K = %d must be a multiple of %d", who, K, kPhaseK);
  WM2F_REQUIRE(M * (int64_t)K * 4 < (1ll << 31) && M * (int64_t)N * 4 < (1ll << 31), "%s: x / out must stay below 2 GiB (32-bit buffer offsets)", who);
  WM2F_REQUIRE(pos_rows * (int64_t)N * 4 < (1ll << 31), "%s: pos must stay below 2 GiB", who);
  WM2F_REQUIRE((ln_gamma == nullptr) == (ln_beta == nullptr), "%s: LayerNorm needs both gamma and beta", who);
  WM2F_REQUIRE(!residual || ln_gamma, "%s: the residual belongs to the LayerNorm epilogue", who);
  WM2F_REQUIRE(!out_plus_pos || (pos && ln_gamma && pos_rows > 0), "%s: out_plus_pos needs pos, pos_rows and the LayerNorm epilogue", who);
  WM2F_REQUIRE(out_group == 0 || (out_group > 0 && out_group % 4 == 0 && N % out_group == 0 && !ln_gamma),
               "%s: out_group = %d must divide N, be a multiple of 4 and exclude the LayerNorm epilogue", who, out_group);
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      set_error("%s: cannot query the device", who);
      return WM2F_ELAUNCH;
    }
    n_cu = prop.multiProcessorCount;
  }
  TgArgs a;
  a.x = (const float*)x;
  a.w = (const float*)w;
  a.bias = (const float*)bias;
  a.residual = (const float*)residual;
  a.gamma = (const float*)ln_gamma;
  a.beta = (const float*)ln_beta;
  a.pos = (const float*)pos;
  a.out = (float*)out;
  a.out_pos = (float*)out_plus_pos;
  a.M = M;
  a.K = K;
  a.N = N;
  a.relu = relu;
  a.out_group = out_group;
  a.pos_rows = pos_rows;
  a.eps = eps;
  a.tiles_total = (int)ceil_div64(M, 16);
  int grid = n_cu;
  if (grid > a.tiles_total) grid = a.tiles_total;
  const size_t lds = (size_t)2 * (N / 16) * 4 * 64 * 4 * sizeof(float) + (size_t)3 * N * sizeof(float);
  auto kfn = N == 256 ? token_gemm_kernel<16, 1> : token_gemm_kernel<18, 1>;
#ifdef WM2F_PROFILING
  {  // timing ablations (outputs not valid): WM2F_TG_MODE = 1 no x loads, 2 no LDS reads, 3 neither, 4 no barriers, 7 all
    const char* e = getenv("WM2F_TG_MODE");
    const int mode = e ? atoi(e) : 0;
    if (N == 256 && mode == 1) kfn = token_gemm_kernel<16, 1, 1>;
    if (N == 256 && mode == 2) kfn = token_gemm_kernel<16, 1, 2>;
    if (N == 256 && mode == 3) kfn = token_gemm_kernel<16, 1, 3>;
    if (N == 256 && mode == 4) kfn = token_gemm_kernel<16, 1, 4>;
    if (N == 256 && mode == 7) kfn = token_gemm_kernel<16, 1, 7>;
    if (N == 256 && mode == 8) kfn = token_gemm_kernel<16, 1, 8>;
    if (N == 256 && mode == 15) kfn = token_gemm_kernel<16, 1, 15>;
  }
#endif
  hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) {
    set_error("%s: cannot raise dynamic LDS to %zu: %s", who, lds, hipGetErrorString(e));
    return WM2F_ELAUNCH;
  }
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(kTgThreads), lds, (hipStream_t)stream, a);
  WM2F_CHECK_LAUNCH(who);
  return WM2F_OK;
}
