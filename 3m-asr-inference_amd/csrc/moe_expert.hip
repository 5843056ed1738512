// Grouped 32-expert FFN  Y = SiLU(X W1[e]^T + b1[e]) W2[e]^T + b2[e]  for all experts in ONE launch.
//
// Replaces compute_fmoe_expert's host loop (TRTAPI++/plugin/fmoe_expert_plugin/fmoe_expert_plugin.cpp:36-142):
// D2H copy of the histogram + stream sync (:75-78), <=4*E launches on 8 streams (cublasSgemm,
// BiasSiluKernel, cublasSgemm, BiasKernel; fmoe_expert_kernel.cu:130-189) and 8 host syncs (:130).
// Here the per-expert row ranges are read on the device from acc_hist, so there is no host
// round trip and the op is graph-capturable.
//
// Work item = (expert e, 64-wide slice of the hidden dimension).  The workgroup
//   1. gathers its expert's token rows x[pos[acc[e]+i]] into LDS (the local_scatter is fused away),
//   2. phase 1: H[:, slice] = SiLU(X . W1[e][slice,:]^T + b1)   -- each wave owns 16 hidden units and
//      streams their W1 rows (2 KB each) straight into VGPRs, A fragments come from LDS,
//   3. keeps H in LDS, phase 2: Ypart = H[:, slice] . W2[e][:, slice]^T -- each wave owns D/4 output
//      columns and streams the 256-B row pieces of W2,
//   4. writes Ypart to slab[slice][sorted row][D]; moe_combine_kernel sums the F/64 slabs in a fixed
//      order (bitwise reproducible, no float atomics), adds b2, applies gate * ff_scale, the
//      residual and optionally the block's final LayerNorm, and un-permutes rows (local_gather).
// Arithmetic: v_mfma_f32_16x16x4_f32 (exact fp32).  At B=1x206 (S=50, ~2 rows/expert) the launch is
// pure weight streaming: 4.2 MB per touched expert, E*F/64 = 512 workgroups of 256 KB each.
// Measured dead ends: a persistent work-queue form (atomic item counter) was 30 % slower,
// 32-wide slices and 3-4 deep load rings did not move the time: the kernel sits at the ~24 GB/s a CU can pull.
#include "common.h"
#include "kernels.h"
#include "moe_gate.h"

namespace m3 {

// Self-routing form (RE = number of experts > 0; S <= 64 * waves rows, all experts local): the work-group derives its expert's
// rows from the router logits itself -- SoftmaxTopK (softmax_topk_kernel.cu:26-120, the reference's arg-max tree) on one lane
// per row, one ballot for "routed to my expert", the set bits in lane order ARE the stable order of ScatterMapping
// (fmoe_expert_kernel.cu:25-90) -- so the single-work-group index launch in front of it disappears (5.2 us x 18 layers at
// B = 1).  The work-groups of slice 0 leave gate_idx / gate_value / mapping / acc_histogram / pos behind (the combine kernel
// reads the first two, the rest are the reference's taps): acc[e] = #rows routed below e is one more ballot.  Partial outputs
// go to slab[slice][ORIGINAL row] (no un-permute in the combine) and slice 0 adds b2[e].
struct ExpertRoute {
  const float* logits = nullptr;            // [S][RE] router logits
  const int32_t* row_len = nullptr; int rows_per_batch = 0;   // frame t of utterance b is padding when t >= row_len[b]
  int32_t* gate_idx = nullptr; float* gate_value = nullptr;
  int32_t* mapping = nullptr; int32_t* acc_hist = nullptr; int32_t* pos = nullptr;
  const float* b2 = nullptr;                // [E][D], added by slice 0
};

// LNS: apply the layer's LayerNorm while gathering rows (fused-route engines); a separate instantiation so the
// default path does not carry its registers.
template <int MT, bool LNS, int RE = 0>
__global__ __launch_bounds__(64 * (kExpertSlice / 16)) void expert_ffn_f32_kernel(const float* __restrict__ x, int ldx,
                                                             const int32_t* __restrict__ pos,
                                                             const int32_t* __restrict__ acc_hist, int S, int D,
                                                             int F, const float* __restrict__ w1,
                                                             const float* __restrict__ b1,
                                                             const float* __restrict__ w2, int w2_row_stride,
                                                             int w2_slice_stride, float* __restrict__ slab,
                                                             const float* __restrict__ ln_gamma,
                                                             const float* __restrict__ ln_beta, float ln_eps, ExpertRoute rt) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int e = blockIdx.y, slice = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, kq = lane >> 4;

  const int xs_ld = D + 8;                   // +8 floats: conflict-free ds_read_b128 of A fragments
  constexpr int hs_ld = kExpertSlice + 8;
  constexpr int NWV = kExpertSlice / 16;     // waves per workgroup, 16 hidden units each
  constexpr int KS2 = kExpertSlice / 16;     // 16-deep k-steps of phase 2
  constexpr int SPG = 8 / KS2;               // phase-2 output tiles per 8-float4 load group
  constexpr int NB = 2;   // load groups in the register ring
  static_assert(kExpertSlice == 16 || kExpertSlice == 32 || kExpertSlice == 64, "slice must be 16/32/64");
  float* xs = lds;                           // [16*MT][D+8]
  float* hs = lds + 16 * MT * xs_ld;         // [16*MT][64+8]
  const int f0 = slice * kExpertSlice;
  const int ksteps1 = D >> 4;

  const float* w1row = w1 + ((size_t)e * F + f0 + 16 * wave + col) * D + 4 * kq;
  const float bias1 = b1[(size_t)e * F + f0 + 16 * wave + col];
  const int nsub = D >> 4;                   // 16-column output tiles of phase 2
  // W2[e][:, slice]: reference layout [E][D][F] (row stride F, slices 64 floats apart) or the plan's
  // slice-major repack [E][F/64][D][64] (row stride 64: the workgroup's 128 KB are contiguous)
  const float* w2_slice = w2 + (size_t)e * D * F + (size_t)slice * w2_slice_stride;

  // One stream of weight loads per wave: g1 groups of 8 W1 k-steps, then g2 groups of 2 W2 tiles
  // (8 float4 each), double-buffered so 8-16 float4 per lane stay in flight across the phase change.
  const int g1 = (ksteps1 + 7) >> 3;
  const int g2 = ((nsub + NWV - 1) / NWV + SPG - 1) / SPG;   // same for every wave (the barrier sits inside the loop)
  const int total = g1 + g2;

  f32x4 wb[NB][8];
  float bb[NB][SPG];      // self-routing, slice 0: b2 of the group's output tiles travels with their weights (no extra wait)
  const bool add_b2 = RE > 0 && slice == 0 && rt.b2 != nullptr;
  auto load_group = [&](int g, int buf) {      // buf is a compile-time constant at every call site
    if (g < g1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int s = min(8 * g + i, ksteps1 - 1);   // clamped, not branched: loads stay back to back
        wb[buf][i] = ldg4_w(w1row + (s << 4));
      }
    } else {
#pragma unroll
      for (int j = 0; j < SPG; ++j) {
        const int sub = min(wave + NWV * (SPG * (g - g1) + j), nsub - 1);
        const float* p = w2_slice + (size_t)(16 * sub + col) * w2_row_stride + 4 * kq;
#pragma unroll
        for (int st = 0; st < KS2; ++st) wb[buf][KS2 * j + st] = ldg4_w(p + 16 * st);
        if (RE > 0) bb[buf][j] = add_b2 ? rt.b2[(size_t)e * D + 16 * sub + col] : 0.f;
      }
    }
  };
  // self-routing: the row's logits are requested first.  (Requesting the first weight group here as well, before the routing is
  // known, was measured and LOSES: expert launch 32.1 vs 30.6 us in situ, 214 k vs 220 k frames/s -- profiles/r04_ab_self_route.txt)
  constexpr int REW = RE > 0 ? RE : 8;
  f32x4 lrow[REW / 4];
  const int rt_r = 64 * wave + lane;
  bool rt_live = false;
  if (RE > 0) {
    rt_live = rt_r < S && (rt.row_len == nullptr || (rt_r % rt.rows_per_batch) < rt.row_len[rt_r / rt.rows_per_batch]);
    const float* lp = rt.logits + (size_t)min(rt_r, S - 1) * RE;
#pragma unroll
    for (int j = 0; j < REW / 4; ++j) lrow[j] = ldg4(lp + 4 * j);
  }
  int row_lo, row_hi;
  __shared__ int32_t route_rows[RE > 0 ? 64 * (kExpertSlice / 16) : 1];   // self-routing: my expert's rows in stable order
  __shared__ int route_cnt[RE > 0 ? 2 * (kExpertSlice / 16) : 1];
  if (RE > 0) {
    constexpr int NWR = kExpertSlice / 16;
    const int r = rt_r;
    const bool live = rt_live;
    int gi = -1;
    float gv = 0.f;
    gate_top1_regs<REW>(lrow, &gi, &gv, slice == 0);   // (the gate value is a tap of slice 0 only)
    if (!live) { gi = -1; gv = 0.f; }
    const unsigned long long mine = __ballot(gi == e), below = __ballot(live && gi < e);
    if (lane == 0) {
      route_cnt[wave] = __popcll(mine);
      route_cnt[NWR + wave] = __popcll(below);
    }
    __syncthreads();
    int off = 0, n_e = 0, acc_e = 0;
#pragma unroll
    for (int w = 0; w < NWR; ++w) {
      if (w < wave) off += route_cnt[w];
      n_e += route_cnt[w];
      acc_e += route_cnt[NWR + w];
    }
    const int rank = off + __popcll(mine & ((1ull << lane) - 1ull));
    if (gi == e) route_rows[rank] = r;
    if (slice == 0) {      // what the index launch used to leave behind
      if (gi == e) {
        rt.gate_idx[r] = gi;
        rt.gate_value[r] = gv;
        if (rt.mapping) rt.mapping[r] = acc_e + rank;
        if (rt.pos) rt.pos[acc_e + rank] = r;
      } else if (e == 0 && r < S && !live) {
        rt.gate_idx[r] = -1;
        rt.gate_value[r] = 0.f;
        if (rt.mapping) rt.mapping[r] = -1;
      }
      if (rt.acc_hist && threadIdx.x == 0) {
        rt.acc_hist[e] = acc_e;
        if (e == RE - 1) rt.acc_hist[RE] = acc_e + n_e;
      }
    }
    row_lo = 0;
    row_hi = n_e;          // (route_rows is complete behind the tile loop's first barrier)
  } else {
    row_lo = acc_hist[e];
    row_hi = acc_hist[e + 1];
  }
  if (row_hi <= row_lo) return;  // empty expert: nothing (more) streamed

  // an expert's row tiles are spread over blockIdx.z (long batches, unbalanced routing: no serial tile loop)
  for (int r0 = row_lo + 16 * MT * blockIdx.z; r0 < row_hi; r0 += 16 * MT * gridDim.z) {
    const int nrows = min(16 * MT, row_hi - r0);
    float* slab_base = slab + ((size_t)slice * S + (RE > 0 ? 0 : r0)) * D;   // self-routing: rows land at their ORIGINAL index

    load_group(0, 0);   // issued before the X staging so HBM latency overlaps it

    // ---- gather token rows into LDS (fused local_scatter) ----
    __syncthreads();    // previous row tile is done with xs / hs
    for (int i = wave; i < 16 * MT; i += NWV) {
      float* dst = xs + i * xs_ld;
      if (i < nrows) {
        const float* src = x + (size_t)(RE > 0 ? route_rows[r0 + i] : pos[r0 + i]) * ldx;
        if (LNS) {
          // the layer's LayerNorm (norm_ff) applied on the fly: x is the raw residual stream and the normalised
          // MoE input never exists in memory (two-pass statistics; the row is L1/L2-resident)
          f32x4 v[8];                              // the row stays in registers: one memory round trip (D <= 2048)
          float s = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int c = lane * 4 + 256 * j;
            const f32x4 t = ldg4(src + min(c, D - 4));
            const bool in = c < D;
            v[j] = f32x4{in ? t[0] : 0.f, in ? t[1] : 0.f, in ? t[2] : 0.f, in ? t[3] : 0.f};
            s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
          }
          const float mean = wave_sum(s) / (float)D;
          float q = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float on = (lane * 4 + 256 * j < D) ? 1.f : 0.f;
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
              const float d = (v[j][e4] - mean) * on;
              q += d * d;
            }
          }
          const float rstd = rsqrtf(wave_sum(q) / (float)D + ln_eps);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int c = lane * 4 + 256 * j;
            if (c < D) {
              const f32x4 g = ldg4(ln_gamma + c), be = ldg4(ln_beta + c);
              f32x4 o;
#pragma unroll
              for (int e4 = 0; e4 < 4; ++e4) o[e4] = (v[j][e4] - mean) * rstd * g[e4] + be[e4];
              stg4(dst + c, o);
            }
          }
        } else {
          for (int c = lane * 4; c < D; c += 256) stg4(dst + c, ldg4(src + c));
        }
      } else {
        for (int c = lane * 4; c < D; c += 256) stg4(dst + c, f32x4{0.f, 0.f, 0.f, 0.f});
      }
    }
    __syncthreads();

    f32x4 acc1[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc1[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 hfrag[MT][KS2];
    int orow[MT][4];     // self-routing: output row of accumulator row (mt, r)
    if (RE > 0) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) orow[mt][r] = route_rows[min(r0 + 16 * mt + 4 * kq + r, row_hi - 1)];
    }

    // phase change: H = SiLU(acc1 + b1) -> LDS -> A fragments of phase 2
    auto transition = [&]() {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          hs[(16 * mt + 4 * kq + r) * hs_ld + 16 * wave + col] = silu(acc1[mt][r] + bias1);
      __syncthreads();
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int st = 0; st < KS2; ++st)
          hfrag[mt][st] = *reinterpret_cast<const f32x4*>(hs + (16 * mt + col) * hs_ld + 16 * st + 4 * kq);
    };
    auto compute = [&](int g, int buf) {
      if (g < g1) {   // phase 1: H[:, f0+16w .. +16) += X[:, 16s..] . W1^T
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int s = 8 * g + i;
          if (s < ksteps1) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              const f32x4 a = *reinterpret_cast<const f32x4*>(xs + (16 * mt + col) * xs_ld + (s << 4) + 4 * kq);
#pragma unroll
              for (int j = 0; j < 4; ++j) acc1[mt] = mfma16(a[j], wb[buf][i][j], acc1[mt]);
            }
          }
        }
      } else {        // phase 2: Ypart[:, 16*sub .. +16) = H[:, slice] . W2[e][:, slice]^T
#pragma unroll
        for (int j = 0; j < SPG; ++j) {
          const int sub = wave + NWV * (SPG * (g - g1) + j);
          if (sub < nsub) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              f32x4 acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
              const float bias2 = RE > 0 ? bb[buf][j] : 0.f;
#pragma unroll
              for (int st = 0; st < KS2; ++st)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) acc2 = mfma16(hfrag[mt][st][jj], wb[buf][KS2 * j + st][jj], acc2);
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int i = 16 * mt + 4 * kq + r;
                if (RE > 0) {
                  if (i < nrows) slab_base[(size_t)orow[mt][r] * D + 16 * sub + col] = acc2[r] + bias2;
                } else {
                  if (i < nrows) slab_base[(size_t)i * D + 16 * sub + col] = acc2[r];
                }
              }
            }
          }
        }
      }
    };

    // ring of NB groups: while group g is consumed, groups g+1 .. g+NB-1 are in flight (8*(NB-1) float4 per lane)
#pragma unroll
    for (int b = 1; b < NB - 1; ++b)
      if (b < total) load_group(b, b);
    for (int g0 = 0; g0 < total; g0 += NB) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int g = g0 + b;
        if (g < total) {
          if (g + NB - 1 < total) load_group(g + NB - 1, (b + NB - 1) % NB);
          if (g == g1) transition();
          compute(g, b);
        }
      }
    }
  }
}

// Opt the big-tile variants into > 64 KB of dynamic LDS once (not a stream operation: must not
// happen inside a hipGraph capture, so the engine calls this at prepare time).
int init_expert_ffn_kernels() {
  static PerDeviceOnce once;
  if (once.done()) return 0;
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)expert_ffn_f32_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)expert_ffn_f32_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)expert_ffn_f32_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)expert_ffn_f32_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  once.mark();
  return 0;
}

size_t expert_ffn_slab_bytes(int S, int D, int F) {
  return (size_t)(F / kExpertSlice) * (size_t)S * (size_t)D * sizeof(float);
}

int launch_expert_ffn_f32(const float* x, int ldx, const int32_t* pos, const int32_t* acc_hist, int S, int E, int D,
                          int F, const float* w1, const float* b1, const float* w2, int w2_sliced, float* slab,
                          const float* ln_gamma, const float* ln_beta, float ln_eps, hipStream_t stream) {
  M3_REQUIRE(S > 0 && E > 0, "expert_ffn: empty problem S=%d E=%d", S, E);
  M3_REQUIRE((D & 15) == 0 && D <= 2048, "expert_ffn: idim=%d must be a multiple of 16 (<=2048)", D);
  M3_REQUIRE(F % kExpertSlice == 0, "expert_ffn: hidden_units=%d must be a multiple of %d", F, kExpertSlice);
  M3_REQUIRE((ldx & 3) == 0, "expert_ffn: ldx=%d must be a multiple of 4", ldx);
  if (ln_gamma == nullptr && expert_ffn_f32_tiled(S, E, D, F))   // long batches: two grouped LDS-tiled GEMMs
    return launch_expert_ffn_f32_tiled(x, ldx, pos, acc_hist, S, E, D, F, w1, b1, w2, w2_sliced, slab,
                                       expert_ffn_f32_rows(slab, S, E, D, F), stream);
  // rows per tile: small batches keep LDS small (more workgroups per CU -> more bytes in flight)
  const int mt = S <= 64 ? 1 : (S <= 512 ? 2 : 4);
  const size_t lds_bytes = (size_t)16 * mt * ((D + 8) + (kExpertSlice + 8)) * sizeof(float);
  M3_REQUIRE(lds_bytes <= 160 * 1024, "expert_ffn: LDS tile of %zu bytes does not fit", lds_bytes);
  int zt = cdiv(S, 16 * mt);
  dim3 grid(F / kExpertSlice, E, zt < 8 ? zt : 8);
  const int w2_row_stride = w2_sliced ? kExpertSlice : F;
  const int w2_slice_stride = w2_sliced ? D * kExpertSlice : kExpertSlice;
  if (int rc = init_expert_ffn_kernels()) return rc;
#define M3_EXPERT_CASE2(MT_, LNS_)                                                                      \
  hipLaunchKernelGGL((expert_ffn_f32_kernel<MT_, LNS_>), grid, dim3(64 * (kExpertSlice / 16)), lds_bytes, stream, x, ldx, pos,    \
                     acc_hist, S, D, F, w1, b1, w2, w2_row_stride, w2_slice_stride, slab, ln_gamma, ln_beta, ln_eps, ExpertRoute())
#define M3_EXPERT_CASE(MT_) do { if (ln_gamma) M3_EXPERT_CASE2(MT_, true); else M3_EXPERT_CASE2(MT_, false); } while (0)
  if (mt == 1) M3_EXPERT_CASE(1); else if (mt == 2) M3_EXPERT_CASE(2); else M3_EXPERT_CASE(4);
#undef M3_EXPERT_CASE
#undef M3_EXPERT_CASE2
  M3_LAUNCH_CHECK();
  return 0;
}

// Self-routing launch (see ExpertRoute): SoftmaxTopK + ScatterMapping + grouped expert FFN in one launch for S <= 256 rows.
// slab [F/64][S][D] receives the partial outputs at ORIGINAL rows with b2 already added by slice 0: combine with
// mapping = NULL, b2 = NULL.  Bit-identical to launch_moe_gate_index + launch_expert_ffn_f32 + combine(mapping, b2).
bool expert_ffn_f32_self_routing(int S, int E) { return S >= 1 && S <= 64 * (kExpertSlice / 16) && (E == 8 || E == 16 || E == 32 || E == 64); }

int launch_expert_route_ffn_f32(const float* x, int ldx, const float* logits, const int32_t* row_len, int rows_per_batch, int S, int E,
                                int D, int F, const float* w1, const float* b1, const float* w2, int w2_sliced, const float* b2,
                                float* slab, int32_t* gate_idx, float* gate_value, int32_t* mapping, int32_t* acc_hist, int32_t* pos,
                                hipStream_t stream, const float* ln_gamma, const float* ln_beta, float ln_eps) {
  // ln_gamma != null: x is the RAW residual stream and the layer's LayerNorm is applied while the rows are gathered (the split-route
  // engines: the normalised MoE input is never written)
  M3_REQUIRE(expert_ffn_f32_self_routing(S, E), "expert_route_ffn: needs 1 <= S <= %d rows and 8/16/32/64 experts (S=%d E=%d)", 64 * (kExpertSlice / 16), S, E);
  M3_REQUIRE((D & 15) == 0 && D <= 2048 && F % kExpertSlice == 0 && (ldx & 3) == 0, "expert_route_ffn: bad dims D=%d F=%d ldx=%d", D, F, ldx);
  M3_REQUIRE(logits && gate_idx && gate_value && slab, "expert_route_ffn: null pointer");
  M3_REQUIRE(row_len == nullptr || rows_per_batch > 0, "expert_route_ffn: rows_per_batch missing");
  const int mt = S <= 64 ? 1 : 2;               // (an expert with more rows than a tile walks its tiles in turn)
  const size_t lds_bytes = (size_t)16 * mt * ((D + 8) + (kExpertSlice + 8)) * sizeof(float);
  M3_REQUIRE(lds_bytes <= 150 * 1024, "expert_route_ffn: LDS tile of %zu bytes does not fit", lds_bytes);
  static PerDeviceOnce once;
  if (!once.done()) {
#define M3_ROUTE_ATTR(MT_, E_)                                                                                                                            \
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)expert_ffn_f32_kernel<MT_, false, E_>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); \
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)expert_ffn_f32_kernel<MT_, true, E_>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024))
    M3_ROUTE_ATTR(2, 8); M3_ROUTE_ATTR(2, 16); M3_ROUTE_ATTR(2, 32); M3_ROUTE_ATTR(2, 64);
#undef M3_ROUTE_ATTR
    once.mark();
  }
  ExpertRoute rt;
  rt.logits = logits; rt.row_len = row_len; rt.rows_per_batch = rows_per_batch; rt.gate_idx = gate_idx; rt.gate_value = gate_value;
  rt.mapping = mapping; rt.acc_hist = acc_hist; rt.pos = pos; rt.b2 = b2;
  dim3 grid(F / kExpertSlice, E, 1);
  const int w2_row_stride = w2_sliced ? kExpertSlice : F;
  const int w2_slice_stride = w2_sliced ? D * kExpertSlice : kExpertSlice;
#define M3_ROUTE_CASE(MT_, E_)                                                                                              \
  do {                                                                                                                      \
    if (ln_gamma != nullptr)                                                                                                \
      hipLaunchKernelGGL((expert_ffn_f32_kernel<MT_, true, E_>), grid, dim3(64 * (kExpertSlice / 16)), lds_bytes, stream, x, ldx, \
                         nullptr, nullptr, S, D, F, w1, b1, w2, w2_row_stride, w2_slice_stride, slab, ln_gamma, ln_beta, ln_eps, rt); \
    else                                                                                                                    \
      hipLaunchKernelGGL((expert_ffn_f32_kernel<MT_, false, E_>), grid, dim3(64 * (kExpertSlice / 16)), lds_bytes, stream, x, ldx, \
                         nullptr, nullptr, S, D, F, w1, b1, w2, w2_row_stride, w2_slice_stride, slab, nullptr, nullptr, 0.f, rt); \
  } while (0)
#define M3_ROUTE_E(E_) do { if (mt == 1) M3_ROUTE_CASE(1, E_); else M3_ROUTE_CASE(2, E_); } while (0)
  switch (E) {
    case 8: M3_ROUTE_E(8); break;
    case 16: M3_ROUTE_E(16); break;
    case 32: M3_ROUTE_E(32); break;
    default: M3_ROUTE_E(64); break;
  }
#undef M3_ROUTE_E
#undef M3_ROUTE_CASE
  M3_LAUNCH_CHECK();
  return 0;
}

// out[s] = resid[s] + alpha * gate[s] * (b2[g_s] + sum_k slab[k][mapping[s]])  (+ optional LayerNorm)
// mapping == NULL: the slabs hold ORIGINAL rows (self-routing expert launch); a row is dropped when gate_idx[s] < 0.
// One wave per token row; NV = float4 per lane.
template <int NV>
__global__ __launch_bounds__(256) void moe_combine_kernel(const float* __restrict__ slab, int n_slices,
                                                          const int32_t* __restrict__ mapping,
                                                          const int32_t* __restrict__ gate_idx,
                                                          const float* __restrict__ gate_value,
                                                          const float* __restrict__ b2,
                                                          const float* resid, float alpha,
                                                          const float* __restrict__ ln_gamma,
                                                          const float* __restrict__ ln_beta, float ln_eps,
                                                          float* out, int S, int D, bf16_t* out_b, float* out_stats,
                                                          const int32_t* __restrict__ n_slices_dev) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s = blockIdx.x * 4 + wave;
  if (s >= S) return;
  if (n_slices_dev != nullptr) n_slices = min(n_slices, *n_slices_dev);   // (the fused fp8 kernel chose its F split on the device)
  const int g = gate_idx ? gate_idx[s] : 0;
  const int m = mapping ? mapping[s] : (g >= 0 ? s : -1);
  const float gate = (m >= 0) ? (gate_value ? gate_value[s] : 1.f) : 0.f;
  f32x4 v[NV];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (lane + 64 * i) * 4;
    v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < D) {
      f32x4 y = f32x4{0.f, 0.f, 0.f, 0.f};
      constexpr int INF = NV <= 2 ? 16 : 8;            // slab rows in flight (one memory round trip for F/64 = 16 slices), summed in slice order
      if (mapping == nullptr) {
        // ORIGINAL-row slabs: the addresses do not depend on the gate, so the loads go out beside the gate's (one round trip);
        // a dropped row's slab rows are never written: selected away, not multiplied away
        for (int k0 = 0; k0 < n_slices; k0 += INF) {
          f32x4 t[INF];
#pragma unroll
          for (int j = 0; j < INF; ++j)
            t[j] = (k0 + j < n_slices) ? ldg4(slab + ((size_t)(k0 + j) * S + s) * D + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < INF; ++j) y += t[j];
        }
        if (b2) y += ldg4(b2 + (size_t)max(g, 0) * D + c);
        if (m < 0) y = f32x4{0.f, 0.f, 0.f, 0.f};
      } else if (m >= 0) {
        if (b2) y = ldg4(b2 + (size_t)g * D + c);
        for (int k0 = 0; k0 < n_slices; k0 += INF) {
          f32x4 t[INF];
#pragma unroll
          for (int j = 0; j < INF; ++j)
            t[j] = (k0 + j < n_slices) ? ldg4(slab + ((size_t)(k0 + j) * S + m) * D + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < INF; ++j) y += t[j];
        }
      }
      y *= (alpha * gate);
      if (resid) y += ldg4(resid + (size_t)s * D + c);
      v[i] = y;
      sum += (y[0] + y[1]) + (y[2] + y[3]);
    }
  }
  if (ln_gamma) {
    const float mean = wave_sum(sum) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c < D) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float d = v[i][j] - mean;
          q += d * d;
        }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + ln_eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c < D) {
        const f32x4 ga = ldg4(ln_gamma + c), be = ldg4(ln_beta + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[i][j] = (v[i][j] - mean) * rstd * ga[j] + be[j];
      }
    }
  }
  float t1 = 0.f, t2 = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < D) {
      stg4(out + (size_t)s * D + c, v[i]);
      if (out_b != nullptr) {                         // bf16 copy of the residual stream for the next GEMMs' A operand
        bf16x4 h;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          h[j] = (bf16_t)v[i][j];
          const float f = (float)h[j];
          t1 += f;
          t2 += f * f;
        }
        *reinterpret_cast<bf16x4*>(out_b + (size_t)s * D + c) = h;
      }
    }
  }
  if (out_stats != nullptr) {      // row statistics of the bf16 copy (kernels.h: Yb_stats): total in part 0, zeros elsewhere
    t1 = wave_sum(t1);
    t2 = wave_sum(t2);
    if (lane < 2 * kXbStatParts) out_stats[(size_t)s * 2 * kXbStatParts + lane] = lane == 0 ? t1 : (lane == 1 ? t2 : 0.f);
  }
}

int launch_moe_combine(const float* slab, int n_slices, const int32_t* mapping, const int32_t* gate_idx,
                       const float* gate_value, const float* b2, const float* resid, float alpha,
                       const float* ln_gamma, const float* ln_beta, float ln_eps, float* out, int S, int D,
                       hipStream_t stream, void* out_bf16, float* out_stats, const int32_t* n_slices_dev) {
  M3_REQUIRE((D & 3) == 0 && D <= 2048, "moe_combine: D=%d must be a multiple of 4 (<=2048)", D);
  if (S == 0) return 0;
  const int nv = cdiv(D, 256);
  dim3 grid(cdiv(S, 4));
#define M3_COMBINE_CASE(NV_)                                                                              \
  hipLaunchKernelGGL((moe_combine_kernel<NV_>), grid, dim3(256), 0, stream, slab, n_slices, mapping,     \
                     gate_idx, gate_value, b2, resid, alpha, ln_gamma, ln_beta, ln_eps, out, S, D, (bf16_t*)out_bf16, out_stats, n_slices_dev)
  if (nv <= 1) M3_COMBINE_CASE(1); else if (nv <= 2) M3_COMBINE_CASE(2); else if (nv <= 4) M3_COMBINE_CASE(4); else M3_COMBINE_CASE(8);
#undef M3_COMBINE_CASE
  M3_LAUNCH_CHECK();
  return 0;
}

}  // namespace m3
