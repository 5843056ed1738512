// Fused relative-position multi-head attention (flash-style, fp32 MFMA 16x16x4).
//
// Replaces the layer chain emitted by RelPositionMultiHeadedAttention.forward
// (trainer_3m_fix/layer/attention.py:320-384) + forward_attention_trt (:199-239):
// 6 shuffles, 2 bias adds, q.k^T and q.p^T batched matmuls, add, AttMaskedSoftmax plugin
// (att_masked_softmax_kernel.cu:224-272 / common.cuh:264-360), p.v matmul, transpose+reshape --
// with the (B,h,T',T') score tensor never materialised:
//   s_ij = ((q_i + u) . k_j + (q_i + v) . p_j) * scale          (NO rel_shift, as in the reference)
//   a_ij = softmax_j(s_ij | j < len[b]),  a_ij = 0 for j >= len[b];   ctx_i = sum_j a_ij v_j
// Inputs: qkv [B*T][ldq] = (q | k | v) rows from the fused QKV projection, p [T][ldp] = linear_pos(pos_emb)
// (shared by the batch), pos_bias_u/v [h][dk].  Output ctx [B*T][ldo] with heads merged, i.e. the
// "attn_transpose_and_reshape" shuffle is folded into the store.
// One workgroup (4 waves) per (batch, head, 16-query tile): wave w walks key tiles w, w+4, ... with an
// online softmax (so a 50-frame utterance is one key tile per wave = one memory round trip), the
// probability tile goes C-layout -> A-layout through a 1 KB LDS patch per wave, and the four partial
// (max, sum, O) states are merged through LDS.
#include "common.h"
#include "kernels.h"

namespace m3 {

// (the body as a device function of the work-group id: relpos_attention_dual_kernel runs two independent attention problems --
//  the embed encoder's and the main encoder's first block, different head widths -- in one launch; engine.hip "horizontal fusion")
template <int DK>
__device__ __forceinline__ void relpos_attention_body(const float* __restrict__ qkv, int ldq, const float* __restrict__ pmat, int ldp,
                                                      const float* __restrict__ pos_u, const float* __restrict__ pos_v,
                                                      const int32_t* __restrict__ row_len, int T, int D, float scale,
                                                      float* __restrict__ out, int ldo, int out_bf16, const int32_t* __restrict__ row0,
                                                      int QT, int H, int xcd_map, int chunk, int left_chunks, const int wg_id) {
  constexpr int KS = DK / 16;
  // one batch of kernel-argument loads instead of one per first use (see gemm.hip: ~5 dependent s_load rounds otherwise)
  asm volatile("" ::"s"(qkv), "s"(ldq), "s"(pmat), "s"(ldp), "s"(pos_u), "s"(pos_v), "s"(row_len), "s"(T), "s"(D), "s"(scale),
               "s"(out), "s"(ldo), "s"(out_bf16), "s"(row0), "s"(QT), "s"(H), "s"(xcd_map), "s"(chunk), "s"(left_chunks));
  __shared__ __attribute__((aligned(16))) float ps_all[4][16][20];
  __shared__ float mo[4][16][DK + 1];   // per-wave partial O
  __shared__ float mm[4][16], ml[4][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, kq = lane >> 4;
  float (*ps)[20] = ps_all[wave];
  // work-group -> (batch, head, query tile).  The QT query tiles of one (batch, head) read the same K / V / P rows: with
  // xcd_map they get block ids that are congruent mod 8, i.e. one XCD and one L2 (blocks are dealt round-robin over the 8
  // XCDs), so those rows leave memory once instead of once per query tile (PMC: 1.55 MB -> see DESIGN.md 3).  Placement
  // only changes speed, never the result.
  int b, h, q0;
  {
    const int id = wg_id;
    int bh, qt;
    if (xcd_map) {
      const int g = id & 7, slot = id >> 3;
      bh = (slot / QT) * 8 + g;
      qt = slot - (slot / QT) * QT;
    } else {
      bh = id / QT;
      qt = id - bh * QT;
    }
    b = bh / H;
    h = bh - b * H;
    q0 = qt * 16;
  }
  const int len = min(row_len ? row_len[b] : T, T);
  // packed rows (row0 != null): utterance b owns rows [row0[b], row0[b] + len) -- nothing beyond its last frame may be
  // read (it is another utterance's) or written; padded rows: [b T, (b+1) T), frames >= len hold defined values
  if (row0 != nullptr && q0 >= len) return;
  const size_t brow = row0 ? (size_t)row0[b] : (size_t)b * T;
  const int last = row0 ? len - 1 : T - 1;          // clamp for the addresses of masked lanes
  const int q_end = row0 ? len : T;                 // query rows this utterance owns

  const int qi = min(q0 + col, last);
  const float* qrow = qkv + (brow + qi) * ldq + h * DK + 4 * kq;
  f32x4 qu[KS], qv[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const f32x4 q4 = ldg4(qrow + 16 * s);
    qu[s] = q4 + ldg4(pos_u + h * DK + 16 * s + 4 * kq);
    qv[s] = q4 + ldg4(pos_v + h * DK + 16 * s + 4 * kq);
  }
  f32x4 o[KS];
#pragma unroll
  for (int n = 0; n < KS; ++n) o[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run[4], l_run[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    m_run[r] = -INFINITY;
    l_run[r] = 0.f;
  }

  // static chunk mask (utils/mask.py:42-75 subsequent_chunk_mask, :127-134 static_chunk_size): query i sees keys
  // [max((i / chunk - left_chunks) chunk, 0), min((i / chunk + 1) chunk, T)) -- all left chunks when left_chunks < 0 --
  // besides the padding mask key < len.  chunk <= 0: full context.  Rows of this lane's accumulators: 4 kq + r.
  int klo[4], khi[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qrow = q0 + 4 * kq + r;
    klo[r] = 0;
    khi[r] = len;
    if (chunk > 0) {
      const int c = qrow / chunk;
      klo[r] = left_chunks < 0 ? 0 : max((c - left_chunks) * chunk, 0);
      khi[r] = min(min((c + 1) * chunk, T), len);
    }
  }
  for (int j0 = 16 * wave; j0 < len; j0 += 64) {
    const int kj = min(j0 + col, last);
    const float* krow = qkv + (brow + kj) * ldq + D + h * DK + 4 * kq;
    const float* prow = pmat + (size_t)kj * ldp + h * DK + 4 * kq;
    f32x4 kb[KS], pb[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      kb[s] = ldg4(krow + 16 * s);
      pb[s] = ldg4(prow + 16 * s);
    }
    // V fragments for this key tile: B[k = key 4kq+jj][n = channel 16n+col]
    float vb[KS][4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int vj = min(j0 + 4 * kq + jj, last);
      const float* vrow = qkv + (brow + vj) * ldq + 2 * D + h * DK + col;
#pragma unroll
      for (int n = 0; n < KS; ++n) vb[n][jj] = vrow[16 * n];
    }
    f32x4 sc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sc = mfma16(qu[s][j], kb[s][j], sc);
        sc = mfma16(qv[s][j], pb[s][j], sc);
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool valid = (j0 + col) >= klo[r] && (j0 + col) < khi[r];
      const float sv = valid ? sc[r] * scale : -INFINITY;
      const float m_new = fmaxf(m_run[r], group16_max(sv));
      const float pexp = valid ? expf(sv - m_new) : 0.f;
      const float corr = (m_new == -INFINITY) ? 1.f : expf(m_run[r] - m_new);   // (a key tile wholly outside the row's chunk window)
      l_run[r] = l_run[r] * corr + group16_sum(pexp);
      m_run[r] = m_new;
#pragma unroll
      for (int n = 0; n < KS; ++n) o[n][r] *= corr;
      ps[4 * kq + r][col] = pexp;
    }
    // ps is private to this wave and a wave's LDS operations execute in order: only the compiler has
    // to be kept from moving the transposed read across the writes (no s_barrier needed)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const f32x4 pa = *reinterpret_cast<const f32x4*>(&ps[col][4 * kq]);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int n = 0; n < KS; ++n)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) o[n] = mfma16(pa[jj], vb[n][jj], o[n]);
  }

  // ---- merge the four waves' partial softmax states ----
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (col == 0) {
      mm[wave][4 * kq + r] = m_run[r];
      ml[wave][4 * kq + r] = l_run[r];
    }
#pragma unroll
    for (int n = 0; n < KS; ++n) mo[wave][4 * kq + r][16 * n + col] = o[n][r];
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 16 * DK; idx += 256) {
    const int i = idx / DK, d = idx - i * DK;
    const int qrow = q0 + i;
    if (qrow >= q_end) continue;
    const float m_tot = fmaxf(fmaxf(mm[0][i], mm[1][i]), fmaxf(mm[2][i], mm[3][i]));
    float l_tot = 0.f, acc = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float f = (mm[w][i] == -INFINITY) ? 0.f : expf(mm[w][i] - m_tot);   // waves without a (visible) key tile: factor 0
      l_tot += ml[w][i] * f;
      acc += mo[w][i][d] * f;
    }
    const float o = l_tot > 0.f ? acc / l_tot : 0.f;       // a row with no visible key (padded query under a chunk mask): zeros
    if (out_bf16) reinterpret_cast<bf16_t*>(out)[(brow + qrow) * ldo + h * DK + d] = (bf16_t)o;
    else out[(brow + qrow) * ldo + h * DK + d] = o;
  }
}

template <int DK>
__global__ __launch_bounds__(256) void relpos_attention_kernel(const float* __restrict__ qkv, int ldq,
                                                              const float* __restrict__ pmat, int ldp,
                                                              const float* __restrict__ pos_u,
                                                              const float* __restrict__ pos_v,
                                                              const int32_t* __restrict__ row_len, int T, int D,
                                                              float scale, float* __restrict__ out, int ldo, int out_bf16,
                                                              const int32_t* __restrict__ row0, int QT, int H, int xcd_map,
                                                              int chunk, int left_chunks) {
  relpos_attention_body<DK>(qkv, ldq, pmat, ldp, pos_u, pos_v, row_len, T, D, scale, out, ldo, out_bf16, row0, QT, H, xcd_map, chunk,
                            left_chunks, (int)blockIdx.x);
}
struct AttKernArgs {
  const float *qkv; int ldq; const float* pmat; int ldp; const float *pos_u, *pos_v; const int32_t* row_len; int T, D; float scale;
  float* out; int ldo, out_bf16; const int32_t* row0; int QT, H, xcd_map, chunk, left_chunks, n_wg;
};
// work-groups [0, n0) the first problem (n0 = its count rounded up to a multiple of 8: "id % 8 = XCD" holds for both), the rest the second
template <int DK0, int DK1>
__global__ __launch_bounds__(256) void relpos_attention_dual_kernel(const AttKernArgs a, const AttKernArgs b, const int n0) {
  if ((int)blockIdx.x < n0) {
    if ((int)blockIdx.x < a.n_wg)
      relpos_attention_body<DK0>(a.qkv, a.ldq, a.pmat, a.ldp, a.pos_u, a.pos_v, a.row_len, a.T, a.D, a.scale, a.out, a.ldo, a.out_bf16, a.row0,
                                 a.QT, a.H, a.xcd_map, a.chunk, a.left_chunks, (int)blockIdx.x);
  } else {
    relpos_attention_body<DK1>(b.qkv, b.ldq, b.pmat, b.ldp, b.pos_u, b.pos_v, b.row_len, b.T, b.D, b.scale, b.out, b.ldo, b.out_bf16, b.row0,
                               b.QT, b.H, b.xcd_map, b.chunk, b.left_chunks, (int)blockIdx.x - n0);
  }
}

int launch_relpos_attention_args(const AttArgs& a, hipStream_t stream) {
  return launch_relpos_attention(a.qkv, a.ldq, a.pmat, a.ldp, a.pos_u, a.pos_v, a.row_len, a.B, a.T, a.H, a.dk, a.scale, a.out, a.ldo, stream,
                                 a.out_bf16, a.row0, a.chunk, a.left_chunks);
}
static bool att_dual_dk(int dk) { return dk == 64 || dk == 128; }
bool relpos_attention_dual_fusable(const AttArgs& a, const AttArgs& b) {
  return att_dual_dk(a.dk) && att_dual_dk(b.dk) && a.B > 0 && a.T > 0 && b.B > 0 && b.T > 0 && (a.ldq & 3) == 0 && (a.ldp & 3) == 0 &&
         (b.ldq & 3) == 0 && (b.ldp & 3) == 0;
}
int launch_relpos_attention_dual(const AttArgs& a, const AttArgs& b, hipStream_t stream) {
  M3_REQUIRE(relpos_attention_dual_fusable(a, b), "attention dual: head widths %d / %d are not a dual instantiation", a.dk, b.dk);
  auto pack = [](const AttArgs& q) {
    AttKernArgs k;
    k.qkv = q.qkv; k.ldq = q.ldq; k.pmat = q.pmat; k.ldp = q.ldp; k.pos_u = q.pos_u; k.pos_v = q.pos_v; k.row_len = q.row_len; k.T = q.T;
    k.D = q.H * q.dk; k.scale = q.scale; k.out = q.out; k.ldo = q.ldo; k.out_bf16 = q.out_bf16; k.row0 = q.row0; k.QT = cdiv(q.T, 16); k.H = q.H;
    k.xcd_map = ((q.H * q.B) % 8 == 0) ? 1 : 0; k.chunk = q.chunk; k.left_chunks = q.left_chunks; k.n_wg = k.QT * q.H * q.B;
    return k;
  };
  const AttKernArgs ka = pack(a), kb = pack(b);
  const int n0 = (int)align_up((size_t)ka.n_wg, 8);
  dim3 grid(n0 + kb.n_wg);
#define M3_ATTD(D0_, D1_) hipLaunchKernelGGL((relpos_attention_dual_kernel<D0_, D1_>), grid, dim3(256), 0, stream, ka, kb, n0)
  if (a.dk == 128 && b.dk == 64) M3_ATTD(128, 64);
  else if (a.dk == 64 && b.dk == 128) M3_ATTD(64, 128);
  else if (a.dk == 64) M3_ATTD(64, 64);
  else M3_ATTD(128, 128);
#undef M3_ATTD
  M3_LAUNCH_CHECK();
  return 0;
}

int launch_relpos_attention(const float* qkv, int ldq, const float* pmat, int ldp, const float* pos_u,
                            const float* pos_v, const int32_t* row_len, int B, int T, int H, int dk, float scale,
                            float* out, int ldo, hipStream_t stream, int out_bf16, const int32_t* row0, int chunk, int left_chunks) {
  M3_REQUIRE(B > 0 && T > 0 && H > 0, "attention: empty problem");
  M3_REQUIRE((ldq & 3) == 0 && (ldp & 3) == 0, "attention: row strides must be multiples of 4");
  const int QT = cdiv(T, 16);
  dim3 grid(QT * H * B);
  const int D = H * dk;
  const int xcd_map = ((H * B) % 8 == 0) ? 1 : 0;
#define M3_ATT_CASE(DK_)                                                                                   \
  hipLaunchKernelGGL((relpos_attention_kernel<DK_>), grid, dim3(256), 0, stream, qkv, ldq, pmat, ldp, pos_u, \
                     pos_v, row_len, T, D, scale, out, ldo, out_bf16, row0, QT, H, xcd_map, chunk, left_chunks)
  switch (dk) {
    case 16: M3_ATT_CASE(16); break;
    case 32: M3_ATT_CASE(32); break;
    case 64: M3_ATT_CASE(64); break;
    case 128: M3_ATT_CASE(128); break;
    default: M3_REQUIRE(false, "attention: d_k=%d unsupported (16/32/64/128)", dk);
  }
#undef M3_ATT_CASE
  M3_LAUNCH_CHECK();
  return 0;
}


// ------------------------------------------------------------------------------------------------------------------------
// Chunk-by-chunk (streaming) form: the queries are the C frames of the current chunk, the keys / values are every frame the
// static chunk mask lets them see -- the chunk itself (still in the QKV GEMM's output) and the frames to its left, which come
// from the per-layer history `hist` [B][cap][2 D] (K | V rows of all earlier chunks, a ring when cap < the stream's length).
// This is what the reference's cache plugins were written for: CatSplitCachePluginDynamic concatenates the cached K / V with
// the chunk's and splits the new cache off (cat_split_cache_kernel.cu:30-107), AttStreamSoftmaxPluginDynamic masks the scores
// to [ld - decode_frame_num, mask_idx + cache_len) (att_stream_softmax_kernel.cu:136-191), RelPositionalEncoding's streaming
// form takes pe[offset : offset + T] (rel_positional_encoding_kernel.cu:108-123).  Here they are one kernel: nothing is
// concatenated or copied, the work-group that owns 16 query frames of a head also appends that head's K / V rows of those
// frames to the history (read only by LATER chunks: keys of the current chunk are taken from the QKV buffer), and p is
// indexed by the key's absolute position.  The chunk counter lives on the device (`step`), so the launch is replayable from
// a hipGraph.  Key tiles sit at absolute multiples of 16 and wave w owns tiles w, w + 4, ... exactly as in the full-utterance
// kernel above; tiles left of every row's window are skipped (there they leave the softmax state untouched), so a chunk's
// context equals the rows the full-utterance kernel computes under the same static chunk mask bit for bit.
template <int DK>
__global__ __launch_bounds__(256) void relpos_attention_stream_kernel(const float* __restrict__ qkv, int ldq,
                                                                     float* __restrict__ hist, int cap,
                                                                     const float* __restrict__ pmat, int ldp,
                                                                     const float* __restrict__ pos_u, const float* __restrict__ pos_v,
                                                                     const int32_t* __restrict__ chunk_len, const int32_t* __restrict__ step,
                                                                     int C, int D, float scale, float* __restrict__ out, int ldo,
                                                                     int QT, int H, int left_chunks) {
  constexpr int KS = DK / 16;
  asm volatile("" ::"s"(qkv), "s"(ldq), "s"(hist), "s"(cap), "s"(pmat), "s"(ldp), "s"(pos_u), "s"(pos_v), "s"(chunk_len), "s"(step),
               "s"(C), "s"(D), "s"(scale), "s"(out), "s"(ldo), "s"(QT), "s"(H), "s"(left_chunks));
  __shared__ __attribute__((aligned(16))) float ps_all[4][16][20];
  __shared__ float mo[4][16][DK + 1];
  __shared__ float mm[4][16], ml[4][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, kq = lane >> 4;
  float (*ps)[20] = ps_all[wave];
  const int bh = blockIdx.x / QT, qt = blockIdx.x - bh * QT;
  const int b = bh / H, h = bh - b * H, q0 = qt * 16;
  const int off = *step * C;                              // absolute index of the chunk's first frame
  const int nlive = min(max(chunk_len[b], 0), C);         // valid frames of utterance b in this chunk
  const int len = off + nlive;                            // keys the utterance has so far
  const size_t brow = (size_t)b * C;
  const float* hb = hist + (size_t)b * cap * 2 * D;

  // ---- this work-group's share of the history append: K | V of head h, frames q0 .. q0 + 15 of the chunk ----
  for (int idx = threadIdx.x; idx < 16 * 2 * (DK / 4); idx += 256) {
    const int r = idx / (2 * (DK / 4)), rem = idx - r * 2 * (DK / 4), kv = rem / (DK / 4), c4 = (rem - kv * (DK / 4)) * 4;
    if (q0 + r < C) {
      const f32x4 val = ldg4(qkv + (brow + q0 + r) * ldq + (1 + kv) * D + h * DK + c4);
      stg4(hist + ((size_t)b * cap + (size_t)((off + q0 + r) % cap)) * 2 * D + kv * D + h * DK + c4, val);
    }
  }

  const int qi = min(q0 + col, C - 1);
  const float* qrow = qkv + (brow + qi) * ldq + h * DK + 4 * kq;
  f32x4 qu[KS], qv[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const f32x4 q4 = ldg4(qrow + 16 * s);
    qu[s] = q4 + ldg4(pos_u + h * DK + 16 * s + 4 * kq);
    qv[s] = q4 + ldg4(pos_v + h * DK + 16 * s + 4 * kq);
  }
  f32x4 o[KS];
#pragma unroll
  for (int n = 0; n < KS; ++n) o[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run[4], l_run[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    m_run[r] = -INFINITY;
    l_run[r] = 0.f;
  }
  // every query of the chunk sits in chunk number *step: one window [klo, khi) for all of them
  const int cidx = off / C;
  const int klo = left_chunks < 0 ? 0 : max((cidx - left_chunks) * C, 0);
  const int khi = min((cidx + 1) * C, len);
  const int last = max(len - 1, 0);
  for (int j0 = 16 * wave; j0 < len; j0 += 64) {
    if (j0 + 16 <= klo) continue;                          // wholly left of the window: the full kernel's state is untouched there
    const int kj = min(j0 + col, last);
    const float* krow = (kj >= off ? qkv + (brow + kj - off) * ldq + D : hb + (size_t)(kj % cap) * 2 * D) + h * DK + 4 * kq;
    const float* prow = pmat + (size_t)kj * ldp + h * DK + 4 * kq;
    f32x4 kb[KS], pb[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      kb[s] = ldg4(krow + 16 * s);
      pb[s] = ldg4(prow + 16 * s);
    }
    float vb[KS][4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int vj = min(j0 + 4 * kq + jj, last);
      const float* vrow = (vj >= off ? qkv + (brow + vj - off) * ldq + 2 * D : hb + (size_t)(vj % cap) * 2 * D + D) + h * DK + col;
#pragma unroll
      for (int n = 0; n < KS; ++n) vb[n][jj] = vrow[16 * n];
    }
    f32x4 sc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sc = mfma16(qu[s][j], kb[s][j], sc);
        sc = mfma16(qv[s][j], pb[s][j], sc);
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool valid = (j0 + col) >= klo && (j0 + col) < khi;
      const float sv = valid ? sc[r] * scale : -INFINITY;
      const float m_new = fmaxf(m_run[r], group16_max(sv));
      const float pexp = valid ? expf(sv - m_new) : 0.f;
      const float corr = (m_new == -INFINITY) ? 1.f : expf(m_run[r] - m_new);
      l_run[r] = l_run[r] * corr + group16_sum(pexp);
      m_run[r] = m_new;
#pragma unroll
      for (int n = 0; n < KS; ++n) o[n][r] *= corr;
      ps[4 * kq + r][col] = pexp;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const f32x4 pa = *reinterpret_cast<const f32x4*>(&ps[col][4 * kq]);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int n = 0; n < KS; ++n)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) o[n] = mfma16(pa[jj], vb[n][jj], o[n]);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (col == 0) {
      mm[wave][4 * kq + r] = m_run[r];
      ml[wave][4 * kq + r] = l_run[r];
    }
#pragma unroll
    for (int n = 0; n < KS; ++n) mo[wave][4 * kq + r][16 * n + col] = o[n][r];
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 16 * DK; idx += 256) {
    const int i = idx / DK, d = idx - i * DK;
    const int qrow = q0 + i;
    if (qrow >= C) continue;
    const float m_tot = fmaxf(fmaxf(mm[0][i], mm[1][i]), fmaxf(mm[2][i], mm[3][i]));
    float l_tot = 0.f, acc = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float f = (mm[w][i] == -INFINITY) ? 0.f : expf(mm[w][i] - m_tot);
      l_tot += ml[w][i] * f;
      acc += mo[w][i][d] * f;
    }
    out[(brow + qrow) * ldo + h * DK + d] = l_tot > 0.f ? acc / l_tot : 0.f;
  }
}

int launch_relpos_attention_stream(const float* qkv, int ldq, float* hist, int cap, const float* pmat, int ldp, const float* pos_u,
                                   const float* pos_v, const int32_t* chunk_len, const int32_t* step, int B, int C, int H, int dk,
                                   float scale, float* out, int ldo, int left_chunks, hipStream_t stream) {
  M3_REQUIRE(B > 0 && C > 0 && H > 0 && cap >= C, "attention (stream): empty problem or history shorter than a chunk");
  M3_REQUIRE(left_chunks < 0 || cap >= (left_chunks + 1) * C, "attention (stream): a history of %d frames cannot hold %d left chunks of %d", cap, left_chunks, C);
  M3_REQUIRE((ldq & 3) == 0 && (ldp & 3) == 0 && (dk & 3) == 0, "attention (stream): row strides must be multiples of 4");
  M3_REQUIRE(hist && chunk_len && step, "attention (stream): null state");
  const int QT = cdiv(C, 16);
  dim3 grid(QT * H * B);
  const int D = H * dk;
#define M3_ATT_CASE(DK_)                                                                                            \
  hipLaunchKernelGGL((relpos_attention_stream_kernel<DK_>), grid, dim3(256), 0, stream, qkv, ldq, hist, cap, pmat, ldp, \
                     pos_u, pos_v, chunk_len, step, C, D, scale, out, ldo, QT, H, left_chunks)
  switch (dk) {
    case 16: M3_ATT_CASE(16); break;
    case 32: M3_ATT_CASE(32); break;
    case 64: M3_ATT_CASE(64); break;
    case 128: M3_ATT_CASE(128); break;
    default: M3_REQUIRE(false, "attention (stream): d_k=%d unsupported (16/32/64/128)", dk);
  }
#undef M3_ATT_CASE
  M3_LAUNCH_CHECK();
  return 0;
}


// ------------------------------------------------------------------------------------------------------------------------
// The same operator on bf16 rows for the 16-bit modes of long batches (qkv written as bf16 by the QKV GEMM, context
// written as bf16 for the output projection), T' <= 128 keys:  v_mfma_f32_16x16x32_bf16, fp32 softmax.
//
// One work-group per (utterance, head) instead of one per (utterance, head, 16-query tile): K, P = linear_pos(pos_emb) and
// V of the head are staged ONCE into LDS (full-line loads; the fp32 kernel above re-reads them per query tile and loads V
// with 4-byte accesses), then each wave takes query tiles.  Scores are computed TRANSPOSED, S^T = K (q+u)^T + P (q+v)^T
// (keys on the MFMA row axis): a lane then holds, for ITS query (column), 4 consecutive keys per 16-key tile in registers --
// the whole row of probabilities for T' <= 128 is 32 registers, softmax needs no online rescaling and no LDS transpose, and
// two key tiles packed together ARE the B operand of  O^T = V^T P^T  (k index j < 4: tile 2c key 4kq+j, j >= 4: tile 2c+1);
// V is kept transposed in LDS ([channel][key]) so that the matching A operand is two 8-byte reads.
namespace {
constexpr int kAttTP = 128;                                    // keys a work-group can hold
template <int DK> constexpr int att16_lds_bytes() { return 2 * kAttTP * (DK + 8) * 2 + DK * (kAttTP + 8) * 2; }
}  // namespace

template <int DK>
__global__ __launch_bounds__(256) void relpos_attention_bf16_kernel(const bf16_t* __restrict__ qkv, int ldq,
                                                                   const float* __restrict__ pmat, int ldp,
                                                                   const float* __restrict__ pos_u,
                                                                   const float* __restrict__ pos_v,
                                                                   const int32_t* __restrict__ row_len, int T, int D, int H,
                                                                   float scale, bf16_t* __restrict__ out, int ldo,
                                                                   const int32_t* __restrict__ row0, int chunk, int left_chunks) {
  constexpr int KLD = DK + 8;                                  // bf16 elements per K / P row in LDS (conflict-free 16-B fragment reads)
  constexpr int VLD = kAttTP + 8;                              // bf16 elements per Vt row (one channel, all keys)
  constexpr int KS = DK / 32;                                  // 32-deep k-steps of the score products
  constexpr int NCH = DK / 16;                                 // 16-channel tiles of the output
  extern __shared__ __attribute__((aligned(16))) unsigned char att_lds[];
  bf16_t* Ks = reinterpret_cast<bf16_t*>(att_lds);             // [kAttTP][KLD]
  bf16_t* Ps = Ks + kAttTP * KLD;                              // [kAttTP][KLD]
  bf16_t* Vt = Ps + kAttTP * KLD;                              // [DK][VLD]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, kq = lane >> 4;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int len = min(row_len ? row_len[b] : T, T);
  if (len <= 0) return;
  const size_t brow = row0 ? (size_t)row0[b] : (size_t)b * T;
  const int q_end = row0 ? len : T;                            // query rows this utterance owns (padded layout: all T)
  const int nkt = (len + 15) >> 4;                             // 16-key tiles with at least one valid key
  const int nk32 = ((len + 31) >> 5) << 5;                     // keys rounded up to the pairs the P.V product consumes

  // ---- all loads of the work-group are issued up front, clamped instead of branched (a load under a branch is one memory
  //      round trip per loop iteration): the query rows of this wave's (at most two) tiles, then K / P / V of the head ----
  constexpr int QT = kAttTP / 64;                              // query tiles per wave
  bf16x8 q8[QT][KS];
#pragma unroll
  for (int i = 0; i < QT; ++i) {
    const int qi = min(16 * wave + 64 * i + col, q_end - 1);
    const bf16_t* qrow = qkv + (brow + qi) * ldq + h * DK + 8 * kq;
#pragma unroll
    for (int s = 0; s < KS; ++s) q8[i][s] = *reinterpret_cast<const bf16x8*>(qrow + 32 * s);
  }
  f32x4 pu[KS][2], pv[KS][2];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    pu[s][0] = ldg4(pos_u + h * DK + 32 * s + 8 * kq);
    pu[s][1] = ldg4(pos_u + h * DK + 32 * s + 8 * kq + 4);
    pv[s][0] = ldg4(pos_v + h * DK + 32 * s + 8 * kq);
    pv[s][1] = ldg4(pos_v + h * DK + 32 * s + 8 * kq + 4);
  }
  {
    constexpr int CPR = DK / 8;                                // 16-byte chunks per row
    constexpr int RPP = 256 / CPR;                             // rows per pass
    constexpr int NP = kAttTP / RPP;                           // passes over the key rows
    const int c = tid % CPR, r0 = tid / CPR;
    u32x4 kv[NP], vv[NP];
    f32x4 p0[NP], p1[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int j = min(r0 + RPP * i, len - 1);                // keys >= len: a valid row is read and replaced by zeros below
      const bf16_t* rowp = qkv + (brow + j) * ldq + h * DK + 8 * c;
      kv[i] = *reinterpret_cast<const u32x4*>(rowp + D);
      vv[i] = *reinterpret_cast<const u32x4*>(rowp + 2 * D);
      const float* pr = pmat + (size_t)j * ldp + h * DK + 8 * c;
      p0[i] = ldg4(pr);
      p1[i] = ldg4(pr + 4);
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int j = r0 + RPP * i;
      if (j < nk32) {
        const bool on = j < len;
        *reinterpret_cast<u32x4*>(Ks + j * KLD + 8 * c) = on ? kv[i] : u32x4{0u, 0u, 0u, 0u};
        *reinterpret_cast<bf16x8*>(Ps + j * KLD + 8 * c) = cvt8(on ? p0[i] : f32x4{0.f, 0.f, 0.f, 0.f}, on ? p1[i] : f32x4{0.f, 0.f, 0.f, 0.f});
        const bf16x8 v8 = __builtin_bit_cast(bf16x8, on ? vv[i] : u32x4{0u, 0u, 0u, 0u});
#pragma unroll
        for (int e = 0; e < 8; ++e) Vt[(8 * c + e) * VLD + j] = v8[e];
      }
    }
  }
  __syncthreads();

  // ---- query tiles: wave w takes tiles w and w + 4 ----
#pragma unroll
  for (int i = 0; i < QT; ++i) {
    const int q0 = 16 * wave + 64 * i;
    if (q0 >= q_end) break;
    bf16x8 qu[KS], qv[KS];                                     // B operands: (q + u)^T, (q + v)^T -- lane (query col, k = 8 kq + j)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        qu[s][j] = (bf16_t)((float)q8[i][s][j] + pu[s][0][j]);
        qu[s][4 + j] = (bf16_t)((float)q8[i][s][4 + j] + pu[s][1][j]);
        qv[s][j] = (bf16_t)((float)q8[i][s][j] + pv[s][0][j]);
        qv[s][4 + j] = (bf16_t)((float)q8[i][s][4 + j] + pv[s][1][j]);
      }
    }
    // static chunk mask (see the fp32 kernel): this lane's query sees keys [klo, khi)
    int klo = 0, khi = len;
    if (chunk > 0) {
      const int c = (q0 + col) / chunk;
      klo = left_chunks < 0 ? 0 : max((c - left_chunks) * chunk, 0);
      khi = min(min((c + 1) * chunk, T), len);
    }
    // scores, transposed: sT[t][r] = s(query col, key 16 t + 4 kq + r)
    f32x4 sT[kAttTP / 16];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < kAttTP / 16; ++t) {
      sT[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (t < nkt) {
        const bf16_t* krow = Ks + (16 * t + col) * KLD + 8 * kq;
        const bf16_t* prow = Ps + (16 * t + col) * KLD + 8 * kq;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          sT[t] = mfma16h(*reinterpret_cast<const bf16x8*>(krow + 32 * s), qu[s], sT[t]);
          sT[t] = mfma16h(*reinterpret_cast<const bf16x8*>(prow + 32 * s), qv[s], sT[t]);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = 16 * t + 4 * kq + r;
        const bool valid = key >= klo && key < khi;
        sT[t][r] = valid ? sT[t][r] * scale : -INFINITY;
        mx = fmaxf(mx, sT[t][r]);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));                    // the four key quarters of this query's column
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
    const float mref = mx == -INFINITY ? 0.f : mx;            // (no visible key at all: a padded query under a chunk mask)
#pragma unroll
    for (int t = 0; t < kAttTP / 16; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sT[t][r] = __expf(sT[t][r] - mref);                    // masked keys: exp(-inf) = 0
        sum += sT[t][r];
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = sum > 0.f ? 1.f / sum : 0.f;
    // O^T = V^T P^T over pairs of key tiles
    f32x4 oT[NCH];
#pragma unroll
    for (int n = 0; n < NCH; ++n) oT[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c2 = 0; c2 < kAttTP / 32; ++c2) {
      if (2 * c2 < nkt) {
        bf16x8 pb;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          pb[j] = (bf16_t)sT[2 * c2][j];
          pb[4 + j] = (bf16_t)sT[2 * c2 + 1][j];
        }
#pragma unroll
        for (int n = 0; n < NCH; ++n) {
          const bf16_t* vrow = Vt + (16 * n + col) * VLD + 32 * c2 + 4 * kq;
          const bf16x4 lo = *reinterpret_cast<const bf16x4*>(vrow), hi = *reinterpret_cast<const bf16x4*>(vrow + 16);
          bf16x8 va;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            va[j] = lo[j];
            va[4 + j] = hi[j];
          }
          oT[n] = mfma16h(va, pb, oT[n]);
        }
      }
    }
    // oT[n][r] = O(query col, channel 16 n + 4 kq + r): 4 consecutive channels per lane
    const int qrow_out = q0 + col;
    if (qrow_out < q_end) {
      bf16_t* orow = out + (brow + qrow_out) * ldo + h * DK + 4 * kq;
#pragma unroll
      for (int n = 0; n < NCH; ++n) {
        bf16x4 o4;
#pragma unroll
        for (int r = 0; r < 4; ++r) o4[r] = (bf16_t)(oT[n][r] * inv);
        *reinterpret_cast<bf16x4*>(orow + 16 * n) = o4;
      }
    }
  }
}

bool relpos_attention_bf16_supports(int T, int dk) { return T <= kAttTP && (dk == 64 || dk == 128); }

int init_relpos_attention_bf16_kernels() {
  static PerDeviceOnce once;
  if (once.done()) return 0;
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)relpos_attention_bf16_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, att16_lds_bytes<64>()));
  M3_CHECK_HIP(hipFuncSetAttribute((const void*)relpos_attention_bf16_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, att16_lds_bytes<128>()));
  once.mark();
  return 0;
}

int launch_relpos_attention_bf16(const void* qkv, int ldq, const float* pmat, int ldp, const float* pos_u, const float* pos_v,
                                 const int32_t* row_len, int B, int T, int H, int dk, float scale, void* out, int ldo,
                                 hipStream_t stream, const int32_t* row0, int chunk, int left_chunks) {
  M3_REQUIRE(B > 0 && T > 0 && H > 0, "attention: empty problem");
  M3_REQUIRE(relpos_attention_bf16_supports(T, dk), "attention (bf16 rows): T'=%d > %d keys or d_k=%d not 64 / 128", T, kAttTP, dk);
  M3_REQUIRE((ldq & 7) == 0 && (ldp & 3) == 0 && (ldo & 3) == 0, "attention (bf16 rows): row strides must be multiples of 8 / 4 / 4");
  if (int rc = init_relpos_attention_bf16_kernels()) return rc;
  const int D = H * dk;
  if (dk == 64)
    hipLaunchKernelGGL((relpos_attention_bf16_kernel<64>), dim3(B * H), dim3(256), att16_lds_bytes<64>(), stream, (const bf16_t*)qkv,
                       ldq, pmat, ldp, pos_u, pos_v, row_len, T, D, H, scale, (bf16_t*)out, ldo, row0, chunk, left_chunks);
  else
    hipLaunchKernelGGL((relpos_attention_bf16_kernel<128>), dim3(B * H), dim3(256), att16_lds_bytes<128>(), stream, (const bf16_t*)qkv,
                       ldq, pmat, ldp, pos_u, pos_v, row_len, T, D, H, scale, (bf16_t*)out, ldo, row0, chunk, left_chunks);
  M3_LAUNCH_CHECK();
  return 0;
}

}  // namespace m3
