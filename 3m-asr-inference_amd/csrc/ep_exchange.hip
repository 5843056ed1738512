// Device-side bookkeeping of the expert-parallel exchange (no host round trip inside a layer).
//
// Reference semantics: the FastMoE path of the trainer (inference in the reference is single-GPU),
//   moe_prepare_forward  trainer_3m_fix/fmoe/functions.py:13-52   local_expert_count by GLOBAL expert id, count exchange,
//                                                                  fwd_expert_count = global_expert_count.view(world, E_loc).sum(0)
//   MOEScatter.forward   fmoe/functions.py:63-86                  local_scatter (rows sorted by global expert) + global_scatter
//   MOEGather.forward    fmoe/functions.py:175-199                global_gather + local_gather
// whose host code reads the counts back (`.cpu()`) to size every all-to-all-v.  Here the exchange has a FIXED shape, so
// nothing has to be read back and the whole layer is enqueued asynchronously:
//
//   wire buffer  [world][1 + C][D]      chunk j = what this rank sends to rank j (and, after the all-to-all with equal
//                                       splits, what it received from rank j): a header row carrying the E_loc row
//                                       counts of the chunk (int32, "count exchange" fused into the payload), then up
//                                       to C rows sorted by the destination's local expert id.  C = rows per rank.
//
//   ep_send_map_kernel   gate_idx / mapping / acc_histogram of the local index step  ->  map_send[s] = wire row of token s
//                        (= where local_scatter puts it, and where its result comes back), headers written
//   ep_recv_gate_kernel  headers of the received chunks  ->  gate_recv[row] = local expert id of every received wire row
//                        (-1 for header rows and unused capacity), the input of the grouped expert FFN
//
// The receiver's stable counting sort (moe_index) orders the rows by local expert, then by source rank, then by their
// order on the wire -- FastMoE's receive order.
#include "common.h"
#include "kernels.h"

namespace m3 {

__global__ __launch_bounds__(256) void ep_send_map_kernel(const int32_t* __restrict__ gate_idx, const int32_t* __restrict__ mapping,
                                                          const int32_t* __restrict__ acc, int S, int world, int e_loc, int cap,
                                                          int32_t* __restrict__ map_send, int32_t* __restrict__ wire, int row_words,
                                                          int32_t* __restrict__ overflow) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid < world * e_loc) {                     // header of chunk j: rows per local expert of rank j
    const int j = tid / e_loc, i = tid - j * e_loc;
    wire[(size_t)j * (cap + 1) * row_words + i] = acc[tid + 1] - acc[tid];
  }
  for (int s = tid; s < S; s += gridDim.x * blockDim.x) {
    const int g = gate_idx[s];
    int m = -1;
    if (g >= 0 && g < world * e_loc) {
      const int j = g / e_loc;
      const int off = mapping[s] - acc[j * e_loc];            // position among the rows bound for rank j
      m = off < cap ? j * (cap + 1) + 1 + off : -1;           // (off < cap always holds when cap >= S)
      if (off >= cap && overflow != nullptr) atomicMax(overflow, off + 1);   // bounded wire: rows the chunk would have needed
    }
    map_send[s] = m;
  }
}

// ep_send_map_kernel + local_scatter in ONE launch (the engine's "moe_ep.send" stage): one wave per token computes the token's
// wire row, records it in map_send and copies the row there (16 B per lane per access); work-group 0 also writes the headers.
__global__ __launch_bounds__(256) void ep_send_rows_kernel(const int32_t* __restrict__ gate_idx, const int32_t* __restrict__ mapping,
                                                           const int32_t* __restrict__ acc, int S, int world, int e_loc, int cap,
                                                           int32_t* __restrict__ map_send, const uint4* __restrict__ x, int row16,
                                                           uint4* __restrict__ wire, int32_t* __restrict__ overflow) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (blockIdx.x == 0)
    for (int t = threadIdx.x; t < world * e_loc; t += blockDim.x) {
      const int j = t / e_loc, i = t - j * e_loc;
      reinterpret_cast<int32_t*>(wire + (size_t)j * (cap + 1) * row16)[i] = acc[t + 1] - acc[t];
    }
  for (int s = blockIdx.x * 4 + wave; s < S; s += gridDim.x * 4) {
    const int g = gate_idx[s];
    int m = -1;
    if (g >= 0 && g < world * e_loc) {
      const int j = g / e_loc;
      const int off = mapping[s] - acc[j * e_loc];
      m = off < cap ? j * (cap + 1) + 1 + off : -1;
      // bounded wire (cap < S): a chunk that would need more rows than it has reports how many; the forward's result is then
      // invalid and the driver repeats it with a larger capacity (m3asr/ep.py) -- rows are never dropped silently
      if (off >= cap && overflow != nullptr && lane == 0) atomicMax(overflow, off + 1);
    }
    if (lane == 0) map_send[s] = m;
    if (m >= 0) {
      const uint4* src = x + (size_t)s * row16;
      uint4* dst = wire + (size_t)m * row16;
      for (int c = lane; c < row16; c += 64) dst[c] = src[c];
    }
  }
}

__global__ __launch_bounds__(256) void ep_recv_gate_kernel(const int32_t* __restrict__ wire, int world, int e_loc, int cap,
                                                           int row_words, int32_t* __restrict__ gate_recv) {
  // one work-group per source rank j
  __shared__ int32_t off[1025];
  const int j = blockIdx.x;
  const int32_t* hdr = wire + (size_t)j * (cap + 1) * row_words;
  if (threadIdx.x == 0) {
    int run = 0;
    for (int i = 0; i < e_loc; ++i) {
      off[i] = run;
      int c = hdr[i];
      c = c < 0 ? 0 : c;
      run = min(run + c, cap);                   // counts come off the wire: never index past the chunk
    }
    off[e_loc] = run;
  }
  __syncthreads();
  int32_t* out = gate_recv + (size_t)j * (cap + 1);
  if (threadIdx.x == 0) out[0] = -1;             // the header row is not a token
  for (int t = threadIdx.x; t < cap; t += blockDim.x) {
    int g = -1;
    if (t < off[e_loc]) {
      int lo = 0, hi = e_loc;                    // last i with off[i] <= t
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (off[mid] <= t) lo = mid; else hi = mid;
      }
      g = lo;
    }
    out[1 + t] = g;
  }
}

int launch_ep_send_map(const int32_t* gate_idx, const int32_t* mapping, const int32_t* acc_hist, int S, int world, int e_loc,
                       int capacity, int32_t* map_send, void* wire, int row_bytes, hipStream_t stream, int32_t* overflow) {
  M3_REQUIRE(S > 0 && world > 0 && e_loc > 0 && capacity > 0, "ep_send_map: bad sizes S=%d world=%d e_loc=%d capacity=%d", S,
             world, e_loc, capacity);
  M3_REQUIRE((row_bytes & 15) == 0 && row_bytes >= 4 * e_loc, "ep_send_map: a wire row of %d bytes cannot carry %d counts", row_bytes, e_loc);
  M3_REQUIRE(capacity >= S || overflow != nullptr, "ep_send_map: capacity %d < rows %d needs an overflow counter (a rank may send all of its rows to one peer)", capacity, S);
  const int n = S > world * e_loc ? S : world * e_loc;
  hipLaunchKernelGGL(ep_send_map_kernel, dim3(min(cdiv(n, 256), 1024)), dim3(256), 0, stream, gate_idx, mapping, acc_hist, S,
                     world, e_loc, capacity, map_send, (int32_t*)wire, row_bytes / 4, overflow);
  M3_LAUNCH_CHECK();
  return 0;
}

int launch_ep_send_rows(const int32_t* gate_idx, const int32_t* mapping, const int32_t* acc_hist, int S, int world, int e_loc,
                        int capacity, int32_t* map_send, const void* x, void* wire, int row_bytes, hipStream_t stream, int32_t* overflow) {
  M3_REQUIRE(S > 0 && world > 0 && e_loc > 0 && capacity > 0, "ep_send_rows: bad sizes S=%d world=%d e_loc=%d capacity=%d", S,
             world, e_loc, capacity);
  M3_REQUIRE((row_bytes & 15) == 0 && row_bytes >= 4 * e_loc, "ep_send_rows: a wire row of %d bytes cannot carry %d counts", row_bytes, e_loc);
  M3_REQUIRE(capacity >= S || overflow != nullptr, "ep_send_rows: capacity %d < rows %d needs an overflow counter (a rank may send all of its rows to one peer)", capacity, S);
  hipLaunchKernelGGL(ep_send_rows_kernel, dim3(min(cdiv(S, 4), 2048)), dim3(256), 0, stream, gate_idx, mapping, acc_hist, S, world,
                     e_loc, capacity, map_send, (const uint4*)x, row_bytes / 16, (uint4*)wire, overflow);
  M3_LAUNCH_CHECK();
  return 0;
}

int launch_ep_recv_gate(const void* wire, int world, int e_loc, int capacity, int row_bytes, int32_t* gate_recv,
                        hipStream_t stream) {
  M3_REQUIRE(world > 0 && e_loc > 0 && e_loc <= 1024 && capacity > 0, "ep_recv_gate: bad sizes world=%d e_loc=%d capacity=%d",
             world, e_loc, capacity);
  M3_REQUIRE((row_bytes & 15) == 0 && row_bytes >= 4 * e_loc, "ep_recv_gate: a wire row of %d bytes cannot carry %d counts", row_bytes, e_loc);
  hipLaunchKernelGGL(ep_recv_gate_kernel, dim3(world), dim3(256), 0, stream, (const int32_t*)wire, world, e_loc, capacity,
                     row_bytes / 4, gate_recv);
  M3_LAUNCH_CHECK();
  return 0;
}

}  // namespace m3
